// The W half-step inside pass A (round 4; round-3 review, next #2).
//
// Pass A leaves XHt = H_all . X^T in HBM and the W half-step (sweep_kernel) reads it back: 2 x 205 MB per 1024-column
// iteration at 50 000 cells.  A stream-K workgroup that computes a tile in ONE piece (two thirds of the 784 tiles at 1024
// columns) holds that tile's products in its accumulators: here it runs the coordinate-descent update of the tile's 256
// cells itself -- for every restart whose columns lie inside one 128-column half of the tile's component group -- and
// never writes those products.  What it does per restart is sweep_body's arithmetic, operation for operation (same
// component order, same 64-row MFMA accumulation of the Gram partial, same reductions), on the same float32 values, so the
// result is BIT-IDENTICAL to the stand-alone sweep (CNMF_FUSE_A=0; tests/test_gpu_nmf.py::test_fused_w_half_step_*):
//   * phase g = 0, 1: the four waves holding components [g * 128, g * 128 + 128) of the group write their scaled
//     accumulators into an LDS tile Pt[128 components][256 cells] (the DMA images are dead by then); barrier;
//   * every wave takes restarts (slots) of that half round-robin; ONE wave runs a restart over the tile's four 64-cell
//     chunks -- chunk c is what wave c of the stand-alone sweep's workgroup does --: W from HBM (the next chunk's rows are
//     requested before this chunk's update), products from Pt, update, W and its two f16 planes back to HBM, Gram partial
//     of the updated rows on the f32 matrix pipe; then the restart's partial for this tile (Gram, violation, row-scale
//     bound), combined over the four chunks in the stand-alone kernel's order;
//   * rows of the tile that belong to no such restart (ranks above 16, restarts straddling a 128-column boundary, empty
//     columns are skipped) are stored as before and swept by the stand-alone kernel, which skips exactly the
//     (restart, tile) pairs done here: fusable(off, k) = k <= 16 and (off % 128) + k <= 128, tile not cut.
#pragma once
#include "kernels_sweep.hip.h"

namespace cnmf {

// wave-private LDS behind the Pt tile: Gram of the other factor [16][17] | exponents [16] | strip (Gram staging 32 x 17
// floats, then the plane transposition of 8 components x 64 cells)
constexpr int FW_LIST = 1 + 128;                               // slots of one 128-column block (rank >= 1)
constexpr int FW_PT_BYTES = 128 * 256 * 4;
constexpr int FW_GS = 16 * 17;
constexpr int FW_STRIP = 32 * 17;                              // floats; >= 8 * 64 dwords
constexpr int FW_PRIV_BYTES = (FW_GS + 16 + FW_STRIP) * 4;     // 3 328 B per wave
constexpr int FW_LDS_BYTES = FW_PT_BYTES + 8 * FW_PRIV_BYTES;  // 157 696 B
static_assert(FW_STRIP >= 8 * 64, "the strip holds the plane transposition of 8 components");

struct FusedW {
    static constexpr bool enabled = true;
    int on;                                   // 0: behave like the plain store
    float* V; int ldv; int L;                 // Wt_all [KC][ldv], cells
    const float* gram;                        // H.H^T (+ l2) per slot [slot][GRAM_SZ]
    const SlotDesc* slots;
    // built by the host whenever the slot table changes (a scan of the table by every wave cost more than the sweeps):
    const int* lists;                         // [128-column block][FW_LIST]: count, then the ids of the block's fusable slots
    const unsigned* need;                     // [component group][8]: rows the stand-alone sweep still needs from HBM
    float l1;
    float* gram_part; double* viol_part; float* rmax_part;
    int n_parts, gld;                         // partials per slot (= cell tiles), leading dimension of a Gram partial
    PlaneOut po;

    // one restart, one tile, one wave
    template <int KP>
    __device__ __forceinline__ void unit(const SlotDesc& sd, int slot, int part, const float* __restrict__ Pt, int cb,
                                         float* priv) const
    {
        static_assert(KP <= 16, "ranks <= 16 only");
        const int lane = threadIdx.x & 63;
        const int k = sd.k, off = sd.off;
        float* GSw = priv;                                    // [KP][17]
        int* shl = reinterpret_cast<int*>(priv + FW_GS);      // [16]
        float* strip = priv + FW_GS + 16;
#define FGS(t_, r_) GSw[(t_) * 17 + (r_)]
        for (int e = lane; e < KP * KP; e += 64) {
            const int r = e / KP, c = e % KP;
            FGS(r, c) = (r < k && c < k) ? gram[(size_t)slot * GRAM_SZ + r * GRAM_LD + c] : 0.f;
        }
        if (lane < KP) shl[lane] = po.shift[off + min(lane, k - 1)];
        __builtin_amdgcn_wave_barrier();
        // The four 64-cell chunks of the tile are updated SIDE BY SIDE: four independent coordinate-descent chains per lane
        // (one per chunk) share every read of the Gram matrix -- a single wave has nobody to hide the latency of the
        // dependent multiply-adds and of the LDS reads behind, unlike the 16-20 waves per CU of the stand-alone sweep.
        // (Each chunk's own arithmetic is unchanged: chunk c is wave c of the stand-alone workgroup.)
        f32x4 g4[4];
        float violc[4];
        constexpr int NI = 2;                                 // chunks side by side (4: the 64-bit row addresses alone spill)
        const unsigned ldu = (unsigned)ldv;
#pragma unroll 1
        for (int c2 = 0; c2 < 4; c2 += NI) {
        float w[NI][KP], p[NI][KP];
        unsigned vo[NI];                                      // element offset of (component off, this lane's cell) in V
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int rowc = min(part * 256 + (c2 + i) * 64 + lane, L - 1);
            vo[i] = (unsigned)off * ldu + (unsigned)rowc;
#pragma unroll
            for (int c = 0; c < KP; ++c) w[i][c] = V[vo[i] + (unsigned)min(c, k - 1) * ldu];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool live = part * 256 + (c2 + i) * 64 + lane < L;
#pragma unroll
            for (int c = 0; c < KP; ++c) {
                const float pv = Pt[(unsigned)(off - cb + min(c, k - 1)) * 256u + (unsigned)((c2 + i) * 64 + lane)];
                const bool on_ = live && (c < k);
                w[i][c] = on_ ? w[i][c] : 0.f;
                p[i][c] = on_ ? (pv - l1) : 0.f;
            }
        }
        float vl[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) vl[i] = 0.f;
#pragma unroll
        for (int t = 0; t < KP; ++t) {
            if (t < k) {
                float grad[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) grad[i] = -p[i][t];
#pragma unroll
                for (int r = 0; r < KP; ++r) {
                    const float gtr = FGS(t, r);
#pragma unroll
                    for (int i = 0; i < NI; ++i) grad[i] = fmaf(gtr, w[i][r], grad[i]);
                }
                const float hess = FGS(t, t);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const bool live = part * 256 + (c2 + i) * 64 + lane < L;  // (dead rows: w = p = 0 -> grad = 0, nothing moves)
                    const float pg = (w[i][t] == 0.f) ? fminf(0.f, grad[i]) : grad[i];
                    if (live) vl[i] += fabsf(pg);
                    if (live && hess != 0.f) w[i][t] = fmaxf(w[i][t] - grad[i] / hess, 0.f);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            violc[c2 + i] = vl[i];
            if (part * 256 + (c2 + i) * 64 + lane < L) {
                const unsigned so = (unsigned)off * ldu + (unsigned)(part * 256 + (c2 + i) * 64 + lane);
#pragma unroll
                for (int c = 0; c < KP; ++c)
                    if (c < k) V[so + (unsigned)c * ldu] = w[i][c];
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int ch = c2 + i;
            // Gram of the updated rows: sweep_body's GMODE 0 sequence on this wave's strip
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            {
                const int li = lane & 15, q = lane >> 4;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if ((lane >> 5) == hf) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) strip[(lane & 31) * 17 + c] = (c < KP) ? w[i][c < KP ? c : 0] : 0.f;
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const float a = strip[(4 * s + q) * 17 + li];
                        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, g, 0, 0, 0);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            g4[ch] = g;
            // the two f16 planes of the updated rows (sweep_body PLN), eight components at a time through the strip
            {
                unsigned* tw = reinterpret_cast<unsigned*>(strip);
                const int kb0 = (part * 256 + ch * 64) >> 4;                         // first 16-cell block of the chunk
#pragma unroll
                for (int c0 = 0; c0 < KP; c0 += 8) {
                    if (c0 < k) {
#pragma unroll
                        for (int c = c0; c < c0 + 8 && c < KP; ++c) {
                            if (c < k) {
                                const float y = ldexpf(w[i][c], shl[c]);
                                unsigned short hb, mb;
                                split2h(y, hb, mb);
                                tw[(c - c0) * 64 + lane] = (unsigned)hb | ((unsigned)mb << 16);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        const int kc = min(8, k - c0);
                        {
                            const int item = lane;                                   // 8 components x 4 blocks x 2 halves = 64 items
                            const int cc = item >> 3, b = (item >> 1) & 3, hf = item & 1;
                            if (cc < kc && kb0 + b < po.Kb) {
                                const u32x4 d0 = *reinterpret_cast<const u32x4*>(tw + cc * 64 + b * 16 + hf * 8);
                                const u32x4 d1 = *reinterpret_cast<const u32x4*>(tw + cc * 64 + b * 16 + hf * 8 + 4);
                                u32x4 oh, om;
                                oh.x = (d0.x & 0xffffu) | (d0.y << 16); oh.y = (d0.z & 0xffffu) | (d0.w << 16);
                                oh.z = (d1.x & 0xffffu) | (d1.y << 16); oh.w = (d1.z & 0xffffu) | (d1.w << 16);
                                om.x = (d0.x >> 16) | (d0.y & 0xffff0000u); om.y = (d0.z >> 16) | (d0.w & 0xffff0000u);
                                om.z = (d1.x >> 16) | (d1.y & 0xffff0000u); om.w = (d1.z >> 16) | (d1.w & 0xffff0000u);
                                const int r = off + c0 + cc, tr = r / po.TR, rin = r % po.TR, swz = (rin >> 2) & 3;
                                unsigned short* gdst = po.dst + (((size_t)tr * po.Kb + (kb0 + b)) * po.TR + rin) * 32;
                                *reinterpret_cast<u32x4*>(gdst + ((0 + hf) ^ swz) * 8) = oh;
                                *reinterpret_cast<u32x4*>(gdst + ((2 + hf) ^ swz) * 8) = om;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
        }
        // ---- this restart's partial for the tile: the four chunks in the order the stand-alone workgroup adds its waves
        {
            const int li = lane & 15, q = lane >> 4;
            float* gp = gram_part + ((size_t)slot * n_parts + part) * (size_t)(gld * gld);
            float diag = 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float v = g4[0][rr] + g4[1][rr] + g4[2][rr] + g4[3][rr];
                const int r = 4 * q + rr;
                if (r < k && li < k) gp[r * gld + li] = v;
                if (r == li) diag = v;
            }
            double vr[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                double dv = (double)violc[ch];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) dv += __shfl_xor(dv, o, 64);
                vr[ch] = dv;
            }
            if (lane == 0) viol_part[(size_t)slot * n_parts + part] = vr[0] + vr[1] + vr[2] + vr[3];
            // row-scale bound of the f16 planes: sqrt of the diagonal (lane holds (r, r) when 4 q + rr == li)
            if (rmax_part && li < k && (li >> 2) == q) rmax_part[(size_t)(off + li) * n_parts + part] = sqrtf(diag) * 1.0001f;
        }
        __builtin_amdgcn_wave_barrier();
#undef FGS
    }

    // scaled accumulators of one wave group (128 components x this wave's 64 cells) -> Pt[component][cell]
    __device__ __forceinline__ void write_pt(f32x16_3 (&acc)[4][2], const float* __restrict__ rscale, float* Pt, int c0,
                                             int wn, int li, int h) const
    {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = m * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                    Pt[(size_t)rl * 256 + wn * 64 + n * 32 + li] = acc[m][n][r] * rscale[c0 + rl];
                }
    }

    // the restarts of components [cb, cb + 128), dealt round-robin to `nsweep` waves (this wave is number `me`)
    __device__ __forceinline__ void sweep_half(const float* Pt, float* priv, int part, int cb, int me, int nsweep) const
    {
        const int* lst = lists + (size_t)(cb >> 7) * FW_LIST;
        const int n = lst[0];
#pragma unroll 1
        for (int i = me; i < n; i += nsweep) {
            const int s = lst[1 + i];
            const SlotDesc sd = slots[s];
            if (!sd.active) continue;                         // converged since the list was built
            switch ((sd.k + 3) / 4) {
                case 1: unit<4>(sd, s, part, Pt, cb, priv); break;
                case 2: unit<8>(sd, s, part, Pt, cb, priv); break;
                case 3: unit<12>(sd, s, part, Pt, cb, priv); break;
                default: unit<16>(sd, s, part, Pt, cb, priv); break;
            }
        }
    }

    // NOT inlined: a real call gives the epilogue a register allocation of its own -- inlined into the GEMM kernel (231
    // registers in its MFMA loop) the allocator spilled 300+ registers around both; the accumulators cross the call through
    // the stack (128 stores + loads per lane and tile, L1 / L2 resident).
    __device__ __attribute__((noinline)) void operator()(f32x16_3 (&acc)[4][2], const float* __restrict__ rscale, float* __restrict__ C,
                                               int ldc, int m0, int j0, unsigned char* smem) const
    {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // (scalar: the branches below hold barriers)
        const int grp = wave >> 2, wn = wave & 3, li = lane & 31, h = lane >> 5;
        float* Pt = reinterpret_cast<float*>(smem);
        float* priv = reinterpret_cast<float*>(smem + FW_PT_BYTES + wave * FW_PRIV_BYTES);
        const int part = j0 / G3C_JW;
        // which of the group's 256 component rows does the STAND-ALONE sweep still need from HBM?  Rows of installed restarts
        // that are not run here (ranks above 16, restarts straddling a 128-column boundary); rows of restarts run below and
        // of empty columns are not stored at all.
        const unsigned* need_g = need + (size_t)(m0 / G3_MW) * 8;
        const int j = j0 + wn * 64 + li;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned nw = __builtin_amdgcn_readfirstlane(need_g[grp * 4 + m]);
            if (nw == 0u) continue;                           // (wave-uniform)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 4 * h + (r & 3) + 8 * (r >> 2);
                    if ((nw >> rl) & 1u) {
                        const int row = m0 + grp * 128 + m * 32 + rl;
                        C[(size_t)row * ldc + j + n * 32] = acc[m][n][r] * rscale[row];
                    }
                }
        }
        G3_WAIT_VM(0);
        __syncthreads();                                      // every wave has left the MFMA loop: the images are free
        // The two halves of the component group go through Pt one after the other.  The code is split by wave group so that
        // no wave carries live accumulators through a sweep (128 registers on top of the sweep's own spilled):
        //   group 0 : write | S1 | sweep half 0 (4 waves) | S2 |               | S3 | sweep half 1 (8 waves) | S4
        //   group 1 :       | S1 |        (idle)          | S2 | write         | S3 | sweep half 1 (8 waves) | S4
        if (grp == 0) {
            write_pt(acc, rscale, Pt, m0, wn, li, h);
            __syncthreads();
            sweep_half(Pt, priv, part, m0, wn, 4);
            __syncthreads();
        } else {
            __syncthreads();
            __syncthreads();
            write_pt(acc, rscale, Pt, m0 + 128, wn, li, h);
        }
        __syncthreads();
        sweep_half(Pt, priv, part, m0 + 128, wave, 8);
        __syncthreads();
    }
};

}  // namespace cnmf
