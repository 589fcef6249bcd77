// Coordinate-descent sweep + bookkeeping kernels of the batched NMF engine.
//
// sweep_kernel restates sklearn's `_update_cdnmf_fast`
// (sklearn/decomposition/_cdnmf_fast.pyx:8-38) with one LANE per row: rows are
// independent inside the sweep, components are visited in order t = 0..k-1 and
// every update sees the already-updated components r < t, exactly as the Cython
// loop.  All factors are stored component-major ([KC][L], L = cells or genes) so
// that a wave's 64 lanes read 64 consecutive rows of one component (coalesced).
//
// The same kernel serves both half-steps (sklearn _nmf.py:500 and :505):
//   W half-step : V = Wt_all [KC][N],  P = X.Ht products [1][KC][N],   gram = HHt (+l2_reg_W)
//   H half-step : V = H_all  [KC][G],  P = Xt.W split-K partials [S][KC][G], gram = WtW (+l2_reg_H)
// and additionally produces, on the otherwise idle matrix pipe, the Gram matrix of the
// UPDATED rows (V_slot . V_slot^T, needed by the next half-step, sklearn _nmf.py:386) and
// the projected-gradient violation, both as per-workgroup partials that
// finalize_kernel reduces in a fixed order (deterministic, no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_gemm.hip.h"
#include "kernels_gemm2h.hip.h"

namespace cnmf {

constexpr int KMAX = 128;           // largest rank: <= 64 in the register-resident sweep_kernel, 65..128 in sweep_big_kernel
constexpr int KSMALL = 64;          // largest rank of sweep_kernel (tiers 0..2)
constexpr int GRAM_LD = KMAX;       // final gram matrices are stored [slot][64][64]
constexpr int GRAM_SZ = KMAX * KMAX;

// dynamic LDS of sweep_kernel for a batch whose largest rank is kmax:
//   Gs [KG][KG+4] | vred [4] doubles | rmx [4][64] | Ws [4][64][wstride]      (KG = 16 / 32 / 64)
static inline int sweep_kg(int kmax) { return kmax <= 16 ? 16 : (kmax <= 32 ? 32 : 64); }      // (ranks > 64: sweep_big_kernel)
static inline int sweep_wstride(int kmax) { return sweep_kg(kmax) + 1; }
static inline size_t sweep_lds_bytes(int kmax, bool planes = false)
{
    const int kg = sweep_kg(kmax);
    if (planes)         // + the transposition strips [4 waves][kg][64] dwords and kg exponents (PLN)
        return sweep_lds_bytes(kmax, false) + sizeof(float) * (size_t)(4 * kg * 64 + kg);
    // ranks <= 16 stage the Gram operand 32 rows at a time: 11 KB per workgroup, so that a sweep workgroup fits beside
    // a GEMM workgroup (144 KB of the 160 KB) when two batches share the GPU
    return sizeof(float) * (size_t)(kg * (kg + 4) + 8 + 256 + 4 * (kg == 16 ? 32 : 64) * sweep_wstride(kmax));
}

struct SlotDesc {                   // one restart in flight (device + host mirror)
    int off;                        // first packed component column
    int k;                          // rank
    int active;                     // 1 while iterating, 0 once converged / empty
    int iter;                       // completed outer iterations
    int restart;                    // index into the caller's restart list (-1 = empty)
    int pad_;
    double viol;                    // violation accumulated in the current outer iteration
    double viol_init;               // violation of outer iteration 1
    double viol_last;               // violation/viol_init at the last completed iteration
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SplitInfo {                  // further partial planes of a stream-K product (or plane1 == nullptr)
    const float* plane1;
    const unsigned char* split;     // [tiles] bit 0 = the tile was cut: add plane 1; bit 1 = cut twice: add plane 2 too
    int tile_rows, tile_cols, mgroups;
    const float* plane2 = nullptr;
};

// PLN (round 3, the W half-step of the f16 paths): the sweep also writes the two f16 planes of the rows it has just
// updated, in the block-major layout pass B multiplies (kernels_gemm2h.hip.h), scaled by the per-component exponent
// `shift[component]` chosen one iteration earlier (split2h_tiled_body's fused mode checks it afterwards and re-converts the
// rare row whose exponent was off).  A wave holds 64 consecutive cells of k components in registers; the layout wants, per
// component and 16-cell block, 8 cells of one plane per 16-byte slot: each lane packs (h | m << 16) of one cell, the wave
// transposes through a private LDS strip [component][64 cells], and every lane then gathers 8 cells of one (component,
// block, half) and stores the h slot and the m slot.  Saves the plane split's re-read of the factor (205 MB per 1024
// columns at 50 000 cells) and its launch.
struct PlaneOut {
    unsigned short* dst;            // planes of the packed factor (nullptr: none)
    const int* shift;               // [KC] exponent per component row
    int Kb, TR;                     // 16-cell blocks per row (cells_pad / 16), row-tile height (256)
};

// Body of the sweep for one (row chunk, slot); KP = k rounded up (compile-time register array
// size: multiples of 4 up to 32, then 48 and 64).  Gram of the updated rows on the matrix pipe:
//   KP <= 16 : v_mfma_f32_16x16x4_f32  (16 MFMAs of 32 cycles per 64 rows)
//   KP <= 32 : v_mfma_f32_32x32x2_f32  (32 MFMAs of 64 cycles per 64 rows)
//   KP <= 64 : 2 x 2 tiles of 32x32x2
// Partials are written compactly: entry (r,c) at r*gld + c, gld = largest rank of the batch.
// RMX: also report the largest updated entry per component (x rmax_scale[row]) EXACTLY, at the price of KP
// registers (the H half-step: few rows).  Without it (the W half-step: keeps 5 waves per SIMD) the report is the
// bound  sqrt(sum_rows w^2)  from the diagonal of the workgroup's Gram partial -- at most sqrt(rows per workgroup)
// = 32 x the true maximum, which the f16 plane split tolerates (kernels_gemm2h.hip.h: 5 bits of exponent slack only
// move the threshold below which tiny entries keep an absolute rather than a relative accuracy).
// PSUM: the products arrive as `sp.mgroups` split-K partial planes (stride sp.tile_rows * 2^20 + sp.tile_cols floats,
// see psum_info) that are summed here in split order and scaled by the per-row constant sp.split (reinterpreted as
// const double*) -- the work of reduce_splits_kernel folded into the H half-step (no extra launch, no extra pass).
#ifndef CNMF_SWEEP_PF
#define CNMF_SWEEP_PF 0
#endif
template <int KP, bool RMX, bool PSUM = false, bool PLN = false>
__device__ __forceinline__ void sweep_body(
    float* __restrict__ V, int ldv, int L, const float* __restrict__ P, const SplitInfo& sp,
    const float* __restrict__ gram, const SlotDesc& sd, int slot, float l1_reg,
    float* __restrict__ gram_part, double* __restrict__ viol_part,
    int chunks_per_block, int want_gram, float* lds, int kg, int gld,
    float* __restrict__ rmax_part, const double* __restrict__ rmax_scale, const PlaneOut& po = PlaneOut{nullptr, nullptr, 0, 0})
{
    constexpr int GMODE = (KP <= 16) ? 0 : ((KP <= 32) ? 1 : 2);
    constexpr int GR = (GMODE == 0) ? 16 : ((GMODE == 1) ? 32 : 64);       // gram tile edge
    const int gs = kg + 4, wstride = kg + 1;
    float* Gsb = lds;                                                     // [kg][kg+4]
    double* vred = reinterpret_cast<double*>(lds + kg * gs);
    float* rmx = lds + kg * gs + 8;                                       // [4][64] per-wave row maxima
    float* Wsb = rmx + 256;
    // PLN: [4 waves][kg components][64 cells] dwords + kg exponents, behind the Gram staging (sweep_lds_bytes(kmax, true))
    unsigned* Tpl = reinterpret_cast<unsigned*>(Wsb + 4 * (kg == 16 ? 32 : 64) * (kg + 1));
    int* shl = reinterpret_cast<int*>(Tpl + 4 * kg * 64);
#define GS(t_, r_) Gsb[(t_) * gs + (r_)]
    constexpr int SR = (GMODE == 0) ? 32 : 64;                            // rows of a wave staged at a time
#define WS(wv_, r_, c_) Wsb[((wv_) * SR + (r_)) * wstride + (c_)]
    const int k = sd.k, off = sd.off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < KP * KP; e += 256) {
        const int r = e / KP, c = e % KP;
        GS(r, c) = (r < k && c < k) ? gram[(size_t)slot * GRAM_SZ + r * GRAM_LD + c] : 0.f;
    }
    if constexpr (PLN) { if (tid < KP) shl[tid] = po.shift[off + min(tid, k - 1)]; }
    __syncthreads();

    f32x16 gacc[GMODE == 2 ? 4 : 1];
    f32x4 gacc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < (GMODE == 2 ? 4 : 1); ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[a][r] = 0.f;
    float viol = 0.f;
    // largest updated entry per component over this workgroup's rows (x the per-row scale of the f16 plane split,
    // kernels_gemm2h.hip.h) -- only when the caller wants it
    float mx[RMX ? KP : 1];
#pragma unroll
    for (int c = 0; c < (RMX ? KP : 1); ++c) mx[c] = 0.f;

    // Software prefetch (round 6 experiment, -DCNMF_SWEEP_PF=1; W half-step only; NOT adopted): the factor and product values
    // of chunk ch + 1 are requested before chunk ch is computed.  Same arithmetic, same bits; 2 KP more registers = 128 + 20
    // spilled and 4 instead of 5 waves per SIMD: 217.6 against 228.9 restarts/s (profiles/r6_sweep_prefetch_ab.txt).
    constexpr bool PF = (CNMF_SWEEP_PF != 0) && !PSUM;
    float wn[PF ? KP : 1], pn[PF ? KP : 1];
    if constexpr (PF) {
        const int rowc0 = min((int)(blockIdx.x * chunks_per_block) * 256 + tid, L - 1);
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            const size_t idx = (size_t)(off + min(c, k - 1)) * ldv + rowc0;
            wn[c] = V[idx];
            pn[c] = P[idx];
        }
    }
    for (int ch = 0; ch < chunks_per_block; ++ch) {
        const int row = (blockIdx.x * chunks_per_block + ch) * 256 + tid;
        const bool live = row < L;
        // stream-K pass A: was this (row tile, component group) cut between two workgroups?
        // A slot spans at most two component groups and a wave's 64 rows lie in one row tile,
        // so two wave-uniform flags cover every element.
        int cut0 = 0, cut1 = 0;
        int mg_edge = 1 << 30;
        if (!PSUM && sp.plane1) {
            const int rt = __builtin_amdgcn_readfirstlane(min(row, L - 1) / sp.tile_rows);
            const int g0 = off / sp.tile_cols, g1 = (off + k - 1) / sp.tile_cols;
            mg_edge = (g0 + 1) * sp.tile_cols;
            cut0 = sp.split[rt * sp.mgroups + g0];
            cut1 = sp.split[rt * sp.mgroups + g1];
        }
        float w[KP], p[KP];
        float dsc = 1.0f;                          // per-row scale of the exact row-maximum report (issued with the loads)
        if constexpr (RMX) { if (rmax_scale) dsc = (float)rmax_scale[min(row, L - 1)]; }
        {
            // branch-free load phase: every lane issues all 2*KP (3*KP) loads back to back with
            // clamped (always valid) addresses; the values of dead lanes / columns >= k are
            // discarded by selects.  (Per-column branches serialise the memory latency.)
            const int rowc = min(row, L - 1);
            if constexpr (PF) {
#pragma unroll
                for (int c = 0; c < KP; ++c) { w[c] = wn[c]; p[c] = pn[c]; }
                if (ch + 1 < chunks_per_block) {                 // (uniform over the workgroup)
                    const int rown = min(row + 256, L - 1);
#pragma unroll
                    for (int c = 0; c < KP; ++c) {
                        const size_t idx = (size_t)(off + min(c, k - 1)) * ldv + rown;
                        wn[c] = V[idx];
                        pn[c] = P[idx];
                    }
                }
            } else {
#pragma unroll
            for (int c = 0; c < KP; ++c) {
                const size_t idx = (size_t)(off + min(c, k - 1)) * ldv + rowc;
                w[c] = V[idx];
                p[c] = P[idx];
            }
            }
            if constexpr (PSUM) {
                const int nsplit = sp.mgroups;
                const size_t stride = (size_t)sp.tile_rows * (1u << 20) + (size_t)sp.tile_cols;
                // U planes' loads in flight at a time; the additions keep the split order (bit-identical to
                // reduce_splits_kernel)
                // the remainder of the split count is batched too (clamped plane index, wave-uniform guard on the
                // addition): with 8 planes (a 1024-column batch) the former `s + U <= nsplit` loop never took a full
                // batch and added all seven planes one memory round trip after the other
                constexpr int U = KP <= 16 ? 8 : (KP <= 32 ? 2 : 1);
                for (int s = 1; s < nsplit; s += U) {
                    float q[U][KP];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int ss = min(s + u, nsplit - 1);
#pragma unroll
                        for (int c = 0; c < KP; ++c)
                            q[u][c] = P[ss * stride + (size_t)(off + min(c, k - 1)) * ldv + rowc];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (s + u < nsplit) {
#pragma unroll
                            for (int c = 0; c < KP; ++c) p[c] += q[u][c];
                        }
                }
                const double* colscale = reinterpret_cast<const double*>(sp.split);
                if (colscale) {
                    const double cs = colscale[rowc];
#pragma unroll
                    for (int c = 0; c < KP; ++c) p[c] = (float)((double)p[c] * cs);
                }
            }
            if ((cut0 | cut1) & 1) {              // wave-uniform: one branch per chunk
                float qq[KP];
#pragma unroll
                for (int c = 0; c < KP; ++c) {
                    const size_t idx = (size_t)(off + min(c, k - 1)) * ldv + rowc;
                    qq[c] = sp.plane1[idx];
                }
#pragma unroll
                for (int c = 0; c < KP; ++c) {
                    const int cut = ((off + c) < mg_edge) ? cut0 : cut1;
                    p[c] += (cut & 1) ? qq[c] : 0.f;
                }
            }
            if ((cut0 | cut1) & 2) {
                float qq[KP];
#pragma unroll
                for (int c = 0; c < KP; ++c) {
                    const size_t idx = (size_t)(off + min(c, k - 1)) * ldv + rowc;
                    qq[c] = sp.plane2[idx];
                }
#pragma unroll
                for (int c = 0; c < KP; ++c) {
                    const int cut = ((off + c) < mg_edge) ? cut0 : cut1;
                    p[c] += (cut & 2) ? qq[c] : 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < KP; ++c) {
                const bool on = live && (c < k);
                w[c] = on ? w[c] : 0.f;
                p[c] = on ? (p[c] - l1_reg) : 0.f;
            }
        }
        if (live) {
#pragma unroll
            for (int t = 0; t < KP; ++t) {
                if (t < k) {
                    float grad = -p[t];
#pragma unroll
                    for (int r = 0; r < KP; ++r) grad = fmaf(GS(t, r), w[r], grad);
                    const float pg = (w[t] == 0.f) ? fminf(0.f, grad) : grad;
                    viol += fabsf(pg);
                    const float hess = GS(t, t);
                    if (hess != 0.f) w[t] = fmaxf(w[t] - grad / hess, 0.f);
                }
            }
#pragma unroll
            for (int c = 0; c < KP; ++c)
                if (c < k) V[(size_t)(off + c) * ldv + row] = w[c];
            if constexpr (RMX) {
#pragma unroll
                for (int c = 0; c < KP; ++c) mx[c] = fmaxf(mx[c], w[c] * dsc);
            }
        }
        if (want_gram) {
            // Gram of the updated rows on the (otherwise idle) matrix pipe: acc += Wrows^T . Wrows
            if constexpr (GMODE == 0) {
                const int li = lane & 15, q = lane >> 4;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {   // rows 0..31, then 32..63 of the wave (same order of the additions)
                    if ((lane >> 5) == hf) {
#pragma unroll
                        for (int c = 0; c < GR; ++c) WS(wave, lane & 31, c) = (c < KP) ? w[c < KP ? c : 0] : 0.f;
                    }
                    __builtin_amdgcn_wave_barrier();   // wave-private tile: LDS ops of one wave are in order
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const float a = WS(wave, 4 * s + q, li);
                        gacc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, gacc4, 0, 0, 0);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else if constexpr (GMODE == 1) {
#pragma unroll
                for (int c = 0; c < GR; ++c) WS(wave, lane, c) = (c < KP) ? w[c < KP ? c : 0] : 0.f;
                __builtin_amdgcn_wave_barrier();
                const int li = lane & 31, h = lane >> 5;
#pragma unroll 8
                for (int s = 0; s < 32; ++s) {
                    const float a = WS(wave, 2 * s + h, li);
                    gacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, gacc[0], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < GR; ++c) WS(wave, lane, c) = (c < KP) ? w[c < KP ? c : 0] : 0.f;
                __builtin_amdgcn_wave_barrier();
                const int li = lane & 31, h = lane >> 5;
#pragma unroll 4
                for (int s = 0; s < 32; ++s) {
                    const float a0 = WS(wave, 2 * s + h, li), a1 = WS(wave, 2 * s + h, 32 + li);
                    gacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, gacc[0], 0, 0, 0);
                    gacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, gacc[1], 0, 0, 0);
                    gacc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a0, gacc[2], 0, 0, 0);
                    gacc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, gacc[3], 0, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (PLN) {
            // (h | m << 16) of this lane's cell for every component -> the wave's strip
            unsigned* tw = Tpl + wave * (kg * 64);
#pragma unroll
            for (int c = 0; c < KP; ++c) {
                if (c < k) {
                    const float y = ldexpf(w[c], shl[c]);
                    unsigned short hb, mb;
                    split2h(y, hb, mb);
                    tw[c * 64 + lane] = (unsigned)hb | ((unsigned)mb << 16);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int kb0 = ((blockIdx.x * chunks_per_block + ch) * 256 + wave * 64) >> 4;       // first 16-cell block of the wave
            for (int item = lane; item < 8 * k; item += 64) {
                const int c = item >> 3, b = (item >> 1) & 3, hf = item & 1;
                if (kb0 + b >= po.Kb) continue;                                  // row blocks past the padded length
                const u32x4 d0 = *reinterpret_cast<const u32x4*>(tw + c * 64 + b * 16 + hf * 8);
                const u32x4 d1 = *reinterpret_cast<const u32x4*>(tw + c * 64 + b * 16 + hf * 8 + 4);
                u32x4 oh, om;
                oh.x = (d0.x & 0xffffu) | (d0.y << 16); oh.y = (d0.z & 0xffffu) | (d0.w << 16);
                oh.z = (d1.x & 0xffffu) | (d1.y << 16); oh.w = (d1.z & 0xffffu) | (d1.w << 16);
                om.x = (d0.x >> 16) | (d0.y & 0xffff0000u); om.y = (d0.z >> 16) | (d0.w & 0xffff0000u);
                om.z = (d1.x >> 16) | (d1.y & 0xffff0000u); om.w = (d1.z >> 16) | (d1.w & 0xffff0000u);
                const int r = off + c, tr = r / po.TR, rin = r % po.TR, swz = (rin >> 2) & 3;
                unsigned short* g = po.dst + (((size_t)tr * po.Kb + (kb0 + b)) * po.TR + rin) * 32;
                *reinterpret_cast<u32x4*>(g + ((0 + hf) ^ swz) * 8) = oh;
                *reinterpret_cast<u32x4*>(g + ((2 + hf) ^ swz) * 8) = om;
            }
            __builtin_amdgcn_wave_barrier();                                     // the strip is rewritten by the next chunk
        }
    }

    // ---- row maxima: wave reduce -> LDS -> one partial per (component, workgroup)
    if constexpr (RMX) {
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            float v = mx[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
            if (lane == 0) rmx[wave * 64 + c] = v;
        }
    }
    // ---- violation: wave reduce (double) -> block partial
    double dv = (double)viol;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dv += __shfl_xor(dv, o, 64);
    if (lane == 0) vred[wave] = dv;

    // ---- gram: sum the 4 waves' accumulators through LDS, write the block partial
    __syncthreads();
    float* gred = Wsb;                   // reused as [4][GR][GR+1]   (<= 4*64*(kg+1) floats)
    if (want_gram) {
        if constexpr (GMODE == 0) {
            const int li = lane & 15, q = lane >> 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) gred[(wave * GR + 4 * q + r) * (GR + 1) + li] = gacc4[r];
        } else {
            const int li = lane & 31, h = lane >> 5;
#pragma unroll
            for (int a = 0; a < (GMODE == 2 ? 4 : 1); ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (a >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    gred[(wave * GR + rr) * (GR + 1) + (a & 1) * 32 + li] = gacc[a][r];
                }
        }
    }
    __syncthreads();
    if (want_gram) {
        float* gp = gram_part + ((size_t)slot * gridDim.x + blockIdx.x) * (size_t)(gld * gld);
        for (int e = tid; e < k * k; e += 256) {
            const int r = e / k, c = e % k;
            gp[r * gld + c] = gred[(0 * GR + r) * (GR + 1) + c] + gred[(1 * GR + r) * (GR + 1) + c] +
                              gred[(2 * GR + r) * (GR + 1) + c] + gred[(3 * GR + r) * (GR + 1) + c];
        }
    }
    if (tid == 0)
        viol_part[(size_t)slot * gridDim.x + blockIdx.x] = vred[0] + vred[1] + vred[2] + vred[3];
    if (rmax_part && tid < k && (RMX || want_gram)) {
        float v;
        if constexpr (RMX) v = fmaxf(fmaxf(rmx[tid], rmx[64 + tid]), fmaxf(rmx[128 + tid], rmx[192 + tid]));
        else v = sqrtf(gred[(0 * GR + tid) * (GR + 1) + tid] + gred[(1 * GR + tid) * (GR + 1) + tid] +
                       gred[(2 * GR + tid) * (GR + 1) + tid] + gred[(3 * GR + tid) * (GR + 1) + tid]) * 1.0001f;
        rmax_part[(size_t)(off + tid) * gridDim.x + blockIdx.x] = v;
    }
#undef WS
#undef GS
}

// One launch sweeps every slot in flight: grid = (row blocks, slots); the workgroup
// dispatches on its slot's rank to the right register-array size.
// TIER 0 handles ranks <= 16, TIER 1 ranks 17..32, TIER 2 ranks 33..64: one kernel for all ranks would
// give the common small-rank case the register allocation of the largest (232 VGPR + 64 AGPR = one
// wave per SIMD).  The host launches only the tiers present in the batch.
// (TIER 0 without the exact report is the W half-step of the common ranks: held to 5 waves per SIMD)
template <int TIER, bool RMX = false, bool PSUM = false, bool PLN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((TIER == 0 && !RMX && !PLN) ? 5 : (TIER == 0 && !RMX ? 4 : 1), 8))) void sweep_kernel(
    float* __restrict__ V, int ldv, int L,
    const float* __restrict__ P,             // [KC][ldv] products (split-K already reduced)
    SplitInfo sp,
    const float* __restrict__ gram,          // [nslots][32][32], regularised diagonal included
    const SlotDesc* __restrict__ slots,
    float l1_reg,
    float* __restrict__ gram_part,           // [nslots][gridDim.x][32][32]
    double* __restrict__ viol_part,          // [nslots][gridDim.x]
    int chunks_per_block, int want_gram, int kg, int gld,
    float* __restrict__ rmax_part = nullptr,     // [KC][gridDim.x] largest updated entry per component and workgroup
    const double* __restrict__ rmax_scale = nullptr,
    PlaneOut po = PlaneOut{nullptr, nullptr, 0, 0})
{
    const int slot = blockIdx.y;
    const SlotDesc sd = slots[slot];
    if (!sd.active) return;
    extern __shared__ __attribute__((aligned(16))) float sweep_lds[];
#define CNMF_SW(KP_)                                                                              \
        sweep_body<KP_, RMX, PSUM, PLN>(V, ldv, L, P, sp, gram, sd, slot, l1_reg, gram_part, viol_part, \
                        chunks_per_block, want_gram, sweep_lds, kg, gld, rmax_part, rmax_scale, po);
    const int k = sd.k;
    if (TIER == 0) {
        if (k > 16) return;
        switch ((k + 3) / 4) {
            case 1: CNMF_SW(4) break;   case 2: CNMF_SW(8) break;   case 3: CNMF_SW(12) break;
            case 4: CNMF_SW(16) break;  default: break;
        }
    } else if (TIER == 1) {
        if (k <= 16 || k > 32) return;
        switch ((k + 3) / 4) {
            case 5: CNMF_SW(20) break;  case 6: CNMF_SW(24) break;
            case 7: CNMF_SW(28) break;  case 8: CNMF_SW(32) break;  default: break;
        }
    } else {
        if (k <= 32) return;
        if (k <= 48) { CNMF_SW(48) } else if (k <= 64) { CNMF_SW(64) }
    }
#undef CNMF_SW
}

// ------------------------------------------------------------------------------------------
// Ranks 65 .. 128 (tier 3).  The register-resident body above unrolls k x k multiply-adds and holds w, p (and the Gram
// accumulators of the matrix pipe) in registers -- at rank 128 that is 16 384 unrolled FMAs and > 512 registers.  Here:
//   * one lane per row as before, the row's w[128] in registers; the component loop runs over blocks of 32 components
//     (unrolled 4 x) with a RUNTIME loop inside a block: the dot product  grad = -p_t + sum_r G[t][r] w[r]  reads all
//     128 registers statically (same summation order r = 0 .. k-1 as _update_cdnmf_fast: the terms r >= k are + 0), the
//     one dynamically indexed register write w[t] = new value is a 32-way select chain on the wave-uniform index;
//   * p_t and the running w_t of the current block sit in a per-lane LDS strip (stride 33: conflict-free), G [128][132]
//     in LDS is read with broadcast 16-byte loads;
//   * the Gram matrix of the updated rows is a separate launch (gram_big_kernel, matrix pipe, re-reads the rows) and the
//     split-K partial planes of pass B are reduced by reduce_splits_kernel first (no PSUM) -- ranks this large are the
//     rare case; what matters is that they run, correctly, next to the small ones in the same batch.
// RMX: exact per-component maxima (x the per-row scale) for the f16 plane split of the H half-step.
constexpr int KBIG = 128;
constexpr int KBIG_GS = KBIG + 4;
static inline size_t sweep_big_lds_bytes() { return sizeof(float) * (size_t)(KBIG * KBIG_GS + 2 * 256 * 33 + 4 * KBIG + 16); }

template <bool RMX>
__global__ __launch_bounds__(256) void sweep_big_kernel(
    float* __restrict__ V, int ldv, int L, const float* __restrict__ P, SplitInfo sp, const float* __restrict__ gram,
    const SlotDesc* __restrict__ slots, float l1_reg, double* __restrict__ viol_part, int chunks_per_block,
    float* __restrict__ rmax_part, const double* __restrict__ rmax_scale)
{
    const int slot = blockIdx.y;
    const SlotDesc sd = slots[slot];
    if (!sd.active || sd.k <= KSMALL) return;
    extern __shared__ __attribute__((aligned(16))) float sweep_lds[];
    float* Gs = sweep_lds;                                  // [128][132]
    float* Pl = Gs + KBIG * KBIG_GS;                        // [256][33]  p_t - l1 of the current 32-block, per lane
    float* Wl = Pl + 256 * 33;                              // [256][33]  running w_t of the current 32-block, per lane
    float* rmx = Wl + 256 * 33;                             // [4][128]
    double* vred = reinterpret_cast<double*>(rmx + 4 * KBIG);
    const int k = sd.k, off = sd.off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < KBIG * KBIG; e += 256) {
        const int r = e / KBIG, c = e % KBIG;
        Gs[r * KBIG_GS + c] = (r < k && c < k) ? gram[(size_t)slot * GRAM_SZ + r * GRAM_LD + c] : 0.f;
    }
    if (RMX) for (int e = tid; e < 4 * KBIG; e += 256) rmx[e] = 0.f;
    __syncthreads();
    float viol = 0.f;
    float* pl = Pl + tid * 33;
    float* wl = Wl + tid * 33;
    for (int ch = 0; ch < chunks_per_block; ++ch) {
        const int row = (blockIdx.x * chunks_per_block + ch) * 256 + tid;
        const bool live = row < L;
        const int rowc = min(row, L - 1);
        int cut0 = 0, cut1 = 0, mg_edge = 1 << 30;
        if (sp.plane1) {                                    // stream-K pass A: cut flags of this wave's row tile
            const int rt = __builtin_amdgcn_readfirstlane(rowc / sp.tile_rows);
            const int g0 = off / sp.tile_cols, g1 = (off + k - 1) / sp.tile_cols;
            mg_edge = (g0 + 1) * sp.tile_cols;
            cut0 = sp.split[rt * sp.mgroups + g0];
            cut1 = sp.split[rt * sp.mgroups + g1];
        }
        float w[KBIG];
#pragma unroll
        for (int c = 0; c < KBIG; ++c) {
            const float v = V[(size_t)(off + min(c, k - 1)) * ldv + rowc];
            w[c] = (live && c < k) ? v : 0.f;
        }
        float dsc = 1.0f;
        if (RMX && rmax_scale) dsc = (float)rmax_scale[rowc];
#pragma unroll
        for (int tb = 0; tb < KBIG / 32; ++tb) {
            if (tb * 32 < k) {                              // wave-uniform
#pragma unroll
                for (int rr = 0; rr < 32; ++rr) {
                    const int c = tb * 32 + rr;
                    const size_t idx = (size_t)(off + min(c, k - 1)) * ldv + rowc;
                    float pv = P[idx];
                    const int cut = ((off + c) < mg_edge) ? cut0 : cut1;
                    if (cut & 1) pv += sp.plane1[idx];      // (wave-uniform per component)
                    if (cut & 2) pv += sp.plane2[idx];
                    pl[rr] = (live && c < k) ? pv - l1_reg : 0.f;
                    wl[rr] = w[c];
                }
                __builtin_amdgcn_wave_barrier();            // per-lane strips: LDS operations of one wave are in order
                const int tend = min(32, k - tb * 32);
                for (int tt = 0; tt < tend; ++tt) {
                    const int t = tb * 32 + tt;
                    const float4* g4 = reinterpret_cast<const float4*>(Gs + t * KBIG_GS);
                    float grad = -pl[tt];
#pragma unroll
                    for (int r4 = 0; r4 < KBIG / 4; ++r4) {
                        const float4 g = g4[r4];
                        grad = fmaf(g.x, w[4 * r4 + 0], grad);
                        grad = fmaf(g.y, w[4 * r4 + 1], grad);
                        grad = fmaf(g.z, w[4 * r4 + 2], grad);
                        grad = fmaf(g.w, w[4 * r4 + 3], grad);
                    }
                    const float wt = wl[tt];
                    const float pg = (wt == 0.f) ? fminf(0.f, grad) : grad;
                    viol += live ? fabsf(pg) : 0.f;
                    const float hess = Gs[t * KBIG_GS + t];
                    float wn = wt;
                    if (hess != 0.f) wn = fmaxf(wt - grad / hess, 0.f);
#pragma unroll
                    for (int rr = 0; rr < 32; ++rr) w[tb * 32 + rr] = (rr == tt) ? wn : w[tb * 32 + rr];
                }
            }
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < KBIG; ++c)
                if (c < k) V[(size_t)(off + c) * ldv + row] = w[c];
        }
        if (RMX) {
#pragma unroll
            for (int c = 0; c < KBIG; ++c) {
                if (c < k) {                                // wave-uniform
                    float v = w[c] * dsc;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
                    if (lane == 0) rmx[wave * KBIG + c] = fmaxf(rmx[wave * KBIG + c], v);
                }
            }
        }
    }
    double dv = (double)viol;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dv += __shfl_xor(dv, o, 64);
    if (lane == 0) vred[wave] = dv;
    __syncthreads();
    if (tid == 0) viol_part[(size_t)slot * gridDim.x + blockIdx.x] = vred[0] + vred[1] + vred[2] + vred[3];
    if (RMX && rmax_part && tid < k)
        rmax_part[(size_t)(off + tid) * gridDim.x + blockIdx.x] =
            fmaxf(fmaxf(rmx[tid], rmx[KBIG + tid]), fmaxf(rmx[2 * KBIG + tid], rmx[3 * KBIG + tid]));
}

// Gram partial of the rows of one (row block, slot) for ranks 65..128: gram_part[slot][block] = V_rows^T . V_rows on the
// exact-f32 matrix pipe, 32 rows staged through LDS per step, each wave a 64 x 64 quadrant (2 x 2 tiles of 32 x 32).
// Also (rmax_part != nullptr) the row-scale bound of the f16 plane split, sqrt of the diagonal (see sweep_body).
__global__ __launch_bounds__(256) void gram_big_kernel(const float* __restrict__ V, int ldv, int L,
                                                       const SlotDesc* __restrict__ slots, float* __restrict__ gram_part,
                                                       int chunks_per_block, int gld, float* __restrict__ rmax_part)
{
    const int slot = blockIdx.y;
    const SlotDesc sd = slots[slot];
    if (!sd.active || sd.k <= KSMALL) return;
    __shared__ float T[32][KBIG_GS];
    __shared__ float diag[KBIG];
    const int k = sd.k, off = sd.off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int r_begin = blockIdx.x * chunks_per_block * 256, r_end = min(L, r_begin + chunks_per_block * 256);
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i, c = e >> 5, rr = e & 31;
            T[rr][c] = (r0 + rr < r_end && c < k) ? V[(size_t)(off + c) * ldv + r0 + rr] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int s2 = 0; s2 < 16; ++s2) {
            const float a0 = T[2 * s2 + h][wr * 64 + li], a1 = T[2 * s2 + h][wr * 64 + 32 + li];
            const float b0 = T[2 * s2 + h][wc * 64 + li], b1 = T[2 * s2 + h][wc * 64 + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    float* gp = gram_part + ((size_t)slot * gridDim.x + blockIdx.x) * (size_t)(gld * gld);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, gc = wc * 64 + j * 32 + li;
                if (gr < k && gc < k) gp[gr * gld + gc] = acc[i][j][r];
                if (gr == gc && gr < KBIG) diag[gr] = acc[i][j][r];
            }
    __syncthreads();
    if (rmax_part && tid < k) rmax_part[(size_t)(off + tid) * gridDim.x + blockIdx.x] = sqrtf(diag[tid]) * 1.0001f;
}

// Sum the split-K partials of pass B in split order (deterministic): out = sum_s P[s].
// float4 grid-stride; rows of inactive / unused component columns are skipped via `rows`.
__global__ __launch_bounds__(256) void reduce_splits_kernel(
    const float* __restrict__ P, int nsplit, long long split_stride, float* __restrict__ out,
    long long n_vec4, const double* __restrict__ colscale = nullptr, int ld = 1)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_vec4) return;
    const v4f* p = reinterpret_cast<const v4f*>(P) + i;
    const long long sv = split_stride / 4;
    v4f acc = p[0];
    int s = 1;
    for (; s + 4 <= nsplit; s += 4) {
        const v4f a = p[(long long)s * sv], b = p[(long long)(s + 1) * sv];
        const v4f c = p[(long long)(s + 2) * sv], d = p[(long long)(s + 3) * sv];
        acc += a; acc += b; acc += c; acc += d;
    }
    for (; s < nsplit; ++s) acc += p[(long long)s * sv];
    if (colscale) {                                       // count-structured data: X^T.W = d * (n^T.W)
        const int c0 = (int)((i * 4) % ld);               // ld is a multiple of 4: the four lanes share a row
        acc.x = (float)((double)acc.x * colscale[c0]);     acc.y = (float)((double)acc.y * colscale[c0 + 1]);
        acc.z = (float)((double)acc.z * colscale[c0 + 2]); acc.w = (float)((double)acc.w * colscale[c0 + 3]);
    }
    reinterpret_cast<v4f*>(out)[i] = acc;
}

// grid = (slots, ceil(kmax^2/256)): reduce the sweep's partials in a fixed order, add the l2 regulariser to
// the diagonal (sklearn _nmf.py:389-392); block (slot,0) also runs the stopping rule of
// _fit_coordinate_descent (sklearn _nmf.py:496-521) on the device.
//   phase 0 : after the W half-step (update_H=True)  -> store violation, no decision
//   phase 1 : after the H half-step                  -> total violation, decide
//   phase 2 : after the W half-step (update_H=False) -> decide on the W violation alone
// NB the `active` flag is read by all four blocks of a slot and cleared by block 0 of the
// Zero-copy snapshot: the slot state goes straight into a host-mapped (coherent, pinned) ring entry --
// no copy kernel, no event per iteration.  The payload is written first, then (after a system-scope
// fence) the stamp `pad_` = iteration number + 1; the host polls the stamp and reads the payload
// only once it matches.
__device__ __forceinline__ void publish_slot(SlotDesc* dst, const SlotDesc& s, int stamp)
{
    volatile SlotDesc* d = dst;
    d->off = s.off; d->k = s.k; d->active = s.active; d->iter = s.iter; d->restart = s.restart;
    d->viol = s.viol; d->viol_init = s.viol_init; d->viol_last = s.viol_last;
    __threadfence_system();
    d->pad_ = stamp;
}

// same launch; the flag copy `was_active` taken at entry keeps the other three consistent
// enough: a block that sees the cleared flag skips a gram nobody will read again.
struct FinalizeArgs {
    const float* gram_part; const double* viol_part; int nparts;
    float* gram_out; float l2_reg;
    SlotDesc* slots; int phase; double tol; int max_iter, want_gram, gld;
    SlotDesc* snap_out; int stamp;
};

// `red` = 256 doubles of LDS; (slot, by) = the block's coordinates in the (slots, ceil(kmax^2/256)) grid
__device__ __forceinline__ void finalize_body(const FinalizeArgs& a, int slot, int by, double* red)
{
    const float* __restrict__ gram_part = a.gram_part;
    const double* __restrict__ viol_part = a.viol_part;
    const int nparts = a.nparts;
    float* __restrict__ gram_out = a.gram_out;
    const float l2_reg = a.l2_reg;
    SlotDesc* __restrict__ slots = a.slots;
    const int phase = a.phase, max_iter = a.max_iter, want_gram = a.want_gram, gld = a.gld, stamp = a.stamp;
    const double tol = a.tol;
    SlotDesc* snap_out = a.snap_out;
    SlotDesc* sd = &slots[slot];
    const int tid = threadIdx.x;
    if (!sd->active) {
        // the host's view of this iteration (zero-copy snapshot, see publish_slot): unchanged state
        if (snap_out && by == 0 && tid == 0) publish_slot(snap_out + slot, *sd, stamp);
        return;
    }
    const int k = sd->k;
    if (want_gram) {
        const int e = by * 256 + tid;          // grid.y covers kmax*kmax entries
        if (e < k * k) {
            const int r = e / k, c = e % k;
            const size_t gsz = (size_t)gld * gld;
            const float* gp = gram_part + (size_t)slot * nparts * gsz + r * gld + c;
            // the partials of up to 32 workgroups are requested together, then added in the order of the plain loop
            // (s_j += part[4 i + j]; the remainder into s0): this launch is pure latency -- 49 partials fetched four
            // at a time were twelve dependent round trips (~10 of the launch's 13 us, twice per iteration)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const int n4 = nparts & ~3;
            for (int p0 = 0; p0 < nparts; p0 += 32) {
                float v[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) v[j] = gp[(size_t)min(p0 + j, nparts - 1) * gsz];
#pragma unroll
                for (int j = 0; j < 32; j += 4) {
                    if (p0 + j + 4 <= n4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (p0 + j + q >= n4 && p0 + j + q < nparts) s0 += v[j + q];
                    }
                }
            }
            float s = (s0 + s1) + (s2 + s3);
            if (r == c) s += l2_reg;
            gram_out[(size_t)slot * GRAM_SZ + r * GRAM_LD + c] = s;
        }
    }
    if (by != 0) return;
    double v = 0.0;
    for (int pI = tid; pI < nparts; pI += 256) v += viol_part[(size_t)slot * nparts + pI];
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        const double vsum = red[0];
        if (phase == 0) {
            sd->viol = vsum;
        } else {
            const double viol = (phase == 1) ? sd->viol + vsum : vsum;
            const int it = sd->iter + 1;
            sd->iter = it;
            if (it == 1) sd->viol_init = viol;
            bool done = false;
            if (sd->viol_init == 0.0) { done = true; sd->viol_last = 0.0; }
            else {
                sd->viol_last = viol / sd->viol_init;
                if (sd->viol_last <= tol) done = true;
            }
            if (it >= max_iter) done = true;
            sd->viol = 0.0;
            if (done) sd->active = 0;
        }
        if (snap_out) publish_slot(snap_out + slot, *sd, stamp);
    }
}

__global__ __launch_bounds__(256) void finalize_kernel(
    const float* __restrict__ gram_part, const double* __restrict__ viol_part, int nparts,
    float* __restrict__ gram_out, float l2_reg,
    SlotDesc* __restrict__ slots, int phase, double tol, int max_iter, int want_gram, int gld,
    SlotDesc* snap_out = nullptr, int stamp = 0)
{
    __shared__ double red[256];
    const FinalizeArgs a{gram_part, viol_part, nparts, gram_out, l2_reg, slots, phase, tol, max_iter, want_gram, gld,
                         snap_out, stamp};
    finalize_body(a, blockIdx.x, blockIdx.y, red);
}

// Horizontal fusion for the split-operand modes: ONE launch carries the plane split of a factor (the first
// split_bx * split_by workgroups) and the finalize of the sweep that produced it (the rest) -- the two only
// read what the sweep wrote, and the ~9 us latency of the tiny finalize disappears behind the split.
__global__ __launch_bounds__(256) void split3_finalize_kernel(const float* __restrict__ src, int ld, int K, int TR,
                                                              unsigned short* __restrict__ dst,
                                                              const double* __restrict__ kscale, int split_bx,
                                                              int split_by, FinalizeArgs fa, int fin_y)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 64 * 48 * 2];
    const int b = blockIdx.x, nsplit = split_bx * split_by;
    if (b < nsplit) {
        split3_tiled_body(src, ld, K, TR, dst, kscale, b % split_bx, b / split_bx,
                          reinterpret_cast<unsigned short (*)[64][48]>(lds));
    } else {
        const int f = b - nsplit;
        finalize_body(fa, f / fin_y, f % fin_y, reinterpret_cast<double*>(lds));
    }
}

// The same for the f16 two-plane split (kernels_gemm2h.hip.h).
template <int TROWS, int TKB>
__global__ __launch_bounds__(256) void split2h_finalize_kernel(const float* __restrict__ src, int ld, int K, int TR,
                                                               unsigned short* __restrict__ dst,
                                                               const double* __restrict__ kscale,
                                                               const float* __restrict__ rmax_part, int parts,
                                                               float* __restrict__ inv_scale, int split_bx,
                                                               int split_by, FinalizeArgs fa, int fin_y,
                                                               SplitFused fu = SplitFused{nullptr, nullptr})
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 32 * 2];      // 16 KB (see split2h_tiled_kernel)
    const int b = blockIdx.x, nsplit = split_bx * split_by;
    if (b < nsplit) {
        split2h_tiled_body<TROWS, TKB>(src, ld, K, TR, dst, kscale, rmax_part, parts, inv_scale, b % split_bx, split_bx,
                                       b / split_bx, reinterpret_cast<unsigned short*>(lds),
                                       reinterpret_cast<float*>(lds), fu);
    } else {
        const int f = b - nsplit;
        finalize_body(fa, f / fin_y, f % fin_y, reinterpret_cast<double*>(lds));
    }
}

// Gram matrix of the k rows [off, off+k) of a component-major factor (used once
// per restart, for the initial HHt of H0; sklearn _nmf.py:386).  One workgroup per slot.
template <int KR, int GC>
__device__ __forceinline__ void gram_rows_body(const float* __restrict__ V, int ldv, int L, int k, int off, int slot,
                                               float* __restrict__ gram_out, float l2_reg, float* lds)
{
    float (*tile)[GC + 1] = reinterpret_cast<float (*)[GC + 1]>(lds);
    const int tid = threadIdx.x;
    // thread owns the (a,b) pairs e = tid + 256*i, e < k*k
    constexpr int NP = KR * KR / 256;
    double acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = 0.0;
    for (int g0 = 0; g0 < L; g0 += GC) {
        for (int e = tid; e < k * GC; e += 256) {
            const int c = e / GC, g = e % GC;
            tile[c][g] = (g0 + g < L) ? V[(size_t)(off + c) * ldv + g0 + g] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = tid + 256 * i;
            if (e < k * k) {
                const int a = e / k, b = e % k;
                float s = 0.f;
                for (int g = 0; g < GC; ++g) s = fmaf(tile[a][g], tile[b][g], s);
                acc[i] += (double)s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = tid + 256 * i;
        if (e < k * k) {
            const int a = e / k, b = e % k;
            float s = (float)acc[i];
            if (a == b) s += l2_reg;
            gram_out[(size_t)slot * GRAM_SZ + a * GRAM_LD + b] = s;
        }
    }
}

__global__ __launch_bounds__(256) void gram_rows_kernel(
    const float* __restrict__ V, int ldv, int L,
    const SlotDesc* __restrict__ slots, const int* __restrict__ slot_list,
    float* __restrict__ gram_out, float l2_reg)
{
    const int slot = slot_list[blockIdx.x];
    const SlotDesc sd = slots[slot];
    __shared__ __attribute__((aligned(16))) float lds[KSMALL * 129];          // = 64 x 129 >= 128 x 65
    // ranks <= 64: 128 columns per step (as in rounds 1-2: same bits); 65..128: 64 columns per step
    if (sd.k <= KSMALL) gram_rows_body<KSMALL, 128>(V, ldv, L, sd.k, sd.off, slot, gram_out, l2_reg, lds);
    else gram_rows_body<KMAX, 64>(V, ldv, L, sd.k, sd.off, slot, gram_out, l2_reg, lds);
}

// Install a restart into its slot: H0 [k][G] row-major -> H_all rows, W0 [N][k]
// row-major (sklearn layout) -> Wt_all rows (transposed).  Also clears the slot's
// padding.  grid = (ceil(max(N,G)/256), k)
__global__ void install_kernel(const float* __restrict__ H0, const float* __restrict__ W0,
                               float* __restrict__ H, int ldh, int G,
                               float* __restrict__ Wt, int ldw, int N, int off, int k)
{
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G) H[(size_t)(off + c) * ldh + i] = H0 ? H0[(size_t)c * G + i] : 0.f;
    if (i < N) Wt[(size_t)(off + c) * ldw + i] = W0 ? W0[(size_t)i * k + c] : 0.f;
}

// Same from a component-major source (the device-generated init store): H0 [k][G], Wt0 [k][N].
__global__ void install_cm_kernel(const float* __restrict__ H0, const float* __restrict__ Wt0,
                                  float* __restrict__ H, int ldh, int G,
                                  float* __restrict__ Wt, int ldw, int N, int off)
{
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G) H[(size_t)(off + c) * ldh + i] = H0[(size_t)c * G + i];
    if (i < N) Wt[(size_t)(off + c) * ldw + i] = Wt0[(size_t)c * N + i];
}

__global__ void set_slot_off_kernel(SlotDesc* slots, int slot, int off) { slots[slot].off = off; }

// Re-packing of the component columns in ONE pass: stage[j][:] = V[colmap[j]][:]  (colmap[j] < 0: a zero row), then the
// staged rows are copied back with one device-to-device copy.  grid = (ceil(ld / 1024), rows), 256 threads x float4.
__global__ __launch_bounds__(256) void gather_rows_cm_kernel(const float* __restrict__ V, int ld,
                                                             const int* __restrict__ colmap, float* __restrict__ stage)
{
    const int j = blockIdx.y, i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= ld) return;                                        // ld is a multiple of 4 (padded leading dimensions)
    const int c = colmap[j];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c >= 0) v = *reinterpret_cast<const float4*>(V + (size_t)c * ld + i);
    *reinterpret_cast<float4*>(stage + (size_t)j * ld + i) = v;
}

__global__ void set_ints_kernel(int* __restrict__ a, int* __restrict__ b, int off, int n, int value)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[off + i] = value; b[off + i] = value; }
}
// dst[j] = colmap[j] >= 0 ? src[colmap[j]] : 0   (the per-column exponents follow their columns through a re-packing)
__global__ void permute_ints_kernel(const int* __restrict__ src, const int* __restrict__ colmap, int* __restrict__ dst, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) { const int c = colmap[j]; dst[j] = c >= 0 ? src[c] : 0; }
}

__global__ void set_slot_offs_kernel(SlotDesc* slots, const int* __restrict__ ids, const int* __restrict__ offs, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[ids[i]].off = offs[i];
}

// Zero a range of packed component rows (freed slot -> contributes nothing to the GEMMs).
__global__ void clear_rows_kernel(float* __restrict__ V, int ldv, int L, int off, int k)
{
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < k && i < L) V[(size_t)(off + c) * ldv + i] = 0.f;
}

// Copy k packed rows out to a dense [k][L] result block (optionally transposed to [L][k]).
__global__ void extract_kernel(const float* __restrict__ V, int ldv, int L, int off, int k,
                               float* __restrict__ out, int transpose)
{
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < k && i < L) {
        const float v = V[(size_t)(off + c) * ldv + i];
        if (transpose) out[(size_t)i * k + c] = v; else out[(size_t)c * L + i] = v;
    }
}

// fp32 copy of the caller's matrix into the zero-padded device layout.
__global__ void pad_copy_kernel(const float* __restrict__ src, int rows, int cols,
                                float* __restrict__ dst, int ld)
{
    const int r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows && c < cols) dst[(size_t)r * ld + c] = src[(size_t)r * cols + c];
}

}  // namespace cnmf
