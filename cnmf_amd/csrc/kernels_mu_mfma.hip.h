// Kullback-Leibler multiplicative updates on the matrix pipe, batched over restarts.
// Restates sklearn/decomposition/_nmf.py:526-631 (_multiplicative_update_w), :634-728 (_multiplicative_update_h)
// and :84-194 (_beta_divergence) for beta = 1 and dense X -- the solver the reference keeps for
// beta_loss='kullback-leibler' (cnmf.py:618-631).  kernels_mu.hip.h holds the vector-ALU version (Itakura-Saito,
// ranks above 32, and the A/B fallback CNMF_MU_VALU=1).
//
// Per 32 x 32 tile of X and per restart, two chained MFMA products with the elementwise quotient in between:
//   S = W.H        (v_mfma_f32_32x32x16_bf16, the rank is the k dimension; factors as two bf16 planes hi + lo,
//                   all four plane products: 2^-17 relative)
//   Q = X / max(S, eps)                       (v_rcp_f32 on the accumulator registers, in the MFMA C layout)
//   numerator += Q.H^T  (W half-step)  or  W^T.Q  (H half-step)
// The second product takes Q straight from the registers of the first: the C layout of v_mfma_f32_32x32 (lane l holds
// column l % 32 and the 16 rows 8 * (r / 4) + 4 * (l / 32) + r % 4) IS a B operand of the next v_mfma_32x32x16 when
// the reduction runs over those rows -- registers 0..7 are the 8 k values of one MFMA, 8..15 of a second -- provided
// the A operand lists its k values in the same order.  That order (per 16 block: 0..3, 8..11 | 4..7, 12..15) is baked
// into the component-major plane copies of the factors (`pos16`), so every operand fragment is one 16-byte load.
//   H half-step: tile rows = cells (reduced), columns = genes  -> X read as stored (coalesced along genes)
//   W half-step: tile rows = genes (reduced), columns = cells  -> X^T (a resident transposed copy, built once)
// For ranks <= 16 the hi and lo planes of the second product's A operand are stacked in the M dimension
// (rows 0..15 = hi, 16..31 = lo of the same 16 components): one MFMA per Q plane covers both.
//
// Restarts share each X tile: the waves of a workgroup are different restarts on the SAME tile (the L1 serves the
// repeats), further restart groups sit in gridDim.z.  Each wave owns two tiles along the non-reduced dimension, so a
// streamed factor fragment is used twice.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_gemm3.hip.h"
#include "kernels_mu.hip.h"

namespace cnmf {

typedef __attribute__((ext_vector_type(8))) __bf16 mu_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 mu_bf16x2;
typedef float mu_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short mu_u16;

constexpr int MU_MAXSLOTS = 16;

struct MuSlotDev {                     // one restart in flight (all pointers device memory)
    float* W;                          // [Np][KP]
    float* Ht;                         // [Gs][KP]
    mu_u16 *Wp_hi, *Wp_lo;             // row-major bf16 planes of W   [Np][KP]
    mu_u16 *Wc_hi, *Wc_lo;             // component-major, pos16 order [KP][Np]; ONE allocation, lo follows hi
    mu_u16 *Hp_hi, *Hp_lo;             // [Gs][KP]
    mu_u16 *Hc_hi, *Hc_lo;             // [KP][Gs]; ONE allocation, lo follows hi
    float *Hsum, *Wsum;                // [KP] column sums
    float* pnum;                       // [nchunks][Gs][KP] partial numerators of the H half-step
    double* divpart;                   // [nstrips] partial divergences
    double* cspart;                    // [256][KP] column-sum partials
};
struct MuBatch {
    int n;
    MuSlotDev s[MU_MAXSLOTS];
};

// position of row / gene `i` inside its 16 block in the component-major planes (swap bits 2 and 3)
__device__ __host__ __forceinline__ int mu_pos16(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

__device__ __forceinline__ unsigned mu_pack_bf16(float a, float b)
{
    const mu_bf16x2 v = __builtin_convertvector(mu_f32x2{a, b}, mu_bf16x2);     // round to nearest even
    return __builtin_bit_cast(unsigned, v);
}
// x = hi + lo (+ 2^-17 |x|): both planes as bf16 bit patterns
__device__ __forceinline__ void mu_split_bf16(float x, mu_u16& hi, mu_u16& lo)
{
    const unsigned p = mu_pack_bf16(x, 0.f);
    const float h = __uint_as_float(p << 16);
    const unsigned q = mu_pack_bf16(x - h, 0.f);
    hi = (mu_u16)(p & 0xffffu); lo = (mu_u16)(q & 0xffffu);
}

__device__ __forceinline__ mu_bf16x8 mu_ld8(const mu_u16* p)
{
    return __builtin_bit_cast(mu_bf16x8, *reinterpret_cast<const u32x4*>(p));
}

// max(s, eps) as ONE instruction: v_med3_f32(s, eps, +inf)  (fmaxf adds a canonicalising v_max for signalling NaNs
// an MFMA result cannot be; inline assembly is not an option -- the hazard recogniser does not see an asm statement
// reading MFMA results and omits the wait states)
__device__ __forceinline__ float mu_clamp_eps(float s)
{
    return __builtin_amdgcn_fmed3f(s, MU_EPS, __builtin_inff());
}

// quotient planes of one tile: q[r] = x[r] / max(s[r], eps) as bf16 hi / lo, packed in register order
// (registers 0..7 -> first B operand, 8..15 -> second).  v_rcp_f32 is a quarter-rate instruction: one reciprocal
// serves two elements, q0 = x0 s1 / (s0 s1), q1 = x1 s0 / (s0 s1)  (s >= eps = 1.2e-7: the product cannot underflow).
__device__ __forceinline__ void mu_quotient_planes(const f32x16& s, const float* x, u32x4 (&qh)[2], u32x4 (&ql)[2])
{
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        unsigned ph[4], pl[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int r = 8 * c + 2 * d;
            const float s0 = mu_clamp_eps(s[r]), s1 = mu_clamp_eps(s[r + 1]);
            const float rc = __builtin_amdgcn_rcpf(s0 * s1);
            const float q0 = (x[r] * s1) * rc, q1 = (x[r + 1] * s0) * rc;
            ph[d] = mu_pack_bf16(q0, q1);
            const float h0 = __uint_as_float(ph[d] << 16), h1 = __uint_as_float(ph[d] & 0xffff0000u);
            pl[d] = mu_pack_bf16(q0 - h0, q1 - h1);
        }
        qh[c] = u32x4{ph[0], ph[1], ph[2], ph[3]};
        ql[c] = u32x4{pl[0], pl[1], pl[2], pl[3]};
    }
}

#define MU_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)

// S = A.B over the rank (KS = KP / 16 k steps), all four plane products
template <int KS>
__device__ __forceinline__ f32x16 mu_product(const mu_bf16x8 (&a)[KS][2], const mu_bf16x8 (&b)[KS][2])
{
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s = MU_MFMA(a[0][1], b[0][1], zero);      // smallest terms first; C = inline constant 0
    s = MU_MFMA(a[0][0], b[0][1], s);
    s = MU_MFMA(a[0][1], b[0][0], s);
    s = MU_MFMA(a[0][0], b[0][0], s);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) {
        s = MU_MFMA(a[ks][1], b[ks][1], s);
        s = MU_MFMA(a[ks][0], b[ks][1], s);
        s = MU_MFMA(a[ks][1], b[ks][0], s);
        s = MU_MFMA(a[ks][0], b[ks][0], s);
    }
    return s;
}

// numerator += A2 . Q  (A2: component planes in pos16 order, Q: the quotient planes)
//   KP = 16: a2[c][0] holds hi (rows 0..15) and lo (rows 16..31) stacked; KP = 32: a2[c][0] = hi, a2[c][1] = lo
template <int KP>
__device__ __forceinline__ void mu_accumulate(f32x16& acc, const mu_bf16x8 (&a2)[2][KP == 16 ? 1 : 2],
                                              const u32x4 (&qh)[2], const u32x4 (&ql)[2])
{
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const mu_bf16x8 bh = __builtin_bit_cast(mu_bf16x8, qh[c]), bl = __builtin_bit_cast(mu_bf16x8, ql[c]);
        if constexpr (KP == 16) {
            acc = MU_MFMA(a2[c][0], bl, acc);
            acc = MU_MFMA(a2[c][0], bh, acc);
        } else {
            acc = MU_MFMA(a2[c][1], bl, acc);
            acc = MU_MFMA(a2[c][1], bh, acc);
            acc = MU_MFMA(a2[c][0], bl, acc);
            acc = MU_MFMA(a2[c][0], bh, acc);
        }
    }
}

// wave -> (slot, sub): `sw` (1, 2 or 4) waves of a workgroup are different restarts, the other 4 / sw are further tiles
__device__ __forceinline__ bool mu_wave_role(const MuBatch& mb, int sw, int& slot, int& sub)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    slot = blockIdx.z * sw + (wave % sw);
    sub = wave / sw;
    return slot < mb.n;
}

// Streamed operands come in through buffer loads: resource descriptor (wave-uniform base) + per-lane 32-bit byte
// offset + scalar offset -- no 64-bit per-lane address arithmetic in the tile loop.
typedef __amdgpu_buffer_rsrc_t mu_rsrc;
__device__ __forceinline__ mu_rsrc mu_make_rsrc(const void* base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float mu_bld(mu_rsrc rs, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ mu_bf16x8 mu_bld8(mu_rsrc rs, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(mu_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// The tile loop shared by both half-steps.  Streams, per 32-deep step t of the reduced dimension:
//   x      : X values of the wave's two 32 x 32 tiles in the C layout -- row (8 (r / 4) + 4 h + r % 4) of the step,
//            column l32 (+32 for the second tile); `xs` = bytes between consecutive rows of the stepped dimension
//   a1     : first product's A fragments (row-major planes of the streamed factor, KP bf16 per row)
//   a2     : second product's A fragments (component-major planes, `cs` elements per component row, lo plane after hi)
// Every fragment is re-requested for step t + 1 as soon as its last use in step t has been issued (same registers:
// the loads travel under the rest of the step), pinned by scheduling barriers so that the compiler does not sink them
// back to their next use.
template <int KP, bool SECOND>
struct MuStream {
    mu_rsrc rs_a1h, rs_a1l, rs_a2;
    const float* xbase;            // X (H half-step) or X^T (W half-step)
    size_t xstep;                  // floats per 32-step of the reduced dimension
    unsigned xv;                   // per-lane byte offset inside a step
    unsigned xs;                   // bytes per row of the reduced dimension
    unsigned a1o, a2o, plane_b;
    float x[2][16];
    mu_bf16x8 a1[KP / 16][2];
    mu_bf16x8 a2[2][KP == 16 ? 1 : 2];

    __device__ __forceinline__ void load_x(int jt, int t)
    {
        const mu_rsrc rs = mu_make_rsrc(xbase + (size_t)t * xstep);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[jt][r] = mu_bld(rs, xv, (unsigned)(8 * (r >> 2) + (r & 3)) * xs + 128u * jt);
    }
    __device__ __forceinline__ void load_a1(int t)
    {
        const unsigned so = 2u * (unsigned)(t * 32 * KP);
#pragma unroll
        for (int ks = 0; ks < KP / 16; ++ks) {
            a1[ks][0] = mu_bld8(rs_a1h, a1o, so + 32u * ks);
            a1[ks][1] = mu_bld8(rs_a1l, a1o, so + 32u * ks);
        }
    }
    __device__ __forceinline__ void load_a2(int t)
    {
        if constexpr (SECOND) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int p = 0; p < (KP == 16 ? 1 : 2); ++p)
                    a2[c][p] = mu_bld8(rs_a2, a2o, 2u * (unsigned)(t * 32 + 16 * c) + p * plane_b);
        }
    }
};

#define MU_PIN() __builtin_amdgcn_sched_barrier(0)

// numerators of steps [t0, t1) into acc[2]
template <int KP>
__device__ __forceinline__ void mu_tile_loop(MuStream<KP, true>& st, const mu_bf16x8 (&b1)[2][KP / 16][2],
                                             f32x16 (&acc)[2], int t0, int t1)
{
    if (t0 >= t1) return;
    st.load_x(0, t0); st.load_x(1, t0); st.load_a1(t0); st.load_a2(t0);
    for (int t = t0; t < t1; ++t) {
        const int tn = min(t + 1, t1 - 1);
        u32x4 qh[2], ql[2];
        f32x16 s = mu_product<KP / 16>(st.a1, b1[0]);
        mu_quotient_planes(s, st.x[0], qh, ql);
        MU_PIN(); st.load_x(0, tn); MU_PIN();
        mu_accumulate<KP>(acc[0], st.a2, qh, ql);
        s = mu_product<KP / 16>(st.a1, b1[1]);
        MU_PIN(); st.load_a1(tn); MU_PIN();
        mu_quotient_planes(s, st.x[1], qh, ql);
        MU_PIN(); st.load_x(1, tn); MU_PIN();
        mu_accumulate<KP>(acc[1], st.a2, qh, ql);
        MU_PIN(); st.load_a2(tn); MU_PIN();
    }
}

// ---- H half-step partials.  grid = (gene strips of 64, row chunks / (4 / sw), restart groups), block = 256
//   pnum[chunk][g][c] = sum_{i in chunk} W[i][c] Q[i][g]
template <int KP>
__global__ __launch_bounds__(256) void mu_h_mfma_kernel(const float* __restrict__ X, int ldx, int Np, int Gs,
                                                        MuBatch mb, int tiles_per_chunk, int nchunks, int sw)
{
    constexpr int KS = KP / 16;
    int slot, sub;
    if (!mu_wave_role(mb, sw, slot, sub)) return;
    const MuSlotDev& sd = mb.s[slot];
    const int lane = threadIdx.x & 63, l32 = lane & 31, h = lane >> 5;
    const int g0 = blockIdx.x * 64;
    const int chunk = blockIdx.y * (4 / sw) + sub;
    if (chunk >= nchunks) return;
    const int ntiles = Np / 32;
    const int rt0 = chunk * tiles_per_chunk, rt1 = min(ntiles, rt0 + tiles_per_chunk);

    mu_bf16x8 b1[2][KS][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t o = (size_t)(g0 + 32 * jt + l32) * KP + 16 * ks + 8 * h;
            b1[jt][ks][0] = mu_ld8(sd.Hp_hi + o);
            b1[jt][ks][1] = mu_ld8(sd.Hp_lo + o);
        }
    f32x16 acc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
    // second product's A operand: row m of the 32 = (plane, component); Wc_lo follows Wc_hi in memory.
    // (a last gene strip beyond ldx reads into the next row -- X has a slack row -- and only feeds padded genes)
    MuStream<KP, true> st;
    st.rs_a1h = mu_make_rsrc(sd.Wp_hi); st.rs_a1l = mu_make_rsrc(sd.Wp_lo); st.rs_a2 = mu_make_rsrc(sd.Wc_hi);
    st.xbase = X; st.xstep = (size_t)32 * ldx; st.xs = 4u * (unsigned)ldx;
    st.xv = 4u * (unsigned)(4 * h * ldx + g0 + l32);
    st.a1o = 2u * (unsigned)(l32 * KP + 8 * h);
    st.plane_b = 2u * (unsigned)KP * (unsigned)Np;
    st.a2o = (KP == 16) ? (2u * (unsigned)((l32 & 15) * Np + 8 * h) + (l32 >> 4) * st.plane_b)
                        : 2u * (unsigned)(l32 * Np + 8 * h);
    mu_tile_loop<KP>(st, b1, acc, rt0, rt1);
    // C layout: register r <-> row m = 8 (r / 4) + 4 h + r % 4, column (gene) l32
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        float* pn = sd.pnum + ((size_t)chunk * Gs + g0 + 32 * jt + l32) * KP;
        if constexpr (KP == 16) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                v4f v;
                v.x = acc[jt][4 * q + 0] + acc[jt][4 * q + 8]; v.y = acc[jt][4 * q + 1] + acc[jt][4 * q + 9];
                v.z = acc[jt][4 * q + 2] + acc[jt][4 * q + 10]; v.w = acc[jt][4 * q + 3] + acc[jt][4 * q + 11];
                *reinterpret_cast<v4f*>(pn + 8 * q + 4 * h) = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f v = {acc[jt][4 * q], acc[jt][4 * q + 1], acc[jt][4 * q + 2], acc[jt][4 * q + 3]};
                *reinterpret_cast<v4f*>(pn + 8 * q + 4 * h) = v;
            }
        }
    }
}

// ---- H half-step finish: Ht[g][c] *= num / den, planes refreshed.  grid = (ceil(Gs * KP / 256), slots)
template <int KP>
__global__ __launch_bounds__(256) void mu_h_finish_mfma_kernel(MuBatch mb, int G, int Gs, int nchunks, float l1, float l2)
{
    const MuSlotDev& sd = mb.s[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Gs * KP) return;
    const int g = e / KP, c = e % KP;
    float v = 0.f;
    if (g < G) {
        float n = 0.f;
        for (int q = 0; q < nchunks; ++q) n += sd.pnum[(size_t)q * Gs * KP + e];
        float dn = sd.Wsum[c];
        if (dn == 0.f) dn = 1.0f;                              // sklearn _nmf.py:684-686
        const float hv = sd.Ht[e];
        if (l1 > 0.f) dn += l1;
        if (l2 > 0.f) dn += l2 * hv;
        if (dn == 0.f) dn = MU_EPS;
        v = hv * (n / dn);
        if (v < F64_EPS_AS_F32) v = 0.f;                       // sklearn _nmf.py:868-869
    }
    sd.Ht[e] = v;
    mu_u16 hi, lo;
    mu_split_bf16(v, hi, lo);
    sd.Hp_hi[e] = hi; sd.Hp_lo[e] = lo;
    const size_t o = (size_t)c * Gs + (g & ~15) + mu_pos16(g & 15);
    sd.Hc_hi[o] = hi; sd.Hc_lo[o] = lo;
}

// planes of a freshly installed factor M [L][KP] (L = Np or Gs; rows >= the live count must already be zero)
template <int KP>
__global__ __launch_bounds__(256) void mu_planes_kernel(const float* __restrict__ M, int L, mu_u16* __restrict__ p_hi,
                                                        mu_u16* __restrict__ p_lo, mu_u16* __restrict__ c_hi,
                                                        mu_u16* __restrict__ c_lo)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= L * KP) return;
    const int i = e / KP, c = e % KP;
    mu_u16 hi, lo;
    mu_split_bf16(M[e], hi, lo);
    p_hi[e] = hi; p_lo[e] = lo;
    const size_t o = (size_t)c * L + (i & ~15) + mu_pos16(i & 15);
    c_hi[o] = hi; c_lo[o] = lo;
}

// ---- W half-step (MODE 0) / divergence of the current factors (MODE 1).
// grid = (row strips of 64 / (4 / sw), 1, restart groups), block = 256.  A wave owns 64 cells and walks all genes.
template <int KP, int MODE>
__global__ __launch_bounds__(256) void mu_w_mfma_kernel(const float* __restrict__ Xt, int ldxt, int N, int Gs,
                                                        MuBatch mb, int sw, float l1, float l2)
{
    constexpr int KS = KP / 16;
    int slot, sub;
    if (!mu_wave_role(mb, sw, slot, sub)) return;
    const MuSlotDev& sd = mb.s[slot];
    const int lane = threadIdx.x & 63, l32 = lane & 31, h = lane >> 5;
    const int strip = blockIdx.x * (4 / sw) + sub;
    const int r0 = strip * 64;
    if (r0 >= N) return;

    mu_bf16x8 b1[2][KS][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t o = (size_t)(r0 + 32 * jt + l32) * KP + 16 * ks + 8 * h;
            b1[jt][ks][0] = mu_ld8(sd.Wp_hi + o);
            b1[jt][ks][1] = mu_ld8(sd.Wp_lo + o);
        }
    f32x16 acc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
    double dv = 0.0;
    const int ngt = Gs / 32;
    if constexpr (MODE == 0) {
        MuStream<KP, true> st;                                                  // Hc_lo follows Hc_hi in memory
        st.rs_a1h = mu_make_rsrc(sd.Hp_hi); st.rs_a1l = mu_make_rsrc(sd.Hp_lo); st.rs_a2 = mu_make_rsrc(sd.Hc_hi);
        st.xbase = Xt; st.xstep = (size_t)32 * ldxt; st.xs = 4u * (unsigned)ldxt;
        st.xv = 4u * (unsigned)(4 * h * ldxt + r0 + l32);
        st.a1o = 2u * (unsigned)(l32 * KP + 8 * h);
        st.plane_b = 2u * (unsigned)KP * (unsigned)Gs;
        st.a2o = (KP == 16) ? (2u * (unsigned)((l32 & 15) * Gs + 8 * h) + (l32 >> 4) * st.plane_b)
                            : 2u * (unsigned)(l32 * Gs + 8 * h);
        mu_tile_loop<KP>(st, b1, acc, 0, ngt);
    } else {
        // sum over X > eps of  X log(X / WH) - X   (sklearn _nmf.py:125-141; + sum WH added by the host)
        MuStream<KP, false> st;
        st.rs_a1h = mu_make_rsrc(sd.Hp_hi); st.rs_a1l = mu_make_rsrc(sd.Hp_lo);
        st.xbase = Xt; st.xstep = (size_t)32 * ldxt; st.xs = 4u * (unsigned)ldxt;
        st.xv = 4u * (unsigned)(4 * h * ldxt + r0 + l32);
        st.a1o = 2u * (unsigned)(l32 * KP + 8 * h);
        for (int gt = 0; gt < ngt; ++gt) {
            st.load_x(0, gt); st.load_x(1, gt); st.load_a1(gt);
            float part = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const f32x16 s = mu_product<KS>(st.a1, b1[jt]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float xv = st.x[jt][r];
                    if (xv > MU_EPS) part += xv * logf(xv / fmaxf(s[r], MU_EPS)) - xv;
                }
            }
            dv += (double)part;
        }
    }
    if constexpr (MODE == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dv += __shfl_xor(dv, o, 64);
        if (lane == 0) sd.divpart[strip] = dv;
        return;
    } else {
        // C layout: register r <-> component m = 8 (r / 4) + 4 h + r % 4, column (cell) l32
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int row = r0 + 32 * jt + l32;
            if (row >= N) continue;
            constexpr int NQ = KP / 8;                     // groups of 4 components held by this lane
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c0 = 8 * q + 4 * h;
                float num[4];
                if constexpr (KP == 16) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) num[t] = acc[jt][4 * q + t] + acc[jt][4 * q + t + 8];
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) num[t] = acc[jt][4 * q + t];
                }
                float* wp = sd.W + (size_t)row * KP + c0;
                const v4f wv = *reinterpret_cast<const v4f*>(wp);
                const float w[4] = {wv.x, wv.y, wv.z, wv.w};
                float o[4];
                mu_u16 hi[4], lo[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float dn = sd.Hsum[c0 + t];
                    if (l1 > 0.f) dn += l1;
                    if (l2 > 0.f) dn += l2 * w[t];
                    if (dn == 0.f) dn = MU_EPS;
                    o[t] = w[t] * (num[t] / dn);
                    mu_split_bf16(o[t], hi[t], lo[t]);
                    const size_t co = (size_t)(c0 + t) * ldxt + (row & ~15) + mu_pos16(row & 15);
                    sd.Wc_hi[co] = hi[t]; sd.Wc_lo[co] = lo[t];
                }
                *reinterpret_cast<v4f*>(wp) = v4f{o[0], o[1], o[2], o[3]};
                uint2 ph, pl;
                ph.x = hi[0] | ((unsigned)hi[1] << 16); ph.y = hi[2] | ((unsigned)hi[3] << 16);
                pl.x = lo[0] | ((unsigned)lo[1] << 16); pl.y = lo[2] | ((unsigned)lo[3] << 16);
                *reinterpret_cast<uint2*>(sd.Wp_hi + (size_t)row * KP + c0) = ph;
                *reinterpret_cast<uint2*>(sd.Wp_lo + (size_t)row * KP + c0) = pl;
            }
        }
    }
}

// ---- column sums of a factor, batched over the slots: which = 0 -> W [rows N] into Wsum, 1 -> Ht [rows G] into Hsum.
// Two levels in a fixed order (as mu_colsum_*_kernel).  grid = (nb, slots) then (1, slots)
template <int KP>
__global__ __launch_bounds__(256) void mu_colsum_batch_part_kernel(MuBatch mb, int which, int R)
{
    __shared__ double red[256 / KP][KP];
    constexpr int RG = 256 / KP;
    const MuSlotDev& sd = mb.s[blockIdx.y];
    const float* M = which ? sd.Ht : sd.W;
    const int c = threadIdx.x % KP, rg = threadIdx.x / KP;
    const int per = (R + gridDim.x - 1) / gridDim.x;
    const int rb = blockIdx.x * per, re = min(R, rb + per);
    double s = 0.0;
    for (int r = rb + rg; r < re; r += RG) s += (double)M[(size_t)r * KP + c];
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0) {
        double t = 0.0;
        for (int q = 0; q < RG; ++q) t += red[q][c];
        sd.cspart[(size_t)blockIdx.x * KP + c] = t;
    }
}
// block = 1024 threads = 16 waves, a wave per component (two for KP = 32): lanes take the partials round robin, then a
// fixed xor tree
template <int KP>
__global__ __launch_bounds__(1024) void mu_colsum_batch_final_kernel(MuBatch mb, int which, int nb)
{
    const MuSlotDev& sd = mb.s[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = wave; c < KP; c += 16) {
        double t = 0.0;
        for (int b = lane; b < nb; b += 64) t += sd.cspart[(size_t)b * KP + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) (which ? sd.Hsum : sd.Wsum)[c] = (float)t;
    }
}

// X [N_pad][ldx] -> Xt [Gs][ldxt] (zero beyond N x G).  grid = (ceil(Gs / 32), ceil(ldxt / 32)), block = (32, 8)
__global__ __launch_bounds__(256) void mu_transpose_kernel(const float* __restrict__ X, int ldx, int N, int G,
                                                           float* __restrict__ Xt, int ldxt, int Gs)
{
    __shared__ float tile[32][33];
    const int g0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, g = g0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < N && g < G) ? X[(size_t)r * ldx + g] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int g = g0 + j, r = r0 + threadIdx.x;
        if (g < Gs && r < ldxt) Xt[(size_t)g * ldxt + r] = tile[threadIdx.x][j];
    }
}

}  // namespace cnmf
