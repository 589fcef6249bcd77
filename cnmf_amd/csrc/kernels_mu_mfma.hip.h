// Multiplicative updates (Kullback-Leibler, Itakura-Saito) on the matrix pipe, batched over restarts.
// Restates sklearn/decomposition/_nmf.py:526-631 (_multiplicative_update_w), :634-728 (_multiplicative_update_h)
// and :84-194 (_beta_divergence) for beta = 1 / 0 and dense X -- the solver the reference keeps for
// beta_loss != 'frobenius' (cnmf.py:618-631).  kernels_mu.hip.h holds the vector-ALU version (the A/B fallback
// CNMF_MU_VALU=1 and matrices beyond the 2^24 addressing bound).  Below the text describes beta = 1; Itakura-Saito carries a second ratio
// (1 / S for the denominator next to X / S^2 for the numerator) through a second set of accumulators.
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
// Restarts share each X tile: a workgroup is 4 restarts on the SAME block of X (staged once through LDS), further
// restart groups sit in gridDim.z.  Each wave owns two 32 x 32 tiles along the non-reduced dimension.
// Measured (PMC, 16 restarts of rank 9 at 50000 x 2000): vector ALU 60 % busy (the quotient: clamp, reciprocal, two
// bf16 planes = about 8 issue slots per element), matrix pipe 24 % -- they do not overlap inside a SIMD here, so the step is
// bound by their sum.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_gemm3.hip.h"
#include "kernels_mu.hip.h"

namespace cnmf {

typedef __attribute__((ext_vector_type(8))) __bf16 mu_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 mu_bf16x2;
typedef float mu_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short mu_u16;

constexpr int MU_MAXSLOTS = 32;       // (the batch travels as a kernel argument: 32 x 120 B < the 4 KB limit)

struct MuSlotDev {                     // one restart in flight (all pointers device memory)
    float* W;                          // [Np][KP]
    float* Ht;                         // [Gs][KP]
    mu_u16 *Wp_hi, *Wp_lo;             // row-major bf16 planes of W   [Np][KP]
    mu_u16* Wc_hi;                     // component-major, pos16 order [2][KP][Np]: the lo plane follows the hi plane
    mu_u16 *Hp_hi, *Hp_lo;             // [Gs][KP]
    mu_u16* Hc_hi;                     // [2][KP][Gs], likewise
    float *Hsum, *Wsum;                // [KP] column sums
    float* pnum;                       // [nchunks][Gs][KP] partial numerators of the H half-step
    float* pden;                       // ... and denominators (Itakura-Saito only)
    double* divpart;                   // [nstrips] partial divergences
    double* cspart;                    // [256][KP] column-sum partials
    int k;                             // rank of the restart (the non-zero path skips the padding quads of a factor row)
};
struct MuBatch {
    int n;
    MuSlotDev s[MU_MAXSLOTS];
};
static_assert(sizeof(MuBatch) <= 3968, "MuBatch travels by value: keep the kernel arguments below 4 KB");
static_assert(sizeof(MuBatch) <= 3968, "MuBatch travels by value: kernel arguments are limited to 4 KB");

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

// max(s, eps) as ONE instruction: a signed integer maximum of the bit patterns (the order of non-negative floats is the
// order of their bits; a negative s -- impossible for a product of non-negative factors -- would also clamp to eps).
// fmaxf / v_med3_f32 cost a second, canonicalising v_max; inline assembly is not an option -- the hazard recogniser
// does not see an asm statement reading MFMA results and omits the wait states.
__device__ __forceinline__ float mu_clamp_eps(float s)
{
    return __int_as_float(max(__float_as_int(s), __float_as_int(MU_EPS)));
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

// bf16 hi / lo planes of 8 values (register order), as MFMA B operands
__device__ __forceinline__ void mu_planes8(const float (&q)[8], mu_bf16x8& bh, mu_bf16x8& bl)
{
    unsigned ph[4], pl[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        ph[d] = mu_pack_bf16(q[2 * d], q[2 * d + 1]);
        const float h0 = __uint_as_float(ph[d] << 16), h1 = __uint_as_float(ph[d] & 0xffff0000u);
        pl[d] = mu_pack_bf16(q[2 * d] - h0, q[2 * d + 1] - h1);
    }
    bh = __builtin_bit_cast(mu_bf16x8, u32x4{ph[0], ph[1], ph[2], ph[3]});
    bl = __builtin_bit_cast(mu_bf16x8, u32x4{pl[0], pl[1], pl[2], pl[3]});
}
// Shape of one restart inside a workgroup, by padded rank:
//   KP = 16, 32: 4 restarts x 2 halves of the 128-wide block, two 32-wide tiles per wave, one 32-row M tile of the
//                second product (for 16 the hi and lo planes are stacked in it);
//   KP = 64    : 2 restarts x 4 quarters, ONE tile per wave, TWO M tiles (components 0..31 and 32..63) -- the same
//                register and LDS budget, twice the MFMAs per element, the same quotient work.
template <int KP> struct MuShape {
    static constexpr int RPW = KP == 64 ? 2 : 4;         // restarts per workgroup
    static constexpr int NJT = KP == 64 ? 1 : 2;         // 32-wide tiles per wave
    static constexpr int NMH = KP == 64 ? 2 : 1;         // M tiles of the second product
    static constexpr int NA2 = KP == 16 ? 1 : 2 * NMH;   // A fragments of the second product per 16-deep chunk
    static constexpr int TPR = 512 / RPW;                // loader threads per restart
};
template <int KP> struct MuAcc { f32x16 v[MuShape<KP>::NMH]; };

template <int KP>
__device__ __forceinline__ void mu_accumulate8(MuAcc<KP>& acc, const mu_bf16x8 (&a2c)[MuShape<KP>::NA2], mu_bf16x8 bh, mu_bf16x8 bl)
{
    if constexpr (KP == 16) {
        acc.v[0] = MU_MFMA(a2c[0], bl, acc.v[0]);
        acc.v[0] = MU_MFMA(a2c[0], bh, acc.v[0]);
    } else {
#pragma unroll
        for (int mh = 0; mh < MuShape<KP>::NMH; ++mh) {
            acc.v[mh] = MU_MFMA(a2c[2 * mh + 1], bl, acc.v[mh]);
            acc.v[mh] = MU_MFMA(a2c[2 * mh + 1], bh, acc.v[mh]);
            acc.v[mh] = MU_MFMA(a2c[2 * mh], bl, acc.v[mh]);
            acc.v[mh] = MU_MFMA(a2c[2 * mh], bh, acc.v[mh]);
        }
    }
}

// one 16-deep chunk c of a tile: the ratio planes of registers 8c .. 8c+7, then the MFMAs that consume them.
//   Kullback-Leibler (BETA1): numerator ratio X / S.
//   Itakura-Saito: numerator ratio X / S^2 into acc, denominator ratio 1 / S into accd (sklearn _nmf.py:580-604).
// v_rcp_f32 is a quarter-rate instruction: one reciprocal serves two elements, 1/s0 = s1 / (s0 s1)
// (s >= eps = 1.2e-7: the product cannot underflow).
template <int KP, bool BETA1>
__device__ __forceinline__ void mu_quotient_accumulate_chunk(MuAcc<KP>& acc, MuAcc<KP>& accd, const f32x16& s, const float (&x)[16],
                                                             int c, const mu_bf16x8 (&a2c)[MuShape<KP>::NA2])
{
    float qn[8], qd[BETA1 ? 1 : 8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int r = 8 * c + 2 * d;
        const float s0 = mu_clamp_eps(s[r]), s1 = mu_clamp_eps(s[r + 1]);
        const float rc = __builtin_amdgcn_rcpf(s0 * s1);
        if constexpr (BETA1) {
            qn[2 * d] = (x[r] * s1) * rc; qn[2 * d + 1] = (x[r + 1] * s0) * rc;
        } else {
            const float i0 = s1 * rc, i1 = s0 * rc;
            qd[2 * d] = i0; qd[2 * d + 1] = i1;
            qn[2 * d] = (x[r] * i0) * i0; qn[2 * d + 1] = (x[r + 1] * i1) * i1;
        }
    }
    mu_bf16x8 bh, bl;
    mu_planes8(qn, bh, bl);
    mu_accumulate8<KP>(acc, a2c, bh, bl);
    if constexpr (!BETA1) {
        mu_planes8(qd, bh, bl);
        mu_accumulate8<KP>(accd, a2c, bh, bl);
    }
}

// ---- H half-step finish: Ht[g][c] *= (num / den)^gamma, planes refreshed.  grid = (ceil(Gs * KP / 256), slots)
template <int KP, bool BETA1>
__global__ __launch_bounds__(256) void mu_h_finish_mfma_kernel(MuBatch mb, int G, int Gs, int nchunks, float l1, float l2)
{
    const MuSlotDev& sd = mb.s[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Gs * KP) return;
    const int g = e / KP, c = e % KP;
    float v = 0.f;
    if (g < G) {
        float n = 0.f, dn = 0.f;
        for (int q = 0; q < nchunks; ++q) n += sd.pnum[(size_t)q * Gs * KP + e];
        if constexpr (BETA1) {
            dn = sd.Wsum[c];
            if (dn == 0.f) dn = 1.0f;                          // sklearn _nmf.py:684-686
        } else {
            for (int q = 0; q < nchunks; ++q) dn += sd.pden[(size_t)q * Gs * KP + e];
        }
        const float hv = sd.Ht[e];
        if (l1 > 0.f) dn += l1;
        if (l2 > 0.f) dn += l2 * hv;
        if (dn == 0.f) dn = MU_EPS;
        float delta = n / dn;
        if (!BETA1) delta = sqrtf(delta);                      // gamma = 1 / (2 - beta) = 1/2
        v = hv * delta;
        if (v < F64_EPS_AS_F32) v = 0.f;                       // sklearn _nmf.py:868-869 (beta <= 1)
    }
    sd.Ht[e] = v;
    mu_u16 hi, lo;
    mu_split_bf16(v, hi, lo);
    sd.Hp_hi[e] = hi; sd.Hp_lo[e] = lo;
    const size_t o = (size_t)c * Gs + (g & ~15) + mu_pos16(g & 15);
    sd.Hc_hi[o] = hi; sd.Hc_hi[(size_t)KP * Gs + o] = lo;
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

// W half-step epilogue: W[row][c] *= numerator / denominator for the wave's 64 cells, planes refreshed.
// C layout: register r <-> component m = 8 (r / 4) + 4 h + r % 4, column (cell) l32
template <int KP, bool BETA1>
__device__ __forceinline__ void mu_w_epilogue(const MuSlotDev& sd, const MuAcc<KP> (&acc)[MuShape<KP>::NJT],
                                              const MuAcc<KP> (&accd)[MuShape<KP>::NJT], int r0,
                                              int l32, int h, int N, int ldxt, float l1, float l2)
{
#pragma unroll
    for (int jt = 0; jt < MuShape<KP>::NJT; ++jt) {
        const int row = r0 + 32 * jt + l32;
        if (row >= N) continue;
        constexpr int NQ = KP / 8;                     // groups of 4 components held by this lane
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c0 = 8 * q + 4 * h;
            float num[4], den[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (KP == 16) {
#pragma unroll
                for (int t = 0; t < 4; ++t) num[t] = acc[jt].v[0][4 * q + t] + acc[jt].v[0][4 * q + t + 8];
                if constexpr (!BETA1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) den[t] = accd[jt].v[0][4 * q + t] + accd[jt].v[0][4 * q + t + 8];
                }
            } else {                                   // component 8 q + 4 h + t = 32 (q / 4) + register 4 (q % 4) + t of that M tile
#pragma unroll
                for (int t = 0; t < 4; ++t) num[t] = acc[jt].v[q / 4][4 * (q % 4) + t];
                if constexpr (!BETA1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) den[t] = accd[jt].v[q / 4][4 * (q % 4) + t];
                }
            }
            float* wp = sd.W + (size_t)row * KP + c0;
            const v4f wv = *reinterpret_cast<const v4f*>(wp);
            const float w[4] = {wv.x, wv.y, wv.z, wv.w};
            float o[4];
            mu_u16 hi[4], lo[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float dn = BETA1 ? sd.Hsum[c0 + t] : den[t];
                if (l1 > 0.f) dn += l1;
                if (l2 > 0.f) dn += l2 * w[t];
                if (dn == 0.f) dn = MU_EPS;
                float delta = num[t] / dn;
                if (!BETA1) delta = sqrtf(delta);              // gamma = 1/2
                o[t] = w[t] * delta;
                if (!BETA1 && o[t] < F64_EPS_AS_F32) o[t] = 0.f;      // sklearn _nmf.py:849-850 (beta < 1 only)
                mu_split_bf16(o[t], hi[t], lo[t]);
                const size_t co = (size_t)(c0 + t) * ldxt + (row & ~15) + mu_pos16(row & 15);
                sd.Wc_hi[co] = hi[t]; sd.Wc_hi[(size_t)KP * ldxt + co] = lo[t];
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

// ---- the workgroup: 8 waves = 4 restarts x 2 halves (rank <= 32; 2 restarts x 4 quarters for ranks 33..64, MuShape) of a
// 128-wide block of the non-reduced dimension.  Per 32-deep
// step it brings in ONE copy of the X block (32 x 128 floats) and one copy of each restart's factor fragments through
// LDS (double buffered, one barrier per step); every wave then reads its operands from LDS in the fragment layouts
// (2 KB of L2 -> CU traffic per tile and restart).
typedef __amdgpu_buffer_rsrc_t mu_rsrc;
template <int KP>
struct MuLds {
    static constexpr int RPW = MuShape<KP>::RPW, TPR = MuShape<KP>::TPR;
    static constexpr int XS = 136;                                   // floats per X row: 4 rows apart = 32 banks apart
    static constexpr int A1ROW = KP * 2 + (KP >= 32 ? 16 : 0);       // bytes per row of the row-major planes
    static constexpr int A2ROW = 80;                                 // bytes per (plane, component): 32 positions + pad
    static constexpr int X_BYTES = 32 * XS * 4;
    static constexpr int A1_BYTES = RPW * 2 * 32 * A1ROW;            // [restart][plane][row]
    static constexpr int A2_BYTES = RPW * 2 * KP * A2ROW;            // [restart][plane][component]
    static constexpr int BUF = X_BYTES + A1_BYTES + A2_BYTES;
    static constexpr int NA1 = (2 * 32 * KP * 2 / 16) / TPR;         // 16-byte chunks per thread and restart
    static constexpr int NA2 = (2 * KP * 4) / TPR;
};

struct MuCoopSrc {                   // what one restart streams (the H half-step streams W, the W half-step H)
    const mu_u16 *a1h, *a1l;         // row-major planes [L][KP]
    const mu_u16* a2;                // component-major planes: hi [KP][cs], lo after it
    int cs;
};

// SECOND = false: only the first product's operands (divergence pass)
template <int KP, bool SECOND>
struct MuCoop {
    using L = MuLds<KP>;
    unsigned char* lds;
    // loader role
    const float* xbase; size_t xld; unsigned long long xbytes;       // X / X^T, floats per reduced row, bytes in all
    int c0;                                                          // first non-reduced column of the workgroup
    MuCoopSrc src; bool src_live;
    int tid;
    u32x4 rx[2], ra1[L::NA1], ra2[L::NA2];

    __device__ __forceinline__ void gload(int t)
    {
        const unsigned long long off = (unsigned long long)t * 32ull * xld * 4ull;
        const unsigned long long rem = xbytes - off;
        const mu_rsrc rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(xbase) + off), 0,
                                                             (int)(rem > 0x7fffffffull ? 0x7fffffffull : rem), 0x00020000);
        const unsigned vo = 4u * (unsigned)((tid >> 5) * xld + c0 + (tid & 31) * 4);
        rx[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
        rx[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (unsigned)(64ull * xld), 0);          // row + 16
        if (src_live) {
            const int tt = tid & (L::TPR - 1);
#pragma unroll
            for (int i = 0; i < L::NA1; ++i) {
                const int q = tt + L::TPR * i, plane = q / (2 * KP * 2), within = q % (2 * KP * 2);
                const mu_u16* pp = (plane ? src.a1l : src.a1h) + (size_t)t * 32 * KP;
                ra1[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(pp) + within * 16);
            }
            if constexpr (SECOND) {
#pragma unroll
                for (int i = 0; i < L::NA2; ++i) {
                    const int q = tt + L::TPR * i, plane = q / (4 * KP), rem2 = q % (4 * KP), comp = rem2 >> 2, part = rem2 & 3;
                    ra2[i] = *reinterpret_cast<const u32x4*>(src.a2 + ((size_t)plane * KP + comp) * src.cs + (size_t)t * 32 + part * 8);
                }
            }
        }
    }
    __device__ __forceinline__ void lstore(int buf)
    {
        unsigned char* b = lds + buf * L::BUF;
        float* xs = reinterpret_cast<float*>(b);
        *reinterpret_cast<u32x4*>(xs + (tid >> 5) * L::XS + (tid & 31) * 4) = rx[0];
        *reinterpret_cast<u32x4*>(xs + ((tid >> 5) + 16) * L::XS + (tid & 31) * 4) = rx[1];
        if (src_live) {
            const int tt = tid & (L::TPR - 1), rsi = tid / L::TPR;
#pragma unroll
            for (int i = 0; i < L::NA1; ++i) {
                const int q = tt + L::TPR * i, plane = q / (2 * KP * 2), within = q % (2 * KP * 2);
                const int row = within / (KP / 8), part = within % (KP / 8);
                *reinterpret_cast<u32x4*>(b + L::X_BYTES + ((rsi * 2 + plane) * 32 + row) * L::A1ROW + part * 16) = ra1[i];
            }
            if constexpr (SECOND) {
#pragma unroll
                for (int i = 0; i < L::NA2; ++i) {
                    const int q = tt + L::TPR * i, plane = q / (4 * KP), rem2 = q % (4 * KP), comp = rem2 >> 2, part = rem2 & 3;
                    *reinterpret_cast<u32x4*>(b + L::X_BYTES + L::A1_BYTES + ((rsi * 2 + plane) * KP + comp) * L::A2ROW + part * 16) = ra2[i];
                }
            }
        }
    }
    // fragment reads of the computing wave (restart rs, half gs)
    __device__ __forceinline__ void read_x(int buf, int gs, int jt, int l32, int h, float (&x)[16]) const
    {
        const float* xs = reinterpret_cast<const float*>(lds + buf * L::BUF) + 4 * h * L::XS + gs * 32 * MuShape<KP>::NJT + jt * 32 + l32;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = xs[(8 * (r >> 2) + (r & 3)) * L::XS];
    }
    __device__ __forceinline__ void read_a1(int buf, int rs, int l32, int h, mu_bf16x8 (&a1)[KP / 16][2]) const
    {
        const unsigned char* b = lds + buf * L::BUF + L::X_BYTES;
#pragma unroll
        for (int ks = 0; ks < KP / 16; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                a1[ks][p] = __builtin_bit_cast(mu_bf16x8, *reinterpret_cast<const u32x4*>(b + ((rs * 2 + p) * 32 + l32) * L::A1ROW + (16 * ks + 8 * h) * 2));
    }
    __device__ __forceinline__ void read_a2(int buf, int rs, int l32, int h, int c, mu_bf16x8 (&a2c)[MuShape<KP>::NA2]) const
    {
        const unsigned char* b = lds + buf * L::BUF + L::X_BYTES + L::A1_BYTES;
        if constexpr (KP == 16)
            a2c[0] = __builtin_bit_cast(mu_bf16x8, *reinterpret_cast<const u32x4*>(b + ((rs * 2 + (l32 >> 4)) * 16 + (l32 & 15)) * L::A2ROW + (16 * c + 8 * h) * 2));
        else {
#pragma unroll
            for (int mh = 0; mh < MuShape<KP>::NMH; ++mh)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    a2c[2 * mh + p] = __builtin_bit_cast(mu_bf16x8, *reinterpret_cast<const u32x4*>(b + ((rs * 2 + p) * KP + 32 * mh + l32) * L::A2ROW + (16 * c + 8 * h) * 2));
        }
    }
};

// steps [t0, t1): numerators of the wave's two tiles into acc (MODE 0) or the divergence partial into dv (MODE 1)
template <int KP, int MODE, bool BETA1>
__device__ __forceinline__ void mu_coop_loop(MuCoop<KP, MODE == 0>& co, bool active, int rs, int gs, int l32, int h,
                                             const mu_bf16x8 (&b1)[MuShape<KP>::NJT][KP / 16][2], MuAcc<KP> (&acc)[MuShape<KP>::NJT],
                                             MuAcc<KP> (&accd)[MuShape<KP>::NJT], double& dv, int t0, int t1)
{
    if (t0 >= t1) return;
    co.gload(t0);
    co.lstore(0);
    __syncthreads();
    int buf = 0;
    for (int t = t0; t < t1; ++t, buf ^= 1) {
        const bool more = t + 1 < t1;
        if (more) co.gload(t + 1);
        if (active) {
            mu_bf16x8 a1[KP / 16][2];
            co.read_a1(buf, rs, l32, h, a1);
            if constexpr (MODE == 0) {
#pragma unroll
                for (int jt = 0; jt < MuShape<KP>::NJT; ++jt) {
                    float x[16];
                    co.read_x(buf, gs, jt, l32, h, x);
                    const f32x16 s = mu_product<KP / 16>(a1, b1[jt]);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        mu_bf16x8 a2c[MuShape<KP>::NA2];
                        co.read_a2(buf, rs, l32, h, c, a2c);
                        mu_quotient_accumulate_chunk<KP, BETA1>(acc[jt], accd[jt], s, x, c, a2c);
                    }
                }
            } else {
                float part = 0.f;
#pragma unroll
                for (int jt = 0; jt < MuShape<KP>::NJT; ++jt) {
                    float x[16];
                    co.read_x(buf, gs, jt, l32, h, x);
                    const f32x16 s = mu_product<KP / 16>(a1, b1[jt]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // hardware reciprocal and logarithm (1 ulp each): the sum below carries ~1e8 terms of mixed
                        // sign, a relative 1e-7 per term is far inside the 1e-4 convergence test it feeds
                        // KL: X log(X / S) - X;  IS: X / S - log(X / S)   (sklearn _nmf.py:125-141, 143-147)
                        const float dq = x[r] * __builtin_amdgcn_rcpf(mu_clamp_eps(s[r]));
                        const float lg = __builtin_amdgcn_logf(dq) * 0.69314718056f;     // v_log_f32 (log2); dq is never denormal
                        const float t = BETA1 ? (x[r] * lg - x[r]) : (dq - lg);
                        part += (x[r] > MU_EPS) ? t : 0.f;
                    }
                }
                dv += (double)part;
            }
        }
        if (more) co.lstore(buf ^ 1);
        __syncthreads();
    }
}

// ---- H half-step partials, cooperative.  grid = (gene blocks of 128, row chunks, restart groups of 4), block = 512
template <int KP, bool BETA1>
__global__ __launch_bounds__(512) void mu_h_coop_kernel(const float* __restrict__ X, int ldx, int Np, int Gs,
                                                        MuBatch mb, int tiles_per_chunk, int nchunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mu_lds[];
    constexpr int KS = KP / 16;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    using SH = MuShape<KP>;
    const int rs = wave % SH::RPW, gs = wave / SH::RPW;
    const int slot = blockIdx.z * SH::RPW + rs;
    const bool active = slot < mb.n;
    const MuSlotDev& sd = mb.s[active ? slot : 0];
    const int g0 = blockIdx.x * 128 + gs * 32 * SH::NJT;
    const int chunk = blockIdx.y;
    const int ntiles = Np / 32;
    const int rt0 = chunk * tiles_per_chunk, rt1 = min(ntiles, rt0 + tiles_per_chunk);

    MuCoop<KP, true> co;
    co.lds = mu_lds; co.tid = tid;
    co.xbase = X; co.xld = (size_t)ldx; co.xbytes = (unsigned long long)(Np + 1) * ldx * 4ull;     // the slack row included
    co.c0 = blockIdx.x * 128;
    {
        const int ls = blockIdx.z * SH::RPW + tid / SH::TPR;         // the restart this thread loads for
        co.src_live = ls < mb.n;
        const MuSlotDev& sl = mb.s[co.src_live ? ls : 0];
        co.src = MuCoopSrc{sl.Wp_hi, sl.Wp_lo, sl.Wc_hi, Np};
    }
    mu_bf16x8 b1[SH::NJT][KS][2];
#pragma unroll
    for (int jt = 0; jt < SH::NJT; ++jt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t o = (size_t)(g0 + 32 * jt + l32) * KP + 16 * ks + 8 * h;
            b1[jt][ks][0] = mu_ld8(sd.Hp_hi + o);
            b1[jt][ks][1] = mu_ld8(sd.Hp_lo + o);
        }
    MuAcc<KP> acc[SH::NJT], accd[SH::NJT];
#pragma unroll
    for (int jt = 0; jt < SH::NJT; ++jt)
#pragma unroll
        for (int mh = 0; mh < SH::NMH; ++mh)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[jt].v[mh][r] = 0.f; accd[jt].v[mh][r] = 0.f; }
    double dv = 0.0;
    mu_coop_loop<KP, 0, BETA1>(co, active, rs, gs, l32, h, b1, acc, accd, dv, rt0, rt1);
    if (!active) return;
    // C layout: register r <-> row m = 8 (r / 4) + 4 h + r % 4, column (gene) l32
    auto store = [&](float* base, const MuAcc<KP> (&a)[SH::NJT]) {
#pragma unroll
        for (int jt = 0; jt < SH::NJT; ++jt) {
            float* pn = base + ((size_t)chunk * Gs + g0 + 32 * jt + l32) * KP;
            if constexpr (KP == 16) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    v4f v;
                    v.x = a[jt].v[0][4 * q + 0] + a[jt].v[0][4 * q + 8]; v.y = a[jt].v[0][4 * q + 1] + a[jt].v[0][4 * q + 9];
                    v.z = a[jt].v[0][4 * q + 2] + a[jt].v[0][4 * q + 10]; v.w = a[jt].v[0][4 * q + 3] + a[jt].v[0][4 * q + 11];
                    *reinterpret_cast<v4f*>(pn + 8 * q + 4 * h) = v;
                }
            } else {
#pragma unroll
                for (int mh = 0; mh < SH::NMH; ++mh)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v4f v = {a[jt].v[mh][4 * q], a[jt].v[mh][4 * q + 1], a[jt].v[mh][4 * q + 2], a[jt].v[mh][4 * q + 3]};
                        *reinterpret_cast<v4f*>(pn + 32 * mh + 8 * q + 4 * h) = v;
                    }
            }
        }
    };
    store(sd.pnum, acc);
    if constexpr (!BETA1) store(sd.pden, accd);
}

// ---- W half-step (MODE 0) / divergence (MODE 1), cooperative.  grid = (cell blocks of 128, 1, restart groups of 4)
template <int KP, int MODE, bool BETA1>
__global__ __launch_bounds__(512) void mu_w_coop_kernel(const float* __restrict__ Xt, int ldxt, int N, int Gs,
                                                        MuBatch mb, float l1, float l2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mu_lds[];
    constexpr int KS = KP / 16;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    using SH = MuShape<KP>;
    const int rs = wave % SH::RPW, gs = wave / SH::RPW;
    const int slot = blockIdx.z * SH::RPW + rs;
    const bool active = slot < mb.n;
    const MuSlotDev& sd = mb.s[active ? slot : 0];
    const int r0 = blockIdx.x * 128 + gs * 32 * SH::NJT;             // (ldxt = N_pad is a multiple of 128)

    MuCoop<KP, MODE == 0> co;
    co.lds = mu_lds; co.tid = tid;
    co.xbase = Xt; co.xld = (size_t)ldxt; co.xbytes = (unsigned long long)Gs * ldxt * 4ull;
    co.c0 = blockIdx.x * 128;
    {
        const int ls = blockIdx.z * SH::RPW + tid / SH::TPR;
        co.src_live = ls < mb.n;
        const MuSlotDev& sl = mb.s[co.src_live ? ls : 0];
        co.src = MuCoopSrc{sl.Hp_hi, sl.Hp_lo, sl.Hc_hi, Gs};
    }
    mu_bf16x8 b1[SH::NJT][KS][2];
#pragma unroll
    for (int jt = 0; jt < SH::NJT; ++jt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t o = (size_t)(r0 + 32 * jt + l32) * KP + 16 * ks + 8 * h;
            b1[jt][ks][0] = mu_ld8(sd.Wp_hi + o);
            b1[jt][ks][1] = mu_ld8(sd.Wp_lo + o);
        }
    MuAcc<KP> acc[SH::NJT], accd[SH::NJT];
#pragma unroll
    for (int jt = 0; jt < SH::NJT; ++jt)
#pragma unroll
        for (int mh = 0; mh < SH::NMH; ++mh)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[jt].v[mh][r] = 0.f; accd[jt].v[mh][r] = 0.f; }
    double dv = 0.0;
    mu_coop_loop<KP, MODE, BETA1>(co, active, rs, gs, l32, h, b1, acc, accd, dv, 0, Gs / 32);
    if (!active) return;
    if constexpr (MODE == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dv += __shfl_xor(dv, o, 64);
        if (lane == 0 && r0 < N) sd.divpart[blockIdx.x * (8 / SH::RPW) + gs] = dv;    // one partial per strip of 32 NJT cells
        return;
    } else {
        mu_w_epilogue<KP, BETA1>(sd, acc, accd, r0, l32, h, N, ldxt, l1, l2);
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

// X [N_pad][ldx] -> Xt [Gs][ldxt] (zero beyond N x G).  grid = (ceil(ldxt / 32), ceil(Gs / 32)), block = (32, 8): the CELL
// dimension rides on grid.x (2^31 - 1 blocks), the gene dimension (<= 2^24 / 32 ... in practice a few hundred) on grid.y
__global__ __launch_bounds__(256) void mu_transpose_kernel(const float* __restrict__ X, int ldx, int N, int G,
                                                           float* __restrict__ Xt, int ldxt, int Gs)
{
    __shared__ float tile[32][33];
    const int g0 = blockIdx.y * 32, r0 = blockIdx.x * 32;
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
