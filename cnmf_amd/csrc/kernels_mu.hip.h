// Multiplicative-update NMF for beta_loss in {Kullback-Leibler (beta=1), Itakura-Saito (beta=0)}
// -- the solver the reference keeps when beta_loss != 'frobenius' (cnmf.py:618-631).
// Restates sklearn/decomposition/_nmf.py:526-631 (_multiplicative_update_w), :634-728
// (_multiplicative_update_h), :84-194 (_beta_divergence) for dense X, without ever
// materialising the N x G matrices WH and X / WH: every workgroup recomputes its slice
// of W.H on the fly from the rank-k factors (k <= 32), so X is the only large operand read.
//
// Layouts: W [N][KP] row-major, Ht [G][KP] row-major (KP = k rounded up to 8/16/32, padding 0).
//   W half-step : one lane per cell, genes looped (X row segments streamed as float4)
//   H half-step : one lane per gene (X loads coalesced across lanes), cells looped in chunks,
//                 W rows come in as wave-uniform scalars; per-chunk partials reduced in order
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_gemm.hip.h"

namespace cnmf {

constexpr float MU_EPS = 1.1920928955078125e-07f;      // np.finfo(np.float32).eps (sklearn EPSILON)
constexpr float F64_EPS_AS_F32 = 2.220446049250313e-16f;

// numer / denom contributions of one element; BETA1: KL, else IS
template <bool BETA1>
__device__ __forceinline__ void mu_ratio(float x, float wh, float& r, float& d)
{
    const float whs = fmaxf(wh, MU_EPS);
    if (BETA1) { r = x / whs; d = 0.f; }
    else { const float inv = 1.0f / whs; r = x * inv * inv; d = inv; }   // X*WH^-2 , WH^-1 (clamped)
}

// ---- W half-step: W[i][c] *= (num[i][c] / den)^gamma
//   num[i][c] = sum_g R[i][g] Ht[g][c];  den = Hsum[c] (KL)  or  sum_g WH^-1 Ht[g][c] (IS)
// block = 256 threads = 64 cells x 4 gene quarters; grid.x = ceil(N/64).  X goes through an LDS
// tile [64 cells][64 genes] (loaded coalesced, 256 B per row; read back one row per lane, stride 65
// -> conflict-free) and the matching Ht tile [64 genes][KP] is read back as wave-wide broadcasts.
// NQ waves per workgroup share the 64 cells (each takes 64 / NQ genes of every gene tile): NQ = 8 (512 threads) doubles
// the waves per SIMD of this latency-bound kernel (PMC, round 2: 3 waves per SIMD, 49 % of the wave cycles waiting).
template <int KP, bool BETA1, int NQ = 8>
__global__ __launch_bounds__(64 * NQ) void mu_w_kernel(const float* __restrict__ X, int ldx, int N, int G,
                                                       float* __restrict__ W, const float* __restrict__ Ht,
                                                       const float* __restrict__ Hsum, float l1, float l2)
{
    constexpr int NT = 64 * NQ, GPW = 64 / NQ;           // threads; genes per wave and tile
    constexpr int XS = 64 * 65, HS = 64 * KP;
    constexpr int RED = (BETA1 ? 1 : 2) * (NQ - 1) * 64 * (KP + 1);
    __shared__ __attribute__((aligned(16))) float lds[(XS + HS) > RED ? (XS + HS) : RED];
    float* xs = lds;                 // [64][65]
    float* hs = lds + XS;            // [64][KP]
    const int tid = threadIdx.x, ci = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave index: provably uniform
    const int i0 = blockIdx.x * 64;
    const int i = i0 + ci;
    const bool live = i < N;
    float w[KP], num[KP], den[KP];
#pragma unroll
    for (int c = 0; c < KP; ++c) { w[c] = live ? W[(size_t)i * KP + c] : 0.f; num[c] = 0.f; den[c] = 0.f; }
    const int lrow = tid >> 4, lc4 = (tid & 15) * 4;              // tile loader: 16 float4 per row
    for (int g0 = 0; g0 < ldx; g0 += 64) {
        // X and Ht are zero padded to ldx (a multiple of 32) columns / rows; a padded gene adds 0
#pragma unroll
        for (int j = 0; j < 64 / (NT / 16); ++j) {
            const int r = lrow + (NT / 16) * j;
            const int row = min(i0 + r, N - 1);
            v4f v = v4f{0.f, 0.f, 0.f, 0.f};
            if (g0 + lc4 < ldx) v = *reinterpret_cast<const v4f*>(X + (size_t)row * ldx + g0 + lc4);
            float* d = xs + r * 65 + lc4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        for (int e = tid * 4; e < HS; e += NT * 4) {
            const int g = g0 + e / KP;
            v4f v = v4f{0.f, 0.f, 0.f, 0.f};
            if (g < ldx) v = *reinterpret_cast<const v4f*>(Ht + (size_t)g0 * KP + e);
            *reinterpret_cast<v4f*>(hs + e) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int u = 0; u < GPW; ++u) {
            const int gl = q * GPW + u;
            const float x = xs[ci * 65 + gl];
            const float* h = hs + gl * KP;                        // same address for the whole wave
            float wh = 0.f;
#pragma unroll
            for (int c = 0; c < KP; ++c) wh = fmaf(w[c], h[c], wh);
            float r, d;
            mu_ratio<BETA1>(x, wh, r, d);
#pragma unroll
            for (int c = 0; c < KP; ++c) { num[c] = fmaf(r, h[c], num[c]); if (!BETA1) den[c] = fmaf(d, h[c], den[c]); }
        }
        __syncthreads();
    }
    // waves 1..NQ-1 -> wave 0, added in wave order
    float (*red)[NQ - 1][64][KP + 1] = reinterpret_cast<float (*)[NQ - 1][64][KP + 1]>(lds);
    if (q > 0) {
#pragma unroll
        for (int c = 0; c < KP; ++c) { red[0][q - 1][ci][c] = num[c]; if (!BETA1) red[BETA1 ? 0 : 1][q - 1][ci][c] = den[c]; }
    }
    __syncthreads();
    if (q == 0 && live) {
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            float n = num[c], dsum = den[c];
#pragma unroll
            for (int z = 0; z < NQ - 1; ++z) { n += red[0][z][ci][c]; if (!BETA1) dsum += red[BETA1 ? 0 : 1][z][ci][c]; }
            float dn = BETA1 ? Hsum[c] : dsum;
            if (l1 > 0.f) dn += l1;
            if (l2 > 0.f) dn += l2 * w[c];
            if (dn == 0.f) dn = MU_EPS;
            float delta = n / dn;
            if (!BETA1) delta = sqrtf(delta);             // gamma = 1/(2-beta) = 1/2
            float v = w[c] * delta;
            if (!BETA1 && v < F64_EPS_AS_F32) v = 0.f;    // sklearn _nmf.py:849-850 (beta < 1 only)
            W[(size_t)i * KP + c] = v;
        }
    }
}

// ---- H half-step partials: lanes along genes, one chunk of cells per blockIdx.y
//   pnum[chunk][g][c] = sum_{i in chunk} R[i][g] W[i][c];  pden likewise with WH^-1 (IS)
template <int KP, bool BETA1>
__global__ __launch_bounds__(256) void mu_h_partial_kernel(const float* __restrict__ X, int ldx, int N, int G,
                                                           const float* __restrict__ W, const float* __restrict__ Ht,
                                                           int rows_per_chunk, float* __restrict__ pnum,
                                                           float* __restrict__ pden)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < G;
    float h[KP], num[KP], den[KP];
#pragma unroll
    for (int c = 0; c < KP; ++c) { h[c] = live ? Ht[(size_t)g * KP + c] : 0.f; num[c] = 0.f; den[c] = 0.f; }
    const int ib = blockIdx.y * rows_per_chunk, ie = min(N, ib + rows_per_chunk);
    const int gc = live ? g : 0;
    int i = ib;
    for (; i + 4 <= ie; i += 4) {                         // 4 cells per trip: loads issued before use
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = X[(size_t)(i + u) * ldx + gc];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* wr = W + (size_t)(i + u) * KP;   // wave-uniform -> scalar loads
            float wh = 0.f;
#pragma unroll
            for (int c = 0; c < KP; ++c) wh = fmaf(wr[c], h[c], wh);
            float r, d;
            mu_ratio<BETA1>(live ? x[u] : 0.f, wh, r, d);
#pragma unroll
            for (int c = 0; c < KP; ++c) { num[c] = fmaf(r, wr[c], num[c]); if (!BETA1) den[c] = fmaf(d, wr[c], den[c]); }
        }
    }
    for (; i < ie; ++i) {
        const float x = live ? X[(size_t)i * ldx + g] : 0.f;
        const float* wr = W + (size_t)i * KP;
        float wh = 0.f;
#pragma unroll
        for (int c = 0; c < KP; ++c) wh = fmaf(wr[c], h[c], wh);
        float r, d;
        mu_ratio<BETA1>(x, wh, r, d);
#pragma unroll
        for (int c = 0; c < KP; ++c) { num[c] = fmaf(r, wr[c], num[c]); if (!BETA1) den[c] = fmaf(d, wr[c], den[c]); }
    }
    if (live) {
        float* pn = pnum + ((size_t)blockIdx.y * G + g) * KP;
#pragma unroll
        for (int c = 0; c < KP; ++c) pn[c] = num[c];
        if (!BETA1) {
            float* pd = pden + ((size_t)blockIdx.y * G + g) * KP;
#pragma unroll
            for (int c = 0; c < KP; ++c) pd[c] = den[c];
        }
    }
}

// ---- H half-step finish: Ht[g][c] *= (num/den)^gamma, clamp < float64 eps -> 0 (beta <= 1)
template <int KP, bool BETA1>
__global__ void mu_h_finish_kernel(float* __restrict__ Ht, int G, const float* __restrict__ pnum,
                                   const float* __restrict__ pden, int nchunks,
                                   const float* __restrict__ Wsum, float l1, float l2)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= G * KP) return;
    const int c = e % KP;
    float n = 0.f, dn = 0.f;
    for (int q = 0; q < nchunks; ++q) { n += pnum[(size_t)q * G * KP + e]; if (!BETA1) dn += pden[(size_t)q * G * KP + e]; }
    if (BETA1) { dn = Wsum[c]; if (dn == 0.f) dn = 1.0f; }     // sklearn _nmf.py:684-686
    const float hv = Ht[e];
    if (l1 > 0.f) dn += l1;
    if (l2 > 0.f) dn += l2 * hv;
    if (dn == 0.f) dn = MU_EPS;
    float delta = n / dn;
    if (!BETA1) delta = sqrtf(delta);
    float v = hv * delta;
    if (v < F64_EPS_AS_F32) v = 0.f;                           // sklearn _nmf.py:868-869 (beta <= 1)
    Ht[e] = v;
}

// column sums: out[c] = sum_r M[r][c] for M [R][KP]   (Hsum over genes, Wsum over cells).
// Two levels, fixed order: `nb` blocks write partial sums [nb][KP], block 0 of the second launch adds them.
template <int KP>
__global__ __launch_bounds__(256) void mu_colsum_part_kernel(const float* __restrict__ M, int R, double* __restrict__ part)
{
    __shared__ double red[256 / KP > 0 ? 256 / KP : 1][KP];
    constexpr int RG = 256 / KP;                  // row groups per block (KP = 8,16,32,64 -> 32,16,8,4)
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
        part[(size_t)blockIdx.x * KP + c] = t;
    }
}
template <int KP>
__global__ void mu_colsum_final_kernel(const double* __restrict__ part, int nb, float* __restrict__ out)
{
    const int c = threadIdx.x;
    if (c >= KP) return;
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += part[(size_t)b * KP + c];
    out[c] = (float)t;
}

// ---- beta divergence partials (sklearn _nmf.py:84-194), entries with X > EPSILON only:
//   KL: sum X log(X/WH) - sum X   (+ sum WH added by the host from the column sums)
//   IS: sum X/WH - log(X/WH)      (- N*G subtracted by the host)
template <int KP, bool BETA1>
__global__ __launch_bounds__(256) void mu_divergence_kernel(const float* __restrict__ X, int ldx, int N, int G,
                                                            const float* __restrict__ W, const float* __restrict__ Ht,
                                                            int rows_per_chunk, double* __restrict__ part)
{
    __shared__ double red[4];
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < G;
    float h[KP];
#pragma unroll
    for (int c = 0; c < KP; ++c) h[c] = live ? Ht[(size_t)g * KP + c] : 0.f;
    const int ib = blockIdx.y * rows_per_chunk, ie = min(N, ib + rows_per_chunk);
    double acc = 0.0;
    for (int i = ib; i < ie; ++i) {
        const float x = live ? X[(size_t)i * ldx + g] : 0.f;
        const float* wr = W + (size_t)i * KP;
        float wh = 0.f;
#pragma unroll
        for (int c = 0; c < KP; ++c) wh = fmaf(wr[c], h[c], wh);
        if (x > MU_EPS) {
            const float div = x / fmaxf(wh, MU_EPS);
            acc += BETA1 ? (double)(x * logf(div) - x) : (double)(div - logf(div));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// pack / unpack between the component-major stores ([k][L]) and the padded row-major MU layout ([L][KP])
__global__ void mu_pack_kernel(const float* __restrict__ cm, int k, int L, float* __restrict__ rm, int KP)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L * KP) return;
    const int r = e / KP, c = e % KP;
    rm[e] = (c < k) ? cm[(size_t)c * L + r] : 0.f;
}
__global__ void mu_unpack_kernel(const float* __restrict__ rm, int k, int L, int KP, float* __restrict__ out,
                                 int transpose /*1: out [k][L], 0: out [L][k]*/)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L * k) return;
    const int r = e / k, c = e % k;
    const float v = rm[(size_t)r * KP + c];
    if (transpose) out[(size_t)c * L + r] = v; else out[(size_t)r * k + c] = v;
}
// row-major source [L][k] -> padded [L][KP]
__global__ void mu_pack_rm_kernel(const float* __restrict__ src, int k, int L, float* __restrict__ rm, int KP)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L * KP) return;
    const int r = e / KP, c = e % KP;
    rm[e] = (c < k) ? src[(size_t)r * k + c] : 0.f;
}
__global__ void mu_fill_kernel(float* __restrict__ rm, int k, int L, int KP, float v)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L * KP) return;
    rm[e] = ((e % KP) < k) ? v : 0.f;
}

}  // namespace cnmf
