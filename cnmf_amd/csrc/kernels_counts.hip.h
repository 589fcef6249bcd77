// Count structure of the data matrix (gfx950 only).
//
// cNMF factorises  X = counts[:, HVGs] / std_g  (cnmf.py:540-548): every column of X is an INTEGER
// matrix times one per-gene constant,  X[i][g] = n[i][g] * d[g].  Counts of high-variance genes are
// small: n <= 256 is exactly representable in ONE bf16 plane; the rare larger count (up to 65 535) is
// n = lo + 256 hi with a second, almost empty plane whose (256 rows x 16 k) blocks are flagged so that the
// GEMM touches it only where it is non-zero.  Folding d into the factor side,
//     pass A :  X . H^T   = n . (H * d)^T          (scale the columns of H before splitting it)
//     pass B :  X^T . W   = d * (n^T . W)          (scale the rows of the product afterwards)
// the f32-accurate product needs 3 bf16 MFMAs (the factor's three planes x one integer plane) instead of
// 6, every partial product is exact, and X costs 2 bytes per element instead of 6.
// The structure is DETECTED on the device from the resident float32 matrix (so it also applies to a
// matrix that arrives already normalised, as the reference hands it to scikit-learn); a matrix without
// it simply keeps the general three-plane path.  Included by cnmf_hip.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_gemm3.hip.h"

namespace cnmf {

constexpr int CNT_ROWS = 256;           // rows per partial of the column passes
constexpr int CNT_MAXMULT = 8;          // the smallest positive entry may be this many counts
constexpr float CNT_MAX = 65535.0f;     // lo + 256 hi, both planes exact in bf16 (lo <= 256, hi <= 255)

// part[chunk][g] = smallest positive entry of column g among the chunk's rows (+inf if none)
__global__ __launch_bounds__(256) void col_minpos_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                         float* __restrict__ part)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const int r0 = blockIdx.y * CNT_ROWS, r1 = min(N, r0 + CNT_ROWS);
    float m = __builtin_inff();
    for (int r = r0; r < r1; ++r) { const float x = X[(size_t)r * ld + g]; if (x > 0.f) m = fminf(m, x); }
    part[(size_t)blockIdx.y * G + g] = m;
}

__global__ __launch_bounds__(256) void col_min_combine_kernel(const float* __restrict__ part, int chunks, int G,
                                                              float* __restrict__ out)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    float m = __builtin_inff();
    for (int c = 0; c < chunks; ++c) m = fminf(m, part[(size_t)c * G + g]);
    out[g] = isinf(m) ? 0.f : m;
}

// fail[g] bit (m-1) set  <=>  for multiplier m some entry x of column g is not (an integer <= 256) * vmin/m
__global__ __launch_bounds__(256) void count_check_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                          const float* __restrict__ vmin, unsigned* __restrict__ fail)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const float v = vmin[g];
    if (v <= 0.f) return;                                 // empty column: n = 0 everywhere
    const int r0 = blockIdx.y * CNT_ROWS, r1 = min(N, r0 + CNT_ROWS);
    unsigned bad = 0;
    for (int r = r0; r < r1; ++r) {
        const float x = X[(size_t)r * ld + g];
        if (x <= 0.f) { if (x < 0.f) bad = 0xFFu; continue; }
        const float q = x / v;                            // in units of the smallest entry
#pragma unroll
        for (int m = 1; m <= CNT_MAXMULT; ++m) {
            const float t = q * (float)m, n = rintf(t);
            // float32 input: t carries ~2^-22 relative error -> 1e-3 absolute below ~2000, relative above
            if (n < 1.f || n > CNT_MAX || fabsf(t - n) > fmaxf(1e-3f, 4e-7f * n)) bad |= 1u << (m - 1);
        }
    }
    if (bad) atomicOr(&fail[g], bad);
}

// partial sums for the scale: sx = sum x, sn = sum n  (float64), n = rint(x / unit[g])
__global__ __launch_bounds__(256) void count_sums_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                         const float* __restrict__ unit, double* __restrict__ psx,
                                                         double* __restrict__ psn)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const float u = unit[g];
    const int r0 = blockIdx.y * CNT_ROWS, r1 = min(N, r0 + CNT_ROWS);
    double sx = 0.0, sn = 0.0;
    if (u > 0.f)
        for (int r = r0; r < r1; ++r) { const float x = X[(size_t)r * ld + g]; sx += (double)x; sn += (double)rintf(x / u); }
    psx[(size_t)blockIdx.y * G + g] = sx;
    psn[(size_t)blockIdx.y * G + g] = sn;
}

__global__ __launch_bounds__(256) void count_scale_kernel(const double* __restrict__ psx, const double* __restrict__ psn,
                                                          int chunks, int G, int G_pad, double* __restrict__ scale)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G_pad) return;
    double sx = 0.0, sn = 0.0;
    if (g < G) for (int c = 0; c < chunks; ++c) { sx += psx[(size_t)c * G + g]; sn += psn[(size_t)c * G + g]; }
    scale[g] = sn > 0.0 ? sx / sn : 0.0;
}

// ---- integer planes (ONE bf16 plane), block-major with row tiles of TR rows:
//      [row tile][16-k block][row][16 bf16] = 32 contiguous bytes per row and block
// rows = cells, k = genes (pass A's operand).  One thread per (row, block).
// n = lo + 256 hi: lo in [0, 256] (256 itself stays in lo), hi in [0, 255]
__device__ __forceinline__ void count_lo_hi(float n, float& lo, float& hi)
{
    if (n <= 256.f) { lo = n; hi = 0.f; }
    else { hi = floorf(n * (1.0f / 256.0f)); lo = n - 256.f * hi; hi *= 256.f; }      // hi plane holds 256 hi
}

__device__ __forceinline__ void store_plane_row(unsigned short* d, const unsigned short* p)
{
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        u32x4 w;
        w.x = p[8 * hf + 0] | ((unsigned)p[8 * hf + 1] << 16); w.y = p[8 * hf + 2] | ((unsigned)p[8 * hf + 3] << 16);
        w.z = p[8 * hf + 4] | ((unsigned)p[8 * hf + 5] << 16); w.w = p[8 * hf + 6] | ((unsigned)p[8 * hf + 7] << 16);
        *reinterpret_cast<u32x4*>(d + hf * 8) = w;
    }
}

__global__ __launch_bounds__(256) void count_planes_kernel(const float* __restrict__ X, int ld, int N, int G, int rows_pad,
                                                           int K, int TR, const float* __restrict__ unit,
                                                           unsigned short* __restrict__ dst,
                                                           unsigned short* __restrict__ dst_hi,
                                                           unsigned int* __restrict__ hiflag)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Kb = K / 16;
    if (t >= (long long)rows_pad * Kb) return;
    const int row = (int)(t / Kb), kb = (int)(t % Kb);
    unsigned short p[16], ph[16];
    bool any_hi = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int g = kb * 16 + i;
        float n = 0.f, lo, hi;
        if (row < N && g < G) { const float u = unit[g]; if (u > 0.f) n = rintf(X[(size_t)row * ld + g] / u); }
        count_lo_hi(n, lo, hi);
        p[i] = bf16_rne(lo); ph[i] = bf16_rne(hi);           // both exact
        any_hi |= hi != 0.f;
    }
    const size_t blk = (size_t)(row / TR) * Kb + kb;
    store_plane_row(dst + (blk * TR + (row % TR)) * 16, p);
    if (dst_hi) {
        store_plane_row(dst_hi + (blk * TR + (row % TR)) * 16, ph);
        // (tile, block) has a non-zero second plane: one bit per block, (Kb + 31) / 32 words per tile row
        if (any_hi) atomicOr(&hiflag[(size_t)(row / TR) * ((Kb + 31) / 32) + (kb >> 5)], 1u << (kb & 31));
    }
}

// rows = genes, k = cells (pass B's operand).  One thread per (gene row j, block); lanes run along j.
__global__ __launch_bounds__(256) void count_planes_transpose_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                                     int rows_pad, int K, int TR,
                                                                     const float* __restrict__ unit,
                                                                     unsigned short* __restrict__ dst,
                                                                     unsigned short* __restrict__ dst_hi,
                                                                     unsigned int* __restrict__ hiflag)
{
    const int j = blockIdx.y * 256 + threadIdx.x;       // (grid = (16-k blocks, row groups): the long dimension on x -- no 65 535 limit)
    const int kb = blockIdx.x;
    if (j >= rows_pad) return;
    const int Kb = K / 16;
    const float u = (j < G) ? unit[j] : 0.f;
    unsigned short p[16], ph[16];
    bool any_hi = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = kb * 16 + i;
        float n = 0.f, lo, hi;
        if (u > 0.f && c < N) n = rintf(X[(size_t)c * ld + j] / u);
        count_lo_hi(n, lo, hi);
        p[i] = bf16_rne(lo); ph[i] = bf16_rne(hi);
        any_hi |= hi != 0.f;
    }
    const size_t blk = (size_t)(j / TR) * Kb + kb;
    store_plane_row(dst + (blk * TR + (j % TR)) * 16, p);
    if (dst_hi) {
        store_plane_row(dst_hi + (blk * TR + (j % TR)) * 16, ph);
        if (any_hi) atomicOr(&hiflag[(size_t)(j / TR) * ((Kb + 31) / 32) + (kb >> 5)], 1u << (kb & 31));
    }
}

// does any column need the second plane?  (max n > 256)
__global__ __launch_bounds__(256) void count_max_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                        const float* __restrict__ unit, unsigned* __restrict__ any_big)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const float u = unit[g];
    if (u <= 0.f) return;
    const int r0 = blockIdx.y * CNT_ROWS, r1 = min(N, r0 + CNT_ROWS);
    bool big = false;
    for (int r = r0; r < r1; ++r) big |= rintf(X[(size_t)r * ld + g] / u) > 256.f;
    if (big) atomicOr(any_big, 1u);
}

}  // namespace cnmf
