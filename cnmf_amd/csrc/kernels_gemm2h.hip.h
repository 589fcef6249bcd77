// Count-structured data on the f16 matrix pipe: 2 MFMAs per f32-class product (gfx950 only).
//
// kernels_counts.hip.h explains the structure X[i][g] = n[i][g] * d[g] (n integer).  gemm3c_* (kernels_gemm3.hip.h)
// multiplies the integer plane (bf16: n <= 256 exact) with the factor's THREE bf16 planes: 3 MFMAs per product.
// Here both operands are f16 (same MFMA rate as bf16, 11 significand bits instead of 8):
//   * n <= 2048 is exact in ONE f16 plane (the second, flagged plane is needed only for counts > 2048: n = lo +
//     2048 hi, hi <= 31);
//   * the factor row  a[c][:]  is held as TWO f16 planes of  y = a * 2^s_c :  h = f16(y), m = f16(y - h).
//     y - h is exact in f32 and has at most 13 significant bits, so  h + m = y  exactly for 3 values in 4 and
//     |y - h - m| <= 1 ulp_f32(y) otherwise (when the 13th bit is set and the value is odd): the factor is
//     represented to ~23.5 bits, every partial product (11 x 11 bits) is exact, accumulation is f32.
//     s_c is a per-ROW exponent (max_j y = 2^14 .. 2^15, from the row maximum the sweep that produced the row
//     reports), so f16's narrow exponent range costs nothing: entries down to 2^-14 of the row maximum keep all 22
//     bits, smaller ones an ABSOLUTE error <= 2^-39 of the row maximum.  The product is scaled back by 2^-s_c
//     (exact) when the tile is stored.
//   => 2 MFMAs per product instead of 3, 24 KB instead of 32 KB of operands per 16-k block.
//
// Plane layouts, block-major like the bf16 ones ([row tile of 256][16-k block][row][...]), one (tile, block) = one
// contiguous run = the LDS image the DMA writes:
//   factor : 64 B per row and block = four 16-B slots  c = 2 q + hf  (q: 0 = h, 1 = m; hf = 8-k half), stored at
//            slot  c ^ ((row >> 2) & 3)  -- a ds_read_b128 lane group (16 rows, one slot each) then covers all 16
//            bank groups: conflict-free without padding;
//   counts : 32 B per row and block = two slots, stored at  hf ^ ((row >> 3) & 1)  (same argument).
// The kernel body is gemm3c's (256 x 256 tile, 8 waves in two groups half a block apart, 4 LDS-DMA images, counted
// vmcnt, raw barriers) with NSUB 16-k sub-blocks per barrier pair: NSUB = 2 halves the number of barriers per flop.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_gemm3.hip.h"

namespace cnmf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_g __attribute__((ext_vector_type(4)));
// (A/B build: -DCNMF_G2_LDS_STORE sends the finished tile through LDS and out with 16-byte stores; measured on one box,
//  two alternations: 216.8 / 215.6 restarts/s against 217.5 / 216.8 with the four-byte stores straight from the
//  accumulator registers -- the stores cost 10-13 % of a pass (CNMF_G2_NOSTORE ablation) but not through their issue)
#ifdef CNMF_G2_LDS_STORE
constexpr bool G2_LDS_STORE = true;
#else
constexpr bool G2_LDS_STORE = false;
#endif

constexpr int G2_ROWB = 64;                        // factor bytes per row and 16-k block
constexpr int G2_A = G3_MW * G2_ROWB;              // 16 384 B of factor planes per 16-k block
constexpr int G2_B = G3C_JW * 32;                  //  8 192 B of one count plane per 16-k block
constexpr float G2_COUNT_BASE = 2048.0f;           // lo plane holds n <= 2048, hi plane 2048 * hi (hi <= 31)

constexpr int g2_imgs(int nsub) { return nsub == 1 ? 4 : 3; }
constexpr int g2_img_bytes(int nsub, bool with_hi) { return nsub * (G2_A + (with_hi ? 2 : 1) * G2_B); }
constexpr int g2_lds_bytes(int nsub, bool with_hi) { return g2_imgs(nsub) * g2_img_bytes(nsub, with_hi); }

__device__ __forceinline__ unsigned short f16_bits(float x)
{
    const _Float16 h = (_Float16)x;                // v_cvt_f16_f32, round to nearest even
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float f16_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// exponent shift of a row whose largest entry is mx:  mx * 2^s in [2^14, 2^15)   (0 for an empty row)
__device__ __forceinline__ int g2_row_shift(float mx)
{
    if (!(mx > 0.f)) return 0;
    int e;
    frexpf(mx, &e);                                // mx = f * 2^e, f in [0.5, 1)  ->  mx in [2^(e-1), 2^e)
    const int s = 15 - e;
    return s > 120 ? 120 : (s < -110 ? -110 : s);
}

// y -> (h, m)
__device__ __forceinline__ void split2h(float y, unsigned short& h, unsigned short& m)
{
    h = f16_bits(y);
    m = f16_bits(y - f16_to_f32(h));
}

// ---- planes of the packed factor, through LDS.  A workgroup converts tiles of TROWS rows x TKB 16-k blocks
// (TROWS * TKB = 256: 16 rows x 256 k -- 1 KB contiguous per row read and per block written -- when K % 256 == 0,
// else 64 rows x 64 k) and walks the k tiles bx, bx + nbx, ... of its rows (the row scale is reduced once).
//   rmax_part [rows][parts] : per-row maxima of the values to convert (after kscale), one per sweep workgroup
//   inv_scale [rows]        : 2^-s_c, written by the workgroups of the first k column (bx == 0)
// `tile` = unsigned short [TKB][TROWS][32] (16 KB), `red` = float [256 / TROWS][TROWS] (1 KB, may alias the tile)
// Fused mode (round 3, the W half-step: shift_in != nullptr): the sweep has ALREADY written the planes of its rows, scaled
// by the exponent shift_in[row] chosen one iteration earlier.  This body then only (i) reduces the row maxima the sweep
// reported, (ii) checks that the exponent used was adequate -- largest scaled entry below the f16 overflow and not more
// than 5 bits under the target (the f16 planes keep an absolute accuracy 2^-39 of the scale: 5 bits of slack is what the
// sqrt(sum w^2) bound costs anyway) --, (iii) publishes 2^-s for pass B and the exponent of the NEXT iteration
// (shift_out: unchanged while the scaled maximum stays in [2^13, 2^15.5), else re-centred), and (iv) re-converts its 16
// rows from the float32 factor ONLY when one of them failed the check (a restart's first iterations).
struct SplitFused { const int* shift_in; int* shift_out; };
template <int TROWS, int TKB>
__device__ __forceinline__ void split2h_tiled_body(const float* __restrict__ src, int ld, int K, int TR,
                                                   unsigned short* __restrict__ dst, const double* __restrict__ kscale,
                                                   const float* __restrict__ rmax_part, int parts,
                                                   float* __restrict__ inv_scale, int bx, int nbx, int by,
                                                   unsigned short* tile_, float* red_, SplitFused fu = SplitFused{nullptr, nullptr})
{
    static_assert(TROWS * TKB == 256, "tile = 256 (row, block) pairs");
    constexpr int NQ = 256 / TROWS;                      // threads per row in the maximum reduction
    constexpr int KQ = TKB * 4;                          // float4 per row of a tile
    constexpr int RPP = 256 / KQ;                        // rows per pass of the conversion
    unsigned short (*tile)[TROWS][32] = reinterpret_cast<unsigned short (*)[TROWS][32]>(tile_);
    float (*red)[TROWS] = reinterpret_cast<float (*)[TROWS]>(red_);
    const int t = threadIdx.x;
    const int r0 = by * TROWS;
    const int kq = t % KQ, rr = t / KQ;                  // float4 index along k, row within a pass
    const int ntiles = K / (16 * TKB);
    // the data of the first k tile and the row maxima are requested together (one memory latency, not two)
    const bool fused = fu.shift_in != nullptr;
    float4 v[4];
    if (!fused) {
        const int k0 = bx * 16 * TKB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = *reinterpret_cast<const float4*>(src + (size_t)(r0 + rr + RPP * i) * ld + k0 + kq * 4);
    }
    {
        const int row = t % TROWS, qt = t / TROWS;
        float mx = 0.f;
        // (one partial per 256-row tile since round 4: 196 of them at 50 000 cells -- eight requests in flight per pass)
        for (int p0 = qt; p0 < parts; p0 += 8 * NQ) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (p0 + j * NQ < parts) ? rmax_part[(size_t)(r0 + row) * parts + p0 + j * NQ] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[j]);
        }
        red[qt][row] = mx;
    }
    __syncthreads();
    int sh[4];
    bool redo = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rr + RPP * i;
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) mx = fmaxf(mx, red[q][row]);
        sh[i] = g2_row_shift(mx);
        if (fused) {
            const int su = fu.shift_in[r0 + row];
            const float ymax = ldexpf(mx, su);
            const bool ok = !(mx > 0.f) || (ymax < 60000.0f && ymax >= 1024.0f);
            const bool keep = !(mx > 0.f) || (ymax < 46340.0f && ymax >= 8192.0f);
            if (bx == 0 && kq == 0) fu.shift_out[r0 + row] = (ok && keep) ? su : sh[i];
            if (ok) sh[i] = su; else redo = true;
        }
        if (bx == 0 && kq == 0) inv_scale[r0 + row] = ldexpf(1.0f, -sh[i]);
    }
    if (fused) {
        // every thread looked at 4 of the TROWS rows: does ANY row of the workgroup need the conversion?
        if (!__syncthreads_or(redo ? 1 : 0)) return;
        const int k0 = bx * 16 * TKB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = *reinterpret_cast<const float4*>(src + (size_t)(r0 + rr + RPP * i) * ld + k0 + kq * 4);
    }
    __syncthreads();                                     // `red` may alias the tile
    const int Kb = K / 16;
    const int tr = r0 / TR, rin = r0 % TR;
    for (int kt = bx; kt < ntiles; kt += nbx) {
        const int k0 = kt * 16 * TKB;
        // request the NEXT tile before converting this one
        float4 vn[4];
        const bool has_next = kt + nbx < ntiles;
        if (has_next) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                vn[i] = *reinterpret_cast<const float4*>(src + (size_t)(r0 + rr + RPP * i) * ld + (k0 + nbx * 16 * TKB) + kq * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rr + RPP * i;
            float x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            if (kscale) {                                 // count-structured data: the per-gene scale rides on the factor
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (float)((double)x[e] * kscale[k0 + kq * 4 + e]);
            }
            unsigned short p[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split2h(ldexpf(x[e], sh[i]), p[0][e], p[1][e]);
            // 16-k block kq >> 2, 8-k half (kq >> 1) & 1, element offset (kq & 1) * 4 inside the slot;
            // (r0, TR multiples of 16: bits 2..3 of the in-tile row are bits 2..3 of the local row)
            const int hf = (kq >> 1) & 1, swz = (row >> 2) & 3;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint2 w;
                w.x = p[q][0] | ((unsigned)p[q][1] << 16); w.y = p[q][2] | ((unsigned)p[q][3] << 16);
                *reinterpret_cast<uint2*>(&tile[kq >> 2][row][((2 * q + hf) ^ swz) * 8 + (kq & 1) * 4]) = w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int c = pass * 256 + t;                 // 16-byte chunk of the tile: [block][row][4 chunks]
            const int b = c / (TROWS * 4), within = c % (TROWS * 4);
            unsigned short* g = dst + (((size_t)tr * Kb + (k0 / 16 + b)) * TR + rin) * 32;
            reinterpret_cast<u32x4*>(g)[within] = reinterpret_cast<const u32x4*>(&tile[b][0][0])[within];
        }
        __syncthreads();
        if (has_next) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = vn[i];
        }
    }
}

// grid = (column groups, rows / TROWS)
template <int TROWS, int TKB>
__global__ __launch_bounds__(256) void split2h_tiled_kernel(const float* __restrict__ src, int ld, int K, int TR,
                                                            unsigned short* __restrict__ dst,
                                                            const double* __restrict__ kscale,
                                                            const float* __restrict__ rmax_part, int parts,
                                                            float* __restrict__ inv_scale)
{
    // 16 KB in all: the maxima scratch lives in the tile (it is consumed before the first tile is written), so that a
    // split workgroup fits beside a GEMM workgroup (144 of the 160 KB) when two batches share the GPU
    __shared__ __attribute__((aligned(16))) unsigned short tile[256 * 32];
    float* red = reinterpret_cast<float*>(tile);
    split2h_tiled_body<TROWS, TKB>(src, ld, K, TR, dst, kscale, rmax_part, parts, inv_scale, blockIdx.x, gridDim.x,
                                   blockIdx.y, tile, red);
}

// per-row maxima of a packed factor in the partials layout the sweep writes ([rows][parts], part p = rows' columns
// [p * span, (p+1) * span)): used for rows the sweep has not produced (freshly installed or moved slots).
// grid = (parts, rows / 4): one wave per row and part.
__global__ __launch_bounds__(256) void rowmax_part_kernel(const float* __restrict__ V, int ld, int L, int span,
                                                          const double* __restrict__ kscale, int parts,
                                                          float* __restrict__ rmax_part)
{
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, p = blockIdx.x;
    const int j0 = p * span, j1 = min(L, j0 + span);
    float mx = 0.f;
    for (int j = j0 + lane; j < j1; j += 64) {
        float x = V[(size_t)row * ld + j];
        if (kscale) x = (float)((double)x * kscale[j]);
        mx = fmaxf(mx, x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) rmax_part[(size_t)row * parts + p] = mx;
}

// ---- count planes in f16: [row tile][16-k block][row][16 f16], slots swapped for rows with bit 3 set
__device__ __forceinline__ void count_lo_hi_2048(float n, float& lo, float& hi)
{
    if (n <= G2_COUNT_BASE) { lo = n; hi = 0.f; }
    else { hi = floorf(n * (1.0f / G2_COUNT_BASE)); lo = n - G2_COUNT_BASE * hi; hi *= G2_COUNT_BASE; }
}

__device__ __forceinline__ void store_plane_row_swz(unsigned short* d, const unsigned short* p, int row_in_tile)
{
    const int swz = (row_in_tile >> 3) & 1;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        u32x4 w;
        w.x = p[8 * hf + 0] | ((unsigned)p[8 * hf + 1] << 16); w.y = p[8 * hf + 2] | ((unsigned)p[8 * hf + 3] << 16);
        w.z = p[8 * hf + 4] | ((unsigned)p[8 * hf + 5] << 16); w.w = p[8 * hf + 6] | ((unsigned)p[8 * hf + 7] << 16);
        *reinterpret_cast<u32x4*>(d + (hf ^ swz) * 8) = w;
    }
}

// rows = cells, k = genes (pass A's operand).  One thread per (row, block).
__global__ __launch_bounds__(256) void count_planes_f16_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                               int rows_pad, int K, int TR,
                                                               const float* __restrict__ unit,
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
        count_lo_hi_2048(n, lo, hi);
        p[i] = f16_bits(lo); ph[i] = f16_bits(hi);           // both exact
        any_hi |= hi != 0.f;
    }
    const size_t blk = (size_t)(row / TR) * Kb + kb;
    store_plane_row_swz(dst + (blk * TR + (row % TR)) * 16, p, row % TR);
    if (dst_hi) {
        store_plane_row_swz(dst_hi + (blk * TR + (row % TR)) * 16, ph, row % TR);
        if (any_hi) atomicOr(&hiflag[(size_t)(row / TR) * ((Kb + 31) / 32) + (kb >> 5)], 1u << (kb & 31));
    }
}

// rows = genes, k = cells (pass B's operand).  One thread per (gene row j, block); lanes run along j.
__global__ __launch_bounds__(256) void count_planes_f16_transpose_kernel(const float* __restrict__ X, int ld, int N, int G,
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
        count_lo_hi_2048(n, lo, hi);
        p[i] = f16_bits(lo); ph[i] = f16_bits(hi);
        any_hi |= hi != 0.f;
    }
    const size_t blk = (size_t)(j / TR) * Kb + kb;
    store_plane_row_swz(dst + (blk * TR + (j % TR)) * 16, p, j % TR);
    if (dst_hi) {
        store_plane_row_swz(dst_hi + (blk * TR + (j % TR)) * 16, ph, j % TR);
        if (any_hi) atomicOr(&hiflag[(size_t)(j / TR) * ((Kb + 31) / 32) + (kb >> 5)], 1u << (kb & 31));
    }
}

// ---- general (NOT count-structured) X on the f16 pipe, gemm_mode 5: every row r of the operand (a cell for pass A, a
// gene for pass B) is held as TWO f16 planes of  y = x * 2^s_r  (h = f16(y), m = f16(y - h): within 1 ulp_f32 of x,
// exactly like the factor planes), s_r from the row maximum; the GEMM multiplies all four plane pairs (4 MFMAs per
// product, the `HI` instantiation with every block flagged) and its epilogue undoes 2^s_r per OUTPUT column.
// Replaces the 3 x 3 bf16 planes of rounds 1-2 (6 MFMAs per product, terms below 2^-18 dropped).
// row maxima: rows of X (transpose = 0: one wave per row) or columns of X (transpose = 1: one thread per column)
__global__ __launch_bounds__(256) void x2h_rowshift_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                           int transpose, int rows_pad, int* __restrict__ shift,
                                                           float* __restrict__ inv_scale)
{
    if (!transpose) {
        const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= rows_pad) return;
        float mx = 0.f;
        if (row < N) for (int g = lane; g < G; g += 64) mx = fmaxf(mx, X[(size_t)row * ld + g]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) { const int s = g2_row_shift(mx); shift[row] = s; inv_scale[row] = ldexpf(1.0f, -s); }
    } else {
        const int g = blockIdx.x * 256 + threadIdx.x;
        if (g >= rows_pad) return;
        float mx = 0.f;
        if (g < G) for (int r = 0; r < N; ++r) mx = fmaxf(mx, X[(size_t)r * ld + g]);
        const int s = g2_row_shift(mx);
        shift[g] = s; inv_scale[g] = ldexpf(1.0f, -s);
    }
}

// rows = cells, k = genes (pass A's operand).  One thread per (row, 16-k block).
__global__ __launch_bounds__(256) void x2h_planes_kernel(const float* __restrict__ X, int ld, int N, int G, int rows_pad,
                                                         int K, int TR, const int* __restrict__ shift,
                                                         unsigned short* __restrict__ dst_h, unsigned short* __restrict__ dst_m)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Kb = K / 16;
    if (t >= (long long)rows_pad * Kb) return;
    const int row = (int)(t / Kb), kb = (int)(t % Kb);
    const int s = shift[row];
    unsigned short ph[16], pm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int g = kb * 16 + i;
        const float x = (row < N && g < G) ? X[(size_t)row * ld + g] : 0.f;
        split2h(ldexpf(x, s), ph[i], pm[i]);
    }
    const size_t blk = (size_t)(row / TR) * Kb + kb;
    store_plane_row_swz(dst_h + (blk * TR + (row % TR)) * 16, ph, row % TR);
    store_plane_row_swz(dst_m + (blk * TR + (row % TR)) * 16, pm, row % TR);
}

// rows = genes, k = cells (pass B's operand).  One thread per (gene row j, block); lanes run along j.
__global__ __launch_bounds__(256) void x2h_planes_transpose_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                                   int rows_pad, int K, int TR,
                                                                   const int* __restrict__ shift,
                                                                   unsigned short* __restrict__ dst_h,
                                                                   unsigned short* __restrict__ dst_m)
{
    const int j = blockIdx.y * 256 + threadIdx.x;       // (grid = (16-k blocks, row groups): the long dimension on x -- no 65 535 limit)
    const int kb = blockIdx.x;
    if (j >= rows_pad) return;
    const int Kb = K / 16;
    const int s = shift[j];
    unsigned short ph[16], pm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = kb * 16 + i;
        const float x = (j < G && c < N) ? X[(size_t)c * ld + j] : 0.f;
        split2h(ldexpf(x, s), ph[i], pm[i]);
    }
    const size_t blk = (size_t)(j / TR) * Kb + kb;
    store_plane_row_swz(dst_h + (blk * TR + (j % TR)) * 16, ph, j % TR);
    store_plane_row_swz(dst_m + (blk * TR + (j % TR)) * 16, pm, j % TR);
}

// does any column need the second plane?  (max n > base)
__global__ __launch_bounds__(256) void count_max_base_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                             const float* __restrict__ unit, float base,
                                                             unsigned* __restrict__ any_big)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const float u = unit[g];
    if (u <= 0.f) return;
    const int r0 = blockIdx.y * 256, r1 = min(N, r0 + 256);
    bool big = false;
    for (int r = r0; r < r1; ++r) big |= rintf(X[(size_t)r * ld + g] / u) > base;
    if (big) atomicOr(any_big, 1u);
}

// LDS-DMA issued from inline asm (cdna_hip_programming.md section 5.7): hipcc does not see an LDS write, so it neither
// drains the pending fragment reads in front of it (the builtin makes it wait lgkmcnt(0): a possible alias) nor counts it
// in vmcnt -- the kernel counts by hand anyway.  M0 is saved and restored inside the statement.
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_asm_nt(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ------------------------------------------------------------------------------------------
// One K segment [kb0, kb0 + nkb) (in 16-k blocks; kb0 and nkb multiples of NSUB) of one 256 x 256 tile.
//   A2 : f16 planes of the component-major factor (rows m0..), B1 : f16 count plane (rows j0..),
//   Bhi / hiflag : second count plane and its per-(tile, block) flags (nullptr: none),
//   rscale : 2^-s_c per component row, applied to the stored product.
// Timeline exactly as gemm3c_segment, in units of "steps" of NSUB blocks.
// VAR (A/B knobs of the instruction stream; NSUB = 2 without a second count plane only):
//   bit 0: the six LDS-DMA pieces of a step are spread through the MFMA stream (one after every 4th MFMA of the step's
//          first 24) instead of issued in one burst behind the X barrier -- a burst of DMA issues in front of 20
//          ds_read_b128 is the expensive place for them (MI355X_MICROARCH.md, "LDS-DMA piece issue cost");
//   VAR 4: VAR 1 with the fragment reads in two batches (see G2_READ_A / G2_READ_B);
//   VAR 5: VAR 4 with the steady-state DMA pieces issued from inline asm (glds16_asm);
//   VAR 2 / 3 (timing ablations, results meaningless): the spread stream WITHOUT its MFMAs (fill + fragment reads
//          only) / WITHOUT its steady-state DMA (MFMAs + fragment reads on stale images).  s_setprio around the MFMA
//          halves was tried and is neutral (profiles/r2_probe_gemm2h_variants.txt).
// NTB: the count plane is loaded non-temporally (every tile is read by ONE workgroup per pass: 256 packed columns, or pass B
// with its XCD-aware order); with several component groups in pass A the workgroups p, p + P / MG stream the SAME tile at
// the same time and must find it in their L2: default cache policy (PMC, 1024 columns: 607 -> see profiles/r3_pmc_*).
// PART: only the 32-row component tiles named in `live` (bit t = rows 32 t .. 32 t + 31 of this 256-row group) are
// multiplied and stored -- the tail of a call, when restarts have finished and nothing is left to refill their columns:
// a wave skips the MFMAs of its dead tiles (wave-uniform scalar branches), so a pass costs what its live columns cost,
// down to the floor of streaming the count plane.  The operand traffic is unchanged.
// GEN (round 4, general matrices = X itself as two f16 planes, HI instantiation): the product of the two SMALL planes,
// x_m . f_m, is not formed -- |x_m| <= 2^-11 |x_h| and |f_m| <= 2^-11 |f_h|, so the term is <= 2^-22 of the product, random
// in sign, against the 2^-24 rounding of every float32 accumulation step (and 16 x below the 2^-18 the 3 x 3 bf16 planes of
// rounds 1-2 dropped): 3 MFMAs per product instead of 4.  The count path (HI there = a second EXACT integer plane) keeps all.
// EPI (round 4): what happens to a WHOLE tile's accumulators instead of the plain store -- the W half-step of the tile's
// restarts run by the pass-A workgroup itself (kernels_fusedw.hip.h); G2NoEpi: the store below.
struct G2NoEpi { static constexpr bool enabled = false; };
template <int NSUB, bool HI, int VAR = 0, bool NTB = true, bool PART = false, bool GEN = false, class EPI = G2NoEpi>
__device__ __forceinline__ void gemm2h_segment(const unsigned char* __restrict__ A2, const unsigned char* __restrict__ B1,
                                               const unsigned char* __restrict__ Bhi,
                                               const unsigned int* __restrict__ hiflag,
                                               const float* __restrict__ rscale,
                                               int Kb, float* __restrict__ C, int ldc, int m0, int j0, int kb0,
                                               int nkb, unsigned char* smem, const float* __restrict__ cscale = nullptr,
                                               unsigned live = 0xffu, const EPI* epi = nullptr, bool whole_tile = false)
{
    constexpr int IMGS = g2_imgs(NSUB);
    constexpr int IMG = g2_img_bytes(NSUB, HI);
    constexpr int OFF_B = NSUB * G2_A;                   // image: [A sub-blocks][B sub-blocks][hi sub-blocks]
    constexpr int OFF_H = OFF_B + NSUB * G2_B;
    constexpr int NA = NSUB * 2, NB = NSUB;              // DMA instructions per step and wave (8 KB each over the workgroup)
    constexpr int NDMA = NA + NB;                        // minimum per step (flagged steps issue NB more)
    const int tid = threadIdx.x;                         // 0..511
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wn = wave & 3;
    const int li = lane & 31, h = lane >> 5;
    const int nst = nkb / NSUB;                          // steps in this segment
    // the wave's four component tiles (rows grp * 128 + m * 32 ...): which of them hold live columns?
    const unsigned lv = PART ? (unsigned)__builtin_amdgcn_readfirstlane((int)((live >> (grp * 4)) & 0xfu)) : 0xfu;
#define G2_LIVE(m_) (!PART || ((lv >> (m_)) & 1u))

    const size_t jblk = ((size_t)(j0 / G3C_JW) * Kb + kb0);
    const unsigned char* abase = A2 + ((size_t)(m0 / G3_MW) * Kb + kb0) * G2_A + tid * 16;
    const unsigned char* bbase = B1 + jblk * G2_B + tid * 16;
    const unsigned char* hbase = HI ? Bhi + jblk * G2_B + tid * 16 : nullptr;
    // flag bits of blocks kb0 .. kb0 + nkb - 1 (at most 5 words for up to 129 blocks; longer segments: all set)
    unsigned int fw[5] = {0u, 0u, 0u, 0u, 0u};
    if (HI) {
        const int KW = (Kb + 31) >> 5, w0 = kb0 >> 5;
        const unsigned int* fr = hiflag + (size_t)(j0 / G3C_JW) * KW;
#pragma unroll
        for (int w = 0; w < 5; ++w)
            fw[w] = (nkb > 129) ? 0xFFFFFFFFu : ((w0 + w < KW) ? __builtin_amdgcn_readfirstlane(fr[w0 + w]) : 0u);
    }
    const int fbit0 = kb0 & 31;

    f32x16_3 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // fragment addresses inside an image: factor row (grp*128 + m*32 + li), slot (2q + h) ^ ((li >> 2) & 3);
    // count row (wn*64 + n*32 + li), slot h ^ ((li >> 3) & 1)
    const int a_row = (grp * 128 + li) * G2_ROWB;
    const int a_s0 = ((0 + h) ^ ((li >> 2) & 3)) * 16, a_s1 = ((2 + h) ^ ((li >> 2) & 3)) * 16;
    const int b_off = OFF_B + (wn * 64 + li) * 32 + (h ^ ((li >> 3) & 1)) * 16;

#define G2_BLKFLAG(b_) (HI && ((fw[(fbit0 + (b_)) >> 5 > 4 ? 4 : (fbit0 + (b_)) >> 5] >> ((fbit0 + (b_)) & 31)) & 1u))
#define G2_ISSUE(s_)                                                                               \
    {                                                                                              \
        unsigned char* d_ = smem + ((s_) % IMGS) * IMG + wave * 1024;                              \
        const unsigned char* a_ = abase + (size_t)(s_) * (NSUB * G2_A);                            \
        if constexpr (VAR == 5) {                       /* no LDS-DMA builtin anywhere in this variant */ \
            const unsigned l_ = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)G3_AS3(d_)); \
            _Pragma("unroll") for (int i = 0; i < NA; ++i) glds16_asm(a_ + i * 8192, l_ + i * 8192); \
            _Pragma("unroll") for (int i = 0; i < NB; ++i)                                         \
                glds16_asm_nt(bbase + (size_t)(s_) * (NSUB * G2_B) + i * 8192, l_ + OFF_B + i * 8192); \
        } else {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                             \
            __builtin_amdgcn_global_load_lds(G3_AS1(a_ + i * 8192), G3_AS3(d_ + i * 8192), 16, 0, 0); \
        /* the count planes are read once per pass: non-temporal */                                \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                             \
            __builtin_amdgcn_global_load_lds(G3_AS1(bbase + (size_t)(s_) * (NSUB * G2_B) + i * 8192), \
                                             G3_AS3(d_ + OFF_B + i * 8192), 16, 0, NTB ? 2 : 0);   \
        if (HI) {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < NB; ++i)                                         \
                if (G2_BLKFLAG((s_) * NSUB + i))                                                   \
                    __builtin_amdgcn_global_load_lds(G3_AS1(hbase + (size_t)(s_) * (NSUB * G2_B) + i * 8192), \
                                                     G3_AS3(d_ + OFF_H + i * 8192), 16, 0, NTB ? 2 : 0); \
        }                                                                                          \
        }                                                                                          \
    }
#define G2_FRAG(ptr_) __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ptr_))
#define G2_READ(s_)                                                                                \
    {                                                                                              \
        const unsigned char* bb = smem + ((s_) % IMGS) * IMG;                                      \
        _Pragma("unroll") for (int u = 0; u < NSUB; ++u) {                                         \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) bq[u][n] = G2_FRAG(bb + b_off + u * G2_B + n * 32 * 32); \
            if (HI) { _Pragma("unroll") for (int n = 0; n < 2; ++n)                                \
                          bh[u][n] = G2_FRAG(bb + b_off + NSUB * G2_B + u * G2_B + n * 32 * 32); } \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                        \
                aq[u][m][0] = G2_FRAG(bb + u * G2_A + a_row + m * 32 * G2_ROWB + a_s0);            \
                aq[u][m][1] = G2_FRAG(bb + u * G2_A + a_row + m * 32 * G2_ROWB + a_s1);            \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
// two component tiles at a time, small plane first: consecutive MFMAs cycle through four accumulators
#define G2_MFMA(b_, u_, m_, q_)                                                                    \
    if (G2_LIVE(m_)) {                                                                             \
    acc[m_][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[u_][m_][q_], b_[u_][0], acc[m_][0], 0, 0, 0); \
    acc[m_][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[u_][m_][q_], b_[u_][1], acc[m_][1], 0, 0, 0); }
#define G2_MFMA4(b_, u_, ma_, mb_)                                                                 \
    G2_MFMA(b_, u_, ma_, 1) G2_MFMA(b_, u_, mb_, 1) G2_MFMA(b_, u_, ma_, 0) G2_MFMA(b_, u_, mb_, 0)
#define G2_MFMA2H(b_, u_, ma_, mb_) G2_MFMA(b_, u_, ma_, 0) G2_MFMA(b_, u_, mb_, 0)
#define G2_HALF(ma_, mb_)                                                                          \
    _Pragma("unroll") for (int u = 0; u < NSUB; ++u) {                                             \
        G2_MFMA4(bq, u, ma_, mb_)                                                                  \
        if (HI) { if (hi_blk[u]) { if constexpr (GEN) { G2_MFMA2H(bh, u, ma_, mb_) } else { G2_MFMA4(bh, u, ma_, mb_) } } } \
    }
// "the step after this one has landed; the ones behind it may stay in flight"
#define G2_WAIT_AHEAD(cnt_)                                                                        \
    {                                                                                              \
        if constexpr ((cnt_) * NDMA == 0) G3_WAIT_VM(0);                                           \
        else if constexpr ((cnt_) * NDMA == 3) G3_WAIT_VM(3);                                      \
        else if constexpr ((cnt_) * NDMA == 6) G3_WAIT_VM(6);                                      \
        else if constexpr ((cnt_) * NDMA == 12) G3_WAIT_VM(12);                                    \
        else G3_WAIT_VM(0);                                                                        \
    }

    f16x8 bq[NSUB][2], bh[NSUB][2], aq[NSUB][4][2];
    constexpr int AHEAD = IMGS - 1;                      // steps requested ahead of the one being multiplied
    constexpr bool SPREAD = VAR >= 1 && NSUB == 2 && !HI;
    // round 6, general matrices (GEN: both X planes in every block, flags all set): the count kernels' spread stream for the
    // one-block-per-step instantiation -- the four LDS-DMA pieces of step s + 3 ride between the MFMA groups instead of a
    // burst behind the X barrier, the fragment reads come in two batches (8 + 4).  Same MFMA order per accumulator as the
    // burst loop below: results are bit-identical (tests/test_gpu_nmf.py).
    constexpr bool SPREADG = VAR >= 1 && NSUB == 1 && ((HI && GEN) || !HI);   // (!HI: the count plane alone, A/B via CNMF_G2_NSUB=1)
    constexpr bool PRIO = false;
    constexpr bool NOMFMA = VAR == 2, NODMA = VAR == 3 || VAR == 6 || VAR == 7, SPLITRD = VAR >= 4, ASMDMA = VAR == 5;
    constexpr bool NOREAD = VAR == 6 || VAR == 7;         // timing ablations: fragments read once / ... and no barriers
    constexpr bool NOBAR = VAR == 7;
    G3_WAIT_VM(0);                                          // stores of a previous segment
    G2_ISSUE(0)
    if (nst > 1) G2_ISSUE(1)
    if (AHEAD > 2 && nst > 2) G2_ISSUE(2)
    // step 0 landed (everything issued after it may stay in flight)
    if (AHEAD > 2 && nst > 2) G2_WAIT_AHEAD(2) else if (nst > 1) G2_WAIT_AHEAD(1) else G2_WAIT_AHEAD(0)
    if (grp == 1) G3_RAW_BARRIER()
    if constexpr (SPREADG) {
        static_assert(!SPREADG || (IMGS == 4 && NA == 2 && NB == 1), "one-block spread stream: four images, pieces A0 A1 B (Bhi)");
#define G2G_PIECE(s_, i_)                                                                          \
        {                                                                                          \
            unsigned char* d_ = smem + ((s_) % IMGS) * IMG + wave * 1024;                          \
            if ((i_) < NA)                                                                         \
                __builtin_amdgcn_global_load_lds(G3_AS1(abase + (size_t)(s_) * G2_A + (i_) * 8192), \
                                                 G3_AS3(d_ + (i_) * 8192), 16, 0, 0);              \
            else if ((i_) == NA)                                                                   \
                __builtin_amdgcn_global_load_lds(G3_AS1(bbase + (size_t)(s_) * G2_B), G3_AS3(d_ + OFF_B), 16, 0, NTB ? 2 : 0); \
            else if constexpr (HI)                                                                 \
                __builtin_amdgcn_global_load_lds(G3_AS1(hbase + (size_t)(s_) * G2_B), G3_AS3(d_ + OFF_H), 16, 0, NTB ? 2 : 0); \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
#define G2G_READ_A(s_)                                                                             \
        {                                                                                          \
            const unsigned char* bb = smem + ((s_) % IMGS) * IMG;                                  \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) bq[0][n] = G2_FRAG(bb + b_off + n * 32 * 32); \
            _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                        \
                aq[0][m][1] = G2_FRAG(bb + a_row + m * 32 * G2_ROWB + a_s1);                       \
                aq[0][m][0] = G2_FRAG(bb + a_row + m * 32 * G2_ROWB + a_s0);                       \
            }                                                                                      \
            if constexpr (HI) { _Pragma("unroll") for (int n = 0; n < 2; ++n) bh[0][n] = G2_FRAG(bb + b_off + G2_B + n * 32 * 32); } \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
#define G2G_READ_B(s_)                                                                             \
        {                                                                                          \
            const unsigned char* bb = smem + ((s_) % IMGS) * IMG;                                  \
            _Pragma("unroll") for (int m = 2; m < 4; ++m) {                                        \
                aq[0][m][1] = G2_FRAG(bb + a_row + m * 32 * G2_ROWB + a_s1);                       \
                aq[0][m][0] = G2_FRAG(bb + a_row + m * 32 * G2_ROWB + a_s0);                       \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
#define G2G_STEP(MORE_)                                                                            \
        {                                                                                          \
            G3_RAW_BARRIER()                                        /* X_s: image (s + 3) % 4 = (s - 1) % 4 is free */ \
            G2G_READ_A(s)                                                                          \
            G2_MFMA(bq, 0, 0, 1) G2_MFMA(bq, 0, 1, 1)                                              \
            if (MORE_) G2G_PIECE(s + AHEAD, 0)                                                     \
            G2_MFMA(bq, 0, 0, 0) G2_MFMA(bq, 0, 1, 0)                                              \
            G2G_READ_B(s)                                                                          \
            if (MORE_) G2G_PIECE(s + AHEAD, 1)                                                     \
            if constexpr (HI) { G2_MFMA(bh, 0, 0, 0) G2_MFMA(bh, 0, 1, 0) }                        \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
            /* step s+1 must have landed before Y_s; step s+2 (4 / 3 pieces) and the two pieces of step s+3 just issued may stay in flight */ \
            if (s + 1 < nst) { if (MORE_) { if constexpr (HI) G3_WAIT_VM(6); else G3_WAIT_VM(5); } else G3_WAIT_VM(0); } \
            G3_RAW_BARRIER()                                        /* Y_s */                      \
            G2_MFMA(bq, 0, 2, 1) G2_MFMA(bq, 0, 3, 1)                                              \
            if (MORE_) G2G_PIECE(s + AHEAD, 2)                                                     \
            G2_MFMA(bq, 0, 2, 0) G2_MFMA(bq, 0, 3, 0)                                              \
            if constexpr (HI) { if (MORE_) G2G_PIECE(s + AHEAD, 3)                                 \
                                G2_MFMA(bh, 0, 2, 0) G2_MFMA(bh, 0, 3, 0) }                        \
        }
        {
            int s = 0;
            const int n_main = nst - AHEAD;                         // steps that still have a step to request
            for (; s < n_main; ++s) G2G_STEP(true)
            for (; s < nst; ++s) G2G_STEP(false)
        }
#undef G2G_STEP
#undef G2G_READ_B
#undef G2G_READ_A
#undef G2G_PIECE
    } else if constexpr (!SPREAD) {
        for (int s = 0; s < nst; ++s) {
            bool hi_blk[NSUB];
#pragma unroll
            for (int u = 0; u < NSUB; ++u) hi_blk[u] = G2_BLKFLAG(s * NSUB + u);
            G3_RAW_BARRIER()                                        // X_s
            if (s + AHEAD < nst) G2_ISSUE(s + AHEAD)
            G2_READ(s)
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            G2_HALF(0, 1)
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this image is free once both groups pass here
            // step s+1 must have landed before Y_s; later ones may stay in flight
            if (s + 1 < nst) {
                if (AHEAD > 2 && s + 3 < nst) G2_WAIT_AHEAD(2) else if (s + 2 < nst) G2_WAIT_AHEAD(1) else G2_WAIT_AHEAD(0)
            }
            G3_RAW_BARRIER()                                        // Y_s
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            G2_HALF(2, 3)
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }
    } else {
        // NSUB = 2, six pieces per step (A 0..3, B 4..5): pieces 0-2 ride in the first half, 3-5 in the second
#define G2_PIECE(s_, i_)                                                                           \
        {                                                                                          \
            unsigned char* d_ = smem + ((s_) % IMGS) * IMG + wave * 1024;                          \
            if constexpr (ASMDMA) {                                                                \
                const unsigned l_ = __builtin_amdgcn_readfirstlane(                                \
                    (unsigned)(unsigned long long)G3_AS3(d_ + ((i_) < NA ? (i_) * 8192 : OFF_B + ((i_) - NA) * 8192))); \
                if ((i_) < NA) glds16_asm(abase + (size_t)(s_) * (NSUB * G2_A) + (i_) * 8192, l_);  \
                else glds16_asm_nt(bbase + (size_t)(s_) * (NSUB * G2_B) + ((i_) - NA) * 8192, l_);  \
            } else if ((i_) < NA)                                                                  \
                __builtin_amdgcn_global_load_lds(G3_AS1(abase + (size_t)(s_) * (NSUB * G2_A) + (i_) * 8192), \
                                                 G3_AS3(d_ + (i_) * 8192), 16, 0, 0);              \
            else                                                                                   \
                __builtin_amdgcn_global_load_lds(G3_AS1(bbase + (size_t)(s_) * (NSUB * G2_B) + ((i_) - NA) * 8192), \
                                                 G3_AS3(d_ + OFF_B + ((i_) - NA) * 8192), 16, 0, NTB ? 2 : 0); \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
#define G2_MF(u_, m_, q_)                                                                          \
        if constexpr (NOMFMA) { asm volatile("" :: "v"(aq[u_][m_][q_]), "v"(bq[u_][0]), "v"(bq[u_][1])); }        \
        else { G2_MFMA(bq, u_, m_, q_) }
// fragment reads in two batches (VAR 4): the 12 the first half needs, then -- behind the first 8 MFMAs -- the 8 of the
// second half: at most 12 LDS reads are outstanding, so the waits in front of the MFMAs can be counted (lgkmcnt is a
// 4-bit counter: behind 20 reads the compiler can only wait for all of them)
#define G2_READ_A(s_)                                                                              \
    {                                                                                              \
        const unsigned char* bb = smem + ((s_) % IMGS) * IMG;                                      \
        _Pragma("unroll") for (int u = 0; u < NSUB; ++u) {                                         \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) bq[u][n] = G2_FRAG(bb + b_off + u * G2_B + n * 32 * 32); \
            _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                        \
                aq[u][m][1] = G2_FRAG(bb + u * G2_A + a_row + m * 32 * G2_ROWB + a_s1);            \
                aq[u][m][0] = G2_FRAG(bb + u * G2_A + a_row + m * 32 * G2_ROWB + a_s0);            \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#define G2_READ_B(s_)                                                                              \
    {                                                                                              \
        const unsigned char* bb = smem + ((s_) % IMGS) * IMG;                                      \
        _Pragma("unroll") for (int u = 0; u < NSUB; ++u)                                           \
            _Pragma("unroll") for (int m = 2; m < 4; ++m) {                                        \
                aq[u][m][1] = G2_FRAG(bb + u * G2_A + a_row + m * 32 * G2_ROWB + a_s1);            \
                aq[u][m][0] = G2_FRAG(bb + u * G2_A + a_row + m * 32 * G2_ROWB + a_s0);            \
            }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#define G2_STEP(MORE_)                                                                             \
        {                                                                                          \
            if constexpr (!NOBAR) G3_RAW_BARRIER()                  /* X_s */                      \
            if constexpr (NOREAD) { if (s == 0) { G2_READ(s) } }                                   \
            else if constexpr (SPLITRD) { G2_READ_A(s) } else { G2_READ(s) }                       \
            G2_MF(0, 0, 1) G2_MF(0, 1, 1)                                                          \
            if (MORE_) G2_PIECE(s + AHEAD, 0)                                                      \
            G2_MF(0, 0, 0) G2_MF(0, 1, 0)                                                          \
            if constexpr (SPLITRD && !NOREAD) { G2_READ_B(s) }                                     \
            if (MORE_) G2_PIECE(s + AHEAD, 1)                                                      \
            G2_MF(1, 0, 1) G2_MF(1, 1, 1)                                                          \
            if (MORE_) G2_PIECE(s + AHEAD, 2)                                                      \
            G2_MF(1, 0, 0) G2_MF(1, 1, 0)                                                          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
            /* step s+1 must have landed before Y_s; the three pieces of step s+2 just issued may stay in flight */ \
            if (s + 1 < nst) { if (MORE_) G3_WAIT_VM(3); else G3_WAIT_VM(0); }                     \
            if constexpr (!NOBAR) G3_RAW_BARRIER()                  /* Y_s */                      \
            G2_MF(0, 2, 1) G2_MF(0, 3, 1)                                                          \
            if (MORE_) G2_PIECE(s + AHEAD, 3)                                                      \
            G2_MF(0, 2, 0) G2_MF(0, 3, 0)                                                          \
            if (MORE_) G2_PIECE(s + AHEAD, 4)                                                      \
            G2_MF(1, 2, 1) G2_MF(1, 3, 1)                                                          \
            if (MORE_) G2_PIECE(s + AHEAD, 5)                                                      \
            G2_MF(1, 2, 0) G2_MF(1, 3, 0)                                                          \
        }
        {
            int s = 0;
            const int n_main = NODMA ? 0 : nst - AHEAD;             // steps that still have a step to request
            for (; s < n_main; ++s) G2_STEP(true)
            for (; s < nst; ++s) G2_STEP(false)
        }
#undef G2_STEP
#undef G2_READ_B
#undef G2_READ_A
#undef G2_MF
#undef G2_PIECE
    }
    if (grp == 0) G3_RAW_BARRIER()
#undef G2_WAIT_AHEAD
#undef G2_HALF
#undef G2_MFMA2H
#undef G2_MFMA4
#undef G2_MFMA
#undef G2_READ
#undef G2_FRAG
#undef G2_ISSUE
#undef G2_BLKFLAG

    if constexpr (EPI::enabled) {
        // a tile computed in one piece: the epilogue takes the accumulators (and stores what it does not consume itself)
        if (whole_tile && epi->on) { (*epi)(acc, rscale, C, ldc, m0, j0, smem); return; }
    }
    if (C == nullptr) return;                                // (timing ablation CNMF_G2_NOSTORE: the pass without its stores)
    const int j = j0 + wn * 64 + li;
    // general (not count-structured) X as two f16 planes of x * 2^s_j (x2h_planes_kernel): the per-column exponent is
    // undone here (a power of two: exact); 1 otherwise
    const float cs[2] = {(HI && cscale) ? cscale[j] : 1.0f, (HI && cscale) ? cscale[j + 32] : 1.0f};
    if constexpr (!PART && G2_LDS_STORE && g2_lds_bytes(NSUB, HI) >= 8 * 16384) {
        // round-4 experiment (build flag, not adopted): the tile leaves through LDS with 16-byte stores instead of 128
        // four-byte stores per lane.  The stores are 10-13 % of a pass (CNMF_G2_NOSTORE ablation: pass A 325 -> 291 us,
        // pass B 197 -> 171 us) but this form measured 0.4 % SLOWER end to end.  The DMA images are dead here (both wave
        // groups are past their last fragment read), every wave transposes its own 128 components x 64 cells in two rounds
        // of 64 x 64 through a private 16 KB strip and writes rows of 256 contiguous bytes.  Same values, same places.
        float* tw = reinterpret_cast<float*>(smem) + wave * 4096;
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int m = 2 * mp + mm;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = 4 * h + (r & 3) + 8 * (r >> 2);
                        tw[(mm * 32 + rl) * 64 + n * 32 + li] = (acc[m][n][r] * rscale[m0 + grp * 128 + m * 32 + rl]) * cs[n];
                    }
            }
            __builtin_amdgcn_wave_barrier();                 // (LDS operations of one wave execute in order)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 4 * i + (lane >> 4);
                const f32x4_g v = *reinterpret_cast<const f32x4_g*>(tw + c * 64 + 4 * (lane & 15));
                *reinterpret_cast<f32x4_g*>(C + (size_t)(m0 + grp * 128 + mp * 64 + c) * ldc + j0 + wn * 64 + 4 * (lane & 15)) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (!G2_LIVE(m)) continue;                       // nobody reads the products of dead columns
        const int cbase = m0 + grp * 128 + m * 32 + 4 * h;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = cbase + (r & 3) + 8 * (r >> 2);
                C[(size_t)row * ldc + j + n * 32] = (acc[m][n][r] * rscale[row]) * cs[n];
            }
    }
    }
#undef G2_LIVE
}

// pass B (and pass A on few tiles): split-K launch, XCD-aware order as gemm3c_kernel.  kb_per is a multiple of NSUB.
template <int NSUB, bool HI, int VAR = 0, bool PART = false, bool GEN = false>
__global__ __launch_bounds__(512) void gemm2h_kernel(const unsigned char* __restrict__ A2,
                                                     const unsigned char* __restrict__ B1,
                                                     const unsigned char* __restrict__ Bhi,
                                                     const unsigned int* __restrict__ hiflag,
                                                     const float* __restrict__ rscale, int Kb,
                                                     float* __restrict__ C, int ldc, long long c_split_stride,
                                                     int kb_per, const float* __restrict__ cscale = nullptr,
                                                     unsigned long long livemask = ~0ull)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    int jt = blockIdx.x, mg = blockIdx.y, z = blockIdx.z;
    if ((gridDim.z & 7) == 0) {
        // XCD-aware order (block L runs on XCD L % 8, each XCD has its own L2): one XCD gets ALL row tiles jt and ALL
        // component groups mg of the K splits z = xcd, xcd + 8, ... -- the workgroups that share a factor segment
        // (same mg, z) or a count-plane segment (same jt, z) then share it in one L2.
        const int L = blockIdx.x + (int)gridDim.x * (blockIdx.y + (int)gridDim.y * blockIdx.z);
        const int xcd = L & 7, idx = L >> 3;
        jt = idx % (int)gridDim.x;
        mg = (idx / (int)gridDim.x) % (int)gridDim.y;
        z = xcd + 8 * (idx / (int)(gridDim.x * gridDim.y));
    }
    const int kb0 = z * kb_per;
    const int nkb = min(kb_per, Kb - kb0);
    gemm2h_segment<NSUB, HI, VAR, true, PART, GEN>(A2, B1, Bhi, hiflag, rscale, Kb, C ? C + (size_t)z * c_split_stride : nullptr, ldc, mg * G3_MW,
                                              jt * G3C_JW, kb0, nkb, smem3, cscale, (unsigned)((livemask >> (8 * mg)) & 0xffu));
}

// pass A: stream-K over persistent workgroups, unit = one step of NSUB blocks (Kb % NSUB == 0)
template <int NSUB, bool HI, int VAR = 0, bool NTB = true, bool PART = false, bool GEN = false, class EPI = G2NoEpi>
__global__ __launch_bounds__(512) void gemm2h_streamk_kernel(const unsigned char* __restrict__ A2,
                                                             const unsigned char* __restrict__ B1,
                                                             const unsigned char* __restrict__ Bhi,
                                                             const unsigned int* __restrict__ hiflag,
                                                             const float* __restrict__ rscale, int Kb,
                                                             float* __restrict__ C0, float* __restrict__ C1,
                                                             float* __restrict__ C2, int ldc, int MG, int T, int xmap,
                                                             const float* __restrict__ cscale = nullptr,
                                                             unsigned long long livemask = ~0ull, EPI epi = EPI{})
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int Ks = Kb / NSUB;                             // steps per tile
    const long long U = (long long)T * Ks;
    // Work order: component group major (tile o = mg * NJ + jt), and -- xmap -- workgroup p takes the q-th share with
    // q = (p % MG) * (P / MG) + p / MG: block p runs on XCD p % 8, so each XCD works inside ONE component group and its
    // 4 MB L2 keeps that group's factor planes (2 MB at 2000 genes; the planes of four groups would thrash it: 1.5 GB
    // fetched per launch at 1024 columns, PMC), while neighbouring XCDs stream the SAME row tile jt of the count plane at
    // the same time (one HBM fetch, the others find it in the Infinity Cache).  The cut flags the sweep reads are indexed
    // jt * MG + mg (plan_streamk3); the set of cut positions does not depend on the permutation.
    const int P = (int)gridDim.x;
    const int q = (xmap && MG > 1 && P % MG == 0) ? ((int)blockIdx.x % MG) * (P / MG) + (int)blockIdx.x / MG : (int)blockIdx.x;
    long long u = U * q / P;
    const long long u1 = U * (q + 1) / P;
    const int NJ = T / MG;
    while (u < u1) {
        const int tile = (int)(u / Ks), ks = (int)(u % Ks);
        const int ke = (int)min((long long)Ks, ks + (u1 - u));
        const int mg = tile / NJ, jt = tile % NJ;
        gemm2h_segment<NSUB, HI, VAR, NTB, PART, GEN, EPI>(A2, B1, Bhi, hiflag, rscale, Kb, (ks == 0) ? C0 : (ke == Ks ? C1 : C2), ldc, mg * G3_MW,
                                                 jt * G3C_JW, ks * NSUB, (ke - ks) * NSUB, smem3, cscale,
                                                 (unsigned)((livemask >> (8 * mg)) & 0xffu), &epi, ks == 0 && ke == Ks);
        u += ke - ks;
        G3_WAIT_VM(0);
        __syncthreads();                 // the images are refilled by the next segment's DMA
    }
}

}  // namespace cnmf
