// Split-operand MFMA GEMM: f32-accurate products on the bf16 matrix pipe (gfx950 only).
//
// The exact-f32 MFMA (kernels_gemm.hip.h) runs at 1/16 of the bf16 MFMA rate.  Here every f32
// operand x is held as THREE bf16 planes  x = h + m + l  (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m); 3 x 8 significand bits >= the 24 of f32, the subtractions are exact), and a
// product a*b is accumulated in f32 from the six partial products whose weight is >= 2^-18:
//
//      a*b  ~=  ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)
//
// The three dropped ones (am*bl, al*bm, al*bl) are below 2^-25 |a*b|, i.e. below the rounding of
// the f32 product itself, so the result carries f32-class error (tests/test_gpu_nmf.py holds it to
// the same 2e-6 bound as the f32 pipe).  6 bf16 MFMAs per 16 k at 16x the f32 rate = 2.67x the
// f32 matrix peak.
//
// Both passes are "NT" products  C[c][j] = sum_k A[c][k] * B[j][k]  with K contiguous in both
// operands: pass A uses planes of X (cells x genes), pass B planes of X^T (genes x cells) -- the
// data matrix is resident twice (288 GB of HBM; 1.2 GB at 50k x 2000).
//
// Plane layout in memory (A and B alike), "block-major":
//      [row tile of TR rows][16-k block kb][row in tile][h: 16 bf16 | m: 16 bf16 | l: 16 bf16]
// i.e. one row of one block is 96 contiguous bytes and ONE (tile, block) is ONE contiguous run of
// TR * 96 bytes (TR = 256 for the packed factor, 128 for X / X^T) in exactly the order the kernel wants
// it in LDS.  A stage is therefore fetched as whole 128-byte lines, 1 KB per wave instruction (rows
// of a row-major layout would be 96-byte segments 12 KB apart: twice the L1/TA line requests).
// Inside a row the two 8-k halves of every plane are SWAPPED for rows with bit 3 set (G3_SWZ): a
// ds_read_b128 lane group holds 16 rows of one half; with dense 96-byte rows their bank groups are
// 24 r mod 64 = only 8 distinct values (2-way conflicts), and moving half of the rows by 16 bytes makes
// all 16 distinct -- conflict-free fragment reads straight from the lane-linear DMA image, no padding.
// The register-staged variant un-swaps while writing its padded (112 B = 28 dword) LDS rows.
//
// v_mfma_f32_32x32x16_bf16: lane l supplies row/col (l & 31) and the 8 k's of half (l >> 5) of the
// block; A and B use the same assignment, so the order of k inside a block is immaterial.
// Workgroup = 4 waves, tile 256 components x 128 j; wave (wm, wn) owns 128 x 64 = 4 x 2 MFMA tiles
// (128 accumulator registers).  X is read ONCE per pass for all 256 packed columns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnmf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef float f32x16_3 __attribute__((ext_vector_type(16)));

constexpr int G3_MW = 256;            // component rows per workgroup tile
constexpr int G3_JW = 128;            // j columns per workgroup tile
constexpr int G3_BK = 16;             // k per stage = one MFMA block
constexpr int G3_ROWB = 96;           // bytes of one row-block in memory
constexpr int G3_LDSROW = 112;        // bytes of one row in LDS
constexpr int G3_ROWS = G3_MW + G3_JW;
constexpr int G3_LDS_BYTES = G3_ROWS * G3_LDSROW;       // 43 008 B -> 2 workgroups per CU
constexpr int G3_CHUNKS = G3_ROWS * 6 / 256;            // 16-byte chunks staged per thread (9)

// 1 if the 8-k halves of row r (index within its row tile) are stored swapped
#define G3_SWZ(r_) (((r_) >> 3) & 1)

__device__ __forceinline__ unsigned short bf16_rne(float x)
{
    unsigned int u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// x -> (h, m, l)
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l)
{
    h = bf16_rne(x);
    const float r1 = x - bf16_to_f32(h);
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(m);
    l = bf16_rne(r2);
}

// src [rows][ld] f32 (K-contiguous)  ->  block-major planes with row tiles of TR rows.
// One thread per (row, 16-k block): reads 64 contiguous bytes, writes the row's 96 bytes of that block.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, int ld, int rows, int K, int TR,
                                                     unsigned short* __restrict__ dst,
                                                     const double* __restrict__ kscale = nullptr)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Kb = K / 16;
    if (t >= (long long)rows * Kb) return;
    const int row = (int)(t / Kb), kb = (int)(t % Kb);
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)row * ld + kb * 16);
    float x[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 v = s4[q]; x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
    if (kscale) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = (float)((double)x[i] * kscale[kb * 16 + i]);
    }
    unsigned short p[3][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) split3(x[i], p[0][i], p[1][i], p[2][i]);
    unsigned short* d = dst + (((size_t)(row / TR) * Kb + kb) * TR + (row % TR)) * 48;
    const int swz = G3_SWZ(row % TR);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            u32x4 w;
            w.x = p[q][8 * hf + 0] | ((unsigned)p[q][8 * hf + 1] << 16); w.y = p[q][8 * hf + 2] | ((unsigned)p[q][8 * hf + 3] << 16);
            w.z = p[q][8 * hf + 4] | ((unsigned)p[q][8 * hf + 5] << 16); w.w = p[q][8 * hf + 6] | ((unsigned)p[q][8 * hf + 7] << 16);
            *reinterpret_cast<u32x4*>(d + q * 16 + (hf ^ swz) * 8) = w;
        }
}

// planes of the TRANSPOSE: src [K rows][ld] f32 (J-contiguous, J = dst rows) -> block-major planes of the
// J x K matrix.  One thread per (dst row j, 16-k block); lanes run along j so the reads coalesce.  One-off.
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ src, int ld, int src_rows,
                                                               int J, int K, int TR, unsigned short* __restrict__ dst)
{
    const int j = blockIdx.y * 256 + threadIdx.x;       // (grid = (16-k blocks, row groups): the long dimension on x -- no 65 535 limit)
    const int kb = blockIdx.x;
    if (j >= J) return;
    const int Kb = K / 16;
    unsigned short p[3][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = kb * 16 + i;
        const float x = (k < src_rows) ? src[(size_t)k * ld + j] : 0.f;
        split3(x, p[0][i], p[1][i], p[2][i]);
    }
    unsigned short* d = dst + (((size_t)(j / TR) * Kb + kb) * TR + (j % TR)) * 48;
    const int swz = G3_SWZ(j % TR);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            u32x4 w;
            w.x = p[q][8 * hf + 0] | ((unsigned)p[q][8 * hf + 1] << 16); w.y = p[q][8 * hf + 2] | ((unsigned)p[q][8 * hf + 3] << 16);
            w.z = p[q][8 * hf + 4] | ((unsigned)p[q][8 * hf + 5] << 16); w.w = p[q][8 * hf + 6] | ((unsigned)p[q][8 * hf + 7] << 16);
            *reinterpret_cast<u32x4*>(d + q * 16 + (hf ^ swz) * 8) = w;
        }
}

// The same through LDS for the per-iteration split of the packed factors: a workgroup converts 64 rows
// x 64 k (four blocks).  Reads: 16 lanes cover 256 contiguous bytes of a row; writes: the 64 rows of
// one block are 6 KB contiguous in the block-major layout.  (rows, K, TR multiples of 64.)
// `tile` = unsigned short [4][64][48] ([block][row][h16|m16|l16], 24 KB of LDS)
__device__ __forceinline__ void split3_tiled_body(const float* __restrict__ src, int ld, int K, int TR,
                                                  unsigned short* __restrict__ dst, const double* __restrict__ kscale,
                                                  int bx, int by, unsigned short (*tile)[64][48])
{
    const int t = threadIdx.x;
    const int k0 = bx * 64, r0 = by * 64;
    const int kq = t & 15, rr = t >> 4;                  // float4 index along k, row within a pass of 16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rr + 16 * i;
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(r0 + row) * ld + k0 + kq * 4);
        float x[4] = {v.x, v.y, v.z, v.w};
        if (kscale) {                                     // count-structured data: the per-gene scale rides on the factor
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (float)((double)x[e] * kscale[k0 + kq * 4 + e]);
        }
        unsigned short p[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(x[e], p[0][e], p[1][e], p[2][e]);
        // (r0 and TR are multiples of 64: bit 3 of the in-tile row is bit 3 of the local row)
        unsigned short* d = &tile[kq >> 2][row][((((kq & 3) >> 1) ^ G3_SWZ(row)) * 8) + (kq & 1) * 4];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            uint2 w;
            w.x = p[q][0] | ((unsigned)p[q][1] << 16); w.y = p[q][2] | ((unsigned)p[q][3] << 16);
            *reinterpret_cast<uint2*>(d + q * 16) = w;
        }
    }
    __syncthreads();
    const int Kb = K / 16;
    const int tr = r0 / TR, rin = r0 % TR;               // the 64 rows lie in one row tile (TR % 64 == 0)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        unsigned short* g = dst + (((size_t)tr * Kb + (k0 / 16 + b)) * TR + rin) * 48;
        const u32x4* s4 = reinterpret_cast<const u32x4*>(&tile[b][0][0]);          // 64 rows x 96 B = 384 chunks
        u32x4* g4 = reinterpret_cast<u32x4*>(g);
        g4[t] = s4[t];
        if (t < 128) g4[256 + t] = s4[256 + t];
    }
}

__global__ __launch_bounds__(256) void split3_tiled_kernel(const float* __restrict__ src, int ld, int K, int TR,
                                                           unsigned short* __restrict__ dst,
                                                           const double* __restrict__ kscale = nullptr)
{
    __shared__ __attribute__((aligned(16))) unsigned short tile[4][64][48];
    split3_tiled_body(src, ld, K, TR, dst, kscale, blockIdx.x, blockIdx.y, tile);
}

// One K segment [kb0, kb0 + nkb) (in 16-k blocks) of one 256 x 128 tile, stored to C.
//   A3 : planes of the component-major factor, rows m0.. (KC rows in total), Kb blocks per row
//   B3 : planes of X (or X^T), rows j0..
// Rows of B beyond the allocation are never touched: the caller pads the plane buffers to whole tiles.
__device__ __forceinline__ void gemm3_segment(const unsigned char* __restrict__ A3, const unsigned char* __restrict__ B3,
                                              int Kb, float* __restrict__ C, int ldc, int m0, int j0, int kb0,
                                              int nkb, unsigned char* smem)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;

    // ---- staging map: chunk c = tid + 256*i ; row = c / 6 ; part = c % 6 ; rows 0..255 = A, 256..383 = B
    // (two 64-bit bases + 32-bit per-chunk offsets: a tile's rows span < 2^31 bytes)
    // block-major planes: block (tile, kb) is one contiguous run, chunk c of it at byte 16 c
    const unsigned char* abase = A3 + ((size_t)(m0 / G3_MW) * Kb + kb0) * (G3_MW * G3_ROWB) + tid * 16;
    const unsigned char* bbase = B3 + ((size_t)(j0 / G3_JW) * Kb + kb0) * (G3_JW * G3_ROWB) + tid * 16;
    int lds_off[G3_CHUNKS];
#pragma unroll
    for (int i = 0; i < G3_CHUNKS; ++i) {
        const int c = tid + 256 * i;
        const int row = c / 6, part = c - row * 6;
        // rows of the A region start at 0, of the B region at 256: bit 3 of the in-tile row = bit 3 of `row`
        lds_off[i] = row * G3_LDSROW + (part ^ G3_SWZ(row)) * 16;
    }
#define G3_SRC(i_) ((i_) < 6 ? abase + (i_) * 4096 : bbase + ((i_) - 6) * 4096)
#define G3_BLK(i_) ((i_) < 6 ? G3_MW * G3_ROWB : G3_JW * G3_ROWB)
    u32x4 stage[G3_CHUNKS];

    f32x16_3 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const unsigned char* a_lds = smem + (wm * 128 + li) * G3_LDSROW + h * 16;
    const unsigned char* b_lds = smem + (G3_MW + wn * 64 + li) * G3_LDSROW + h * 16;

#pragma unroll
    for (int i = 0; i < G3_CHUNKS; ++i) stage[i] = *reinterpret_cast<const u32x4*>(G3_SRC(i));
#pragma unroll
    for (int i = 0; i < G3_CHUNKS; ++i) *reinterpret_cast<u32x4*>(smem + lds_off[i]) = stage[i];
    __syncthreads();

    for (int s = 0; s < nkb; ++s) {
        // prefetch the next block into registers (clamped on the last stage: a harmless re-read)
        const int sn = (s + 1 < nkb) ? s + 1 : s;
#pragma unroll
        for (int i = 0; i < G3_CHUNKS; ++i)
            stage[i] = *reinterpret_cast<const u32x4*>(G3_SRC(i) + (size_t)sn * G3_BLK(i));
        __builtin_amdgcn_sched_barrier(0);

        bf16x8 bq[2][3];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bq[n][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(b_lds + n * 32 * G3_LDSROW + p * 32));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            bf16x8 aq[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                aq[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_lds + m * 32 * G3_LDSROW + p * 32));
            // smallest terms first, alternating accumulators so that dependent MFMAs are two apart
#define G3_MFMA(pa, pb)                                                                                   \
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[pa], bq[0][pb], acc[m][0], 0, 0, 0);   \
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[pa], bq[1][pb], acc[m][1], 0, 0, 0);
            G3_MFMA(0, 2) G3_MFMA(1, 1) G3_MFMA(2, 0)
            G3_MFMA(0, 1) G3_MFMA(1, 0)
            G3_MFMA(0, 0)
#undef G3_MFMA
        }
        __syncthreads();                                   // every wave is done reading this block
#pragma unroll
        for (int i = 0; i < G3_CHUNKS; ++i) *reinterpret_cast<u32x4*>(smem + lds_off[i]) = stage[i];
        __syncthreads();
    }

#undef G3_SRC
#undef G3_BLK
    const int j = j0 + wn * 64 + li;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int cbase = m0 + wm * 128 + m * 32 + 4 * h;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = cbase + (r & 3) + 8 * (r >> 2);
                C[(size_t)row * ldc + j + n * 32] = acc[m][n][r];
            }
    }
}

// grid-mapped launch: blockIdx = (j tile, component group of 256, K split)
__global__ __launch_bounds__(256, 2) void gemm3_kernel(const unsigned char* __restrict__ A3,
                                                       const unsigned char* __restrict__ B3, int Kb,
                                                       float* __restrict__ C, int ldc, long long c_split_stride,
                                                       int kb_per)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int kb0 = blockIdx.z * kb_per;
    const int nkb = min(kb_per, Kb - kb0);
    gemm3_segment(A3, B3, Kb, C + (size_t)blockIdx.z * c_split_stride, ldc, blockIdx.y * G3_MW,
                  blockIdx.x * G3_JW, kb0, nkb, smem3);
}

// stream-K launch: gridDim.x persistent workgroups share T tiles x Kb blocks evenly.  With
// gridDim.x <= 2 T a tile is cut at most twice: the piece that starts at block 0 goes to plane C0, the
// piece that ends at Kb to C1, a piece in the middle to C2 (sweep_kernel adds the planes the tile's
// flags name).  Fixed piece -> plane mapping and fixed summation order: bit-reproducible.
__global__ __launch_bounds__(256, 2) void gemm3_streamk_kernel(const unsigned char* __restrict__ A3,
                                                               const unsigned char* __restrict__ B3, int Kb,
                                                               float* __restrict__ C0, float* __restrict__ C1,
                                                               float* __restrict__ C2, int ldc, int MG, int T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const long long U = (long long)T * Kb;
    long long u = U * blockIdx.x / gridDim.x;
    const long long u1 = U * (blockIdx.x + 1) / gridDim.x;
    while (u < u1) {
        const int tile = (int)(u / Kb), kb = (int)(u % Kb);
        const int ke = (int)min((long long)Kb, kb + (u1 - u));
        const int mg = tile / (T / MG), jt = tile % (T / MG);     // component-group-major work order (see gemm2h_streamk_kernel)
        gemm3_segment(A3, B3, Kb, (kb == 0) ? C0 : (ke == Kb ? C1 : C2), ldc, mg * G3_MW, jt * G3_JW, kb,
                      ke - kb, smem3);
        u += ke - kb;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA ping-pong variant (the production kernel).  ONE workgroup of 8 waves per CU, as two groups
// of 4 waves that own the SAME 256 x 128 tile and alternate k blocks (group 0 even, group 1 odd): the
// matrix pipe never sees two equal bursts at once (two independent workgroups per CU fall into
// lockstep: both in the MFMA burst, then both in the memory phase).  Blocks go global -> LDS directly
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into FOUR dense 36 KB images (two
// per group), so a block is requested three phases (~5000 cycles) before it is multiplied.  At the
// end group 1 hands its partial accumulators to group 0 through LDS (thread t of either group owns
// the same tile elements): a fixed two-term sum.  One raw s_barrier per phase, placed in the middle
// of the computing group's 48 MFMAs: the other group's fragment reads overlap that tail, so the
// matrix pipe goes from one group's burst straight into the other's.
//   * LDS image of a block: chunk c = row*6 + part at byte 16 c (lane-linear, as the DMA writes it);
//     dense 96-byte rows make ds_read_b128 2-way bank conflicted -- irrelevant at < 40 % LDS load.
//   * vmcnt is counted by hand (9 DMA instructions per block and wave): "my next block has landed"
//     = vmcnt(9) while the block after it is still in flight.  __syncthreads() would drain it.
constexpr int G3G_BLK = G3_ROWS * G3_ROWB;              // 36 864 B
constexpr int G3G_LDS_BYTES = 4 * G3G_BLK;              // 147 456 B (the 128 KB exchange area overlays it)

#define G3_AS1(p_) ((const __attribute__((address_space(1))) void*)(p_))
#define G3_AS3(p_) ((__attribute__((address_space(3))) void*)(p_))
#define G3_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")
#define G3_RAW_BARRIER()                                   \
    {                                                      \
        asm volatile("" ::: "memory");                     \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory");                     \
    }

__device__ __forceinline__ void gemm3g_segment(const unsigned char* __restrict__ A3, const unsigned char* __restrict__ B3,
                                               int Kb, float* __restrict__ C, int ldc, int m0, int j0, int kb0,
                                               int nkb, unsigned char* smem)
{
    const int grp = threadIdx.x >> 8;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    unsigned char* gbuf = smem + grp * 2 * G3G_BLK;

    // block-major planes: block (tile, kb) is one contiguous run and IS the LDS image (chunk c at byte 16 c)
    const unsigned char* abase = A3 + ((size_t)(m0 / G3_MW) * Kb + kb0 + grp) * (G3_MW * G3_ROWB) + tid * 16;
    const unsigned char* bbase = B3 + ((size_t)(j0 / G3_JW) * Kb + kb0 + grp) * (G3_JW * G3_ROWB) + tid * 16;
    f32x16_3 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int hs = h ^ G3_SWZ(li);                       // every row offset below is a multiple of 32
    const int a_off = (wm * 128 + li) * G3_ROWB + hs * 16;
    const int b_off = (G3_MW + wn * 64 + li) * G3_ROWB + hs * 16;
    const int n_own = (nkb - grp + 1) >> 1;                 // this group's blocks: grp, grp + 2, ...

#define G3_SRC(i_) ((i_) < 6 ? abase + (i_) * 4096 + oa_ : bbase + ((i_) - 6) * 4096 + ob_)
#define G3G_ISSUE(own_)                                                                            \
    {                                                                                              \
        const size_t oa_ = (size_t)(own_) * (2 * G3_MW * G3_ROWB), ob_ = (size_t)(own_) * (2 * G3_JW * G3_ROWB); \
        unsigned char* d_ = gbuf + ((own_) & 1) * G3G_BLK + wave * 1024;                           \
        _Pragma("unroll") for (int i = 0; i < G3_CHUNKS; ++i) {                                    \
            /* the X planes are read once: non-temporal, so that they do not evict the factor planes */ \
            if (i >= 6) __builtin_amdgcn_global_load_lds(G3_AS1(G3_SRC(i)), G3_AS3(d_ + i * 4096), 16, 0, 2); \
            else        __builtin_amdgcn_global_load_lds(G3_AS1(G3_SRC(i)), G3_AS3(d_ + i * 4096), 16, 0, 0); \
        }                                                                                          \
    }
#define G3_FRAG(ptr_) __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ptr_))
#define G3_MFMA(m_, aq_, pa, pb)                                                                   \
    acc[m_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq_[pa], bq[0][pb], acc[m_][0], 0, 0, 0); \
    acc[m_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq_[pa], bq[1][pb], acc[m_][1], 0, 0, 0);
#define G3_MFMA6(m_, aq_)                                                                          \
    G3_MFMA(m_, aq_, 0, 2) G3_MFMA(m_, aq_, 1, 1) G3_MFMA(m_, aq_, 2, 0)                           \
    G3_MFMA(m_, aq_, 0, 1) G3_MFMA(m_, aq_, 1, 0) G3_MFMA(m_, aq_, 0, 0)

// two component tiles at a time: consecutive MFMAs cycle through FOUR accumulators, so an accumulator
// is touched again only after three other 8-pass MFMAs
#define G3_MFMA2(ma_, mb_, pa, pb)                                                                 \
    G3_MFMA(ma_, aq[ma_], pa, pb) G3_MFMA(mb_, aq[mb_], pa, pb)
#define G3_MFMA12(ma_, mb_)                                                                        \
    G3_MFMA2(ma_, mb_, 0, 2) G3_MFMA2(ma_, mb_, 1, 1) G3_MFMA2(ma_, mb_, 2, 0)                     \
    G3_MFMA2(ma_, mb_, 0, 1) G3_MFMA2(ma_, mb_, 1, 0) G3_MFMA2(ma_, mb_, 0, 0)
    G3_WAIT_VM(0);                                          // stores of a previous segment
    if (n_own > 0) G3G_ISSUE(0)
    if (n_own > 1) G3G_ISSUE(1)
    if (grp == 0) { if (n_own > 1) G3_WAIT_VM(9); else G3_WAIT_VM(0); }
    G3_RAW_BARRIER()

    bf16x8 bq[2][3], aq[4][3];
    for (int p = 0; p < nkb; ++p) {
        const bool mine = ((p & 1) == grp);
        const int own = p >> 1;
        if (mine) {
            // all 18 fragment reads first (they return in order: the B and m = 0 fragments feed the first
            // MFMAs while the rest land), pinned so that the scheduler cannot sink them between MFMAs
            const unsigned char* bb = gbuf + (own & 1) * G3G_BLK;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 3; ++q) bq[n][q] = G3_FRAG(bb + b_off + n * 32 * G3_ROWB + q * 32);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 3; ++q) aq[m][q] = G3_FRAG(bb + a_off + m * 32 * G3_ROWB + q * 32);
            __builtin_amdgcn_sched_barrier(0);
            G3_MFMA12(0, 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of this image has returned
        } else if (p + 1 < nkb) {
            const int j = (p + 1) >> 1;                     // my next block must have landed before the barrier
            if (j + 1 < n_own) G3_WAIT_VM(9); else G3_WAIT_VM(0);
        }
        G3_RAW_BARRIER()
        if (mine) {
            G3_MFMA12(2, 3)                                 // the second half of the burst overlaps the
                                                            // other group's fragment reads
            if (own + 2 < n_own) G3G_ISSUE(own + 2)         // refill the image just consumed
        }
    }
#undef G3_MFMA12
#undef G3_MFMA2
#undef G3_MFMA6
#undef G3_MFMA
#undef G3_FRAG
#undef G3G_ISSUE
#undef G3_SRC

    // ---- group 1 hands its partial sums to group 0 (nothing is in flight any more)
    __syncthreads();
    float* xch = reinterpret_cast<float*>(smem);
    if (grp == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((m * 2 + n) * 16 + r) * 256 + tid] = acc[m][n][r];
    }
    __syncthreads();
    if (grp == 0) {
        const int j = j0 + wn * 64 + li;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int cbase = m0 + wm * 128 + m * 32 + 4 * h;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = cbase + (r & 3) + 8 * (r >> 2);
                    C[(size_t)row * ldc + j + n * 32] = acc[m][n][r] + xch[((m * 2 + n) * 16 + r) * 256 + tid];
                }
        }
    }
}

__global__ __launch_bounds__(512) void gemm3g_kernel(const unsigned char* __restrict__ A3,
                                                     const unsigned char* __restrict__ B3, int Kb,
                                                     float* __restrict__ C, int ldc, long long c_split_stride,
                                                     int kb_per)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round robin, and every j tile of one K
    // split re-reads the same slice of A3.  Give each XCD whole K splits (all their j tiles), so a
    // slice is fetched into ONE L2 instead of eight.  Needs gridDim.z % 8 == 0.
    // (several component groups, gridDim.y > 1: one XCD gets all j tiles AND all groups of its K splits)
    int jt = blockIdx.x, mg = blockIdx.y, z = blockIdx.z;
    if ((gridDim.z & 7) == 0) {
        const int L = blockIdx.x + (int)gridDim.x * (blockIdx.y + (int)gridDim.y * blockIdx.z);
        const int xcd = L & 7, idx = L >> 3;
        jt = idx % (int)gridDim.x;
        mg = (idx / (int)gridDim.x) % (int)gridDim.y;
        z = xcd + 8 * (idx / (int)(gridDim.x * gridDim.y));
    }
    const int kb0 = z * kb_per;
    const int nkb = min(kb_per, Kb - kb0);
    gemm3g_segment(A3, B3, Kb, C + (size_t)z * c_split_stride, ldc, mg * G3_MW, jt * G3_JW, kb0, nkb,
                   smem3);
}

__global__ __launch_bounds__(512) void gemm3g_streamk_kernel(const unsigned char* __restrict__ A3,
                                                             const unsigned char* __restrict__ B3, int Kb,
                                                             float* __restrict__ C0, float* __restrict__ C1,
                                                             float* __restrict__ C2, int ldc, int MG, int T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const long long U = (long long)T * Kb;
    long long u = U * blockIdx.x / gridDim.x;
    const long long u1 = U * (blockIdx.x + 1) / gridDim.x;
    while (u < u1) {
        const int tile = (int)(u / Kb), kb = (int)(u % Kb);
        const int ke = (int)min((long long)Kb, kb + (u1 - u));
        const int mg = tile / (T / MG), jt = tile % (T / MG);     // component-group-major work order (see gemm2h_streamk_kernel)
        gemm3g_segment(A3, B3, Kb, (kb == 0) ? C0 : (ke == Kb ? C1 : C2), ldc, mg * G3_MW, jt * G3_JW, kb,
                       ke - kb, smem3);
        u += ke - kb;
        __syncthreads();                 // the exchange area is overwritten by the next segment's DMA
    }
}

// ------------------------------------------------------------------------------------------
// Count-structured data (kernels_counts.hip.h): B is ONE integer bf16 plane, A the factor's three planes.
// 3 MFMAs per product, every partial product exact.  With half the MFMAs per block the operand feed would
// be the limit on a 256 x 128 tile (the 24 KB of factor planes per block dominate), so this kernel works on
// 256 components x 256 j: 8 waves, wave (g, wn) owns rows 128 g.., columns 64 wn.., EVERY wave multiplies
// every k block -- 32 KB of operands per block for 2 x 12 MFMAs per SIMD lane-pair.  The two wave groups run
// half a block apart: ONE instruction stream, group 1 enters it one barrier later, so its X_s barrier is
// group 0's Y_s and its Y_s is group 0's X_{s+1} (s_barrier only counts arrivals):
//     event 2s   : group 0 passes X_s      group 1 passes Y_{s-1}
//     event 2s+1 : group 0 passes Y_s      group 1 passes X_s
// Between X_s and Y_s a group reads its 14 fragments of block s and issues its first 12 MFMAs while the other
// group issues its last 12 of the previous block.  Four 32 KB block images; block s+3 is requested after X_s
// into the image whose last reader (group 1, block s-1) waited for its fragments before event 2s; block s+1
// is awaited before Y_s (for group 1 that is event 2s+2, just before group 0 reads it).
// Operand planes: block-major with 256-row tiles on both sides; B rows are 32 dense bytes.
constexpr int G3C_JW = 256;
constexpr int G3C_A = G3_MW * G3_ROWB;                  // 24 576 B of factor planes per block
constexpr int G3C_B = G3C_JW * 32;                      //  8 192 B of one integer plane per block
constexpr int G3C_IMGS = 4;
// LDS: four images of 32 KB (factor planes + integer plane), 40 KB when the matrix has a second integer plane
constexpr int g3c_lds_bytes(bool with_hi) { return G3C_IMGS * (G3C_A + (with_hi ? 2 : 1) * G3C_B); }

// Bhi / hiflag: second integer plane (256 hi) and its flags, one bit per (tile row, block), (Kb + 31) / 32 words
// per tile row (nullptr: no second plane).  Only flagged blocks fetch it (a 5th DMA instruction) and spend 3 more
// MFMAs per product.  The flag bits of the segment are fetched BEFORE the first DMA (a load inside the loop would
// sit in the same vmcnt queue as the DMAs).  The counted vmcnt waits assume the minimum of four DMA instructions
// per block: with extra ones in flight they only wait longer.
__device__ __forceinline__ void gemm3c_segment(const unsigned char* __restrict__ A3, const unsigned char* __restrict__ B1,
                                               const unsigned char* __restrict__ Bhi,
                                               const unsigned int* __restrict__ hiflag,
                                               int Kb, float* __restrict__ C, int ldc, int m0, int j0, int kb0,
                                               int nkb, unsigned char* smem)
{
    const int tid = threadIdx.x;                         // 0..511
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wn = wave & 3;
    const int li = lane & 31, h = lane >> 5;
    const int img_bytes = G3C_A + (Bhi ? 2 : 1) * G3C_B;

    // chunk c = tid + 512 i of a block image: i < 3 -> factor planes, i = 3 -> integer plane, i = 4 -> its second plane
    const size_t jblk = ((size_t)(j0 / G3C_JW) * Kb + kb0);
    const unsigned char* abase = A3 + ((size_t)(m0 / G3_MW) * Kb + kb0) * G3C_A + tid * 16;
    const unsigned char* bbase = B1 + jblk * G3C_B + tid * 16;
    const unsigned char* hbase = Bhi ? Bhi + jblk * G3C_B + tid * 16 : nullptr;
    // flag bits of blocks kb0 .. kb0 + nkb - 1 (at most 5 words for up to 129 blocks; longer segments: all set)
    unsigned int fw[5] = {0u, 0u, 0u, 0u, 0u};
    if (Bhi) {
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

    const int a_off = (grp * 128 + li) * G3_ROWB + (h ^ G3_SWZ(li)) * 16;
    const int b_off = G3C_A + (wn * 64 + li) * 32 + h * 16;

#define G3C_FLAG(s_) (Bhi && ((fw[(fbit0 + (s_)) >> 5 > 4 ? 4 : (fbit0 + (s_)) >> 5] >> ((fbit0 + (s_)) & 31)) & 1u))
#define G3C_ISSUE(s_)                                                                              \
    {                                                                                              \
        unsigned char* d_ = smem + ((s_) % G3C_IMGS) * img_bytes + wave * 1024;                    \
        const unsigned char* a_ = abase + (size_t)(s_) * G3C_A;                                    \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                              \
            __builtin_amdgcn_global_load_lds(G3_AS1(a_ + i * 8192), G3_AS3(d_ + i * 8192), 16, 0, 0); \
        /* the integer planes are read once per pass: non-temporal */                              \
        __builtin_amdgcn_global_load_lds(G3_AS1(bbase + (size_t)(s_) * G3C_B), G3_AS3(d_ + 3 * 8192), 16, 0, 2); \
        if (G3C_FLAG(s_))                                                                          \
            __builtin_amdgcn_global_load_lds(G3_AS1(hbase + (size_t)(s_) * G3C_B), G3_AS3(d_ + 4 * 8192), 16, 0, 2); \
    }
#define G3_FRAG(ptr_) __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ptr_))
#define G3C_READ(s_)                                                                               \
    {                                                                                              \
        const unsigned char* bb = smem + ((s_) % G3C_IMGS) * img_bytes;                            \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) bq[n] = G3_FRAG(bb + b_off + n * 32 * 32);   \
        if (hi_blk) { _Pragma("unroll") for (int n = 0; n < 2; ++n) bh[n] = G3_FRAG(bb + b_off + G3C_B + n * 32 * 32); } \
        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                              \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) aq[m][q] = G3_FRAG(bb + a_off + m * 32 * G3_ROWB + q * 32); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
// two component tiles at a time, smallest plane first: consecutive MFMAs cycle through four accumulators
#define G3C_MFMA(b_, m_, q_)                                                                       \
    acc[m_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[m_][q_], b_[0], acc[m_][0], 0, 0, 0);  \
    acc[m_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[m_][q_], b_[1], acc[m_][1], 0, 0, 0);
#define G3C_MFMA6(b_, ma_, mb_)                                                                    \
    G3C_MFMA(b_, ma_, 2) G3C_MFMA(b_, mb_, 2) G3C_MFMA(b_, ma_, 1) G3C_MFMA(b_, mb_, 1) G3C_MFMA(b_, ma_, 0) G3C_MFMA(b_, mb_, 0)

    bf16x8 bq[2], bh[2], aq[4][3];
    G3_WAIT_VM(0);                                          // stores of a previous segment
    G3C_ISSUE(0)
    if (nkb > 1) G3C_ISSUE(1)
    if (nkb > 2) G3C_ISSUE(2)
    if (nkb > 2) G3_WAIT_VM(8); else if (nkb > 1) G3_WAIT_VM(4); else G3_WAIT_VM(0);      // block 0 landed
    if (grp == 1) G3_RAW_BARRIER()
    for (int s = 0; s < nkb; ++s) {
        const bool hi_blk = G3C_FLAG(s);
        G3_RAW_BARRIER()                                        // X_s
        if (s + 3 < nkb) G3C_ISSUE(s + 3)
        G3C_READ(s)
        G3C_MFMA6(bq, 0, 1)
        if (hi_blk) { G3C_MFMA6(bh, 0, 1) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this image is free once both groups pass here
        // block s+1 must have landed before Y_s; s+2, s+3 may stay in flight
        if (s + 1 < nkb) {
            if (s + 3 < nkb) G3_WAIT_VM(8); else if (s + 2 < nkb) G3_WAIT_VM(4); else G3_WAIT_VM(0);
        }
        G3_RAW_BARRIER()                                        // Y_s
        G3C_MFMA6(bq, 2, 3)
        if (hi_blk) { G3C_MFMA6(bh, 2, 3) }
    }
    if (grp == 0) G3_RAW_BARRIER()
#undef G3C_MFMA6
#undef G3C_MFMA
#undef G3C_READ
#undef G3_FRAG
#undef G3C_ISSUE
#undef G3C_FLAG

    const int j = j0 + wn * 64 + li;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int cbase = m0 + grp * 128 + m * 32 + 4 * h;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = cbase + (r & 3) + 8 * (r >> 2);
                C[(size_t)row * ldc + j + n * 32] = acc[m][n][r];
            }
    }
}

__global__ __launch_bounds__(512) void gemm3c_kernel(const unsigned char* __restrict__ A3,
                                                     const unsigned char* __restrict__ B1,
                                                     const unsigned char* __restrict__ Bhi,
                                                     const unsigned int* __restrict__ hiflag, int Kb,
                                                     float* __restrict__ C, int ldc, long long c_split_stride,
                                                     int kb_per)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    int jt = blockIdx.x, mg = blockIdx.y, z = blockIdx.z;       // XCD-aware order, as in gemm3g_kernel
    if ((gridDim.z & 7) == 0) {
        const int L = blockIdx.x + (int)gridDim.x * (blockIdx.y + (int)gridDim.y * blockIdx.z);
        const int xcd = L & 7, idx = L >> 3;
        jt = idx % (int)gridDim.x;
        mg = (idx / (int)gridDim.x) % (int)gridDim.y;
        z = xcd + 8 * (idx / (int)(gridDim.x * gridDim.y));
    }
    const int kb0 = z * kb_per;
    const int nkb = min(kb_per, Kb - kb0);
    gemm3c_segment(A3, B1, Bhi, hiflag, Kb, C + (size_t)z * c_split_stride, ldc, mg * G3_MW, jt * G3C_JW, kb0,
                   nkb, smem3);
}

__global__ __launch_bounds__(512) void gemm3c_streamk_kernel(const unsigned char* __restrict__ A3,
                                                             const unsigned char* __restrict__ B1,
                                                             const unsigned char* __restrict__ Bhi,
                                                             const unsigned int* __restrict__ hiflag, int Kb,
                                                             float* __restrict__ C0, float* __restrict__ C1,
                                                             float* __restrict__ C2, int ldc, int MG, int T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const long long U = (long long)T * Kb;
    long long u = U * blockIdx.x / gridDim.x;
    const long long u1 = U * (blockIdx.x + 1) / gridDim.x;
    while (u < u1) {
        const int tile = (int)(u / Kb), kb = (int)(u % Kb);
        const int ke = (int)min((long long)Kb, kb + (u1 - u));
        const int mg = tile / (T / MG), jt = tile % (T / MG);     // component-group-major work order (see gemm2h_streamk_kernel)
        gemm3c_segment(A3, B1, Bhi, hiflag, Kb, (kb == 0) ? C0 : (ke == Kb ? C1 : C2), ldc, mg * G3_MW, jt * G3C_JW, kb,
                       ke - kb, smem3);
        u += ke - kb;
        G3_WAIT_VM(0);
        __syncthreads();                 // the images are refilled by the next segment's DMA
    }
}

}  // namespace cnmf
