// Consensus-step kernels (gfx950), all arithmetic in float64.
//
// The consensus core of the reference (cnmf.py:871-936) is float64 end to end and its
// outputs are index-like (density filter, k-means labels, per-cluster medians), so the
// device path keeps float64: the stacked spectra are tiny (R <= ~5000 rows x 2000 genes)
// and the f64 matrix pipe (v_mfma_f64_16x16x4_f64, 78 TF) makes the R x R Gram matrix a
// ~1 ms kernel.  Restated functions:
//   row_norm / l2        cnmf.py:882
//   dist_sym             sklearn/metrics/pairwise.py:419-438  (-2 X.Xt + |x|^2 + |y|^2, clamp, diag=0, sqrt)
//   knn_density          cnmf.py:893-898  (sum of the n+1 smallest per row / n)
//   kmeans++ / Lloyd     sklearn/cluster/_kmeans.py:174-272, :624-752; _k_means_lloyd.pyx:26-219
//   cluster_median       pandas groupby().median(), cnmf.py:913
//   silhouette           sklearn/metrics/cluster/_unsupervised.py:141-201
//   residual_sq          cnmf.py:926-930
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnmf {

constexpr int KM_CID = 128;          // largest number of clusters (= largest rank, CNMF_KMAX): centre ids per init, per-cluster sums
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
struct KmState;
__device__ __forceinline__ bool km_tile_done(const KmState* st, int m0, int k, int n_init);

// ---------------------------------------------------------------- f64 MFMA GEMM, C = A . B^T
// A [M][lda], B [N][ldb] both K-contiguous, C [M][ldc].  Workgroup tile 64 x 64, 4 waves
// (2 x 2), each wave 32 x 32 = 2 x 2 MFMA tiles of 16 x 16 x 4.  M, N multiples of 64 and
// K multiple of 16 (callers zero-pad).
//   A operand lane l: A[i=l&15][k=l>>4]   B operand lane l: B[k=l>>4][j=l&15]
//   D reg r lane l  : row = (l>>4) + 4*r, col = l&15          (f64 layout differs from f32!)
constexpr int DBK = 16;
constexpr int DLD = DBK + 2;     // padded LDS row (doubles): 18*8 B = 144 B -> ds_read_b64 conflict-free

// All-pairs distances of the rows of A (sklearn/metrics/pairwise.py:419-438) in ONE launch: the same 64 x 64 MFMA tile
// loop over the LOWER-TRIANGULAR tiles only (blockIdx.x = by*(by+1)/2 + bx, bx <= by) with the distance epilogue
//   D[i][j] = sqrt(max(0, (-2 G_ij + sq_i) + sq_j)), D[i][i] = 0
// applied to both (i, j) and its mirror (j, i) -- G_ij and G_ji are the same sum of the same products, so the mirror
// is what the full Gram would have held; only the order in which the two norms are added follows the element's own
// (row, column), as in the two-sided epilogue this replaces.  The mirrored tile goes through LDS so that its stores
// are row-contiguous too.  Halves the f64 MFMA work of the dominant consensus kernel and drops one 2 x R^2 x 8 B pass.
__global__ __launch_bounds__(256) void dist_sym_kernel(const double* __restrict__ A, int lda, int K,
                                                       const double* __restrict__ sq, int R,
                                                       double* __restrict__ D, int ldd)
{
    constexpr int TLD = 65;
    __shared__ __attribute__((aligned(16))) double smem[4 * 64 * DLD > 64 * TLD ? 4 * 64 * DLD : 64 * TLD];
    double* As0 = smem;                      // [2][64*DLD]
    double* Bs0 = smem + 2 * 64 * DLD;       // [2][64*DLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    int by = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((by + 1) * (by + 2) / 2 <= (int)blockIdx.x) ++by;
    while (by * (by + 1) / 2 > (int)blockIdx.x) --by;
    const int bx = (int)blockIdx.x - by * (by + 1) / 2;
    const int m0 = by * 64, n0 = bx * 64;
    const int s_row = tid >> 3, s_k = (tid & 7) * 2;
    const double* a_src = A + (size_t)(m0 + s_row) * lda + s_k;
    const double* b_src = A + (size_t)(n0 + s_row) * lda + s_k;
    v2d ar[2], br[2];
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int nk = K / DBK;
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ar[i] = *reinterpret_cast<const v2d*>(a_src + (size_t)(32 * i) * lda + kt * DBK);
            br[i] = *reinterpret_cast<const v2d*>(b_src + (size_t)(32 * i) * lda + kt * DBK);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<v2d*>(&As0[buf * 64 * DLD + (s_row + 32 * i) * DLD + s_k]) = ar[i];
            *reinterpret_cast<v2d*>(&Bs0[buf * 64 * DLD + (s_row + 32 * i) * DLD + s_k]) = br[i];
        }
    };
    if (nk > 0) { load(0); store(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load(kt + 1);
        const double* as = &As0[buf * 64 * DLD + (wm * 32 + li) * DLD + lk];
        const double* bs = &Bs0[buf * 64 * DLD + (wn * 32 + li) * DLD + lk];
#pragma unroll
        for (int q = 0; q < DBK / 4; ++q) {
            const double a0 = as[q * 4], a1 = as[16 * DLD + q * 4];
            const double b0 = bs[q * 4], b1 = bs[16 * DLD + q * 4];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store(buf ^ 1);
        __syncthreads();
    }
    // direct tile (rows of the m block) + the raw Gram tile into LDS, transposed, for the mirror
    double* T = smem;                        // [64 cols of the tile][TLD]: T[c][r] = G[m0 + r][n0 + c]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm * 32 + i * 16 + lk + 4 * r, cl = wn * 32 + j * 16 + li;
                const int row = m0 + rl, col = n0 + cl;
                const double g = acc[i][j][r];
                T[cl * TLD + rl] = g;
                if (row < R && col < R) {
                    double dd = -2.0 * g;
                    dd += sq[row];
                    dd += sq[col];
                    dd = fmax(dd, 0.0);
                    if (row == col) dd = 0.0;
                    D[(size_t)row * ldd + col] = sqrt(dd);
                }
            }
    if (bx == by) return;                    // a diagonal tile is its own mirror (uniform)
    __syncthreads();
    const int cl = tid & 63;
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
        const int rl = rr * 4 + (tid >> 6);
        const int row = n0 + rl, col = m0 + cl;          // element (row, col) of D = mirror of G[col][row]
        if (row < R && col < R) {
            double dd = -2.0 * T[rl * TLD + cl];
            dd += sq[row];
            dd += sq[col];
            dd = fmax(dd, 0.0);
            D[(size_t)row * ldd + col] = sqrt(dd);
        }
    }
}

// Few-rows variant for the k-means products (M = 64 centre / candidate rows against all N rows):
// workgroup tile 64 x 16 so that the launch has N/16 workgroups instead of N/64.
// st != nullptr: rows are the centres of n_init k-means runs (k rows each); a tile whose runs have all finished is skipped.
__global__ __launch_bounds__(256) void dgemm_nt_small_kernel(const double* __restrict__ A, int lda,
                                                             const double* __restrict__ B, int ldb,
                                                             double* __restrict__ C, int ldc, int K,
                                                             const KmState* __restrict__ st, int k, int n_init)
{
    __shared__ __attribute__((aligned(16))) double As[2][64 * DLD];
    __shared__ __attribute__((aligned(16))) double Bs[2][16 * DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 16;
    if (st && km_tile_done(st, m0, k, n_init)) return;
    const int s_row = tid >> 3, s_k = (tid & 7) * 2;
    const double* a_src = A + (size_t)(m0 + s_row) * lda + s_k;
    const double* b_src = B + (size_t)(n0 + (s_row & 15)) * ldb + s_k;
    const bool b_thr = tid < 128;
    v2d ar[2], br = v2d{0.0, 0.0};
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
    const int nk = K / DBK;
    auto load = [&](int kt) {
        ar[0] = *reinterpret_cast<const v2d*>(a_src + kt * DBK);
        ar[1] = *reinterpret_cast<const v2d*>(a_src + (size_t)32 * lda + kt * DBK);
        if (b_thr) br = *reinterpret_cast<const v2d*>(b_src + kt * DBK);
    };
    auto store = [&](int buf) {
        *reinterpret_cast<v2d*>(&As[buf][s_row * DLD + s_k]) = ar[0];
        *reinterpret_cast<v2d*>(&As[buf][(s_row + 32) * DLD + s_k]) = ar[1];
        if (b_thr) *reinterpret_cast<v2d*>(&Bs[buf][s_row * DLD + s_k]) = br;
    };
    if (nk > 0) { load(0); store(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load(kt + 1);
        const double* as = &As[buf][(wave * 16 + li) * DLD + lk];
        const double* bs = &Bs[buf][li * DLD + lk];
#pragma unroll
        for (int q = 0; q < DBK / 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(as[q * 4], bs[q * 4], acc, 0, 0, 0);
        if (kt + 1 < nk) store(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        C[(size_t)(m0 + wave * 16 + lk + 4 * r) * ldc + n0 + li] = acc[r];
}

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ double block_sum(double v, double* red)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// out[r][:] = S[r][:] / sqrt(sum S[r]^2) into a zero-padded [Rp][ld] buffer; sq[r] = |out[r]|^2
__global__ __launch_bounds__(256) void l2_rows_kernel(const double* __restrict__ S, int R, int G,
                                                      double* __restrict__ out, int ld,
                                                      double* __restrict__ sq)
{
    __shared__ double red[4];
    const int r = blockIdx.x;
    double s = 0.0;
    for (int g = threadIdx.x; g < G; g += 256) { const double v = S[(size_t)r * G + g]; s += v * v; }
    const double nrm = sqrt(block_sum(s, red));
    double s2 = 0.0;
    for (int g = threadIdx.x; g < G; g += 256) {
        const double v = S[(size_t)r * G + g] / nrm;
        out[(size_t)r * ld + g] = v;
        s2 += v * v;
    }
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) sq[r] = s2;
}

// out[r][:] = (double) store[rows[r]][:]   (merged spectra of one k out of the resident float32 store; grid (G / 256, R))
__global__ __launch_bounds__(256) void gather_store_rows_kernel(const float* __restrict__ store, int G,
                                                                const long long* __restrict__ rows, double* __restrict__ out)
{
    const int g = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (g < G) out[(size_t)r * G + g] = (double)store[(size_t)rows[r] * G + g];
}

// the same without the normalisation (cnmf_pairwise_distances: euclidean_distances of the rows as they are)
__global__ __launch_bounds__(256) void copy_rows_sq_kernel(const double* __restrict__ S, int R, int G,
                                                           double* __restrict__ out, int ld,
                                                           double* __restrict__ sq)
{
    __shared__ double red[4];
    const int r = blockIdx.x;
    double s2 = 0.0;
    for (int g = threadIdx.x; g < G; g += 256) {
        const double v = S[(size_t)r * G + g];
        out[(size_t)r * ld + g] = v;
        s2 += v * v;
    }
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) sq[r] = s2;
}

// density[i] = (sum of the m smallest entries of row i) / n      (m = n+1, self distance 0 included)
// Exact selection by bisection on the IEEE bit pattern (non-negative doubles order like uint64).
// Fallback for more than 20 480 merged spectra: the ~64 selection passes re-read the row from global memory (it stays
// in L2) -- any R.
__global__ __launch_bounds__(256) void knn_density_kernel(const double* __restrict__ Dm, int ld, int R,
                                                          int m, int n, double* __restrict__ density)
{
    __shared__ double red[4];
    __shared__ int cnt_s[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const double* __restrict__ rowbuf = Dm + (size_t)i * ld;
    unsigned long long lo = 0ull, hi = 0x7ff0000000000000ull;   // find smallest T with count(x<=T) >= m
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int j = tid; j < R; j += 256) c += (__double_as_longlong(rowbuf[j]) <= (long long)mid) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) cnt_s[tid >> 6] = c;
        __syncthreads();
        c = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
        if (c >= m) hi = mid; else lo = mid + 1;
    }
    const double T = __longlong_as_double((long long)lo);
    double s = 0.0; int c = 0;
    for (int j = tid; j < R; j += 256) { const double v = rowbuf[j]; if (v < T) { s += v; ++c; } }
    s = block_sum(s, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) cnt_s[tid >> 6] = c;
    __syncthreads();
    c = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
    if (tid == 0) density[i] = (s + (double)(m - c) * T) / (double)n;
}

// The same selection with the row in REGISTERS (VPT values per thread, R <= 256 * VPT) and a 4-way search: three pivots
// per pass, their three counts packed into one 64-bit word (21 bits each) for a single reduction -- 32 passes of
// register compares instead of 63 passes over memory, and no LDS footprint to limit occupancy.  The final sum runs in
// the order of the fallback kernel (thread-strided, then block_sum), so the density is bit-identical to it.
template <int VPT>
__global__ __launch_bounds__(256) void knn_density_reg_kernel(const double* __restrict__ Dm, int ld, int R,
                                                              int m, int n, double* __restrict__ density)
{
    __shared__ double red[4];
    __shared__ unsigned long long cnt_s[2][4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const double* __restrict__ grow = Dm + (size_t)i * ld;
    long long key[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int j = tid + 256 * v;
        key[v] = (j < R) ? __double_as_longlong(grow[j]) : 0x7fffffffffffffffll;
    }
    unsigned long long lo = 0ull, hi = 0x7ff0000000000000ull;   // smallest T with count(x <= T) >= m
    int pass = 0;
    while (lo < hi) {
        const unsigned long long q = (hi - lo) >> 2;
        const long long p1 = (long long)(lo + q), p2 = (long long)(lo + 2 * q), p3 = (long long)(lo + 3 * q);
        unsigned long long c = 0ull;
#pragma unroll
        for (int v = 0; v < VPT; ++v)
            c += (key[v] <= p1 ? 1ull : 0ull) + (key[v] <= p2 ? (1ull << 21) : 0ull) + (key[v] <= p3 ? (1ull << 42) : 0ull);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        unsigned long long* cs = cnt_s[pass & 1];             // two buffers: one barrier per pass
        if ((tid & 63) == 0) cs[tid >> 6] = c;
        __syncthreads();
        c = cs[0] + cs[1] + cs[2] + cs[3];
        const int c1 = (int)(c & 0x1fffff), c2 = (int)((c >> 21) & 0x1fffff), c3 = (int)(c >> 42);
        if (c1 >= m) hi = (unsigned long long)p1;
        else if (c2 >= m) { lo = (unsigned long long)p1 + 1; hi = (unsigned long long)p2; }
        else if (c3 >= m) { lo = (unsigned long long)p2 + 1; hi = (unsigned long long)p3; }
        else lo = (unsigned long long)p3 + 1;
        ++pass;
    }
    const double T = __longlong_as_double((long long)lo);
    double s = 0.0; int c = 0;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const double x = __longlong_as_double(key[v]);
        if (tid + 256 * v < R && x < T) { s += x; ++c; }
    }
    s = block_sum(s, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __shared__ int ci[4];
    if ((tid & 63) == 0) ci[tid >> 6] = c;
    __syncthreads();
    c = ci[0] + ci[1] + ci[2] + ci[3];
    if (tid == 0) density[i] = (s + (double)(m - c) * T) / (double)n;
}

// gather rows: out[q][:] = in[idx[q]][:]   (zero-padded destination rows are cleared by the caller)
__global__ void gather_rows_kernel(const double* __restrict__ in, int ld_in, const int* __restrict__ idx,
                                   int nrows, int G, double* __restrict__ out, int ld_out)
{
    const int q = blockIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nrows && g < G) out[(size_t)q * ld_out + g] = in[(size_t)idx[q] * ld_in + g];
}

// The same statistics with the rows spread over many workgroups (col_stats_kernel walks all R rows with 4 row groups
// per 64 columns: 32 workgroups at G = 2000 -- 0.7 ms at R = 4900).  part[chunk][g] = sum over the chunk's rows of x
// (mean == nullptr) or (x - mean[g])^2; col_combine_f64_kernel adds the chunks in order and divides by the row count.
__global__ __launch_bounds__(256) void col_partial_f64_kernel(const double* __restrict__ X, int ld, int R, int G,
                                                              int rows_per_chunk, const double* __restrict__ mean,
                                                              double* __restrict__ part)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    const double mu = mean ? mean[g] : 0.0;
    double s = 0.0;
    if (mean) for (int r = r0; r < r1; ++r) { const double d = X[(size_t)r * ld + g] - mu; s += d * d; }
    else      for (int r = r0; r < r1; ++r) s += X[(size_t)r * ld + g];
    part[(size_t)blockIdx.y * G + g] = s;
}

__global__ __launch_bounds__(256) void col_combine_f64_kernel(const double* __restrict__ part, int chunks, int G,
                                                              double divisor, double* __restrict__ out)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += part[(size_t)c * G + g];
    out[g] = s / divisor;               // a true division, like numpy's mean (sum * (1/R) can differ by 1 ulp)
}

// X[r][g] -= mean[g];  sq[r] = |X[r]|^2
__global__ __launch_bounds__(256) void center_rows_kernel(double* __restrict__ X, int ld, int G,
                                                          const double* __restrict__ mean,
                                                          double* __restrict__ sq)
{
    __shared__ double red[4];
    const int r = blockIdx.x;
    double s = 0.0;
    for (int g = threadIdx.x; g < G; g += 256) {
        const double v = X[(size_t)r * ld + g] - mean[g];
        X[(size_t)r * ld + g] = v;
        s += v * v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) sq[r] = s;
}

// ---------------------------------------------------------------- k-means (all inits batched)
// The n_init runs of KMeans are independent given their random draws (whose count is data
// independent), so every kernel below processes ALL inits at once: blockIdx.z (or .y / .x where
// noted) is the init index.  Per-init arrays are laid out [init][...]; centre rows of init i are
// rows [i*k, (i+1)*k) of one tall matrix so that ONE f64 MFMA product serves every init.
struct KmState {          // device-resident scalars of one k-means run
    double pot;           // current potential
    double shift_tot;     // sum_j |new_j - old_j|^2 of the last Lloyd step
    double inertia;
    int changed;          // number of labels that changed in the last E step
    int n_empty;
    int cand[8];          // candidate row ids of the current k-means++ step
    int best;
    int done;             // Lloyd loop finished for this init (lloyd_decide_kernel sets it)
    int iters;            // Lloyd iterations run
    int strict;           // stopped because no label changed (no final E step needed)
};

__device__ __forceinline__ bool km_tile_done(const KmState* st, int m0, int k, int n_init)
{   // rows m0 .. m0+63 of the stacked centre matrix belong to inits m0/k .. (m0+63)/k
    const int i0 = m0 / k, i1 = min((m0 + 63) / k, n_init - 1);
    for (int i = i0; i <= i1; ++i) if (!st[i].done) return false;
    return true;
}

struct KmDims { int R, Rp, G, ld, k, L; };   // kept rows, padded rows, genes, padded genes, clusters, local trials

// ---------------------------------------------------------------- k-means++ in one launch
// The whole seeding of one init in ONE workgroup (sklearn _kmeans.py:174-272): every quantity it needs is a squared
// distance between two DATA points, and those are already on the device -- the all-pairs matrix of the L2-normalised
// spectra (distances do not change when the column means are subtracted).  So the k-1 draws need no product and no
// second launch: cumulative sum + searchsorted of the L trial values, min(closest, D[cand]^2) and its potential for the
// L candidates (one pass, one reduction), first arg-min, next draw.  ~10 us per draw instead of five dependent launches.
template <int NT>
__global__ __launch_bounds__(NT) void pp_fused_kernel(const double* __restrict__ Dm, int ldD,
                                                      const int* __restrict__ keep, KmDims d,
                                                      const int* __restrict__ c0, const double* __restrict__ u,
                                                      int ustride, double* closest, double* cum, double* dmin,
                                                      int* __restrict__ center_ids)
{
    constexpr int NW = NT / 64;
    __shared__ double part[256];
    __shared__ double red[NW][8];
    __shared__ int cand_s[8];
    const int init = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, R = d.R, L = d.L;
    double* cl = closest + (size_t)init * d.Rp;
    double* cm = cum + (size_t)init * d.Rp;
    double* dm = dmin + (size_t)init * 8 * d.Rp;
    double pot;
    {   // first centre: closest = D[c0]^2, pot = sum
        const int first = c0[init];
        const double* drow = Dm + (size_t)keep[first] * ldD;
        double s = 0.0;
        for (int r = tid; r < R; r += NT) { double v = drow[keep[r]]; v *= v; cl[r] = v; s += v; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave][0] = s;
        __syncthreads();
        pot = 0.0;
        for (int w = 0; w < NW; ++w) pot += red[w][0];
        if (tid == 0) center_ids[init * KM_CID] = first;
    }
    const int per = (R + 255) / 256;
    const int sb = min(tid * per, R), se = (tid < 256) ? min(sb + per, R) : sb;     // scan chunk of threads 0..255
    for (int c = 1; c < d.k; ++c) {
        __syncthreads();
        {   // cumulative sum of closest[]: 256 chunks, chunk totals scanned serially, like np.cumsum up to rounding
            double s = 0.0;
            for (int r = sb; r < se; ++r) s += cl[r];
            if (tid < 256) part[tid] = s;
            __syncthreads();
            if (tid == 0) {
                double run = 0.0;
#pragma unroll 32
                for (int t = 0; t < 256; ++t) { const double v = part[t]; part[t] = run; run += v; }
            }
            __syncthreads();
            if (tid < 256) { double run = part[tid]; for (int r = sb; r < se; ++r) { run += cl[r]; cm[r] = run; } }
        }
        __syncthreads();
        if (tid < L) {      // searchsorted(cum, u * pot, side='left'), clipped
            const double v = u[(size_t)init * ustride + 1 + (size_t)(c - 1) * L + tid] * pot;
            int lo = 0, hi = R;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cm[mid] < v) lo = mid + 1; else hi = mid; }
            cand_s[tid] = min(lo, R - 1);
        }
        __syncthreads();
        const double* drow[8];
        double s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.0; drow[j] = Dm + (size_t)keep[cand_s[j < L ? j : 0]] * ldD; }
        for (int r = tid; r < R; r += NT) {
            const double cr = cl[r];
            const int kr = keep[r];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < L) {
                    double v = drow[j][kr];
                    v *= v;
                    v = fmin(cr, v);
                    dm[(size_t)j * d.Rp + r] = v;
                    s[j] += v;
                }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < L) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_xor(s[j], o, 64);
                if (lane == 0) red[wave][j] = s[j];
            }
        }
        __syncthreads();
        int best = 0; double bp = 0.0;
        for (int j = 0; j < L; ++j) {
            double t = 0.0;
            for (int w = 0; w < NW; ++w) t += red[w][j];
            if (j == 0 || t < bp) { bp = t; best = j; }        // first minimum, np.argmin
        }
        pot = bp;
        if (tid == 0) center_ids[init * KM_CID + c] = cand_s[best];
        const double* db = dm + (size_t)best * d.Rp;
        for (int r = tid; r < R; r += NT) cl[r] = db[r];       // each thread re-reads what it wrote itself
    }
}

__device__ __forceinline__ double uniform_f64(double v)
{   // a value every lane holds identically, moved to scalar registers
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// The same seeding with the init's closest[] in REGISTERS (blocked layout: thread t owns rows [t*VPT, (t+1)*VPT); up to
// 1024*VPT kept rows).  Nothing is materialised: the cumulative sum is a block scan of the thread totals, and
// searchsorted(cum, v, 'left') is the NUMBER of rows with cum < v -- one packed block reduction for the L trial values.
// Three barriers per draw; the winning candidate's row of D is re-read (cache hit) instead of keeping L x VPT minima.
template <int NT, int VPT>
__global__ __launch_bounds__(NT) void pp_fused_reg_kernel(const double* __restrict__ Dm, int ldD,
                                                          const int* __restrict__ keep, KmDims d,
                                                          const int* __restrict__ c0, const double* __restrict__ u,
                                                          int ustride, int* __restrict__ center_ids)
{
    constexpr int NW = NT / 64;
    __shared__ double wtot[NW];
    __shared__ double red[NW][8];
    __shared__ unsigned long long cred[NW][2];
    const int init = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, R = d.R, L = d.L;
    const int base = tid * VPT;
    double cl[VPT];
    int kr[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) kr[v] = (base + v < R) ? keep[base + v] : -1;
    double pot;
    {
        const int first = c0[init];
        const double* drow = Dm + (size_t)keep[first] * ldD;
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            double x = 0.0;
            if (kr[v] >= 0) { x = drow[kr[v]]; x *= x; }
            cl[v] = x; s += x;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave][0] = s;
        __syncthreads();
        pot = 0.0;
        for (int w = 0; w < NW; ++w) pot += red[w][0];
        pot = uniform_f64(pot);
        if (tid == 0) center_ids[init * KM_CID] = first;
    }
    for (int c = 1; c < d.k; ++c) {
        // exclusive prefix of this thread's block of closest[]
        double tt = 0.0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) tt += cl[v];
        double x = tt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        double excl = 0.0;
        for (int w = 0; w < wave; ++w) excl += wtot[w];
        excl += x - tt;
        // searchsorted of the L trial values: rows with cum < v
        double tv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tv[j] = (j < L) ? u[(size_t)init * ustride + 1 + (size_t)(c - 1) * L + j] * pot : 0.0;
        unsigned long long ca = 0ull, cb = 0ull;
        double run = excl;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            run += cl[v];
            if (kr[v] >= 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ca += (run < tv[j]) ? (1ull << (16 * j)) : 0ull;
#pragma unroll
                for (int j = 4; j < 8; ++j) cb += (run < tv[j]) ? (1ull << (16 * (j - 4))) : 0ull;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ca += __shfl_xor(ca, o, 64); cb += __shfl_xor(cb, o, 64); }
        if (lane == 0) { cred[wave][0] = ca; cred[wave][1] = cb; }
        __syncthreads();
        ca = cb = 0ull;
        for (int w = 0; w < NW; ++w) { ca += cred[w][0]; cb += cred[w][1]; }
        int cand[8];
        const double* drow[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cnt = __builtin_amdgcn_readfirstlane((int)(((j < 4 ? ca : cb) >> (16 * (j & 3))) & 0xffffull));
            cand[j] = min(cnt, R - 1);
            drow[j] = Dm + (size_t)keep[j < L ? cand[j] : 0] * ldD;
        }
        double s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.0;
#pragma unroll
        for (int v = 0; v < VPT; ++v)
            if (kr[v] >= 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < L) { double t = drow[j][kr[v]]; t *= t; s[j] += fmin(cl[v], t); }
            }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < L) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_xor(s[j], o, 64);
                if (lane == 0) red[wave][j] = s[j];
            }
        __syncthreads();
        int best = 0; double bp = 0.0;
        for (int j = 0; j < L; ++j) {
            double t = 0.0;
            for (int w = 0; w < NW; ++w) t += red[w][j];
            if (j == 0 || t < bp) { bp = t; best = j; }
        }
        pot = uniform_f64(bp);
        best = __builtin_amdgcn_readfirstlane(best);
        int bc = cand[0]; const double* db = drow[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) if (j == best) { bc = cand[j]; db = drow[j]; }
        if (tid == 0) center_ids[init * KM_CID + c] = bc;
#pragma unroll
        for (int v = 0; v < VPT; ++v)
            if (kr[v] >= 0) { double t = db[kr[v]]; t *= t; cl[v] = fmin(cl[v], t); }
    }
}

// centres of every init from the chosen row ids: centers[init*k + c] = X[center_ids[init][c]]   grid (ld/256, k, n_init)
__global__ void pp_centers_kernel(const double* __restrict__ X, KmDims d, const int* __restrict__ center_ids,
                                  double* __restrict__ centers)
{
    const int init = blockIdx.z, c = blockIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < d.ld)
        centers[(size_t)(init * d.k + c) * d.ld + g] = (g < d.G) ? X[(size_t)center_ids[init * KM_CID + c] * d.ld + g] : 0.0;
}

// *tol_out = mean(var) * tol  (sklearn _kmeans.py:279-288)
__global__ __launch_bounds__(256) void km_tolerance_kernel(const double* __restrict__ var, int G, double tol,
                                                           double* __restrict__ tol_out)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int g = threadIdx.x; g < G; g += 256) s += var[g];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *tol_out = (s / G) * tol;
}

// ---------------------------------------------------------------- Lloyd
// csq[row] = |centers[row]|^2 for all n_init*k centre rows                             grid = n_init*k
__global__ __launch_bounds__(256) void center_norms_kernel(const double* __restrict__ centers, int ld,
                                                           int G, double* __restrict__ csq)
{
    __shared__ double red[4];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int g = threadIdx.x; g < G; g += 256) { const double v = centers[(size_t)j * ld + g]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) csq[j] = s;
}

// labels[r] = first argmin_j (csq[j] - 2 dots[j][r]); counts changed labels      grid (R/256, n_init)
// mode 0: Lloyd E step of the inits that are not done; mode 1: final E step of the inits flagged in `need`
__global__ __launch_bounds__(256) void assign_kernel(const double* __restrict__ dots, KmDims d,
                                                     const double* __restrict__ csq,
                                                     int* __restrict__ labels, KmState* st, int mode,
                                                     const int* __restrict__ need)
{
    const int init = blockIdx.y;
    if (mode == 0 ? st[init].done : !need[init]) return;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const double* dr = dots + (size_t)(init * d.k) * d.Rp;
    const double* cs = csq + init * d.k;
    int* lab = labels + (size_t)init * d.Rp;
    int ch = 0;
    if (r < d.R) {
        int best = 0; double bv = cs[0] - 2.0 * dr[r];
        for (int j = 1; j < d.k; ++j) {
            const double v = cs[j] - 2.0 * dr[(size_t)j * d.Rp + r];
            if (v < bv) { bv = v; best = j; }
        }
        ch = (lab[r] != best) ? 1 : 0;
        lab[r] = best;
    }
    if (mode == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ch += __shfl_xor(ch, o, 64);
        if ((threadIdx.x & 63) == 0 && ch) atomicAdd(&st[init].changed, ch);
    }
}

// partial[init][chunk][j][g] = sum over the chunk's rows with label j of X[r][g]   grid (G/256, nchunks, n_init)
__global__ __launch_bounds__(256) void accumulate_kernel(const double* __restrict__ X, KmDims d,
                                                         const int* __restrict__ labels,
                                                         int rows_per_chunk, int nchunks,
                                                         double* __restrict__ partial,
                                                         int* __restrict__ pcount, const KmState* __restrict__ st)
{
    extern __shared__ __attribute__((aligned(16))) double accs[];     // [k][blockDim.x]  (256 columns per block; 128 for k > 64)
    const int init = blockIdx.z;
    if (st[init].done) return;
    const int bw = (int)blockDim.x;
    const int g = blockIdx.x * bw + threadIdx.x, chunk = blockIdx.y, k = d.k;
    const int* lab = labels + (size_t)init * d.Rp;
    for (int j = 0; j < k; ++j) accs[j * bw + threadIdx.x] = 0.0;
    const int rb = chunk * rows_per_chunk, re = min(rb + rows_per_chunk, d.R);
    if (g < d.G)
        for (int r = rb; r < re; r += 16) {             // 16 rows' loads in flight, added in row order as before
            double x[16]; int l[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { const int rr = min(r + q, re - 1); x[q] = X[(size_t)rr * d.ld + g]; l[q] = lab[rr]; }
#pragma unroll
            for (int q = 0; q < 16; ++q) if (r + q < re) accs[l[q] * bw + threadIdx.x] += x[q];
        }
    double* pp = partial + (((size_t)init * nchunks + chunk) * k) * d.ld;
    if (g < d.G)
        for (int j = 0; j < k; ++j) pp[(size_t)j * d.ld + g] = accs[j * bw + threadIdx.x];
    if (blockIdx.x == 0 && (int)threadIdx.x < k) {
        int c = 0;
        for (int r = rb; r < re; ++r) c += (lab[r] == (int)threadIdx.x) ? 1 : 0;
        pcount[(init * nchunks + chunk) * k + threadIdx.x] = c;
    }
}

// sums[init][j][g] = sum_chunks partial; counts[init][j] = sum_chunks pcount     grid (G/256, k, n_init)
__global__ void reduce_partial_kernel(const double* __restrict__ partial, const int* __restrict__ pcount,
                                      int nchunks, KmDims d, double* __restrict__ sums,
                                      int* __restrict__ counts, KmState* st)
{
    const int init = blockIdx.z;
    if (st[init].done) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, k = d.k;
    if (g < d.G) {
        double s = 0.0;
        for (int c = 0; c < nchunks; ++c) s += partial[(((size_t)init * nchunks + c) * k + j) * d.ld + g];
        sums[((size_t)init * k + j) * d.ld + g] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int c = 0;
        for (int q = 0; q < nchunks; ++q) c += pcount[(init * nchunks + q) * k + j];
        counts[init * k + j] = c;
        if (c == 0) atomicAdd(&st[init].n_empty, 1);
    }
}

// squared distance of every row to its assigned centre, for the inits that have an empty cluster to relocate (rare: the
// launch is 16 rows per workgroup so that the usual "nothing to do" exit is cheap)         grid (R/16, n_init)
constexpr int RCD_ROWS = 16;
__global__ __launch_bounds__(256) void row_center_dist_kernel(const double* __restrict__ X, KmDims d,
                                                              const double* __restrict__ centers,
                                                              const int* __restrict__ labels,
                                                              double* __restrict__ dist,
                                                              const KmState* __restrict__ st)
{
    __shared__ double red[4];
    const int init = blockIdx.y;
    if (st[init].done || st[init].n_empty == 0) return;
    for (int r = blockIdx.x * RCD_ROWS; r < min(d.R, (int)(blockIdx.x + 1) * RCD_ROWS); ++r) {
        const double* c = centers + (size_t)(init * d.k + labels[(size_t)init * d.Rp + r]) * d.ld;
        double s = 0.0;
        for (int g = threadIdx.x; g < d.G; g += 256) { const double v = X[(size_t)r * d.ld + g] - c[g]; s += v * v; }
        s = block_sum(s, red);
        if (threadIdx.x == 0) dist[(size_t)init * d.Rp + r] = s;
    }
}

// The final pass (inertia of every init): one workgroup per row walks all inits, so the row of X is fetched from HBM
// once and the n_init x k centre rows come out of L2.                                          grid = R
__global__ __launch_bounds__(256) void row_center_dist_all_kernel(const double* __restrict__ X, KmDims d,
                                                                  const double* __restrict__ centers,
                                                                  const int* __restrict__ labels,
                                                                  double* __restrict__ dist, int n_init)
{
    __shared__ double red[4];
    const int r = blockIdx.x;
    const double* x = X + (size_t)r * d.ld;
    for (int init = 0; init < n_init; ++init) {
        const double* c = centers + (size_t)(init * d.k + labels[(size_t)init * d.Rp + r]) * d.ld;
        double s = 0.0;
        for (int g = threadIdx.x; g < d.G; g += 256) { const double v = x[g] - c[g]; s += v * v; }
        s = block_sum(s, red);
        if (threadIdx.x == 0) dist[(size_t)init * d.Rp + r] = s;
        __syncthreads();
    }
}

// relocate empty clusters to the farthest points (sklearn _k_means_common.pyx:167-211): clusters in
// ascending id, points in descending distance.                                         grid = n_init
__global__ __launch_bounds__(256) void relocate_empty_kernel(const double* __restrict__ X, KmDims d,
                                                             const int* __restrict__ labels,
                                                             double* __restrict__ dist,
                                                             double* __restrict__ sums, int* __restrict__ counts,
                                                             const KmState* __restrict__ st)
{
    const int init = blockIdx.x;
    if (st[init].done || st[init].n_empty == 0) return;
    __shared__ double bv[256];
    __shared__ int bi[256];
    const int tid = threadIdx.x, k = d.k;
    const int* lab = labels + (size_t)init * d.Rp;
    double* di = dist + (size_t)init * d.Rp;
    double* sm = sums + (size_t)init * k * d.ld;
    int* cn = counts + init * k;
    for (int e = 0; e < k; ++e) {
        if (cn[e] != 0) continue;               // uniform
        double v = -1.0; int idx = -1;
        for (int r = tid; r < d.R; r += 256) if (di[r] > v) { v = di[r]; idx = r; }
        bv[tid] = v; bi[tid] = idx;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o && (bv[tid + o] > bv[tid] || (bv[tid + o] == bv[tid] && bi[tid + o] < bi[tid] && bi[tid + o] >= 0))) {
                bv[tid] = bv[tid + o]; bi[tid] = bi[tid + o];
            }
            __syncthreads();
        }
        const int far = bi[0];
        const double fv = bv[0];
        __syncthreads();
        if (far < 0 || fv <= 0.0) return;           // all points coincide with their centres
        const int old = lab[far];
        for (int g = tid; g < d.G; g += 256) {
            const double x = X[(size_t)far * d.ld + g];
            sm[(size_t)old * d.ld + g] -= x;
            sm[(size_t)e * d.ld + g] = x;
        }
        __syncthreads();
        if (tid == 0) { cn[e] = 1; cn[old] -= 1; di[far] = -1.0; }
        __syncthreads();
    }
}

// new centre j of init i = sums * (1/count) (empty -> copy of the heaviest cluster); shift2[i][j] = |new - old|^2.
// A finished init just carries its centres over so both buffers stay valid.             grid (k, n_init)
__global__ __launch_bounds__(256) void finish_centers_kernel(const double* __restrict__ sums,
                                                             const int* __restrict__ counts, KmDims d,
                                                             const double* __restrict__ old_c,
                                                             double* __restrict__ new_c, const KmState* __restrict__ st,
                                                             double* __restrict__ shift2)
{
    __shared__ double red[4];
    const int j = blockIdx.x, init = blockIdx.y, k = d.k;
    const double* oc = old_c + ((size_t)init * k + j) * d.ld;
    double* nc = new_c + ((size_t)init * k + j) * d.ld;
    if (st[init].done) {
        for (int e = threadIdx.x; e < d.ld; e += 256) nc[e] = oc[e];
        return;
    }
    const int* cn = counts + init * k;
    int src = j;
    if (cn[j] <= 0) { src = 0; for (int q = 1; q < k; ++q) if (cn[q] > cn[src]) src = q; }
    const double* sm = sums + ((size_t)init * k + src) * d.ld;
    const double alpha = 1.0 / (double)cn[src];
    double s = 0.0;
    for (int g = threadIdx.x; g < d.ld; g += 256) {
        double v = 0.0;
        if (g < d.G) {
            v = sm[g] * alpha;
            const double df = v - oc[g];
            s += df * df;
        }
        nc[g] = v;
    }
    s = block_sum(s, red);
    const double sh = sqrt(s);
    if (threadIdx.x == 0) shift2[init * KM_CID + j] = sh * sh;
}

// The stopping rule of one Lloyd iteration, per init (sklearn _kmeans.py:700-730): no label changed -> strict
// convergence; else total centre shift <= tol -> converged.  Also clears the per-iteration counters.   one thread per init
__global__ void lloyd_decide_kernel(KmState* st, const double* __restrict__ shift2, int k, int n_init,
                                    const double* __restrict__ tol, int it)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_init) return;
    if (!st[i].done) {
        double tot = 0.0;
        for (int j = 0; j < k; ++j) tot += shift2[i * KM_CID + j];
        st[i].shift_tot = tot;
        st[i].iters = it + 1;
        if (st[i].changed == 0) { st[i].strict = 1; st[i].done = 1; }
        else if (tot <= *tol) st[i].done = 1;
    }
    st[i].changed = 0;
    st[i].n_empty = 0;
}

// inertia[init] = sum_r dist[init][r]                                                   grid = n_init
__global__ __launch_bounds__(256) void inertia_kernel(const double* __restrict__ dist, KmDims d, KmState* st)
{
    __shared__ double red[4];
    const int init = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < d.R; r += 256) s += dist[(size_t)init * d.Rp + r];
    s = block_sum(s, red);
    if (threadIdx.x == 0) st[init].inertia = s;
}

__global__ __launch_bounds__(256) void sum_kernel(const double* __restrict__ v, int n, double* out)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s;
}

// ---------------------------------------------------------------- per-cluster per-gene median
// order[]: row ids sorted by cluster, seg[j]..seg[j+1] = cluster j.  One thread per (cluster, gene):
// exact order statistics by bisection on the bit pattern of (non-negative) doubles.
__device__ __forceinline__ unsigned long long f64_key(double x)
{   // order-preserving map double -> uint64 (handles negative values and -0.0 < +0.0 as ints; the
    // spectra are non-negative, this just makes the selection total)
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unkey(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ double kth_smallest(const double* __restrict__ X, int ld, int g,
                                               const int* __restrict__ rows, int m, int kth /*1-based*/)
{
    unsigned long long lo = 0ull, hi = ~0ull;
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int q = 0; q < m; ++q) c += (f64_key(X[(size_t)rows[q] * ld + g]) <= mid) ? 1 : 0;
        if (c >= kth) hi = mid; else lo = mid + 1;
    }
    return f64_unkey(lo);
}

// One WAVE per (cluster, gene): the lanes split the cluster's m rows, keep their values in
// registers (VPL per lane) and count with a wave reduction -- 64 bisection steps on the key.
template <int VPL>
__device__ __forceinline__ double wave_kth(const unsigned long long (&key)[VPL], int kth)
{
    unsigned long long lo = 0ull, hi = ~0ull;
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int v = 0; v < VPL; ++v) c += (key[v] <= mid) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (c >= kth) hi = mid; else lo = mid + 1;
    }
    return f64_unkey(lo);
}

template <int VPL>
__global__ __launch_bounds__(256) void cluster_median_kernel(const double* __restrict__ X, int ld, int G,
                                                             const int* __restrict__ order,
                                                             const int* __restrict__ seg,
                                                             double* __restrict__ med)
{
    const int j = blockIdx.y, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int b = seg[j], m = seg[j + 1] - b;
    unsigned long long key[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int q = lane + 64 * v;
        key[v] = (q < m) ? f64_key(X[(size_t)order[b + q] * ld + g]) : ~0ull;   // padding sorts last
    }
    double val;
    if (m & 1) val = wave_kth<VPL>(key, m / 2 + 1);
    else val = (wave_kth<VPL>(key, m / 2) + wave_kth<VPL>(key, m / 2 + 1)) / 2.0;
    if (lane == 0) med[(size_t)j * G + g] = val;
}

// fallback for clusters with more than 2048 members: one thread per (cluster, gene), values re-read
__global__ void cluster_median_big_kernel(const double* __restrict__ X, int ld, int G,
                                          const int* __restrict__ order, const int* __restrict__ seg,
                                          double* __restrict__ med)
{
    const int j = blockIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int b = seg[j], m = seg[j + 1] - b;
    const int* rows = order + b;
    double v;
    if (m & 1) v = kth_smallest(X, ld, g, rows, m, m / 2 + 1);
    else v = (kth_smallest(X, ld, g, rows, m, m / 2) + kth_smallest(X, ld, g, rows, m, m / 2 + 1)) / 2.0;
    med[(size_t)j * G + g] = v;
}

__global__ __launch_bounds__(256) void normalise_rows_sum_kernel(double* __restrict__ M, int G)
{
    __shared__ double red[4];
    const int r = blockIdx.x;
    double s = 0.0;
    for (int g = threadIdx.x; g < G; g += 256) s += M[(size_t)r * G + g];
    s = block_sum(s, red);
    for (int g = threadIdx.x; g < G; g += 256) M[(size_t)r * G + g] /= s;
}

// ---------------------------------------------------------------- silhouette
// rows in `order` (sorted by cluster): sample q = order position.  One workgroup per sample.
__global__ __launch_bounds__(256) void silhouette_kernel(const double* __restrict__ Dm, int ld,
                                                         const int* __restrict__ rowid,   // kept -> original row
                                                         const int* __restrict__ order, const int* __restrict__ seg,
                                                         const int* __restrict__ labels, int Rk, int k,
                                                         double* __restrict__ sil)
{
    __shared__ double red[4];
    __shared__ double csum[KM_CID];
    const int q = blockIdx.x;
    const int i = rowid[q];
    for (int j = 0; j < k; ++j) {
        double s = 0.0;
        for (int p = seg[j] + threadIdx.x; p < seg[j + 1]; p += 256) s += Dm[(size_t)i * ld + rowid[order[p]]];
        s = block_sum(s, red);
        if (threadIdx.x == 0) csum[j] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int li = labels[q];
        const int ni = seg[li + 1] - seg[li];
        double a = csum[li] / (double)(ni - 1);
        double b = 1.0 / 0.0;
        for (int j = 0; j < k; ++j) {
            const int nj = seg[j + 1] - seg[j];
            if (j == li || nj == 0) continue;
            b = fmin(b, csum[j] / (double)nj);
        }
        double s = (b - a) / fmax(a, b);
        if (ni <= 1 || s != s) s = 0.0;
        sil[q] = s;
    }
}

// ---------------------------------------------------------------- prediction error
// err = sum_{i,g} (X[i][g] - sum_c W[i][c] H[c][g])^2 ; X float32 padded [N_pad][ldx]; W [N][k], H [k][G] f64
__global__ __launch_bounds__(256) void residual_sq_kernel(const float* __restrict__ X, int ldx, int N, int G,
                                                          const double* __restrict__ W, const double* __restrict__ H,
                                                          int k, int rows_per_block, double* __restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) double Hs[];       // [k][blockDim.x]  (256 genes per block; 128 for k > 64)
    __shared__ double red[4];
    const int bw = (int)blockDim.x;
    const int g = blockIdx.x * bw + threadIdx.x;
    for (int c = 0; c < k; ++c) Hs[c * bw + threadIdx.x] = (g < G) ? H[(size_t)c * G + g] : 0.0;
    const int rb = blockIdx.y * rows_per_block, re = min(rb + rows_per_block, N);
    double s = 0.0;
    if (g < G)
        for (int i = rb; i < re; ++i) {
            double p = 0.0;
            for (int c = 0; c < k; ++c) p += W[(size_t)i * k + c] * Hs[c * bw + threadIdx.x];
            const double d = (double)X[(size_t)i * ldx + g] - p;
            s += d * d;
        }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

}  // namespace cnmf
