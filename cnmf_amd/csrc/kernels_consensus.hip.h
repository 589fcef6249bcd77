// Consensus-step kernels (gfx950), all arithmetic in float64.
//
// The consensus core of the reference (cnmf.py:871-936) is float64 end to end and its
// outputs are index-like (density filter, k-means labels, per-cluster medians), so the
// device path keeps float64: the stacked spectra are tiny (R <= ~5000 rows x 2000 genes)
// and the f64 matrix pipe (v_mfma_f64_16x16x4_f64, 78 TF) makes the R x R Gram matrix a
// ~1 ms kernel.  Restated functions:
//   row_norm / l2        cnmf.py:882
//   dgemm_nt + dist_epi  sklearn/metrics/pairwise.py:419-438  (-2 X.Xt + |x|^2 + |y|^2, clamp, diag=0, sqrt)
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

// ---------------------------------------------------------------- f64 MFMA GEMM, C = A . B^T
// A [M][lda], B [N][ldb] both K-contiguous, C [M][ldc].  Workgroup tile 64 x 64, 4 waves
// (2 x 2), each wave 32 x 32 = 2 x 2 MFMA tiles of 16 x 16 x 4.  M, N multiples of 64 and
// K multiple of 16 (callers zero-pad).
//   A operand lane l: A[i=l&15][k=l>>4]   B operand lane l: B[k=l>>4][j=l&15]
//   D reg r lane l  : row = (l>>4) + 4*r, col = l&15          (f64 layout differs from f32!)
constexpr int DBK = 16;
constexpr int DLD = DBK + 2;     // padded LDS row (doubles): 18*8 B = 144 B -> ds_read_b64 conflict-free

__global__ __launch_bounds__(256) void dgemm_nt_kernel(const double* __restrict__ A, int lda,
                                                       const double* __restrict__ B, int ldb,
                                                       double* __restrict__ C, int ldc, int K)
{
    __shared__ __attribute__((aligned(16))) double As[2][64 * DLD];
    __shared__ __attribute__((aligned(16))) double Bs[2][64 * DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    // staging: 64 rows x 16 doubles = 512 v2d per operand -> 2 per thread
    const int s_row = tid >> 3, s_k = (tid & 7) * 2;
    const double* a_src = A + (size_t)(m0 + s_row) * lda + s_k;
    const double* b_src = B + (size_t)(n0 + s_row) * ldb + s_k;
    v2d ar[2], br[2];
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int nk = K / DBK;
#define DG_LOAD(kt_)                                                                          \
    {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                       \
            ar[i] = *reinterpret_cast<const v2d*>(a_src + (size_t)(32 * i) * lda + (kt_) * DBK); \
            br[i] = *reinterpret_cast<const v2d*>(b_src + (size_t)(32 * i) * ldb + (kt_) * DBK); \
        }                                                                                     \
    }
#define DG_STORE(buf_)                                                                        \
    {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                       \
            *reinterpret_cast<v2d*>(&As[buf_][(s_row + 32 * i) * DLD + s_k]) = ar[i];         \
            *reinterpret_cast<v2d*>(&Bs[buf_][(s_row + 32 * i) * DLD + s_k]) = br[i];         \
        }                                                                                     \
    }
    if (nk > 0) { DG_LOAD(0) DG_STORE(0) }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) DG_LOAD(kt + 1)
        const double* as = &As[buf][(wm * 32 + li) * DLD + lk];
        const double* bs = &Bs[buf][(wn * 32 + li) * DLD + lk];
#pragma unroll
        for (int q = 0; q < DBK / 4; ++q) {
            const double a0 = as[q * 4], a1 = as[16 * DLD + q * 4];
            const double b0 = bs[q * 4], b1 = bs[16 * DLD + q * 4];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) DG_STORE(buf ^ 1)
        __syncthreads();
    }
#undef DG_LOAD
#undef DG_STORE
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + lk + 4 * r;
                const int col = n0 + wn * 32 + j * 16 + li;
                C[(size_t)row * ldc + col] = acc[i][j][r];
            }
}

// Few-rows variant for the k-means products (M = 64 centre / candidate rows against all N rows):
// workgroup tile 64 x 16 so that the launch has N/16 workgroups instead of N/64.
__global__ __launch_bounds__(256) void dgemm_nt_small_kernel(const double* __restrict__ A, int lda,
                                                             const double* __restrict__ B, int ldb,
                                                             double* __restrict__ C, int ldc, int K)
{
    __shared__ __attribute__((aligned(16))) double As[2][64 * DLD];
    __shared__ __attribute__((aligned(16))) double Bs[2][16 * DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 16;
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

// D[i][j] = sqrt(max(0, sq_i + sq_j - 2 G_ij)), diagonal forced to 0 (in place on the Gram)
__global__ void dist_epilogue_kernel(double* __restrict__ Dm, int ld, int R, const double* __restrict__ sq)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (i >= R || j >= R) return;
    double d = -2.0 * Dm[(size_t)i * ld + j];
    d += sq[i];
    d += sq[j];
    d = fmax(d, 0.0);
    if (i == j) d = 0.0;
    Dm[(size_t)i * ld + j] = sqrt(d);
}

// density[i] = (sum of the m smallest entries of row i) / n      (m = n+1, self distance 0 included)
// Exact selection by bisection on the IEEE bit pattern (non-negative doubles order like uint64).
// LDS = true: the row is staged in LDS (R <= 19 200); false: the ~64 selection passes re-read the row from
// global memory (it stays in L2) -- any R.
template <bool LDS>
__global__ __launch_bounds__(256) void knn_density_kernel(const double* __restrict__ Dm, int ld, int R,
                                                          int m, int n, double* __restrict__ density)
{
    extern __shared__ __attribute__((aligned(16))) double rowbuf_lds[];
    __shared__ double red[4];
    __shared__ int cnt_s[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const double* __restrict__ grow = Dm + (size_t)i * ld;
    if (LDS) {
        for (int j = tid; j < R; j += 256) rowbuf_lds[j] = grow[j];
        __syncthreads();
    }
#define rowbuf (LDS ? (const double*)rowbuf_lds : grow)
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
#undef rowbuf
}

// gather rows: out[q][:] = in[idx[q]][:]   (zero-padded destination rows are cleared by the caller)
__global__ void gather_rows_kernel(const double* __restrict__ in, int ld_in, const int* __restrict__ idx,
                                   int nrows, int G, double* __restrict__ out, int ld_out)
{
    const int q = blockIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nrows && g < G) out[(size_t)q * ld_out + g] = in[(size_t)idx[q] * ld_in + g];
}

// column statistics of X [R][ld]: mean[g], var[g] (population).  Block = 64 columns x 4 row groups;
// fixed summation order (row group partials added in group order).
__global__ __launch_bounds__(256) void col_stats_kernel(const double* __restrict__ X, int ld, int R, int G,
                                                        double* __restrict__ mean, double* __restrict__ var)
{
    __shared__ double part[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + c;
    const int per = (R + 3) / 4, rb = rg * per, re = min(R, rb + per);
    double s = 0.0;
    if (g < G) for (int r = rb; r < re; ++r) s += X[(size_t)r * ld + g];
    part[rg][c] = s;
    __syncthreads();
    const double mu = (part[0][c] + part[1][c] + part[2][c] + part[3][c]) / R;
    __syncthreads();
    double v = 0.0;
    if (g < G) for (int r = rb; r < re; ++r) { const double d = X[(size_t)r * ld + g] - mu; v += d * d; }
    part[rg][c] = v;
    __syncthreads();
    if (rg == 0 && g < G) {
        mean[g] = mu;
        var[g] = (part[0][c] + part[1][c] + part[2][c] + part[3][c]) / R;
    }
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
    int done;             // Lloyd loop finished for this init (host sets it)
};

struct KmDims { int R, Rp, G, ld, k, L; };   // kept rows, padded rows, genes, padded genes, clusters, local trials

// first centre: centers[init*k] = X[c0[init]]
__global__ void pp_seed_kernel(const double* __restrict__ X, KmDims d, const int* __restrict__ c0,
                               double* __restrict__ centers, int* __restrict__ center_ids)
{
    const int init = blockIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < d.ld) centers[(size_t)(init * d.k) * d.ld + g] = (g < d.G) ? X[(size_t)c0[init] * d.ld + g] : 0.0;
    if (g == 0) center_ids[init * KM_CID] = c0[init];
}

// closest[r] = max(0, sq_c + sq_r - 2 dot[centre 0][r]); pot = sum closest          grid = n_init
__global__ __launch_bounds__(256) void pp_first_kernel(const double* __restrict__ dots, KmDims d,
                                                       const double* __restrict__ sq, const int* __restrict__ c0,
                                                       double* __restrict__ closest, KmState* st)
{
    __shared__ double red[4];
    const int init = blockIdx.x;
    const double* drow = dots + (size_t)(init * d.k) * d.Rp;
    double* cl = closest + (size_t)init * d.Rp;
    const double sc = sq[c0[init]];
    double s = 0.0;
    for (int r = threadIdx.x; r < d.R; r += 256) {
        double v = -2.0 * drow[r];
        v += sc; v += sq[r];
        v = fmax(v, 0.0);
        cl[r] = v;
        s += v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) st[init].pot = s;
}

// cumulative sum of closest[] (chunked serial scan, like np.cumsum up to rounding) + searchsorted of
// the L random values u*pot (side='left'), clipped to R-1.                             grid = n_init
__global__ __launch_bounds__(256) void pp_candidates_kernel(const double* __restrict__ closest, KmDims d,
                                                            const double* __restrict__ u, int ustride, int uoff,
                                                            double* __restrict__ cum, KmState* st)
{
    __shared__ double part[256];
    const int init = blockIdx.x, tid = threadIdx.x, R = d.R;
    const double* cl = closest + (size_t)init * d.Rp;
    double* cm = cum + (size_t)init * d.Rp;
    const int per = (R + 255) / 256;
    const int b = tid * per, e = min(b + per, R);
    double s = 0.0;
    for (int r = b; r < e; ++r) s += cl[r];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { double run = 0.0; for (int t = 0; t < 256; ++t) { const double v = part[t]; part[t] = run; run += v; } }
    __syncthreads();
    double run = part[tid];
    for (int r = b; r < e; ++r) { run += cl[r]; cm[r] = run; }
    __syncthreads();
    if (tid < d.L) {
        const double v = u[(size_t)init * ustride + uoff + tid] * st[init].pot;
        int lo = 0, hi = R;                       // first index with cum[idx] >= v
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cm[mid] < v) lo = mid + 1; else hi = mid; }
        st[init].cand[tid] = min(lo, R - 1);
    }
}

// candidate rows of every init into one tall matrix: cand[init*L + j] = X[st[init].cand[j]]   grid (G/256, L, n_init)
__global__ void pp_gather_kernel(const double* __restrict__ X, KmDims d, const KmState* __restrict__ st,
                                 double* __restrict__ cand)
{
    const int init = blockIdx.z, j = blockIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < d.G) cand[(size_t)(init * d.L + j) * d.ld + g] = X[(size_t)st[init].cand[j] * d.ld + g];
}

// dmin[init][j][r] = min(closest[r], max(0, sq_cand + sq_r - 2 dots[init*L+j][r])); cpot = sum   grid (L, n_init)
__global__ __launch_bounds__(256) void pp_update_kernel(const double* __restrict__ dots, KmDims d,
                                                        const double* __restrict__ sq,
                                                        const double* __restrict__ closest,
                                                        const KmState* __restrict__ st, double* __restrict__ dmin,
                                                        double* __restrict__ cpot)
{
    __shared__ double red[4];
    const int j = blockIdx.x, init = blockIdx.y;
    const int c = st[init].cand[j];
    const double* drow = dots + (size_t)(init * d.L + j) * d.Rp;
    const double* cl = closest + (size_t)init * d.Rp;
    double* dm = dmin + ((size_t)init * 8 + j) * d.Rp;
    double s = 0.0;
    for (int r = threadIdx.x; r < d.R; r += 256) {
        double v = -2.0 * drow[r];
        v += sq[c]; v += sq[r];
        v = fmax(v, 0.0);
        v = fmin(cl[r], v);
        dm[r] = v;
        s += v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) cpot[init * 8 + j] = s;
}

// choose the candidate with the smallest potential (first min), make it centre `c`      grid = n_init
__global__ __launch_bounds__(256) void pp_pick_kernel(const double* __restrict__ cpot, KmDims d,
                                                      const double* __restrict__ dmin,
                                                      double* __restrict__ closest, KmState* st,
                                                      const double* __restrict__ X,
                                                      double* __restrict__ centers, int c,
                                                      int* __restrict__ center_ids)
{
    __shared__ int best_s;
    const int init = blockIdx.x;
    if (threadIdx.x == 0) {
        int best = 0;
        for (int j = 1; j < d.L; ++j) if (cpot[init * 8 + j] < cpot[init * 8 + best]) best = j;
        best_s = best;
        st[init].pot = cpot[init * 8 + best];
        st[init].best = st[init].cand[best];
        center_ids[init * KM_CID + c] = st[init].cand[best];
    }
    __syncthreads();
    const int best = best_s, row = st[init].cand[best];
    const double* dm = dmin + ((size_t)init * 8 + best) * d.Rp;
    double* cl = closest + (size_t)init * d.Rp;
    for (int r = threadIdx.x; r < d.R; r += 256) cl[r] = dm[r];
    double* cdst = centers + (size_t)(init * d.k + c) * d.ld;
    for (int g = threadIdx.x; g < d.ld; g += 256) cdst[g] = (g < d.G) ? X[(size_t)row * d.ld + g] : 0.0;
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

__global__ void km_reset_kernel(KmState* st, int n_init)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_init) { st[i].changed = 0; st[i].n_empty = 0; }
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
        for (int r = rb; r < re; ++r) accs[lab[r] * bw + threadIdx.x] += X[(size_t)r * d.ld + g];
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

// squared distance of every row to its assigned centre                          grid (R, n_init)
// mode 0: only for inits with an empty cluster (relocation); mode 1: all inits (inertia)
__global__ __launch_bounds__(256) void row_center_dist_kernel(const double* __restrict__ X, KmDims d,
                                                              const double* __restrict__ centers,
                                                              const int* __restrict__ labels,
                                                              double* __restrict__ dist,
                                                              const KmState* __restrict__ st, int mode)
{
    __shared__ double red[4];
    const int init = blockIdx.y;
    if (mode == 0 && (st[init].done || st[init].n_empty == 0)) return;
    const int r = blockIdx.x;
    const double* c = centers + (size_t)(init * d.k + labels[(size_t)init * d.Rp + r]) * d.ld;
    double s = 0.0;
    for (int g = threadIdx.x; g < d.G; g += 256) { const double v = X[(size_t)r * d.ld + g] - c[g]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) dist[(size_t)init * d.Rp + r] = s;
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

// new centres = sums * (1/count) (empty -> copy of the heaviest cluster); shift_tot = sum |new-old|^2.
// A finished init just carries its centres over so both buffers stay valid.             grid = n_init
__global__ __launch_bounds__(256) void finish_centers_kernel(const double* __restrict__ sums,
                                                             const int* __restrict__ counts, KmDims d,
                                                             const double* __restrict__ old_c,
                                                             double* __restrict__ new_c, KmState* st)
{
    __shared__ double red[4];
    __shared__ int amax_s;
    const int init = blockIdx.x, k = d.k;
    const double* oc = old_c + (size_t)init * k * d.ld;
    double* nc = new_c + (size_t)init * k * d.ld;
    if (st[init].done) {
        for (int e = threadIdx.x; e < k * d.ld; e += 256) nc[e] = oc[e];
        return;
    }
    const double* sm = sums + (size_t)init * k * d.ld;
    const int* cn = counts + init * k;
    if (threadIdx.x == 0) {
        int am = 0;
        for (int j = 1; j < k; ++j) if (cn[j] > cn[am]) am = j;
        amax_s = am;
    }
    __syncthreads();
    const int am = amax_s;
    double tot = 0.0;
    for (int j = 0; j < k; ++j) {
        const int src = (cn[j] > 0) ? j : am;
        const double alpha = 1.0 / (double)cn[src];
        double s = 0.0;
        for (int g = threadIdx.x; g < d.ld; g += 256) {
            double v = 0.0;
            if (g < d.G) {
                v = sm[(size_t)src * d.ld + g] * alpha;
                const double df = v - oc[(size_t)j * d.ld + g];
                s += df * df;
            }
            nc[(size_t)j * d.ld + g] = v;
        }
        s = block_sum(s, red);
        const double sh = sqrt(s);
        tot += sh * sh;
    }
    if (threadIdx.x == 0) st[init].shift_tot = tot;
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
