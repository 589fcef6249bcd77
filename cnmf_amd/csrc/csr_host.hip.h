// The resident matrix as compressed rows -- of X (cells x genes: "CSR") and of X^T (genes x cells: "CSC") -- for the paths
// that only touch the stored entries: the float64 Kullback-Leibler refits of the consensus tail (mu_refit_host.hip.h) and
// the sliced-ELL images of the Kullback-Leibler restarts (mu_host.hip.h).  scikit-learn walks exactly these arrays when
// the reference hands it a scipy.sparse matrix (sklearn/decomposition/_nmf.py:192 `_special_sparse_dot`; cnmf.py:726, 873,
// 950: `norm_counts.X` / `tpm.X` as stored).
//
//   * cnmf_set_matrix_csr KEEPS the uploaded arrays on the device (round 5; they used to be dropped after the densify);
//   * a matrix that arrived dense gets its compressed rows from the resident float32 matrix on first use (two passes);
//   * X^T's compressed rows are built ON THE DEVICE from those of X -- no transposed upload: a counting sort whose order
//     is fixed by construction (row chunks in order, the rows of a chunk one after the other, the entries of a row -- distinct
//     columns -- side by side), so every column lists its entries by ascending row, run to run identical.
// Included by cnmf_hip.hip (after runtime.hip.h).
#pragma once

namespace cnmf {

// ---- dense -> compressed rows.  Pass 1: stored (non-zero) entries per row
__global__ __launch_bounds__(256) void csr_count_dense_kernel(const float* __restrict__ M, int ld, int R, int C,
                                                              long long* __restrict__ cnt)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* m = M + (size_t)row * ld;
    int n = 0;
    for (int c = lane; c < C; c += 64) n += (m[c] != 0.f) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0) cnt[row] = n;
}

// Pass 2: a wavefront per row compacts the row in column order
__global__ __launch_bounds__(256) void csr_fill_dense_kernel(const float* __restrict__ M, int ld, int R, int C,
                                                             const long long* __restrict__ ptr, int* __restrict__ idx,
                                                             float* __restrict__ val)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* m = M + (size_t)row * ld;
    long long base = ptr[row];
    for (int cb = 0; cb < C; cb += 64) {
        const int c = cb + lane;
        const float x = c < C ? m[c] : 0.f;
        const bool nz = x != 0.f;
        const unsigned long long mask = __ballot(nz);
        if (nz) {
            const long long p = base + __popcll(mask & ((1ull << lane) - 1ull));
            idx[p] = c; val[p] = x;
        }
        base += __popcll(mask);
    }
}

__global__ __launch_bounds__(256) void csr_widen_ptr_kernel(const int* __restrict__ p32, long long n, long long* __restrict__ p64)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p64[i] = p32[i];
}

// ---- transpose.  Pass 1: entries per (row chunk, column)
__global__ __launch_bounds__(256) void csr_tr_hist_kernel(const long long* __restrict__ ptr, const int* __restrict__ idx,
                                                          int R, int C, int rows_per_chunk, int* __restrict__ cnt)
{
    const int t = blockIdx.x;
    const int r0 = t * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    if (r0 >= r1) return;
    int* c = cnt + (size_t)t * C;
    for (long long p = ptr[r0] + threadIdx.x; p < ptr[r1]; p += 256) atomicAdd(&c[idx[p]], 1);
}

// Pass 2a: entries per column
__global__ __launch_bounds__(256) void csr_tr_total_kernel(const int* __restrict__ cnt, int T, int C, long long* __restrict__ total)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    long long s = 0;
    for (int t = 0; t < T; ++t) s += cnt[(size_t)t * C + c];
    total[c] = s;
}

// Pass 2b: cnt[t][c] := first position of chunk t inside column c (64-bit positions kept relative to the column start)
__global__ __launch_bounds__(256) void csr_tr_offsets_kernel(int* __restrict__ cnt, int T, int C)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    int run = 0;
    for (int t = 0; t < T; ++t) { const int v = cnt[(size_t)t * C + c]; cnt[(size_t)t * C + c] = run; run += v; }
}

// Pass 3: ONE wavefront per chunk walks its rows in order; the entries of a row name distinct columns, so the 64 lanes
// never meet on a counter, and the counters of a chunk belong to this wavefront alone (the atomic is only there to go
// through L2: a plain load could see the line as it was before the previous row's store)
__global__ __launch_bounds__(64) void csr_tr_fill_kernel(const long long* __restrict__ ptr, const int* __restrict__ idx,
                                                         const float* __restrict__ val, int R, int C, int rows_per_chunk,
                                                         int* __restrict__ cnt, const long long* __restrict__ tptr,
                                                         int* __restrict__ tidx, float* __restrict__ tval)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const int r0 = t * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    int* c = cnt + (size_t)t * C;
    for (int r = r0; r < r1; ++r) {
        const long long b = ptr[r], e = ptr[r + 1];
        for (long long p = b + lane; p < e; p += 64) {
            const int col = idx[p];
            const long long q = tptr[col] + atomicAdd(&c[col], 1);
            tidx[q] = r; tval[q] = val[p];
        }
        __builtin_amdgcn_s_waitcnt(0);            // the counters of this row are back before the next row asks for them
    }
}


// ---- the float64 statistics and products of the consensus tail on the compressed rows of X^T (round 5): a wavefront per
// gene, its stored entries lane-strided, fixed-order butterfly sums -- so that a Kullback-Leibler run's tail (whose refits
// walk the stored entries already) never needs the dense N x G_all image of the TPM matrix.
// mean[g] = sum x / N ;  ssd[g] = sum over ALL cells of (x - mean)^2 = sum_stored (x - mean)^2 + (N - stored) mean^2
__global__ __launch_bounds__(256) void csc_col_moments_kernel(const long long* __restrict__ tptr, const float* __restrict__ tval,
                                                              int G, int N, double* __restrict__ mean, double* __restrict__ ssd)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    const long long b = tptr[g], e = tptr[g + 1];
    double s = 0.0;
    for (long long p = b + lane; p < e; p += 64) s += (double)tval[p];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const double mu = s / (double)N;
    double q = 0.0;
    for (long long p = b + lane; p < e; p += 64) { const double d = (double)tval[p] - mu; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if (lane == 0) { mean[g] = mu; ssd[g] = q + (double)(N - (e - b)) * mu * mu; }
}

// column sums of W [N][k] (float64), one workgroup per component, fixed order
__global__ __launch_bounds__(256) void colsum_f64_kernel(const double* __restrict__ W, int N, int k, double* __restrict__ out)
{
    __shared__ double red[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s += W[(size_t)i * k + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[c] = red[0];
}

// out[t][g] = sum_i W[i][t] z(x_ig),  z(x) = (x - mean[g]) inv_std[g] (zs) or x:  with the zeros of the column folded
// into the constant term,  = inv_std[g] (sum_stored W[i][t] x_ig - mean[g] wsum[t]).  Components in chunks of 16.
__global__ __launch_bounds__(256) void csc_xtw_f64_kernel(const long long* __restrict__ tptr, const int* __restrict__ tidx,
                                                          const float* __restrict__ tval, int G, const double* __restrict__ W,
                                                          int k, int zs, const double* __restrict__ mean,
                                                          const double* __restrict__ inv_std, const double* __restrict__ wsum,
                                                          double* __restrict__ out)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    const long long b = tptr[g], e = tptr[g + 1];
    const double mu = zs ? mean[g] : 0.0, is = zs ? inv_std[g] : 1.0;
    for (int t0 = 0; t0 < k; t0 += 16) {
        const int nt = min(16, k - t0);
        double acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0;
        for (long long p = b + lane; p < e; p += 64) {
            const double x = (double)tval[p];
            const double* w = W + (size_t)tidx[p] * k + t0;
#pragma unroll
            for (int t = 0; t < 16; ++t) if (t < nt) acc[t] = fma(w[t], x, acc[t]);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            double a = acc[t];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane == 0 && t < nt) out[(size_t)(t0 + t) * G + g] = (a - mu * wsum[t0 + t]) * is;
        }
    }
}


// sum over the STORED entries of row i of  (x - w_i.h_j)^2 - (w_i.h_j)^2  (float64): with  ||W H||_F^2 = tr(W^T W . H H^T)  added
// by the host this is ||X - W H||_F^2 without ever touching the zeros of X.  Ht: [G][k].  A wavefront per row.
__global__ __launch_bounds__(256) void csr_residual_rows_kernel(const long long* __restrict__ ptr, const int* __restrict__ idx,
                                                                const float* __restrict__ val, int N, const double* __restrict__ W,
                                                                const double* __restrict__ Ht, int k, double* __restrict__ part)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const long long b = ptr[row], e = ptr[row + 1];
    const double* w = W + (size_t)row * k;
    double acc = 0.0;
    for (long long p = b + lane; p < e; p += 64) {
        const double* h = Ht + (size_t)idx[p] * k;
        double s = 0.0;
        for (int c = 0; c < k; ++c) s = fma(w[c], h[c], s);
        const double d = (double)val[p] - s;
        acc += d * d - s * s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) part[row] = acc;
}

}  // namespace cnmf

static void free_csr(cnmf_ctx* c)
{
    hipFree(c->csr_ptr); hipFree(c->csr_idx); hipFree(c->csr_val);
    hipFree(c->csc_ptr); hipFree(c->csc_idx); hipFree(c->csc_val);
    c->csr_ptr = c->csc_ptr = nullptr; c->csr_idx = c->csc_idx = nullptr; c->csr_val = c->csc_val = nullptr;
    c->csr_nnz = -1;
}

// exclusive scan of n device counts (64-bit) into ptr[0..n] through the host (n <= 2^30 rows / 2^24 columns: a few MB)
static int csr_scan_to_ptr(cnmf_ctx* ctx, long long* d_cnt_in_ptr_out, size_t n, long long* total)
{
    std::vector<long long> h(n + 1);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_cnt_in_ptr_out, n * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    long long run = 0;
    for (size_t i = 0; i < n; ++i) { const long long v = h[i]; h[i] = run; run += v; }
    h[n] = run;
    *total = run;
    HIP_TRY(ctx, hipMemcpyAsync(d_cnt_in_ptr_out, h.data(), (n + 1) * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                   // (h is a stack object)
    return CNMF_OK;
}

// compressed rows of X: the uploaded ones, or built here from the resident dense matrix
static int ensure_csr(cnmf_ctx* ctx)
{
    using namespace cnmf;
    if (ctx->csr_ptr) return CNMF_OK;
    if (!ctx->X) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }       // (neither image: no matrix)
    const int N = (int)ctx->N, G = (int)ctx->G;
    hipStream_t st = ctx->stream;
    long long* ptr = nullptr;
    int* idx = nullptr;
    float* val = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&ptr, ((size_t)N + 1) * sizeof(long long)));
    csr_count_dense_kernel<<<(N + 3) / 4, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, ptr);
    long long nnz = 0;
    int rc = hipGetLastError() == hipSuccess ? csr_scan_to_ptr(ctx, ptr, (size_t)N, &nnz) : CNMF_EHIP;
    hipError_t e = hipSuccess;
    if (!rc) e = hipMalloc((void**)&idx, (size_t)std::max<long long>(nnz, 1) * sizeof(int));
    if (!rc && e == hipSuccess) e = hipMalloc((void**)&val, (size_t)std::max<long long>(nnz, 1) * sizeof(float));
    if (!rc && e == hipSuccess) {
        csr_fill_dense_kernel<<<(N + 3) / 4, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, ptr, idx, val);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (rc || e != hipSuccess) {
        hipFree(ptr); hipFree(idx); hipFree(val);
        if (rc) return rc;
        HIP_TRY(ctx, e);
    }
    ctx->csr_ptr = ptr; ctx->csr_idx = idx; ctx->csr_val = val; ctx->csr_nnz = nnz;
    return CNMF_OK;
}

// compressed rows of X^T, from those of X (device counting sort, fixed order)
static int ensure_csc(cnmf_ctx* ctx)
{
    using namespace cnmf;
    if (ctx->csc_ptr) return CNMF_OK;
    int rc = ensure_csr(ctx);
    if (rc) return rc;
    const int N = (int)ctx->N, G = (int)ctx->G;
    const long long nnz = ctx->csr_nnz;
    hipStream_t st = ctx->stream;
    // row chunks: as many as keep the counters at <= 64 M ints, between 64 and 4096
    int T = (int)std::min<long long>(4096, std::max<long long>(64, (64ll << 20) / std::max(1, G)));
    T = std::min(T, N);
    const int rpc = (N + T - 1) / T;
    T = (N + rpc - 1) / rpc;
    DevPool pool;
    int* cnt = pool.get<int>((size_t)T * G, true, st);
    POOL_TRY(ctx, pool);
    long long* tptr = nullptr;
    int* tidx = nullptr;
    float* tval = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&tptr, ((size_t)G + 1) * sizeof(long long)));
    csr_tr_hist_kernel<<<T, 256, 0, st>>>(ctx->csr_ptr, ctx->csr_idx, N, G, rpc, cnt);
    csr_tr_total_kernel<<<(G + 255) / 256, 256, 0, st>>>(cnt, T, G, tptr);
    csr_tr_offsets_kernel<<<(G + 255) / 256, 256, 0, st>>>(cnt, T, G);
    long long total = 0;
    rc = hipGetLastError() == hipSuccess ? csr_scan_to_ptr(ctx, tptr, (size_t)G, &total) : CNMF_EHIP;
    hipError_t e = hipSuccess;
    if (!rc && total != nnz) { SET_ERR(ctx, "transpose of the compressed rows: %lld of %lld entries counted", total, nnz); rc = CNMF_EHIP; }
    if (!rc) e = hipMalloc((void**)&tidx, (size_t)std::max<long long>(nnz, 1) * sizeof(int));
    if (!rc && e == hipSuccess) e = hipMalloc((void**)&tval, (size_t)std::max<long long>(nnz, 1) * sizeof(float));
    if (!rc && e == hipSuccess) {
        csr_tr_fill_kernel<<<T, 64, 0, st>>>(ctx->csr_ptr, ctx->csr_idx, ctx->csr_val, N, G, rpc, cnt, tptr, tidx, tval);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (rc || e != hipSuccess) {
        hipFree(tptr); hipFree(tidx); hipFree(tval);
        if (rc) return rc;
        HIP_TRY(ctx, e);
    }
    ctx->csc_ptr = tptr; ctx->csc_idx = tidx; ctx->csc_val = tval;
    return CNMF_OK;
}
