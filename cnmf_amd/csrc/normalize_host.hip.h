// Gene-wise unit-variance scaling of the resident matrix on the device -- the dense branch of the
// reference's get_norm_counts (cnmf.py:540-554):
//     norm_counts.X = counts[:, hvgs].astype(float64);  norm_counts.X /= norm_counts.X.std(axis=0, ddof=1)
//     zerocells = norm_counts.X.sum(axis=1) == 0  ->  Exception
// Statistics are accumulated in float64 (two passes: mean, then sum of squared deviations) in a fixed
// order; the division is done in float64 and rounded once to the float32 the kernels work in.
// Included by cnmf_hip.hip.
#pragma once

namespace cnmf {

constexpr int NORM_ROWS = 256;          // rows per partial of the column statistics

// out[chunk][g] = sum over the chunk's rows of x (mean == nullptr) or of (x - mean[g])^2
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                          const double* __restrict__ mean, double* __restrict__ out)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const int r0 = blockIdx.y * NORM_ROWS, r1 = min(N, r0 + NORM_ROWS);
    const double mu = mean ? mean[g] : 0.0;
    double s = 0.0;
    if (mean) for (int r = r0; r < r1; ++r) { const double d = (double)X[(size_t)r * ld + g] - mu; s += d * d; }
    else      for (int r = r0; r < r1; ++r) s += (double)X[(size_t)r * ld + g];
    out[(size_t)blockIdx.y * G + g] = s;
}

// sum the partials in chunk order (deterministic)
__global__ __launch_bounds__(256) void col_combine_kernel(const double* __restrict__ part, int chunks, int G,
                                                          double scale, double* __restrict__ out)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += part[(size_t)c * G + g];
    out[g] = s * scale;
}

__global__ __launch_bounds__(256) void scale_cols_kernel(float* __restrict__ X, int ld, int N, int G,
                                                         const double* __restrict__ divisor)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const double d = divisor[g];
    const int r0 = blockIdx.y * NORM_ROWS, r1 = min(N, r0 + NORM_ROWS);
    for (int r = r0; r < r1; ++r) {
        float* p = X + (size_t)r * ld + g;
        *p = (float)((double)*p / d);
    }
}

// one wave per row: float64 row sums (lanes stride the genes; fixed butterfly order)
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ X, int ld, int N, int G,
                                                      double* __restrict__ out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    double s = 0.0;
    for (int g = lane; g < G; g += 64) s += (double)X[(size_t)row * ld + g];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[row] = s;
}

}  // namespace cnmf

// the planes of X (split-operand GEMM) and the count structure (integer planes, per-gene scale) describe
// the old values: drop them whenever X changes (mirrors alloc_matrix)
static void invalidate_planes(cnmf_ctx* ctx)
{
    hipStreamSynchronize(ctx->stream);
    hipFree(ctx->X3); hipFree(ctx->Xt3); hipFree(ctx->XtF);
    ctx->X3 = ctx->Xt3 = nullptr; ctx->XtF = nullptr;
    free_mu_sparse(ctx); free_csr(ctx);
    hipFree(ctx->C1); hipFree(ctx->Ct1); hipFree(ctx->d_scale);
    hipFree(ctx->C1h); hipFree(ctx->Ct1h); hipFree(ctx->hiA); hipFree(ctx->hiB);
    ctx->C1 = ctx->Ct1 = ctx->C1h = ctx->Ct1h = nullptr; ctx->hiA = ctx->hiB = nullptr;
    ctx->d_scale = nullptr; ctx->count_state = 0; ctx->count_fmt = 0;
    // (the resident spectra store outlives a change of the matrix, like in alloc_matrix: it carries its own gene count)
}

extern "C" int cnmf_col_moments(cnmf_ctx* ctx, double* mean_out, double* ssd_out)
{
    using namespace cnmf;
    if (!ctx || !mean_out || !ssd_out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (!ctx->X && ctx->csr_ptr) {
        // a matrix that lives as compressed rows only (a CSR upload nobody has asked the dense image of): the same
        // statistics from the stored entries of every column, float64 (round 5)
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (int rc_ = ensure_csc(ctx)) return rc_;
        const int N_ = (int)ctx->N, G_ = (int)ctx->G;
        DevPool pool_;
        double* m_ = pool_.get<double>(G_);
        double* s_ = pool_.get<double>(G_);
        POOL_TRY(ctx, pool_);
        csc_col_moments_kernel<<<(G_ + 3) / 4, 256, 0, ctx->stream>>>(ctx->csc_ptr, ctx->csc_val, G_, N_, m_, s_);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(mean_out, m_, (size_t)G_ * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ssd_out, s_, (size_t)G_ * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return CNMF_OK;
    }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int N = (int)ctx->N, G = (int)ctx->G, chunks = (N + NORM_ROWS - 1) / NORM_ROWS;
    DevPool pool;
    double* part = pool.get<double>((size_t)chunks * G);
    double* mean = pool.get<double>(G);
    double* ssd = pool.get<double>(G);
    POOL_TRY(ctx, pool);
    dim3 grid((G + 255) / 256, chunks);
    col_partial_kernel<<<grid, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, nullptr, part);
    col_combine_kernel<<<(G + 255) / 256, 256, 0, st>>>(part, chunks, G, 1.0 / (double)N, mean);
    col_partial_kernel<<<grid, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, mean, part);
    col_combine_kernel<<<(G + 255) / 256, 256, 0, st>>>(part, chunks, G, 1.0, ssd);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(mean_out, mean, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(ssd_out, ssd, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return CNMF_OK;
}

extern "C" int cnmf_scale_columns(cnmf_ctx* ctx, const double* divisor)
{
    using namespace cnmf;
    if (!ctx || !divisor) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
    const int N = (int)ctx->N, G = (int)ctx->G;
    for (int g = 0; g < G; ++g)
        if (!(divisor[g] > 0.0) || !std::isfinite(divisor[g])) {
            SET_ERR(ctx, "divisor of gene %d is %g (a gene without variance cannot be scaled to unit variance)", g, divisor[g]);
            return CNMF_EINVAL;
        }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevPool pool;
    double* d = pool.get<double>(G);
    POOL_TRY(ctx, pool);
    HIP_TRY(ctx, hipMemcpyAsync(d, divisor, (size_t)G * sizeof(double), hipMemcpyHostToDevice, st));
    dim3 grid((G + 255) / 256, (N + NORM_ROWS - 1) / NORM_ROWS);
    scale_cols_kernel<<<grid, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, d);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    invalidate_planes(ctx);
    return CNMF_OK;
}

extern "C" int cnmf_row_sums(cnmf_ctx* ctx, double* out)
{
    using namespace cnmf;
    if (!ctx || !out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int N = (int)ctx->N, G = (int)ctx->G;
    DevPool pool;
    double* d = pool.get<double>(N);
    POOL_TRY(ctx, pool);
    row_sum_kernel<<<(N + 3) / 4, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, d);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, d, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return CNMF_OK;
}

extern "C" int cnmf_get_matrix(cnmf_ctx* ctx, float* out)
{
    if (!ctx || !out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy2DAsync(out, (size_t)ctx->G * sizeof(float), ctx->X, (size_t)ctx->G_pad * sizeof(float),
                                  (size_t)ctx->G * sizeof(float), (size_t)ctx->N, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNMF_OK;
}
