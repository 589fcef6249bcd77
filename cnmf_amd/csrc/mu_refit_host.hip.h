// Kullback-Leibler refit with one factor fixed, in FLOAT64, on the stored entries of the resident matrix (round 5).
//
// What it replaces: the three `non_negative_factorization(X, H=..., update_H=False, solver='mu', beta_loss='kullback-leibler')`
// calls of a consensus run whose restarts used the Kullback-Leibler loss (cnmf.py:776-820 through :920, :952, :972) --
//   refit_usage(norm_counts.X, median_spectra)           rows = cells  of the normalised counts,  fixed H = spectra
//   refit_spectra(tpm.X, norm_usages)  = on tpm.X^T      rows = GENES of the TPM matrix,          fixed H = usages^T
//   refit_usage(tpm[:, hvgs] / std, spectra_tpm / std)   rows = cells  of a column subset, columns divided by a constant
// on float64 matrices (cnmf.py:534), i.e. scikit-learn's `_fit_multiplicative_update` with update_H=False
// (sklearn/decomposition/_nmf.py:731-893, `_multiplicative_update_w` :526-631, `_beta_divergence` :84-194) -- which for
// beta = 1 only touches the stored entries of a scipy.sparse X (`_special_sparse_dot`, :192).
//
// With H fixed the rows of W do not interact: row i iterates
//     w_i <- w_i * ( sum_j  x_ij / max(w_i . h_j, eps32) * h_j ) / ( sum_j h_j + l1 + l2 w_i )
// and only the stopping rule -- every 10 iterations, (previous - error) / error_at_init < tol on the GLOBAL divergence --
// couples them.  One launch therefore runs up to TEN iterations of every row (a wavefront per row, its entries lane-strided,
// the k-vector in registers, fixed-order butterfly sums) and leaves the row's share of the divergence; the host adds the
// shares in row order and applies scikit-learn's rule.  The transposed problem walks the compressed rows of X^T that
// csr_host.hip.h builds on the device: no `.todense()`, no transposed upload.
// Arithmetic is float64 throughout (the matrix values are the resident float32 image, as for the float64 NNLS refits of
// tail_host.hip.h).  Included by cnmf_hip.hip.
#pragma once

namespace cnmf {

constexpr double MU_EPS32 = 1.1920928955078125e-07;        // scikit-learn's EPSILON = np.finfo(np.float32).eps (_nmf.py:39)

// Ht: [ncols][KP] (the fixed factor, one row per COLUMN of the walked matrix, zero padded to KP); coldiv (nullable):
// x'_ij = x_ij / coldiv[j], coldiv[j] == 0 drops column j (the column subset of the final usage refit);
// W: [nrows][KP] in / out;  hsum: [KP] = sum_j Ht[j][c] over the columns that count;
// err_part[row] = sum_{x' > eps} (x' log(x' / max(wh, eps)) - x') + sum_c w_c hsum_c   after the last iteration
// BETA = 0 (Itakura-Saito, round 6): scikit-learn refuses beta_loss <= 0 on a matrix that contains a zero (_nmf.py:1679-1684),
// so every entry of the walked matrix is a STORED entry and the same walk covers the dense update
//     w_i <- w_i * sqrt( (sum_j x_ij / s_ij^2 h_j) / (sum_j 1 / s_ij h_j + l1 + l2 w_i) ),   s_ij = max(w_i . h_j, eps32)
// (`_multiplicative_update_w` :575-631 with gamma = 1 / (2 - beta) = 1/2, :848-851; W < eps64 -> 0, :858-860); a dropped
// column (coldiv == 0) has x' = 0 and h = 0 and adds nothing to either sum.  Divergence: sum (x / s - log(x / s)) - rows x
// columns (`_beta_divergence` :158-161; ncount = the column count of the matrix meant).
template <int KP, int BETA = 1>
__global__ __launch_bounds__(256) void mu_refit_f64_kernel(const long long* __restrict__ ptr, const int* __restrict__ idx,
                                                           const float* __restrict__ val, int nrows,
                                                           const double* __restrict__ Ht, const double* __restrict__ coldiv,
                                                           double* __restrict__ W, const double* __restrict__ hsum,
                                                           double l1, double l2, int n_inner, double* __restrict__ err_part,
                                                           double ncount = 0.0)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const long long b = ptr[row], e = ptr[row + 1];
    double w[KP];
#pragma unroll
    for (int c = 0; c < KP; ++c) w[c] = W[(size_t)row * KP + c];
    for (int it = 0; it < n_inner; ++it) {
        double acc[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) acc[c] = 0.0;
        if constexpr (BETA == 1) {
        for (long long p = b + lane; p < e; p += 64) {
            const int j = idx[p];
            double x = (double)val[p];
            if (coldiv) { const double d = coldiv[j]; x = d != 0.0 ? x / d : 0.0; }
            const double* h = Ht + (size_t)j * KP;
            if constexpr (KP <= 32) {
                double hv[KP];
#pragma unroll
                for (int c = 0; c < KP; ++c) hv[c] = h[c];
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < KP; ++c) s += w[c] * hv[c];
                const double q = x / fmax(s, MU_EPS32);
#pragma unroll
                for (int c = 0; c < KP; ++c) acc[c] += q * hv[c];
            } else {                               // (rank 33..64: the factor row is read twice instead of held)
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < KP; ++c) s += w[c] * h[c];
                const double q = x / fmax(s, MU_EPS32);
#pragma unroll
                for (int c = 0; c < KP; ++c) acc[c] += q * h[c];
            }
        }
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            double a = acc[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            double den = hsum[c] + l1 + l2 * w[c];
            if (den == 0.0) den = MU_EPS32;
            w[c] *= a / den;
        }
        } else {
        double dacc[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) dacc[c] = 0.0;
        for (long long p = b + lane; p < e; p += 64) {
            const int j = idx[p];
            double x = (double)val[p];
            if (coldiv) { const double d = coldiv[j]; x = d != 0.0 ? x / d : 0.0; }
            const double* h = Ht + (size_t)j * KP;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < KP; ++c) s += w[c] * h[c];
            s = fmax(s, MU_EPS32);
            const double r = 1.0 / s;               // numerator: WH ** -1, ** 2, * X (_nmf.py:590-595); denominator: WH ** (beta - 1) (:618)
            const double q = (r * r) * x;
#pragma unroll
            for (int c = 0; c < KP; ++c) { const double hv = h[c]; acc[c] += q * hv; dacc[c] += r * hv; }
        }
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            double a = acc[c], d = dacc[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); d += __shfl_xor(d, o, 64); }
            double den = d + l1 + l2 * w[c];
            if (den == 0.0) den = MU_EPS32;
            double nw = w[c] * sqrt(a / den);
            if (nw < 2.220446049250313e-16) nw = 0.0;
            w[c] = nw;
        }
        }
    }
    if (n_inner > 0 && lane == 0) {
#pragma unroll
        for (int c = 0; c < KP; ++c) W[(size_t)row * KP + c] = w[c];
    }
    if (err_part) {
        double r = 0.0;
        for (long long p = b + lane; p < e; p += 64) {
            const int j = idx[p];
            double x = (double)val[p];
            if (coldiv) { const double d = coldiv[j]; x = d != 0.0 ? x / d : 0.0; }
            if (!(x > MU_EPS32)) continue;
            const double* h = Ht + (size_t)j * KP;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < KP; ++c) s += w[c] * h[c];
            if constexpr (BETA == 1) r += x * log(x / fmax(s, MU_EPS32)) - x;
            else { const double dv = x / fmax(s, MU_EPS32); r += dv - log(dv); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        if (lane == 0) {
            if constexpr (BETA == 1) {
                double swh = 0.0;
#pragma unroll
                for (int c = 0; c < KP; ++c) swh += w[c] * hsum[c];
                err_part[row] = r + swh;
            } else {
                err_part[row] = r - ncount;          // - prod(X.shape), one row's share
            }
        }
    }
}

// fixed-order sum of the per-row shares: 1024 strided partial sums, then one thread
__global__ __launch_bounds__(1024) void mu_refit_err_kernel(const double* __restrict__ part, int n, double* __restrict__ out)
{
    __shared__ double red[1024];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

__global__ __launch_bounds__(256) void mu_refit_fill_kernel(double* __restrict__ W, long long n, int KP, int k, double v)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) W[i] = ((int)(i % KP) < k) ? v : 0.0;
}

}  // namespace cnmf

// side 0: rows = cells (H is k x n_genes); side 1: rows = genes of the resident matrix, i.e. the problem on X^T (H is
// k x n_cells).  W_out: [rows][k] float64.  w_init: scikit-learn's avg = sqrt(X.mean() / k) of the matrix the caller MEANS
// (with coldiv: of the divided column subset).  prm: tol, max_iter, l1_reg_W, l2_reg_W.
extern "C" int cnmf_mu_refit_f64(cnmf_ctx* ctx, int side, int beta, int k, const double* H, const double* coldiv, double w_init,
                                 const cnmf_cd_params* prm, double* W_out, int32_t* n_iter_out, double* err_out)
{
    using namespace cnmf;
    if (!ctx || !H || !prm || !W_out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (beta != 0 && beta != 1) { SET_ERR(ctx, "beta must be 1 (Kullback-Leibler) or 0 (Itakura-Saito)"); return CNMF_EINVAL; }
    if (!ctx->X && !ctx->csr_ptr) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }
    if (k < 1 || k > CNMF_MU_KMAX) { SET_ERR(ctx, "rank %d outside 1..%d (multiplicative updates)", k, CNMF_MU_KMAX); return CNMF_EUNSUPPORTED; }
    if (side != 0 && side != 1) { SET_ERR(ctx, "side must be 0 (rows = cells) or 1 (rows = genes)"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = side == 0 ? ensure_csr(ctx) : ensure_csc(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    const int nrows = side == 0 ? (int)ctx->N : (int)ctx->G, ncols = side == 0 ? (int)ctx->G : (int)ctx->N;
    const long long* ptr = side == 0 ? ctx->csr_ptr : ctx->csc_ptr;
    const int* idx = side == 0 ? ctx->csr_idx : ctx->csc_idx;
    const float* val = side == 0 ? ctx->csr_val : ctx->csc_val;
    const int KP = k <= 8 ? 8 : (k <= 16 ? 16 : (k <= 32 ? 32 : 64));
    if (beta == 0) {
        // scikit-learn's own refusal (_nmf.py:1679-1684): every entry must be positive, i.e. every entry is stored
        long long nnz = 0;
        HIP_TRY(ctx, hipMemcpy(&nnz, ptr + nrows, sizeof(long long), hipMemcpyDeviceToHost));
        if (nnz != (long long)nrows * ncols) {
            SET_ERR(ctx, "When beta_loss <= 0 and X contains zeros, the solver may diverge. Please add small values to X, or use a positive beta_loss.");
            return CNMF_EINVAL;
        }
    }
    double ncount = 0.0;                            // columns of the matrix meant (Itakura-Saito divergence: - prod(X.shape))
    for (int j = 0; j < ncols; ++j) ncount += (coldiv && coldiv[j] == 0.0) ? 0.0 : 1.0;
    // the fixed factor, one row per column of the walked matrix; its row sums over the columns that count
    std::vector<double> ht((size_t)ncols * KP, 0.0), hsum(KP, 0.0);
    for (int c = 0; c < k; ++c) {
        const double* h = H + (size_t)c * ncols;
        double s = 0.0;
        for (int j = 0; j < ncols; ++j) {
            const double v = (coldiv && coldiv[j] == 0.0) ? 0.0 : h[j];
            ht[(size_t)j * KP + c] = v;
            s += v;
        }
        hsum[c] = s;
    }
    DevPool pool;
    double* dHt = pool.get<double>(ht.size());
    double* dHs = pool.get<double>(KP);
    double* dDiv = coldiv ? pool.get<double>(ncols) : nullptr;
    double* dW = pool.get<double>((size_t)nrows * KP);
    double* dPart = pool.get<double>(nrows);
    double* dErr = pool.get<double>(1);
    POOL_TRY(ctx, pool);
    HIP_TRY(ctx, hipMemcpyAsync(dHt, ht.data(), ht.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dHs, hsum.data(), KP * sizeof(double), hipMemcpyHostToDevice, st));
    if (coldiv) HIP_TRY(ctx, hipMemcpyAsync(dDiv, coldiv, (size_t)ncols * sizeof(double), hipMemcpyHostToDevice, st));
    {
        const long long n = (long long)nrows * KP;
        mu_refit_fill_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(dW, n, KP, k, w_init);
    }
    const double l1 = prm->l1_reg_W, l2 = prm->l2_reg_W;
    auto step = [&](int n_inner, double* err) -> int {
        const unsigned grid = (unsigned)((nrows + 3) / 4);
#define CNMF_MU_REFIT_LAUNCH(KPV, BV) mu_refit_f64_kernel<KPV, BV><<<grid, 256, 0, st>>>(ptr, idx, val, nrows, dHt, dDiv, dW, dHs, l1, l2, \
                                                                                     n_inner, err ? dPart : nullptr, ncount)
        if (beta == 1) {
            switch (KP) {
                case 8:  CNMF_MU_REFIT_LAUNCH(8, 1); break;
                case 16: CNMF_MU_REFIT_LAUNCH(16, 1); break;
                case 32: CNMF_MU_REFIT_LAUNCH(32, 1); break;
                default: CNMF_MU_REFIT_LAUNCH(64, 1); break;
            }
        } else {
            switch (KP) {
                case 8:  CNMF_MU_REFIT_LAUNCH(8, 0); break;
                case 16: CNMF_MU_REFIT_LAUNCH(16, 0); break;
                case 32: CNMF_MU_REFIT_LAUNCH(32, 0); break;
                default: CNMF_MU_REFIT_LAUNCH(64, 0); break;
            }
        }
#undef CNMF_MU_REFIT_LAUNCH
        HIP_TRY(ctx, hipGetLastError());
        if (err) {
            double res = 0.0;
            mu_refit_err_kernel<<<1, 1024, 0, st>>>(dPart, nrows, dErr);
            HIP_TRY(ctx, hipMemcpyAsync(&res, dErr, sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
            *err = std::sqrt(2.0 * std::max(res, 0.0));                  // _beta_divergence(square_root=True)
        }
        return CNMF_OK;
    };
    double err0 = 0.0, prev = 0.0, err = 0.0;
    rc = step(0, &err0);
    if (rc) return rc;
    prev = err = err0;
    int it = 0;
    bool err_current = true;
    while (it < prm->max_iter) {
        const int n_inner = std::min(10 - it % 10, prm->max_iter - it);
        const bool check = prm->tol > 0 && (it + n_inner) % 10 == 0;
        rc = step(n_inner, check ? &err : nullptr);
        if (rc) return rc;
        it += n_inner;
        err_current = check;
        if (check) {
            if ((prev - err) / err0 < prm->tol) break;
            prev = err;
        }
    }
    if (!err_current) {                                                   // report the divergence of the FINAL factors
        rc = step(0, &err);
        if (rc) return rc;
    }
    std::vector<double> hw((size_t)nrows * KP);
    HIP_TRY(ctx, hipMemcpyAsync(hw.data(), dW, hw.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int r = 0; r < nrows; ++r)
        for (int c = 0; c < k; ++c) W_out[(size_t)r * k + c] = hw[(size_t)r * KP + c];
    if (n_iter_out) *n_iter_out = it;
    if (err_out) *err_out = err;
    return CNMF_OK;
}
