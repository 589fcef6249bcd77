// Randomized range finder of scikit-learn's `randomized_svd` (sklearn/utils/extmath.py:287-357, 531-602) -- the O(N G)
// part of init='nndsvd' (sklearn/decomposition/_nmf.py:316-354; the reference's `--init nndsvd`, cnmf.py:1252) -- for a
// GROUP of restarts in one call: their (k + 10)-column blocks sit side by side as packed component columns (up to 256),
// every product with X is one pass of the exact-f32 matrix pipe over the whole group, and the normalisation between two
// products never leaves the device.
//
// scikit-learn normalises with a pivoted LU after every product (power_iteration_normalizer='auto' -> 'LU').  ANY
// normaliser Q <- Y R^-1 with invertible R leaves range(Q) -- all the power iteration propagates -- unchanged; here it is
// Cholesky-QR: G = Y^T Y per block on the device (gram_rows_kernel), the c x c Cholesky factor and its inverse on the
// host in float64 (c <= 128), Q = Y R^-1 on the device.  Between two passes over X the condition of a block is that of
// ONE application of X to orthonormal columns, far below the 1/sqrt(eps) limit of Cholesky-QR.  The final basis gets two
// rounds (CholeskyQR2: orthonormal to float32 round-off), then B = Q^T M comes from one more pass.  The (k + 10) x G SVD,
// the sign flip and the positive / negative split stay on the host (Engine.nndsvd_init_batch).
#pragma once

namespace cnmf {

// Q[:, j] = sum_{i <= j} Y[:, i] * Rinv[i][j], in place, one thread per position; j descending so that no input is
// overwritten before its last use.  grid = (ceil(L / 256), blocks); Rinv [block][cmax][cmax] row-major (upper triangular)
__global__ __launch_bounds__(256) void block_apply_rinv_kernel(float* __restrict__ V, int ldv, int L,
                                                               const SlotDesc* __restrict__ blocks,
                                                               const float* __restrict__ Rinv, int cmax)
{
    const SlotDesc sd = blocks[blockIdx.y];
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= L) return;
    const float* R = Rinv + (size_t)blockIdx.y * cmax * cmax;
    float* v = V + (size_t)sd.off * ldv + pos;
    for (int j = sd.k - 1; j >= 0; --j) {
        float acc = 0.f;
        for (int i = 0; i <= j; ++i) acc = fmaf(v[(size_t)i * ldv], R[i * cmax + j], acc);
        v[(size_t)j * ldv] = acc;
    }
}

// Gram matrices of the blocks for LONG vectors (L = cells): gram_rows_kernel walks all L positions of a block in ONE
// workgroup (1.1 ms at L = 50 000 -- a third of the whole initialisation).  Here the positions are cut into chunks:
// part[block][chunk][c][c] (float64) from grid (blocks, chunks), then one workgroup per block adds the chunks in order.
template <int NP>                                    // pairs (a, b) per thread: c * c <= 256 * NP
__global__ __launch_bounds__(256) void gram_chunk_kernel(const float* __restrict__ V, int ldv, int L,
                                                         const SlotDesc* __restrict__ blocks, int chunk_len,
                                                         double* __restrict__ part, int cmax)
{
    constexpr int GC = 64;
    __shared__ float tile[KMAX][GC + 1];
    const SlotDesc sd = blocks[blockIdx.x];
    const int c = sd.k, off = sd.off, tid = threadIdx.x;
    const int g_lo = blockIdx.y * chunk_len, g_hi = min(L, g_lo + chunk_len);
    double acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = 0.0;
    for (int g0 = g_lo; g0 < g_hi; g0 += GC) {
        for (int e = tid; e < c * GC; e += 256) {
            const int r = e / GC, g = e % GC;
            tile[r][g] = (g0 + g < g_hi) ? V[(size_t)(off + r) * ldv + g0 + g] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = tid + 256 * i;
            if (e < c * c) {
                const int a = e / c, b = e % c;
                float s = 0.f;
#pragma unroll 8
                for (int g = 0; g < GC; ++g) s = fmaf(tile[a][g], tile[b][g], s);
                acc[i] += (double)s;
            }
        }
        __syncthreads();
    }
    double* out = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * cmax * cmax;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = tid + 256 * i;
        if (e < c * c) out[(e / c) * cmax + e % c] = acc[i];
    }
}

__global__ __launch_bounds__(256) void gram_chunk_reduce_kernel(const double* __restrict__ part, int nchunks, int cmax,
                                                                const SlotDesc* __restrict__ blocks,
                                                                float* __restrict__ gram_out)
{
    const int c = blocks[blockIdx.x].k;
    for (int e = threadIdx.x; e < c * c; e += 256) {
        const int a = e / c, b = e % c;
        double s = 0.0;
        for (int q = 0; q < nchunks; ++q) s += part[((size_t)blockIdx.x * nchunks + q) * cmax * cmax + a * cmax + b];
        gram_out[(size_t)blockIdx.x * GRAM_SZ + a * GRAM_LD + b] = (float)s;
    }
}

}  // namespace cnmf

// transpose = 0: M = X (rows = cells), 1: M = X^T.  Q0 [M_cols][C] row-major (C = sum of widths <= 256);
// Q_out [M_rows][C]; B_out [C][M_cols] = Q^T M.
extern "C" int cnmf_range_finder(cnmf_ctx* ctx, int transpose, int nblocks, const int32_t* widths, const float* Q0,
                                 int n_iter, float* Q_out, float* B_out)
{
    using namespace cnmf;
    if (!ctx || !widths || !Q0 || !Q_out || !B_out || nblocks < 1 || n_iter < 0) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
    int C = 0, cmax = 0;
    for (int b = 0; b < nblocks; ++b) {
        if (widths[b] < 1 || widths[b] > KMAX) { SET_ERR(ctx, "block width %d outside 1..%d", widths[b], KMAX); return CNMF_EUNSUPPORTED; }
        C += widths[b]; cmax = std::max(cmax, (int)widths[b]);
    }
    if (C > 256) { SET_ERR(ctx, "%d columns exceed one pass (256)", C); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int N = (int)ctx->N, G = (int)ctx->G, Np = ctx->N_pad, Gp = ctx->G_pad;
    const int KC = C <= 32 ? 32 : (C <= 64 ? 64 : (C <= 128 ? 128 : 256));
    // M_rows / M_cols and their padded lengths
    const int Lr = transpose ? G : N, Lrp = transpose ? Gp : Np;
    const int Lc = transpose ? N : G, Lcp = transpose ? Np : Gp;
    DevPool pool;
    float* dIn = pool.get<float>((size_t)std::max(Lr, Lc) * C);
    float* dR = pool.get<float>((size_t)KC * Lrp, true, st);          // component-major [KC][M_rows]
    float* dCm = pool.get<float>((size_t)KC * Lcp, true, st);         // component-major [KC][M_cols]
    const int nsplit = std::max(1, std::min(16, Np / 2048));          // K split of the products that contract over cells
    float* dS = pool.get<float>((size_t)nsplit * KC * Gp);            // their split-K partial planes
    SlotDesc* dBlk = pool.get<SlotDesc>(nblocks);
    int* dList = pool.get<int>(nblocks);
    float* dGram = pool.get<float>((size_t)nblocks * GRAM_SZ);
    float* dRinv = pool.get<float>((size_t)nblocks * cmax * cmax);
    constexpr int GRAM_CHUNKS = 64;
    double* dGpart = pool.get<double>((size_t)nblocks * GRAM_CHUNKS * cmax * cmax);
    POOL_TRY(ctx, pool);
    std::vector<SlotDesc> blk(nblocks);
    std::vector<int> list(nblocks);
    int off = 0;
    for (int b = 0; b < nblocks; ++b) {
        memset(&blk[b], 0, sizeof(SlotDesc));
        blk[b].off = off; blk[b].k = widths[b]; blk[b].active = 1; blk[b].restart = b;
        list[b] = b; off += widths[b];
    }
    HIP_TRY(ctx, hipMemcpyAsync(dBlk, blk.data(), nblocks * sizeof(SlotDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dList, list.data(), nblocks * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dIn, Q0, (size_t)Lc * C * sizeof(float), hipMemcpyHostToDevice, st));
    {   // row-major [M_cols][C] -> component-major rows (install_kernel's W path transposes)
        dim3 gI((Lc + 255) / 256, C);
        install_kernel<<<gI, 256, 0, st>>>(nullptr, dIn, dCm, Lcp, 0, dCm, Lcp, Lc, 0, C);
        HIP_TRY(ctx, hipGetLastError());
    }
    // dst[KC][rows of M] = src . M^T   (contracts over M's columns)
    auto mul_M = [&](const float* src, float* dst) -> int {
        if (!transpose) HIP_TRY(ctx, launch_gemm<false>(st, 0, src, Gp, ctx->X, Gp, dst, Np, 0, KC, Gp, Np, 1));
        else {
            HIP_TRY(ctx, launch_gemm<true>(st, 0, src, Np, ctx->X, Gp, dS, Gp, (long long)KC * Gp, KC, Np, Gp, nsplit));
            HIP_TRY(ctx, launch_reduce_splits(st, dS, nsplit, (long long)KC * Gp, (long long)KC * Gp));
            HIP_TRY(ctx, hipMemcpyAsync(dst, dS, (size_t)KC * Gp * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        return CNMF_OK;
    };
    // dst[KC][cols of M] = src . M    (contracts over M's rows)
    auto mul_Mt = [&](const float* src, float* dst) -> int {
        if (transpose) HIP_TRY(ctx, launch_gemm<false>(st, 0, src, Gp, ctx->X, Gp, dst, Np, 0, KC, Gp, Np, 1));
        else {
            HIP_TRY(ctx, launch_gemm<true>(st, 0, src, Np, ctx->X, Gp, dS, Gp, (long long)KC * Gp, KC, Np, Gp, nsplit));
            HIP_TRY(ctx, launch_reduce_splits(st, dS, nsplit, (long long)KC * Gp, (long long)KC * Gp));
            HIP_TRY(ctx, hipMemcpyAsync(dst, dS, (size_t)KC * Gp * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        return CNMF_OK;
    };
    std::vector<float> hg((size_t)nblocks * GRAM_SZ), hr((size_t)nblocks * cmax * cmax);
    // one Cholesky-QR round of every block of V [KC][ld] (L valid positions)
    auto normalise = [&](float* V, int ld, int L) -> int {
        if (L >= 8192) {              // long vectors: positions cut into chunks, partial Grams added in chunk order
            const int chunk_len = round_up((L + GRAM_CHUNKS - 1) / GRAM_CHUNKS, 64);
            const int nch = (L + chunk_len - 1) / chunk_len;
            const dim3 gg(nblocks, nch);
            if (cmax <= 32) gram_chunk_kernel<4><<<gg, 256, 0, st>>>(V, ld, L, dBlk, chunk_len, dGpart, cmax);
            else if (cmax <= 64) gram_chunk_kernel<16><<<gg, 256, 0, st>>>(V, ld, L, dBlk, chunk_len, dGpart, cmax);
            else gram_chunk_kernel<64><<<gg, 256, 0, st>>>(V, ld, L, dBlk, chunk_len, dGpart, cmax);
            gram_chunk_reduce_kernel<<<nblocks, 256, 0, st>>>(dGpart, nch, cmax, dBlk, dGram);
        } else {
            gram_rows_kernel<<<nblocks, 256, 0, st>>>(V, ld, L, dBlk, dList, dGram, 0.f);
        }
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(hg.data(), dGram, hg.size() * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        std::fill(hr.begin(), hr.end(), 0.f);
        std::vector<double> Rm((size_t)cmax * cmax), Ri((size_t)cmax * cmax);
        for (int b = 0; b < nblocks; ++b) {
            const int c = widths[b];
            const float* g = hg.data() + (size_t)b * GRAM_SZ;
            // upper Cholesky factor R (G = R^T R), float64.  A rank-deficient block (more columns than the matrix has rank:
            // min(N, G) < k + 10, a noiseless low-rank X, duplicated rows) leaves a pivot that is zero up to the round-off
            // of the float32 Gram matrix: that direction lies numerically in the span of the earlier columns and is DROPPED
            // -- row j of R and column j of R^-1 are zero, the column of Q becomes a zero vector and stays one through the
            // remaining passes.  (Dividing by a floored pivot instead compounds over several deficient columns: R^-1
            // overflows float32 and the basis turns into NaN -- round-3 advisor finding.)  range(Q) is still the range
            // of the block; scikit-learn's pivoted LU reaches the same subspace with arbitrary vectors in the null part.
            std::fill(Rm.begin(), Rm.end(), 0.0);
            std::vector<char> dead(c, 0);
            for (int j = 0; j < c; ++j) {
                for (int i = 0; i <= j; ++i) {
                    if (i < j && dead[i]) continue;                    // row i of R is zero
                    double s = g[i * GRAM_LD + j];
                    for (int p = 0; p < i; ++p) s -= Rm[(size_t)p * cmax + i] * Rm[(size_t)p * cmax + j];
                    if (i == j) {
                        // the residual of column j against its own squared norm: below ~10 x the round-off of the Gram
                        // entries (products in float32) it carries no direction
                        const double own = (double)g[j * GRAM_LD + j];
                        if (!(s > own * 1e-6) || !(s > 1e-300) || !std::isfinite(s)) dead[j] = 1;
                        else Rm[(size_t)j * cmax + j] = std::sqrt(s);
                    } else Rm[(size_t)i * cmax + j] = s / Rm[(size_t)i * cmax + i];
                }
            }
            // inverse of the upper-triangular R by back substitution, column by column (dead rows / columns stay zero)
            std::fill(Ri.begin(), Ri.end(), 0.0);
            for (int j = 0; j < c; ++j) {
                if (dead[j]) continue;
                Ri[(size_t)j * cmax + j] = 1.0 / Rm[(size_t)j * cmax + j];
                for (int i = j - 1; i >= 0; --i) {
                    if (dead[i]) continue;
                    double s = 0.0;
                    for (int p = i + 1; p <= j; ++p) s += Rm[(size_t)i * cmax + p] * Ri[(size_t)p * cmax + j];
                    Ri[(size_t)i * cmax + j] = -s / Rm[(size_t)i * cmax + i];
                }
            }
            float* o = hr.data() + (size_t)b * cmax * cmax;
            for (int i = 0; i < c; ++i)
                for (int j = i; j < c; ++j) o[i * cmax + j] = (float)Ri[(size_t)i * cmax + j];
        }
        HIP_TRY(ctx, hipMemcpyAsync(dRinv, hr.data(), hr.size() * sizeof(float), hipMemcpyHostToDevice, st));
        block_apply_rinv_kernel<<<dim3((L + 255) / 256, nblocks), 256, 0, st>>>(V, ld, L, dBlk, dRinv, cmax);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(st));               // hr is reused by the next round
        return CNMF_OK;
    };
    int rc;
    for (int it = 0; it < n_iter; ++it) {
        if ((rc = mul_M(dCm, dR))) return rc;                 // Y = M Q
        if ((rc = normalise(dR, Lrp, Lr))) return rc;
        if ((rc = mul_Mt(dR, dCm))) return rc;                // Z = M^T Y
        if ((rc = normalise(dCm, Lcp, Lc))) return rc;
    }
    if ((rc = mul_M(dCm, dR))) return rc;                     // the final basis: Q = qr(M Q), twice for orthonormality
    if ((rc = normalise(dR, Lrp, Lr))) return rc;
    if ((rc = normalise(dR, Lrp, Lr))) return rc;
    if ((rc = mul_Mt(dR, dCm))) return rc;                    // B = Q^T M  (component-major = [C][M_cols])
    dim3 gQ((Lr + 255) / 256, C), gB((Lc + 255) / 256, C);
    extract_kernel<<<gQ, 256, 0, st>>>(dR, Lrp, Lr, 0, C, dIn, 1);                   // -> [M_rows][C]
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(Q_out, dIn, (size_t)Lr * C * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    extract_kernel<<<gB, 256, 0, st>>>(dCm, Lcp, Lc, 0, C, dIn, 0);                  // -> [C][M_cols]
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(B_out, dIn, (size_t)Lc * C * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return CNMF_OK;
}
