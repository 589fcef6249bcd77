// Host driver of the multiplicative-update solver (included by cnmf_hip.hip).
// Restates _fit_multiplicative_update (sklearn/decomposition/_nmf.py:731-893): W then H
// half-steps, beta-divergence every 10 iterations, stop when (previous - error)/error_at_init < tol.
// Restarts run one after the other (each needs its own N x G ratio pass; batching them over
// a shared X tile is a round-2 item) -- this is the non-default path of the reference
// (beta_loss='frobenius' -> CD is the default, cnmf.py:334,630).
#pragma once
#include "kernels_mu.hip.h"

namespace cnmf {

template <int KP, bool BETA1>
static int mu_run_one(cnmf_ctx* ctx, hipStream_t st, int N, int G, int k, float* dW, float* dHt,
                      float* dHsum, float* dWsum, float* pnum, float* pden, int nchunks, int rpc,
                      double* dpart, double* dcs, int update_H, const cnmf_cd_params* prm, int* n_iter_out,
                      double* err_out)
{
    const int ldx = ctx->G_pad;
    const dim3 gW((N + 63) / 64), gH((G + 255) / 256, nchunks);
    const int nfin = (G * KP + 255) / 256;
    const int npart = (int)(gH.x * gH.y);
    std::vector<double> hpart(npart);
    float hs[2][KP];
    auto colsum = [&](const float* M, int R, float* out) {
        const int nb = std::max(1, std::min(256, R / 256));
        mu_colsum_part_kernel<KP><<<nb, 256, 0, st>>>(M, R, dcs);
        mu_colsum_final_kernel<KP><<<1, 64, 0, st>>>(dcs, nb, out);
    };
    auto divergence = [&](double* err) -> int {
        colsum(dHt, G, dHsum);
        colsum(dW, N, dWsum);
        mu_divergence_kernel<KP, BETA1><<<gH, 256, 0, st>>>(ctx->X, ldx, N, G, dW, dHt, rpc, dpart);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(hpart.data(), dpart, (size_t)npart * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(hs[0], dHsum, KP * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(hs[1], dWsum, KP * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        double res = 0.0;
        for (double v : hpart) res += v;
        if (BETA1) { double swh = 0.0; for (int c = 0; c < KP; ++c) swh += (double)hs[0][c] * (double)hs[1][c]; res += swh; }
        else res -= (double)N * (double)G;
        *err = std::sqrt(2.0 * std::max(res, 0.0));
        return CNMF_OK;
    };
    double err0 = 0.0;
    int rc = divergence(&err0);
    if (rc) return rc;
    double prev = err0, err = err0;
    const float l1W = (float)prm->l1_reg_W, l2W = (float)prm->l2_reg_W;
    const float l1H = (float)prm->l1_reg_H, l2H = (float)prm->l2_reg_H;
    int it = 0;
    bool hsum_valid = false;
    for (it = 1; it <= prm->max_iter; ++it) {
        if (BETA1 && !hsum_valid) { colsum(dHt, G, dHsum); hsum_valid = true; }
        mu_w_kernel<KP, BETA1, (KP <= 32 ? 8 : 4)><<<gW, 64 * (KP <= 32 ? 8 : 4), 0, st>>>(ctx->X, ldx, N, G, dW, dHt, dHsum, l1W, l2W);
        if (update_H) {
            if (BETA1) colsum(dW, N, dWsum);
            mu_h_partial_kernel<KP, BETA1><<<gH, 256, 0, st>>>(ctx->X, ldx, N, G, dW, dHt, rpc, pnum, pden);
            mu_h_finish_kernel<KP, BETA1><<<nfin, 256, 0, st>>>(dHt, G, pnum, pden, nchunks, dWsum, l1H, l2H);
            hsum_valid = false;
        }
        HIP_TRY(ctx, hipGetLastError());
        if (prm->tol > 0 && it % 10 == 0) {
            rc = divergence(&err);
            if (rc) return rc;
            hsum_valid = true;                       // divergence() refreshed Hsum
            if ((prev - err) / err0 < prm->tol) break;
            prev = err;
        }
    }
    *n_iter_out = std::min(it, prm->max_iter);
    *err_out = err;
    return CNMF_OK;
}

}  // namespace cnmf

extern "C" int cnmf_nmf_mu_batch(cnmf_ctx* ctx, int n, const int32_t* kk, int init_mode,
                                 const uint32_t* seeds, const double* avg, const float* W0,
                                 const float* H0, int beta, int update_H, const cnmf_cd_params* prm,
                                 float* H_out, float* W_out, int32_t* n_iter_out, double* err_out)
{
    using namespace cnmf;
    if (!ctx) { SET_ERR(ctx, "ctx is NULL"); return CNMF_EINVAL; }
    if (!ctx->X) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }
    int rc = validate_params(ctx, prm);
    if (rc) return rc;
    if (beta != 0 && beta != 1) { SET_ERR(ctx, "beta_loss must be 1 (kullback-leibler) or 0 (itakura-saito)"); return CNMF_EUNSUPPORTED; }
    if (n < 0 || (n > 0 && !kk)) { SET_ERR(ctx, "bad restart list"); return CNMF_EINVAL; }
    if (!update_H && (!H0 || !avg)) { SET_ERR(ctx, "update_H=0 needs H0 and avg"); return CNMF_EINVAL; }
    if (update_H && init_mode == 0 && n > 0 && (!W0 || !H0)) { SET_ERR(ctx, "init_mode 0 needs W0 and H0"); return CNMF_EINVAL; }
    if (update_H && init_mode == 1 && n > 0 && (!seeds || !avg)) { SET_ERR(ctx, "init_mode 1 needs seeds and avg"); return CNMF_EINVAL; }
    if (update_H && n > 0 && !H_out) { SET_ERR(ctx, "H_out is NULL"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int N = (int)ctx->N, G = (int)ctx->G, Gp = ctx->G_pad;
    DevPool pool;
    size_t hoff = 0, woff = 0;
    for (int r = 0; r < n; ++r) {
        const int k = kk[r];
        if (k < 1) { SET_ERR(ctx, "n_components must be >= 1"); return CNMF_EINVAL; }
        if (k > KMAX) { SET_ERR(ctx, "n_components=%d > CNMF_KMAX=%d", k, KMAX); return CNMF_EUNSUPPORTED; }
        if (k > 32 && beta != 1) { SET_ERR(ctx, "itakura-saito with n_components > 32 is not supported on the device"); return CNMF_EUNSUPPORTED; }
        const int KP = k <= 8 ? 8 : (k <= 16 ? 16 : (k <= 32 ? 32 : 64));
        // row chunks of the H half-step / divergence kernels: ~8 waves per SIMD (2048 workgroups), >= 64 rows each
        const int nchunks = std::max(1, std::min(std::max(64, 2048 / std::max(1, (G + 255) / 256)), N / 64));
        const int rpc = (N + nchunks - 1) / nchunks;
        DevPool rp;                                   // per-restart scratch
        float* dW = rp.get<float>((size_t)N * KP);
        float* dHt = rp.get<float>((size_t)Gp * KP, true, st);
        float* dHsum = rp.get<float>(KP, true, st);
        float* dWsum = rp.get<float>(KP, true, st);
        float* pnum = rp.get<float>((size_t)nchunks * G * KP);
        float* pden = rp.get<float>(beta == 0 ? (size_t)nchunks * G * KP : 1);
        double* dpart = rp.get<double>((size_t)((G + 255) / 256) * nchunks);
        double* dcs = rp.get<double>((size_t)256 * 64);
        float* cmH = rp.get<float>((size_t)k * G);
        float* cmW = rp.get<float>((size_t)k * N);
        if (rp.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        const int gHt = (G * KP + 255) / 256, gWp = (int)(((size_t)N * KP + 255) / 256);
        if (!update_H) {
            HIP_TRY(ctx, hipMemcpyAsync(cmH, H0 + hoff, (size_t)k * G * sizeof(float), hipMemcpyHostToDevice, st));
            mu_pack_kernel<<<gHt, 256, 0, st>>>(cmH, k, G, dHt, KP);
            mu_fill_kernel<<<gWp, 256, 0, st>>>(dW, k, N, KP, (float)avg[r]);   // sklearn _nmf.py:1229-1231
        } else if (init_mode == 0) {
            HIP_TRY(ctx, hipMemcpyAsync(cmH, H0 + hoff, (size_t)k * G * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_TRY(ctx, hipMemcpyAsync(cmW, W0 + woff, (size_t)k * N * sizeof(float), hipMemcpyHostToDevice, st));
            mu_pack_kernel<<<gHt, 256, 0, st>>>(cmH, k, G, dHt, KP);
            mu_pack_rm_kernel<<<gWp, 256, 0, st>>>(cmW, k, N, dW, KP);
        } else {
            RngJob job{seeds[r], k, 0, avg[r], (long long)k * ((long long)G + N)};
            RngJob* dj = rp.get<RngJob>(1);
            if (rp.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
            HIP_TRY(ctx, hipMemcpy(dj, &job, sizeof job, hipMemcpyHostToDevice));
            rng_kernel<1><<<1, 256, 0, st>>>(dj, nullptr, cmH, G, G, cmW, N, N);
            mu_pack_kernel<<<gHt, 256, 0, st>>>(cmH, k, G, dHt, KP);
            mu_pack_kernel<<<gWp, 256, 0, st>>>(cmW, k, N, dW, KP);
        }
        HIP_TRY(ctx, hipGetLastError());
        int nit = 0; double err = 0.0;
#define MU_GO(KP_)                                                                                             \
        rc = (beta == 1) ? mu_run_one<KP_, true>(ctx, st, N, G, k, dW, dHt, dHsum, dWsum, pnum, pden, nchunks, rpc, \
                                                 dpart, dcs, update_H, prm, &nit, &err)                              \
                         : mu_run_one<KP_, false>(ctx, st, N, G, k, dW, dHt, dHsum, dWsum, pnum, pden, nchunks, rpc, \
                                                  dpart, dcs, update_H, prm, &nit, &err)
        if (KP == 8) { MU_GO(8); } else if (KP == 16) { MU_GO(16); } else if (KP == 32) { MU_GO(32); }
        else rc = mu_run_one<64, true>(ctx, st, N, G, k, dW, dHt, dHsum, dWsum, pnum, pden, nchunks, rpc, dpart,
                                       dcs, update_H, prm, &nit, &err);
#undef MU_GO
        if (rc) return rc;
        if (H_out && update_H) {
            mu_unpack_kernel<<<(G * k + 255) / 256, 256, 0, st>>>(dHt, k, G, KP, cmH, 1);
            HIP_TRY(ctx, hipMemcpyAsync(H_out + hoff, cmH, (size_t)k * G * sizeof(float), hipMemcpyDeviceToHost, st));
        }
        if (W_out) {
            mu_unpack_kernel<<<(int)(((size_t)N * k + 255) / 256), 256, 0, st>>>(dW, k, N, KP, cmW, 0);
            HIP_TRY(ctx, hipMemcpyAsync(W_out + woff, cmW, (size_t)k * N * sizeof(float), hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (n_iter_out) n_iter_out[r] = nit;
        if (err_out) err_out[r] = err;
        hoff += (size_t)k * G; woff += (size_t)k * N;
    }
    return CNMF_OK;
}
