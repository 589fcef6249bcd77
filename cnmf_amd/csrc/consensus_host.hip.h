// Host orchestration of the consensus step (included by cnmf_hip.hip).
//
// cnmf_consensus() restates cNMF.consensus' numerical core (cnmf.py:871-916, stats
// 922-923) on the device: L2-normalise -> all-pairs distances -> KNN local density ->
// density filter -> KMeans(k, n_init, random_state) -> per-cluster medians -> rows / sum.
// KMeans' random draws (numpy RandomState(1): one uniform for the first centre, then
// 2+int(log k) per further centre, per init; the count is data independent) are passed in
// by the caller so that seeds match scikit-learn exactly (SURVEY.md "hard parts").
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "kernels_consensus.hip.h"

namespace cnmf {

// (DevPool, the scope-bound device allocator, is defined in cnmf_hip.hip)

static inline bool same_clustering(const std::vector<int>& a, const std::vector<int>& b, int k)
{   // sklearn _k_means_common.pyx:314-330
    std::vector<int> map(k, -1);
    for (size_t i = 0; i < a.size(); ++i) {
        if (map[a[i]] == -1) map[a[i]] = b[i];
        else if (map[a[i]] != b[i]) return false;
    }
    return true;
}

// numpy RandomState.choice(n, p=ones(n)/n) for one uniform draw u:
//   cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, side='right')
static inline int first_center_index(int n, double u)
{
    std::vector<double> cdf(n);
    const double p = 1.0 / (double)n;
    double run = 0.0;
    for (int i = 0; i < n; ++i) { run += p; cdf[i] = run; }
    const double last = cdf[n - 1];
    for (int i = 0; i < n; ++i) cdf[i] /= last;
    return (int)(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
}

}  // namespace cnmf

#define CONS_TRY(call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            SET_ERR(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? CNMF_ENOMEM : CNMF_EHIP;                       \
        }                                                                                       \
    } while (0)

extern "C" int cnmf_consensus(cnmf_ctx* ctx, const double* spectra, int R, int G,
                              const cnmf_consensus_params* prm, const double* uniforms,
                              double* density_out, int32_t* keep_out, int32_t* labels_out,
                              double* median_out, double* dist_out, double* stats_out)
{
    using namespace cnmf;
    if (!ctx || !spectra || !prm || !labels_out || !median_out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    const int k = prm->k;
    if (R < 1 || G < 1 || k < 1 || k > R) { SET_ERR(ctx, "bad shape R=%d G=%d k=%d", R, G, k); return CNMF_EINVAL; }
    if (k > KM_CID) { SET_ERR(ctx, "k=%d > %d clusters is not supported", k, KM_CID); return CNMF_EUNSUPPORTED; }
    const int n_init = prm->n_init > 0 ? prm->n_init : 10;
    const int max_iter = prm->max_iter > 0 ? prm->max_iter : 300;
    const double tol = prm->tol >= 0 ? prm->tol : 1e-4;
    const int L = 2 + (int)std::log((double)k);
    if (!uniforms) { SET_ERR(ctx, "uniforms (n_init x (1+(k-1)*(2+int(log k)))) is NULL"); return CNMF_EINVAL; }
    if (L > 8) { SET_ERR(ctx, "too many local trials"); return CNMF_EUNSUPPORTED; }
    if (n_init > 64) { SET_ERR(ctx, "n_init > 64 is not supported"); return CNMF_EUNSUPPORTED; }
    if (!prm->skip_density && prm->n_neighbors < 1) { SET_ERR(ctx, "n_neighbors must be >= 1"); return CNMF_EINVAL; }
    if (!prm->skip_density && prm->n_neighbors + 1 > R) { SET_ERR(ctx, "n_neighbors+1 > number of spectra"); return CNMF_EINVAL; }
    CONS_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bool dbg = getenv("CNMF_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        hipStreamSynchronize(st);
        const double t = now();
        fprintf(stderr, "[cnmf consensus] %-28s %8.2f ms\n", what, t - t_prev);
        t_prev = t;
    };
    DevPool pool;
    const int ld = round_up(G, 16);
    const int Rp = round_up(R, 64);

    // ---- L2-normalised spectra (cnmf.py:882)
    double* dS = pool.get<double>((size_t)R * G);
    double* dL2 = pool.get<double>((size_t)Rp * ld, true, st);
    double* dsq = pool.get<double>(Rp, true, st);
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    CONS_TRY(hipMemcpyAsync(dS, spectra, (size_t)R * G * sizeof(double), hipMemcpyHostToDevice, st));
    l2_rows_kernel<<<R, 256, 0, st>>>(dS, R, G, dL2, ld, dsq);
    lap("alloc + upload + l2");

    // ---- all-pairs distances + KNN local density (cnmf.py:891-898)
    const bool need_dist = !prm->skip_density || prm->want_silhouette || dist_out;
    double* dD = nullptr;
    std::vector<double> density(R, 0.0);
    std::vector<int> keep_idx;
    if (need_dist) {
        dD = pool.get<double>((size_t)Rp * Rp);
        if (pool.err) { SET_ERR(ctx, "device allocation failed (distance matrix)"); return CNMF_ENOMEM; }
        dgemm_nt_kernel<<<dim3(Rp / 64, Rp / 64), 256, 0, st>>>(dL2, ld, dL2, ld, dD, Rp, ld);
        dist_epilogue_kernel<<<dim3((R + 255) / 256, R), 256, 0, st>>>(dD, Rp, R, dsq);
        CONS_TRY(hipGetLastError());
        if (dist_out)
            CONS_TRY(hipMemcpy2DAsync(dist_out, (size_t)R * sizeof(double), dD, (size_t)Rp * sizeof(double),
                                      (size_t)R * sizeof(double), R, hipMemcpyDeviceToHost, st));
    }
    lap("distance matrix");
    if (!prm->skip_density) {
        double* ddens = pool.get<double>(R);
        if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        const size_t lds = (size_t)R * sizeof(double);
        if (lds <= 150 * 1024 && !getenv("CNMF_KNN_GLOBAL")) {
            CONS_TRY(hipFuncSetAttribute((const void*)knn_density_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            knn_density_kernel<true><<<R, 256, lds, st>>>(dD, Rp, R, prm->n_neighbors + 1, prm->n_neighbors, ddens);
        } else {                      // more than 19 200 merged spectra: selection passes over the L2-resident row
            knn_density_kernel<false><<<R, 256, 0, st>>>(dD, Rp, R, prm->n_neighbors + 1, prm->n_neighbors, ddens);
        }
        CONS_TRY(hipGetLastError());
        CONS_TRY(hipMemcpyAsync(density.data(), ddens, (size_t)R * sizeof(double), hipMemcpyDeviceToHost, st));
        CONS_TRY(hipStreamSynchronize(st));
        for (int r = 0; r < R; ++r) if (density[r] < prm->density_threshold) keep_idx.push_back(r);   // strict <, cnmf.py:903
    } else {
        for (int r = 0; r < R; ++r) keep_idx.push_back(r);
    }
    if (density_out) memcpy(density_out, density.data(), (size_t)R * sizeof(double));
    if (keep_out) { for (int r = 0; r < R; ++r) keep_out[r] = 0; for (int r : keep_idx) keep_out[r] = 1; }
    for (int r = 0; r < R; ++r) labels_out[r] = -1;
    const int Rk = (int)keep_idx.size();
    if (stats_out) { stats_out[0] = Rk; stats_out[1] = stats_out[2] = stats_out[3] = 0.0; }
    if (Rk == 0) { SET_ERR(ctx, "Zero components remain after density filtering. Consider increasing density threshold"); return CNMF_ESTATE; }
    if (Rk < k) { SET_ERR(ctx, "n_samples=%d should be >= n_clusters=%d.", Rk, k); return CNMF_EINVAL; }

    lap("knn density + filter");
    // ---- KMeans on the kept rows (cnmf.py:908-909; sklearn _kmeans.py:1427-1555), all inits batched
    const int Rkp = round_up(Rk, 64);
    const int I = n_init;
    const int MT = round_up(I * k, 64), MC = round_up(I * L, 64), MD = std::max(MT, MC);
    const size_t ustride = 1 + (size_t)(k - 1) * L;
    const KmDims kd{Rk, Rkp, G, ld, k, L};
    int* dkeep = pool.get<int>(Rk);
    double* dX = pool.get<double>((size_t)Rkp * ld, true, st);
    double* dxsq = pool.get<double>(Rkp, true, st);
    double* dmean = pool.get<double>(ld, true, st);
    double* dvar = pool.get<double>(ld, true, st);
    double* dcA = pool.get<double>((size_t)MT * ld, true, st);
    double* dcB = pool.get<double>((size_t)MT * ld, true, st);
    double* dcand = pool.get<double>((size_t)MC * ld, true, st);
    double* ddots = pool.get<double>((size_t)MD * Rkp, true, st);
    double* dclosest = pool.get<double>((size_t)I * Rkp);
    double* dcum = pool.get<double>((size_t)I * Rkp);
    double* ddmin = pool.get<double>((size_t)I * 8 * Rkp);
    double* dcpot = pool.get<double>((size_t)I * 8);
    double* dcsq = pool.get<double>(MT);
    int* dlabels = pool.get<int>((size_t)I * Rkp);
    const int rows_per_chunk = std::max(64, (Rk + 15) / 16);
    const int nchunks = (Rk + rows_per_chunk - 1) / rows_per_chunk;
    double* dpartial = pool.get<double>((size_t)I * nchunks * k * ld);
    int* dpcount = pool.get<int>((size_t)I * nchunks * k);
    double* dsums = pool.get<double>((size_t)I * k * ld);
    int* dcounts = pool.get<int>((size_t)I * k);
    double* ddist = pool.get<double>((size_t)I * Rkp);
    int* dcids = pool.get<int>((size_t)I * KM_CID);
    int* dc0 = pool.get<int>(I);
    int* dneed = pool.get<int>(I);
    KmState* dst = pool.get<KmState>(I, true, st);
    const size_t nu = (size_t)I * ustride;
    double* du = pool.get<double>(nu);
    KmState* hst = nullptr;
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    CONS_TRY(hipHostMalloc(&hst, sizeof(KmState) * I));
    struct HostFree { void* p; ~HostFree() { hipHostFree(p); } } hf{hst};
    CONS_TRY(hipMemcpyAsync(dkeep, keep_idx.data(), (size_t)Rk * sizeof(int), hipMemcpyHostToDevice, st));
    CONS_TRY(hipMemcpyAsync(du, uniforms, nu * sizeof(double), hipMemcpyHostToDevice, st));
    gather_rows_kernel<<<dim3((G + 255) / 256, Rk), 256, 0, st>>>(dL2, ld, dkeep, Rk, G, dX, ld);
    {   // column mean / population variance of the kept rows (sklearn _kmeans.py:1477-1484, 279-288), rows spread
        // over 64-row chunks, chunk partials added in order
        const int rpc = 64, chunks = (Rk + rpc - 1) / rpc;
        double* dcpart = pool.get<double>((size_t)chunks * G);
        if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        dim3 gp((G + 255) / 256, chunks);
        col_partial_f64_kernel<<<gp, 256, 0, st>>>(dX, ld, Rk, G, rpc, nullptr, dcpart);
        col_combine_f64_kernel<<<(G + 255) / 256, 256, 0, st>>>(dcpart, chunks, G, (double)Rk, dmean);
        col_partial_f64_kernel<<<gp, 256, 0, st>>>(dX, ld, Rk, G, rpc, dmean, dcpart);
        col_combine_f64_kernel<<<(G + 255) / 256, 256, 0, st>>>(dcpart, chunks, G, (double)Rk, dvar);
    }
    center_rows_kernel<<<Rk, 256, 0, st>>>(dX, ld, G, dmean, dxsq);
    std::vector<double> hvar(G);
    CONS_TRY(hipMemcpyAsync(hvar.data(), dvar, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, st));
    CONS_TRY(hipStreamSynchronize(st));
    double vsum = 0.0;
    for (int g = 0; g < G; ++g) vsum += hvar[g];
    const double tol_ = (vsum / G) * tol;                      // _tolerance, sklearn _kmeans.py:279-288

    const int acc_bw = k <= 64 ? 256 : 128;                      // columns per block of the M step: k x acc_bw doubles of LDS
    const size_t acc_lds = (size_t)k * acc_bw * sizeof(double);
    CONS_TRY(dyn_lds_optin((const void*)accumulate_kernel, 160 * 1024 - 64));
    auto dots = [&](const double* A, int M) {     // ddots[M][Rkp] = A[M][ld] . X^T  (one product for every init)
        dgemm_nt_small_kernel<<<dim3(Rkp / 16, M / 64), 256, 0, st>>>(A, ld, dX, ld, ddots, Rkp, ld);
    };
    lap("kmeans setup (gather, stats)");

    // -- k-means++ for all inits in lock step (sklearn _kmeans.py:174-272)
    std::vector<int> c0(I);
    for (int i = 0; i < I; ++i) c0[i] = std::min(first_center_index(Rk, uniforms[(size_t)i * ustride]), Rk - 1);
    CONS_TRY(hipMemcpyAsync(dc0, c0.data(), (size_t)I * sizeof(int), hipMemcpyHostToDevice, st));
    pp_seed_kernel<<<dim3((ld + 255) / 256, I), 256, 0, st>>>(dX, kd, dc0, dcA, dcids);
    dots(dcA, MT);
    pp_first_kernel<<<I, 256, 0, st>>>(ddots, kd, dxsq, dc0, dclosest, dst);
    for (int c = 1; c < k; ++c) {
        pp_candidates_kernel<<<I, 256, 0, st>>>(dclosest, kd, du, (int)ustride, 1 + (c - 1) * L, dcum, dst);
        pp_gather_kernel<<<dim3((G + 255) / 256, L, I), 256, 0, st>>>(dX, kd, dst, dcand);
        dots(dcand, MC);
        pp_update_kernel<<<dim3(L, I), 256, 0, st>>>(ddots, kd, dxsq, dclosest, dst, ddmin, dcpot);
        pp_pick_kernel<<<I, 256, 0, st>>>(dcpot, kd, ddmin, dclosest, dst, dX, dcA, c, dcids);
    }
    CONS_TRY(hipGetLastError());
    lap("kmeans++ (all inits)");

    // -- Lloyd for all inits in lock step; an init leaves the loop at its own iteration (sklearn _kmeans.py:624-752)
    double* cur = dcA; double* nxt = dcB;
    CONS_TRY(hipMemsetAsync(dlabels, 0xff, (size_t)I * Rkp * sizeof(int), st));          // labels = -1
    std::vector<char> done(I, 0), strict(I, 0);
    std::vector<int> iters(I, 0);
    int n_done = 0;
    for (int it = 0; it < max_iter && n_done < I; ++it) {
        center_norms_kernel<<<I * k, 256, 0, st>>>(cur, ld, G, dcsq);
        dots(cur, MT);
        km_reset_kernel<<<1, 64, 0, st>>>(dst, I);
        assign_kernel<<<dim3((Rk + 255) / 256, I), 256, 0, st>>>(ddots, kd, dcsq, dlabels, dst, 0, nullptr);
        accumulate_kernel<<<dim3((G + acc_bw - 1) / acc_bw, nchunks, I), acc_bw, acc_lds, st>>>(dX, kd, dlabels, rows_per_chunk, nchunks, dpartial, dpcount, dst);
        reduce_partial_kernel<<<dim3((G + 255) / 256, k, I), 256, 0, st>>>(dpartial, dpcount, nchunks, kd, dsums, dcounts, dst);
        row_center_dist_kernel<<<dim3(Rk, I), 256, 0, st>>>(dX, kd, cur, dlabels, ddist, dst, 0);
        relocate_empty_kernel<<<I, 256, 0, st>>>(dX, kd, dlabels, ddist, dsums, dcounts, dst);
        finish_centers_kernel<<<I, 256, 0, st>>>(dsums, dcounts, kd, cur, nxt, dst);
        CONS_TRY(hipGetLastError());
        CONS_TRY(hipMemcpyAsync(hst, dst, sizeof(KmState) * I, hipMemcpyDeviceToHost, st));
        CONS_TRY(hipStreamSynchronize(st));
        std::swap(cur, nxt);
        bool any_new = false;
        for (int i = 0; i < I; ++i) {
            if (done[i]) continue;
            iters[i] = it + 1;
            if (hst[i].changed == 0) { strict[i] = 1; done[i] = 1; }
            else if (hst[i].shift_tot <= tol_) done[i] = 1;
            if (done[i]) { hst[i].done = 1; ++n_done; any_new = true; }
        }
        if (any_new)      // publish the done flags (only that field changes; the device copy is otherwise current)
            for (int i = 0; i < I; ++i)
                if (done[i]) CONS_TRY(hipMemcpyAsync(&dst[i].done, &hst[i].done, sizeof(int), hipMemcpyHostToDevice, st));
    }
    // inits that stopped on the tolerance (or max_iter) re-run the E step so labels match the final centres
    std::vector<int> need(I, 0);
    bool any_need = false;
    for (int i = 0; i < I; ++i) { need[i] = strict[i] ? 0 : 1; any_need |= need[i] != 0; }
    if (any_need) {
        CONS_TRY(hipMemcpyAsync(dneed, need.data(), (size_t)I * sizeof(int), hipMemcpyHostToDevice, st));
        center_norms_kernel<<<I * k, 256, 0, st>>>(cur, ld, G, dcsq);
        dots(cur, MT);
        assign_kernel<<<dim3((Rk + 255) / 256, I), 256, 0, st>>>(ddots, kd, dcsq, dlabels, dst, 1, dneed);
    }
    row_center_dist_kernel<<<dim3(Rk, I), 256, 0, st>>>(dX, kd, cur, dlabels, ddist, dst, 1);
    inertia_kernel<<<I, 256, 0, st>>>(ddist, kd, dst);
    CONS_TRY(hipGetLastError());
    std::vector<int> all_labels((size_t)I * Rkp);
    CONS_TRY(hipMemcpyAsync(hst, dst, sizeof(KmState) * I, hipMemcpyDeviceToHost, st));
    CONS_TRY(hipMemcpyAsync(all_labels.data(), dlabels, all_labels.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    CONS_TRY(hipStreamSynchronize(st));
    // best-of-n_init in init order (sklearn _kmeans.py:1525-1533)
    std::vector<int> best_labels, labels(Rk);
    double best_inertia = 0.0; int best_iter = 0; bool have_best = false;
    for (int i = 0; i < I; ++i) {
        std::copy(all_labels.begin() + (size_t)i * Rkp, all_labels.begin() + (size_t)i * Rkp + Rk, labels.begin());
        const double inertia = hst[i].inertia;
        if (!have_best || (inertia < best_inertia && !same_clustering(labels, best_labels, k))) {
            best_labels = labels; best_inertia = inertia; best_iter = std::min(iters[i], max_iter); have_best = true;
        }
    }
    lap("kmeans (all inits)");
    for (int q = 0; q < Rk; ++q) labels_out[keep_idx[q]] = best_labels[q];

    // ---- per-cluster per-gene median, rows normalised to sum 1 (cnmf.py:913-916)
    std::vector<int> seg(k + 1, 0), order(Rk), order_rows(Rk);
    for (int q = 0; q < Rk; ++q) seg[best_labels[q] + 1]++;
    for (int j = 0; j < k; ++j) seg[j + 1] += seg[j];
    { std::vector<int> pos(seg.begin(), seg.end() - 1);
      for (int q = 0; q < Rk; ++q) { const int p = pos[best_labels[q]]++; order[p] = q; order_rows[p] = keep_idx[q]; } }
    for (int j = 0; j < k; ++j)
        if (seg[j + 1] == seg[j]) { SET_ERR(ctx, "k-means produced an empty cluster (%d)", j); return CNMF_ESTATE; }
    int* dorder_rows = pool.get<int>(Rk);
    int* dorder = pool.get<int>(Rk);
    int* dseg = pool.get<int>(k + 1);
    double* dmed = pool.get<double>((size_t)k * G);
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    CONS_TRY(hipMemcpyAsync(dorder_rows, order_rows.data(), (size_t)Rk * sizeof(int), hipMemcpyHostToDevice, st));
    CONS_TRY(hipMemcpyAsync(dorder, order.data(), (size_t)Rk * sizeof(int), hipMemcpyHostToDevice, st));
    CONS_TRY(hipMemcpyAsync(dseg, seg.data(), (size_t)(k + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    int max_m = 0;
    for (int j = 0; j < k; ++j) max_m = std::max(max_m, seg[j + 1] - seg[j]);
    if (max_m <= 512)
        cluster_median_kernel<8><<<dim3((G + 3) / 4, k), 256, 0, st>>>(dL2, ld, G, dorder_rows, dseg, dmed);
    else if (max_m <= 2048)
        cluster_median_kernel<32><<<dim3((G + 3) / 4, k), 256, 0, st>>>(dL2, ld, G, dorder_rows, dseg, dmed);
    else
        cluster_median_big_kernel<<<dim3((G + 63) / 64, k), 64, 0, st>>>(dL2, ld, G, dorder_rows, dseg, dmed);
    normalise_rows_sum_kernel<<<k, 256, 0, st>>>(dmed, G);
    CONS_TRY(hipGetLastError());
    CONS_TRY(hipMemcpyAsync(median_out, dmed, (size_t)k * G * sizeof(double), hipMemcpyDeviceToHost, st));

    lap("medians");
    // ---- silhouette (cnmf.py:923)
    double sil = 0.0;
    if (prm->want_silhouette) {
        double* dsil = pool.get<double>(Rk);
        double* dsum = pool.get<double>(1);
        if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        CONS_TRY(hipMemcpyAsync(dlabels, best_labels.data(), (size_t)Rk * sizeof(int), hipMemcpyHostToDevice, st));
        // sample order = kept order q; silhouette_kernel wants sample index in kept space
        silhouette_kernel<<<Rk, 256, 0, st>>>(dD, Rp, dkeep, dorder, dseg, dlabels, Rk, k, dsil);
        sum_kernel<<<1, 256, 0, st>>>(dsil, Rk, dsum);
        CONS_TRY(hipGetLastError());
        CONS_TRY(hipMemcpyAsync(&sil, dsum, sizeof(double), hipMemcpyDeviceToHost, st));
    }
    CONS_TRY(hipStreamSynchronize(st));
    if (stats_out) { stats_out[0] = Rk; stats_out[1] = best_inertia; stats_out[2] = sil / Rk; stats_out[3] = best_iter; }
    return CNMF_OK;
}

// Replaces the dense residual of cnmf.py:926-930: sum((X - W.H)^2) with X the resident matrix.
extern "C" int cnmf_prediction_error(cnmf_ctx* ctx, int k, const double* W, const double* H, double* err_out)
{
    using namespace cnmf;
    if (!ctx || !W || !H || !err_out || k < 1) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    if (!ctx->X) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }
    if (k > KMAX) { SET_ERR(ctx, "k=%d > %d", k, KMAX); return CNMF_EUNSUPPORTED; }
    CONS_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int N = (int)ctx->N, G = (int)ctx->G;
    DevPool pool;
    double* dW = pool.get<double>((size_t)N * k);
    double* dH = pool.get<double>((size_t)k * G);
    const int rpb = 512;
    const int bw = k <= 64 ? 256 : 128;                     // genes per block: the block's H columns sit in LDS (k x bw doubles)
    dim3 grid((G + bw - 1) / bw, (N + rpb - 1) / rpb);
    double* dpart = pool.get<double>((size_t)grid.x * grid.y);
    double* dsum = pool.get<double>(1);
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    CONS_TRY(hipMemcpyAsync(dW, W, (size_t)N * k * sizeof(double), hipMemcpyHostToDevice, st));
    CONS_TRY(hipMemcpyAsync(dH, H, (size_t)k * G * sizeof(double), hipMemcpyHostToDevice, st));
    const size_t lds = (size_t)k * bw * sizeof(double);
    CONS_TRY(dyn_lds_optin((const void*)residual_sq_kernel, 160 * 1024 - 64));
    residual_sq_kernel<<<grid, bw, lds, st>>>(ctx->X, ctx->G_pad, N, G, dW, dH, k, rpb, dpart);
    sum_kernel<<<1, 256, 0, st>>>(dpart, (int)(grid.x * grid.y), dsum);
    CONS_TRY(hipGetLastError());
    CONS_TRY(hipMemcpyAsync(err_out, dsum, sizeof(double), hipMemcpyDeviceToHost, st));
    CONS_TRY(hipStreamSynchronize(st));
    return CNMF_OK;
}
