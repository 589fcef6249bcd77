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

// `spectra` (host, float64 [R][G]) or -- round 4 -- `store_rows` [R]: row indices into the context's RESIDENT spectra store
// (float32, filled by cnmf_nmf_cd_batch_resident): the merged spectra of a k are gathered and widened on the device, no
// upload (80 MB of float64 at 5 000 x 2 000: 1.5 of the 4.4 ms of the round-3 call).
static int consensus_impl(cnmf_ctx* ctx, const double* spectra, const int64_t* store_rows, int R, int G,
                          const cnmf_consensus_params* prm, const double* uniforms,
                          double* density_out, int32_t* keep_out, int32_t* labels_out,
                          double* median_out, double* dist_out, double* stats_out)
{
    using namespace cnmf;
    if (!ctx || (!spectra && !store_rows) || !prm || !labels_out || !median_out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (store_rows) {
        if ((int64_t)G != ctx->spectra_G) { SET_ERR(ctx, "the resident store holds spectra over %lld genes, not %d", (long long)ctx->spectra_G, G); return CNMF_EINVAL; }
        for (int r = 0; r < R; ++r)
            if (store_rows[r] < 0 || (size_t)store_rows[r] >= ctx->spectra_rows) { SET_ERR(ctx, "store row %lld outside 0..%lld", (long long)store_rows[r], (long long)ctx->spectra_rows - 1); return CNMF_EINVAL; }
    }
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
    const bool dbg = ctx_getenv(ctx, "CNMF_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        hipStreamSynchronize(st);
        const double t = now();
        fprintf(stderr, "[cnmf consensus] %-28s %8.2f ms\n", what, t - t_prev);
        t_prev = t;
    };
    Arena& pool = ctx->cons_ws;                      // blocks stay with the context: no hipMalloc / hipFree per call
    pool.reset();
    struct Trim { Arena& a; ~Trim() { if (a.total() > Arena::keep_limit) a.release(); } } trim{pool};
    const int ld = round_up(G, 16);
    const int Rp = round_up(R, 64);

    // ---- L2-normalised spectra (cnmf.py:882)
    double* dS = pool.get<double>((size_t)R * G);
    double* dL2 = pool.get<double>((size_t)Rp * ld, true, st);
    double* dsq = pool.get<double>(Rp, true, st);
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    if (store_rows) {
        long long* drows = pool.get<long long>(R);
        if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        CONS_TRY(hipMemcpyAsync(drows, store_rows, (size_t)R * sizeof(long long), hipMemcpyHostToDevice, st));
        gather_store_rows_kernel<<<dim3((G + 255) / 256, R), 256, 0, st>>>(ctx->spectra, G, drows, dS);
    } else {
        CONS_TRY(hipMemcpyAsync(dS, spectra, (size_t)R * G * sizeof(double), hipMemcpyHostToDevice, st));
    }
    l2_rows_kernel<<<R, 256, 0, st>>>(dS, R, G, dL2, ld, dsq);
    lap("alloc + upload + l2");

    // ---- all-pairs distances + KNN local density (cnmf.py:891-898)
    const bool need_dist = true;      // k-means++ reads its squared distances from this matrix (pp_fused_kernel)
    double* dD = nullptr;
    std::vector<double> density(R, 0.0);
    std::vector<int> keep_idx;
    if (need_dist) {
        dD = pool.get<double>((size_t)Rp * Rp);
        if (pool.err) { SET_ERR(ctx, "device allocation failed (distance matrix)"); return CNMF_ENOMEM; }
        const int nt = Rp / 64;                      // lower-triangular tiles; the epilogue writes both halves
        dist_sym_kernel<<<nt * (nt + 1) / 2, 256, 0, st>>>(dL2, ld, ld, dsq, R, dD, Rp);
        CONS_TRY(hipGetLastError());
        if (dist_out)
            CONS_TRY(hipMemcpy2DAsync(dist_out, (size_t)R * sizeof(double), dD, (size_t)Rp * sizeof(double),
                                      (size_t)R * sizeof(double), R, hipMemcpyDeviceToHost, st));
    }
    lap("distance matrix");
    if (!prm->skip_density) {
        double* ddens = pool.get<double>(R);
        if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        const int m_sel = prm->n_neighbors + 1, vpt = (R + 255) / 256;
#define KNN_REG(V) knn_density_reg_kernel<V><<<R, 256, 0, st>>>(dD, Rp, R, m_sel, prm->n_neighbors, ddens)
        if (vpt <= 80 && !ctx_getenv(ctx, "CNMF_KNN_GLOBAL")) {   // the row in registers, 4-way search (up to 20 480 merged spectra)
            if (vpt <= 4) KNN_REG(4); else if (vpt <= 8) KNN_REG(8); else if (vpt <= 16) KNN_REG(16);
            else if (vpt <= 24) KNN_REG(24); else if (vpt <= 40) KNN_REG(40); else KNN_REG(80);
        } else {                      // selection passes over the L2-resident row: any R
            knn_density_kernel<<<R, 256, 0, st>>>(dD, Rp, R, m_sel, prm->n_neighbors, ddens);
        }
#undef KNN_REG
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
    const int MT = round_up(I * k, 64);
    const size_t ustride = 1 + (size_t)(k - 1) * L;
    const KmDims kd{Rk, Rkp, G, ld, k, L};
    int* dkeep = pool.get<int>(Rk);
    double* dX = pool.get<double>((size_t)Rkp * ld, true, st);
    double* dxsq = pool.get<double>(Rkp, true, st);
    double* dmean = pool.get<double>(ld, true, st);
    double* dvar = pool.get<double>(ld, true, st);
    double* dtol = pool.get<double>(1);
    double* dcA = pool.get<double>((size_t)MT * ld, true, st);
    double* dcB = pool.get<double>((size_t)MT * ld, true, st);
    double* ddots = pool.get<double>((size_t)MT * Rkp, true, st);
    double* dcsq = pool.get<double>(MT);
    double* dshift2 = pool.get<double>((size_t)I * KM_CID);
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
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    // pinned host block (kept by the context): k-means state | labels of every init | first-centre ids / need flags |
    // the three index arrays of the median step
    const size_t pin_state = round_up((int64_t)sizeof(KmState) * I, 256), pin_lab = (size_t)I * Rkp * sizeof(int);
    const size_t pin_ints = (size_t)(2 * I + 2 * Rk + k + 1) * sizeof(int);
    const size_t pin_need = pin_state + pin_lab + round_up((int64_t)pin_ints, 256);
    if (ctx->cons_pinned_bytes < pin_need) {
        if (ctx->cons_pinned) { hipHostFree(ctx->cons_pinned); ctx->cons_pinned = nullptr; ctx->cons_pinned_bytes = 0; }
        CONS_TRY(hipHostMalloc(&ctx->cons_pinned, pin_need + pin_need / 2));
        ctx->cons_pinned_bytes = pin_need + pin_need / 2;
    }
    KmState* hst = (KmState*)ctx->cons_pinned;
    int* h_labels = (int*)((char*)ctx->cons_pinned + pin_state);
    int* h_ints = (int*)((char*)ctx->cons_pinned + pin_state + pin_lab);
    CONS_TRY(hipMemcpyAsync(dkeep, keep_idx.data(), (size_t)Rk * sizeof(int), hipMemcpyHostToDevice, st));
    CONS_TRY(hipMemcpyAsync(du, uniforms, nu * sizeof(double), hipMemcpyHostToDevice, st));
    gather_rows_kernel<<<dim3((G + 255) / 256, Rk), 256, 0, st>>>(dL2, ld, dkeep, Rk, G, dX, ld);
    {   // column mean / population variance of the kept rows (sklearn _kmeans.py:1477-1484, 279-288), rows spread
        // over 64-row chunks, chunk partials added in order; the tolerance stays on the device (lloyd_decide_kernel)
        const int rpc = 64, chunks = (Rk + rpc - 1) / rpc;
        double* dcpart = pool.get<double>((size_t)chunks * G);
        if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        dim3 gp((G + 255) / 256, chunks);
        col_partial_f64_kernel<<<gp, 256, 0, st>>>(dX, ld, Rk, G, rpc, nullptr, dcpart);
        col_combine_f64_kernel<<<(G + 255) / 256, 256, 0, st>>>(dcpart, chunks, G, (double)Rk, dmean);
        col_partial_f64_kernel<<<gp, 256, 0, st>>>(dX, ld, Rk, G, rpc, dmean, dcpart);
        col_combine_f64_kernel<<<(G + 255) / 256, 256, 0, st>>>(dcpart, chunks, G, (double)Rk, dvar);
        km_tolerance_kernel<<<1, 256, 0, st>>>(dvar, G, tol, dtol);
    }
    center_rows_kernel<<<Rk, 256, 0, st>>>(dX, ld, G, dmean, dxsq);

    const int acc_bw = k <= 64 ? 256 : 128;                      // columns per block of the M step: k x acc_bw doubles of LDS
    const size_t acc_lds = (size_t)k * acc_bw * sizeof(double);
    CONS_TRY(dyn_lds_optin((const void*)accumulate_kernel, 160 * 1024 - 64));
    auto dots = [&](const double* A, const KmState* live) {   // ddots[MT][Rkp] = A[MT][ld] . X^T  (one product for every init)
        dgemm_nt_small_kernel<<<dim3(Rkp / 16, MT / 64), 256, 0, st>>>(A, ld, dX, ld, ddots, Rkp, ld, live, k, I);
    };
    lap("kmeans setup (gather, stats)");

    // -- k-means++: one workgroup per init runs all k draws on the distance matrix (sklearn _kmeans.py:174-272)
    int* c0 = h_ints;
    for (int i = 0; i < I; ++i) c0[i] = std::min(first_center_index(Rk, uniforms[(size_t)i * ustride]), Rk - 1);
    CONS_TRY(hipMemcpyAsync(dc0, c0, (size_t)I * sizeof(int), hipMemcpyHostToDevice, st));
    const int pp_vpt = (Rk + 1023) / 1024;
#define PP_REG(V) pp_fused_reg_kernel<1024, V><<<I, 1024, 0, st>>>(dD, Rp, dkeep, kd, dc0, du, (int)ustride, dcids)
    if (pp_vpt <= 8 && !ctx_getenv(ctx, "CNMF_PP_GLOBAL")) {       // closest[] in registers: up to 8192 kept spectra
        if (pp_vpt <= 1) PP_REG(1); else if (pp_vpt <= 2) PP_REG(2); else if (pp_vpt <= 4) PP_REG(4); else PP_REG(8);
    } else {
        double* dclosest = pool.get<double>((size_t)I * Rkp);
        double* dcum = pool.get<double>((size_t)I * Rkp);
        double* ddmin = pool.get<double>((size_t)I * 8 * Rkp);
        if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        pp_fused_kernel<1024><<<I, 1024, 0, st>>>(dD, Rp, dkeep, kd, dc0, du, (int)ustride, dclosest, dcum, ddmin, dcids);
    }
#undef PP_REG
    pp_centers_kernel<<<dim3((ld + 255) / 256, k, I), 256, 0, st>>>(dX, kd, dcids, dcA);
    CONS_TRY(hipGetLastError());
    lap("kmeans++ (all inits)");

    // -- Lloyd for all inits in lock step; an init leaves the loop at its own iteration (sklearn _kmeans.py:624-752).
    // The stopping rule is applied on the device, so iterations are queued LLOYD_BATCH at a time and the host looks at
    // the state once per batch (kernels of a finished init return at once; its tiles of the product are skipped).
    double* cur = dcA; double* nxt = dcB;
    CONS_TRY(hipMemsetAsync(dlabels, 0xff, (size_t)I * Rkp * sizeof(int), st));          // labels = -1
    constexpr int LLOYD_BATCH = 3;
    for (int it = 0; it < max_iter; ) {
        const int nb = std::min(LLOYD_BATCH, max_iter - it);
        for (int b = 0; b < nb; ++b, ++it) {
            center_norms_kernel<<<I * k, 256, 0, st>>>(cur, ld, G, dcsq);
            dots(cur, dst);
            assign_kernel<<<dim3((Rk + 255) / 256, I), 256, 0, st>>>(ddots, kd, dcsq, dlabels, dst, 0, nullptr);
            accumulate_kernel<<<dim3((G + acc_bw - 1) / acc_bw, nchunks, I), acc_bw, acc_lds, st>>>(dX, kd, dlabels, rows_per_chunk, nchunks, dpartial, dpcount, dst);
            reduce_partial_kernel<<<dim3((G + 255) / 256, k, I), 256, 0, st>>>(dpartial, dpcount, nchunks, kd, dsums, dcounts, dst);
            row_center_dist_kernel<<<dim3((Rk + RCD_ROWS - 1) / RCD_ROWS, I), 256, 0, st>>>(dX, kd, cur, dlabels, ddist, dst);
            relocate_empty_kernel<<<I, 256, 0, st>>>(dX, kd, dlabels, ddist, dsums, dcounts, dst);
            finish_centers_kernel<<<dim3(k, I), 256, 0, st>>>(dsums, dcounts, kd, cur, nxt, dst, dshift2);
            lloyd_decide_kernel<<<1, 64, 0, st>>>(dst, dshift2, k, I, dtol, it);
            std::swap(cur, nxt);
        }
        CONS_TRY(hipGetLastError());
        CONS_TRY(hipMemcpyAsync(hst, dst, sizeof(KmState) * I, hipMemcpyDeviceToHost, st));
        CONS_TRY(hipStreamSynchronize(st));
        bool all_done = true;
        for (int i = 0; i < I; ++i) all_done &= hst[i].done != 0;
        if (all_done) break;
    }
    // inits that stopped on the tolerance (or max_iter) re-run the E step so labels match the final centres
    std::vector<int> iters(I, 0);
    int* need = h_ints + I;
    bool any_need = false;
    for (int i = 0; i < I; ++i) { iters[i] = hst[i].iters; need[i] = hst[i].strict ? 0 : 1; any_need |= need[i] != 0; }
    if (any_need) {
        CONS_TRY(hipMemcpyAsync(dneed, need, (size_t)I * sizeof(int), hipMemcpyHostToDevice, st));
        center_norms_kernel<<<I * k, 256, 0, st>>>(cur, ld, G, dcsq);
        dots(cur, nullptr);
        assign_kernel<<<dim3((Rk + 255) / 256, I), 256, 0, st>>>(ddots, kd, dcsq, dlabels, dst, 1, dneed);
    }
    row_center_dist_all_kernel<<<Rk, 256, 0, st>>>(dX, kd, cur, dlabels, ddist, I);
    inertia_kernel<<<I, 256, 0, st>>>(ddist, kd, dst);
    CONS_TRY(hipGetLastError());
    const int* all_labels = h_labels;
    CONS_TRY(hipMemcpyAsync(hst, dst, sizeof(KmState) * I, hipMemcpyDeviceToHost, st));
    CONS_TRY(hipMemcpyAsync(h_labels, dlabels, (size_t)I * Rkp * sizeof(int), hipMemcpyDeviceToHost, st));
    CONS_TRY(hipStreamSynchronize(st));
    // best-of-n_init in init order (sklearn _kmeans.py:1525-1533)
    std::vector<int> best_labels, labels(Rk);
    double best_inertia = 0.0; int best_iter = 0; bool have_best = false;
    for (int i = 0; i < I; ++i) {
        std::copy(all_labels + (size_t)i * Rkp, all_labels + (size_t)i * Rkp + Rk, labels.begin());
        const double inertia = hst[i].inertia;
        if (!have_best || (inertia < best_inertia && !same_clustering(labels, best_labels, k))) {
            best_labels = labels; best_inertia = inertia; best_iter = std::min(iters[i], max_iter); have_best = true;
        }
    }
    lap("kmeans (all inits)");
    for (int q = 0; q < Rk; ++q) labels_out[keep_idx[q]] = best_labels[q];

    // ---- per-cluster per-gene median, rows normalised to sum 1 (cnmf.py:913-916)
    int* order_rows = h_ints + 2 * I;            // pinned, one upload: order_rows | order | seg
    int* order = order_rows + Rk;
    int* seg = order + Rk;
    for (int j = 0; j <= k; ++j) seg[j] = 0;
    for (int q = 0; q < Rk; ++q) seg[best_labels[q] + 1]++;
    for (int j = 0; j < k; ++j) seg[j + 1] += seg[j];
    { std::vector<int> pos(seg, seg + k);
      for (int q = 0; q < Rk; ++q) { const int p = pos[best_labels[q]]++; order[p] = q; order_rows[p] = keep_idx[q]; } }
    for (int j = 0; j < k; ++j)
        if (seg[j + 1] == seg[j]) { SET_ERR(ctx, "k-means produced an empty cluster (%d)", j); return CNMF_ESTATE; }
    int* dorder_rows = pool.get<int>((size_t)2 * Rk + k + 1);
    int* dorder = dorder_rows + Rk;
    int* dseg = dorder + Rk;
    double* dmed = pool.get<double>((size_t)k * G);
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    CONS_TRY(hipMemcpyAsync(dorder_rows, order_rows, (size_t)(2 * Rk + k + 1) * sizeof(int), hipMemcpyHostToDevice, st));
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

extern "C" int cnmf_consensus(cnmf_ctx* ctx, const double* spectra, int R, int G,
                              const cnmf_consensus_params* prm, const double* uniforms,
                              double* density_out, int32_t* keep_out, int32_t* labels_out,
                              double* median_out, double* dist_out, double* stats_out)
{
    if (!spectra) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    return consensus_impl(ctx, spectra, nullptr, R, G, prm, uniforms, density_out, keep_out, labels_out, median_out, dist_out, stats_out);
}

extern "C" int cnmf_consensus_store(cnmf_ctx* ctx, const int64_t* store_rows, int R, int G,
                                    const cnmf_consensus_params* prm, const double* uniforms,
                                    double* density_out, int32_t* keep_out, int32_t* labels_out,
                                    double* median_out, double* dist_out, double* stats_out)
{
    if (!store_rows) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    return consensus_impl(ctx, nullptr, store_rows, R, G, prm, uniforms, density_out, keep_out, labels_out, median_out, dist_out, stats_out);
}

// All-pairs Euclidean distances of the rows AS GIVEN (sklearn.metrics.euclidean_distances(X), cnmf.py:891 / :988) and /
// or the mean silhouette coefficient of a labelling of them (sklearn.metrics.silhouette_score(X, labels,
// metric='euclidean'), cnmf.py:923) in float64 -- the two scikit-learn calls of the reference's consensus body that are
// not k-means, as entry points of their own (INTEGRATION.md Option B: no consensus run behind a distance matrix, no
// scikit-learn behind a silhouette).  dist_out (nullable) [R][R]; labels (nullable) [R] with values 0..k-1 and
// silhouette_out (nullable) go together.
extern "C" int cnmf_pairwise_distances(cnmf_ctx* ctx, const double* rows, int R, int G, const int32_t* labels, int k,
                                       double* dist_out, double* silhouette_out)
{
    using namespace cnmf;
    if (!ctx || !rows || R < 1 || G < 1 || (!dist_out && !silhouette_out)) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    if (silhouette_out && (!labels || k < 2 || k > KM_CID || k >= R)) {
        SET_ERR(ctx, "Number of labels is %d. Valid values are 2 to n_samples - 1 (inclusive)", k); return CNMF_EINVAL;
    }
    CONS_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Arena& pool = ctx->cons_ws;
    pool.reset();
    struct Trim { Arena& a; ~Trim() { if (a.total() > Arena::keep_limit) a.release(); } } trim{pool};
    const int ld = round_up(G, 16), Rp = round_up(R, 64);
    double* dS = pool.get<double>((size_t)R * G);
    double* dL2 = pool.get<double>((size_t)Rp * ld, true, st);
    double* dsq = pool.get<double>(Rp, true, st);
    double* dD = pool.get<double>((size_t)Rp * Rp);
    if (pool.err) { SET_ERR(ctx, "device allocation failed (distance matrix)"); return CNMF_ENOMEM; }
    CONS_TRY(hipMemcpyAsync(dS, rows, (size_t)R * G * sizeof(double), hipMemcpyHostToDevice, st));
    copy_rows_sq_kernel<<<R, 256, 0, st>>>(dS, R, G, dL2, ld, dsq);
    const int nt = Rp / 64;
    dist_sym_kernel<<<nt * (nt + 1) / 2, 256, 0, st>>>(dL2, ld, ld, dsq, R, dD, Rp);
    CONS_TRY(hipGetLastError());
    if (dist_out)
        CONS_TRY(hipMemcpy2DAsync(dist_out, (size_t)R * sizeof(double), dD, (size_t)Rp * sizeof(double),
                                  (size_t)R * sizeof(double), R, hipMemcpyDeviceToHost, st));
    if (silhouette_out) {
        std::vector<int> ints((size_t)3 * R + k + 1);
        int* rowid = ints.data(), *order = rowid + R, *lab = order + R, *seg = lab + R;
        for (int j = 0; j <= k; ++j) seg[j] = 0;
        for (int q = 0; q < R; ++q) {
            if (labels[q] < 0 || labels[q] >= k) { SET_ERR(ctx, "label %d of row %d outside 0..%d", (int)labels[q], q, k - 1); return CNMF_EINVAL; }
            rowid[q] = q; lab[q] = labels[q]; seg[labels[q] + 1]++;
        }
        for (int j = 0; j < k; ++j) seg[j + 1] += seg[j];
        { std::vector<int> pos(seg, seg + k); for (int q = 0; q < R; ++q) order[pos[labels[q]]++] = q; }
        int* dints = pool.get<int>(ints.size());
        double* dsil = pool.get<double>(R);
        double* dsum = pool.get<double>(1);
        if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        CONS_TRY(hipMemcpyAsync(dints, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice, st));
        silhouette_kernel<<<R, 256, 0, st>>>(dD, Rp, dints, dints + R, dints + 3 * R, dints + 2 * R, R, k, dsil);
        sum_kernel<<<1, 256, 0, st>>>(dsil, R, dsum);
        CONS_TRY(hipGetLastError());
        double sum = 0.0;
        CONS_TRY(hipMemcpyAsync(&sum, dsum, sizeof(double), hipMemcpyDeviceToHost, st));
        CONS_TRY(hipStreamSynchronize(st));
        *silhouette_out = sum / R;
    }
    CONS_TRY(hipStreamSynchronize(st));
    return CNMF_OK;
}

// Replaces the dense residual of cnmf.py:926-930: sum((X - W.H)^2) with X the resident matrix.
extern "C" int cnmf_prediction_error(cnmf_ctx* ctx, int k, const double* W, const double* H, double* err_out)
{
    using namespace cnmf;
    if (!ctx || !W || !H || !err_out || k < 1) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    if (k > KMAX) { SET_ERR(ctx, "k=%d > %d", k, KMAX); return CNMF_EUNSUPPORTED; }
    if (!ctx->X && ctx->csr_ptr) {
        // a matrix that lives as compressed rows only (round 5; the reference densifies it here, cnmf.py:927-928):
        // ||X - W H||^2 = sum_stored [(x - wh)^2 - (wh)^2] + tr(W^T W . H H^T), float64
        CONS_TRY(hipSetDevice(ctx->device));
        hipStream_t st_ = ctx->stream;
        const int N_ = (int)ctx->N, G_ = (int)ctx->G;
        std::vector<double> ht((size_t)G_ * k), wtw((size_t)k * k, 0.0), hht((size_t)k * k, 0.0);
        for (int c = 0; c < k; ++c) for (int j = 0; j < G_; ++j) ht[(size_t)j * k + c] = H[(size_t)c * G_ + j];
        for (int i = 0; i < N_; ++i) { const double* w = W + (size_t)i * k; for (int a = 0; a < k; ++a) for (int b = 0; b < k; ++b) wtw[a * k + b] += w[a] * w[b]; }
        for (int j = 0; j < G_; ++j) { const double* h = ht.data() + (size_t)j * k; for (int a = 0; a < k; ++a) for (int b = 0; b < k; ++b) hht[a * k + b] += h[a] * h[b]; }
        double tr = 0.0;
        for (int a = 0; a < k * k; ++a) tr += wtw[a] * hht[a];
        DevPool pool_;
        double* dW_ = pool_.get<double>((size_t)N_ * k);
        double* dHt_ = pool_.get<double>((size_t)G_ * k);
        double* dpart_ = pool_.get<double>(N_);
        double* dsum_ = pool_.get<double>(1);
        if (pool_.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
        CONS_TRY(hipMemcpyAsync(dW_, W, (size_t)N_ * k * sizeof(double), hipMemcpyHostToDevice, st_));
        CONS_TRY(hipMemcpyAsync(dHt_, ht.data(), (size_t)G_ * k * sizeof(double), hipMemcpyHostToDevice, st_));
        csr_residual_rows_kernel<<<(N_ + 3) / 4, 256, 0, st_>>>(ctx->csr_ptr, ctx->csr_idx, ctx->csr_val, N_, dW_, dHt_, k, dpart_);
        sum_kernel<<<1, 256, 0, st_>>>(dpart_, N_, dsum_);
        CONS_TRY(hipGetLastError());
        double s_ = 0.0;
        CONS_TRY(hipMemcpyAsync(&s_, dsum_, sizeof(double), hipMemcpyDeviceToHost, st_));
        CONS_TRY(hipStreamSynchronize(st_));
        *err_out = s_ + tr;
        return CNMF_OK;
    }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
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
