// Host driver of the multiplicative-update solver (included by cnmf_hip.hip).
// Restates _fit_multiplicative_update (sklearn/decomposition/_nmf.py:731-893): W then H
// half-steps, beta-divergence every 10 iterations, stop when (previous - error)/error_at_init < tol.
// Restarts run one after the other (each needs its own N x G ratio pass; batching them over
// a shared X tile is a round-2 item) -- this is the non-default path of the reference
// (beta_loss='frobenius' -> CD is the default, cnmf.py:334,630).
#pragma once
#include "kernels_mu.hip.h"
#include "kernels_mu_mfma.hip.h"
#include "kernels_mu_sparse.hip.h"

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
    int it = 0, err_it = 0;
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
            err_it = it;
            hsum_valid = true;                       // divergence() refreshed Hsum
            if ((prev - err) / err0 < prm->tol) break;
            prev = err;
        }
    }
    *n_iter_out = std::min(it, prm->max_iter);
    if (err_it != *n_iter_out) {                     // report the divergence of the FINAL factors (tol = 0, or max_iter % 10 != 0)
        rc = divergence(&err);
        if (rc) return rc;
    }
    *err_out = err;
    return CNMF_OK;
}

// ---- Kullback-Leibler restarts on the matrix pipe, up to MU_MAXSLOTS in flight (kernels_mu_mfma.hip.h).
// All live restarts advance one iteration per round of launches; the divergence is evaluated (and slots are retired /
// refilled) only at rounds where every live restart sits at a multiple of 10 iterations, so one host synchronisation
// per 10 iterations serves the whole batch (sklearn evaluates it every 10 iterations, _nmf.py:871-884).
static int mu_ensure_xt(cnmf_ctx* ctx, int Gs)
{
    if (ctx->XtF) return CNMF_OK;
    const size_t n = (size_t)Gs * ctx->N_pad;
    if (Gs / 32 > 65535) { SET_ERR(ctx, "X^T build: %d gene rows exceed the launch grid", Gs); return CNMF_EUNSUPPORTED; }
    float* xt = nullptr;
    HIP_TRY(ctx, hipMalloc(&xt, n * sizeof(float)));
    dim3 grid(ctx->N_pad / 32, Gs / 32), block(32, 8);        // cells on grid.x: no 65 535-block limit on N_pad
    mu_transpose_kernel<<<grid, block, 0, ctx->stream>>>(ctx->X, ctx->G_pad, (int)ctx->N, (int)ctx->G, xt, ctx->N_pad, Gs);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);       // built once per matrix: check it really ran
    if (e != hipSuccess) {
        // never leave a half-built X^T behind: the next call would skip the build and iterate on garbage
        hipFree(xt);
        HIP_TRY(ctx, e);
    }
    ctx->XtF = xt;
    return CNMF_OK;
}

// ---- Kullback-Leibler on the non-zeros: the blocked sliced-ELL images (kernels_mu_sparse.hip.h) of a matrix of R rows x C
// columns given as compressed rows (round 5: csr_host.hip.h -- the CSR upload as it came, or X^T's rows built on the device;
// the N x G dense image and its transposed copy are not touched on this path)
static int sp_build_image(cnmf_ctx* ctx, const long long* ptr, const int* idx, const float* val, int R, int C, int BS, SpImage& im)
{
    hipStream_t st = ctx->stream;
    const int nblk = (C + BS - 1) / BS, nslice = (R + 63) / 64, npos = nslice * 64;
    DevPool pool;
    int* dcnt = pool.get<int>((size_t)R * nblk, true, st);
    int* dpre = pool.get<int>((size_t)R * nblk);
    POOL_TRY(ctx, pool);
    sp_count_csr_kernel<<<(R + 3) / 4, 256, 0, st>>>(ptr, idx, R, BS, nblk, dcnt);
    sp_prefix_kernel<<<(R + 255) / 256, 256, 0, st>>>(dcnt, R, nblk, dpre);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<int> cnt((size_t)R * nblk);
    HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), dcnt, cnt.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    // rows by decreasing non-zero count (ties in row order): the rows of a slice are about equally long
    std::vector<long long> tot(R, 0);
    for (int r = 0; r < R; ++r) for (int b = 0; b < nblk; ++b) tot[r] += cnt[(size_t)r * nblk + b];
    std::vector<int> perm(npos, -1);
    for (int r = 0; r < R; ++r) perm[r] = r;
    std::stable_sort(perm.begin(), perm.begin() + R, [&](int a, int b) { return tot[a] > tot[b]; });
    std::vector<int> len((size_t)nslice * nblk);
    std::vector<long long> off((size_t)nslice * nblk);
    long long n_ent = 0;
    for (int s = 0; s < nslice; ++s)
        for (int b = 0; b < nblk; ++b) {
            int mx = 0;
            for (int l = 0; l < 64; ++l) { const int r = perm[(size_t)s * 64 + l]; if (r >= 0) mx = std::max(mx, cnt[(size_t)r * nblk + b]); }
            mx = round_up(mx, SP_UNROLL);
            len[(size_t)s * nblk + b] = mx; off[(size_t)s * nblk + b] = n_ent;
            n_ent += (long long)mx * 64;
        }
    im.release();
    hipError_t e = hipMalloc((void**)&im.perm, perm.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&im.off, off.size() * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc((void**)&im.len, len.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&im.ent, (size_t)std::max<long long>(n_ent, 64) * sizeof(uint2));
    if (e == hipSuccess) e = hipMemsetAsync(im.ent, 0, (size_t)std::max<long long>(n_ent, 64) * sizeof(uint2), st);
    if (e == hipSuccess) e = hipMemcpyAsync(im.perm, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(im.off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(im.len, len.data(), len.size() * sizeof(int), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        sp_fill_csr_kernel<<<(npos + 3) / 4, 256, 0, st>>>(ptr, idx, val, BS, nblk, npos, im.perm, im.off, dpre, (uint2*)im.ent, SP_LDS_BYTES / 4 / BS);
        e = hipGetLastError();
    }
    // a conflict-free order of every row's entries (CNMF_SP_ORDER=0: storage order, the A/B reference)
    const char* eo = ctx_getenv(ctx, "CNMF_SP_ORDER");
    if (e == hipSuccess && !(eo && atoi(eo) == 0) && n_ent > 0 && BS / 16 <= 255) {     // (its per-residue counters are bytes)
        uint2* tmp = nullptr;
        e = hipMalloc((void**)&tmp, (size_t)n_ent * sizeof(uint2));
        if (e == hipSuccess) {
            const long long jobs = (long long)nslice * nblk * 4;
            sp_reorder_kernel<<<(unsigned)((jobs + 63) / 64), 64, 0, st>>>(nslice, nblk, C, BS, SP_LDS_BYTES / 4 / BS, im.off, im.len,
                                                                          (uint2*)im.ent, tmp);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            hipFree(tmp);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);               // the host vectors above are stack objects
    if (e != hipSuccess) {
        im.release();
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return CNMF_ENOMEM; }      // (the caller falls back to the dense kernels)
        HIP_TRY(ctx, e);
    }
    im.R = R; im.C = C; im.BS = BS; im.nblk = nblk; im.nslice = nslice; im.n_ent = (size_t)n_ent;
    return CNMF_OK;
}

// Does this call take the non-zero path?  CNMF_MU_SPARSE=0 never, =1 always, otherwise when at most a quarter of X is
// non-zero (the dense matrix-pipe kernels cost ~0.9 ns per ELEMENT and restart-iteration at k <= 16, these ~3 ns per NON-ZERO)
// and its images and partial numerators fit the free device memory (else: the dense kernels, as before round 4).
static int mu_sparse_prepare(cnmf_ctx* ctx, int KP, bool* use)
{
    *use = false;
    const char* ev = ctx_getenv(ctx, "CNMF_MU_SPARSE");
    const int mode = ev ? atoi(ev) : -1;
    if (mode == 0 || (KP != 16 && KP != 32)) return CNMF_OK;
    const int N = (int)ctx->N, G = (int)ctx->G, idx = KP == 16 ? 0 : 1, BS = SP_LDS_BYTES / (KP * 4);
    // how many entries are stored?  From the compressed rows when the matrix came as CSR; a dense upload is only COUNTED here
    // (N + 1 counters) -- its compressed rows (12 B per entry) are built below, once the density and memory rules have chosen
    // this path, and not at all when they send the call back to the dense kernels (round-5 advice)
    int rc = CNMF_OK;
    if (ctx->csr_ptr) ctx->x_nnz = ctx->csr_nnz;
    else if (ctx->x_nnz < 0) {
        long long* cnt = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&cnt, ((size_t)N + 1) * sizeof(long long)));
        csr_count_dense_kernel<<<(N + 3) / 4, 256, 0, ctx->stream>>>(ctx->X, ctx->G_pad, N, G, cnt);
        long long nnz = 0;
        rc = hipGetLastError() == hipSuccess ? csr_scan_to_ptr(ctx, cnt, (size_t)N, &nnz) : CNMF_EHIP;
        hipFree(cnt);
        if (rc) return rc;
        ctx->x_nnz = nnz;
    }
    if (mode != 1 && (double)ctx->x_nnz > 0.25 * (double)N * (double)G) return CNMF_OK;
    if (!ctx->spA[idx].ent || !ctx->spB[idx].ent) {
        // the partial numerators of a half-step whose other side needs several blocks ([blocks][own rows][KP] per slot), the
        // two images (8 B per entry, padded: x 1.25 allowed for), X^T's compressed rows and the build's scratch
        const double nbA = std::ceil((double)G / BS), nbB = std::ceil((double)N / BS);
        const double part = 4.0 * KP * MU_MAXSLOTS * std::max(nbA > 1 ? nbA * ctx->N_pad : 0.0, nbB > 1 ? nbB * round_up(ctx->G_pad, 128) : 0.0);
        const double need = part + 2.0 * 1.25 * 8.0 * (double)ctx->x_nnz * 2.0 + 8.0 * (double)ctx->x_nnz + 8.0 * ((double)N * nbA + (double)G * nbB);
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        if (mode != 1 && (part > 64e9 || need > 0.9 * (double)free_b)) {
            if (ctx_getenv(ctx, "CNMF_DEBUG")) fprintf(stderr, "[cnmf] KL on the non-zeros: %.1f GB needed, %.1f GB free -> dense kernels\n", need / 1e9, free_b / 1e9);
            return CNMF_OK;
        }
    }
    rc = ensure_csr(ctx);                                       // (kept from the CSR upload, or two passes over the dense image)
    if (rc == CNMF_ENOMEM && mode != 1) return CNMF_OK;
    if (rc) return rc;
    if (!ctx->spA[idx].ent) rc = sp_build_image(ctx, ctx->csr_ptr, ctx->csr_idx, ctx->csr_val, N, G, BS, ctx->spA[idx]);
    if (!rc && !ctx->spB[idx].ent) {
        rc = ensure_csc(ctx);
        if (!rc) rc = sp_build_image(ctx, ctx->csc_ptr, ctx->csc_idx, ctx->csc_val, G, N, BS, ctx->spB[idx]);
    }
    if (rc == CNMF_ENOMEM && mode != 1) {                       // an allocation failed after all: the dense kernels still work
        ctx->spA[idx].release(); ctx->spB[idx].release();
        return CNMF_OK;
    }
    if (rc) return rc;
    if (ctx_getenv(ctx, "CNMF_DEBUG"))
        fprintf(stderr, "[cnmf] KL on the non-zeros: %lld of %lld elements (%.1f %%), padded entries A %.3f x, B %.3f x\n",
                ctx->x_nnz, (long long)N * G, 100.0 * ctx->x_nnz / ((double)N * G),
                ctx->spA[idx].n_ent / std::max(1.0, (double)ctx->x_nnz), ctx->spB[idx].n_ent / std::max(1.0, (double)ctx->x_nnz));
    *use = true;
    return CNMF_OK;
}

struct MuJob { int restart; int k; size_t hoff, woff; };

template <int KP, bool BETA1>
static int mu_batch_mfma(cnmf_ctx* ctx, const std::vector<MuJob>& jobs, int init_mode, const uint32_t* seeds,
                         const double* avg, const float* W0, const float* H0, int update_H,
                         const cnmf_cd_params* prm, float* H_out, float* W_out, int32_t* n_iter_out, double* err_out,
                         bool sparse = false)
{
    hipStream_t st = ctx->stream;
    const int N = (int)ctx->N, G = (int)ctx->G, ldx = ctx->G_pad, Np = ctx->N_pad;
    // the non-zero path (Kullback-Leibler, padded ranks 16 / 32): same slots, same schedule, other kernels
    BSellDev dA{}, dB{};
    int grpA = 0, grpB = 0, tilesA = 0, spwB = 1;
    if (sparse) {
        const SpImage &a = ctx->spA[KP == 16 ? 0 : 1], &b = ctx->spB[KP == 16 ? 0 : 1];
        dA = BSellDev{a.R, a.C, a.BS, a.nblk, a.nslice, a.perm, a.off, a.len, (const uint2*)a.ent};
        dB = BSellDev{b.R, b.C, b.BS, b.nblk, b.nslice, b.perm, b.off, b.len, (const uint2*)b.ent};
        // cells: one slice per wave (sorted rows: the slices of a workgroup are equally long); genes: all slices of up to
        // 8192 genes in ONE workgroup per block of cells, long and short slices paired on the waves
        spwB = std::max(1, std::min(8, (b.nslice + SP_WAVES - 1) / SP_WAVES));
        grpA = (a.nslice + SP_WAVES - 1) / SP_WAVES; grpB = (b.nslice + SP_WAVES * spwB - 1) / (SP_WAVES * spwB);
        tilesA = grpA * a.nblk;
    }
    const int Gs = round_up(ctx->G_pad, 128);
    int rc = sparse ? CNMF_OK : ensure_dense(ctx);               // (the non-zero path walks its images only)
    if (!rc && !sparse) rc = mu_ensure_xt(ctx, Gs);
    if (rc) return rc;
    constexpr int RPW = MuShape<KP>::RPW, SW = 32 * MuShape<KP>::NJT;      // restarts per workgroup, cells per divergence strip
    const int nstrips = std::max((N + SW - 1) / SW, tilesA), ntiles = Np / 32;
    const int ndiv = sparse ? tilesA : (N + SW - 1) / SW;                   // divergence partials per slot
    const int nchunks = std::min(32, ntiles), tpc = (ntiles + nchunks - 1) / nchunks;
    const int R = (int)std::min<size_t>(MU_MAXSLOTS, jobs.size());
    const float l1W = (float)prm->l1_reg_W, l2W = (float)prm->l2_reg_W;
    const float l1H = (float)prm->l1_reg_H, l2H = (float)prm->l2_reg_H;

    struct Slot { MuSlotDev d; int job = -1; int it = 0, err_it = -1; double err0 = 0, prev = 0, err = 0; bool fresh = false; };
    std::vector<Slot> slots(R);
    DevPool pool;
    int kmax = 1;
    for (const MuJob& j : jobs) kmax = std::max(kmax, j.k);
    for (Slot& s : slots) {
        MuSlotDev& d = s.d;
        d.W = pool.get<float>((size_t)Np * KP, true, st);
        d.Ht = pool.get<float>((size_t)Gs * KP, true, st);
        d.Wp_hi = pool.get<mu_u16>((size_t)Np * KP, true, st); d.Wp_lo = pool.get<mu_u16>((size_t)Np * KP, true, st);
        d.Wc_hi = pool.get<mu_u16>((size_t)2 * Np * KP, true, st);
        d.Hp_hi = pool.get<mu_u16>((size_t)Gs * KP, true, st); d.Hp_lo = pool.get<mu_u16>((size_t)Gs * KP, true, st);
        d.Hc_hi = pool.get<mu_u16>((size_t)2 * Gs * KP, true, st);
        d.Hsum = pool.get<float>(KP, true, st); d.Wsum = pool.get<float>(KP, true, st);
        d.pnum = sparse ? pool.get<float>(std::max<size_t>(dA.nblk > 1 ? (size_t)dA.nblk * Np : 0, dB.nblk > 1 ? (size_t)dB.nblk * Gs : 0) * KP)
                        : pool.get<float>((size_t)nchunks * Gs * KP);
        d.pden = BETA1 ? nullptr : pool.get<float>((size_t)nchunks * Gs * KP);
        d.divpart = pool.get<double>(nstrips);
        d.cspart = pool.get<double>((size_t)256 * KP);
    }
    // component-major staging of the initial / final factors, one kmax-row band per slot (the seeded initialisation of
    // all fresh slots is ONE launch: a workgroup per restart, the Mersenne twister is serial inside it)
    float* cmH = pool.get<float>((size_t)R * kmax * G);
    float* cmW = pool.get<float>((size_t)R * kmax * N);
    RngJob* dj = pool.get<RngJob>(R);
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    std::vector<double> hdiv((size_t)R * nstrips);
    std::vector<float> hsums((size_t)R * 2 * KP);

    constexpr int coop_lds = 2 * MuLds<KP>::BUF;
    {
        {
            HIP_TRY(ctx, dyn_lds_optin((const void*)mu_h_coop_kernel<KP, BETA1>, coop_lds));
            HIP_TRY(ctx, dyn_lds_optin((const void*)mu_w_coop_kernel<KP, 0, BETA1>, coop_lds));
            HIP_TRY(ctx, dyn_lds_optin((const void*)mu_w_coop_kernel<KP, 1, BETA1>, coop_lds));
            if constexpr (BETA1 && KP <= 32) if (sparse) {
                HIP_TRY(ctx, dyn_lds_optin((const void*)mu_sp_kernel<KP, 0, 0>, SP_LDS_BYTES));
                HIP_TRY(ctx, dyn_lds_optin((const void*)mu_sp_kernel<KP, 0, 1>, SP_LDS_BYTES));
                HIP_TRY(ctx, dyn_lds_optin((const void*)mu_sp_kernel<KP, 1, 0>, SP_LDS_BYTES));
            }
        }
    }
    auto batch_of = [&](const std::vector<int>& ids) { MuBatch mb; mb.n = (int)ids.size(); for (int i = 0; i < mb.n; ++i) mb.s[i] = slots[ids[i]].d; return mb; };
    auto colsum = [&](const MuBatch& mb, int which) {
        const int Rr = which ? G : N;
        const int nb = std::max(1, std::min(256, Rr / 256));
        mu_colsum_batch_part_kernel<KP><<<dim3(nb, mb.n), 256, 0, st>>>(mb, which, Rr);
        mu_colsum_batch_final_kernel<KP><<<dim3(1, mb.n), 1024, 0, st>>>(mb, which, nb);
    };
    auto install = [&](const std::vector<int>& sis, const std::vector<int>& jis) -> int {
        const int gHt = (G * KP + 255) / 256, gWp = (int)(((size_t)N * KP + 255) / 256);
        if (update_H && init_mode == 1) {
            std::vector<RngJob> hj(sis.size());
            for (size_t i = 0; i < sis.size(); ++i) {
                const MuJob& jb = jobs[jis[i]];
                hj[i] = RngJob{seeds[jb.restart], jb.k, sis[i] * kmax, avg[jb.restart], (long long)jb.k * ((long long)G + N)};
            }
            HIP_TRY(ctx, hipMemcpyAsync(dj, hj.data(), hj.size() * sizeof(RngJob), hipMemcpyHostToDevice, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));                      // `hj` is a stack temporary
            rng_kernel<1><<<(int)sis.size(), 256, 0, st>>>(dj, nullptr, cmH, G, G, cmW, N, N);
        }
        for (size_t i = 0; i < sis.size(); ++i) {
            Slot& s = slots[sis[i]];
            const MuJob& jb = jobs[jis[i]];
            const int k = jb.k, r = jb.restart;
            float* cH = cmH + (size_t)sis[i] * kmax * G;
            float* cW = cmW + (size_t)sis[i] * kmax * N;
            if (!update_H) {
                HIP_TRY(ctx, hipMemcpyAsync(cH, H0 + jb.hoff, (size_t)k * G * sizeof(float), hipMemcpyHostToDevice, st));
                mu_pack_kernel<<<gHt, 256, 0, st>>>(cH, k, G, s.d.Ht, KP);
                mu_fill_kernel<<<gWp, 256, 0, st>>>(s.d.W, k, N, KP, (float)avg[r]);       // sklearn _nmf.py:1229-1231
            } else if (init_mode == 0) {
                HIP_TRY(ctx, hipMemcpyAsync(cH, H0 + jb.hoff, (size_t)k * G * sizeof(float), hipMemcpyHostToDevice, st));
                HIP_TRY(ctx, hipMemcpyAsync(cW, W0 + jb.woff, (size_t)k * N * sizeof(float), hipMemcpyHostToDevice, st));
                mu_pack_kernel<<<gHt, 256, 0, st>>>(cH, k, G, s.d.Ht, KP);
                mu_pack_rm_kernel<<<gWp, 256, 0, st>>>(cW, k, N, s.d.W, KP);
            } else {
                mu_pack_kernel<<<gHt, 256, 0, st>>>(cH, k, G, s.d.Ht, KP);
                mu_pack_kernel<<<gWp, 256, 0, st>>>(cW, k, N, s.d.W, KP);
            }
            mu_planes_kernel<KP><<<(int)(((size_t)Np * KP + 255) / 256), 256, 0, st>>>(s.d.W, Np, s.d.Wp_hi, s.d.Wp_lo, s.d.Wc_hi, s.d.Wc_hi + (size_t)Np * KP);
            mu_planes_kernel<KP><<<(Gs * KP + 255) / 256, 256, 0, st>>>(s.d.Ht, Gs, s.d.Hp_hi, s.d.Hp_lo, s.d.Hc_hi, s.d.Hc_hi + (size_t)Gs * KP);
            s.job = jis[i]; s.it = 0; s.fresh = true; s.err0 = s.prev = s.err = 0.0; s.d.k = k;
        }
        HIP_TRY(ctx, hipGetLastError());
        return CNMF_OK;
    };
    auto retire = [&](int si) -> int {
        Slot& s = slots[si];
        const MuJob& jb = jobs[s.job];
        const int k = jb.k;
        float* cH = cmH + (size_t)si * kmax * G;
        float* cW = cmW + (size_t)si * kmax * N;
        if (H_out && update_H) {
            mu_unpack_kernel<<<(G * k + 255) / 256, 256, 0, st>>>(s.d.Ht, k, G, KP, cH, 1);
            HIP_TRY(ctx, hipMemcpyAsync(H_out + jb.hoff, cH, (size_t)k * G * sizeof(float), hipMemcpyDeviceToHost, st));
        }
        if (W_out) {
            mu_unpack_kernel<<<(int)(((size_t)N * k + 255) / 256), 256, 0, st>>>(s.d.W, k, N, KP, cW, 0);
            HIP_TRY(ctx, hipMemcpyAsync(W_out + jb.woff, cW, (size_t)k * N * sizeof(float), hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (n_iter_out) n_iter_out[jb.restart] = s.it;
        if (err_out) err_out[jb.restart] = s.err;
        s.job = -1;
        return CNMF_OK;
    };
    std::function<int(const std::vector<int>&, std::vector<double>&)> divergence_fn;
    // err_out is the divergence of the FINAL factors (what sklearn's reconstruction_err_ reports): a slot whose last
    // evaluation is older than its last iteration (tol = 0, or max_iter not a multiple of 10) is evaluated once more
    auto retire_final = [&](int si) -> int {
        Slot& s = slots[si];
        if (err_out && s.err_it != s.it) {
            std::vector<double> e;
            int rcf = divergence_fn(std::vector<int>{si}, e);
            if (rcf) return rcf;
            s.err = e[0]; s.err_it = s.it;
        }
        return retire(si);
    };
    // divergence of the current factors of the slots `ids` -> err[] (host)
    auto divergence = [&](const std::vector<int>& ids, std::vector<double>& err) -> int {
        const MuBatch mb = batch_of(ids);
        if (!update_H) colsum(mb, 0);                 // refit: the iterations do not need the column sums of W
        if (sparse) {
            if constexpr (BETA1 && KP <= 32)
                mu_sp_kernel<KP, 1, 0><<<(tilesA + 7) / 8 * 8 * mb.n, SP_WAVES * 64, SP_LDS_BYTES, st>>>(dA, mb, grpA, 1, 0.f, 0.f, Np);
        } else
            mu_w_coop_kernel<KP, 1, BETA1><<<dim3((N + 127) / 128, 1, (mb.n + RPW - 1) / RPW), 512, coop_lds, st>>>(ctx->XtF, Np, N, Gs, mb, 0.f, 0.f);
        HIP_TRY(ctx, hipGetLastError());
        for (int i = 0; i < mb.n; ++i) {
            HIP_TRY(ctx, hipMemcpyAsync(hdiv.data() + (size_t)i * nstrips, mb.s[i].divpart, (size_t)ndiv * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipMemcpyAsync(hsums.data() + (size_t)i * 2 * KP, mb.s[i].Hsum, KP * sizeof(float), hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipMemcpyAsync(hsums.data() + (size_t)i * 2 * KP + KP, mb.s[i].Wsum, KP * sizeof(float), hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(ctx, hipStreamSynchronize(st));
        err.resize(mb.n);
        for (int i = 0; i < mb.n; ++i) {
            double res = 0.0;
            for (int q = 0; q < ndiv; ++q) res += hdiv[(size_t)i * nstrips + q];
            if (BETA1) {                                   // + sum(WH) from the column sums
                double swh = 0.0;
                for (int c = 0; c < KP; ++c) swh += (double)hsums[(size_t)i * 2 * KP + c] * (double)hsums[(size_t)i * 2 * KP + KP + c];
                res += swh;
            } else res -= (double)N * (double)G;           // Itakura-Saito: sum(X/WH) - N G - sum log(X/WH)
            err[i] = std::sqrt(2.0 * std::max(res, 0.0));
        }
        return CNMF_OK;
    };

    divergence_fn = divergence;
    size_t next = 0;
    std::vector<int> ids;
    std::vector<double> errs;
    for (;;) {
        // ---- aligned point: every live slot sits at a multiple of 10 iterations (or nothing is live)
        bool aligned = true;
        for (const Slot& s : slots) if (s.job >= 0 && s.it % 10 != 0 && s.it < prm->max_iter) aligned = false;
        if (aligned) {
            // refill
            std::vector<int> fresh, fjobs;
            for (int si = 0; si < R && next < jobs.size(); ++si)
                if (slots[si].job < 0) { fresh.push_back(si); fjobs.push_back((int)next++); }
            if (!fresh.empty()) {
                rc = install(fresh, fjobs);
                if (rc) return rc;
                const MuBatch mb = batch_of(fresh); colsum(mb, 0); colsum(mb, 1);
            }
            // divergence of the fresh slots (error at init) and of the slots due for the convergence test
            ids.clear();
            for (int si = 0; si < R; ++si) {
                const Slot& s = slots[si];
                if (s.job < 0) continue;
                if (s.fresh || (prm->tol > 0 && s.it > 0 && s.it % 10 == 0)) ids.push_back(si);
            }
            if (!ids.empty()) {
                rc = divergence(ids, errs);
                if (rc) return rc;
                for (size_t i = 0; i < ids.size(); ++i) {
                    Slot& s = slots[ids[i]];
                    s.err_it = s.it;
                    if (s.fresh) { s.err0 = s.prev = s.err = errs[i]; s.fresh = false; continue; }
                    s.err = errs[i];
                    if ((s.prev - s.err) / s.err0 < prm->tol) { rc = retire(ids[i]); if (rc) return rc; }
                    else s.prev = s.err;
                }
            }
        }
        // slots that have used up their iterations
        for (int si = 0; si < R; ++si)
            if (slots[si].job >= 0 && slots[si].it >= prm->max_iter) { rc = retire_final(si); if (rc) return rc; }
        ids.clear();
        for (int si = 0; si < R; ++si) if (slots[si].job >= 0) ids.push_back(si);
        if (ids.empty()) { if (next >= jobs.size()) break; else continue; }
        // ---- one iteration of every live slot
        const MuBatch mb = batch_of(ids);
        const int gz4 = (mb.n + RPW - 1) / RPW;
        if (sparse) {
            if constexpr (BETA1 && KP <= 32) {
                mu_sp_kernel<KP, 0, 0><<<(tilesA + 7) / 8 * 8 * mb.n, SP_WAVES * 64, SP_LDS_BYTES, st>>>(dA, mb, grpA, 1, l1W, l2W, Np);
                if (dA.nblk > 1)
                    mu_sp_finish_kernel<KP><<<dim3((unsigned)(((size_t)N * KP + 255) / 256), mb.n), 256, 0, st>>>(mb, 0, N, Np, dA.nblk, l1W, l2W);
                if (update_H) {
                    colsum(mb, 0);
                    const int tilesB = grpB * dB.nblk;
                    mu_sp_kernel<KP, 0, 1><<<(tilesB + 7) / 8 * 8 * mb.n, SP_WAVES * 64, SP_LDS_BYTES, st>>>(dB, mb, grpB, spwB, l1H, l2H, Gs);
                    if (dB.nblk > 1)
                        mu_sp_finish_kernel<KP><<<dim3((G * KP + 255) / 256, mb.n), 256, 0, st>>>(mb, 1, G, Gs, dB.nblk, l1H, l2H);
                    colsum(mb, 1);
                }
            }
        } else {
        mu_w_coop_kernel<KP, 0, BETA1><<<dim3((N + 127) / 128, 1, gz4), 512, coop_lds, st>>>(ctx->XtF, Np, N, Gs, mb, l1W, l2W);
        if (update_H) {
            colsum(mb, 0);
            mu_h_coop_kernel<KP, BETA1><<<dim3(Gs / 128, nchunks, gz4), 512, coop_lds, st>>>(ctx->X, ldx, Np, Gs, mb, tpc, nchunks);
            mu_h_finish_mfma_kernel<KP, BETA1><<<dim3((Gs * KP + 255) / 256, mb.n), 256, 0, st>>>(mb, G, Gs, nchunks, l1H, l2H);
            colsum(mb, 1);
        }
        }
        HIP_TRY(ctx, hipGetLastError());
        for (int si : ids) slots[si].it++;
    }
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
    if (!ctx->X && !ctx->csr_ptr) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }
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
    for (int r = 0; r < n; ++r) {
        const int k = kk[r];
        if (k < 1) { SET_ERR(ctx, "n_components must be >= 1"); return CNMF_EINVAL; }
        if (k > CNMF_MU_KMAX) { SET_ERR(ctx, "n_components=%d > %d is not supported by the multiplicative-update solver on the device", k, CNMF_MU_KMAX); return CNMF_EUNSUPPORTED; }
    }
    // batched on the matrix pipe (kernels_mu_mfma.hip.h: padded ranks 16 / 32 / 64); the vector-ALU kernels below, one
    // restart at a time, serve CNMF_MU_VALU=1 and matrices beyond the 32-bit addressing bound
    std::vector<char> done(n, 0);
    {
        const char* e = ctx_getenv(ctx, "CNMF_MU_VALU");
        // (the matrix-pipe kernels address a 32-row step of X / X^T with 32-bit byte offsets below 2^31 -- the buffer
        //  descriptor's range: 4 * 32 * row length -> up to 2^24 cells or genes)
        if (!(e && atoi(e) != 0) && ctx->N_pad <= (1 << 24) && ctx->G_pad <= (1 << 24)) {
            std::vector<MuJob> j16, j32, j64;
            size_t ho = 0, wo = 0;
            for (int r = 0; r < n; ++r) {
                const int k = kk[r];
                if (k <= 16) j16.push_back(MuJob{r, k, ho, wo});
                else if (k <= 32) j32.push_back(MuJob{r, k, ho, wo});
                else j64.push_back(MuJob{r, k, ho, wo});
                done[r] = 1;
                ho += (size_t)k * G; wo += (size_t)k * N;
            }
#define MU_BATCH(KP_, B1_, jobs_, ...) mu_batch_mfma<KP_, B1_>(ctx, jobs_, init_mode, seeds, avg, W0, H0, update_H, prm, H_out, W_out, n_iter_out, err_out, ##__VA_ARGS__)
            // Kullback-Leibler at padded ranks 16 / 32 on a matrix that is mostly zeros: only the non-zeros are touched
            bool sp16 = false, sp32 = false;
            if (beta == 1 && !j16.empty()) { rc = mu_sparse_prepare(ctx, 16, &sp16); if (rc) return rc; }
            if (beta == 1 && !j32.empty()) { rc = mu_sparse_prepare(ctx, 32, &sp32); if (rc) return rc; }
            if (!j16.empty()) { rc = beta == 1 ? MU_BATCH(16, true, j16, sp16) : MU_BATCH(16, false, j16); if (rc) return rc; }
            if (!j32.empty()) { rc = beta == 1 ? MU_BATCH(32, true, j32, sp32) : MU_BATCH(32, false, j32); if (rc) return rc; }
            if (!j64.empty()) { rc = beta == 1 ? MU_BATCH(64, true, j64) : MU_BATCH(64, false, j64); if (rc) return rc; }
#undef MU_BATCH
        }
    }
    size_t hoff = 0, woff = 0;
    for (int r = 0; r < n; ++r) {
        const int k = kk[r];
        if (done[r]) { hoff += (size_t)k * G; woff += (size_t)k * N; continue; }
        const int KP = k <= 8 ? 8 : (k <= 16 ? 16 : (k <= 32 ? 32 : 64));
        if (int rcd_ = ensure_dense(ctx)) return rcd_;
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
        if (KP == 8) { MU_GO(8); } else if (KP == 16) { MU_GO(16); } else if (KP == 32) { MU_GO(32); } else { MU_GO(64); }
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
