// The restart engine: batch buffers, the slot scheduler (run_batch) and the NNLS refit
// (included by cnmf_hip.hip after gemm_host.hip.h).
#pragma once

// ------------------------------------------------------------------ batch buffers
static int sweep_max_parts()
{
    static const int v = getenv("CNMF_SWEEP_PARTS") ? atoi(getenv("CNMF_SWEEP_PARTS")) : 64;
    return std::max(1, v);
}
static int sweep_chunks(int L) { const int64_t m = 256ll * sweep_max_parts(); return std::max(1, (int)(((int64_t)L + m - 1) / m)); }
// partials per slot of a half-step: one per sweep workgroup (sweep_chunks 256-row tiles each)
static int sweep_parts(int L) { const int c = sweep_chunks(L); return (L + 256 * c - 1) / (256 * c); }

static int pick_nsplit(const cnmf_ctx* ctx, int KC)
{
    if (const char* s = ctx_getenv(ctx, "CNMF_NSPLIT")) { int v = atoi(s); if (v > 0) return v; }
    // pass B grid = ceil(G_pad/128) x (KC/128 or 1) x nsplit ; aim at ~2 workgroups per CU
    const int jt = (ctx->G_pad + 127) / 128;
    const int mg = std::max(1, KC / 128);
    int s = std::max(1, 512 / (jt * mg));
    const int max_by_k = std::max(1, ctx->N_pad / 256);   // at least 256 cells of K per split
    return std::min(s, max_by_k);
}

// splits that actually receive work once the per-split K range is rounded up to whole stages
static int effective_splits(int Ktot, int nsplit)
{
    const int Kper = round_up((Ktot + nsplit - 1) / nsplit, BK);
    return (Ktot + Kper - 1) / Kper;
}

// pass A on few cells: fewer than one 128-cell tile per CU -> split the gene (K) range too, so
// that ~2 workgroups per CU are in flight; the planes are summed by reduce_splits_kernel.
static int pick_nsplit_A(const cnmf_ctx* ctx, int KC)
{
    if (const char* s = ctx_getenv(ctx, "CNMF_NSPLIT_A")) { int v = atoi(s); if (v > 0) return effective_splits(ctx->G_pad, v); }
    const int T = (ctx->N_pad / 128) * std::max(1, KC / 128);
    if (T > 256) return 1;                                  // stream-K territory
    int s = std::max(1, 512 / T);
    s = std::min(s, std::max(1, ctx->G_pad / (4 * BK)));    // at least 4 stages per split
    return effective_splits(ctx->G_pad, std::min(s, 16));
}

// pass A of the split-operand kernels on few cell tiles (no stream-K below 3/4 of the workgroup slots):
// K splits so that about one workgroup per slot is in flight, >= 8 blocks each
static int pick_nsplit_A3(const cnmf_ctx* ctx, int KC, int jw)
{
    const int T = std::max(1, ctx->N_pad / jw) * std::max(1, KC / G3_MW);
    const int Kb = ctx->G_pad / G3_BK;
    int s = std::max(1, std::min(gemm3_wg_slots() / T, Kb / 8));
    const int kb_per = (Kb + s - 1) / s;
    return (Kb + kb_per - 1) / kb_per;
}

static int ensure_batch(cnmf_ctx* ctx, int KC, int max_k = KMAX, int min_k = 1)
{
    const bool use3 = gemm3_enabled(ctx, KC);
    const int nsplit = use3 ? std::max(pick_nsplit(ctx, KC), std::max(pick_nsplit3(ctx, KC, G3_JW), pick_nsplit3(ctx, KC, G3C_JW)))
                            : pick_nsplit(ctx, KC);
    const int nsplitA = use3 ? std::max(pick_nsplit_A(ctx, KC), std::max(pick_nsplit_A3(ctx, KC, G3_JW), pick_nsplit_A3(ctx, KC, G3C_JW)))
                             : pick_nsplit_A(ctx, KC);
    const int parts = std::max(sweep_parts((int)ctx->N), sweep_parts((int)ctx->G));
    const size_t gp_need = (size_t)(KC / std::max(1, min_k) + 1) * parts * max_k * max_k;
    if (ctx->kc_alloc == KC && ctx->nsplit_alloc == nsplit && ctx->nsplitA_alloc == nsplitA &&
        ctx->parts_alloc == parts && ctx->gram_part_floats >= gp_need && (!use3 || ctx->H3)) return CNMF_OK;
    free_batch(ctx);
    const size_t hb = (size_t)KC * ctx->G_pad * sizeof(float);
    const size_t wb = (size_t)KC * ctx->N_pad * sizeof(float);
    HIP_TRY(ctx, hipMalloc(&ctx->H, hb));
    HIP_TRY(ctx, hipMalloc(&ctx->Wt, wb));
    HIP_TRY(ctx, hipMalloc(&ctx->XHt, wb * nsplitA));
    HIP_TRY(ctx, hipMalloc(&ctx->XHt1, wb));
    if (use3) {
        HIP_TRY(ctx, hipMalloc(&ctx->XHt2, wb));
        HIP_TRY(ctx, hipMalloc(&ctx->H3, (size_t)KC * (ctx->G_pad / 16) * G3_ROWB));
        HIP_TRY(ctx, hipMalloc(&ctx->Wt3, (size_t)KC * (ctx->N_pad / 16) * G3_ROWB));
        HIP_TRY(ctx, hipMemsetAsync(ctx->Wt3, 0, (size_t)KC * (ctx->N_pad / 16) * G3_ROWB, ctx->stream));
        // f16 two-plane split: row-maximum partials written by the sweeps ([KC][parts]) and 2^-s per row
        HIP_TRY(ctx, hipMalloc(&ctx->rmaxH, (size_t)KC * parts * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&ctx->rmaxW, (size_t)KC * parts * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&ctx->iscaleH, (size_t)KC * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&ctx->iscaleW, (size_t)KC * sizeof(float)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->rmaxH, 0, (size_t)KC * parts * sizeof(float), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->rmaxW, 0, (size_t)KC * parts * sizeof(float), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->iscaleH, 0, (size_t)KC * sizeof(float), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->iscaleW, 0, (size_t)KC * sizeof(float), ctx->stream));
        HIP_TRY(ctx, hipMalloc(&ctx->shiftW, (size_t)3 * KC * sizeof(int)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->shiftW, 0, (size_t)3 * KC * sizeof(int), ctx->stream));
    }
    HIP_TRY(ctx, hipMalloc(&ctx->d_split, (size_t)(KC / 32 + 1) * (ctx->N_pad / 128 + 1)));
    HIP_TRY(ctx, hipMalloc(&ctx->XtW, hb * nsplit));
    HIP_TRY(ctx, hipMalloc(&ctx->gramH, (size_t)KC * GRAM_SZ * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&ctx->gramW, (size_t)KC * GRAM_SZ * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&ctx->gram_part, gp_need * sizeof(float)));
    ctx->gram_part_floats = gp_need;
    HIP_TRY(ctx, hipMalloc(&ctx->viol_part, (size_t)KC * parts * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_slots, (size_t)KC * sizeof(SlotDesc)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_slot_list, (size_t)KC * RING * sizeof(int)));
    HIP_TRY(ctx, hipHostMalloc(&ctx->h_slots, (size_t)KC * sizeof(SlotDesc)));
    // device-written, host-polled: coherent mapped pinned memory (zero-copy snapshots of the slot table)
    HIP_TRY(ctx, hipHostMalloc(&ctx->h_snap, (size_t)KC * RING * sizeof(SlotDesc),
                               hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->h_snap, 0, (size_t)KC * RING * sizeof(SlotDesc));
    HIP_TRY(ctx, hipHostMalloc(&ctx->h_slot_list, (size_t)KC * RING * sizeof(int)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->H, 0, hb, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->Wt, 0, wb, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->XHt, 0, wb * nsplitA, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->XtW, 0, hb * nsplit, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_slots, 0, (size_t)KC * sizeof(SlotDesc), ctx->stream));
    ctx->kc_alloc = KC; ctx->nsplit_alloc = nsplit; ctx->nsplitA_alloc = nsplitA; ctx->parts_alloc = parts;
    return CNMF_OK;
}

static int ensure_stage(cnmf_ctx* ctx, size_t wfloats, size_t hfloats)
{
    if (wfloats > ctx->stageW_sz) {
        hipFree(ctx->stageW); ctx->stageW = nullptr;
        HIP_TRY(ctx, hipMalloc(&ctx->stageW, wfloats * sizeof(float)));
        ctx->stageW_sz = wfloats;
    }
    if (hfloats > ctx->stageH_sz) {
        hipFree(ctx->stageH); ctx->stageH = nullptr;
        HIP_TRY(ctx, hipMalloc(&ctx->stageH, hfloats * sizeof(float)));
        ctx->stageH_sz = hfloats;
    }
    return CNMF_OK;
}

// simple first-fit interval allocator over the packed component columns
struct ColAlloc {
    std::vector<std::pair<int, int>> free_;   // (begin, length), sorted by begin
    explicit ColAlloc(int n) { free_.push_back({0, n}); }
    int alloc(int k) {
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].second >= k) {
                int b = free_[i].first;
                free_[i].first += k; free_[i].second -= k;
                if (free_[i].second == 0) free_.erase(free_.begin() + i);
                return b;
            }
        return -1;
    }
    void release(int b, int k) {
        auto it = std::lower_bound(free_.begin(), free_.end(), std::make_pair(b, 0));
        it = free_.insert(it, {b, k});
        if (it + 1 != free_.end() && it->first + it->second == (it + 1)->first) {
            it->second += (it + 1)->second; free_.erase(it + 1);
        }
        if (it != free_.begin() && (it - 1)->first + (it - 1)->second == it->first) {
            (it - 1)->second += it->second; free_.erase(it);
        }
    }
};

struct HostSlot { int state = 0; int restart = -1; int off = 0; int k = 0; int64_t installed_at = 0; };

// Poll the stamps of one zero-copy snapshot (finalize_kernel -> publish_slot) until all `n` slots carry
// `stamp`.  The snapshot is `lag` iterations old when it is needed, so this normally returns at once.
static int wait_snapshot(cnmf_ctx* ctx, const SlotDesc* sp, int n, int stamp)
{
    // Pure spinning on the host-mapped stamp.  The stream is queried (to notice a dead stream instead of spinning for
    // ever) only after 20 ms without progress: hipStreamQuery on a busy stream makes the runtime append a completion
    // marker behind the last enqueued kernel -- here always the H finalize -- and the GPU then pays ~6 us for that
    // barrier packet in front of every pass A (measured: the finalize -> pass A gap of round 2's kernel traces).
    using clk = std::chrono::steady_clock;
    static const bool eager_query = getenv("CNMF_SPIN_QUERY") != nullptr;       // A/B: the round-1 behaviour
    for (int s = 0; s < n; ++s) {
        const volatile int* flag = &sp[s].pad_;
        long spins = 0;
        clk::time_point t0;
        bool timing = false;
        while (*flag != stamp) {
            if (++spins % 4096 != 0) continue;
            if (!timing) { t0 = clk::now(); timing = true; continue; }
            if (!eager_query && std::chrono::duration_cast<std::chrono::milliseconds>(clk::now() - t0).count() < 20) continue;
            t0 = clk::now();
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {                       // stream drained: the stamp must be there
                if (*flag == stamp) break;
                SET_ERR(ctx, "slot snapshot %d was never published (slot %d)", stamp, s);
                return CNMF_EHIP;
            }
            if (q != hipErrorNotReady) {
                SET_ERR(ctx, "stream failed while waiting for a slot snapshot: %s", hipGetErrorString(q));
                return CNMF_EHIP;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return CNMF_OK;
}

// Packed component columns of a call.  32 ... 256 by the total rank of the job (powers of two); on the matrix-pipe
// split-operand paths (`wide_ok`) a larger job runs WIDE, up to 1024 columns = four 256-column component groups per GEMM
// pass: both passes then read every X tile once per 1024 columns instead of once per 256, the stream-K partial planes of
// pass A and the latency-bound H half-step are amortised over four times the columns (169 -> 198 -> 215 restarts/s at
// 256 / 512 / 1024 columns, 50 000 x 2000); the batch narrows in 256-column steps once the queue is dry (compact()).
// kc_max: 0 = auto, else an upper bound (multiple of 32; above 256 in steps of 256, up to CNMF_KC_LIMIT).
constexpr int CNMF_KC_LIMIT_DEFAULT = 1024;
static int kc_limit(const cnmf_ctx* ctx)   // (CNMF_KC_LIMIT: A/B knob, up to 2048 = the 64 tile bits of the live mask; read from
{                                          //  the context's snapshot of the environment like every per-call switch)
    const char* s = ctx_getenv(ctx, "CNMF_KC_LIMIT");
    const int v = s ? atoi(s) : CNMF_KC_LIMIT_DEFAULT;
    return std::max(256, std::min(2048, (v / 256) * 256));
}
#define CNMF_KC_LIMIT kc_limit(ctx)
static int pick_kc(const cnmf_ctx* ctx, int64_t total_k, int max_k, int kc_max, bool wide_ok)
{
    bool forced = false;
    if (const char* s = ctx_getenv(ctx, "CNMF_KC")) { int v = atoi(s); if (v >= 32) { kc_max = v; forced = true; } }
    const bool autosize = kc_max <= 0;
    if (autosize) kc_max = 256;
    kc_max = std::max(32, std::min(CNMF_KC_LIMIT, (kc_max / 32) * 32));
    if (kc_max > 256) kc_max = wide_ok ? (kc_max / 256) * 256 : 256;
    int kc = 32;
    while (kc < std::min(kc_max, 256) && kc < total_k) kc *= 2;
    kc = std::min(kc, kc_max);
    if (wide_ok && kc == 256 && total_k > 256) {
        // as wide as the job, up to the limit: a job that fits entirely starts every restart at once and the batch
        // narrows behind the ones that finish (compact()); measured and simulated on the iteration counts of the
        // north-star job (tools/sim_schedule.py): 113 restarts run 187 / 172 / 145 restarts/s at 1024 / 512 / 256 columns
        if (autosize && !ctx_getenv(ctx, "CNMF_NO_WIDE")) kc = (int)std::min<int64_t>(CNMF_KC_LIMIT, round_up(total_k, 256));
        else if (!autosize && kc_max > 256) kc = (int)std::min<int64_t>(kc_max, round_up(total_k, 256));
    }
    (void)forced;
    if (kc < max_k) kc = round_up(max_k, 32);
    return kc;
}

static int validate_params(cnmf_ctx* ctx, const cnmf_cd_params* p)
{
    if (!p) { SET_ERR(ctx, "params is NULL"); return CNMF_EINVAL; }
    if (!(p->tol >= 0) || p->max_iter < 1) { SET_ERR(ctx, "bad tol/max_iter"); return CNMF_EINVAL; }
    if (p->l1_reg_W < 0 || p->l2_reg_W < 0 || p->l1_reg_H < 0 || p->l2_reg_H < 0) {
        SET_ERR(ctx, "negative regularisation"); return CNMF_EINVAL;
    }
    return CNMF_OK;
}

// ------------------------------------------------------------------ the restart hot loop
static int run_batch(cnmf_ctx* ctx, int n, const int32_t* kk, int init_mode, const uint32_t* seeds,
                     const double* avg, const float* W0, const float* H0, const cnmf_cd_params* prm,
                     float* H_out, float* W_out, bool resident, int32_t* n_iter_out,
                     double* viol_out, cnmf_batch_stats* stats)
{
    if (!ctx) { SET_ERR(ctx, "ctx is NULL"); return CNMF_EINVAL; }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
    refresh_gemm3_mode(ctx);
    int rc = validate_params(ctx, prm);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !kk)) { SET_ERR(ctx, "bad restart list"); return CNMF_EINVAL; }
    if (init_mode == 0 && n > 0 && (!W0 || !H0)) { SET_ERR(ctx, "init_mode 0 needs W0 and H0"); return CNMF_EINVAL; }
    if (init_mode == 1 && n > 0 && (!seeds || !avg)) { SET_ERR(ctx, "init_mode 1 needs seeds and avg"); return CNMF_EINVAL; }
    if (init_mode != 0 && init_mode != 1) { SET_ERR(ctx, "unknown init_mode %d", init_mode); return CNMF_EINVAL; }
    if (!resident && n > 0 && !H_out) { SET_ERR(ctx, "H_out is NULL"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof *stats);
    if (n == 0) return CNMF_OK;

    const int N = (int)ctx->N, G = (int)ctx->G;
    int64_t total_k = 0; int max_k = 0, min_k = 1 << 30;
    std::vector<size_t> hoff(n + 1, 0), woff(n + 1, 0);
    for (int r = 0; r < n; ++r) {
        if (kk[r] < 1) { SET_ERR(ctx, "n_components must be >= 1 (restart %d)", r); return CNMF_EINVAL; }
        if (kk[r] > KMAX) { SET_ERR(ctx, "n_components=%d > CNMF_KMAX=%d is not supported by the device sweep", kk[r], KMAX); return CNMF_EUNSUPPORTED; }
        total_k += kk[r]; max_k = std::max(max_k, (int)kk[r]); min_k = std::min(min_k, (int)kk[r]);
        hoff[r + 1] = hoff[r] + (size_t)kk[r] * G;
        woff[r + 1] = woff[r] + (size_t)kk[r] * N;
    }
    // (wide batches need the whole-tile matrix-pipe kernels and a matrix large enough to be worth them)
    const bool wide_can = gemm3_mode() != 0 && gemm3_enabled(ctx, 512);
    // round 6: a SMALL matrix (BASELINE config 2: 2 700 x 2 000) is launch-latency-bound -- an iteration costs its six
    // launches whatever the width -- so a job of >= 512 columns also runs wide there (C2: 1 000 columns in one batch instead
    // of four 256-column rounds, 17 -> 8 ms per job); CNMF_WIDE_SMALL=0: the round-5 rule (A/B)
    const char* ws_ = ctx_getenv(ctx, "CNMF_WIDE_SMALL");
    const bool wide_small = total_k >= 512 && !(ws_ && atoi(ws_) == 0);
    const bool wide_auto = wide_can && ((int64_t)ctx->N_pad * ctx->G_pad >= (1ll << 24) || wide_small);
    int KC = pick_kc(ctx, total_k, max_k, prm->kc_max, (prm->kc_max > 256 || ctx_getenv(ctx, "CNMF_KC")) ? wide_can : wide_auto);
    // 65..128 columns of a count-structured matrix: the 256-column integer-plane kernels (half empty) are still
    // faster than 128 columns on the f32 pipe
    if (KC == 128 && prm->kc_max <= 0 && !ctx_getenv(ctx, "CNMF_KC") && gemm3_mode() >= 3 && gemm3_enabled(ctx, 256) &&
        (int64_t)ctx->N_pad * ctx->G_pad >= (1ll << 24)) {
        rc = ensure_counts(ctx);
        if (rc) return rc;
        if (ctx->count_state == 1) KC = 256;
    }
    const int KC0 = KC;
    rc = ensure_batch(ctx, KC, max_k, min_k);
    if (rc) return rc;
    rc = ensure_stage(ctx, (size_t)N * KMAX, (size_t)G * KMAX);
    if (rc) return rc;
    int nsplit = std::min(pick_nsplit(ctx, KC), ctx->nsplit_alloc);
    bool use3 = gemm3_enabled(ctx, KC);            // split-operand bf16 MFMA path (whole 256-column tiles only)
    bool usec = false;                             // ... with X as one integer plane (count-structured data)
    if (use3 && gemm3_mode() >= 3) {
        rc = ensure_counts(ctx);
        if (rc) return rc;
        usec = ctx->count_state == 1;
    }
    // any OTHER matrix in the default mode: X itself as two f16 planes with a per-row exponent, 4 MFMAs per product
    // (gemm_mode 5; CNMF_G2G=0 keeps the 3 x 3 bf16 planes of rounds 1-2, 6 MFMAs)
    static const bool g2g_off = getenv("CNMF_G2G") && atoi(getenv("CNMF_G2G")) == 0;
    bool use2g = use3 && !usec && gemm3_mode() >= 4 && !g2g_off && ctx->G_pad % G3C_JW == 0 && ctx->N_pad % G3C_JW == 0;
    if (use2g) { rc = ensure_x2planes(ctx); if (rc) return rc; }
    bool use2h = (usec && ctx->count_fmt == 4) || use2g;      // ... on the f16 pipe: two f16 factor planes
    const int gemm_mode_used = !use3 ? 0 : (use2g ? 5 : (usec ? (use2h ? 4 : 3) : std::min(gemm3_mode(), 2)));
    const int KbA = ctx->G_pad / 16, KbB = ctx->N_pad / 16;
    // the X-side operands of the f16 kernels: the integer count plane(s), or the two planes of a general matrix
    const unsigned char *xA = use2g ? ctx->X2h : ctx->C1, *xAhi = use2g ? ctx->X2m : ctx->C1h;
    const unsigned char *xB = use2g ? ctx->Xt2h : ctx->Ct1, *xBhi = use2g ? ctx->Xt2m : ctx->Ct1h;
    const unsigned int *xAfl = use2g ? ctx->onesA : ctx->hiA, *xBfl = use2g ? ctx->onesB : ctx->hiB;
    const float *csA = use2g ? ctx->x2sA : nullptr, *csB = use2g ? ctx->x2sB : nullptr;      // 2^-s per output column
    const double* dsc = use2g ? nullptr : ctx->d_scale;                                      // per-gene scale of the count path
    const int nsubA = use2h ? gemm2h_nsub(xAhi != nullptr, KbA) : 1, nsubB = use2h ? gemm2h_nsub(xBhi != nullptr, KbB) : 1;
    // W planes written by the W sweep itself (kernels_sweep.hip.h, PLN) instead of a separate pass over W; ranks above
    // 64 (sweep_big_kernel) keep the separate split for the whole call.  CNMF_FUSE_W=0: the round-2 scheme (A/B).
    static const bool fuse_off = getenv("CNMF_FUSE_W") && atoi(getenv("CNMF_FUSE_W")) == 0;
    const bool fuseW = use2h && !fuse_off && max_k <= KSMALL && ctx->shiftW != nullptr;
    int shgen = 0;                                        // generation of the exponents the NEXT W sweep uses
    int* const shW[2] = {ctx->shiftW, ctx->shiftW ? ctx->shiftW + ctx->kc_alloc : nullptr};
    if (fuseW) HIP_TRY(ctx, hipMemsetAsync(ctx->shiftW, 0, (size_t)3 * ctx->kc_alloc * sizeof(int), ctx->stream));
    if (use3 && !usec && !use2g) { rc = ensure_planes(ctx); if (rc) return rc; }
    const int jwA = (usec || use2g) ? G3C_JW : G3_JW;         // width of a pass-A / pass-B tile
    int nsplit3 = use3 ? pick_nsplit3(ctx, KC, jwA) : 1;
    const int nsplit3_first = nsplit3;             // (reported: the tail narrows the batch and re-plans)
    const int fin_y = (max_k * max_k + 255) / 256;       // finalize blocks per slot
    const int lag_env = ctx_getenv(ctx, "CNMF_LAG") ? atoi(ctx_getenv(ctx, "CNMF_LAG")) : 0;
    const int lag = std::max(1, std::min(RING - 2, prm->lag > 0 ? prm->lag : (lag_env > 0 ? lag_env : 2)));
    hipStream_t st = ctx->stream;

    // device result buffers
    DevPool pool;
    EventPool events;
    float* d_Hres = nullptr; float* d_Wres = nullptr;
    if (resident) {
        if (ctx->spectra_rows && ctx->spectra_G != G) {
            SET_ERR(ctx, "the resident store holds spectra over %lld genes, this matrix has %d: cnmf_spectra_reset first",
                    (long long)ctx->spectra_G, G);
            return CNMF_ESTATE;
        }
        ctx->spectra_G = G;
        const size_t need = (ctx->spectra_rows + (size_t)total_k) * G;
        if (need > ctx->spectra_cap) {
            float* nb = nullptr;
            const size_t cap = std::max(need, ctx->spectra_cap * 2);
            HIP_TRY(ctx, hipMalloc(&nb, cap * sizeof(float)));
            hipError_t ce = hipSuccess;
            if (ctx->spectra_rows)
                ce = hipMemcpyAsync(nb, ctx->spectra, ctx->spectra_rows * G * sizeof(float), hipMemcpyDeviceToDevice, st);
            if (ce == hipSuccess) ce = hipStreamSynchronize(st);
            if (ce != hipSuccess) { hipFree(nb); HIP_TRY(ctx, ce); }
            hipFree(ctx->spectra);
            ctx->spectra = nb; ctx->spectra_cap = cap;
        }
        d_Hres = ctx->spectra + ctx->spectra_rows * G;
    } else {
        d_Hres = pool.get<float>(hoff[n]);
    }
    if (W_out) d_Wres = pool.get<float>(woff[n]);
    int* d_repack = pool.get<int>((size_t)3 * KC0);            // re-packing: column map, moved slot ids, their new offsets
    POOL_TRY(ctx, pool);
    struct PinnedInts { int* p = nullptr; ~PinnedInts() { if (p) hipHostFree(p); } } repack_host;
    HIP_TRY(ctx, hipHostMalloc(&repack_host.p, (size_t)RING * 3 * KC0 * sizeof(int)));
    int* h_repack = repack_host.p;

    // init_mode 1: sklearn's init='random' for EVERY restart of the call, generated up front on the
    // device (one workgroup per restart) into a component-major store; install = row copy.
    float *d_H0 = nullptr, *d_Wt0 = nullptr;
    RngJob* d_jobs = nullptr;
    if (init_mode == 1) {
        d_H0 = pool.get<float>(hoff[n]);
        d_Wt0 = pool.get<float>(woff[n]);
        d_jobs = pool.get<RngJob>((size_t)n);
        POOL_TRY(ctx, pool);
        std::vector<RngJob> jobs(n);
        int rowoff = 0;
        for (int r = 0; r < n; ++r) {
            jobs[r] = RngJob{seeds[r], kk[r], rowoff, avg[r], (long long)kk[r] * ((long long)G + N)};
            rowoff += kk[r];
        }
        HIP_TRY(ctx, hipMemcpy(d_jobs, jobs.data(), (size_t)n * sizeof(RngJob), hipMemcpyHostToDevice));
        rng_kernel<1><<<n, 256, 0, st>>>(d_jobs, nullptr, d_H0, G, G, d_Wt0, N, N);
        HIP_TRY(ctx, hipGetLastError());
    }

    // Queue order.  Restarts are independent, so the order they run in is free; what it decides is the TAIL of the
    // call: once the queue is dry the columns of finished restarts stay empty, and the call ends when the LONGEST
    // restart still in flight converges (iterations per restart span 35 ... 1000 and depend mostly on the rank: ranks
    // far from the data's own need the most).  Hence longest-expected-first:
    //   * start: the ranks interleaved (round-robin over the distinct ranks, largest first inside a round), so every
    //     rank is sampled within the first fill of the packed columns;
    //   * as restarts retire, the mean iteration count per rank is learned and the pending queue is re-sorted by it,
    //     descending (LPT) -- the short restarts are kept for the end, where they fill the columns the long ones free.
    // CNMF_QUEUE=rank restores the plain descending-rank order (A/B).
    std::vector<int> order(n);
    const bool queue_by_rank = ctx_getenv(ctx, "CNMF_QUEUE") && !strcmp(ctx_getenv(ctx, "CNMF_QUEUE"), "rank");
    // round 4: the caller's iteration hints (cnmf_set_iteration_hints: mean iterations per rank, e.g. what an earlier call
    // on this matrix learned -- cnmf_get_iteration_means) order the queue from the start: a second factorize with more
    // restarts, a resumed ledger do not have to re-learn that k = 13 runs 1000 iterations and k = 9 runs 37.  Explicit only:
    // the order decides the packed columns a restart occupies and with them the last bits of its float32 result
    // (tests/test_gpu_determinism.py); without hints a call's result depends on its own arguments alone.
    const bool have_prior = ctx->iter_hint.size() == (size_t)KMAX + 1;
    auto prior_of = [&](int k) { return have_prior ? ctx->iter_hint[k] : 0.0; };
    if (queue_by_rank) {
        for (int r = 0; r < n; ++r) order[r] = r;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kk[a] > kk[b]; });
    } else {
        std::vector<std::vector<int>> by_k(KMAX + 1);
        for (int r = 0; r < n; ++r) by_k[kk[r]].push_back(r);
        size_t pos = 0;
        for (size_t j = 0; pos < (size_t)n; ++j)
            for (int k = KMAX; k >= 1; --k)
                if (j < by_k[k].size()) order[pos++] = by_k[k][j];
        if (have_prior)            // ranks never seen count as longest (sampled first), the others by their learned mean
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                const double ea = prior_of(kk[a]) > 0 ? prior_of(kk[a]) : 1e30, eb = prior_of(kk[b]) > 0 ? prior_of(kk[b]) : 1e30;
                return ea > eb;
            });
    }
    std::vector<int64_t> k_iters(KMAX + 1, 0), k_done(KMAX + 1, 0);     // learned per rank: sum of n_iter, restarts retired
    std::vector<double> rank_expect(KMAX + 1, 1e30);                   // expected iterations per rank (1e30: nothing known yet)
    static const bool holes_fill = !(getenv("CNMF_HOLES") && !strcmp(getenv("CNMF_HOLES"), "wait"));
    if (have_prior)
        for (int k = 1; k <= KMAX; ++k) if (prior_of(k) > 0) rank_expect[k] = prior_of(k);
    int last_resort_done = 0;
    size_t next = 0;                 // first queue position that may still be pending
    int n_pending = n;

    ColAlloc cols(KC);
    std::vector<HostSlot> hs(KC0);
    int nslots = 0;          // highest used slot index + 1
    int n_active = 0;
    int64_t it = 0;          // batch iterations enqueued so far
    int snap_nslots[RING] = {0};
    const bool no_defrag = ctx_getenv(ctx, "CNMF_NO_DEFRAG") != nullptr;
    const bool no_psum = ctx_getenv(ctx, "CNMF_NO_PSUM") != nullptr;          // (A/B knob: keep the separate split-K reduce)
    // partial-tile passes in the tail (the f16 kernels skip the MFMAs of 32-column tiles without a live restart, the live
    // restarts dealt evenly to the component groups): built and measured in round 3 -- 183.0 vs 182.2 restarts/s on the
    // 113-restart shard, 216.5 vs 215.7 on the full job: within noise (the skipping variant is ~4 % slower with all tiles
    // live, and a tail pass is bound by streaming the count plane as much as by its MFMAs).  Opt-in: CNMF_PART=1.
    const bool no_part = ctx_getenv(ctx, "CNMF_PART") == nullptr || ctx_getenv(ctx, "CNMF_NO_PART") != nullptr;
    int64_t last_tail_repack = -1000;
    int64_t last_defrag = -8, n_defrag = 0;
    bool h3_valid = false;           // H3 holds the planes of the current H (split-operand modes)
    // stamps restart at 1 in every call: forget the ones a previous call left in the ring (nothing is in flight here)
    memset(ctx->h_snap, 0, (size_t)ctx->kc_alloc * RING * sizeof(SlotDesc));
    hipEvent_t ev_begin = events.get(), ev_end = events.get();
    POOL_TRY(ctx, events);
    HIP_TRY(ctx, hipEventRecord(ev_begin, st));
    // HIP events around the two GEMM passes of every `time_stride`-th iteration (an event record costs
    // ~6 us of queue time: bracketing every launch would take 4 % off the throughput it measures)
    const int time_stride = !stats ? 0 : (prm->profile > 0 ? prm->profile : (ctx_getenv(ctx, "CNMF_TIME_GEMM") ? 1 : 0));
    std::vector<hipEvent_t> gev;   // (a0,a1,b0,b1) per iteration when timing is requested

    const int chunksW = sweep_chunks(N), partsW = sweep_parts(N);
    const int chunksH = sweep_chunks(G), partsH = sweep_parts(G);
    const float l1W = (float)prm->l1_reg_W, l2W = (float)prm->l2_reg_W;
    const float l1H = (float)prm->l1_reg_H, l2H = (float)prm->l2_reg_H;
    const int gvarA = ctx_getenv(ctx, "CNMF_GEMM_A") ? atoi(ctx_getenv(ctx, "CNMF_GEMM_A")) : 0;
    const int gvarB = ctx_getenv(ctx, "CNMF_GEMM_B") ? atoi(ctx_getenv(ctx, "CNMF_GEMM_B")) : 0;
    int64_t restart_iters = 0, column_iters = 0, restart_col_iters = 0;
    const bool dbg = ctx_getenv(ctx, "CNMF_DEBUG") != nullptr;
    int64_t dbg_it[65] = {0}, dbg_live[65] = {0};          // by KC / 32 (up to 2048 packed columns)
    const int wg_slots = ctx_getenv(ctx, "CNMF_SK_WGS") ? atoi(ctx_getenv(ctx, "CNMF_SK_WGS")) : 2 * 256;                    // T-layout pass A: 2 workgroups per CU (73.7 KB LDS each)
    StreamK sk = plan_streamk(KC, ctx->N_pad, ctx->G_pad, wg_slots);
    if (sk.on) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_split, sk.split.data(), sk.split.size(), hipMemcpyHostToDevice, st));
    int nsplitA = (sk.on && gvarA == 0) ? 1 : std::min(pick_nsplit_A(ctx, KC), ctx->nsplitA_alloc);
    StreamK3 sk3;
    if (use3) {
        sk3 = plan_streamk3(KC, ctx->N_pad, ctx->G_pad, gemm3_wg_slots(), jwA, nsubA);
        if (sk3.on) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_split, sk3.flags.data(), sk3.flags.size(), hipMemcpyHostToDevice, st));
        else nsplitA = std::min(pick_nsplit_A3(ctx, KC, jwA), ctx->nsplitA_alloc);   // few tiles: K split + reduce
    }
    int n_done = 0;
    hipEvent_t ev_tail = nullptr;    // recorded when the queue runs dry
    int64_t tail_its = 0, tail_live = 0;

    auto retire = [&](int s, const SlotDesc& snap) -> int {
        HostSlot& h = hs[s];
        const int r = h.restart, k = h.k;
        dim3 gH((G + 255) / 256, k), gW((N + 255) / 256, k);
        extract_kernel<<<gH, 256, 0, st>>>(ctx->H, ctx->G_pad, G, h.off, k, d_Hres + hoff[r], 0);
        if (d_Wres) extract_kernel<<<gW, 256, 0, st>>>(ctx->Wt, ctx->N_pad, N, h.off, k, d_Wres + woff[r], 1);
        clear_rows_kernel<<<gH, 256, 0, st>>>(ctx->H, ctx->G_pad, ctx->G_pad, h.off, k);
        clear_rows_kernel<<<gW, 256, 0, st>>>(ctx->Wt, ctx->N_pad, ctx->N_pad, h.off, k);
        HIP_TRY(ctx, hipGetLastError());
        if (n_iter_out) n_iter_out[r] = snap.iter;
        if (viol_out) viol_out[r] = snap.viol_last;
        restart_iters += snap.iter;
        restart_col_iters += (int64_t)snap.iter * k;
        k_iters[k] += snap.iter; k_done[k] += 1;
        if (n_pending > 0) cols.release(h.off, k);       // (after the queue ran dry nothing is allocated any more)
        h.state = 0; h.restart = -1;
        --n_active; ++n_done;
        return CNMF_OK;
    };

    // Move the live slots to the left end of the packed columns (ascending offset order, through the stage
    // buffers) and zero everything from the first free column up to `width`.  Used by the tail compaction
    // (narrower batch) and by the refill's defragmentation (same width).
    // (one launch sequence per re-packing, whatever the number of slots: at 1024 packed columns the per-slot moves of
    //  round 2 -- four launches each, host-bound at ~5 us per launch -- had grown to 10 % of the wall time)
    int n_repack = 0;
    // balanced (the tail of a wide batch only: nothing will be installed any more): the live restarts are dealt to the
    // 256-column component groups so that every group holds about the same number of live 32-column tiles -- the GEMM
    // workgroups of ALL groups then skip the same share of their MFMAs (a pass lasts as long as its slowest workgroup)
    auto repack_left = [&](int width, bool balanced = false) -> int {
        std::vector<int> idx;
        for (int s = 0; s < nslots; ++s) if (hs[s].state) idx.push_back(s);
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return hs[a].off < hs[b].off; });
        int* hm = h_repack + (size_t)(n_repack++ % RING) * 3 * KC0;      // [colmap | slot ids | new offsets], pinned
        int* d_colmap = d_repack, *d_ids = d_repack + KC0, *d_offs = d_repack + 2 * KC0;
        int pos = 0, nmove = 0;
        const int ngroups = width / 256;
        if (balanced && ngroups > 1) {
            for (int c = 0; c < width; ++c) hm[c] = -1;
            std::vector<int> fill(ngroups, 0), byk(idx);
            std::stable_sort(byk.begin(), byk.end(), [&](int a, int b) { return hs[a].k > hs[b].k; });
            bool ok = true;
            std::vector<int> newoff(nslots, -1);
            for (int s : byk) {
                int g = -1;
                for (int q = 0; q < ngroups; ++q) if (fill[q] + hs[s].k <= 256 && (g < 0 || fill[q] < fill[g])) g = q;
                if (g < 0) { ok = false; break; }
                newoff[s] = g * 256 + fill[g]; fill[g] += hs[s].k;
            }
            if (ok) {
                for (int s : idx) {
                    HostSlot& h = hs[s];
                    for (int c = 0; c < h.k; ++c) hm[newoff[s] + c] = h.off + c;
                    if (h.off != newoff[s]) { hm[KC0 + nmove] = s; hm[2 * KC0 + nmove] = newoff[s]; ++nmove; h.off = newoff[s]; }
                }
                pos = 0;                                                  // (there are zero rows inside [0, width) now)
            } else balanced = false;
        } else balanced = false;
        if (!balanced) {
        for (int s : idx) {
            HostSlot& h = hs[s];
            for (int c = 0; c < h.k; ++c) hm[pos + c] = h.off + c;
            if (h.off != pos) { hm[KC0 + nmove] = s; hm[2 * KC0 + nmove] = pos; ++nmove; h.off = pos; }
            pos += h.k;
        }
        for (int c = pos; c < width; ++c) hm[c] = -1;                    // zero rows behind the live ones
        }
        if (nmove || pos < width) {
            HIP_TRY(ctx, hipMemcpyAsync(d_colmap, hm, (size_t)width * sizeof(int), hipMemcpyHostToDevice, st));
            // staging = the product buffers (scratch between two iterations): XHt [>= KC0][N_pad], XtW [>= KC0][G_pad]
            gather_rows_cm_kernel<<<dim3((ctx->N_pad / 4 + 255) / 256, width), 256, 0, st>>>(ctx->Wt, ctx->N_pad, d_colmap, ctx->XHt);
            gather_rows_cm_kernel<<<dim3((ctx->G_pad / 4 + 255) / 256, width), 256, 0, st>>>(ctx->H, ctx->G_pad, d_colmap, ctx->XtW);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(ctx->Wt, ctx->XHt, (size_t)width * ctx->N_pad * sizeof(float), hipMemcpyDeviceToDevice, st));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->H, ctx->XtW, (size_t)width * ctx->G_pad * sizeof(float), hipMemcpyDeviceToDevice, st));
            if (fuseW) {
                int* tmp = ctx->shiftW + 2 * ctx->kc_alloc;
                for (int g2 = 0; g2 < 2; ++g2) {
                    permute_ints_kernel<<<(width + 255) / 256, 256, 0, st>>>(shW[g2], d_colmap, tmp, width);
                    HIP_TRY(ctx, hipMemcpyAsync(shW[g2], tmp, (size_t)width * sizeof(int), hipMemcpyDeviceToDevice, st));
                }
                HIP_TRY(ctx, hipGetLastError());
            }
            if (nmove) {
                HIP_TRY(ctx, hipMemcpyAsync(d_ids, hm + KC0, (size_t)nmove * sizeof(int), hipMemcpyHostToDevice, st));
                HIP_TRY(ctx, hipMemcpyAsync(d_offs, hm + 2 * KC0, (size_t)nmove * sizeof(int), hipMemcpyHostToDevice, st));
                set_slot_offs_kernel<<<(nmove + 255) / 256, 256, 0, st>>>(ctx->d_slots, d_ids, d_offs, nmove);
                HIP_TRY(ctx, hipGetLastError());
            }
        }
        h3_valid = false;                 // rows of H moved: its planes are stale
        // moved rows of slots that no longer iterate are not swept again: refresh their row maxima here
        if (use2h) HIP_TRY(ctx, launch_rowmax_part(st, ctx->Wt, ctx->N_pad, N, KC, chunksW * 256, nullptr, partsW, ctx->rmaxW));
        return CNMF_OK;
    };

    // ---- step 1: refill free columns from the pending list (n_new = slots installed)
    auto refill = [&](int& n_new) -> int {
        n_new = 0;
        int* new_list = ctx->h_slot_list + (size_t)(it % RING) * KC0;
        // pending restarts are sorted by descending rank; a hole too small for the head of the
        // queue is filled with the largest pending rank that fits (restarts are independent, so
        // the order they run in is free) -> the packed columns stay full in the main phase
        // round-4 experiment (CNMF_HOLES=wait, not the default).  Filling every freed slot at once with the largest pending
        // rank that fits keeps the columns full but freezes every rank's share of them at what the first fill gave it: free
        // columns never accumulate, so a rank-13 restart only ever starts where a rank-13 (or larger) restart ended, and the
        // longest-expected-first order decides nothing across ranks (debug trace: 13 restarts of rank 13 in flight from start
        // to end, 31 still running when the queue is dry).  "wait": once the HEAD of the queue does not fit, a later entry may
        // take a hole only if its rank is known to be short against the head's expectation; otherwise the hole waits until the
        // free columns add up to the head's rank and a re-packing makes room.  Measured (same box, two alternations): tail
        // 9.7 -> 7.9 % of the call, but 218 instead of 115 re-packings and idle holes: 221.0 / 220.5 vs 222.4 / 221.7
        // restarts/s -- not adopted.
        for (int attempt = 0; attempt < 2; ++attempt) {
        int failed_k = 1 << 30;                       // smallest rank that did not fit in this pass
        int head_k = 0; double head_expect = 0.0;     // the first pending entry that did not fit, and what it is expected to run
        for (size_t pi = next; pi < order.size() && n_pending > 0; ++pi) {
            const int r = order[pi];
            if (r < 0) { if (pi == next) ++next; continue; }      // already taken
            const int k = kk[r];
            if (k >= failed_k) continue;
            if (head_k && !holes_fill && !(rank_expect[k] < 0.3 * head_expect)) continue;     // (unknown = 1e30: not short)
            const int off = cols.alloc(k);
            if (off < 0) {
                failed_k = k;
                if (!head_k) { head_k = k; head_expect = rank_expect[k]; }
                continue;
            }
            order[pi] = -1; --n_pending;
            if (pi == next) ++next;
            int s = 0;
            while (s < KC0 && hs[s].state != 0) ++s;
            hs[s].state = 1; hs[s].restart = r; hs[s].off = off; hs[s].k = k; hs[s].installed_at = it;
            nslots = std::max(nslots, s + 1);
            dim3 gI((std::max(N, G) + 255) / 256, k);
            if (init_mode == 0) {
                HIP_TRY(ctx, hipMemcpyAsync(ctx->stageH, H0 + hoff[r], (size_t)k * G * sizeof(float), hipMemcpyHostToDevice, st));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->stageW, W0 + woff[r], (size_t)k * N * sizeof(float), hipMemcpyHostToDevice, st));
                install_kernel<<<gI, 256, 0, st>>>(ctx->stageH, ctx->stageW, ctx->H, ctx->G_pad, G, ctx->Wt, ctx->N_pad, N, off, k);
            } else {
                install_cm_kernel<<<gI, 256, 0, st>>>(d_H0 + hoff[r], d_Wt0 + woff[r], ctx->H, ctx->G_pad, G, ctx->Wt, ctx->N_pad, N, off);
            }
            HIP_TRY(ctx, hipGetLastError());
            if (fuseW) {
                // exponent of the planes the FIRST W sweep of this restart writes: from the size of its W0 (random init:
                // avg |N(0,1)|, the maximum of 50 000 draws is ~4.5 avg), two bits of headroom; the check behind the sweep
                // re-converts the rows whose first update left the window
                float mx = 0.f;
                if (init_mode == 1) mx = 6.0f * (float)avg[r];
                else for (size_t q = 0; q < (size_t)k * N; ++q) mx = std::max(mx, W0[woff[r] + q]);
                int e2 = 0;
                if (mx > 0.f) std::frexp(mx, &e2);
                const int sh0 = mx > 0.f ? std::max(-110, std::min(120, 13 - e2)) : 0;
                set_ints_kernel<<<1, 128, 0, st>>>(shW[0], shW[1], off, k, sh0);
                HIP_TRY(ctx, hipGetLastError());
            }
            SlotDesc* d = &ctx->h_slots[s];
            memset(d, 0, sizeof *d);
            d->off = off; d->k = k; d->active = 1; d->iter = 0; d->restart = r;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->d_slots + s, d, sizeof(SlotDesc), hipMemcpyHostToDevice, st));
            new_list[n_new++] = s;
            ++n_active;
        }
        // Defragmentation: a pending restart did not fit although enough columns are free in total (holes
        // left by retired slots of other ranks).  Repacking costs about one iteration and buys a restart
        // that runs for hundreds -- at most once every 8 iterations.
        if (attempt == 0 && n_pending > 0 && failed_k < (1 << 30) && !no_defrag && it - last_defrag >= 8) {
            int live_cols = 0;
            for (int s2 = 0; s2 < nslots; ++s2) if (hs[s2].state) live_cols += hs[s2].k;
            if (KC - live_cols >= (holes_fill ? failed_k : head_k)) {
                int rcd = repack_left(KC);
                if (rcd) return rcd;
                cols = ColAlloc(KC);
                for (int s2 = 0; s2 < nslots; ++s2) if (hs[s2].state) cols.alloc(hs[s2].k);
                last_defrag = it; ++n_defrag;
                continue;                               // place again into the contiguous free tail
            }
        }
        break;
        }
        if (n_new) {
            int* dl = ctx->d_slot_list + (size_t)(it % RING) * KC0;
            HIP_TRY(ctx, hipMemcpyAsync(dl, new_list, n_new * sizeof(int), hipMemcpyHostToDevice, st));
            gram_rows_kernel<<<n_new, 256, 0, st>>>(ctx->H, ctx->G_pad, G, ctx->d_slots, dl, ctx->gramH, l2W);
            HIP_TRY(ctx, hipGetLastError());
        }
        return CNMF_OK;
    };

    // ---- step 2: enqueue one coordinate-descent outer iteration for every slot in flight
    auto iterate = [&](int n_new) -> int {
        int tiers = 0;
        for (int s2 = 0; s2 < nslots; ++s2)
            if (hs[s2].state) tiers |= hs[s2].k <= 16 ? 1 : (hs[s2].k <= 32 ? 2 : (hs[s2].k <= KSMALL ? 4 : 8));
        // the tail (nothing left to refill with): the f16 kernels skip the 32-column tiles without a live restart
        unsigned long long livemask = ~0ull;
        if (use2h && n_pending == 0 && !no_part) {
            livemask = 0ull;
            for (int s2 = 0; s2 < nslots; ++s2)
                if (hs[s2].state)
                    for (int t = hs[s2].off / 32; t <= (hs[s2].off + hs[s2].k - 1) / 32 && t < 64; ++t) livemask |= 1ull << t;
            // the skipping variant of the kernels is ~4 % slower with everything live, and a pass lasts as long as its
            // fullest component group: use it only when EVERY group has at least two dead tiles
            int fullest = 0;
            for (int g = 0; g < KC / 256; ++g) fullest = std::max(fullest, __builtin_popcountll((livemask >> (8 * g)) & 0xffull));
            if (fullest > 6) livemask = ~0ull;
        }
        const bool time_gemm = time_stride > 0 && it % time_stride == 0;
        if (time_gemm) {
            for (int i = 0; i < 4; ++i) gev.push_back(events.get());
            POOL_TRY(ctx, events);
            hipEventRecord(gev[gev.size() - 4], st);
        }
        // pass A : XHt[KC][N] = H_all . X^T                       (sklearn _nmf.py:387)
        SplitInfo spA{nullptr, nullptr, 1, 1, 1};
        if (use3) {
            // H3 was produced together with the previous iteration's H finalize; rows installed since then
            // (and the very first iteration) need a split of their own.  Count path: H' = H * d.
            if ((n_new > 0 || !h3_valid) && use2h) {
                HIP_TRY(ctx, launch_rowmax_part(st, ctx->H, ctx->G_pad, G, KC, chunksH * 256, dsc, partsH, ctx->rmaxH));
                HIP_TRY(ctx, launch_split2h(st, ctx->H, ctx->G_pad, KC, ctx->G_pad, ctx->H3, G3_MW, dsc, ctx->rmaxH,
                                            partsH, ctx->iscaleH));
            } else if (n_new > 0 || !h3_valid)
                HIP_TRY(ctx, launch_split3(st, ctx->H, ctx->G_pad, KC, ctx->G_pad, ctx->H3, G3_MW, usec ? ctx->d_scale : nullptr));
            if (time_gemm) hipEventRecord(gev[gev.size() - 4], st);
            if (sk3.on) {
                if (use2h) {
                    HIP_TRY(ctx, launch_gemm2h_streamk(st, sk3, ctx->H3, xA, xAhi, xAfl, ctx->iscaleH, KbA,
                                                       ctx->XHt, ctx->XHt1, ctx->XHt2, ctx->N_pad, csA, livemask));
                }
                else if (usec)
                    HIP_TRY(ctx, launch_gemm3c_streamk(st, sk3, ctx->H3, ctx->C1, ctx->C1h, ctx->hiA, ctx->XHt, ctx->XHt1,
                                                       ctx->XHt2, ctx->N_pad));
                else
                    HIP_TRY(ctx, launch_gemm3_streamk(st, sk3, ctx->H3, ctx->X3, ctx->XHt, ctx->XHt1, ctx->XHt2, ctx->N_pad));
                spA = SplitInfo{ctx->XHt1, ctx->d_split, jwA, G3_MW, sk3.MG, ctx->XHt2};
            } else if (use2h) {
                HIP_TRY(ctx, launch_gemm2h(st, ctx->H3, xA, xAhi, xAfl, ctx->iscaleH, KbA, ctx->XHt, ctx->N_pad,
                                           (long long)KC * ctx->N_pad, KC, ctx->N_pad, nsplitA, csA, livemask));
            } else if (usec) {
                HIP_TRY(ctx, launch_gemm3c(st, ctx->H3, ctx->C1, ctx->C1h, ctx->hiA, ctx->G_pad / 16, ctx->XHt, ctx->N_pad,
                                           (long long)KC * ctx->N_pad, KC, ctx->N_pad, nsplitA));
            } else {
                HIP_TRY(ctx, launch_gemm3(st, ctx->H3, ctx->X3, ctx->G_pad / 16, ctx->XHt, ctx->N_pad,
                                          (long long)KC * ctx->N_pad, KC, ctx->N_pad, nsplitA));
            }
        } else if (sk.on && gvarA == 0) {
            HIP_TRY(ctx, launch_streamk_passA(st, sk, ctx->H, ctx->G_pad, ctx->X, ctx->G_pad, ctx->XHt,
                                              ctx->XHt1, ctx->N_pad, ctx->N_pad));
            spA = SplitInfo{ctx->XHt1, ctx->d_split, 128, sk.mw, sk.MG};
        } else
            HIP_TRY(ctx, launch_gemm<false>(st, gvarA, ctx->H, ctx->G_pad, ctx->X, ctx->G_pad, ctx->XHt,
                                            ctx->N_pad, (long long)KC * ctx->N_pad, KC, ctx->G_pad, ctx->N_pad, nsplitA));
        if (time_gemm) hipEventRecord(gev[gev.size() - 3], st);
        if (!spA.plane1)
            HIP_TRY(ctx, launch_reduce_splits(st, ctx->XHt, use2h ? gemm2h_splits(KbA, nsplitA, nsubA) : nsplitA,
                                              (long long)KC * ctx->N_pad, (long long)KC * ctx->N_pad));
        // W half-step                                             (sklearn _nmf.py:500)
        const bool fuse_now = fuseW && use2h;
        HIP_TRY(ctx, launch_sweep(st, nslots, ctx->Wt, ctx->N_pad, N, ctx->XHt, ctx->gramH,
                                  ctx->d_slots, l1W, ctx->gram_part, ctx->viol_part, chunksW, partsW, 1, max_k, tiers, spA,
                                  use2h ? ctx->rmaxW : nullptr, nullptr, false,
                                  fuse_now ? PlaneOut{(unsigned short*)ctx->Wt3, shW[shgen], KbB, G3_MW} : PlaneOut{nullptr, nullptr, 0, 0}));
        if (use2h) {
            const FinalizeArgs fa{ctx->gram_part, ctx->viol_part, partsW, ctx->gramW, l2H, ctx->d_slots, 0, prm->tol,
                                  prm->max_iter, 1, max_k, nullptr, 0};
            // (fused: the sweep wrote the planes; this launch checks their exponents, publishes 2^-s and the next exponents)
            HIP_TRY(ctx, launch_split2h_finalize(st, ctx->Wt, ctx->N_pad, KC, ctx->N_pad, ctx->Wt3, G3_MW, nullptr, ctx->rmaxW,
                                                 partsW, ctx->iscaleW, fa, nslots, fin_y,
                                                 fuse_now ? SplitFused{shW[shgen], shW[shgen ^ 1]} : SplitFused{nullptr, nullptr}));
            if (fuse_now) shgen ^= 1;
        } else if (use3) {
            // finalize of the W sweep + the plane split of its result in one launch (writing the planes from
            // inside the sweep was measured slower: 2-byte stores, lower occupancy)
            const FinalizeArgs fa{ctx->gram_part, ctx->viol_part, partsW, ctx->gramW, l2H, ctx->d_slots, 0, prm->tol,
                                  prm->max_iter, 1, max_k, nullptr, 0};
            HIP_TRY(ctx, launch_split3_finalize(st, ctx->Wt, ctx->N_pad, KC, ctx->N_pad, ctx->Wt3, G3_MW, nullptr, fa,
                                                nslots, fin_y));
        } else {
            finalize_kernel<<<dim3(nslots, fin_y), 256, 0, st>>>(ctx->gram_part, ctx->viol_part, partsW, ctx->gramW, l2H,
                                                    ctx->d_slots, 0, prm->tol, prm->max_iter, 1, max_k);
        }
        if (time_gemm) hipEventRecord(gev[gev.size() - 2], st);
        // pass B : XtW[S][KC][G] = Wt_all . X  (split over cells)  (sklearn _nmf.py:505-507)
        const int nsB = use2h ? gemm2h_splits(KbB, nsplit3, nsubB) : (use3 ? nsplit3 : nsplit);
        if (use2h)
            HIP_TRY(ctx, launch_gemm2h(st, ctx->Wt3, xB, xBhi, xBfl, ctx->iscaleW, KbB, ctx->XtW, ctx->G_pad,
                                       (long long)KC * ctx->G_pad, KC, ctx->G_pad, nsplit3, csB, livemask));
        else if (usec)
            HIP_TRY(ctx, launch_gemm3c(st, ctx->Wt3, ctx->Ct1, ctx->Ct1h, ctx->hiB, ctx->N_pad / 16, ctx->XtW, ctx->G_pad,
                                       (long long)KC * ctx->G_pad, KC, ctx->G_pad, nsplit3));
        else if (use3)
            HIP_TRY(ctx, launch_gemm3(st, ctx->Wt3, ctx->Xt3, ctx->N_pad / 16, ctx->XtW, ctx->G_pad,
                                      (long long)KC * ctx->G_pad, KC, ctx->G_pad, nsplit3));
        else
            HIP_TRY(ctx, launch_gemm<true>(st, gvarB, ctx->Wt, ctx->N_pad, ctx->X, ctx->G_pad, ctx->XtW,
                                           ctx->G_pad, (long long)KC * ctx->G_pad, KC, ctx->N_pad,
                                           ctx->G_pad, nsplit));
        if (time_gemm) hipEventRecord(gev[gev.size() - 1], st);
        // H half-step.  On the f16 path the split-K partials are summed (and scaled by d) inside the sweep itself.
        if (use2h && !no_psum && !(tiers & 8)) {      // (ranks above 64 take the separately reduced product)
            HIP_TRY(ctx, launch_sweep(st, nslots, ctx->H, ctx->G_pad, G, ctx->XtW, ctx->gramW,
                                      ctx->d_slots, l1H, ctx->gram_part, ctx->viol_part, chunksH, partsH, 1, max_k, tiers,
                                      psum_info(nsB, (long long)KC * ctx->G_pad, dsc), ctx->rmaxH, dsc, true));
        } else {
            HIP_TRY(ctx, launch_reduce_splits(st, ctx->XtW, nsB, (long long)KC * ctx->G_pad,
                                              (long long)KC * ctx->G_pad, usec ? dsc : nullptr, ctx->G_pad));
            HIP_TRY(ctx, launch_sweep(st, nslots, ctx->H, ctx->G_pad, G, ctx->XtW, ctx->gramW,
                                      ctx->d_slots, l1H, ctx->gram_part, ctx->viol_part, chunksH, partsH, 1, max_k, tiers,
                                      SplitInfo{nullptr, nullptr, 1, 1, 1}, use2h ? ctx->rmaxH : nullptr,
                                      use2h ? dsc : nullptr));
        }
        // the H finalize also publishes every slot's state into the host-mapped ring entry of this
        // iteration (stamp it + 1): no copy kernel and no event per iteration
        SlotDesc* snap = ctx->h_snap + (size_t)(it % RING) * KC0;
        SlotDesc* snap_dev = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&snap_dev, snap, 0));
        if (use2h) {
            const FinalizeArgs fa{ctx->gram_part, ctx->viol_part, partsH, ctx->gramH, l2W, ctx->d_slots, 1, prm->tol,
                                  prm->max_iter, 1, max_k, snap_dev, (int)(it + 1)};
            HIP_TRY(ctx, launch_split2h_finalize(st, ctx->H, ctx->G_pad, KC, ctx->G_pad, ctx->H3, G3_MW, dsc,
                                                 ctx->rmaxH, partsH, ctx->iscaleH, fa, nslots, fin_y));
            h3_valid = true;
        } else if (use3) {
            const FinalizeArgs fa{ctx->gram_part, ctx->viol_part, partsH, ctx->gramH, l2W, ctx->d_slots, 1, prm->tol,
                                  prm->max_iter, 1, max_k, snap_dev, (int)(it + 1)};
            HIP_TRY(ctx, launch_split3_finalize(st, ctx->H, ctx->G_pad, KC, ctx->G_pad, ctx->H3, G3_MW,
                                                usec ? ctx->d_scale : nullptr, fa, nslots, fin_y));
            h3_valid = true;
        } else {
            finalize_kernel<<<dim3(nslots, fin_y), 256, 0, st>>>(ctx->gram_part, ctx->viol_part, partsH, ctx->gramH, l2W,
                                                    ctx->d_slots, 1, prm->tol, prm->max_iter, 1, max_k,
                                                    snap_dev, (int)(it + 1));
        }
        HIP_TRY(ctx, hipGetLastError());
        snap_nslots[it % RING] = nslots;
        column_iters += KC;
        if (n_pending == 0) {
            int live = 0;
            for (int s2 = 0; s2 < nslots; ++s2) if (hs[s2].state) live += hs[s2].k;
            ++tail_its; tail_live += live;
        }
        if (dbg) {
            int live = 0;
            for (int s2 = 0; s2 < nslots; ++s2) if (hs[s2].state) live += hs[s2].k;
            dbg_it[KC / 32] += 1; dbg_live[KC / 32] += live;
        }
        ++it;
        return CNMF_OK;
    };

    // ---- step 3: look at the snapshot `lag` iterations behind the GPU and retire what has converged
    auto inspect = [&]() -> int {
        const int64_t si = it - 1 - lag;   // snapshot index to inspect now
        if (si < 0) return CNMF_OK;
        const SlotDesc* sp = ctx->h_snap + (size_t)(si % RING) * KC0;
        int rc2 = wait_snapshot(ctx, sp, snap_nslots[si % RING], (int)(si + 1));
        if (rc2) return rc2;
        for (int s = 0; s < snap_nslots[si % RING]; ++s)
            if (hs[s].state == 1 && hs[s].installed_at <= si && sp[s].active == 0 && sp[s].restart == hs[s].restart) {
                rc2 = retire(s, sp[s]);
                if (rc2) return rc2;
            }
        // longest-expected-first: re-sort what is still pending by the mean iteration count learned per rank (ranks
        // without a finished restart yet count as longest: they are sampled first), every 8 retirements
        if (!queue_by_rank && n_pending > 1 && n_done - last_resort_done >= 8) {
            last_resort_done = n_done;
            std::vector<int> rest;
            rest.reserve(n_pending);
            for (size_t pi = next; pi < order.size(); ++pi) if (order[pi] >= 0) rest.push_back(order[pi]);
            // expected iterations of a rank: mean over the restarts that have STARTED -- the finished ones with their count,
            // the ones in flight with their age (a lower bound).  Counting only the finished ones is biased low while it
            // matters most: of a rank whose restarts take 300 or 1000 iterations the 300s retire first.  An earlier call's
            // mean (prior) enters as four pseudo-observations.
            std::vector<double> fly_age(KMAX + 1, 0.0); std::vector<int> fly_n(KMAX + 1, 0);
            for (int s2 = 0; s2 < nslots; ++s2)
                if (hs[s2].state) { fly_age[hs[s2].k] += (double)(it - hs[s2].installed_at); fly_n[hs[s2].k] += 1; }
            auto expect = [&](int r) {
                const int k = kk[r];
                double num = (double)k_iters[k] + fly_age[k], den = (double)k_done[k] + fly_n[k];
                if (prior_of(k) > 0) { num += 4.0 * prior_of(k); den += 4.0; }
                return den > 0 ? num / den : 1e30;
            };
            for (int k = 1; k <= KMAX; ++k) {
                const double den = (double)k_done[k] + fly_n[k] + (prior_of(k) > 0 ? 4.0 : 0.0);
                if (den > 0) rank_expect[k] = ((double)k_iters[k] + fly_age[k] + 4.0 * prior_of(k)) / den;
            }
            if (dbg && n_done % 64 < 8) {
                fprintf(stderr, "[cnmf] it %lld done %d pending %d expected per rank:", (long long)it, n_done, n_pending);
                for (int k = 1; k <= KMAX; ++k)
                    if (k_done[k] + fly_n[k] > 0)
                        fprintf(stderr, " k%d=%.0f(%lld done, %d fly)", k, ((double)k_iters[k] + fly_age[k]) / ((double)k_done[k] + fly_n[k]),
                                (long long)k_done[k], fly_n[k]);
                fprintf(stderr, "\n");
            }
            std::stable_sort(rest.begin(), rest.end(), [&](int a, int b) {
                const double ea = expect(a), eb = expect(b);
                return ea != eb ? ea > eb : kk[a] > kk[b];
            });
            order.resize(next);
            order.insert(order.end(), rest.begin(), rest.end());
        }
        return CNMF_OK;
    };

    // ---- step 4: tail compaction -- nothing left to refill with and at most half of the packed
    // columns still iterate: repack the live slots into a narrower batch so the two GEMM passes
    // shrink with the work (their cost is proportional to KC).
    const bool no_compact = ctx_getenv(ctx, "CNMF_NO_COMPACT") != nullptr;
    const bool f32_tail = ctx_getenv(ctx, "CNMF_F32_TAIL") != nullptr;      // (A/B knob: compact the count path at 128 / 64 too)
    auto compact = [&]() -> int {
        if (n_pending == 0 && n_active > 0 && KC > 32 && !no_compact) {
            int live_cols = 0;
            for (int s = 0; s < nslots; ++s) if (hs[s].state) live_cols += hs[s].k;
            int KCn = 32;
            while (KCn < live_cols && KCn < 256) KCn *= 2;
            if (live_cols > 256) KCn = round_up(live_cols, 256);
            // On the count path a 256-column iteration (~250 us of GEMM at 50k x 2000) costs no more than a
            // 64-column one on the f32 pipe and far less than a 128-column one: leave it only for 32 columns.
            // General (non-count) split-operand path, 6 MFMAs per product: a 256-column iteration still beats 128
            // columns on the f32 pipe (417 / 2 vs 157 TF of roof per live column), not 64.
            // A WIDE batch (512+) narrows in steps of 256 columns and stays on the same kernels.
            bool stay3 = false;
            if (use3 && !f32_tail && KCn > (usec ? 32 : 64)) { KCn = round_up(std::max(live_cols, 1), 256); stay3 = true; }
            if (KCn == KC && use2h && !no_part && it - last_tail_repack >= 16) {
                // same width, but the live restarts lie scattered: pack them to the left so that whole 32-column
                // tiles fall dead (the GEMM passes skip those) -- when that frees at least 2 tiles and an eighth of them
                unsigned long long m = 0ull;
                for (int s = 0; s < nslots; ++s)
                    if (hs[s].state)
                        for (int t = hs[s].off / 32; t <= (hs[s].off + hs[s].k - 1) / 32 && t < 64; ++t) m |= 1ull << t;
                const int ng = KC / 256;
                int fullest = 0;
                for (int g = 0; g < ng; ++g) fullest = std::max(fullest, __builtin_popcountll((m >> (8 * g)) & 0xffull));
                const int ideal = ((live_cols + 31) / 32 + ng - 1) / ng;       // tiles per group if dealt evenly
                if (ideal <= 6 && fullest >= ideal + 2) {
                    int rcp = repack_left(KC, true);
                    if (rcp) return rcp;
                    last_tail_repack = it;
                }
            }
            if (KCn < KC) {
                last_tail_repack = it;
                int rcp = repack_left(KCn, stay3 && use2h && !no_part);
                if (rcp) return rcp;
                KC = KCn;
                cols = ColAlloc(KC);
                for (int s = 0; s < nslots; ++s) if (hs[s].state) cols.alloc(hs[s].k);
                const int cap = (ctx->nsplit_alloc * KC0) / KC;
                if (stay3) {
                    nsplit3 = std::max(1, std::min(pick_nsplit3(ctx, KC, jwA), cap));
                    sk3 = plan_streamk3(KC, ctx->N_pad, ctx->G_pad, gemm3_wg_slots(), jwA, nsubA);
                    if (sk3.on) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_split, sk3.flags.data(), sk3.flags.size(), hipMemcpyHostToDevice, st));
                    else nsplitA = std::max(1, std::min(pick_nsplit_A3(ctx, KC, jwA), (ctx->nsplitA_alloc * KC0) / KC));
                } else {
                nsplit = std::max(1, std::min(pick_nsplit(ctx, KC), cap));
                use3 = usec = use2h = use2g = false; // fewer than 256 packed columns: the f32 pipe takes over
                sk = plan_streamk(KC, ctx->N_pad, ctx->G_pad, wg_slots);
                nsplitA = (sk.on && gvarA == 0) ? 1
                        : std::max(1, std::min(pick_nsplit_A(ctx, KC), (ctx->nsplitA_alloc * KC0) / KC));
                if (sk.on) {
                    // the flags of the old plan may still be read by an in-flight sweep: same stream -> ordered
                    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_split, sk.split.data(), sk.split.size(), hipMemcpyHostToDevice, st));
                }
                }
            }
        }
        return CNMF_OK;
    };

    while (true) {
        int n_new = 0;
        if ((rc = refill(n_new))) return rc;
        if (n_active == 0 && n_pending == 0) break;
        if (n_pending == 0 && !ev_tail && stats) {
            ev_tail = events.get();
            POOL_TRY(ctx, events);
            HIP_TRY(ctx, hipEventRecord(ev_tail, st));
        }
        if ((rc = iterate(n_new))) return rc;
        if ((rc = inspect())) return rc;
        if ((rc = compact())) return rc;
    }

    if (dbg) fprintf(stderr, "[cnmf] %lld defragmentations\n", (long long)n_defrag);
    if (dbg)
        for (int i = 1; i <= 64; ++i)
            if (dbg_it[i]) fprintf(stderr, "[cnmf] KC=%d: %lld iterations, mean host-live columns %.1f\n", i * 32,
                                   (long long)dbg_it[i], (double)dbg_live[i] / dbg_it[i]);
    HIP_TRY(ctx, hipEventRecord(ev_end, st));
    if (!resident)
        HIP_TRY(ctx, hipMemcpyAsync(H_out, d_Hres, hoff[n] * sizeof(float), hipMemcpyDeviceToHost, st));
    if (W_out)
        HIP_TRY(ctx, hipMemcpyAsync(W_out, d_Wres, woff[n] * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (resident) ctx->spectra_rows += (size_t)total_k;
    {   // the mean iteration count per rank this call saw (cnmf_get_iteration_means)
        if (ctx->iter_prior.size() != (size_t)KMAX + 1) ctx->iter_prior.assign((size_t)KMAX + 1, 0.0);
        for (int k = 1; k <= KMAX; ++k)
            if (k_done[k]) ctx->iter_prior[k] = (double)k_iters[k] / (double)k_done[k];
    }
    if (stats) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev_begin, ev_end);
        stats->gpu_ms = ms;
        stats->outer_iterations = it;
        stats->restart_iterations = restart_iters;
        stats->column_iterations = column_iters;
        stats->restart_column_iterations = restart_col_iters;
        stats->kc = KC0; stats->nsplit = gemm_mode_used ? nsplit3_first : ctx->nsplit_alloc;
        stats->gemm_mode = gemm_mode_used;
        stats->tail_iterations = tail_its; stats->tail_live_columns = tail_live;
        if (ev_tail) { float tms = 0.f; hipEventElapsedTime(&tms, ev_tail, ev_end); stats->tail_ms = tms; }
        for (size_t i = 0; i + 3 < gev.size(); i += 4) {
            float a = 0.f, b = 0.f;
            hipEventElapsedTime(&a, gev[i], gev[i + 1]);
            hipEventElapsedTime(&b, gev[i + 2], gev[i + 3]);
            stats->passA_ms += a; stats->passB_ms += b;
            stats->passA_launches++; stats->passB_launches++;
        }
    }
    return CNMF_OK;
}

// mean outer iterations per rank seen by the batch calls on the resident matrix: out[KMAX + 1], 0 = rank not seen
extern "C" int cnmf_get_iteration_means(cnmf_ctx* ctx, double* out)
{
    if (!ctx || !out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    for (int k = 0; k <= KMAX; ++k) out[k] = (ctx->iter_prior.size() == (size_t)KMAX + 1) ? ctx->iter_prior[k] : 0.0;
    return CNMF_OK;
}

// expected outer iterations per rank for the queue order of the following batch calls on this matrix (n pairs; n = 0
// clears them; a new matrix clears them too)
extern "C" int cnmf_set_iteration_hints(cnmf_ctx* ctx, int n, const int32_t* k, const double* mean_iterations)
{
    if (!ctx || n < 0 || (n > 0 && (!k || !mean_iterations))) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    ctx->iter_hint.clear();
    if (n == 0) return CNMF_OK;
    ctx->iter_hint.assign((size_t)KMAX + 1, 0.0);
    for (int i = 0; i < n; ++i) {
        if (k[i] < 1 || k[i] > KMAX || !(mean_iterations[i] >= 0)) { ctx->iter_hint.clear(); SET_ERR(ctx, "bad hint %d", i); return CNMF_EINVAL; }
        ctx->iter_hint[k[i]] = mean_iterations[i];
    }
    return CNMF_OK;
}

extern "C" int cnmf_nmf_cd_batch(cnmf_ctx* ctx, int n, const int32_t* k, int init_mode,
                                 const uint32_t* seeds, const double* avg, const float* W0,
                                 const float* H0, const cnmf_cd_params* prm, float* H_out,
                                 float* W_out, int32_t* n_iter_out, double* viol_out,
                                 cnmf_batch_stats* stats)
{
    return run_batch(ctx, n, k, init_mode, seeds, avg, W0, H0, prm, H_out, W_out, false, n_iter_out, viol_out, stats);
}

extern "C" int cnmf_nmf_cd_batch_resident(cnmf_ctx* ctx, int n, const int32_t* k, int init_mode,
                                          const uint32_t* seeds, const double* avg, const float* W0,
                                          const float* H0, const cnmf_cd_params* prm,
                                          int32_t* n_iter_out, double* viol_out,
                                          cnmf_batch_stats* stats)
{
    return run_batch(ctx, n, k, init_mode, seeds, avg, W0, H0, prm, nullptr, nullptr, true, n_iter_out, viol_out, stats);
}

// ------------------------------------------------------------------ NNLS refit
extern "C" int cnmf_nnls(cnmf_ctx* ctx, int k, const float* Hin, const cnmf_cd_params* prm,
                         float* W_out, int32_t* n_iter_out, double* viol_out)
{
    if (!ctx) { SET_ERR(ctx, "ctx is NULL"); return CNMF_EINVAL; }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
    int rc = validate_params(ctx, prm);
    if (rc) return rc;
    if (!Hin || !W_out || k < 1) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    if (k > KMAX) { SET_ERR(ctx, "n_components=%d > CNMF_KMAX=%d", k, KMAX); return CNMF_EUNSUPPORTED; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int N = (int)ctx->N, G = (int)ctx->G;
    const int KC = k <= 32 ? 32 : (k <= 64 ? 64 : 128);
    rc = ensure_batch(ctx, KC, k, k);
    if (rc) return rc;
    rc = ensure_stage(ctx, (size_t)N * KMAX, (size_t)G * KMAX);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stageH, Hin, (size_t)k * G * sizeof(float), hipMemcpyHostToDevice, st));
    dim3 gI((std::max(N, G) + 255) / 256, k);
    install_kernel<<<gI, 256, 0, st>>>(ctx->stageH, nullptr, ctx->H, ctx->G_pad, G, ctx->Wt, ctx->N_pad, N, 0, k);
    if (k < KC) {   // unused component rows of the 32-wide tile must be zero
        dim3 gc((ctx->G_pad + 255) / 256, KC - k), gw((ctx->N_pad + 255) / 256, KC - k);
        clear_rows_kernel<<<gc, 256, 0, st>>>(ctx->H, ctx->G_pad, ctx->G_pad, k, KC - k);
        clear_rows_kernel<<<gw, 256, 0, st>>>(ctx->Wt, ctx->N_pad, ctx->N_pad, k, KC - k);
    }
    SlotDesc* d = &ctx->h_slots[0];
    memset(d, 0, sizeof *d);
    d->off = 0; d->k = k; d->active = 1; d->restart = 0;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_slots, d, sizeof(SlotDesc), hipMemcpyHostToDevice, st));
    ctx->h_slot_list[0] = 0;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_slot_list, ctx->h_slot_list, sizeof(int), hipMemcpyHostToDevice, st));
    gram_rows_kernel<<<1, 256, 0, st>>>(ctx->H, ctx->G_pad, G, ctx->d_slots, ctx->d_slot_list, ctx->gramH, (float)prm->l2_reg_W);
    HIP_TRY(ctx, launch_gemm<false>(st, 0, ctx->H, ctx->G_pad, ctx->X, ctx->G_pad, ctx->XHt, ctx->N_pad, 0, KC, ctx->G_pad, ctx->N_pad, 1));
    const int chunksW = sweep_chunks(N), partsW = sweep_parts(N);
    DevPool pool;
    EventPool events;
    hipEvent_t ev = events.get(hipEventDisableTiming);
    float* d_W = pool.get<float>((size_t)N * k);
    POOL_TRY(ctx, events);
    POOL_TRY(ctx, pool);
    const int burst = 8;        // sweeps enqueued between two looks at the slot state
    int done = 0;
    SlotDesc* snap = ctx->h_snap;
    for (int it = 0; it < prm->max_iter && !done; it += burst) {
        for (int b = 0; b < burst; ++b) {
            HIP_TRY(ctx, launch_sweep(st, 1, ctx->Wt, ctx->N_pad, N, ctx->XHt, ctx->gramH,
                                      ctx->d_slots, (float)prm->l1_reg_W, ctx->gram_part, ctx->viol_part,
                                      chunksW, partsW, 0, k, k <= 16 ? 1 : (k <= 32 ? 2 : (k <= KSMALL ? 4 : 8))));
            finalize_kernel<<<dim3(1, 1), 256, 0, st>>>(ctx->gram_part, ctx->viol_part, partsW, ctx->gramW, 0.f,
                                               ctx->d_slots, 2, prm->tol, prm->max_iter, 0, k);
        }
        HIP_TRY(ctx, hipMemcpyAsync(snap, ctx->d_slots, sizeof(SlotDesc), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipEventRecord(ev, st));
        HIP_TRY(ctx, hipEventSynchronize(ev));
        done = (snap->active == 0);
    }
    dim3 gW((N + 255) / 256, k);
    extract_kernel<<<gW, 256, 0, st>>>(ctx->Wt, ctx->N_pad, N, 0, k, d_W, 1);
    HIP_TRY(ctx, hipMemcpyAsync(W_out, d_W, (size_t)N * k * sizeof(float), hipMemcpyDeviceToHost, st));
    dim3 gH((ctx->G_pad + 255) / 256, k), gWc((ctx->N_pad + 255) / 256, k);
    clear_rows_kernel<<<gH, 256, 0, st>>>(ctx->H, ctx->G_pad, ctx->G_pad, 0, k);
    clear_rows_kernel<<<gWc, 256, 0, st>>>(ctx->Wt, ctx->N_pad, ctx->N_pad, 0, k);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (n_iter_out) *n_iter_out = snap->iter;
    if (viol_out) *viol_out = snap->viol_last;
    return CNMF_OK;
}
