// Host side of the GEMM / sweep kernels: launchers, stream-K plans, operand planes, count-structure
// detection (included by cnmf_hip.hip after runtime.hip.h).
#pragma once

// ------------------------------------------------------------------ GEMM dispatch
// variant : 0 = auto; 1 = "S" (waves split components, 32 j per workgroup);
//           2 = "T" (every wave owns all the workgroup's components, 128 j per workgroup)
//           3 = 2x2 wave grid (64 j per workgroup)
struct GemmPlan { int variant; int mw; int jw; };

template <int MTW, int WM, int WN, bool NN, int TBK = BK>
static hipError_t launch_gemm_t(hipStream_t st, const float* A, int lda, const float* B, int ldb,
                                float* C, int ldc, long long cstride, int KC, int Ktot, int J,
                                int nsplit)
{
    constexpr int MW = WM * MTW * 32, JW = WN * 32;
    const int Kper = round_up((Ktot + nsplit - 1) / nsplit, BK);
    dim3 grid((J + JW - 1) / JW, KC / MW, nsplit);
    constexpr size_t lds = gemm_lds_bytes<MTW, WM, WN, NN, TBK>();
    {
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm_kernel<MTW, WM, WN, NN, TBK>, (int)lds)) return e_;
    }
    gemm_kernel<MTW, WM, WN, NN, TBK><<<grid, 256, lds, st>>>(A, lda, B, ldb, C, ldc, cstride, Kper,
                                                             Ktot, J);
    return hipGetLastError();
}

// A/B knobs of the launch planning, read ONCE per process (they are not part of the per-context snapshot: cnmf_reload_env
// does not refresh them -- documented as process-lifetime in include/cnmf_hip.h)
static bool no_streamk_env() { static const bool v = getenv("CNMF_NO_STREAMK") != nullptr; return v; }
static const bool s_mtw2 = getenv("CNMF_S_MTW2") != nullptr;

template <bool NN>
static hipError_t launch_gemm(hipStream_t st, int variant, const float* A, int lda, const float* B,
                              int ldb, float* C, int ldc, long long cstride, int KC, int Ktot,
                              int J, int nsplit)
{
#define GO(MTW, WM, WN) \
    return launch_gemm_t<MTW, WM, WN, NN>(st, A, lda, B, ldb, C, ldc, cstride, KC, Ktot, J, nsplit)
    if (variant == 0) variant = 2;
    if (variant == 1 && KC < 128) variant = (KC >= 64) ? 3 : 2;
    if (variant == 3 && KC < 64) variant = 2;
    switch (variant) {
        case 1:  // S: 4 waves x (MTW tiles of 32 comps), 32 j
            if (KC % 256 == 0 && KC >= 256 && s_mtw2) GO(2, 4, 1);
            GO(1, 4, 1);
        case 3:  // 2x2
            if (KC % 128 == 0) GO(2, 2, 2);
            GO(1, 2, 2);
        default:  // T: every wave all comps of the M group, 128 j
            if (KC % 128 == 0) GO(4, 1, 4);
            if (KC % 64 == 0) GO(2, 1, 4);
            GO(1, 1, 4);
    }
#undef GO
}

// ------------------------------------------------------------------ sweep dispatch
// SplitInfo of a PSUM sweep (kernels_sweep.hip.h): `nsplit` partial planes `stride` floats apart, scaled per row by
// `colscale` (nullable).  The fields are re-used: mgroups = nsplit, tile_rows / tile_cols = stride, split = colscale.
static SplitInfo psum_info(int nsplit, long long stride, const double* colscale)
{
    SplitInfo sp{nullptr, reinterpret_cast<const unsigned char*>(colscale), (int)(stride >> 20), (int)(stride & ((1 << 20) - 1)), nsplit};
    return sp;
}

static hipError_t launch_sweep(hipStream_t st, int nslots, float* V, int ldv, int L, const float* P,
                               const float* gram, const SlotDesc* slots, float l1, float* gram_part,
                               double* viol_part, int chunks, int parts, int want_gram, int kmax, int tiers,
                               SplitInfo sp = SplitInfo{nullptr, nullptr, 1, 1, 1},
                               float* rmax_part = nullptr, const double* rmax_scale = nullptr, bool psum = false,
                               PlaneOut po = PlaneOut{nullptr, nullptr, 0, 0})
{
    // `parts` = partials per slot = workgroups per slot; a workgroup walks `chunks` 256-row tiles
    dim3 grid(parts, nslots);
    if (po.dst) {
        // the W half-step of the f16 paths, planes written by the sweep itself (ranks <= 64 only: the caller checks)
        if (psum || (rmax_part && rmax_scale) || (tiers & 8)) return hipErrorInvalidValue;
        const size_t pl = sweep_lds_bytes(kmax, true);
        const int kgp = sweep_kg(kmax);
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<0, false, false, true>, (int)sweep_lds_bytes(KSMALL, true))) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<1, false, false, true>, (int)sweep_lds_bytes(KSMALL, true))) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<2, false, false, true>, (int)sweep_lds_bytes(KSMALL, true))) return e_;
#define CNMF_SWEEP_PLN(T_) sweep_kernel<T_, false, false, true><<<grid, 256, pl, st>>>(V, ldv, L, P, sp, gram, slots, l1, gram_part, viol_part, chunks, want_gram, kgp, kmax, rmax_part, nullptr, po)
        if (tiers & 1) CNMF_SWEEP_PLN(0);
        if (tiers & 2) CNMF_SWEEP_PLN(1);
        if (tiers & 4) CNMF_SWEEP_PLN(2);
#undef CNMF_SWEEP_PLN
        return hipGetLastError();
    }
    {      // ranks above 32 need more than the default 64 KB of dynamic LDS
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<0, false>, (int)sweep_lds_bytes(KSMALL))) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<1, false>, (int)sweep_lds_bytes(KSMALL))) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<2, false>, (int)sweep_lds_bytes(KSMALL))) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<0, true>, (int)sweep_lds_bytes(KSMALL))) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<1, true>, (int)sweep_lds_bytes(KSMALL))) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<2, true>, (int)sweep_lds_bytes(KSMALL))) return e_;
    }
    // one launch per rank tier present among the live slots; a launch skips the slots of other tiers at once.
    // rmax_scale != nullptr selects the exact row-maximum report (the H half-step of the f16 plane split).
    const size_t lds = sweep_lds_bytes(kmax);
    const int kg = sweep_kg(kmax);
#define CNMF_SWEEP(T_, R_) sweep_kernel<T_, R_><<<grid, 256, lds, st>>>(V, ldv, L, P, sp, gram, slots, l1, gram_part, viol_part, chunks, want_gram, kg, kmax, rmax_part, rmax_scale)
#define CNMF_SWEEP_PSUM(T_) sweep_kernel<T_, true, true><<<grid, 256, lds, st>>>(V, ldv, L, P, sp, gram, slots, l1, gram_part, viol_part, chunks, want_gram, kg, kmax, rmax_part, rmax_scale)
    if (psum) {                 // split-K partial planes summed (and column-scaled) inside the sweep: sp = psum_info(...)
        {
            if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<0, true, true>, (int)sweep_lds_bytes(KSMALL))) return e_;
            if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<1, true, true>, (int)sweep_lds_bytes(KSMALL))) return e_;
            if (hipError_t e_ = dyn_lds_optin((const void*)sweep_kernel<2, true, true>, (int)sweep_lds_bytes(KSMALL))) return e_;
        }
        if (tiers & 1) CNMF_SWEEP_PSUM(0);
        if (tiers & 2) CNMF_SWEEP_PSUM(1);
        if (tiers & 4) CNMF_SWEEP_PSUM(2);
    } else if (rmax_part && rmax_scale) {
        if (tiers & 1) CNMF_SWEEP(0, true);
        if (tiers & 2) CNMF_SWEEP(1, true);
        if (tiers & 4) CNMF_SWEEP(2, true);
    } else {
        if (tiers & 1) CNMF_SWEEP(0, false);
        if (tiers & 2) CNMF_SWEEP(1, false);
        if (tiers & 4) CNMF_SWEEP(2, false);
    }
#undef CNMF_SWEEP_PSUM
#undef CNMF_SWEEP
    if (tiers & 8) {            // ranks 65..128: sweep_big_kernel (+ the Gram of the updated rows as its own launch)
        if (psum) return hipErrorInvalidValue;                 // the caller reduces the split-K planes first
        const size_t blds = sweep_big_lds_bytes();
        const int bchunks = chunks;
        const dim3 gbig(parts, nslots);
        if (rmax_part && rmax_scale) {
            if (hipError_t e_ = dyn_lds_optin((const void*)sweep_big_kernel<true>, (int)blds)) return e_;
            sweep_big_kernel<true><<<gbig, 256, blds, st>>>(V, ldv, L, P, sp, gram, slots, l1, viol_part, bchunks, rmax_part, rmax_scale);
        } else {
            if (hipError_t e_ = dyn_lds_optin((const void*)sweep_big_kernel<false>, (int)blds)) return e_;
            sweep_big_kernel<false><<<gbig, 256, blds, st>>>(V, ldv, L, P, sp, gram, slots, l1, viol_part, bchunks, nullptr, nullptr);
        }
        if (want_gram)
            gram_big_kernel<<<gbig, 256, 0, st>>>(V, ldv, L, slots, gram_part, bchunks, kmax, (rmax_part && !rmax_scale) ? rmax_part : nullptr);
    }
    return hipGetLastError();
}

// ---- stream-K pass A (T layout: 128 components x 128 cells per tile)
struct StreamK {
    bool on = false;
    int MG = 1, T = 0, nk = 0, P = 0, mw = 128;      // mw: component rows per workgroup tile
    std::vector<unsigned char> split;
};

static StreamK plan_streamk(int KC, int N_pad, int G_pad, int n_wg_slots)
{
    StreamK sk;
    if (KC % 128 != 0 || no_streamk_env()) return sk;
    sk.MG = KC / sk.mw;
    sk.T = sk.MG * (N_pad / 128);
    sk.nk = G_pad / BK;                                    // stages per tile, as the kernel counts them
    sk.P = n_wg_slots;
    if (sk.T <= sk.P) sk.P = n_wg_slots / 2;              // one workgroup per CU
    if (sk.T <= sk.P || sk.T % sk.P == 0) return sk;      // nothing to balance
    sk.on = true;
    sk.split.assign(sk.T, 0);
    const long long U = (long long)sk.T * sk.nk;
    for (int p = 1; p < sk.P; ++p) {
        const long long b = U * p / sk.P;                  // first unit of workgroup p
        if (b % sk.nk) sk.split[b / sk.nk] = 1;            // boundary inside a tile -> that tile is cut
    }
    return sk;
}

static hipError_t launch_streamk_passA(hipStream_t st, const StreamK& sk, const float* A, int lda,
                                       const float* B, int ldb, float* C0, float* C1, int ldc, int Jtot)
{
    constexpr size_t lds = gemm_lds_bytes<4, 1, 4, false>();
    {
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm_streamk_kernel<4, 1, 4, false>, (int)lds)) return e_;
    }
    gemm_streamk_kernel<4, 1, 4, false><<<sk.P, 256, lds, st>>>(A, lda, B, ldb, C0, C1, ldc, sk.MG, sk.T,
                                                               sk.nk, Jtot);
    return hipGetLastError();
}

static hipError_t launch_reduce_splits(hipStream_t st, float* P, int nsplit, long long split_stride,
                                       long long n_floats, const double* colscale = nullptr, int ld = 1)
{
    if (nsplit <= 1 && !colscale) return hipSuccess;
    const long long nv = n_floats / 4;
    reduce_splits_kernel<<<(unsigned)((nv + 255) / 256), 256, 0, st>>>(P, nsplit, split_stride, P, nv, colscale, ld);
    return hipGetLastError();
}


// ------------------------------------------------------------------ split-operand GEMM launchers
// planes of a K-contiguous f32 matrix, block-major with row tiles of TR rows (rows % TR == 0)
static hipError_t launch_split3(hipStream_t st, const float* src, int ld, int rows, int K, unsigned char* dst, int TR,
                                const double* kscale = nullptr)
{
    if (rows % 64 == 0 && K % 64 == 0 && TR % 64 == 0) {        // tiled through LDS: both sides coalesced
        dim3 grid(K / 64, rows / 64);
        split3_tiled_kernel<<<grid, 256, 0, st>>>(src, ld, K, TR, (unsigned short*)dst, kscale);
        return hipGetLastError();
    }
    const long long total = (long long)rows * (K / 16);
    split3_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(src, ld, rows, K, TR, (unsigned short*)dst, kscale);
    return hipGetLastError();
}

// CNMF_GEMM3: 0 = exact-f32 matrix pipe only, 1 = split-operand bf16 path, two register-staged 4-wave
// workgroups per CU (the simple reference variant), 2 = split-operand bf16 path, one 8-wave LDS-DMA
// ping-pong workgroup per CU; 3 (default) = 2, plus the count-structured path (one integer plane for X, 3 MFMAs
// per product on 256 x 256 tiles) whenever the resident matrix has that structure; 4 (default) = 3 on the f16 matrix
// pipe (kernels_gemm2h.hip.h): counts <= 2048 in one f16 plane, the factor as two f16 planes with a per-row exponent,
// 2 MFMAs per product.  Read at every call from the context's snapshot (cnmf_reload_env re-reads it): tests switch it.
// (Tried and dropped, all within 3 % of variant 2 at the 50k x 2000 shape: the same ping-pong with register
//  staging; 256 x 256 tiles with the two wave groups half a block apart (2/3 of the DMA bytes per flop).)
static thread_local int g_gemm3_mode = CNMF_GEMM3_DEFAULT;
static thread_local int g_g2_gvar = 4;      // instruction stream of the general-matrix GEMMs (CNMF_G2_GVAR, refreshed below)
static void refresh_gemm3_mode(const cnmf_ctx* ctx)   // at every API entry that launches GEMMs (not inside the hot loop): from the
{                                                      // context's snapshot of the environment, like every per-call switch
    const char* e = ctx_getenv(ctx, "CNMF_GEMM3");
    const int mode = e ? atoi(e) : CNMF_GEMM3_DEFAULT;
    g_gemm3_mode = (mode < 0 || mode > 4) ? CNMF_GEMM3_DEFAULT : mode;
    const char* gv = ctx_getenv(ctx, "CNMF_G2_GVAR");
    g_g2_gvar = (gv && atoi(gv) == 0) ? 0 : 4;
}
static int gemm3_mode() { return g_gemm3_mode; }
// (CNMF_WG_SLOTS: fewer persistent GEMM workgroups than CUs -- leaves whole CUs to the kernels of another stream)
static int gemm3_wg_slots()
{
    static const int env = getenv("CNMF_WG_SLOTS") ? atoi(getenv("CNMF_WG_SLOTS")) : 0;
    if (env >= 32 && gemm3_mode() >= 2) return env;
    return gemm3_mode() >= 2 ? 256 : 512;
}
static int gemm3_jw() { return G3_JW; }     // j extent of a tile = row tile of the B planes

static hipError_t launch_gemm3(hipStream_t st, const unsigned char* A3, const unsigned char* B3, int Kb,
                               float* C, int ldc, long long cstride, int KC, int Jpad, int nsplit)
{
    {
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm3_kernel, G3_LDS_BYTES)) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm3g_kernel, G3G_LDS_BYTES)) return e_;
    }
    const int kb_per = (Kb + nsplit - 1) / nsplit;
    dim3 grid(Jpad / gemm3_jw(), KC / G3_MW, (Kb + kb_per - 1) / kb_per);
    if (gemm3_mode() >= 2)
        gemm3g_kernel<<<grid, 512, G3G_LDS_BYTES, st>>>(A3, B3, Kb, C, ldc, cstride, kb_per);
    else
        gemm3_kernel<<<grid, 256, G3_LDS_BYTES, st>>>(A3, B3, Kb, C, ldc, cstride, kb_per);
    return hipGetLastError();
}


// plane split of a packed factor + finalize of the sweep that produced it, in one launch (kernels_sweep.hip.h)
static hipError_t launch_split3_finalize(hipStream_t st, const float* src, int ld, int rows, int K, unsigned char* dst,
                                         int TR, const double* kscale, const FinalizeArgs& fa, int nslots, int fin_y)
{
    const int bx = K / 64, by = rows / 64;
    split3_finalize_kernel<<<bx * by + nslots * fin_y, 256, 0, st>>>(src, ld, K, TR, (unsigned short*)dst, kscale, bx, by,
                                                                  fa, fin_y);
    return hipGetLastError();
}

static hipError_t launch_split2h_finalize(hipStream_t st, const float* src, int ld, int rows, int K, unsigned char* dst,
                                          int TR, const double* kscale, const float* rmax_part, int parts,
                                          float* inv_scale, const FinalizeArgs& fa, int nslots, int fin_y,
                                          SplitFused fu = SplitFused{nullptr, nullptr});

// ---- stream-K plan for the split-operand pass A (tile = 256 components x 128 cells, up to 2 cuts per tile)
struct StreamK3 {
    bool on = false;
    int T = 0, Kb = 0, P = 0, MG = 1;     // Kb: work units per tile (16-k blocks / `unit`)
    std::vector<unsigned char> flags;     // bit 0: >= 1 cut (plane 1 holds the tail), bit 1: 2 cuts (plane 2 the middle)
};

static StreamK3 plan_streamk3(int KC, int N_pad, int G_pad, int n_wg_slots, int jw, int unit = 1)
{
    StreamK3 sk;
    sk.MG = KC / G3_MW;
    sk.T = sk.MG * (N_pad / jw);
    sk.Kb = G_pad / (G3_BK * unit);
    sk.P = n_wg_slots;
    if (sk.T < sk.P / 2 + sk.P / 4 || sk.P > 2 * sk.T || no_streamk_env()) return sk;   // few tiles: K split + reduce instead
    sk.on = true;
    sk.flags.assign(sk.T, 0);
    const long long U = (long long)sk.T * sk.Kb;
    // the kernels walk the tiles component-group-major (o = mg * NJ + jt); the sweep looks the flags up by jt * MG + mg
    const int NJ = sk.T / sk.MG;
    for (int p = 1; p < sk.P; ++p) {
        const long long b = U * p / sk.P;
        if (b % sk.Kb) {
            const int o = (int)(b / sk.Kb);
            unsigned char& f = sk.flags[(o % NJ) * sk.MG + o / NJ];
            f = f ? 3 : 1;
        }
    }
    return sk;
}

static hipError_t launch_gemm3_streamk(hipStream_t st, const StreamK3& sk, const unsigned char* A3,
                                       const unsigned char* B3, float* C0, float* C1, float* C2, int ldc)
{
    {
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm3_streamk_kernel, G3_LDS_BYTES)) return e_;
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm3g_streamk_kernel, G3G_LDS_BYTES)) return e_;
    }
    if (gemm3_mode() >= 2)
        gemm3g_streamk_kernel<<<sk.P, 512, G3G_LDS_BYTES, st>>>(A3, B3, sk.Kb, C0, C1, C2, ldc, sk.MG, sk.T);
    else
        gemm3_streamk_kernel<<<sk.P, 256, G3_LDS_BYTES, st>>>(A3, B3, sk.Kb, C0, C1, C2, ldc, sk.MG, sk.T);
    return hipGetLastError();
}

// planes of X (pass A) and of X^T (pass B), built once per matrix on first use
static int ensure_planes(cnmf_ctx* ctx)
{
    const int TR = gemm3_jw();
    if (ctx->X3 && ctx->Xt3 && ctx->planes_tr == TR) return CNMF_OK;
    hipFree(ctx->X3); hipFree(ctx->Xt3); ctx->X3 = ctx->Xt3 = nullptr;
    ctx->planes_tr = TR;
    const size_t bA = (size_t)ctx->N_pad * (ctx->G_pad / 16) * G3_ROWB;
    const size_t bB = (size_t)ctx->G_pad * (ctx->N_pad / 16) * G3_ROWB;
    HIP_TRY(ctx, hipMalloc(&ctx->X3, bA));
    HIP_TRY(ctx, hipMalloc(&ctx->Xt3, bB));
    HIP_TRY(ctx, launch_split3(ctx->stream, ctx->X, ctx->G_pad, ctx->N_pad, ctx->G_pad, ctx->X3, TR));
    dim3 grid(ctx->N_pad / 16, (ctx->G_pad + 255) / 256);
    split3_transpose_kernel<<<grid, 256, 0, ctx->stream>>>(ctx->X, ctx->G_pad, ctx->N_pad, ctx->G_pad, ctx->N_pad,
                                                            TR, (unsigned short*)ctx->Xt3);
    HIP_TRY(ctx, hipGetLastError());
    return CNMF_OK;
}

// planes of a general (not count-structured) matrix for the f16 pipe (gemm_mode 5, kernels_gemm2h.hip.h)
static int ensure_x2planes(cnmf_ctx* ctx)
{
    if (ctx->X2h) return CNMF_OK;
    hipStream_t st = ctx->stream;
    const int N = (int)ctx->N, G = (int)ctx->G, Np = ctx->N_pad, Gp = ctx->G_pad;
    const size_t pb = (size_t)Np * Gp * 2;
    const size_t fa = (size_t)(Np / G3C_JW) * ((Gp / 16 + 31) / 32) * sizeof(unsigned), fb = (size_t)(Gp / G3C_JW) * ((Np / 16 + 31) / 32) * sizeof(unsigned);
    DevPool pool;
    int* shA = pool.get<int>(Np);
    int* shB = pool.get<int>(Gp);
    POOL_TRY(ctx, pool);
    unsigned char *a = nullptr, *am = nullptr, *b = nullptr, *bm = nullptr;
    float *sa = nullptr, *sb = nullptr;
    unsigned *oa = nullptr, *ob = nullptr;
    hipError_t e = hipMalloc(&a, pb);
    if (e == hipSuccess) e = hipMalloc(&am, pb);
    if (e == hipSuccess) e = hipMalloc(&b, pb);
    if (e == hipSuccess) e = hipMalloc(&bm, pb);
    if (e == hipSuccess) e = hipMalloc(&sa, (size_t)Np * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&sb, (size_t)Gp * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&oa, fa);
    if (e == hipSuccess) e = hipMalloc(&ob, fb);
    if (e == hipSuccess) e = hipMemsetAsync(oa, 0xff, fa, st);
    if (e == hipSuccess) e = hipMemsetAsync(ob, 0xff, fb, st);
    if (e == hipSuccess) {
        x2h_rowshift_kernel<<<(Np + 3) / 4, 256, 0, st>>>(ctx->X, Gp, N, G, 0, Np, shA, sa);
        x2h_rowshift_kernel<<<(Gp + 255) / 256, 256, 0, st>>>(ctx->X, Gp, N, G, 1, Gp, shB, sb);
        const long long total = (long long)Np * (Gp / 16);
        x2h_planes_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(ctx->X, Gp, N, G, Np, Gp, G3C_JW, shA,
                                                                            (unsigned short*)a, (unsigned short*)am);
        x2h_planes_transpose_kernel<<<dim3(Np / 16, (Gp + 255) / 256), 256, 0, st>>>(ctx->X, Gp, N, G, Gp, Np, G3C_JW, shB,
                                                                                    (unsigned short*)b, (unsigned short*)bm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);      // the shift scratch is freed on return
    if (e != hipSuccess) {
        hipFree(a); hipFree(am); hipFree(b); hipFree(bm); hipFree(sa); hipFree(sb); hipFree(oa); hipFree(ob);
        HIP_TRY(ctx, e);
    }
    ctx->X2h = a; ctx->X2m = am; ctx->Xt2h = b; ctx->Xt2m = bm; ctx->x2sA = sa; ctx->x2sB = sb; ctx->onesA = oa; ctx->onesB = ob;
    return CNMF_OK;
}

// ---- count-structured data: launchers of the 256 x 256 integer-plane kernel
// Bhi / hiflag: second integer plane and its block flags (nullptr when no count exceeds 256)
static hipError_t launch_gemm3c(hipStream_t st, const unsigned char* A3, const unsigned char* B1,
                                const unsigned char* Bhi, const unsigned int* hiflag, int Kb,
                                float* C, int ldc, long long cstride, int KC, int Jpad, int nsplit)
{
    {
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm3c_kernel, g3c_lds_bytes(true))) return e_;
    }
    const int kb_per = (Kb + nsplit - 1) / nsplit;
    dim3 grid(Jpad / G3C_JW, KC / G3_MW, (Kb + kb_per - 1) / kb_per);
    gemm3c_kernel<<<grid, 512, g3c_lds_bytes(Bhi != nullptr), st>>>(A3, B1, Bhi, hiflag, Kb, C, ldc, cstride, kb_per);
    return hipGetLastError();
}

static hipError_t launch_gemm3c_streamk(hipStream_t st, const StreamK3& sk, const unsigned char* A3,
                                        const unsigned char* B1, const unsigned char* Bhi,
                                        const unsigned int* hiflag, float* C0, float* C1, float* C2, int ldc)
{
    {
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm3c_streamk_kernel, g3c_lds_bytes(true))) return e_;
    }
    gemm3c_streamk_kernel<<<sk.P, 512, g3c_lds_bytes(Bhi != nullptr), st>>>(A3, B1, Bhi, hiflag, sk.Kb, C0, C1, C2, ldc,
                                                                            sk.MG, sk.T);
    return hipGetLastError();
}

// ---- f16 two-plane count path (kernels_gemm2h.hip.h)
// 16-k sub-blocks per barrier pair of the production launches (CNMF_G2_NSUB = 1 | 2)
static int g2_nsub()
{
    static const int v = getenv("CNMF_G2_NSUB") ? atoi(getenv("CNMF_G2_NSUB")) : 2;
    return v == 1 ? 1 : 2;
}
// instruction-stream variant of the NSUB = 2 kernels (kernels_gemm2h.hip.h): bit 0 = DMA pieces spread through the
// MFMA stream, bit 1 = s_setprio around the MFMA halves
#ifndef CNMF_G2_VAR_DEFAULT
#define CNMF_G2_VAR_DEFAULT 4
#endif
// general matrices (GEN): 4 = the spread instruction stream (round 6, default), 0 = the burst loop (A/B; same bits)
static int g2_gvar() { return g_g2_gvar; }
static int g2_var()
{
    static const int v = getenv("CNMF_G2_VAR") ? atoi(getenv("CNMF_G2_VAR")) : CNMF_G2_VAR_DEFAULT;
    return (v >= 0 && v <= 5) ? v : CNMF_G2_VAR_DEFAULT;
}

static inline unsigned long long g2_full_mask(int KC) { const int t = KC / 32; return t >= 64 ? ~0ull : ((1ull << t) - 1ull); }

template <int NSUB, bool HI, int VAR = 0, bool PART = false, bool GEN = false>
static hipError_t launch_gemm2h_t(hipStream_t st, const unsigned char* A2, const unsigned char* B1,
                                  const unsigned char* Bhi, const unsigned int* hiflag, const float* rscale, int Kb,
                                  float* C, int ldc, long long cstride, int KC, int Jpad, int nsplit,
                                  const float* cscale = nullptr, unsigned long long livemask = ~0ull)
{
    constexpr int lds = g2_lds_bytes(NSUB, HI);
    {
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm2h_kernel<NSUB, HI, VAR, PART, GEN>, lds)) return e_;
    }
    int kb_per = (Kb + nsplit - 1) / nsplit;
    kb_per = ((kb_per + NSUB - 1) / NSUB) * NSUB;            // whole steps
    dim3 grid(Jpad / G3C_JW, KC / G3_MW, (Kb + kb_per - 1) / kb_per);
    gemm2h_kernel<NSUB, HI, VAR, PART, GEN><<<grid, 512, lds, st>>>(A2, B1, Bhi, hiflag, rscale, Kb, C, ldc, cstride, kb_per, cscale, livemask);
    return hipGetLastError();
}

// number of K splits the launch above will actually use (the partial planes the reduce must add)
static int gemm2h_splits(int Kb, int nsplit, int nsub)
{
    int kb_per = (Kb + nsplit - 1) / nsplit;
    kb_per = ((kb_per + nsub - 1) / nsub) * nsub;
    return (Kb + kb_per - 1) / kb_per;
}

static hipError_t launch_gemm2h(hipStream_t st, const unsigned char* A2, const unsigned char* B1,
                                const unsigned char* Bhi, const unsigned int* hiflag, const float* rscale, int Kb,
                                float* C, int ldc, long long cstride, int KC, int Jpad, int nsplit,
                                const float* cscale = nullptr, unsigned long long livemask = ~0ull)
{
    // livemask: bit t = the 32 packed columns 32 t .. 32 t + 31 hold a restart that still iterates (all ones: everything)
    static const bool nostore = getenv("CNMF_G2_NOSTORE") != nullptr;      // timing ablation: results meaningless
    if (nostore) C = nullptr;
    const bool part = (livemask & g2_full_mask(KC)) != g2_full_mask(KC);
    // general matrices (cscale != nullptr: X as two f16 planes) multiply three of the four plane pairs (GEN); CNMF_G2_GEN4=1: all four (A/B)
    static const bool gen4 = getenv("CNMF_G2_GEN4") != nullptr;
    if (Bhi && cscale && !gen4) {
        if (g2_gvar() == 0)                 // (A/B: the burst loop of rounds 3-5; bit-identical results)
            return part ? launch_gemm2h_t<1, true, 0, true, true>(st, A2, B1, Bhi, hiflag, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit, cscale, livemask)
                        : launch_gemm2h_t<1, true, 0, false, true>(st, A2, B1, Bhi, hiflag, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit, cscale);
        return part ? launch_gemm2h_t<1, true, 4, true, true>(st, A2, B1, Bhi, hiflag, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit, cscale, livemask)
                    : launch_gemm2h_t<1, true, 4, false, true>(st, A2, B1, Bhi, hiflag, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit, cscale);
    }
    if (Bhi) return part ? launch_gemm2h_t<1, true, 0, true>(st, A2, B1, Bhi, hiflag, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit, cscale, livemask)
                         : launch_gemm2h_t<1, true>(st, A2, B1, Bhi, hiflag, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit, cscale);
    if (cscale) return hipErrorInvalidValue;              // a column scale exists only on the two-plane operand path
    if (g2_nsub() == 2 && Kb % 2 == 0) {
        if (part && g2_var() == CNMF_G2_VAR_DEFAULT)
            return launch_gemm2h_t<2, false, CNMF_G2_VAR_DEFAULT, true>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit, nullptr, livemask);
        switch (g2_var()) {
            case 1: return launch_gemm2h_t<2, false, 1>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit);
            case 2: return launch_gemm2h_t<2, false, 2>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit);
            case 3: return launch_gemm2h_t<2, false, 3>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit);
            case 4: return launch_gemm2h_t<2, false, 4>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit);
            case 5: return launch_gemm2h_t<2, false, 5>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit);
            default: return launch_gemm2h_t<2, false, 0>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit);
        }
    }
    if (g2_var() >= 1) return launch_gemm2h_t<1, false, 4>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit);
    return launch_gemm2h_t<1, false>(st, A2, B1, nullptr, nullptr, rscale, Kb, C, ldc, cstride, KC, Jpad, nsplit);
}
static int gemm2h_nsub(bool hi, int Kb) { return (!hi && g2_nsub() == 2 && Kb % 2 == 0) ? 2 : 1; }

template <int NSUB, bool HI, int VAR = 0, bool NTB = true, bool PART = false, bool GEN = false>
static hipError_t launch_gemm2h_streamk_t(hipStream_t st, const StreamK3& sk, const unsigned char* A2,
                                          const unsigned char* B1, const unsigned char* Bhi,
                                          const unsigned int* hiflag, const float* rscale, int Kb, float* C0, float* C1,
                                          float* C2, int ldc, const float* cscale = nullptr,
                                          unsigned long long livemask = ~0ull)
{
    constexpr int lds = g2_lds_bytes(NSUB, HI);
    {
        if (hipError_t e_ = dyn_lds_optin((const void*)gemm2h_streamk_kernel<NSUB, HI, VAR, NTB, PART, GEN>, lds)) return e_;
    }
    // XCD mapping of the persistent workgroups (kernel comment).  Count path: every XCD inside ONE component group (1).  General
    // path (GEN: both X planes, 410 MB at 50 000 x 2 000, re-streamed per component group): the identity order (0) -- the four
    // groups of a row tile share it in one L2 -- measured +1.3 % (133.7 / 134.1 vs 131.9 / 132.4 restarts/s, round 6); the
    // count path measured -2 % with it (round 3).  CNMF_G2_XMAP=0|1 forces either.
    static const int xmap_env = getenv("CNMF_G2_XMAP") ? atoi(getenv("CNMF_G2_XMAP")) : -1;
    const int xmap = xmap_env >= 0 ? xmap_env : (GEN ? 0 : 1);
    gemm2h_streamk_kernel<NSUB, HI, VAR, NTB, PART, GEN><<<sk.P, 512, lds, st>>>(A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, sk.MG, sk.T, xmap, cscale, livemask);
    return hipGetLastError();
}

// (the plan `sk` must have been made with unit = gemm2h_nsub(Bhi != nullptr, Kb))
static hipError_t launch_gemm2h_streamk(hipStream_t st, const StreamK3& sk, const unsigned char* A2,
                                        const unsigned char* B1, const unsigned char* Bhi, const unsigned int* hiflag,
                                        const float* rscale, int Kb, float* C0, float* C1, float* C2, int ldc,
                                        const float* cscale = nullptr, unsigned long long livemask = ~0ull)
{
    static const bool nostore = getenv("CNMF_G2_NOSTORE") != nullptr;      // timing ablation: results meaningless
    if (nostore) C0 = C1 = C2 = nullptr;
    const bool part = (livemask & g2_full_mask(sk.MG * G3_MW)) != g2_full_mask(sk.MG * G3_MW);
    // several component groups share every count-plane tile in the L2: no non-temporal loads then (CNMF_G2_NT=1: A/B)
    static const bool force_nt = getenv("CNMF_G2_NT") != nullptr;
    const bool shared = sk.MG > 1 && !force_nt;
    static const bool gen4 = getenv("CNMF_G2_GEN4") != nullptr;
    if (Bhi && cscale && !gen4) {          // general matrices: three of the four plane pairs (GEN)
        if (g2_gvar() == 0) {              // (A/B: the burst loop of rounds 3-5; bit-identical results)
            if (part) return launch_gemm2h_streamk_t<1, true, 0, false, true, true>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale, livemask);
            return shared ? launch_gemm2h_streamk_t<1, true, 0, false, false, true>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale)
                          : launch_gemm2h_streamk_t<1, true, 0, true, false, true>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale);
        }
        if (part) return launch_gemm2h_streamk_t<1, true, 4, false, true, true>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale, livemask);
        return shared ? launch_gemm2h_streamk_t<1, true, 4, false, false, true>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale)
                      : launch_gemm2h_streamk_t<1, true, 4, true, false, true>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale);
    }
    if (Bhi) {
        if (part) return launch_gemm2h_streamk_t<1, true, 0, false, true>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale, livemask);
        return shared ? launch_gemm2h_streamk_t<1, true, 0, false>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale)
                      : launch_gemm2h_streamk_t<1, true>(st, sk, A2, B1, Bhi, hiflag, rscale, Kb, C0, C1, C2, ldc, cscale);
    }
    if (cscale) return hipErrorInvalidValue;
    if (gemm2h_nsub(false, Kb) == 2) {
        if (part && g2_var() == CNMF_G2_VAR_DEFAULT)
            return launch_gemm2h_streamk_t<2, false, CNMF_G2_VAR_DEFAULT, false, true>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc, nullptr, livemask);
        if (shared && g2_var() == CNMF_G2_VAR_DEFAULT)
            return launch_gemm2h_streamk_t<2, false, CNMF_G2_VAR_DEFAULT, false>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
        switch (g2_var()) {
            case 1: return launch_gemm2h_streamk_t<2, false, 1>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
            case 2: return launch_gemm2h_streamk_t<2, false, 2>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
            case 3: return launch_gemm2h_streamk_t<2, false, 3>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
            case 4: return launch_gemm2h_streamk_t<2, false, 4>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
            case 5: return launch_gemm2h_streamk_t<2, false, 5>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
            default: return launch_gemm2h_streamk_t<2, false, 0>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
        }
    }
    if (g2_var() >= 1)                     // (CNMF_G2_NSUB=1 A/B: the one-block spread stream on the count plane)
        return shared ? launch_gemm2h_streamk_t<1, false, 4, false>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc)
                      : launch_gemm2h_streamk_t<1, false, 4, true>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
    return launch_gemm2h_streamk_t<1, false>(st, sk, A2, B1, nullptr, nullptr, rscale, Kb, C0, C1, C2, ldc);
}

// geometry of the split launch: tiles of 16 rows x 256 k when K % 256 == 0 (1 KB contiguous per row), else 64 x 64;
// every workgroup walks its k tiles with stride `groups`, so that all workgroups are resident at once (<= 7 per CU)
// and the row scale is reduced once per workgroup
static bool split2h_wide(int K) { return K % 256 == 0; }
static int split2h_col_groups(int K, int rows)
{
    const int tk = split2h_wide(K) ? 256 : 64, trows = split2h_wide(K) ? 16 : 64;
    const int tiles = K / tk, rg = std::max(1, rows / trows);
    const int cap = std::max(1, (256 * 7) / rg);
    int per = (tiles + cap - 1) / cap;                     // k tiles per workgroup
    return (tiles + per - 1) / per;
}

// f16 planes of a packed factor (rows, K multiples of 64) from the sweep's row-maximum partials
static hipError_t launch_split2h(hipStream_t st, const float* src, int ld, int rows, int K, unsigned char* dst, int TR,
                                 const double* kscale, const float* rmax_part, int parts, float* inv_scale)
{
    if (split2h_wide(K)) {
        dim3 grid(split2h_col_groups(K, rows), rows / 16);
        split2h_tiled_kernel<16, 16><<<grid, 256, 0, st>>>(src, ld, K, TR, (unsigned short*)dst, kscale, rmax_part, parts, inv_scale);
    } else {
        dim3 grid(split2h_col_groups(K, rows), rows / 64);
        split2h_tiled_kernel<64, 4><<<grid, 256, 0, st>>>(src, ld, K, TR, (unsigned short*)dst, kscale, rmax_part, parts, inv_scale);
    }
    return hipGetLastError();
}

// row-maximum partials of rows the sweep has not produced (installed / moved slots): [rows][parts]
static hipError_t launch_rowmax_part(hipStream_t st, const float* V, int ld, int L, int rows, int span,
                                     const double* kscale, int parts, float* rmax_part)
{
    dim3 grid(parts, rows / 4);
    rowmax_part_kernel<<<grid, 256, 0, st>>>(V, ld, L, span, kscale, parts, rmax_part);
    return hipGetLastError();
}

// Examine the resident matrix once: is every column (integers <= 256) x one constant?  If so build the
// integer planes of X and X^T and the per-gene scale (kernels_counts.hip.h).
static int ensure_counts(cnmf_ctx* ctx)
{
    const int fmt = gemm3_mode() == 4 ? 4 : 3;      // plane format the current mode multiplies
    if (ctx->count_state == 1 && ctx->count_fmt != fmt) {
        // the mode was switched between two calls on the same matrix (tests, A/B runs): rebuild the planes
        hipStreamSynchronize(ctx->stream);
        hipFree(ctx->C1); hipFree(ctx->Ct1); hipFree(ctx->d_scale);
        hipFree(ctx->C1h); hipFree(ctx->Ct1h); hipFree(ctx->hiA); hipFree(ctx->hiB);
        ctx->C1 = ctx->Ct1 = ctx->C1h = ctx->Ct1h = nullptr; ctx->hiA = ctx->hiB = nullptr;
        ctx->d_scale = nullptr; ctx->count_state = 0;
    }
    if (ctx->count_state != 0) return CNMF_OK;
    ctx->count_state = -1;
    ctx->count_fmt = fmt;
    const int N = (int)ctx->N, G = (int)ctx->G;
    if (ctx->N_pad % G3C_JW || ctx->G_pad % G3C_JW || getenv("CNMF_NO_COUNTS") || !ctx->count_detect) return CNMF_OK;
    hipStream_t st = ctx->stream;
    const int chunks = (N + CNT_ROWS - 1) / CNT_ROWS;
    DevPool pool;
    float* part = pool.get<float>((size_t)chunks * G);
    float* vmin = pool.get<float>(G);
    unsigned* fail = pool.get<unsigned>(G, true, st);
    float* unit = pool.get<float>(G);
    double* psx = pool.get<double>((size_t)chunks * G);
    double* psn = pool.get<double>((size_t)chunks * G);
    POOL_TRY(ctx, pool);
    dim3 grid((G + 255) / 256, chunks);
    col_minpos_kernel<<<grid, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, part);
    col_min_combine_kernel<<<(G + 255) / 256, 256, 0, st>>>(part, chunks, G, vmin);
    count_check_kernel<<<grid, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, vmin, fail);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<float> h_v(G), h_unit(G);
    std::vector<unsigned> h_fail(G);
    HIP_TRY(ctx, hipMemcpyAsync(h_v.data(), vmin, (size_t)G * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(h_fail.data(), fail, (size_t)G * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int g = 0; g < G; ++g) {
        int m = 0;
        for (int c = 1; c <= CNT_MAXMULT && !m; ++c) if (!(h_fail[g] & (1u << (c - 1)))) m = c;
        if (!m) return CNMF_OK;                            // this gene is not (small integers) x constant
        h_unit[g] = h_v[g] > 0.f ? h_v[g] / (float)m : 0.f;
    }
    HIP_TRY(ctx, hipMemcpyAsync(unit, h_unit.data(), (size_t)G * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMalloc(&ctx->d_scale, (size_t)ctx->G_pad * sizeof(double)));
    count_sums_kernel<<<grid, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, unit, psx, psn);
    count_scale_kernel<<<(ctx->G_pad + 255) / 256, 256, 0, st>>>(psx, psn, chunks, G, ctx->G_pad, ctx->d_scale);
    // does any count exceed 256?  then a second plane (256 hi) with per-block flags rides along
    unsigned* any_big = pool.get<unsigned>(1, true, st);
    POOL_TRY(ctx, pool);
    count_max_base_kernel<<<grid, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, unit, fmt == 4 ? G2_COUNT_BASE : 256.0f, any_big);
    unsigned h_big = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&h_big, any_big, sizeof h_big, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const size_t bytes = (size_t)ctx->N_pad * ctx->G_pad * 2;
    HIP_TRY(ctx, hipMalloc(&ctx->C1, bytes));
    HIP_TRY(ctx, hipMalloc(&ctx->Ct1, bytes));
    if (h_big) {
        const size_t nfA = (size_t)(ctx->N_pad / G3C_JW) * ((ctx->G_pad / 16 + 31) / 32) * sizeof(unsigned int);
        const size_t nfB = (size_t)(ctx->G_pad / G3C_JW) * ((ctx->N_pad / 16 + 31) / 32) * sizeof(unsigned int);
        HIP_TRY(ctx, hipMalloc(&ctx->C1h, bytes));
        HIP_TRY(ctx, hipMalloc(&ctx->Ct1h, bytes));
        HIP_TRY(ctx, hipMalloc(&ctx->hiA, nfA));
        HIP_TRY(ctx, hipMalloc(&ctx->hiB, nfB));
        HIP_TRY(ctx, hipMemsetAsync(ctx->hiA, 0, nfA, st));
        HIP_TRY(ctx, hipMemsetAsync(ctx->hiB, 0, nfB, st));
    }
    {
        const long long total = (long long)ctx->N_pad * (ctx->G_pad / 16);
        dim3 gt(ctx->N_pad / 16, (ctx->G_pad + 255) / 256);
        if (fmt == 4) {
            count_planes_f16_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
                ctx->X, ctx->G_pad, N, G, ctx->N_pad, ctx->G_pad, G3C_JW, unit, (unsigned short*)ctx->C1,
                (unsigned short*)ctx->C1h, ctx->hiA);
            count_planes_f16_transpose_kernel<<<gt, 256, 0, st>>>(
                ctx->X, ctx->G_pad, N, G, ctx->G_pad, ctx->N_pad, G3C_JW, unit, (unsigned short*)ctx->Ct1,
                (unsigned short*)ctx->Ct1h, ctx->hiB);
        } else {
            count_planes_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
                ctx->X, ctx->G_pad, N, G, ctx->N_pad, ctx->G_pad, G3C_JW, unit, (unsigned short*)ctx->C1,
                (unsigned short*)ctx->C1h, ctx->hiA);
            count_planes_transpose_kernel<<<gt, 256, 0, st>>>(
                ctx->X, ctx->G_pad, N, G, ctx->G_pad, ctx->N_pad, G3C_JW, unit, (unsigned short*)ctx->Ct1,
                (unsigned short*)ctx->Ct1h, ctx->hiB);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));                // the pool's scratch is freed on return
    ctx->count_state = 1;
    return CNMF_OK;
}

// the split-operand path needs whole 256 x 128 tiles
static bool gemm3_enabled(const cnmf_ctx* ctx, int KC)
{
    // (round 6: the plane builders carry the 16-cell blocks on grid.x, so matrices beyond 65 535 x 16 = 1 048 560 cells keep the
    //  matrix-pipe path -- probed at 1 100 000 x 2 000, 2.25e9 padded elements: tools/probe_big_matrix.py; bounded at 2^24 cells
    //  only because nothing larger has been run; the gene side still rides on grid.y of those builders: G_pad / 256 <= 65 535)
    return gemm3_mode() != 0 && KC % G3_MW == 0 && ctx->G_pad % gemm3_jw() == 0 && ctx->N_pad % gemm3_jw() == 0 &&
           ctx->N_pad <= (1 << 24) && ctx->G_pad / 256 <= 65535;
}

static int pick_nsplit3(const cnmf_ctx* ctx, int KC, int jw)
{
    // pass B grid = (G_pad/jw) x (KC/256) x nsplit; aim at one (two) workgroups per CU, >= 16 blocks per split
    const int tiles = std::max(1, ctx->G_pad / jw) * std::max(1, KC / G3_MW);
    const int Kb = ctx->N_pad / G3_BK;
    int s = std::max(1, std::min(gemm3_wg_slots() / std::max(1, tiles), Kb / 16));
    const int kb_per = (Kb + s - 1) / s;
    return (Kb + kb_per - 1) / kb_per;
}

static hipError_t launch_split2h_finalize(hipStream_t st, const float* src, int ld, int rows, int K, unsigned char* dst,
                                          int TR, const double* kscale, const float* rmax_part, int parts,
                                          float* inv_scale, const FinalizeArgs& fa, int nslots, int fin_y, SplitFused fu)
{
    const int bx = split2h_col_groups(K, rows);
    if (split2h_wide(K)) {
        const int by = rows / 16;
        split2h_finalize_kernel<16, 16><<<bx * by + nslots * fin_y, 256, 0, st>>>(src, ld, K, TR, (unsigned short*)dst, kscale,
                                                                                rmax_part, parts, inv_scale, bx, by, fa, fin_y, fu);
    } else {
        const int by = rows / 64;
        split2h_finalize_kernel<64, 4><<<bx * by + nslots * fin_y, 256, 0, st>>>(src, ld, K, TR, (unsigned short*)dst, kscale,
                                                                               rmax_part, parts, inv_scale, bx, by, fa, fin_y, fu);
    }
    return hipGetLastError();
}
