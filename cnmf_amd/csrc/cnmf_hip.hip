// libcnmf_hip.so : C-ABI + host runtime of the MI355X-native consensus-NMF engine.
//
// Runtime model (one context per GPU, one host thread per context):
//   * X lives zero-padded in HBM for the life of the context ([N_pad][G_pad] fp32).
//   * A batch call packs as many restarts as fit into KC component columns
//     ("slots"), and runs the coordinate-descent outer iteration for ALL of them with
//     two MFMA passes over X per iteration (kernels_gemm.hip.h) + lane-per-row sweeps
//     (kernels_sweep.hip.h).  The stopping rule runs on the device; a converged slot
//     freezes itself at exactly sklearn's iteration.  The host looks at a pinned
//     snapshot `lag` iterations behind the GPU, retires finished slots and refills
//     them from the pending list (a persistent work queue of restarts), so the batch
//     never runs at the speed of its slowest member.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cnmf_hip.h"
#include "kernels_gemm.hip.h"
#include "kernels_gemm3.hip.h"
#include "kernels_counts.hip.h"
#include "kernels_rng.hip.h"
#include "kernels_sweep.hip.h"

using namespace cnmf;

static thread_local std::string g_last_error;

struct cnmf_comm;

struct cnmf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // data matrix
    int64_t N = 0, G = 0;
    int N_pad = 0, G_pad = 0;
    float* X = nullptr;
    unsigned char *X3 = nullptr, *Xt3 = nullptr;   // bf16 planes of X and X^T (split-operand GEMM), built on first use
    int planes_tr = 0;                             // row-tile height they were built with
    // count structure X = n * d (kernels_counts.hip.h): 0 = not examined, 1 = present, -1 = absent
    int count_state = 0;
    unsigned char *C1 = nullptr, *Ct1 = nullptr;   // integer planes of n and n^T (one bf16 plane, 256-row tiles)
    unsigned char *C1h = nullptr, *Ct1h = nullptr; // second planes (256 hi) when some count exceeds 256, else NULL
    unsigned int *hiA = nullptr, *hiB = nullptr;   // their flags: one bit per (tile row, block)
    double* d_scale = nullptr;                     // per-gene scale d [G_pad]

    // batch buffers (sized for kc_alloc columns)
    int kc_alloc = 0, nsplit_alloc = 0, nsplitA_alloc = 0, parts_alloc = 0;
    size_t gram_part_floats = 0;
    float *H = nullptr, *Wt = nullptr, *XHt = nullptr, *XHt1 = nullptr, *XHt2 = nullptr, *XtW = nullptr;
    unsigned char *H3 = nullptr, *Wt3 = nullptr;   // planes of the packed factors, refreshed every iteration
    unsigned char* d_split = nullptr;   // stream-K cut flags of the current plan
    float *gramH = nullptr, *gramW = nullptr, *gram_part = nullptr;
    double* viol_part = nullptr;
    SlotDesc* d_slots = nullptr;
    int* d_slot_list = nullptr;
    SlotDesc* h_slots = nullptr;      // pinned: per-slot install descriptors
    SlotDesc* h_snap = nullptr;       // pinned: snapshot ring [RING][kc_alloc]
    int* h_slot_list = nullptr;       // pinned ring of new-slot lists
    float *stageW = nullptr, *stageH = nullptr;
    size_t stageW_sz = 0, stageH_sz = 0;

    // resident spectra store (device) for the gather / consensus
    float* spectra = nullptr;
    size_t spectra_cap = 0, spectra_rows = 0;

    cnmf_comm* comm = nullptr;        // RCCL communicator (comm_host.hip.h); NULL = single GPU
};

static constexpr int RING = 8;
#ifndef CNMF_GEMM3_DEFAULT
#define CNMF_GEMM3_DEFAULT 3
#endif

#define SET_ERR(ctx, ...)                                                   \
    do {                                                                    \
        char buf_[512];                                                     \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);                           \
        if (ctx) (ctx)->err = buf_;                                         \
        g_last_error = buf_;                                                \
    } while (0)

#define HIP_TRY(ctx, call)                                                  \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            SET_ERR(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? CNMF_ENOMEM : CNMF_EHIP;   \
        }                                                                   \
    } while (0)

// Scope-bound device allocations / events: released when the entry point returns, on EVERY path
// (the HIP_TRY early returns included; hipFree waits for work that still uses the buffer).
struct DevPool {
    std::vector<void*> ptrs;
    hipError_t err = hipSuccess;
    template <typename T> T* get(size_t n, bool zero = false, hipStream_t st = nullptr) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e != hipSuccess) { err = e; return nullptr; }
        ptrs.push_back(p);
        if (zero) hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(T), st);
        return (T*)p;
    }
    ~DevPool() { for (void* p : ptrs) hipFree(p); }
};

struct EventPool {
    std::vector<hipEvent_t> evs;
    hipError_t err = hipSuccess;
    hipEvent_t get(unsigned flags = hipEventDefault) {
        hipEvent_t e = nullptr;
        hipError_t r = hipEventCreateWithFlags(&e, flags);
        if (r != hipSuccess) { err = r; return nullptr; }
        evs.push_back(e);
        return e;
    }
    ~EventPool() { for (hipEvent_t e : evs) hipEventDestroy(e); }
};

#define POOL_TRY(ctx, pool)                                                                   \
    do {                                                                                      \
        if ((pool).err != hipSuccess) {                                                       \
            SET_ERR(ctx, "device allocation failed: %s (%s:%d)", hipGetErrorString((pool).err), __FILE__, __LINE__); \
            return ((pool).err == hipErrorOutOfMemory) ? CNMF_ENOMEM : CNMF_EHIP;             \
        }                                                                                     \
    } while (0)

static inline int round_up(int64_t v, int m) { return (int)(((v + m - 1) / m) * m); }

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
    static bool attr_set = false;
    constexpr size_t lds = gemm_lds_bytes<MTW, WM, WN, NN, TBK>();
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_kernel<MTW, WM, WN, NN, TBK>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    gemm_kernel<MTW, WM, WN, NN, TBK><<<grid, 256, lds, st>>>(A, lda, B, ldb, C, ldc, cstride, Kper,
                                                             Ktot, J);
    return hipGetLastError();
}

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
            if (KC % 256 == 0 && KC >= 256 && getenv("CNMF_S_MTW2")) GO(2, 4, 1);
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
static hipError_t launch_sweep(hipStream_t st, int nslots, float* V, int ldv, int L, const float* P,
                               const float* gram, const SlotDesc* slots, float l1, float* gram_part,
                               double* viol_part, int chunks, int parts, int want_gram, int kmax, int tiers,
                               SplitInfo sp = SplitInfo{nullptr, nullptr, 1, 1, 1})
{
    dim3 grid(parts, nslots);
    static bool attr_set = false;
    if (!attr_set) {      // ranks above 32 need more than the default 64 KB of dynamic LDS
        hipFuncSetAttribute((const void*)sweep_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes(KMAX));
        hipFuncSetAttribute((const void*)sweep_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes(KMAX));
        hipFuncSetAttribute((const void*)sweep_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes(KMAX));
        attr_set = true;
    }
    // one launch per rank tier present among the live slots; a launch skips the slots of other tiers at once
    const size_t lds = sweep_lds_bytes(kmax);
    const int kg = sweep_kg(kmax);
    if (tiers & 1) sweep_kernel<0><<<grid, 256, lds, st>>>(V, ldv, L, P, sp, gram, slots, l1, gram_part, viol_part, chunks, want_gram, kg, kmax);
    if (tiers & 2) sweep_kernel<1><<<grid, 256, lds, st>>>(V, ldv, L, P, sp, gram, slots, l1, gram_part, viol_part, chunks, want_gram, kg, kmax);
    if (tiers & 4) sweep_kernel<2><<<grid, 256, lds, st>>>(V, ldv, L, P, sp, gram, slots, l1, gram_part, viol_part, chunks, want_gram, kg, kmax);
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
    if (KC % 128 != 0 || getenv("CNMF_NO_STREAMK")) return sk;
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
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_streamk_kernel<4, 1, 4, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
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

// ------------------------------------------------------------------ lifecycle
extern "C" int cnmf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* cnmf_version(void) { return "cnmf_hip 0.1.0 (gfx950)"; }

extern "C" const char* cnmf_last_error(const cnmf_ctx* ctx)
{
    return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

extern "C" cnmf_ctx* cnmf_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        SET_ERR((cnmf_ctx*)nullptr, "no HIP device available (%s)", hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        SET_ERR((cnmf_ctx*)nullptr, "device %d out of range (have %d)", device, n);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        SET_ERR((cnmf_ctx*)nullptr, "hipSetDevice(%d) failed", device);
        return nullptr;
    }
    cnmf_ctx* ctx = new cnmf_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        SET_ERR((cnmf_ctx*)nullptr, "hipStreamCreate failed");
        delete ctx;
        return nullptr;
    }
    return ctx;
}

static void free_batch(cnmf_ctx* c)
{
    hipFree(c->H); hipFree(c->Wt); hipFree(c->XHt); hipFree(c->XHt1); hipFree(c->XHt2); hipFree(c->XtW); hipFree(c->d_split);
    hipFree(c->H3); hipFree(c->Wt3);
    c->XHt1 = c->XHt2 = nullptr; c->d_split = nullptr; c->H3 = c->Wt3 = nullptr;
    hipFree(c->gramH); hipFree(c->gramW); hipFree(c->gram_part); hipFree(c->viol_part);
    hipFree(c->d_slots); hipFree(c->d_slot_list);
    if (c->h_slots) hipHostFree(c->h_slots);
    if (c->h_snap) hipHostFree(c->h_snap);
    if (c->h_slot_list) hipHostFree(c->h_slot_list);
    c->H = c->Wt = c->XHt = c->XtW = c->gramH = c->gramW = c->gram_part = nullptr;
    c->viol_part = nullptr; c->d_slots = nullptr; c->d_slot_list = nullptr;
    c->h_slots = c->h_snap = nullptr; c->h_slot_list = nullptr;
    c->kc_alloc = 0; c->gram_part_floats = 0;
}

extern "C" void cnmf_destroy(cnmf_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    free_batch(ctx);
    cnmf_comm_finalize(ctx);
    hipFree(ctx->X); hipFree(ctx->X3); hipFree(ctx->Xt3);
    hipFree(ctx->C1); hipFree(ctx->Ct1); hipFree(ctx->d_scale);
    hipFree(ctx->C1h); hipFree(ctx->Ct1h); hipFree(ctx->hiA); hipFree(ctx->hiB);
    hipFree(ctx->stageW); hipFree(ctx->stageH); hipFree(ctx->spectra);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ------------------------------------------------------------------ data matrix
static int alloc_matrix(cnmf_ctx* ctx, int64_t N, int64_t G)
{
    if (N <= 0 || G <= 0 || N > (1ll << 30) || G > (1ll << 24)) {
        SET_ERR(ctx, "bad matrix shape %lld x %lld", (long long)N, (long long)G);
        return CNMF_EINVAL;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    free_batch(ctx);
    hipFree(ctx->X); ctx->X = nullptr;
    hipFree(ctx->X3); hipFree(ctx->Xt3); ctx->X3 = ctx->Xt3 = nullptr;
    hipFree(ctx->C1); hipFree(ctx->Ct1); hipFree(ctx->d_scale);
    hipFree(ctx->C1h); hipFree(ctx->Ct1h); hipFree(ctx->hiA); hipFree(ctx->hiB);
    ctx->C1 = ctx->Ct1 = ctx->C1h = ctx->Ct1h = nullptr; ctx->hiA = ctx->hiB = nullptr;
    ctx->d_scale = nullptr; ctx->count_state = 0;
    ctx->spectra_rows = 0;            // spectra of another matrix are not comparable
    ctx->N = N; ctx->G = G;
    ctx->N_pad = round_up(N, N >= 512 ? 256 : 128);  // whole 256-wide tiles for the split-operand GEMMs
    ctx->G_pad = round_up(G, G >= 512 ? 256 : 32);
    // one extra row of slack: pass B's last 128-gene tile runs past G_pad into the next row
    // (values that only feed never-stored output columns), so the last row needs a successor
    const size_t bytes = ((size_t)ctx->N_pad + 1) * ctx->G_pad * sizeof(float);
    HIP_TRY(ctx, hipMalloc(&ctx->X, bytes));
    HIP_TRY(ctx, hipMemsetAsync(ctx->X, 0, bytes, ctx->stream));
    return CNMF_OK;
}

extern "C" int cnmf_set_matrix(cnmf_ctx* ctx, const float* X, int64_t N, int64_t G)
{
    if (!ctx || !X) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    int rc = alloc_matrix(ctx, N, G);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy2DAsync(ctx->X, (size_t)ctx->G_pad * sizeof(float), X,
                                  (size_t)G * sizeof(float), (size_t)G * sizeof(float), (size_t)N,
                                  hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNMF_OK;
}

__global__ void csr_densify_kernel(const int* __restrict__ indptr, const int* __restrict__ indices,
                                   const float* __restrict__ data, float* __restrict__ X, int ld,
                                   int n_rows)
{
    const int row = blockIdx.x;
    if (row >= n_rows) return;
    const int b = indptr[row], e = indptr[row + 1];
    for (int p = b + threadIdx.x; p < e; p += blockDim.x)
        atomicAdd(&X[(size_t)row * ld + indices[p]], data[p]);   // duplicates sum, like .toarray()
}

extern "C" int cnmf_set_matrix_csr(cnmf_ctx* ctx, const int32_t* indptr, const int32_t* indices,
                                   const float* data, int64_t N, int64_t G)
{
    if (!ctx || !indptr || (!indices && indptr[N] > 0) || (!data && indptr[N] > 0)) {
        SET_ERR(ctx, "null argument");
        return CNMF_EINVAL;
    }
    int rc = alloc_matrix(ctx, N, G);
    if (rc) return rc;
    const int64_t nnz = indptr[N];
    DevPool pool;
    int* d_ptr = pool.get<int>((size_t)(N + 1));
    int* d_idx = pool.get<int>((size_t)nnz);
    float* d_val = pool.get<float>((size_t)nnz);
    POOL_TRY(ctx, pool);
    HIP_TRY(ctx, hipMemcpyAsync(d_ptr, indptr, (size_t)(N + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_idx, indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_val, data, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        csr_densify_kernel<<<(unsigned)N, 64, 0, ctx->stream>>>(d_ptr, d_idx, d_val, ctx->X, ctx->G_pad, (int)N);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNMF_OK;
}

extern "C" int cnmf_get_shape(const cnmf_ctx* ctx, int64_t* N, int64_t* G)
{
    if (!ctx) return CNMF_EINVAL;
    if (N) *N = ctx->N;
    if (G) *G = ctx->G;
    return ctx->X ? CNMF_OK : CNMF_ESTATE;
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
// per product on 256 x 256 tiles) whenever the resident matrix has that structure.  Read on every call so that
// tests can switch it.
// (Tried and dropped, all within 3 % of variant 2 at the 50k x 2000 shape: the same ping-pong with register
//  staging; 256 x 256 tiles with the two wave groups half a block apart (2/3 of the DMA bytes per flop).)
static int gemm3_mode()
{
    const char* e = getenv("CNMF_GEMM3");
    const int mode = e ? atoi(e) : CNMF_GEMM3_DEFAULT;
    return (mode < 0 || mode > 3) ? CNMF_GEMM3_DEFAULT : mode;
}
static int gemm3_wg_slots() { return gemm3_mode() >= 2 ? 256 : 512; }
static int gemm3_jw() { return G3_JW; }     // j extent of a tile = row tile of the B planes

static hipError_t launch_gemm3(hipStream_t st, const unsigned char* A3, const unsigned char* B3, int Kb,
                               float* C, int ldc, long long cstride, int KC, int Jpad, int nsplit)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES);
        hipFuncSetAttribute((const void*)gemm3g_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G3G_LDS_BYTES);
        attr_set = true;
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

// ---- stream-K plan for the split-operand pass A (tile = 256 components x 128 cells, up to 2 cuts per tile)
struct StreamK3 {
    bool on = false;
    int T = 0, Kb = 0, P = 0, MG = 1;
    std::vector<unsigned char> flags;     // bit 0: >= 1 cut (plane 1 holds the tail), bit 1: 2 cuts (plane 2 the middle)
};

static StreamK3 plan_streamk3(int KC, int N_pad, int G_pad, int n_wg_slots, int jw)
{
    StreamK3 sk;
    sk.MG = KC / G3_MW;
    sk.T = sk.MG * (N_pad / jw);
    sk.Kb = G_pad / G3_BK;
    sk.P = n_wg_slots;
    if (sk.T < sk.P / 2 + sk.P / 4 || sk.P > 2 * sk.T || getenv("CNMF_NO_STREAMK")) return sk;   // few tiles: K split + reduce instead
    sk.on = true;
    sk.flags.assign(sk.T, 0);
    const long long U = (long long)sk.T * sk.Kb;
    for (int p = 1; p < sk.P; ++p) {
        const long long b = U * p / sk.P;
        if (b % sk.Kb) {
            unsigned char& f = sk.flags[b / sk.Kb];
            f = f ? 3 : 1;
        }
    }
    return sk;
}

static hipError_t launch_gemm3_streamk(hipStream_t st, const StreamK3& sk, const unsigned char* A3,
                                       const unsigned char* B3, float* C0, float* C1, float* C2, int ldc)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm3_streamk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES);
        hipFuncSetAttribute((const void*)gemm3g_streamk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G3G_LDS_BYTES);
        attr_set = true;
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
    dim3 grid((ctx->G_pad + 255) / 256, ctx->N_pad / 16);
    split3_transpose_kernel<<<grid, 256, 0, ctx->stream>>>(ctx->X, ctx->G_pad, ctx->N_pad, ctx->G_pad, ctx->N_pad,
                                                            TR, (unsigned short*)ctx->Xt3);
    HIP_TRY(ctx, hipGetLastError());
    return CNMF_OK;
}

// ---- count-structured data: launchers of the 256 x 256 integer-plane kernel
// Bhi / hiflag: second integer plane and its block flags (nullptr when no count exceeds 256)
static hipError_t launch_gemm3c(hipStream_t st, const unsigned char* A3, const unsigned char* B1,
                                const unsigned char* Bhi, const unsigned int* hiflag, int Kb,
                                float* C, int ldc, long long cstride, int KC, int Jpad, int nsplit)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm3c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, g3c_lds_bytes(true));
        attr_set = true;
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
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm3c_streamk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, g3c_lds_bytes(true));
        attr_set = true;
    }
    gemm3c_streamk_kernel<<<sk.P, 512, g3c_lds_bytes(Bhi != nullptr), st>>>(A3, B1, Bhi, hiflag, sk.Kb, C0, C1, C2, ldc,
                                                                            sk.MG, sk.T);
    return hipGetLastError();
}

// Examine the resident matrix once: is every column (integers <= 256) x one constant?  If so build the
// integer planes of X and X^T and the per-gene scale (kernels_counts.hip.h).
static int ensure_counts(cnmf_ctx* ctx)
{
    if (ctx->count_state != 0) return CNMF_OK;
    ctx->count_state = -1;
    const int N = (int)ctx->N, G = (int)ctx->G;
    if (ctx->N_pad % G3C_JW || ctx->G_pad % G3C_JW || getenv("CNMF_NO_COUNTS")) return CNMF_OK;
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
    count_max_kernel<<<grid, 256, 0, st>>>(ctx->X, ctx->G_pad, N, G, unit, any_big);
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
        count_planes_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
            ctx->X, ctx->G_pad, N, G, ctx->N_pad, ctx->G_pad, G3C_JW, unit, (unsigned short*)ctx->C1,
            (unsigned short*)ctx->C1h, ctx->hiA);
        dim3 gt((ctx->G_pad + 255) / 256, ctx->N_pad / 16);
        count_planes_transpose_kernel<<<gt, 256, 0, st>>>(
            ctx->X, ctx->G_pad, N, G, ctx->G_pad, ctx->N_pad, G3C_JW, unit, (unsigned short*)ctx->Ct1,
            (unsigned short*)ctx->Ct1h, ctx->hiB);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));                // the pool's scratch is freed on return
    ctx->count_state = 1;
    return CNMF_OK;
}

// the split-operand path needs whole 256 x 128 tiles
static bool gemm3_enabled(const cnmf_ctx* ctx, int KC)
{
    // (the plane builders index 16-cell blocks with blockIdx.y: up to 65 535 x 16 cells)
    return gemm3_mode() != 0 && KC % G3_MW == 0 && ctx->G_pad % gemm3_jw() == 0 && ctx->N_pad % gemm3_jw() == 0 &&
           ctx->N_pad / 16 <= 65535 && ctx->G_pad / 16 <= 65535;
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

// ------------------------------------------------------------------ batch buffers
static int sweep_max_parts()
{
    static const int v = getenv("CNMF_SWEEP_PARTS") ? atoi(getenv("CNMF_SWEEP_PARTS")) : 64;
    return std::max(1, v);
}
static int sweep_chunks(int L) { const int64_t m = 256ll * sweep_max_parts(); return std::max(1, (int)(((int64_t)L + m - 1) / m)); }
static int sweep_parts(int L) { int c = sweep_chunks(L); return (L + 256 * c - 1) / (256 * c); }

static int pick_nsplit(const cnmf_ctx* ctx, int KC)
{
    if (const char* s = getenv("CNMF_NSPLIT")) { int v = atoi(s); if (v > 0) return v; }
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
    if (const char* s = getenv("CNMF_NSPLIT_A")) { int v = atoi(s); if (v > 0) return effective_splits(ctx->G_pad, v); }
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
    for (int s = 0; s < n; ++s) {
        const volatile int* flag = &sp[s].pad_;
        long spins = 0;
        while (*flag != stamp) {
            if (++spins % 4096 == 0) {
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
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return CNMF_OK;
}

static int pick_kc(int64_t total_k, int max_k, int kc_max)
{
    if (kc_max <= 0) kc_max = 256;
    if (const char* s = getenv("CNMF_KC")) { int v = atoi(s); if (v >= 32) kc_max = v; }
    kc_max = std::max(32, std::min(256, (kc_max / 32) * 32));
    int kc = 32;
    while (kc < kc_max && kc < total_k) kc *= 2;
    kc = std::min(kc, kc_max);
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
    if (!ctx->X) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }
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
    int KC = pick_kc(total_k, max_k, prm->kc_max);
    const int KC0 = KC;
    rc = ensure_batch(ctx, KC, max_k, min_k);
    if (rc) return rc;
    rc = ensure_stage(ctx, (size_t)N * KMAX, (size_t)G * KMAX);
    if (rc) return rc;
    int nsplit = std::min(pick_nsplit(ctx, KC), ctx->nsplit_alloc);
    bool use3 = gemm3_enabled(ctx, KC);            // split-operand bf16 MFMA path (whole 256-column tiles only)
    bool usec = false;                             // ... with X as one integer plane (count-structured data)
    if (use3 && gemm3_mode() == 3) {
        rc = ensure_counts(ctx);
        if (rc) return rc;
        usec = ctx->count_state == 1;
    }
    const int gemm_mode_used = !use3 ? 0 : (usec ? 3 : std::min(gemm3_mode(), 2));
    if (use3 && !usec) { rc = ensure_planes(ctx); if (rc) return rc; }
    const int jwA = usec ? G3C_JW : G3_JW;         // width of a pass-A / pass-B tile
    const int nsplit3 = use3 ? pick_nsplit3(ctx, KC, jwA) : 1;
    const int fin_y = (max_k * max_k + 255) / 256;       // finalize blocks per slot
    const int lag = std::max(1, std::min(RING - 2, prm->lag > 0 ? prm->lag : 2));
    hipStream_t st = ctx->stream;

    // device result buffers
    DevPool pool;
    EventPool events;
    float* d_Hres = nullptr; float* d_Wres = nullptr;
    if (resident) {
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
    POOL_TRY(ctx, pool);

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

    // restarts in descending rank so that freed slots can always be reused
    std::vector<int> order(n);
    for (int r = 0; r < n; ++r) order[r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kk[a] > kk[b]; });
    size_t next = 0;                 // first queue position that may still be pending
    int n_pending = n;

    ColAlloc cols(KC);
    std::vector<HostSlot> hs(KC0);
    int nslots = 0;          // highest used slot index + 1
    int n_active = 0;
    int64_t it = 0;          // batch iterations enqueued so far
    int snap_nslots[RING] = {0};
    bool h3_valid = false;           // H3 holds the planes of the current H (split-operand modes)
    // stamps restart at 1 in every call: forget the ones a previous call left in the ring (nothing is in flight here)
    memset(ctx->h_snap, 0, (size_t)ctx->kc_alloc * RING * sizeof(SlotDesc));
    hipEvent_t ev_begin = events.get(), ev_end = events.get();
    POOL_TRY(ctx, events);
    HIP_TRY(ctx, hipEventRecord(ev_begin, st));
    // HIP events around the two GEMM passes of every `time_stride`-th iteration (an event record costs
    // ~6 us of queue time: bracketing every launch would take 4 % off the throughput it measures)
    const int time_stride = !stats ? 0 : (prm->profile > 0 ? prm->profile : (getenv("CNMF_TIME_GEMM") ? 1 : 0));
    std::vector<hipEvent_t> gev;   // (a0,a1,b0,b1) per iteration when timing is requested

    const int chunksW = sweep_chunks(N), partsW = sweep_parts(N);
    const int chunksH = sweep_chunks(G), partsH = sweep_parts(G);
    const float l1W = (float)prm->l1_reg_W, l2W = (float)prm->l2_reg_W;
    const float l1H = (float)prm->l1_reg_H, l2H = (float)prm->l2_reg_H;
    const int gvarA = getenv("CNMF_GEMM_A") ? atoi(getenv("CNMF_GEMM_A")) : 0;
    const int gvarB = getenv("CNMF_GEMM_B") ? atoi(getenv("CNMF_GEMM_B")) : 0;
    int64_t restart_iters = 0, column_iters = 0, restart_col_iters = 0;
    const bool dbg = getenv("CNMF_DEBUG") != nullptr;
    int64_t dbg_it[9] = {0}, dbg_live[9] = {0};
    const int wg_slots = getenv("CNMF_SK_WGS") ? atoi(getenv("CNMF_SK_WGS")) : 2 * 256;                    // T-layout pass A: 2 workgroups per CU (73.7 KB LDS each)
    StreamK sk = plan_streamk(KC, ctx->N_pad, ctx->G_pad, wg_slots);
    if (sk.on) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_split, sk.split.data(), sk.split.size(), hipMemcpyHostToDevice, st));
    int nsplitA = (sk.on && gvarA == 0) ? 1 : std::min(pick_nsplit_A(ctx, KC), ctx->nsplitA_alloc);
    StreamK3 sk3;
    if (use3) {
        sk3 = plan_streamk3(KC, ctx->N_pad, ctx->G_pad, gemm3_wg_slots(), jwA);
        if (sk3.on) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_split, sk3.flags.data(), sk3.flags.size(), hipMemcpyHostToDevice, st));
        else nsplitA = std::min(pick_nsplit_A3(ctx, KC, jwA), ctx->nsplitA_alloc);   // few tiles: K split + reduce
    }
    int n_done = 0;

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
        cols.release(h.off, k);
        h.state = 0; h.restart = -1;
        --n_active; ++n_done;
        return CNMF_OK;
    };

    while (true) {
        // ---- refill free columns from the pending list
        int n_new = 0;
        int* new_list = ctx->h_slot_list + (size_t)(it % RING) * KC0;
        // pending restarts are sorted by descending rank; a hole too small for the head of the
        // queue is filled with the largest pending rank that fits (restarts are independent, so
        // the order they run in is free) -> the packed columns stay full in the main phase
        int failed_k = 1 << 30;                       // smallest rank that did not fit in this pass
        for (size_t pi = next; pi < order.size() && n_pending > 0; ++pi) {
            const int r = order[pi];
            if (r < 0) { if (pi == next) ++next; continue; }      // already taken
            const int k = kk[r];
            if (k >= failed_k) continue;
            const int off = cols.alloc(k);
            if (off < 0) { failed_k = k; continue; }
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
            SlotDesc* d = &ctx->h_slots[s];
            memset(d, 0, sizeof *d);
            d->off = off; d->k = k; d->active = 1; d->iter = 0; d->restart = r;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->d_slots + s, d, sizeof(SlotDesc), hipMemcpyHostToDevice, st));
            new_list[n_new++] = s;
            ++n_active;
        }
        if (n_new) {
            int* dl = ctx->d_slot_list + (size_t)(it % RING) * KC0;
            HIP_TRY(ctx, hipMemcpyAsync(dl, new_list, n_new * sizeof(int), hipMemcpyHostToDevice, st));
            gram_rows_kernel<<<n_new, 256, 0, st>>>(ctx->H, ctx->G_pad, G, ctx->d_slots, dl, ctx->gramH, l2W);
            HIP_TRY(ctx, hipGetLastError());
        }
        if (n_active == 0 && n_pending == 0) break;

        // ---- one coordinate-descent outer iteration for every slot in flight
        int tiers = 0;
        for (int s2 = 0; s2 < nslots; ++s2)
            if (hs[s2].state) tiers |= hs[s2].k <= 16 ? 1 : (hs[s2].k <= 32 ? 2 : 4);
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
            if (n_new > 0 || !h3_valid)
                HIP_TRY(ctx, launch_split3(st, ctx->H, ctx->G_pad, KC, ctx->G_pad, ctx->H3, G3_MW, usec ? ctx->d_scale : nullptr));
            if (time_gemm) hipEventRecord(gev[gev.size() - 4], st);
            if (sk3.on) {
                if (usec)
                    HIP_TRY(ctx, launch_gemm3c_streamk(st, sk3, ctx->H3, ctx->C1, ctx->C1h, ctx->hiA, ctx->XHt, ctx->XHt1,
                                                       ctx->XHt2, ctx->N_pad));
                else
                    HIP_TRY(ctx, launch_gemm3_streamk(st, sk3, ctx->H3, ctx->X3, ctx->XHt, ctx->XHt1, ctx->XHt2, ctx->N_pad));
                spA = SplitInfo{ctx->XHt1, ctx->d_split, jwA, G3_MW, sk3.MG, ctx->XHt2};
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
            HIP_TRY(ctx, launch_reduce_splits(st, ctx->XHt, nsplitA, (long long)KC * ctx->N_pad,
                                              (long long)KC * ctx->N_pad));
        // W half-step                                             (sklearn _nmf.py:500)
        HIP_TRY(ctx, launch_sweep(st, nslots, ctx->Wt, ctx->N_pad, N, ctx->XHt, ctx->gramH,
                                  ctx->d_slots, l1W, ctx->gram_part, ctx->viol_part, chunksW, partsW, 1, max_k, tiers, spA));
        if (use3) {
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
        const int nsB = use3 ? nsplit3 : nsplit;
        if (usec)
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
        // H half-step
        HIP_TRY(ctx, launch_reduce_splits(st, ctx->XtW, nsB, (long long)KC * ctx->G_pad,
                                          (long long)KC * ctx->G_pad, usec ? ctx->d_scale : nullptr, ctx->G_pad));
        HIP_TRY(ctx, launch_sweep(st, nslots, ctx->H, ctx->G_pad, G, ctx->XtW, ctx->gramW,
                                  ctx->d_slots, l1H, ctx->gram_part, ctx->viol_part, chunksH, partsH, 1, max_k, tiers));
        // the H finalize also publishes every slot's state into the host-mapped ring entry of this
        // iteration (stamp it + 1): no copy kernel and no event per iteration
        SlotDesc* snap = ctx->h_snap + (size_t)(it % RING) * KC0;
        SlotDesc* snap_dev = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&snap_dev, snap, 0));
        if (use3) {
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
        if (dbg) {
            int live = 0;
            for (int s2 = 0; s2 < nslots; ++s2) if (hs[s2].state) live += hs[s2].k;
            dbg_it[KC / 32] += 1; dbg_live[KC / 32] += live;
        }
        ++it;
        // catch up: everything older than `lag` must be inspected; drain fully when idle
        {
            const int64_t si = it - 1 - lag;   // snapshot index to inspect now
            if (si >= 0) {
                const SlotDesc* sp = ctx->h_snap + (size_t)(si % RING) * KC0;
                rc = wait_snapshot(ctx, sp, snap_nslots[si % RING], (int)(si + 1));
                if (rc) return rc;
                for (int s = 0; s < snap_nslots[si % RING]; ++s)
                    if (hs[s].state == 1 && hs[s].installed_at <= si && sp[s].active == 0 && sp[s].restart == hs[s].restart) {
                        rc = retire(s, sp[s]);
                        if (rc) return rc;
                    }
            }
        }
        // ---- tail compaction: nothing left to refill with and at most half of the packed
        // columns still iterate -> repack the live slots into a narrower batch so the two
        // GEMM passes shrink with the work (their cost is proportional to KC).
        if (n_pending == 0 && n_active > 0 && KC > 32 && !getenv("CNMF_NO_COMPACT")) {
            int live_cols = 0;
            for (int s = 0; s < nslots; ++s) if (hs[s].state) live_cols += hs[s].k;
            int KCn = 32;
            while (KCn < live_cols) KCn *= 2;
            if (KCn < KC) {
                std::vector<int> idx;
                for (int s = 0; s < nslots; ++s) if (hs[s].state) idx.push_back(s);
                std::sort(idx.begin(), idx.end(), [&](int a, int b) { return hs[a].off < hs[b].off; });
                int pos = 0;
                for (int s : idx) {
                    HostSlot& h = hs[s];
                    if (h.off != pos) {
                        dim3 gH((G + 255) / 256, h.k), gW((N + 255) / 256, h.k), gI((std::max(N, G) + 255) / 256, h.k);
                        extract_kernel<<<gH, 256, 0, st>>>(ctx->H, ctx->G_pad, G, h.off, h.k, ctx->stageH, 0);
                        extract_kernel<<<gW, 256, 0, st>>>(ctx->Wt, ctx->N_pad, N, h.off, h.k, ctx->stageW, 0);
                        install_cm_kernel<<<gI, 256, 0, st>>>(ctx->stageH, ctx->stageW, ctx->H, ctx->G_pad, G, ctx->Wt, ctx->N_pad, N, pos);
                        set_slot_off_kernel<<<1, 1, 0, st>>>(ctx->d_slots, s, pos);
                        h.off = pos;
                    }
                    pos += h.k;
                }
                if (pos < KCn) {
                    dim3 gc((ctx->G_pad + 255) / 256, KCn - pos), gw((ctx->N_pad + 255) / 256, KCn - pos);
                    clear_rows_kernel<<<gc, 256, 0, st>>>(ctx->H, ctx->G_pad, ctx->G_pad, pos, KCn - pos);
                    clear_rows_kernel<<<gw, 256, 0, st>>>(ctx->Wt, ctx->N_pad, ctx->N_pad, pos, KCn - pos);
                }
                HIP_TRY(ctx, hipGetLastError());
                KC = KCn;
                cols = ColAlloc(KC);
                for (int s : idx) cols.alloc(hs[s].k);
                const int cap = (ctx->nsplit_alloc * KC0) / KC;
                nsplit = std::max(1, std::min(pick_nsplit(ctx, KC), cap));
                use3 = usec = false;                // fewer than 256 packed columns: the f32 pipe takes over
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

    if (dbg)
        for (int i = 1; i <= 8; ++i)
            if (dbg_it[i]) fprintf(stderr, "[cnmf] KC=%d: %lld iterations, mean host-live columns %.1f\n", i * 32,
                                   (long long)dbg_it[i], (double)dbg_live[i] / dbg_it[i]);
    HIP_TRY(ctx, hipEventRecord(ev_end, st));
    if (!resident)
        HIP_TRY(ctx, hipMemcpyAsync(H_out, d_Hres, hoff[n] * sizeof(float), hipMemcpyDeviceToHost, st));
    if (W_out)
        HIP_TRY(ctx, hipMemcpyAsync(W_out, d_Wres, woff[n] * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (resident) ctx->spectra_rows += (size_t)total_k;
    if (stats) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev_begin, ev_end);
        stats->gpu_ms = ms;
        stats->outer_iterations = it;
        stats->restart_iterations = restart_iters;
        stats->column_iterations = column_iters;
        stats->restart_column_iterations = restart_col_iters;
        stats->kc = KC0; stats->nsplit = gemm_mode_used ? nsplit3 : ctx->nsplit_alloc;
        stats->gemm_mode = gemm_mode_used;
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
    if (!ctx->X) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }
    int rc = validate_params(ctx, prm);
    if (rc) return rc;
    if (!Hin || !W_out || k < 1) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    if (k > KMAX) { SET_ERR(ctx, "n_components=%d > CNMF_KMAX=%d", k, KMAX); return CNMF_EUNSUPPORTED; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int N = (int)ctx->N, G = (int)ctx->G;
    const int KC = k <= 32 ? 32 : 64;
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
                                      chunksW, partsW, 0, k, k <= 16 ? 1 : (k <= 32 ? 2 : 4)));
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

// ------------------------------------------------------------------ consensus step
#include "consensus_host.hip.h"

// ------------------------------------------------------------------ multiplicative-update solver
#include "mu_host.hip.h"
#include "comm_host.hip.h"
#include "normalize_host.hip.h"

// ------------------------------------------------------------------ X . Q / X^T . Q
extern "C" int cnmf_x_matmul(cnmf_ctx* ctx, int trans, const float* Q, int ncols, float* out)
{
    if (!ctx || !Q || !out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (!ctx->X) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }
    if (ncols < 1 || ncols > 256 || (trans != 0 && trans != 1)) { SET_ERR(ctx, "bad ncols/trans"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int N = (int)ctx->N, G = (int)ctx->G;
    const int KC = ncols <= 32 ? 32 : (ncols <= 64 ? 64 : (ncols <= 128 ? 128 : 256));
    const int Kin = trans ? N : G, Kp = trans ? ctx->N_pad : ctx->G_pad;     // contraction length
    const int Jout = trans ? G : N, Jp = trans ? ctx->G_pad : ctx->N_pad;
    DevPool pool;
    float* dQ = pool.get<float>((size_t)Kin * ncols);
    float* dA = pool.get<float>((size_t)KC * Kp, true, st);                   // Q^T, component-major, zero padded
    const int nsplit = trans ? std::max(1, std::min(16, Kp / 2048)) : 1;
    float* dC = pool.get<float>((size_t)nsplit * KC * Jp);
    float* dO = pool.get<float>((size_t)Jout * ncols);
    if (pool.err) { SET_ERR(ctx, "device allocation failed"); return CNMF_ENOMEM; }
    HIP_TRY(ctx, hipMemcpyAsync(dQ, Q, (size_t)Kin * ncols * sizeof(float), hipMemcpyHostToDevice, st));
    dim3 gI((Kin + 255) / 256, ncols);
    // install_kernel's W path transposes a row-major [L][k] block into component-major rows
    install_kernel<<<gI, 256, 0, st>>>(nullptr, dQ, dA, Kp, 0, dA, Kp, Kin, 0, ncols);
    if (!trans)
        HIP_TRY(ctx, launch_gemm<false>(st, 0, dA, Kp, ctx->X, ctx->G_pad, dC, Jp, 0, KC, Kp, Jp, 1));
    else {
        HIP_TRY(ctx, launch_gemm<true>(st, 0, dA, Kp, ctx->X, ctx->G_pad, dC, Jp, (long long)KC * Jp, KC, Kp, Jp, nsplit));
        HIP_TRY(ctx, launch_reduce_splits(st, dC, nsplit, (long long)KC * Jp, (long long)KC * Jp));
    }
    dim3 gO((Jout + 255) / 256, ncols);
    extract_kernel<<<gO, 256, 0, st>>>(dC, Jp, Jout, 0, ncols, dO, 1);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, dO, (size_t)Jout * ncols * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return CNMF_OK;
}

// ------------------------------------------------------------------ diagnostics
extern "C" int cnmf_debug_gemm(cnmf_ctx* ctx, int mode, int variant, const float* A, const float* B,
                               float* C, int KC, int K, int J, int nsplit, double* ms_out, int reps)
{
    if (!ctx || !A || !B || !C) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (KC % 32 || K % 32 || J % 32 || nsplit < 1 || (mode != 0 && mode != 1)) {
        SET_ERR(ctx, "debug_gemm needs KC,K,J multiples of 32"); return CNMF_EINVAL;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int Jp = round_up(J, 128);           // J padded like N_pad so any tile shape is addressable
    const int Kp = K;
    DevPool pool;
    EventPool events;
    const size_t bA = (size_t)KC * Kp * sizeof(float);
    const size_t bB = (mode == 0 ? (size_t)Jp * Kp : ((size_t)Kp + 1) * J + 128) * sizeof(float);
    const size_t bC = (size_t)nsplit * KC * Jp * sizeof(float);
    float* dA = pool.get<float>(bA / sizeof(float));
    float* dB = pool.get<float>(bB / sizeof(float));
    float* dC = pool.get<float>(bC / sizeof(float));
    POOL_TRY(ctx, pool);
    HIP_TRY(ctx, hipMemsetAsync(dB, 0, bB, st));
    HIP_TRY(ctx, hipMemcpyAsync(dA, A, bA, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dB, B, (size_t)(mode == 0 ? J : Kp) * (mode == 0 ? Kp : J) * sizeof(float), hipMemcpyHostToDevice, st));
    hipEvent_t e0 = events.get(), e1 = events.get();
    POOL_TRY(ctx, events);
    reps = std::max(1, reps);
    for (int i = 0; i < reps + 1; ++i) {
        if (i == 1) hipEventRecord(e0, st);
        hipError_t e = (mode == 0)
            ? launch_gemm<false>(st, variant, dA, Kp, dB, Kp, dC, Jp, (long long)KC * Jp, KC, Kp, Jp, 1)
            : launch_gemm<true>(st, variant, dA, Kp, dB, J, dC, Jp, (long long)KC * Jp, KC, Kp, J, nsplit);
        HIP_TRY(ctx, e);
    }
    hipEventRecord(e1, st);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    float ms = 0.f;
    if (reps >= 1) hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = (reps >= 1) ? ms / reps : 0.0;
    std::vector<float> hc((size_t)(mode == 0 ? 1 : nsplit) * KC * Jp);
    HIP_TRY(ctx, hipMemcpy(hc.data(), dC, hc.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int c = 0; c < KC; ++c)
        for (int j = 0; j < J; ++j) {
            float s = hc[(size_t)c * Jp + j];
            if (mode == 1)
                for (int z = 1; z < nsplit; ++z) s += hc[((size_t)z * KC + c) * Jp + j];
            C[(size_t)c * J + j] = s;
        }
    return CNMF_OK;
}

// C[KC][J] = A[KC][K] . B[J][K]^T through the split-operand bf16 MFMA path (KC % 256 == 0, K % 16 == 0)
extern "C" int cnmf_debug_gemm3(cnmf_ctx* ctx, const float* A, const float* B, float* C, int KC, int K, int J,
                                int nsplit, double* ms_out, int reps)
{
    if (!ctx || !A || !B || !C) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (KC % 256 || K % 16 || J < 1 || nsplit < 1) { SET_ERR(ctx, "debug_gemm3 needs KC %% 256 == 0, K %% 16 == 0"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int Jp = round_up(J, gemm3_jw()), Kb = K / 16;
    DevPool pool;
    EventPool events;
    float* dA = pool.get<float>((size_t)KC * K);
    float* dB = pool.get<float>((size_t)Jp * K, true, st);
    unsigned char* dA3 = pool.get<unsigned char>((size_t)KC * Kb * G3_ROWB);
    unsigned char* dB3 = pool.get<unsigned char>((size_t)Jp * Kb * G3_ROWB);
    float* dC = pool.get<float>((size_t)nsplit * KC * Jp);
    hipEvent_t e0 = events.get(), e1 = events.get();
    POOL_TRY(ctx, pool);
    POOL_TRY(ctx, events);
    HIP_TRY(ctx, hipMemcpyAsync(dA, A, (size_t)KC * K * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dB, B, (size_t)J * K * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, launch_split3(st, dA, K, KC, K, dA3, G3_MW));
    HIP_TRY(ctx, launch_split3(st, dB, K, Jp, K, dB3, gemm3_jw()));
    reps = std::max(1, reps);
    int zs = 1;
    for (int i = 0; i < reps + 1; ++i) {
        if (i == 1) hipEventRecord(e0, st);
        HIP_TRY(ctx, launch_gemm3(st, dA3, dB3, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit));
    }
    hipEventRecord(e1, st);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    { const int kb_per = (Kb + nsplit - 1) / nsplit; zs = (Kb + kb_per - 1) / kb_per; }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms / reps;
    std::vector<float> hc((size_t)zs * KC * Jp);
    HIP_TRY(ctx, hipMemcpy(hc.data(), dC, hc.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int c = 0; c < KC; ++c)
        for (int j = 0; j < J; ++j) {
            float v = hc[(size_t)c * Jp + j];
            for (int z = 1; z < zs; ++z) v += hc[((size_t)z * KC + c) * Jp + j];
            C[(size_t)c * J + j] = v;
        }
    return CNMF_OK;
}

// C[KC][J] = A[KC][K] . Bn[J][K]^T through the count-path kernel: Bn holds non-negative integers <= 65535
// (lo plane + flagged hi plane), A arbitrary float32 (three planes).  KC % 256 == 0, K % 16 == 0.
extern "C" int cnmf_debug_gemm3c(cnmf_ctx* ctx, const float* A, const float* Bn, float* C, int KC, int K, int J,
                                 int nsplit, double* ms_out, int reps)
{
    if (!ctx || !A || !Bn || !C) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (KC % 256 || K % 16 || J < 1 || nsplit < 1) { SET_ERR(ctx, "debug_gemm3c needs KC %% 256 == 0, K %% 16 == 0"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int Jp = round_up(J, G3C_JW), Kb = K / 16;
    DevPool pool;
    EventPool events;
    float* dA = pool.get<float>((size_t)KC * K);
    float* dB = pool.get<float>((size_t)J * K);
    float* dUnit = pool.get<float>(K);
    unsigned char* dA3 = pool.get<unsigned char>((size_t)KC * Kb * G3_ROWB);
    unsigned char* dB1 = pool.get<unsigned char>((size_t)Jp * Kb * 32);
    unsigned char* dBh = pool.get<unsigned char>((size_t)Jp * Kb * 32);
    unsigned int* dFl = pool.get<unsigned int>((size_t)(Jp / G3C_JW) * ((Kb + 31) / 32), true, st);
    float* dC = pool.get<float>((size_t)nsplit * KC * Jp);
    hipEvent_t e0 = events.get(), e1 = events.get();
    POOL_TRY(ctx, pool);
    POOL_TRY(ctx, events);
    std::vector<float> ones(K, 1.0f);
    HIP_TRY(ctx, hipMemcpyAsync(dA, A, (size_t)KC * K * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dB, Bn, (size_t)J * K * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dUnit, ones.data(), (size_t)K * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, launch_split3(st, dA, K, KC, K, dA3, G3_MW));
    {
        const long long total = (long long)Jp * Kb;
        count_planes_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(dB, K, J, K, Jp, K, G3C_JW, dUnit,
                                                                           (unsigned short*)dB1, (unsigned short*)dBh, dFl);
        HIP_TRY(ctx, hipGetLastError());
    }
    reps = std::max(1, reps);
    for (int i = 0; i < reps + 1; ++i) {
        if (i == 1) hipEventRecord(e0, st);
        HIP_TRY(ctx, launch_gemm3c(st, dA3, dB1, dBh, dFl, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit));
    }
    hipEventRecord(e1, st);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const int kb_per = (Kb + nsplit - 1) / nsplit, zs = (Kb + kb_per - 1) / kb_per;
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms / reps;
    std::vector<float> hc((size_t)zs * KC * Jp);
    HIP_TRY(ctx, hipMemcpy(hc.data(), dC, hc.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int c = 0; c < KC; ++c)
        for (int j = 0; j < J; ++j) {
            float v = hc[(size_t)c * Jp + j];
            for (int z = 1; z < zs; ++z) v += hc[((size_t)z * KC + c) * Jp + j];
            C[(size_t)c * J + j] = v;
        }
    return CNMF_OK;
}

extern "C" int cnmf_debug_standard_normal(cnmf_ctx* ctx, uint32_t seed, int64_t n, double* out)
{
    if (!ctx || !out || n < 0) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevPool pool;
    double* d = pool.get<double>((size_t)n);
    POOL_TRY(ctx, pool);
    launch_standard_normal(ctx->stream, seed, n, d);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNMF_OK;
}
