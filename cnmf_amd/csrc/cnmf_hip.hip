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
//
// One translation unit; the pieces, in include order:
//   kernels_*.hip.h      device code (GEMMs, sweeps, RNG, counts, consensus, MU)
//   runtime.hip.h        cnmf_ctx, error plumbing, scope-bound device buffers / events
//   gemm_host.hip.h      GEMM / sweep launchers, stream-K plans, operand planes, count-structure detection
//   (this file)          lifecycle, upload of the data matrix
//   batch_host.hip.h     batch buffers, the slot scheduler (run_batch), NNLS refit
//   consensus_host / mu_host / comm_host / normalize_host .hip.h   the other entry points
//   debug_host.hip.h     products with X for NNDSVD, diagnostics used by the tests
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/cnmf_hip.h"
#ifdef CNMF_DEBUG_ABI
#include "../../include/cnmf_hip_debug.h"
#endif
#include "kernels_gemm.hip.h"
#include "kernels_gemm3.hip.h"
#include "kernels_counts.hip.h"
#include "kernels_gemm2h.hip.h"
#include "kernels_rng.hip.h"
#include "kernels_sweep.hip.h"

using namespace cnmf;

#include "runtime.hip.h"
#include "csr_host.hip.h"
#include "gemm_host.hip.h"

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
    ctx_snapshot_env(ctx);
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        SET_ERR((cnmf_ctx*)nullptr, "hipStreamCreate failed");
        delete ctx;
        return nullptr;
    }
    return ctx;
}

// re-read the CNMF_* environment variables into the context's snapshot (tools and tests that flip a knob between two calls)
extern "C" int cnmf_reload_env(cnmf_ctx* ctx)
{
    if (!ctx) return CNMF_EINVAL;
    ctx_snapshot_env(ctx);
    return CNMF_OK;
}

static void free_x2(cnmf_ctx* c)
{
    hipFree(c->X2h); hipFree(c->X2m); hipFree(c->Xt2h); hipFree(c->Xt2m); hipFree(c->x2sA); hipFree(c->x2sB);
    hipFree(c->onesA); hipFree(c->onesB);
    c->X2h = c->X2m = c->Xt2h = c->Xt2m = nullptr; c->x2sA = c->x2sB = nullptr; c->onesA = c->onesB = nullptr;
}

static void free_mu_sparse(cnmf_ctx* c)
{
    for (int i = 0; i < 2; ++i) { c->spA[i].release(); c->spB[i].release(); }
    c->x_nnz = -1;
}

static void free_batch(cnmf_ctx* c)
{
    hipFree(c->H); hipFree(c->Wt); hipFree(c->XHt); hipFree(c->XHt1); hipFree(c->XHt2); hipFree(c->XtW); hipFree(c->d_split);
    hipFree(c->H3); hipFree(c->Wt3);
    hipFree(c->rmaxH); hipFree(c->rmaxW); hipFree(c->iscaleH); hipFree(c->iscaleW); hipFree(c->shiftW);
    c->rmaxH = c->rmaxW = c->iscaleH = c->iscaleW = nullptr; c->shiftW = nullptr;
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
    hipFree(ctx->X); hipFree(ctx->X3); hipFree(ctx->Xt3); hipFree(ctx->XtF);
    free_x2(ctx); free_mu_sparse(ctx); free_csr(ctx);
    hipFree(ctx->C1); hipFree(ctx->Ct1); hipFree(ctx->d_scale);
    hipFree(ctx->C1h); hipFree(ctx->Ct1h); hipFree(ctx->hiA); hipFree(ctx->hiB);
    hipFree(ctx->stageW); hipFree(ctx->stageH); hipFree(ctx->spectra);
    ctx->cons_ws.release();
    if (ctx->cons_pinned) hipHostFree(ctx->cons_pinned);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ------------------------------------------------------------------ data matrix
static int alloc_dense(cnmf_ctx* ctx)
{
    // one extra row of slack: pass B's last 128-gene tile runs past G_pad into the next row
    // (values that only feed never-stored output columns), so the last row needs a successor
    const size_t bytes = ((size_t)ctx->N_pad + 1) * ctx->G_pad * sizeof(float);
    HIP_TRY(ctx, hipMalloc(&ctx->X, bytes));
    HIP_TRY(ctx, hipMemsetAsync(ctx->X, 0, bytes, ctx->stream));
    return CNMF_OK;
}

static int alloc_matrix(cnmf_ctx* ctx, int64_t N, int64_t G, bool dense = true)
{
    if (N <= 0 || G <= 0 || N > (1ll << 30) || G > (1ll << 24)) {
        SET_ERR(ctx, "bad matrix shape %lld x %lld", (long long)N, (long long)G);
        return CNMF_EINVAL;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    free_batch(ctx);
    hipFree(ctx->X); ctx->X = nullptr;
    hipFree(ctx->XtF); ctx->XtF = nullptr;
    free_mu_sparse(ctx); free_csr(ctx);
    hipFree(ctx->X3); hipFree(ctx->Xt3); ctx->X3 = ctx->Xt3 = nullptr;
    free_x2(ctx);
    hipFree(ctx->C1); hipFree(ctx->Ct1); hipFree(ctx->d_scale);
    hipFree(ctx->C1h); hipFree(ctx->Ct1h); hipFree(ctx->hiA); hipFree(ctx->hiB);
    ctx->C1 = ctx->Ct1 = ctx->C1h = ctx->Ct1h = nullptr; ctx->hiA = ctx->hiB = nullptr;
    ctx->d_scale = nullptr; ctx->count_state = 0; ctx->count_fmt = 0;
    ctx->iter_prior.clear(); ctx->iter_hint.clear();     // iteration counts of another matrix say nothing about this one
    // (the resident spectra store survives a change of matrix: consensus() alternates between the normalised counts and
    //  the TPM matrix while the spectra of the factorize call keep serving k selection and further consensus calls;
    //  the store carries its own gene count, and a batch over another gene count refuses to append to it)
    ctx->N = N; ctx->G = G;
    ctx->N_pad = round_up(N, N >= 512 ? 256 : 128);  // whole 256-wide tiles for the split-operand GEMMs
    ctx->G_pad = round_up(G, G >= 512 ? 256 : 32);
    return dense ? alloc_dense(ctx) : CNMF_OK;
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

// *bad |= 1: a row whose columns are not strictly increasing (or a stored zero: the dense image would not list it);
// *bad |= 2: a column index outside [0, n_cols) -- such an entry is skipped
// X == nullptr: only the checks (the dense image is formed later, when a path asks for it: ensure_dense)
__global__ void csr_densify_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                   const float* __restrict__ data, float* __restrict__ X, int ld,
                                   int n_rows, int n_cols, int* __restrict__ bad)
{
    const int row = blockIdx.x;
    if (row >= n_rows) return;
    const long long b = indptr[row], e = indptr[row + 1];
    for (long long p = b + threadIdx.x; p < e; p += blockDim.x) {
        const int c = indices[p];
        if (c < 0 || c >= n_cols) { if (bad) atomicOr(bad, 2); continue; }
        if (bad && ((p > b && indices[p - 1] >= c) || data[p] == 0.f)) atomicOr(bad, 1);
        if (X) atomicAdd(&X[(size_t)row * ld + c], data[p]);   // duplicates sum, like .toarray()
    }
}

// The dense float32 image of the resident matrix.  A dense upload has it from the start; a CSR upload (canonical rows)
// keeps only its compressed rows until a path that multiplies the dense matrix asks for it here (round 5: a
// Kullback-Leibler run on a sparse matrix -- restarts on the non-zero images, float64 refits on the compressed rows --
// never does: 1.6 GB at 200 000 x 2000 stay unallocated, and so does the transposed copy of the dense solver).
static int ensure_dense(cnmf_ctx* ctx)
{
    if (ctx->X) return CNMF_OK;
    if (ctx->N <= 0 || !ctx->csr_ptr) { SET_ERR(ctx, "cnmf_set_matrix has not been called"); return CNMF_ESTATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = alloc_dense(ctx);
    if (rc) return rc;
    if (ctx->csr_nnz > 0) {
        csr_densify_kernel<<<(unsigned)ctx->N, 64, 0, ctx->stream>>>(ctx->csr_ptr, ctx->csr_idx, ctx->csr_val, ctx->X, ctx->G_pad,
                                                                    (int)ctx->N, (int)ctx->G, nullptr);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { hipFree(ctx->X); ctx->X = nullptr; HIP_TRY(ctx, e); }
    }
    return CNMF_OK;
}

extern "C" int cnmf_set_matrix_csr(cnmf_ctx* ctx, const int32_t* indptr, const int32_t* indices,
                                   const float* data, int64_t N, int64_t G)
{
    if (!ctx || !indptr || (!indices && indptr[N] > 0) || (!data && indptr[N] > 0)) {
        SET_ERR(ctx, "null argument");
        return CNMF_EINVAL;
    }
    int rc = alloc_matrix(ctx, N, G, false);                     // (no dense image yet: ensure_dense forms it on demand)
    if (rc) return rc;
    const int64_t nnz = indptr[N];
    // the arrays stay on the device (csr_host.hip.h): the paths that walk the stored entries use them as uploaded --
    // provided every row lists strictly increasing columns without stored zeros (scipy's canonical format); otherwise the
    // dense image is formed at once (duplicates summed like .toarray()) and the compressed rows are rebuilt from it on first use
    DevPool pool;
    int* d_ptr32 = pool.get<int>((size_t)(N + 1));
    int* d_bad = pool.get<int>(1, true, ctx->stream);
    POOL_TRY(ctx, pool);
    long long* d_ptr = nullptr;
    int* d_idx = nullptr;
    float* d_val = nullptr;
    hipError_t e = hipMalloc((void**)&d_ptr, (size_t)(N + 1) * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc((void**)&d_idx, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&d_val, (size_t)std::max<int64_t>(nnz, 1) * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_ptr32, indptr, (size_t)(N + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_idx, indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_val, data, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    int bad = 0;
    if (e == hipSuccess) {
        cnmf::csr_widen_ptr_kernel<<<(unsigned)((N + 1 + 255) / 256), 256, 0, ctx->stream>>>(d_ptr32, N + 1, d_ptr);
        if (nnz > 0)
            csr_densify_kernel<<<(unsigned)N, 64, 0, ctx->stream>>>(d_ptr, d_idx, d_val, nullptr, ctx->G_pad, (int)N, (int)G, d_bad);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess || (bad & 2)) { hipFree(d_ptr); hipFree(d_idx); hipFree(d_val); }
    HIP_TRY(ctx, e);
    if (bad & 2) { SET_ERR(ctx, "column index out of range in the CSR arrays"); return CNMF_EINVAL; }
    ctx->csr_ptr = d_ptr; ctx->csr_idx = d_idx; ctx->csr_val = d_val; ctx->csr_nnz = nnz;
    if (bad & 1) {                      // not canonical: the dense image now (sums duplicates), the arrays are dropped
        rc = ensure_dense(ctx);
        free_csr(ctx);
        if (rc) return rc;
    }
    return CNMF_OK;
}

// Count-structure detection on / off (default on).  The detection snaps a matrix whose every entry lies within
// 1e-3 count units of (an integer <= 65 535) x (one constant per gene) onto that grid (kernels_counts.hip.h); a caller
// whose data could satisfy that by accident -- not X = counts / std -- switches it off: the general split-operand
// path then multiplies the float32 values as they are.
extern "C" int cnmf_set_count_detection(cnmf_ctx* ctx, int enabled)
{
    if (!ctx) return CNMF_EINVAL;
    if (ctx->count_detect != (enabled != 0)) {
        ctx->count_detect = enabled != 0;
        if (ctx->count_state == -1 || !ctx->count_detect) {      // forget a previous decision
            hipStreamSynchronize(ctx->stream);
            hipFree(ctx->C1); hipFree(ctx->Ct1); hipFree(ctx->d_scale);
            hipFree(ctx->C1h); hipFree(ctx->Ct1h); hipFree(ctx->hiA); hipFree(ctx->hiB);
            ctx->C1 = ctx->Ct1 = ctx->C1h = ctx->Ct1h = nullptr; ctx->hiA = ctx->hiB = nullptr;
            ctx->d_scale = nullptr; ctx->count_state = 0; ctx->count_fmt = 0;
        }
    }
    return CNMF_OK;
}

extern "C" int cnmf_get_shape(const cnmf_ctx* ctx, int64_t* N, int64_t* G)
{
    if (!ctx) return CNMF_EINVAL;
    if (N) *N = ctx->N;
    if (G) *G = ctx->G;
    return (ctx->X || ctx->csr_ptr) ? CNMF_OK : CNMF_ESTATE;
}

// which images of the matrix are resident (bit 0: dense float32 image, 1: compressed rows of X, 2: compressed rows of X^T,
// 3: dense X^T copy of the dense multiplicative-update kernels, 4 / 5: non-zero images at padded rank 16 / 32, 6: count planes)
extern "C" int cnmf_matrix_images(const cnmf_ctx* ctx, int32_t* flags)
{
    if (!ctx || !flags) return CNMF_EINVAL;
    *flags = (ctx->X ? 1 : 0) | (ctx->csr_ptr ? 2 : 0) | (ctx->csc_ptr ? 4 : 0) | (ctx->XtF ? 8 : 0) |
             ((ctx->spA[0].ent && ctx->spB[0].ent) ? 16 : 0) | ((ctx->spA[1].ent && ctx->spB[1].ent) ? 32 : 0) | (ctx->C1 ? 64 : 0);
    return CNMF_OK;
}

#include "batch_host.hip.h"

// ------------------------------------------------------------------ consensus step
#include "consensus_host.hip.h"

// ------------------------------------------------------------------ multiplicative-update solver
#include "mu_host.hip.h"
#include "mu_refit_host.hip.h"
#include "comm_host.hip.h"
#include "normalize_host.hip.h"
#include "tail_host.hip.h"

#include "debug_host.hip.h"
#include "nndsvd_host.hip.h"
#include "format_host.hip.h"
