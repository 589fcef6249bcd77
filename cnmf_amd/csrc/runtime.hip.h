// Context, error plumbing and scope-bound device resources of libcnmf_hip (included by cnmf_hip.hip).
#pragma once

#include <map>
static thread_local std::string g_last_error;

struct cnmf_comm;

// Workspace of the consensus entry points, kept by the context between calls: blocks are handed out by a bump
// pointer and stay allocated (the ~30 hipMalloc / hipFree pairs of one consensus call cost more than its kernels),
// released when the context is destroyed or when a call left more than `keep_limit` bytes behind.
struct Arena {
    struct Block { char* p; size_t cap, used; };
    std::vector<Block> blocks;
    hipError_t err = hipSuccess;
    static constexpr size_t keep_limit = (size_t)2 << 30;      // (the R x R distance matrix of a large consensus is given back)
    void reset() { err = hipSuccess; for (Block& b : blocks) b.used = 0; }
    size_t total() const { size_t t = 0; for (const Block& b : blocks) t += b.cap; return t; }
    void release() { for (Block& b : blocks) hipFree(b.p); blocks.clear(); }
    template <typename T> T* get(size_t n, bool zero = false, hipStream_t st = nullptr) {
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        Block* hit = nullptr;
        for (Block& b : blocks) if (b.cap - b.used >= bytes) { hit = &b; break; }
        if (!hit) {
            void* p = nullptr;
            const size_t cap = std::max(bytes, (size_t)16 << 20);
            hipError_t e = hipMalloc(&p, cap);
            if (e != hipSuccess) { err = e; return nullptr; }
            blocks.push_back(Block{(char*)p, cap, 0});
            hit = &blocks.back();
        }
        T* out = (T*)(hit->p + hit->used);
        hit->used += bytes;
        if (zero) hipMemsetAsync(out, 0, bytes, st);
        return out;
    }
};

// blocked sliced-ELL image of the non-zeros of one orientation of X (kernels_mu_sparse.hip.h); all pointers device memory
struct SpImage {
    int R = 0, C = 0, BS = 0, nblk = 0, nslice = 0;
    int* perm = nullptr; long long* off = nullptr; int* len = nullptr; void* ent = nullptr;
    size_t n_ent = 0;
    void release() { hipFree(perm); hipFree(off); hipFree(len); hipFree(ent); *this = SpImage{}; }
};

struct cnmf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // the CNMF_* environment variables as they were when the context was created (cnmf_create) or last re-read
    // (cnmf_reload_env): the host paths of a call consult THIS snapshot (ctx_getenv), not the process environment -- a
    // knob that steers a numerics-affecting path cannot change between two calls on one context behind the caller's back
    std::map<std::string, std::string> env;

    // data matrix
    int64_t N = 0, G = 0;
    int N_pad = 0, G_pad = 0;
    float* X = nullptr;
    unsigned char *X3 = nullptr, *Xt3 = nullptr;   // bf16 planes of X and X^T (split-operand GEMM), built on first use
    int planes_tr = 0;                             // row-tile height they were built with
    // count structure X = n * d (kernels_counts.hip.h): 0 = not examined, 1 = present, -1 = absent
    bool count_detect = true;                      // cnmf_set_count_detection(): look for the count structure at all?
    int count_state = 0;
    unsigned char *C1 = nullptr, *Ct1 = nullptr;   // integer planes of n and n^T (one bf16 plane, 256-row tiles)
    unsigned char *C1h = nullptr, *Ct1h = nullptr; // second planes (256 hi) when some count exceeds 256, else NULL
    unsigned int *hiA = nullptr, *hiB = nullptr;   // their flags: one bit per (tile row, block)
    double* d_scale = nullptr;                     // per-gene scale d [G_pad]
    int count_fmt = 0;                             // 3 = bf16 planes (base 256), 4 = f16 planes (base 2048, swizzled slots)
    // any OTHER matrix on the f16 pipe (gemm_mode 5): two f16 planes of x * 2^s_row for X (rows = cells) and X^T (rows =
    // genes), the per-row 2^-s, and all-ones block flags for the two-plane ("HI") instantiation of the count kernels
    unsigned char *X2h = nullptr, *X2m = nullptr, *Xt2h = nullptr, *Xt2m = nullptr;
    float *x2sA = nullptr, *x2sB = nullptr;
    unsigned int *onesA = nullptr, *onesB = nullptr;
    float* XtF = nullptr;                          // X^T [round_up(G_pad, 64)][N_pad] float32, built on first use by the
                                                   // Kullback-Leibler solver (kernels_mu_mfma.hip.h)
    // Kullback-Leibler on the non-zeros (kernels_mu_sparse.hip.h): images for padded ranks 16 and 32, cells x genes (A) and
    // genes x cells (B); x_nnz = -1 until the first Kullback-Leibler call counted the matrix
    SpImage spA[2], spB[2];
    long long x_nnz = -1;
    // compressed rows of X (cells x genes) and of X^T (genes x cells), csr_host.hip.h: kept from cnmf_set_matrix_csr or
    // built from the dense matrix on first use; 64-bit row pointers, float32 values like the dense image
    long long *csr_ptr = nullptr, *csc_ptr = nullptr;
    int *csr_idx = nullptr, *csc_idx = nullptr;
    float *csr_val = nullptr, *csc_val = nullptr;
    long long csr_nnz = -1;

    // batch buffers (sized for kc_alloc columns)
    int kc_alloc = 0, nsplit_alloc = 0, nsplitA_alloc = 0, parts_alloc = 0;
    size_t gram_part_floats = 0;
    float *H = nullptr, *Wt = nullptr, *XHt = nullptr, *XHt1 = nullptr, *XHt2 = nullptr, *XtW = nullptr;
    unsigned char *H3 = nullptr, *Wt3 = nullptr;   // planes of the packed factors, refreshed every iteration
    unsigned char* d_split = nullptr;   // stream-K cut flags of the current plan
    // f16 two-plane factor split (kernels_gemm2h.hip.h): per-row maxima reported by the sweeps, 2^-s per row
    float *rmaxH = nullptr, *rmaxW = nullptr, *iscaleH = nullptr, *iscaleW = nullptr;
    int* shiftW = nullptr;              // [3][kc_alloc]: exponents of the W planes written by the sweep (two generations + scratch)
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
    int64_t spectra_G = 0;             // gene count of the rows in the store (the matrix they were computed on)
    // what the batch calls on THIS matrix learned about the restarts' length: mean outer iterations per rank (0 = nothing
    // yet; cnmf_get_iteration_means), and the caller's hints for the NEXT calls (cnmf_set_iteration_hints): with hints the
    // queue starts longest-expected-first instead of learning the order again.  Never applied implicitly: the queue order
    // decides the packed columns a restart occupies, and its float32 result moves in the last bits with them.
    std::vector<double> iter_prior, iter_hint;

    Arena cons_ws;                    // consensus workspace (consensus_host.hip.h)
    void* cons_pinned = nullptr;      // pinned host block of the consensus calls (k-means state read-backs)
    size_t cons_pinned_bytes = 0;

    cnmf_comm* comm = nullptr;        // RCCL communicator (comm_host.hip.h); NULL = single GPU
};

extern char** environ;
static void ctx_snapshot_env(cnmf_ctx* ctx)
{
    ctx->env.clear();
    for (char** e = environ; e && *e; ++e) {
        if (strncmp(*e, "CNMF_", 5) != 0) continue;
        const char* eq = strchr(*e, '=');
        if (eq) ctx->env[std::string(*e, eq - *e)] = std::string(eq + 1);
    }
}
static const char* ctx_getenv(const cnmf_ctx* ctx, const char* name)
{
    if (!ctx) return getenv(name);
    auto it = ctx->env.find(name);
    return it == ctx->env.end() ? nullptr : it->second.c_str();
}

static constexpr int RING = 8;
#ifndef CNMF_GEMM3_DEFAULT
#define CNMF_GEMM3_DEFAULT 4
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

// Opt a kernel into more than the default 64 KB of dynamic LDS -- once per (kernel, DEVICE): the attribute belongs to
// the device's code object, so a process that drives several GPUs (one context each) must set it on every one of them
// (a function-local `static bool` did it for the first device only).  Thread-safe; the return code is checked.
#include <mutex>
#include <set>
#include <utility>
static hipError_t dyn_lds_optin(const void* fn, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, dev})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}

static inline int round_up(int64_t v, int m) { return (int)(((v + m - 1) / m) * m); }
