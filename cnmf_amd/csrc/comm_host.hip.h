// Multi-GPU exchange: ONE RCCL all-gather of the packed spectra (plus tiny ones for the headers).
//
// Replaces the reference's filesystem "gather" (cnmf.py:755-770: combine_nmf re-reads one
// npz per restart).  The restarts themselves shard with no collective (cnmf.py:52-53).
// RCCL is bound lazily with dlopen, so the library loads -- and every single-GPU entry point
// works -- on a machine without librccl; nothing here is linked at build time.
// Included by cnmf_hip.hip (needs cnmf_ctx, SET_ERR, HIP_TRY).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace cnmf {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

static RcclApi* rccl_api()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char* names[] = {getenv("CNMF_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
        api.why = dlerror();
    }
    if (!api.handle) return nullptr;
#define CNMF_SYM(field, name)                                                   \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name)); \
    if (!api.field) { api.why = std::string("missing symbol ") + name; dlclose(api.handle); api.handle = nullptr; return nullptr; }
    CNMF_SYM(GetUniqueId, "ncclGetUniqueId")
    CNMF_SYM(CommInitRank, "ncclCommInitRank")
    CNMF_SYM(CommDestroy, "ncclCommDestroy")
    CNMF_SYM(AllGather, "ncclAllGather")
    CNMF_SYM(GetErrorString, "ncclGetErrorString")
#undef CNMF_SYM
    return &api;
}

}  // namespace cnmf

struct cnmf_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

#define RCCL_TRY(ctx, api, call)                                                              \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            SET_ERR(ctx, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r_), __FILE__, __LINE__); \
            return CNMF_ECOMM;                                                                \
        }                                                                                     \
    } while (0)

static_assert(sizeof(ncclUniqueId) == CNMF_COMM_ID_BYTES, "CNMF_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int cnmf_comm_unique_id(unsigned char* id_out)
{
    if (!id_out) { SET_ERR((cnmf_ctx*)nullptr, "id_out is NULL"); return CNMF_EINVAL; }
    cnmf::RcclApi* api = cnmf::rccl_api();
    if (!api) { SET_ERR((cnmf_ctx*)nullptr, "RCCL is not available on this machine"); return CNMF_ECOMM; }
    ncclUniqueId id;
    RCCL_TRY((cnmf_ctx*)nullptr, api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return CNMF_OK;
}

extern "C" int cnmf_comm_init(cnmf_ctx* ctx, const unsigned char* id_bytes, int rank, int world)
{
    if (!ctx) return CNMF_EINVAL;
    if (world < 1 || rank < 0 || rank >= world || !id_bytes) { SET_ERR(ctx, "bad rank/world/id"); return CNMF_EINVAL; }
    if (ctx->comm) { SET_ERR(ctx, "communicator already initialised"); return CNMF_ESTATE; }
    cnmf::RcclApi* api = cnmf::rccl_api();
    if (!api) { SET_ERR(ctx, "RCCL is not available on this machine"); return CNMF_ECOMM; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    cnmf_comm* c = new cnmf_comm();
    c->rank = rank; c->world = world;
    ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        SET_ERR(ctx, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(r));
        delete c;
        return CNMF_ECOMM;
    }
    ctx->comm = c;
    return CNMF_OK;
}

extern "C" int cnmf_comm_finalize(cnmf_ctx* ctx)
{
    if (!ctx) return CNMF_EINVAL;
    if (!ctx->comm) return CNMF_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    cnmf::RcclApi* api = cnmf::rccl_api();
    if (api && ctx->comm->comm) api->CommDestroy(ctx->comm->comm);
    delete ctx->comm;
    ctx->comm = nullptr;
    return CNMF_OK;
}

extern "C" int cnmf_comm_rank(const cnmf_ctx* ctx) { return (ctx && ctx->comm) ? ctx->comm->rank : 0; }
extern "C" int cnmf_comm_world(const cnmf_ctx* ctx) { return (ctx && ctx->comm) ? ctx->comm->world : 1; }

// device-side all-gather of `nbytes` per rank; without a communicator (single GPU) it is a copy
static int allgather_device(cnmf_ctx* ctx, const void* d_send, void* d_recv, size_t nbytes)
{
    if (!ctx->comm) {
        HIP_TRY(ctx, hipMemcpyAsync(d_recv, d_send, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
        return CNMF_OK;
    }
    cnmf::RcclApi* api = cnmf::rccl_api();
    RCCL_TRY(ctx, api, api->AllGather(d_send, d_recv, nbytes, ncclInt8, ctx->comm->comm, ctx->stream));
    return CNMF_OK;
}

extern "C" int cnmf_allgather_bytes(cnmf_ctx* ctx, const void* send, int64_t nbytes, void* recv)
{
    if (!ctx) return CNMF_EINVAL;
    if (nbytes < 0 || (nbytes > 0 && (!send || !recv))) { SET_ERR(ctx, "bad buffers"); return CNMF_EINVAL; }
    if (nbytes == 0) return CNMF_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int world = ctx->comm ? ctx->comm->world : 1;
    const size_t pad = ((size_t)nbytes + 15) & ~(size_t)15;
    unsigned char *d_s = nullptr, *d_r = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_s, pad));
    if (hipMalloc(&d_r, pad * world) != hipSuccess) { hipFree(d_s); SET_ERR(ctx, "out of device memory"); return CNMF_ENOMEM; }
    int rc = CNMF_OK;
    hipError_t e = hipMemsetAsync(d_s, 0, pad, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_s, send, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = allgather_device(ctx, d_s, d_r, pad);
    for (int r = 0; r < world && e == hipSuccess && rc == CNMF_OK; ++r)
        e = hipMemcpyAsync((unsigned char*)recv + (size_t)r * nbytes, d_r + (size_t)r * pad, (size_t)nbytes,
                           hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_s); hipFree(d_r);
    if (e != hipSuccess) { SET_ERR(ctx, "all-gather staging failed: %s", hipGetErrorString(e)); return CNMF_EHIP; }
    return rc;
}

extern "C" int64_t cnmf_spectra_rows(const cnmf_ctx* ctx) { return ctx ? (int64_t)ctx->spectra_rows : 0; }

extern "C" int64_t cnmf_spectra_genes(const cnmf_ctx* ctx) { return ctx ? ctx->spectra_G : 0; }

extern "C" int cnmf_spectra_reset(cnmf_ctx* ctx)
{
    if (!ctx) return CNMF_EINVAL;
    ctx->spectra_rows = 0;
    return CNMF_OK;
}

// rows [n][G] float32 appended behind what the store holds: merged spectra that did not come out of a resident batch of
// this context (read from the reference's files, gathered from other GPUs) -- uploaded ONCE, then every cnmf_consensus_store
// call (other k, other density thresholds) works from the device
extern "C" int cnmf_spectra_append(cnmf_ctx* ctx, const float* rows, int64_t n_rows, int64_t n_genes)
{
    if (!ctx || !rows || n_rows < 0 || n_genes < 1) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    if (ctx->spectra_rows && ctx->spectra_G != n_genes) {
        SET_ERR(ctx, "the resident store holds spectra over %lld genes, not %lld: cnmf_spectra_reset first",
                (long long)ctx->spectra_G, (long long)n_genes);
        return CNMF_ESTATE;
    }
    if (n_rows == 0) return CNMF_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t need = (ctx->spectra_rows + (size_t)n_rows) * (size_t)n_genes;
    if (need > ctx->spectra_cap) {
        float* nb = nullptr;
        const size_t cap = std::max(need, ctx->spectra_cap * 2);
        HIP_TRY(ctx, hipMalloc(&nb, cap * sizeof(float)));
        hipError_t ce = hipSuccess;
        if (ctx->spectra_rows)
            ce = hipMemcpyAsync(nb, ctx->spectra, ctx->spectra_rows * (size_t)n_genes * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->stream);
        if (ce != hipSuccess) { hipFree(nb); HIP_TRY(ctx, ce); }
        hipFree(ctx->spectra);
        ctx->spectra = nb; ctx->spectra_cap = cap;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->spectra + ctx->spectra_rows * (size_t)n_genes, rows, (size_t)n_rows * n_genes * sizeof(float),
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->spectra_G = n_genes;
    ctx->spectra_rows += (size_t)n_rows;
    return CNMF_OK;
}

extern "C" int cnmf_spectra_fetch(cnmf_ctx* ctx, float* out)
{
    if (!ctx) return CNMF_EINVAL;
    if (!ctx->spectra_rows) return CNMF_OK;
    if (!out) { SET_ERR(ctx, "out is NULL"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->spectra, ctx->spectra_rows * (size_t)ctx->spectra_G * sizeof(float),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNMF_OK;
}

// rows [row0, row0 + n_rows) of the store only (a batch call's own rows: the store may hold earlier calls' too)
extern "C" int cnmf_spectra_fetch_rows(cnmf_ctx* ctx, int64_t row0, int64_t n_rows, float* out)
{
    if (!ctx) return CNMF_EINVAL;
    if (row0 < 0 || n_rows < 0 || (size_t)(row0 + n_rows) > ctx->spectra_rows) {
        SET_ERR(ctx, "rows %lld..%lld outside the store's %lld rows", (long long)row0, (long long)(row0 + n_rows), (long long)ctx->spectra_rows);
        return CNMF_EINVAL;
    }
    if (!n_rows) return CNMF_OK;
    if (!out) { SET_ERR(ctx, "out is NULL"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->spectra + (size_t)row0 * (size_t)ctx->spectra_G,
                                (size_t)n_rows * (size_t)ctx->spectra_G * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNMF_OK;
}

extern "C" int cnmf_allgather_spectra(cnmf_ctx* ctx, const float* local, int64_t rows_local, int64_t rows_max,
                                      int64_t n_genes, float* out)
{
    if (!ctx) return CNMF_EINVAL;
    if (rows_local < 0 || rows_max < rows_local || n_genes <= 0 || !out) { SET_ERR(ctx, "bad sizes / out is NULL"); return CNMF_EINVAL; }
    if (!local) {   // the context's resident store (filled by cnmf_nmf_cd_batch_resident)
        if ((int64_t)ctx->spectra_rows != rows_local || ctx->spectra_G != n_genes) {
            SET_ERR(ctx, "resident store holds %lld rows x %lld genes, caller said %lld x %lld",
                    (long long)ctx->spectra_rows, (long long)ctx->spectra_G, (long long)rows_local, (long long)n_genes);
            return CNMF_EINVAL;
        }
    }
    if (rows_max == 0) return CNMF_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int world = ctx->comm ? ctx->comm->world : 1;
    const size_t blk = (size_t)rows_max * n_genes * sizeof(float);
    const size_t mine = (size_t)rows_local * n_genes * sizeof(float);
    unsigned char *d_s = nullptr, *d_r = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_s, blk));
    if (hipMalloc(&d_r, blk * world) != hipSuccess) { hipFree(d_s); SET_ERR(ctx, "out of device memory"); return CNMF_ENOMEM; }
    int rc = CNMF_OK;
    hipError_t e = hipSuccess;
    if (mine < blk) e = hipMemsetAsync(d_s + mine, 0, blk - mine, ctx->stream);       // ragged shards are zero-padded
    if (e == hipSuccess && mine)
        e = local ? hipMemcpyAsync(d_s, local, mine, hipMemcpyHostToDevice, ctx->stream)
                  : hipMemcpyAsync(d_s, ctx->spectra, mine, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) rc = allgather_device(ctx, d_s, d_r, blk);                   // THE data-path collective
    if (e == hipSuccess && rc == CNMF_OK) e = hipMemcpyAsync(out, d_r, blk * world, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_s); hipFree(d_r);
    if (e != hipSuccess) { SET_ERR(ctx, "all-gather staging failed: %s", hipGetErrorString(e)); return CNMF_EHIP; }
    return rc;
}
