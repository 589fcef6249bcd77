// Products with the resident matrix and the diagnostics entry points used by the tests
// (included by cnmf_hip.hip last).
#pragma once

// ------------------------------------------------------------------ X . Q / X^T . Q
extern "C" int cnmf_x_matmul(cnmf_ctx* ctx, int trans, const float* Q, int ncols, float* out)
{
    if (!ctx || !Q || !out) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (int rcd_ = ensure_dense(ctx)) return rcd_;
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

#ifdef CNMF_DEBUG_ABI          // test hooks (include/cnmf_hip_debug.h): not compiled into a product build
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
    refresh_gemm3_mode(ctx);
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

// C[KC][J] = A[KC][K] . Bn[J][K]^T through the f16 two-plane count kernel (kernels_gemm2h.hip.h): Bn holds
// non-negative integers <= 65535, A arbitrary non-negative float32.  KC % 256 == 0, K % 64 == 0.  nsub = 1 | 2 sub-blocks
// per barrier pair (2 only without a second count plane, K % 32 == 0).
// The row-scale bound the W half-step reports instead of the exact maximum (kernels_sweep.hip.h, RMX = false):
// sqrt(sum of w^2 over the <= 1024 cells of one sweep workgroup) * 1.0001, one partial per workgroup.
// grid = (parts, rows / 4): one wave per row and part.
__global__ __launch_bounds__(256) void debug_rowbound_part_kernel(const float* __restrict__ V, int ld, int L, int span,
                                                                  int parts, float* __restrict__ rmax_part)
{
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, p = blockIdx.x;
    const int j0 = p * span, j1 = min(L, j0 + span);
    float s = 0.f;
    for (int j = j0 + lane; j < j1; j += 64) { const float x = V[(size_t)row * ld + j]; s = fmaf(x, x, s); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) rmax_part[(size_t)row * parts + p] = sqrtf(s) * 1.0001f;
}

extern "C" int cnmf_debug_gemm2h(cnmf_ctx* ctx, const float* A, const float* Bn, float* C, int KC, int K, int J,
                                 int nsplit, int nsub, double* ms_out, int reps)
{
    const bool sweep_bound = (nsub & 128) != 0;          // bit 7: scale the rows by the W half-step's BOUND, not the exact maximum
    nsub &= 127;
    if (!ctx || !A || !Bn || !C) { SET_ERR(ctx, "null argument"); return CNMF_EINVAL; }
    if (KC % 256 || K % 64 || J < 1 || nsplit < 1) { SET_ERR(ctx, "debug_gemm2h needs KC %% 256 == 0, K %% 64 == 0"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int Jp = round_up(J, G3C_JW), Kb = K / 16;
    bool has_hi = false;
    for (size_t i = 0; i < (size_t)J * K && !has_hi; ++i) has_hi = Bn[i] > G2_COUNT_BASE;
    DevPool pool;
    EventPool events;
    float* dA = pool.get<float>((size_t)KC * K);
    float* dB = pool.get<float>((size_t)J * K);
    float* dUnit = pool.get<float>(K);
    const int bparts = sweep_bound ? (K + 1023) / 1024 : 1;
    float* dRmax = pool.get<float>((size_t)KC * bparts);
    float* dInv = pool.get<float>(KC);
    unsigned char* dA2 = pool.get<unsigned char>((size_t)KC * Kb * G2_ROWB);
    unsigned char* dB1 = pool.get<unsigned char>((size_t)Jp * Kb * 32);
    unsigned char* dBh = has_hi ? pool.get<unsigned char>((size_t)Jp * Kb * 32) : nullptr;
    unsigned int* dFl = pool.get<unsigned int>((size_t)(Jp / G3C_JW) * ((Kb + 31) / 32), true, st);
    float* dC = pool.get<float>((size_t)nsplit * KC * Jp);
    hipEvent_t e0 = events.get(), e1 = events.get();
    POOL_TRY(ctx, pool);
    POOL_TRY(ctx, events);
    std::vector<float> ones(K, 1.0f);
    HIP_TRY(ctx, hipMemcpyAsync(dA, A, (size_t)KC * K * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dB, Bn, (size_t)J * K * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dUnit, ones.data(), (size_t)K * sizeof(float), hipMemcpyHostToDevice, st));
    if (sweep_bound) {
        debug_rowbound_part_kernel<<<dim3(bparts, KC / 4), 256, 0, st>>>(dA, K, K, 1024, bparts, dRmax);
        HIP_TRY(ctx, hipGetLastError());
    } else
        HIP_TRY(ctx, launch_rowmax_part(st, dA, K, K, KC, K, nullptr, 1, dRmax));
    HIP_TRY(ctx, launch_split2h(st, dA, K, KC, K, dA2, G3_MW, nullptr, dRmax, bparts, dInv));
    {
        const long long total = (long long)Jp * Kb;
        count_planes_f16_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(dB, K, J, K, Jp, K, G3C_JW, dUnit,
                                                                               (unsigned short*)dB1, (unsigned short*)dBh, dFl);
        HIP_TRY(ctx, hipGetLastError());
    }
    reps = std::max(1, reps);
    const int var = (nsub >> 4) & 7;                     // (upper bits of `nsub`: instruction-stream variant, A/B probes)
    nsub &= 15;
    const int ns = (!has_hi && nsub == 2 && Kb % 2 == 0) ? 2 : 1;
    for (int i = 0; i < reps + 1; ++i) {
        if (i == 1) hipEventRecord(e0, st);
        hipError_t e;
        if (has_hi) e = launch_gemm2h_t<1, true>(st, dA2, dB1, dBh, dFl, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else if (ns == 2 && var == 1) e = launch_gemm2h_t<2, false, 1>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else if (ns == 2 && var == 2) e = launch_gemm2h_t<2, false, 2>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else if (ns == 2 && var == 3) e = launch_gemm2h_t<2, false, 3>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else if (ns == 2 && var == 4) e = launch_gemm2h_t<2, false, 4>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else if (ns == 2 && var == 5) e = launch_gemm2h_t<2, false, 5>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else if (ns == 2 && var == 6) e = launch_gemm2h_t<2, false, 6>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else if (ns == 2 && var == 7) e = launch_gemm2h_t<2, false, 7>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else if (ns == 2) e = launch_gemm2h_t<2, false, 0>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        else e = launch_gemm2h_t<1, false>(st, dA2, dB1, nullptr, nullptr, dInv, Kb, dC, Jp, (long long)KC * Jp, KC, Jp, nsplit);
        HIP_TRY(ctx, e);
    }
    hipEventRecord(e1, st);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const int zs = gemm2h_splits(Kb, nsplit, ns);
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

// ---- calibration streams for the PMC byte counters (tools/pmc_calibrate.py): a copy of `n_floats` floats with the
// access width of the sweeps (4 B per lane: width = 1) or of the plane kernels (16 B per lane: width = 4), and a read-only
// LDS-DMA stream (global_load_lds_dwordx4, what the GEMMs use: width = 0).  Known bytes -> FETCH_SIZE / WRITE_SIZE ratios.
namespace cnmf {
__global__ __launch_bounds__(256) void calib_copy1_kernel(const float* __restrict__ a, float* __restrict__ b, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void calib_copy4_kernel(const float4* __restrict__ a, float4* __restrict__ b, long long n4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void calib_ldsdma_kernel(const unsigned char* __restrict__ a, long long nbytes, float* __restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf[4][4096];
    const long long chunk = 4096;                                   // one 16-B load per lane x 256 lanes
    float acc = 0.f;
    for (long long off = (long long)blockIdx.x * chunk; off + chunk <= nbytes; off += (long long)gridDim.x * chunk) {
        const int w = (threadIdx.x >> 6);
        __builtin_amdgcn_global_load_lds(G3_AS1(a + off + threadIdx.x * 16), G3_AS3(&buf[0][0] + w * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += reinterpret_cast<const float*>(&buf[0][0])[threadIdx.x];
    }
    if (acc == 1.2345f) sink[0] = acc;                              // keep the loads alive
}
}  // namespace cnmf

extern "C" int cnmf_debug_stream(cnmf_ctx* ctx, int width, long long n_floats, int reps)
{
    using namespace cnmf;
    if (!ctx || n_floats < 1024 || reps < 1) { SET_ERR(ctx, "bad argument"); return CNMF_EINVAL; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevPool pool;
    n_floats = (n_floats / 1024) * 1024;
    float* a = pool.get<float>((size_t)n_floats, true, st);
    float* b = pool.get<float>((size_t)n_floats, true, st);
    POOL_TRY(ctx, pool);
    for (int r = 0; r < reps; ++r) {
        if (width == 1) calib_copy1_kernel<<<4096, 256, 0, st>>>(a, b, n_floats);
        else if (width == 4) calib_copy4_kernel<<<4096, 256, 0, st>>>((const float4*)a, (float4*)b, n_floats / 4);
        else calib_ldsdma_kernel<<<2048, 256, 0, st>>>((const unsigned char*)a, n_floats * 4, b);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
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
#endif  // CNMF_DEBUG_ABI
