// MFMA-only instruction streams on gfx950: what does the matrix pipe sustain on NON-ZERO data (the power budget
// decides the clock, MI355X_MICROARCH.md "DVFS give-back") for the f16 and the i8 opcodes?  Round-4 go/no-go input for an
// int8 count path: counts <= 127 in one i8 plane x three signed base-256 digits of the factor
// = 3 i8 MFMAs per product against today's 2 f16 MFMAs -- worth it only if the i8 stream runs >= 1.5 x the f16 one.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_stream_probe.hip -o gpurun_out/mfma_stream_probe && ./gpurun_out/mfma_stream_probe
// One workgroup of 512 threads per CU-slot (2 waves per SIMD, like the production GEMM), NACC independent accumulators
// per wave, operands random and held in registers (no memory traffic inside the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int NACC = 8;

template <int MODE>
__global__ __launch_bounds__(512) void stream_kernel(const int* __restrict__ seed, int iters, float* __restrict__ out)
{
    const int t = blockIdx.x * 512 + threadIdx.x;
    // per-lane pseudo-random operands (LCG on the lane id and a loaded seed: the compiler cannot fold them)
    unsigned s = (unsigned)seed[t & 1023] * 2654435761u + (unsigned)t * 40503u + 1u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    if constexpr (MODE == 0) {                       // v_mfma_f32_32x32x16_f16
        f16x8 a[2], b[2];
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 8; ++e) {
                a[i][e] = (_Float16)((float)(rnd() >> 21) * (1.0f / 64.0f) - 8.0f);
                b[i][e] = (_Float16)(float)(rnd() >> 26);          // small integers, like counts
            }
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
        }
        float r = 0.f;
        for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) r += acc[i][q];
        out[t] = r;
    } else if constexpr (MODE == 1) {                // v_mfma_i32_32x32x32_i8
        i32x4 a[2], b[2];
        for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) { a[i][e] = (int)rnd(); b[i][e] = (int)(rnd() & 0x0f0f0f0fu); }
        i32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
        }
        int r = 0;
        for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) r += acc[i][q];
        out[t] = (float)r;
    } else {                                         // v_mfma_i32_16x16x64_i8
        i32x4 a[2], b[2];
        for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) { a[i][e] = (int)rnd(); b[i][e] = (int)(rnd() & 0x0f0f0f0fu); }
        i32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
        }
        int r = 0;
        for (int i = 0; i < NACC; ++i) for (int q = 0; q < 4; ++q) r += acc[i][q];
        out[t] = (float)r;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int wgs = argc > 2 ? atoi(argv[2]) : 256;
    int* seed; float* out;
    CK(hipMalloc(&seed, 1024 * sizeof(int)));
    CK(hipMalloc(&out, (size_t)wgs * 512 * sizeof(float)));
    std::vector<int> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = rand();
    CK(hipMemcpy(seed, h.data(), 1024 * sizeof(int), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"v_mfma_f32_32x32x16_f16", "v_mfma_i32_32x32x32_i8", "v_mfma_i32_16x16x64_i8"};
    const double ops_per_mfma[3] = {2.0 * 32 * 32 * 16, 2.0 * 32 * 32 * 32, 2.0 * 16 * 16 * 64};
    printf("{\"iters\": %d, \"workgroups\": %d, \"waves_per_simd\": 2, \"accumulators_per_wave\": %d, \"streams\": [", iters, wgs, NACC);
    for (int mode = 0; mode < 3; ++mode) {
        double best = 0.0, sum = 0.0;
        const int reps = 5;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) stream_kernel<0><<<wgs, 512>>>(seed, iters, out);
            else if (mode == 1) stream_kernel<1><<<wgs, 512>>>(seed, iters, out);
            else stream_kernel<2><<<wgs, 512>>>(seed, iters, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 0) continue;                  // warm-up
            const double tops = ops_per_mfma[mode] * NACC * (double)iters * wgs * 8 / (ms * 1e-3) * 1e-12;
            best = tops > best ? tops : best; sum += tops;
        }
        printf("%s{\"op\": \"%s\", \"tera_ops_best\": %.1f, \"tera_ops_mean\": %.1f}", mode ? ", " : "", names[mode], best, sum / reps);
    }
    printf("]}\n");
    return 0;
}
