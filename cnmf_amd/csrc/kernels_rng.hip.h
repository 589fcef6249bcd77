// sklearn's init='random' reproduced on the device (gfx950).
//
// sklearn/decomposition/_nmf.py:302-314 draws, from numpy's legacy
// RandomState(seed),  H = avg*standard_normal((k,G))  FIRST and then
// W = avg*standard_normal((N,k)), casts to X's dtype and takes |.|.
// RandomState(int seed) = MT19937 seeded with init_genrand; standard_normal is the
// legacy polar (Marsaglia) method:  every candidate consumes exactly FOUR 32-bit
// words (two 53-bit doubles x1,x2 in (-1,1)); it is accepted iff 0 < r2 = x1^2+x2^2 < 1
// and then yields  f*x2  followed by  f*x1,  f = sqrt(-2 log(r2)/r2).  Because the
// words per candidate are fixed, the stream parallelises: one workgroup per restart
// regenerates the 624-word state with the 3-phase parallel twist, evaluates the 156
// candidates of the block concurrently and compacts the accepted ones with a prefix sum.
//
// Host numpy is far too slow to feed the GPU ((N+G)*k normals per restart at ~20 ns
// each = 10 ms per 50k-cell restart; the engine retires a restart every ~1 ms).
// Arithmetic is IEEE double with contraction OFF (numpy's C is built without FMA);
// the only non-bit-exact step is log(): ocml vs glibc may differ by 1 ulp(double),
// which survives the cast to fp32 with probability ~2^-29 per value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnmf {

constexpr int MT_N = 624, MT_M = 397;

__device__ __forceinline__ uint32_t mt_mix(uint32_t cur, uint32_t nxt, uint32_t far_)
{
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far_ ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// MODE 0: out_d[o] = z (raw doubles, for the known-answer test)
// MODE 1: scatter |float(avg*z)| into H_all rows (o < k*G) and Wt_all rows (o >= k*G)
struct RngJob {
    uint32_t seed; int k; int off;
    double avg;
    long long total;      // number of normals to produce
};

template <int MODE>
__global__ __launch_bounds__(256) void rng_kernel(const RngJob* __restrict__ jobs,
                                                  double* __restrict__ out_d,
                                                  float* __restrict__ H, int ldh, int G,
                                                  float* __restrict__ Wt, int ldw, int N)
{
#pragma clang fp contract(off)
    __shared__ uint32_t st[2][MT_N];
    __shared__ uint32_t tw[MT_N];
    __shared__ int wsum[4];
    const RngJob job = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        uint32_t s = job.seed;
        st[0][0] = s;
        for (int i = 1; i < MT_N; ++i) { s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i; st[0][i] = s; }
    }
    __syncthreads();
    int cur = 0;
    long long produced = 0;           // normals emitted so far (uniform)
    const long long kG = (long long)job.k * G;
    while (produced < job.total) {
        uint32_t* o = st[cur];
        uint32_t* n = st[cur ^ 1];
        // ---- 3-phase parallel twist (new[i] needs old[i], old[i+1] and old[i+397] / new[i-227])
        if (tid < MT_N - MT_M) { const uint32_t v = mt_mix(o[tid], o[tid + 1], o[tid + MT_M]); n[tid] = v; tw[tid] = mt_temper(v); }
        __syncthreads();
        if (tid < MT_N - MT_M) { const int i = tid + (MT_N - MT_M); const uint32_t v = mt_mix(o[i], o[i + 1], n[i - (MT_N - MT_M)]); n[i] = v; tw[i] = mt_temper(v); }
        __syncthreads();
        if (tid < MT_N - 2 * (MT_N - MT_M)) {
            const int i = tid + 2 * (MT_N - MT_M);
            const uint32_t nxt = (i == MT_N - 1) ? n[0] : o[i + 1];
            const uint32_t v = mt_mix(o[i], nxt, n[i - (MT_N - MT_M)]);
            n[i] = v; tw[i] = mt_temper(v);
        }
        __syncthreads();
        cur ^= 1;
        // ---- 156 candidates
        bool acc = false; double z0 = 0.0, z1 = 0.0;
        if (tid < MT_N / 4) {
            const uint32_t a0 = tw[4 * tid] >> 5, b0 = tw[4 * tid + 1] >> 6;
            const uint32_t a1 = tw[4 * tid + 2] >> 5, b1 = tw[4 * tid + 3] >> 6;
            const double d0 = ((double)a0 * 67108864.0 + (double)b0) / 9007199254740992.0;
            const double d1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0;
            const double x1 = 2.0 * d0 - 1.0, x2 = 2.0 * d1 - 1.0;
            const double r2 = x1 * x1 + x2 * x2;
            acc = (r2 < 1.0) && (r2 != 0.0);
            if (acc) {
                const double f = sqrt(-2.0 * log(r2) / r2);
                z0 = f * x2;          // returned first
                z1 = f * x1;          // cached "gauss", returned by the next call
            }
        }
        // ---- prefix sum of the accept flags (wave ballot + cross-wave)
        const unsigned long long bal = __ballot(acc);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) before += wsum[w]; tot += wsum[w]; }
        if (acc) {
            const long long p = produced + 2ll * (before + within);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long long oidx = p + e;
                const double z = e ? z1 : z0;
                if (oidx < job.total) {
                    if (MODE == 0) out_d[oidx] = z;
                    else {
                        const float v = fabsf((float)(job.avg * z));
                        if (oidx < kG) {
                            const int c = (int)(oidx / G), g = (int)(oidx % G);
                            H[(size_t)(job.off + c) * ldh + g] = v;
                        } else {
                            const long long q = oidx - kG;
                            const int i = (int)(q / job.k), c = (int)(q % job.k);
                            Wt[(size_t)(job.off + c) * ldw + i] = v;
                        }
                    }
                }
            }
        }
        produced += 2ll * tot;
        __syncthreads();     // wsum / tw reuse
    }
}

// out_d[0..n) = RandomState(seed).standard_normal(n)   (known-answer test hook)
static inline void launch_standard_normal(hipStream_t st, uint32_t seed, long long n, double* out_d)
{
    RngJob h{seed, 1, 0, 1.0, n};
    RngJob* d = nullptr;
    if (hipMalloc(&d, sizeof(RngJob)) != hipSuccess) return;
    hipMemcpyAsync(d, &h, sizeof h, hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);   // h is a stack object
    rng_kernel<0><<<1, 256, 0, st>>>(d, out_d, nullptr, 0, 1, nullptr, 0, 1);
    hipStreamSynchronize(st);
    hipFree(d);
}

}  // namespace cnmf
