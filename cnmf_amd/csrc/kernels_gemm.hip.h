// MFMA skinny-GEMM kernels for the batched-restart NMF engine (gfx950 only).
//
// Every product on the hot path has the form
//
//     C[c][j] = sum_k A[c][k] * B(j,k)
//
// where c runs over the packed component columns of ALL restarts in flight
// (KC = sum of ranks, padded to a multiple of 32), A is a component-major
// factor (H_all [KC][G] or Wt_all [KC][N]) and B is the shared data matrix X:
//
//   pass A  (sklearn _nmf.py:387, XHt = X @ Ht):   B(j,k) = X[j][k]   j=cell, k=gene   ("NT")
//   pass B  (sklearn _nmf.py:505, XtW = X.T @ W):  B(j,k) = X[k][j]   j=gene, k=cell   ("NN")
//
// so X is streamed ONCE per pass for all restarts in the batch (arithmetic
// intensity KC/2 flop/B instead of k/2).  The contraction runs on the exact-f32
// matrix pipe: v_mfma_f32_32x32x2_f32 (64 cyc/SIMD, 157 TF chip peak).
//
// MFMA operand roles: rows (i = lane&31) = component c, cols (j = lane&31) = j.
//   A operand lane l : A[i=l&31][k=l>>5]     B operand lane l : B[k=l>>5][j=l&31]
//   D reg r, lane l  : row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31
// The contraction index inside one 8-wide k group is permuted (half h=l>>5 takes
// k = 8q+4h+i at MFMA step i) so that the K-contiguous operands are fetched from
// LDS with ONE ds_read_b128 per 4 MFMAs; any pairing of k's is legal as long as
// A and B use the same one.
//
// LDS tiles (BK = 32 floats of k per stage, double buffered, one barrier/stage):
//   K-contiguous operand : [rows][36]   (pad 4 -> ds_read_b128 conflict-free: slot = 9*row mod 16)
//   NN-mode X tile       : [32 k][JW]   (lanes run along j -> ds_read_b32 conflict-free)
#pragma once
#include <hip/hip_runtime.h>

namespace cnmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BK = 32;        // default k extent of one LDS stage (template parameter TBK)

// MTW : 32-row component tiles per wave        WM x WN : wave grid (WM*WN == 4)
// NN  : false -> B is [J][ldb] K-contiguous (pass A); true -> B is [K][ldb] J-contiguous (pass B)
template <int MTW, int WM, int WN, bool NN, int TBK>
__device__ __forceinline__ void gemm_segment(
    const float* __restrict__ A, int lda,
    const float* __restrict__ B, int ldb,
    float* __restrict__ C, int ldc,
    int m0, int j0, int kbeg, int nk, int Jtot, float* smem)
{
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BK = TBK;             // shadows the namespace default inside this function
    constexpr int LDK = BK + 4;         // padded row of a K-contiguous LDS tile (ds_read_b128 conflict-free)
    constexpr int QK = BK / 4;          // float4 per K-contiguous row
    constexpr int RP = 256 / QK;        // rows staged per pass of the 256 threads
    constexpr int MW = WM * MTW * 32;   // component rows per workgroup
    constexpr int JW = WN * 32;         // j columns per workgroup
    constexpr int A_F4 = MW / RP;       // float4 per thread per A stage
    constexpr int B_F4 = NN ? (BK * JW) / 1024 : JW / RP;   // float4 per thread per B stage
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for the staging map");
    constexpr int A_TILE = MW * LDK;
    constexpr int B_TILE = NN ? (BK * JW) : (JW * LDK);

    float* As = smem;                   // [2][A_TILE]
    float* Bs = smem + 2 * A_TILE;      // [2][B_TILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, h = lane >> 5;

    // ---- global -> register staging addresses
    const int a_row = tid / QK, a_kq = tid % QK;           // QK float4 per K-contiguous row
    const float* a_src = A + (size_t)(m0 + a_row) * lda + kbeg + a_kq * 4;
    v4f a_reg[A_F4];
    v4f b_reg[B_F4];

    const float* b_src;
    int b_jq = 0, b_kr = 0;
    if (!NN) {
        b_src = B + (size_t)(j0 + a_row) * ldb + kbeg + a_kq * 4;
    } else {
        constexpr int QPR = JW / 4;                        // float4 per k row
        b_kr = tid / QPR; b_jq = tid % QPR;
        // columns j >= Jtot read into the next row of the (over-allocated) buffer; they only
        // feed output columns that are never stored, so no guard (= no branch) is needed
        b_src = B + (size_t)(kbeg + b_kr) * ldb + j0 + b_jq * 4;
    }

#define CNMF_LOAD_STAGE(kt_)                                                                         \
    {                                                                                                \
        _Pragma("unroll") for (int i = 0; i < A_F4; ++i)                                             \
            a_reg[i] = *reinterpret_cast<const v4f*>(a_src + (size_t)(RP * i) * lda + (kt_) * BK); \
        if (!NN) {                                                                                   \
            _Pragma("unroll") for (int i = 0; i < B_F4; ++i)                                         \
                b_reg[i] = *reinterpret_cast<const v4f*>(b_src + (size_t)(RP * i) * ldb + (kt_) * BK); \
        } else {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < B_F4; ++i)                                         \
                b_reg[i] = *reinterpret_cast<const v4f*>(b_src + (size_t)((kt_) * BK + RPI * i) * ldb); \
        }                                                                                            \
    }
#define CNMF_STORE_STAGE(buf_)                                                                       \
    {                                                                                                \
        float* as_ = As + (buf_) * A_TILE;                                                           \
        float* bs_ = Bs + (buf_) * B_TILE;                                                           \
        _Pragma("unroll") for (int i = 0; i < A_F4; ++i)                                             \
            *reinterpret_cast<v4f*>(as_ + (a_row + RP * i) * LDK + a_kq * 4) = a_reg[i];          \
        if (!NN) {                                                                                   \
            _Pragma("unroll") for (int i = 0; i < B_F4; ++i)                                         \
                *reinterpret_cast<v4f*>(bs_ + (a_row + RP * i) * LDK + a_kq * 4) = b_reg[i];      \
        } else {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < B_F4; ++i)                                         \
                *reinterpret_cast<v4f*>(bs_ + (b_kr + RPI * i) * JW + b_jq * 4) = b_reg[i];       \
        }                                                                                            \
    }
    constexpr int RPI = NN ? 256 / (JW / 4) : 0;           // k rows covered per staging step (NN)

    f32x16 acc[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // a wave whose j tile lies wholly in the padding does no math / no stores
    const bool wave_live = (j0 + wn * 32) < Jtot;

    if (nk > 0) {
        CNMF_LOAD_STAGE(0)
        CNMF_STORE_STAGE(0)
    }
    __syncthreads();

    // Branch-free main loop: the prefetch of the last stage is clamped (re-loads the final
    // tile) and every wave computes, live or not -- a conditional around the MFMA block makes
    // the compiler shuttle all accumulators VGPR<->AGPR on every stage.
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const int ktn = min(kt + 1, nk - 1);
        CNMF_LOAD_STAGE(ktn)
        // keep the prefetch at the top of the stage (hipcc otherwise sinks the loads next to
        // their ds_write consumers and exposes the whole HBM latency every stage)
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* as = As + buf * A_TILE + (wm * MTW * 32 + li) * LDK + h * 4;
            const float* bs = NN ? (Bs + buf * B_TILE + (h * 4) * JW + wn * 32 + li)
                                 : (Bs + buf * B_TILE + (wn * 32 + li) * LDK + h * 4);
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                v4f a4[MTW];
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    a4[m] = *reinterpret_cast<const v4f*>(as + m * 32 * LDK + q * 8);
                float b4[4];
                if (!NN) {
                    v4f t = *reinterpret_cast<const v4f*>(bs + q * 8);
                    b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) b4[i] = bs[(q * 8 + i) * JW];
                }
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].x, b4[0], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].y, b4[1], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].z, b4[2], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].w, b4[3], acc[m], 0, 0, 0);
                }
            }
        }
        CNMF_STORE_STAGE(buf ^ 1)
        __syncthreads();
    }

    if (wave_live) {
        float* c = C;
        const int j = j0 + wn * 32 + li;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int cbase = m0 + (wm * MTW + m) * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = cbase + (r & 3) + 8 * (r >> 2);
                c[(size_t)row * ldc + j] = acc[m][r];
            }
        }
    }
}

#undef CNMF_LOAD_STAGE
#undef CNMF_STORE_STAGE

// grid-mapped launch: blockIdx = (j tile, component group, K split)
template <int MTW, int WM, int WN, bool NN, int TBK = BK>
__global__ __launch_bounds__(256) void gemm_kernel(
    const float* __restrict__ A, int lda,
    const float* __restrict__ B, int ldb,
    float* __restrict__ C, int ldc, long long c_split_stride,
    int Kper, int Ktot, int Jtot)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MW = WM * MTW * 32, JW = WN * 32;
    const int kbeg = blockIdx.z * Kper;
    const int kend = min(kbeg + Kper, Ktot);
    gemm_segment<MTW, WM, WN, NN, TBK>(A, lda, B, ldb, C + (size_t)blockIdx.z * c_split_stride, ldc,
                                       blockIdx.y * MW, blockIdx.x * JW, kbeg, (kend - kbeg) / TBK, Jtot, smem);
}

// stream-K launch (pass A): `gridDim.x` persistent workgroups share T tiles x nk stages evenly.
// Unit u = tile * nk + stage, tile = jt * MG + mg (component group fastest, so the groups of one
// j tile sit in the same / neighbouring workgroup and share the X tile in L2).  With at least nk
// units per workgroup a tile is cut at most once: the piece that starts at stage 0 goes to plane 0,
// the piece that ends at stage nk to plane 1 (sweep_kernel adds plane 1 where `split[tile]`).
template <int MTW, int WM, int WN, bool NN, int TBK = BK>
__global__ __launch_bounds__(256) void gemm_streamk_kernel(
    const float* __restrict__ A, int lda,
    const float* __restrict__ B, int ldb,
    float* __restrict__ C0, float* __restrict__ C1, int ldc,
    int MG, int T, int nk, int Jtot)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MW = WM * MTW * 32, JW = WN * 32;
    const long long U = (long long)T * nk;
    long long u = U * blockIdx.x / gridDim.x;
    const long long u1 = U * (blockIdx.x + 1) / gridDim.x;
    while (u < u1) {
        const int tile = (int)(u / nk), kb = (int)(u % nk);
        const int ke = (int)min((long long)nk, kb + (u1 - u));
        const int jt = tile / MG, mg = tile % MG;
        gemm_segment<MTW, WM, WN, NN, TBK>(A, lda, B, ldb, (kb == 0) ? C0 : C1, ldc, mg * MW, jt * JW,
                                           kb * TBK, ke - kb, Jtot, smem);
        u += ke - kb;
        __syncthreads();                 // LDS stages are reused by the next segment
    }
}

template <int MTW, int WM, int WN, bool NN, int TBK = BK>
constexpr size_t gemm_lds_bytes() {
    return sizeof(float) * 2 * (size_t)(WM * MTW * 32 * (TBK + 4) + (NN ? TBK * WN * 32 : WN * 32 * (TBK + 4)));
}

}  // namespace cnmf
