// Kullback-Leibler multiplicative updates on the NON-ZEROS of X (round 4).
//
// For beta = 1 the quotient X / (WH) vanishes wherever X does, so both half-steps and the divergence only ever touch the
// stored entries -- what scikit-learn does for scipy.sparse input (sklearn/decomposition/_nmf.py:84-194 `_beta_divergence`
// with `_special_sparse_dot`, :526-631 `_multiplicative_update_w`, :634-728 `_multiplicative_update_h`), and what cNMF hands
// it whenever the normalised counts are stored sparse (cnmf.py:618-631).  A real count matrix holds 5-10 % non-zeros.
//
// Layout: a blocked, sliced ELL image (`BSell`) of the matrix per orientation.  The OWN side of a half-step (cells for the
// W update, genes for the H update) is cut in slices of 64 rows = one wavefront, a lane per row, the row's factor in registers;
// the OTHER side is cut in blocks of BS rows whose factor block (BS x KP float32 = 128 KB) sits in LDS for the whole
// workgroup.  Entries of (slice, block) are stored position-major -- entry t of the 64 rows side by side, 8 bytes each
// {row of the other side inside its block, value} -- so a wavefront reads 512 contiguous bytes per step; a slice is padded to
// its longest row (rows are sorted by their non-zero count first, so the padding stays at a few per cent).  Per entry a lane
// gathers KP floats from LDS (quads swizzled by the row so that random rows spread over all banks), forms w.h, the quotient
// and the k-vector update: 2 KP + ~8 vector operations and 4 KP bytes of LDS traffic per non-zero and half-step.
// The entry stream (290 MB at 200 000 x 2 000 / 9 %) is read from HBM once for ALL restarts in flight: the workgroups of
// one tile for the restarts of the batch are adjacent in the dispatch order of ONE XCD, so the others hit its L2.
// Measured there: 108 us per restart-iteration at ranks up to 13 (dense matrix-pipe kernels: 340), DESIGN.md section 4.
//
// Summation order is fixed (entries of a row in storage order, blocks in index order): results are run-to-run identical.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_mu.hip.h"
#include "kernels_mu_mfma.hip.h"

namespace cnmf {

struct BSellDev {
    int R, C, BS, nblk, nslice;     // own rows, other rows, block size (other side), blocks, slices of 64 own rows
    const int* perm;                // [nslice * 64]: own row of (slice, lane), -1 = none
    const long long* off;           // [nslice * nblk]: first entry of (slice, block)
    const int* len;                 // [nslice * nblk]: entries per lane of (slice, block)
    const uint2* ent;               // {LDS byte offset of the other side's row (quad 0 of the rotated row), float bits}
};

constexpr int SP_WAVES = 16;                      // slices per workgroup (1024 threads: 4 waves per SIMD, one workgroup per CU)
constexpr int SP_LDS_BYTES = 128 * 1024;
#ifndef CNMF_SP_PF
#define CNMF_SP_PF 1                              // (2 measured in round 6: 128 registers + 4-5 spilled, 117 vs 113-115 us per
#endif                                            //  restart-iteration at 200 000 x 2 000 / 9 %: profiles/r6_mu_sparse_prefetch_ab.txt)
constexpr int SP_PF = CNMF_SP_PF;                 // trips the entry stream is requested ahead of the gathers
constexpr int SP_UNROLL = 4;                      // lengths are padded to it: up to four entries per lane and trip

// ---- build, pass 1: non-zeros of every row of M [R][ld] inside every block of BS columns
__global__ __launch_bounds__(256) void sp_count_kernel(const float* __restrict__ M, int ld, int R, int C, int BS, int nblk,
                                                       int* __restrict__ cnt)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* m = M + (size_t)row * ld;
    for (int b = 0; b < nblk; ++b) {
        const int c0 = b * BS, c1 = min(C, c0 + BS);
        int n = 0;
        for (int c = c0 + lane; c < c1; c += 64) n += (m[c] != 0.f) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
        if (lane == 0) cnt[(size_t)row * nblk + b] = n;
    }
}

// ---- build, pass 2: a wavefront per (slice, lane) position walks its row and drops the non-zeros of every block into the
// position-major entry array (which was zero-filled: {0, 0.0f} is the padding entry -- it adds nothing to any sum)
// An entry names its row of the other side by the LDS byte offset of that row's quad 0 (sp_quad below): row r of a
// [BS][KP] float block whose quads are rotated by r / (rows per 256-byte bank line).
__global__ __launch_bounds__(256) void sp_fill_kernel(const float* __restrict__ M, int ld, int C, int BS, int nblk, int npos,
                                                      const int* __restrict__ perm, const long long* __restrict__ off,
                                                      uint2* __restrict__ ent, int KP)
{
    const unsigned QPR = (unsigned)KP / 4u, RPL = 16u / QPR;
    const int pos = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pos >= npos) return;
    const int row = perm[pos];
    if (row < 0) return;
    const int s = pos >> 6, l = pos & 63;
    const float* m = M + (size_t)row * ld;
    for (int b = 0; b < nblk; ++b) {
        const int c0 = b * BS, c1 = min(C, c0 + BS);
        uint2* e = ent + off[(size_t)s * nblk + b] + l;
        int base = 0;
        for (int cb = c0; cb < c1; cb += 64) {
            const int c = cb + lane;
            const float x = c < c1 ? m[c] : 0.f;
            const bool nz = x != 0.f;
            const unsigned long long mask = __ballot(nz);
            if (nz) {
                const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
                const unsigned r = (unsigned)(c - c0);
                e[(size_t)idx * 64] = uint2{r * (unsigned)(KP * 4) + (((r / RPL) & (QPR - 1u)) << 4), __float_as_uint(x)};
            }
            base += __popcll(mask);
        }
    }
}

// ---- build from the compressed rows (round 5: csr_host.hip.h keeps them from the CSR upload / builds X^T's on the device):
// the same two passes without touching the N x G dense image.  Rows list ascending columns, so the entries of a block are
// contiguous and the image comes out IDENTICAL to the one built from the dense matrix.
__global__ __launch_bounds__(256) void sp_count_csr_kernel(const long long* __restrict__ ptr, const int* __restrict__ idx, int R,
                                                           int BS, int nblk, int* __restrict__ cnt /* zeroed */)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const long long b = ptr[row], e = ptr[row + 1];
    if (nblk == 1) { if (lane == 0) cnt[row] = (int)(e - b); return; }
    // the entries of a block form a run; a lane that sits on a run's LAST entry knows where the run ends:
    // count(block) = end(block) - end(previous non-empty block), as two integer atomics per run (order-independent)
    for (long long p = b + lane; p < e; p += 64) {
        const int blk = idx[p] / BS;
        const int nxt = (p + 1 < e) ? idx[p + 1] / BS : -1;
        if (nxt != blk) atomicAdd(&cnt[(size_t)row * nblk + blk], (int)(p - b + 1));
        if (nxt != blk && nxt >= 0) atomicAdd(&cnt[(size_t)row * nblk + nxt], -(int)(p - b + 1));
    }
}

// pre[row][b] = entries of the row in blocks 0 .. b - 1
__global__ __launch_bounds__(256) void sp_prefix_kernel(const int* __restrict__ cnt, int R, int nblk, int* __restrict__ pre)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= R) return;
    int run = 0;
    for (int b = 0; b < nblk; ++b) { pre[(size_t)row * nblk + b] = run; run += cnt[(size_t)row * nblk + b]; }
}

__global__ __launch_bounds__(256) void sp_fill_csr_kernel(const long long* __restrict__ ptr, const int* __restrict__ idx,
                                                          const float* __restrict__ val, int BS, int nblk, int npos,
                                                          const int* __restrict__ perm, const long long* __restrict__ off,
                                                          const int* __restrict__ pre, uint2* __restrict__ ent, int KP)
{
    const unsigned QPR = (unsigned)KP / 4u, RPL = 16u / QPR;
    const int pos = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pos >= npos) return;
    const int row = perm[pos];
    if (row < 0) return;
    const int s = pos >> 6, l = pos & 63;
    const long long b = ptr[row], e = ptr[row + 1];
    for (long long p = b + lane; p < e; p += 64) {
        const int c = idx[p], blk = c / BS;
        const unsigned r = (unsigned)(c - blk * BS);
        const int t = (int)(p - b) - pre[(size_t)row * nblk + blk];
        ent[off[(size_t)s * nblk + blk] + l + (size_t)t * 64] =
            uint2{r * (unsigned)(KP * 4) + (((r / RPL) & (QPR - 1u)) << 4), __float_as_uint(val[p])};
    }
}

// ---- build, pass 3: the order of a row's entries is free -- choose it so that the gather is (nearly) free of LDS bank
// conflicts.  A `ds_read_b128` is served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table); inside a group two
// lanes collide when their rows agree modulo 16 (the rotation of the quads keeps that a bijection onto the 16 quad slots).
// In storage order the 16 rows of a step are random: the fullest slot holds 3 of them on average, and the counters showed 62 %
// of the LDS cycles as conflicts.  Here one thread per (slice, block, lane group) re-orders the 16 entry lists: every
// step, the lanes (longest row first) take an entry of a residue nobody took yet, the one they hold most of -- a greedy
// edge colouring of the lanes x residues multigraph, 1.3-1.5 rows on the fullest slot instead of 3 in simulation.  No
// entry is added or dropped; padding entries (value bits 0) are pointed at a row of a free residue.
__device__ __constant__ unsigned char SP_LANE_GROUPS[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

__global__ __launch_bounds__(64) void sp_reorder_kernel(int nslice, int nblk, int C, int BS, int KP,
                                                        const long long* __restrict__ off, const int* __restrict__ len,
                                                        uint2* __restrict__ ent, uint2* __restrict__ tmp)
{
    __shared__ unsigned char cnt[16][16][64];                          // [lane of the group][residue][thread]: entries left
    __shared__ unsigned short start[16][16][64];                       // first position of that bucket in the lane's list
    const long long job = (long long)blockIdx.x * 64 + threadIdx.x;    // (slice, block, group)
    if (job >= (long long)nslice * nblk * 4) return;
    const int grp = (int)(job & 3);
    const long long sb = job >> 2;
    const int L = len[sb];
    if (L == 0) return;
    const int blk = (int)(sb % nblk);
    const int nrows = min(BS, C - blk * BS);
    const unsigned rowb = (unsigned)KP * 4u, QPR = (unsigned)KP / 4u, RPL = 16u / QPR;
    uint2* e0 = ent + off[sb];
    uint2* t0 = tmp + off[sb];
    const int tx = threadIdx.x;
#define c_(i, r) cnt[i][r][tx]
#define st_(i, r) start[i][r][tx]
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) c_(i, r) = 0;
    for (int i = 0; i < 16; ++i) {
        const uint2* e = e0 + SP_LANE_GROUPS[grp][i];
        for (int t = 0; t < L; ++t) { const uint2 v = e[(size_t)t * 64]; if (v.y) c_(i, (v.x / rowb) & 15u)++; }
    }
    // bucket the lists by residue into tmp (same position-major layout); cnt becomes the fill count again afterwards
    for (int i = 0; i < 16; ++i) {
        unsigned short a = 0;
        for (int r = 0; r < 16; ++r) { st_(i, r) = a; a += c_(i, r); c_(i, r) = 0; }
        const uint2* e = e0 + SP_LANE_GROUPS[grp][i];
        uint2* d = t0 + SP_LANE_GROUPS[grp][i];
        for (int t = 0; t < L; ++t) {
            const uint2 v = e[(size_t)t * 64];
            if (!v.y) continue;
            const unsigned r = (v.x / rowb) & 15u;
            d[(size_t)(st_(i, r) + c_(i, r)) * 64] = v;
            c_(i, r)++;
        }
    }
    // the steps
    for (int t = 0; t < L; ++t) {
        unsigned taken = 0, idle = 0;
        for (int i = 0; i < 16; ++i) {
            int best = -1, bestc = 0, any = -1, anyc = 0;
            for (int r = 0; r < 16; ++r) {
                const int v = c_(i, r);
                if (v > anyc) { anyc = v; any = r; }
                if (v > bestc && !((taken >> r) & 1u)) { bestc = v; best = r; }
            }
            if (any < 0) { idle |= 1u << i; continue; }
            if (best < 0) best = any;                                   // every residue it holds is taken: a conflict
            c_(i, best)--;
            e0[(size_t)t * 64 + SP_LANE_GROUPS[grp][i]] = t0[(size_t)(st_(i, best) + c_(i, best)) * 64 + SP_LANE_GROUPS[grp][i]];
            taken |= 1u << best;
        }
        for (int i = 0; i < 16; ++i) {
            if (!((idle >> i) & 1u)) continue;
            unsigned r = 0;                                             // a row of a residue nobody reads in this step
            for (unsigned q = 0; q < 16; ++q) if (!((taken >> q) & 1u)) { r = q; break; }
            if ((int)r >= nrows) r = 0;
            taken |= 1u << r;
            e0[(size_t)t * 64 + SP_LANE_GROUPS[grp][i]] = uint2{r * rowb + (((r / RPL) & (QPR - 1u)) << 4), 0u};
        }
    }
#undef c_
#undef st_
}

// physical quad of logical quad q in LDS row r: rows that share a 256-byte bank line are rotated against each other, and
// so is every group of such rows, so that the 16 lanes of one ds_read_b128 group (random rows) spread over all 16 quad slots
template <int KP> __device__ __forceinline__ int sp_quad(int r, int q)
{
    constexpr int QPR = KP / 4, RPL = 16 / QPR;           // quads per row; rows per bank line
    return (q + r / RPL) & (QPR - 1);
}
// (the padding entry {0, 0.0f} names row 0, whose rotation is 0: offset 0 is its quad 0)

// ---- one half-step (MODE 0) or the divergence partials (MODE 1) over the stored entries.
// SIDE 0: own = W (cells), other = Ht (genes), denominator Hsum;  SIDE 1: own = Ht, other = W, denominator Wsum.
// grid.x enumerates (XCD, restart, tile): tile = (slice group, block); see the header for the order.  A workgroup owns
// SP_WAVES * spw consecutive slices of the (sorted) own side; wave v takes slices v, 2 SP_WAVES - 1 - v, 2 SP_WAVES + v, ...
// -- a long one with a short one, so the 16 waves of a workgroup (which holds its CU alone: the factor block fills the LDS)
// finish together.  Measured with one slice per wave on the gene side at 200 000 x 2 000 / 9 %: the workgroup waits for
// the slice of the 64 most expressed genes while the other 15 waves idle (0.72 entries per clock and CU on average).
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));

// the slices of one wave, for a restart whose rank fills NQ of the KP / 4 quads of a factor row: the padding quads are
// neither gathered nor multiplied (they hold zeros and stay zeros)
template <int KP, int MODE, int SIDE, int NQ>
__device__ __forceinline__ void sp_slices(const BSellDev& A, const MuSlotDev& sd, const unsigned char* lds_raw, int grp, int blk,
                                          int spw, int wv, int lane, float l1, float l2, int Rs, double& acc)
{
    constexpr int KH = NQ * 2;
    constexpr unsigned RM = KP * 4 - 1;                                     // byte mask of one row
    float* own = SIDE ? sd.Ht : sd.W;
    const float* osum = SIDE ? sd.Wsum : sd.Hsum;
    for (int j = 0; j < spw; ++j) {
        const int s = (grp * spw + j) * SP_WAVES + ((j & 1) ? SP_WAVES - 1 - wv : wv);
        if (s >= A.nslice) continue;
        const int row = A.perm[(size_t)s * 64 + lane];
        sp_f32x2 w[KH], num[KH];
#pragma unroll
        for (int c = 0; c < KH; ++c) { w[c] = sp_f32x2{0.f, 0.f}; num[c] = sp_f32x2{0.f, 0.f}; }
        if (row >= 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const v4f v = *reinterpret_cast<const v4f*>(own + (size_t)row * KP + q * 4);
                w[q * 2] = sp_f32x2{v.x, v.y}; w[q * 2 + 1] = sp_f32x2{v.z, v.w};
            }
        }
        // (the builder rounds every length up to a multiple of SP_UNROLL)
        const int L = __builtin_amdgcn_readfirstlane(A.len[(size_t)s * A.nblk + blk]);
        const uint2* ep = A.ent + A.off[(size_t)s * A.nblk + blk] + lane;
        constexpr int U = NQ <= 4 ? 4 : (NQ <= 5 ? 2 : 1);                  // gathers in flight per lane (registers: no spills)
        // the entry stream runs SP_PF trips ahead of the gathers (-DCNMF_SP_PF=2: two trips, an A/B build -- measured no faster)
        uint2 nxt[U], nx2[SP_PF > 1 ? U : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = L > 0 ? ep[(size_t)u * 64] : uint2{0u, 0u};
        if constexpr (SP_PF > 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) nx2[u] = U < L ? ep[(size_t)(U + u) * 64] : uint2{0u, 0u};
        }
        for (int t = 0; t < L; t += U) {
            uint2 e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = nxt[u];
            if constexpr (SP_PF > 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = nx2[u];
                if (t + 2 * U < L) {
#pragma unroll
                    for (int u = 0; u < U; ++u) nx2[u] = ep[(size_t)(t + 2 * U + u) * 64];
                }
            } else if (t + U < L) {
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = ep[(size_t)(t + U + u) * 64];
            }
            sp_f32x2 h[U][KH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned a0 = e[u].x;                                 // quad q: same row, (quad 0 + q) modulo the row
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const unsigned a = q == 0 ? a0 : ((a0 & ~RM) | ((a0 + 16u * q) & RM));
                    const v4f v = *reinterpret_cast<const v4f*>(lds_raw + a);
                    h[u][q * 2] = sp_f32x2{v.x, v.y}; h[u][q * 2 + 1] = sp_f32x2{v.z, v.w};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float x = __uint_as_float(e[u].y);
                sp_f32x2 da = w[0] * h[u][0], db = w[1] * h[u][1];         // four partial dot products, packed by two
#pragma unroll
                for (int c = 2; c < KH; c += 2) {
                    da = __builtin_elementwise_fma(w[c], h[u][c], da);
                    db = __builtin_elementwise_fma(w[c + 1], h[u][c + 1], db);
                }
                da += db;
                const float whs = fmaxf(da.x + da.y, MU_EPS);
                if (MODE == 0) {
                    const float rr = x * __builtin_amdgcn_rcpf(whs);          // (1 ulp; the matrix-pipe path carries 2^-17)
                    const sp_f32x2 r2 = sp_f32x2{rr, rr};
#pragma unroll
                    for (int c = 0; c < KH; ++c) num[c] = __builtin_elementwise_fma(r2, h[u][c], num[c]);
                } else if (x > MU_EPS) {
                    acc += (double)(x * logf(x / whs) - x);
                }
            }
        }
        if (MODE == 1 || row < 0) continue;
        if (A.nblk == 1) {
            // the whole other side was one block: finish the update here (sklearn _nmf.py:588-631 / :684-728)
            float wv_[NQ * 4];
#pragma unroll
            for (int c = 0; c < NQ * 4; ++c) {
                const float w0 = (c & 1) ? w[c / 2].y : w[c / 2].x, nm = (c & 1) ? num[c / 2].y : num[c / 2].x;
                float dn = osum[c];
                if (SIDE && dn == 0.f) dn = 1.0f;
                if (l1 > 0.f) dn += l1;
                if (l2 > 0.f) dn += l2 * w0;
                if (dn == 0.f) dn = MU_EPS;
                float v = w0 * (nm / dn);
                if (SIDE && v < F64_EPS_AS_F32) v = 0.f;
                wv_[c] = v;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                *reinterpret_cast<v4f*>(own + (size_t)row * KP + q * 4) = v4f{wv_[q * 4], wv_[q * 4 + 1], wv_[q * 4 + 2], wv_[q * 4 + 3]};
        } else {
            float* pp = sd.pnum + ((size_t)blk * Rs + row) * KP;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                *reinterpret_cast<v4f*>(pp + q * 4) = v4f{num[q * 2].x, num[q * 2].y, num[q * 2 + 1].x, num[q * 2 + 1].y};
        }
    }
}

template <int KP, int MODE, int SIDE>
__global__ __launch_bounds__(SP_WAVES * 64) void mu_sp_kernel(BSellDev A, MuBatch mb, int ngroups, int spw, float l1, float l2,
                                                              int Rs /* padded own rows: row stride of the partials */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds_raw[];
    float* lds = reinterpret_cast<float*>(sp_lds_raw);
    constexpr int QPR = KP / 4;
    // decode: n -> (xcd, m); m -> (restart, tile / 8); tile = (m / n_restarts) * 8 + xcd
    const int n = blockIdx.x, xcd = n & 7, m = n >> 3;
    const int slot = m % mb.n, tile = (m / mb.n) * 8 + xcd;
    const int ntiles = ngroups * A.nblk;
    if (tile >= ntiles) return;
    const int grp = tile % ngroups, blk = tile / ngroups;
    const MuSlotDev& sd = mb.s[slot];
    const float* other = SIDE ? sd.W : sd.Ht;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = __builtin_amdgcn_readfirstlane(min(QPR, max(1, (sd.k + 3) >> 2)));      // live quads of this restart
    // stage the other side's block
    const int o0 = blk * A.BS, nrows = min(A.BS, A.C - o0);
    for (int e = tid; e < nrows * QPR; e += SP_WAVES * 64) {          // (entries only name rows below nrows)
        const int r = e / QPR, q = e % QPR;
        const v4f v = *reinterpret_cast<const v4f*>(other + (size_t)(o0 + r) * KP + q * 4);
        *reinterpret_cast<v4f*>(lds + (r * QPR + sp_quad<KP>(r, q)) * 4) = v;
    }
    __syncthreads();
    double acc = 0.0;
#define SP_CASE(NQ_) case NQ_: if constexpr (NQ_ <= QPR) sp_slices<KP, MODE, SIDE, NQ_>(A, sd, sp_lds_raw, grp, blk, spw, wv, lane, l1, l2, Rs, acc); break;
    switch (nq) { SP_CASE(1) SP_CASE(2) SP_CASE(3) SP_CASE(4) SP_CASE(5) SP_CASE(6) SP_CASE(7) SP_CASE(8) default: break; }
#undef SP_CASE
    if (MODE == 1) {
        __syncthreads();                                    // (every wave is past the factor block: LDS is free)
        double* red = reinterpret_cast<double*>(sp_lds_raw);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) red[wv] = acc;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int q = 0; q < SP_WAVES; ++q) t += red[q];
            sd.divpart[tile] = t;
        }
    }
}

// ---- finish of a half-step whose other side took several blocks: numerators added in block order, then the update
template <int KP>
__global__ __launch_bounds__(256) void mu_sp_finish_kernel(MuBatch mb, int side, int R, int Rs, int nblk, float l1, float l2)
{
    const MuSlotDev& sd = mb.s[blockIdx.y];
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)R * KP) return;
    const int c = (int)(e % KP);
    if (c >= ((sd.k + 3) & ~3)) return;                        // padding quads: never written, stay zero
    float* own = side ? sd.Ht : sd.W;
    float n = 0.f;
    for (int b = 0; b < nblk; ++b) n += sd.pnum[(size_t)b * Rs * KP + e];
    float dn = (side ? sd.Wsum : sd.Hsum)[c];
    if (side && dn == 0.f) dn = 1.0f;                          // sklearn _nmf.py:684-686
    const float v0 = own[e];
    if (l1 > 0.f) dn += l1;
    if (l2 > 0.f) dn += l2 * v0;
    if (dn == 0.f) dn = MU_EPS;
    float v = v0 * (n / dn);
    if (side && v < F64_EPS_AS_F32) v = 0.f;                   // sklearn _nmf.py:868-869
    own[e] = v;
}

}  // namespace cnmf
