/* Diagnostic entry points of libcnmf_hip.so -- TEST HOOKS, not part of the drop-in boundary (include/cnmf_hip.h).
 * They exist only in a library compiled with -DCNMF_DEBUG_ABI (the in-tree build of cnmf_amd/_lib.py defines it because
 * tests/ call them; `CNMF_PRODUCT_BUILD=1 python -c "import __graft_entry__ as g; g.build()"` leaves them out). */
#ifndef CNMF_HIP_DEBUG_H
#define CNMF_HIP_DEBUG_H
#include "cnmf_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* C[KC][J] = A[KC][K] . B  through the engine's MFMA GEMM; mode 0: B is [J][K] (pass A),
 * mode 1: B is [K][J] (pass B, split-K partials summed in split order).                  */
int cnmf_debug_gemm(cnmf_ctx* ctx, int mode, int variant, const float* A, const float* B,
                    float* C, int KC, int K, int J, int nsplit, double* ms_out, int reps);
/* C[KC][J] = A[KC][K] . B[J][K]^T through the split-operand (3 x bf16 planes, f32-accurate) MFMA
 * path; KC % 256 == 0, K % 16 == 0.                                                        */
int cnmf_debug_gemm3(cnmf_ctx* ctx, const float* A, const float* B, float* C, int KC, int K, int J,
                     int nsplit, double* ms_out, int reps);
/* the same for count-structured data: Bn [J][K] holds non-negative integers <= 256 (ONE bf16 plane),
 * A arbitrary float32 (three planes); 3 exact bf16 MFMAs per product on 256 x 256 tiles.           */
int cnmf_debug_gemm3c(cnmf_ctx* ctx, const float* A, const float* Bn, float* C, int KC, int K, int J,
                      int nsplit, double* ms_out, int reps);
/* the same on the f16 matrix pipe (the default for count-structured data): Bn <= 2048 in ONE f16 plane (a
 * flagged second plane above that), A >= 0 as TWO f16 planes with a per-row exponent; 2 MFMAs per product.
 * KC % 256 == 0, K % 64 == 0; nsub = 16-k sub-blocks per barrier pair (1 | 2); nsub | 128: scale every row of A by
 * the BOUND the W half-step reports (sqrt of the sum of squares over 1024-entry blocks x 1.0001) instead of the exact
 * row maximum -- the production pass-B scaling, for the accuracy tests.                                   */
int cnmf_debug_gemm2h(cnmf_ctx* ctx, const float* A, const float* Bn, float* C, int KC, int K, int J,
                      int nsplit, int nsub, double* ms_out, int reps);
/* calibration streams of known byte counts for the PMC counters (tools/pmc_calibrate.py): width 1 = a copy with 4 B per
 * lane, 4 = 16 B per lane, 0 = a read-only LDS-DMA stream (global_load_lds_dwordx4); n_floats floats, `reps` launches.  */
int cnmf_debug_stream(cnmf_ctx* ctx, int width, long long n_floats, int reps);
/* numpy RandomState(seed).standard_normal(n) reproduced on the device. */
int cnmf_debug_standard_normal(cnmf_ctx* ctx, uint32_t seed, int64_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif /* CNMF_HIP_DEBUG_H */
