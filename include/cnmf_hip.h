/* libcnmf_hip.so -- C ABI of the MI355X-native consensus-NMF engine.
 *
 * The reference (dylkot/cNMF v1.7.1) has no FFI: its hot path is Python method
 * dispatch on `class cNMF` that bottoms out in scikit-learn.  The entry points
 * below are what a ctypes binding behind those methods calls; each one names the
 * reference interface it replaces (paths are /root/reference/src/cnmf/cnmf.py
 * unless prefixed sklearn:).  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - plain pointers and sizes only; the CALLER owns every host buffer (numpy),
 *     the library owns all device memory behind the opaque context;
 *   - every call returns 0 on success or a negative CNMF_E* code; the message is
 *     available from cnmf_last_error() (the Python wrapper maps codes to the same
 *     exception types the reference path raises);
 *   - one context per GPU, one host thread per context; no global mutable state
 *     (multiprocessing fan-out with one process per GPU = one `worker_i` works);
 *   - matrices are C-order float32.  Factor packing for a batch of restarts
 *     r = 0..n-1 with ranks k[r]:  H blocks [k[r]][G] concatenated in r order,
 *     W blocks [N][k[r]] concatenated in r order (sklearn's own layouts).
 */
#ifndef CNMF_HIP_H
#define CNMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNMF_OK            0
#define CNMF_EINVAL       -1   /* bad argument (ValueError on the Python side)   */
#define CNMF_EHIP         -2   /* HIP runtime failure (RuntimeError)             */
#define CNMF_ENOMEM       -3   /* device allocation failed (MemoryError)         */
#define CNMF_ESTATE       -4   /* call order violated, e.g. no matrix set        */
#define CNMF_EUNSUPPORTED -5   /* e.g. rank > CNMF_KMAX (NotImplementedError)    */
#define CNMF_ECOMM        -6   /* RCCL failure                                   */

#define CNMF_KMAX 128          /* largest rank of the coordinate-descent / NNLS / consensus entry points
                                  (<= 64: register-resident sweep; 65..128: sweep_big_kernel)              */

#define CNMF_MU_KMAX 64        /* largest rank of the multiplicative-update solver (cnmf_nmf_mu_batch)      */

typedef struct cnmf_ctx cnmf_ctx;

/* Solver parameters = the yaml kwargs the reference persists (cnmf.py:618-631)
 * after sklearn's scaling of the regularisers (sklearn:decomposition/_nmf.py:1254-1265):
 *   l1_reg_W = G*alpha_W*l1_ratio   l2_reg_W = G*alpha_W*(1-l1_ratio)
 *   l1_reg_H = N*alpha_H*l1_ratio   l2_reg_H = N*alpha_H*(1-l1_ratio)        */
typedef struct cnmf_cd_params {
    double tol;          /* 1e-4 in the reference (cnmf.py:624)                  */
    int    max_iter;     /* 1000 (cnmf.py:567,625)                               */
    int    kc_max;       /* max packed component columns in flight: 32..256 in steps of 32, 512 / 768 / 1024 (wide batches:
                            several 256-column component groups per GEMM pass, matrix-pipe paths only); 0 = auto
                            (up to 256; as wide as the job up to 1024 on matrices of >= 2^24 padded entries)        */
    double l1_reg_W, l2_reg_W, l1_reg_H, l2_reg_H;
    int    lag;          /* host polls convergence `lag` iterations behind the GPU; 0 = default (2) */
    int    profile;      /* n > 0: bracket the two GEMM passes of every n-th iteration with HIP events
                            (fills passA_ms/passB_ms and the launch counts of the sampled iterations) */
} cnmf_cd_params;

/* Per-call statistics (optional, may be NULL). */
typedef struct cnmf_batch_stats {
    int64_t outer_iterations;      /* batch iterations enqueued (each = pass A + pass B)        */
    int64_t restart_iterations;    /* sum over restarts of their n_iter                         */
    int64_t column_iterations;     /* sum over batch iterations of KC (incl. idle columns)      */
    int64_t restart_column_iterations; /* sum over restarts of n_iter*k (algorithmic columns)  */
    double  gpu_ms;                /* device time of the whole call (hipEvent)                  */
    double  passA_ms, passB_ms;    /* summed hipEvent time of the two MFMA GEMM passes          */
    int64_t passA_launches, passB_launches;
    int32_t kc;                    /* packed column count used                                  */
    int32_t nsplit;                /* split-K factor of pass B                                  */
    int32_t gemm_mode;             /* GEMM path of the 256-column phase of this call: 0 = exact-f32 MFMA,
                                      1/2 = split-operand bf16 MFMA (3 x 3 planes, any X), 3 = count-structured X
                                      as one integer bf16 plane x 3 factor planes, 4 (the default when the count
                                      structure is detected) = integer f16 plane x 2 f16 factor planes         */
    int32_t reserved_;
    /* the tail of the call: from the moment the queue of pending restarts ran dry (nothing left to refill freed
     * columns with) to the end -- what strong scaling over more GPUs pays for (fewer restarts per rank)          */
    int64_t tail_iterations;       /* batch iterations enqueued after the queue ran dry                         */
    int64_t tail_live_columns;     /* sum over those iterations of the columns still iterating (host view)      */
    double  tail_ms;               /* device time of that phase (hipEvent)                                      */
} cnmf_batch_stats;

/* ---- lifecycle ------------------------------------------------------------------- */
int         cnmf_device_count(void);
cnmf_ctx*   cnmf_create(int device);           /* NULL on failure; see cnmf_last_error(NULL) */
void        cnmf_destroy(cnmf_ctx* ctx);
const char* cnmf_last_error(const cnmf_ctx* ctx);
/* The CNMF_* environment variables (INTEGRATION.md "Runtime switches") that steer per-call host decisions are read ONCE,
 * when the context is created, into the context; cnmf_reload_env re-reads them (A/B tools, tests).  A knob therefore
 * cannot change between two calls on one context unless the caller asks for it.  In the snapshot: the GEMM operand
 * scheme (CNMF_GEMM3), the batch width (CNMF_KC, CNMF_KC_LIMIT, CNMF_NO_WIDE, CNMF_WIDE_SMALL), CNMF_LAG, the consensus and
 * multiplicative-update switches.  PROCESS-LIFETIME (read once per process, NOT refreshed by cnmf_reload_env): the A/B
 * knobs of the kernel launch planning and instruction streams -- CNMF_NO_STREAMK, CNMF_S_MTW2, CNMF_G2_*, CNMF_WG_SLOTS,
 * CNMF_FUSE_W, CNMF_G2G, CNMF_SPIN_QUERY, CNMF_NO_COUNTS -- set them before the library is loaded.                    */
int cnmf_reload_env(cnmf_ctx* ctx);
const char* cnmf_version(void);

/* ---- data matrix ------------------------------------------------------------------
 * Replaces the per-worker `norm_counts = sc.read(...)` + `norm_counts.X` hand-off of
 * cNMF.factorize (cnmf.py:726,741) and cNMF.consensus (cnmf.py:873,919): X (cells x
 * high-variance genes) is uploaded ONCE per context and stays resident in HBM.       */
int cnmf_set_matrix(cnmf_ctx* ctx, const float* X, int64_t n_cells, int64_t n_genes);
/* Count-structure detection (default on): the engine recognises X = (integers <= 65 535) x (one constant per gene)
 * -- what `norm_counts.X /= std` produces (cnmf.py:540-548) -- with a tolerance of 1e-3 count units and then
 * multiplies the exact integer planes; enabled = 0 keeps every matrix on the general float32-operand path.      */
int cnmf_set_count_detection(cnmf_ctx* ctx, int enabled);
/* CSR input (scipy.sparse.csr_matrix of float32, int32 indices/indptr) -- the reference hands `norm_counts.X` / `tpm.X`
 * to scikit-learn as stored (cnmf.py:726, 873, 950).  The arrays STAY on the device and serve the paths that walk the
 * stored entries (cnmf_mu_refit_f64, the non-zero images of cnmf_nmf_mu_batch) -- as uploaded when every row lists
 * strictly increasing columns without stored zeros (scipy's canonical format after eliminate_zeros(); the Python
 * wrapper compacts stored zeros before the call).  The dense float32 image is formed on the device ONLY when a path
 * that multiplies the dense matrix asks for it (coordinate descent, NNLS, Itakura-Saito restarts; duplicates summed like
 * .toarray()); arrays that are not canonical form it at once and the compressed rows are rebuilt from it on first use.
 * A column index outside [0, n_genes): CNMF_EINVAL.                                                              */
int cnmf_set_matrix_csr(cnmf_ctx* ctx, const int32_t* indptr, const int32_t* indices,
                        const float* data, int64_t n_cells, int64_t n_genes);
int cnmf_get_shape(const cnmf_ctx* ctx, int64_t* n_cells, int64_t* n_genes);
/* Which images of the matrix are resident right now (round 5): bit 0 the dense float32 image (a CSR upload forms it only when
 * a path that multiplies the dense matrix asks for it: a Kullback-Leibler run on a sparse matrix never does), bit 1 the
 * compressed rows of X, bit 2 those of X^T, bit 3 the dense X^T copy of the dense multiplicative-update kernels,
 * bits 4 / 5 the non-zero images at padded rank 16 / 32, bit 6 the integer count planes of the coordinate-descent path. */
int cnmf_matrix_images(const cnmf_ctx* ctx, int32_t* flags);
/* the resident matrix back on the host ([n_cells][n_genes] float32) */
int cnmf_get_matrix(cnmf_ctx* ctx, float* out);

/* ---- gene-wise scaling of the resident matrix ------------------------------------------
 * The dense branch of the reference's get_norm_counts (cnmf.py:540-554): upload the raw counts of
 * the high-variance genes with cnmf_set_matrix, then
 *     cnmf_col_moments   -> mean[g] and ssd[g] = sum_i (x_ig - mean_g)^2   (float64, two passes)
 *                           std(ddof=1) = sqrt(ssd / (n_cells - 1)) is formed by the caller
 *     cnmf_scale_columns -> x_ig = float32(float64(x_ig) / divisor[g])     (divisors must be > 0)
 *     cnmf_row_sums      -> float64 sum over the genes of every cell: the "zero cells" check
 *                           (cnmf.py:550-554) and X.mean() for the random init are derived from it */
int cnmf_col_moments(cnmf_ctx* ctx, double* mean_out /* [G] */, double* ssd_out /* [G] */);
int cnmf_scale_columns(cnmf_ctx* ctx, const double* divisor /* [G] */);
int cnmf_row_sums(cnmf_ctx* ctx, double* out /* [N] */);

/* ---- the restart hot loop ---------------------------------------------------------
 * Replaces the loop body of cNMF.factorize (cnmf.py:735-741): for every restart r,
 *   (usages, spectra, n_iter) = non_negative_factorization(X, n_components=k[r],
 *        init='random'|custom, solver='cd', beta_loss='frobenius', tol, max_iter, ...)
 * i.e. cNMF._nmf (cnmf.py:661-674) -> sklearn:decomposition/_nmf.py:905-1131,
 * _fit_coordinate_descent :406-523, _update_cdnmf_fast sklearn:_cdnmf_fast.pyx:8-38.
 * All restarts share each pass over X.
 *
 *   init_mode 0 (custom): W0 / H0 hold the packed initial factors (what sklearn's
 *       _initialize_nmf :302-314 returns for the restart's seed).
 *   init_mode 1 (random): the library reproduces sklearn's init='random' on the device
 *       from `seeds` (numpy RandomState(seed): MT19937 + legacy polar gauss, H drawn
 *       before W) scaled by avg[r] = sqrt(X.mean()/k[r]); W0/H0 are ignored.
 *
 *   H_out  [sum k][G]   required.   W_out packed [N][k[r]] blocks, may be NULL
 *   (cNMF.factorize drops the usages, cnmf.py:741-745).
 *   n_iter_out[r] = sklearn's n_iter;  viol_out[r] = final violation/violation_init.  */
int cnmf_nmf_cd_batch(cnmf_ctx* ctx, int n_restarts, const int32_t* k,
                      int init_mode, const uint32_t* seeds, const double* avg,
                      const float* W0, const float* H0,
                      const cnmf_cd_params* params,
                      float* H_out, float* W_out,
                      int32_t* n_iter_out, double* viol_out,
                      cnmf_batch_stats* stats);

/* Same, but the spectra stay on the device (for the RCCL gather / on-device consensus):
 * results are appended to the context's spectra store in restart order.               */
int cnmf_nmf_cd_batch_resident(cnmf_ctx* ctx, int n_restarts, const int32_t* k,
                               int init_mode, const uint32_t* seeds, const double* avg,
                               const float* W0, const float* H0,
                               const cnmf_cd_params* params,
                               int32_t* n_iter_out, double* viol_out,
                               cnmf_batch_stats* stats);
/* Queue hints (round 4).  A batch call learns, per rank, the mean number of outer iterations its restarts took
 * (cnmf_get_iteration_means: out[CNMF_KMAX + 1], 0 = rank not seen on this matrix).  Handing such numbers back with
 * cnmf_set_iteration_hints makes the FOLLOWING calls on this matrix start their queue longest-expected-first instead of
 * learning the order again (n = 0 clears the hints; so does a new matrix).  Never implicit: the queue order decides which
 * packed columns a restart occupies, and its float32 result moves in the last bits (1e-6 relative) with its placement --
 * without hints a call's result depends on its own arguments alone (bit for bit; tests/test_gpu_determinism.py). */
int cnmf_get_iteration_means(cnmf_ctx* ctx, double* out);
int cnmf_set_iteration_hints(cnmf_ctx* ctx, int n, const int32_t* k, const double* mean_iterations);

/* ---- multiplicative-update solver --------------------------------------------------------
 * The reference keeps solver='mu' whenever beta_loss != 'frobenius' (cnmf.py:618-631):
 *   non_negative_factorization(X, solver='mu', beta_loss='kullback-leibler'|'itakura-saito', ...)
 *   = sklearn:decomposition/_nmf.py:731-893 (_fit_multiplicative_update), :526-728 (updates),
 *     :84-194 (_beta_divergence, evaluated every 10 iterations for the stopping rule).
 * beta: 1 = kullback-leibler, 0 = itakura-saito.  Same packing / init modes as cnmf_nmf_cd_batch.
 * update_H = 0 is the refit (cnmf.py:776-802 with solver 'mu'): H0 holds the fixed spectra and W
 * starts from avg[r] everywhere (sklearn:_nmf.py:1229-1231); H_out is then ignored.
 * err_out[r] = sqrt(2 * beta-divergence) of the FINAL factors (sklearn's reconstruction_err_).
 * Restarts run batched on the matrix pipe (padded rank 16 / 32 / 64, both losses), up to 32 per round of
 * launches sharing each pass over X (kernels_mu_mfma.hip.h); a restart's result does not depend on the
 * batch it ran in.  Ranks above CNMF_MU_KMAX (64): CNMF_EUNSUPPORTED.
 * The first call builds a resident transposed copy of X (freed with the matrix).
 * Kullback-Leibler on the NON-ZEROS (round 4, kernels_mu_sparse.hip.h) -- scikit-learn's own route for scipy.sparse input
 * (sklearn:_nmf.py:192 `_special_sparse_dot`): taken by itself when beta = 1, every rank of the call is <= 32 and at most a
 * quarter of X is non-zero (counted once per matrix); same mathematics (the quotient vanishes where X does), exact float32
 * products, another summation order than the dense kernels (results agree to float32 round-off, not bit for bit).
 * Extra device memory: per padded rank in use (16, 32) two blocked sliced-ELL images, 8 B x non-zeros x 1.0-1.2 each
 * (round 5: built from the compressed rows of the matrix, csr_host.hip.h -- no dense transposed copy on this path), and per
 * restart in flight the partial numerators [blocks of the other side][own rows][padded rank] float32.  When those
 * allocations do not fit the call falls back to the dense kernels.  The environment variable CNMF_MU_SPARSE (read when the
 * context is created; 0 = never, 1 = always where the rank allows) overrides the density rule.                       */
int cnmf_nmf_mu_batch(cnmf_ctx* ctx, int n_restarts, const int32_t* k, int init_mode,
                      const uint32_t* seeds, const double* avg, const float* W0, const float* H0,
                      int beta, int update_H, const cnmf_cd_params* params,
                      float* H_out, float* W_out, int32_t* n_iter_out, double* err_out);

/* Multiplicative-update refit in FLOAT64 on the stored entries (round 5, mu_refit_host.hip.h): the three
 *   non_negative_factorization(X, H=..., update_H=False, solver='mu', beta_loss=...)
 * calls of a consensus run with a beta loss (cnmf.py:776-820 via :920, :952, :972) on float64 matrices --
 * scikit-learn's `_fit_multiplicative_update` with H fixed (sklearn:_nmf.py:731-893, :526-631, :84-194).
 *   beta 1: Kullback-Leibler (only the stored entries count, like scikit-learn on scipy.sparse input);
 *   beta 0: Itakura-Saito (round 6).  scikit-learn refuses beta_loss <= 0 on a matrix that contains a zero
 *           (sklearn:_nmf.py:1679-1684), so must the caller's matrix be strictly positive: every entry is then a stored entry
 *           and the same walk is the dense update.  A matrix with a zero: CNMF_EINVAL with scikit-learn's message.
 *   side 0: rows = cells   (refit_usage):   H [k][n_genes],  W_out [n_cells][k]
 *   side 1: rows = GENES   (refit_spectra = the problem on X^T, cnmf.py:820): H = usages^T [k][n_cells], W_out [n_genes][k];
 *           walks the compressed rows of X^T built on the device -- no todense(), no transposed upload.
 * coldiv (NULL ok, [columns of the walked matrix]): x'_ij = x_ij / coldiv[j], coldiv[j] == 0 drops column j -- the final
 *   usage refit on tpm[:, hvgs] / std (cnmf.py:963-972) against the resident full TPM matrix.
 * w_init: W starts from this value everywhere (sklearn:_nmf.py:1229-1231: sqrt(X.mean() / k) of the matrix meant).
 * params: tol, max_iter, l1_reg_W, l2_reg_W.  n_iter_out as scikit-learn counts (a multiple of 10 or max_iter),
 * err_out = sqrt(2 x divergence) of the final factors.  Ranks above CNMF_MU_KMAX: CNMF_EUNSUPPORTED.             */
int cnmf_mu_refit_f64(cnmf_ctx* ctx, int side, int beta, int k, const double* H, const double* coldiv, double w_init,
                      const cnmf_cd_params* params, double* W_out, int32_t* n_iter_out, double* err_out);

/* ---- NNLS refit ---------------------------------------------------------------------
 * Replaces cNMF.refit_usage (cnmf.py:776-802): non_negative_factorization(X, H=spectra,
 * update_H=False, n_components=k, solver='cd', ...) with W0 = 0 (sklearn:_nmf.py:1232-1233).
 * X.Ht is computed once (sklearn recomputes it every outer iteration although H is fixed). */
int cnmf_nnls(cnmf_ctx* ctx, int k, const float* H /*[k][G]*/, const cnmf_cd_params* params,
              float* W_out /*[N][k]*/, int32_t* n_iter_out, double* viol_out);

/* ---- consensus core ---------------------------------------------------------------------
 * Replaces the numerical core of cNMF.consensus (cnmf.py:871-916 + the silhouette of the stats
 * branch, :923), float64 on the device:
 *   l2 = spectra / |spectra|_2 (:882) -> euclidean_distances (:891; sklearn:metrics/pairwise.py:
 *   391-438) -> local density = sum of the n_neighbors+1 smallest per row / n_neighbors (:893-898)
 *   -> keep density < threshold (:903) -> KMeans(k, n_init, random_state=1) on the kept rows
 *   (:908-909; sklearn:cluster/_kmeans.py:1427-1555) -> per-cluster per-gene median (:913) ->
 *   rows / row sum (:916).
 * `uniforms` = the draws KMeans takes from numpy RandomState(random_state):
 *   [n_init][1 + (k-1)*(2+int(log k))] doubles (first centre, then the local trials).
 * Outputs: density_out[R] (NULL ok; zeros when skip_density), keep_out[R] 0/1 (NULL ok),
 *   labels_out[R] (0..k-1, -1 for filtered rows), median_out[k][G], dist_out[R][R] (NULL ok),
 *   stats_out[4] = {rows kept, best inertia, silhouette (0 unless want_silhouette), n_iter of best run}.
 * Returns CNMF_ESTATE with the reference's message when the filter removes every row (:905-906).
 * Device memory: the float64 distance matrix is ALWAYS formed (k-means++ reads its squared distances from it, also with
 * skip_density): 8 R_pad^2 bytes (R_pad = R rounded up to 64: 0.2 GB at R = 5 000, 3.2 GB at R = 20 000) + ~24 R G
 * bytes of spectra copies; a call that leaves more than 2 GB in the context's workspace releases it on return.
 * k-means++ takes the squared distance of two rows as (Euclidean distance)^2 of the UNcentred rows where scikit-learn
 * subtracts the column means first: the same number up to the last bits, so the labels are scikit-learn's except where a
 * draw falls within rounding of a tie. */
typedef struct cnmf_consensus_params {
    int    k;
    int    n_neighbors;        /* int(local_neighborhood_size * R / k), cnmf.py:879 */
    double density_threshold;
    int    skip_density;       /* 1 = skip_density_and_return_after_stats (k_selection_plot)  */
    int    want_silhouette;
    int    n_init;             /* 10 */
    int    max_iter;           /* 300 (sklearn default) */
    double tol;                /* 1e-4 (sklearn default) */
} cnmf_consensus_params;

int cnmf_consensus(cnmf_ctx* ctx, const double* spectra, int R, int G,
                   const cnmf_consensus_params* params, const double* uniforms,
                   double* density_out, int32_t* keep_out, int32_t* labels_out,
                   double* median_out, double* dist_out, double* stats_out);

/* The other two scikit-learn calls of the reference's consensus body, on their own (round 4; INTEGRATION.md Option B
 * binds them): euclidean_distances(rows) (cnmf.py:891, :988) -> dist_out [R][R] (NULL ok), and
 * silhouette_score(rows, labels, metric='euclidean') (cnmf.py:923) -> *silhouette_out (NULL ok; labels [R] in 0..k-1,
 * 2 <= k <= n_samples - 1 as scikit-learn requires).  Float64; the rows are taken AS GIVEN (no normalisation).   */
int cnmf_pairwise_distances(cnmf_ctx* ctx, const double* rows, int R, int G, const int32_t* labels, int k,
                            double* dist_out, double* silhouette_out);

/* sum((X - W.H)^2) over the resident matrix: the prediction error of cnmf.py:926-930
 * (W [N][k], H [k][G], float64).  A matrix that lives as compressed rows only is NOT densified (the reference calls
 * todense() there): sum over the stored entries of (x - wh)^2 - (wh)^2, plus tr(W^T W . H H^T). */
int cnmf_prediction_error(cnmf_ctx* ctx, int k, const double* W, const double* H, double* err_out);

/* ---- products with the resident matrix ----------------------------------------------------
 * out = X . Q (trans = 0: Q [G][ncols] -> out [N][ncols]) or X^T . Q (trans = 1: Q [N][ncols] ->
 * out [G][ncols]), ncols <= 256, through the engine's MFMA GEMM.  This is the only O(N.G) work in
 * sklearn's NNDSVD initialisation (`--init nndsvd`, cnmf.py:1252 -> sklearn:_nmf.py:317 ->
 * sklearn:utils/extmath.py:287-357 `_randomized_range_finder`: 2*n_iter+2 such products); the small
 * LU / QR / SVD factorizations stay on the host (cnmf_amd/engine.py::Engine.nndsvd_init).        */
int cnmf_x_matmul(cnmf_ctx* ctx, int trans, const float* Q, int ncols, float* out);

/* ---- multi-GPU exchange (RCCL over xGMI) ---------------------------------------------------
 * The restarts shard with no collective: ledger row idx runs on rank idx % world, the
 * reference's worker_filter (cnmf.py:52-53).  The reference's "gather" is the filesystem:
 * combine_nmf re-reads one npz per restart (cnmf.py:755-770).  Here it is ONE ncclAllGather of
 * the packed float32 spectra.  One process per GPU, one context per process.  RCCL is bound
 * lazily (dlopen) on the first cnmf_comm_* call; single-GPU use never touches it, and without
 * a communicator the gathers below degenerate to copies (world = 1).
 *
 *   rank 0:     cnmf_comm_unique_id(id)  -> ship the 128 bytes to the other ranks out of band
 *               (a file, an environment variable, MPI, a torch.distributed store ...)
 *   every rank: cnmf_comm_init(ctx, id, rank, world)             (collective: ncclCommInitRank)
 *               cnmf_allgather_bytes(ctx, mine, n, all)          headers / row counts, host buffers
 *               cnmf_allgather_spectra(ctx, local, rows_local, rows_max, G, out)
 *                   local [rows_local][G] host floats, or NULL = the context's resident store
 *                   (cnmf_nmf_cd_batch_resident) -> then the spectra never visit the host before
 *                   the exchange;  out [world][rows_max][G] host floats, shards zero-padded to
 *                   rows_max (= max over ranks, from the header gather).
 *               cnmf_comm_finalize(ctx)                          (also done by cnmf_destroy)      */
#define CNMF_COMM_ID_BYTES 128
int cnmf_comm_unique_id(unsigned char* id_out /* [CNMF_COMM_ID_BYTES] */);
int cnmf_comm_init(cnmf_ctx* ctx, const unsigned char* id /* [CNMF_COMM_ID_BYTES] */, int rank, int world);
int cnmf_comm_finalize(cnmf_ctx* ctx);
int cnmf_comm_rank(const cnmf_ctx* ctx);
int cnmf_comm_world(const cnmf_ctx* ctx);
int cnmf_allgather_bytes(cnmf_ctx* ctx, const void* send, int64_t nbytes, void* recv /* [world][nbytes] */);
int cnmf_allgather_spectra(cnmf_ctx* ctx, const float* local, int64_t rows_local, int64_t rows_max,
                           int64_t n_genes, float* out);
/* the resident spectra store filled by cnmf_nmf_cd_batch_resident (rows in restart order) */
int64_t cnmf_spectra_rows(const cnmf_ctx* ctx);
int cnmf_spectra_reset(cnmf_ctx* ctx);
int cnmf_spectra_fetch(cnmf_ctx* ctx, float* out /* [rows][G] */);
/* rows [row0, row0 + n_rows) of the store: what ONE batch call appended (the store may hold earlier calls' rows too) */
int cnmf_spectra_fetch_rows(cnmf_ctx* ctx, int64_t row0, int64_t n_rows, float* out);
/* gene count of the rows in the store (round 4: the store outlives cnmf_set_matrix -- consensus() alternates between the
 * normalised counts and the TPM matrix while the spectra keep serving k selection and further consensus calls) */
int64_t cnmf_spectra_genes(const cnmf_ctx* ctx);
/* append rows [n_rows][n_genes] (float32) to the store: merged spectra read from the reference's files or gathered from
 * other GPUs, uploaded once for any number of cnmf_consensus_store calls (other k, other density thresholds)         */
int cnmf_spectra_append(cnmf_ctx* ctx, const float* rows, int64_t n_rows, int64_t n_genes);
/* cnmf_consensus / cnmf_kselect_stats with the merged spectra taken from the RESIDENT store instead of a host buffer:
 * store_rows[r] = row of the store holding merged row r (float32 there, widened to float64 on the device) -- what
 * combine_nmf (cnmf.py:748-773) would have concatenated, without the trip through the host.                   */
int cnmf_consensus_store(cnmf_ctx* ctx, const int64_t* store_rows, int R, int G,
                         const cnmf_consensus_params* params, const double* uniforms,
                         double* density_out, int32_t* keep_out, int32_t* labels_out,
                         double* median_out, double* dist_out, double* stats_out);
int cnmf_kselect_stats_store(cnmf_ctx* ctx, int n, const int32_t* ks, const int32_t* R, const int64_t* store_rows,
                             const cnmf_consensus_params* cprm, const double* uniforms, const cnmf_cd_params* prm,
                             double* silhouette_out, double* pred_err_out, double* median_out, int32_t* nnls_iter_out);

/* ---- consensus tail + k selection on the device ------------------------------------------------------------ */
/* out[k][G] (float64) = W^T . X, or W^T . zscore(X) with z = (x - mean[g]) * inv_std[g]: the X^T Y accumulation of
 * efficient_ols_all_cols(normalize_y=True) (cnmf.py:55-125, called at :958) over the RESIDENT matrix (the TPM matrix,
 * dense or uploaded as CSR) in float64 like the reference.  W is [N][k] float64 (k <= CNMF_KMAX).  A matrix that lives as
 * compressed rows only (a CSR upload whose dense image nobody asked for) is walked on its stored entries -- the zeros of
 * a column folded into the constant term; cnmf_col_moments likewise.                                          */
int cnmf_xt_matmul_f64(cnmf_ctx* ctx, int k, const double* W, int zscore, const double* mean,
                       const double* inv_std, double* out);
/* cNMF.refit_spectra (cnmf.py:805-820 = refit_usage(X.T, usage.T).T, sklearn:_nmf.py:1210-1233): NNLS for the
 * spectra H [k][G] with the usages W [N][k] fixed, H from zero, sklearn's CD stopping rule -- on the resident
 * matrix, without uploading its transpose (the constant product is W^T.X).  The solved factor takes sklearn's
 * "W" role: prm->l1_reg_W / l2_reg_W apply to it.  Float64 throughout (round 4: product, Gram, sweeps and result):
 * the reference pins gene_spectra_tpm -- values up to 1e5 TPM units -- to sum(diff^2) < 1e-4
 * (/root/reference/tests/test_reproducibility.py:96-115).                                                   */
int cnmf_nnls_spectra(cnmf_ctx* ctx, int k, const double* W, const cnmf_cd_params* prm, double* H_out,
                      int32_t* n_iter_out, double* viol_out);
/* cNMF.refit_usage (cnmf.py:776-802) in FLOAT64 -- what scikit-learn computes when the reference hands it float64
 * matrices (the consensus tail: rf_usages feed refit_spectra on the TPM matrix).  H [k][G] fixed, W_out [N][k] from
 * zero; the product X.H^T, the Gram matrix and the sweeps (sklearn:_cdnmf_fast.pyx:8-38) in float64 over the resident
 * (float32) matrix.  gram (nullable) [k][k]: used INSTEAD of H.H^T, as in cnmf_nnls_gram below.              */
int cnmf_nnls_f64(cnmf_ctx* ctx, int k, const double* H, const double* gram, const cnmf_cd_params* prm,
                  double* W_out, int32_t* n_iter_out, double* viol_out);
/* cnmf_nnls with a caller-supplied Gram matrix gram[k][k] = H.H^T: the rows H_prod [k][G] only enter the product
 * X.H_prod^T.  Lets the final usage refit of consensus() (cnmf.py:960-975: X = tpm[:, hvgs] / std) run on the
 * resident TPM matrix: H_prod = spectra / std on the HVG columns and 0 elsewhere, gram from the HVG block.    */
int cnmf_nnls_gram(cnmf_ctx* ctx, int k, const float* H_prod, const float* gram, const cnmf_cd_params* prm,
                   float* W_out, int32_t* n_iter_out, double* viol_out);
/* n usage refits (cnmf.py:776-802) with ranks ks[r] and fixed spectra H_r (packed [sum k][G]) as ONE pass over X
 * (all H_r are columns of a single X.H^T product) and joint sweeps.  W_out (nullable): [N][k_r] blocks;
 * err_out (nullable): ||X - W_r.H_r||^2 in float64 (cnmf.py:926-930) with W_r taken from the device.         */
int cnmf_nnls_batch(cnmf_ctx* ctx, int n, const int32_t* ks, const float* H, const cnmf_cd_params* prm,
                    float* W_out, int32_t* n_iter_out, double* viol_out, double* err_out);
/* The statistics loop of k_selection_plot (cnmf.py:1119-1135; per k: the stats branch of consensus, :871-936) in one
 * call: for each of the n values ks[i] the merged spectra (R[i] rows, concatenated in `spectra`) go through the
 * consensus core in stats mode (cprm[i]: skip_density = 1, want_silhouette = 1; `uniforms` = the KMeans draws of
 * every k concatenated), then ALL refits run as one batched cnmf_nnls_batch with the prediction errors.
 * Outputs: silhouette_out[n], pred_err_out[n], median_out (nullable) [sum k][G], nnls_iter_out (nullable) [n]. */
int cnmf_kselect_stats(cnmf_ctx* ctx, int n, const int32_t* ks, const int32_t* R, const double* spectra,
                       const cnmf_consensus_params* cprm, const double* uniforms, const cnmf_cd_params* prm,
                       double* silhouette_out, double* pred_err_out, double* median_out, int32_t* nnls_iter_out);

/* ---- text artefacts (host only, no context) ------------------------------------------------
 * rows x cols FINITE doubles (row-major) as the reference's DataFrame.to_csv(sep) prints them (cnmf.py:34-35): per row the
 * label (row_labels: the labels separated by '\n', labels_bytes bytes; NULL: none) and the values, every value as Python's
 * repr(float), `sep` between the fields, '\n' behind the row.  Returns the bytes written, or -(capacity needed) when `cap`
 * is too small (33 bytes per value + the labels + one separator per row).                                   */
int64_t cnmf_format_rows_f64(const double* vals, int64_t rows, int64_t cols, char sep, const char* row_labels,
                             int64_t labels_bytes, char* out, int64_t cap);

/* The range finder of sklearn's randomized_svd (utils/extmath.py:287-357) -- the O(N G) part of init='nndsvd'
 * (decomposition/_nmf.py:316-354; `--init nndsvd`, cnmf.py:1252) -- for a GROUP of restarts: `nblocks` blocks of
 * widths[b] = k_b + 10 columns side by side (sum <= 256, each <= CNMF_KMAX).  transpose = 0: M = X; 1: M = X^T
 * (sklearn transposes when n_samples < n_features).  Q0 [M_cols][C] row-major: the Gaussian start (host RNG, numpy's
 * stream).  n_iter power iterations, normalised by Cholesky-QR on the device; Q_out [M_rows][C]: orthonormal columns
 * per block, B_out [C][M_cols] = Q^T M.  The small SVD / sign flip / NNDSVD split stay with the caller.           */
int cnmf_range_finder(cnmf_ctx* ctx, int transpose, int nblocks, const int32_t* widths, const float* Q0, int n_iter,
                      float* Q_out, float* B_out);
/* (the diagnostic entry points the tests use -- cnmf_debug_* -- are declared in cnmf_hip_debug.h and exist only in a
 * library built with -DCNMF_DEBUG_ABI; they are not part of the drop-in boundary) */

#ifdef __cplusplus
}
#endif
#endif /* CNMF_HIP_H */
