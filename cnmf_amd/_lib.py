"""ctypes loader for ``libcnmf_hip.so`` (the C-ABI declared in include/cnmf_hip.h).

The library is built IN-TREE (``cnmf_amd/libcnmf_hip.so``) by ``build()`` below /
``__graft_entry__.build()`` with ``hipcc --offload-arch=gfx950``.  There is no CPU
fallback anywhere in this package: if the shared object is missing or no GPU is
visible, the product path raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcnmf_hip.so")
SRC_DIR = os.path.join(_HERE, "csrc")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

# every symbol include/cnmf_hip.h declares (tests/test_abi.py checks the header against this)
SYMBOLS = [
    "cnmf_device_count", "cnmf_create", "cnmf_destroy", "cnmf_last_error", "cnmf_reload_env", "cnmf_version",
    "cnmf_set_matrix", "cnmf_set_matrix_csr", "cnmf_set_count_detection", "cnmf_get_shape", "cnmf_matrix_images", "cnmf_get_matrix",
    "cnmf_col_moments", "cnmf_scale_columns", "cnmf_row_sums",
    "cnmf_nmf_cd_batch", "cnmf_nmf_cd_batch_resident", "cnmf_get_iteration_means", "cnmf_set_iteration_hints", "cnmf_nnls",
    "cnmf_consensus", "cnmf_pairwise_distances", "cnmf_prediction_error", "cnmf_nmf_mu_batch", "cnmf_mu_refit_f64", "cnmf_x_matmul",
    "cnmf_xt_matmul_f64", "cnmf_nnls_spectra", "cnmf_nnls_f64", "cnmf_nnls_gram", "cnmf_nnls_batch", "cnmf_kselect_stats",
    "cnmf_comm_unique_id", "cnmf_comm_init", "cnmf_comm_finalize", "cnmf_comm_rank", "cnmf_comm_world",
    "cnmf_allgather_bytes", "cnmf_allgather_spectra",
    "cnmf_spectra_rows", "cnmf_spectra_reset", "cnmf_spectra_fetch", "cnmf_spectra_fetch_rows", "cnmf_spectra_genes", "cnmf_spectra_append",
    "cnmf_consensus_store", "cnmf_kselect_stats_store",
    "cnmf_range_finder", "cnmf_format_rows_f64",
]
# test hooks (include/cnmf_hip_debug.h): present only in a library built with -DCNMF_DEBUG_ABI -- the in-tree default,
# because tests/ call them; CNMF_PRODUCT_BUILD=1 in the environment of build() leaves them out
DEBUG_SYMBOLS = ["cnmf_debug_stream", "cnmf_debug_gemm", "cnmf_debug_gemm3", "cnmf_debug_gemm3c", "cnmf_debug_gemm2h",
                 "cnmf_debug_standard_normal"]

COMM_ID_BYTES = 128

CNMF_KMAX = 128
CNMF_MU_KMAX = 64


class CdParams(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int), ("kc_max", C.c_int),
                ("l1_reg_W", C.c_double), ("l2_reg_W", C.c_double),
                ("l1_reg_H", C.c_double), ("l2_reg_H", C.c_double),
                ("lag", C.c_int), ("profile", C.c_int)]


class BatchStats(C.Structure):
    _fields_ = [("outer_iterations", C.c_int64), ("restart_iterations", C.c_int64),
                ("column_iterations", C.c_int64), ("restart_column_iterations", C.c_int64),
                ("gpu_ms", C.c_double),
                ("passA_ms", C.c_double), ("passB_ms", C.c_double),
                ("passA_launches", C.c_int64), ("passB_launches", C.c_int64),
                ("kc", C.c_int32), ("nsplit", C.c_int32), ("gemm_mode", C.c_int32), ("reserved_", C.c_int32),
                ("tail_iterations", C.c_int64), ("tail_live_columns", C.c_int64), ("tail_ms", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class ConsensusParams(C.Structure):
    _fields_ = [("k", C.c_int), ("n_neighbors", C.c_int), ("density_threshold", C.c_double),
                ("skip_density", C.c_int), ("want_silhouette", C.c_int), ("n_init", C.c_int),
                ("max_iter", C.c_int), ("tol", C.c_double)]


def sources():
    return sorted(os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(INCLUDE_DIR, "cnmf_hip.h"), os.path.join(INCLUDE_DIR, "cnmf_hip_debug.h")]
    return any(os.path.getmtime(s) > t for s in deps)


def build(force=False, verbose=False, out=None):
    """Compile csrc/cnmf_hip.hip for gfx950 into cnmf_amd/libcnmf_hip.so (in-tree).  The diagnostic entry points the tests
    use (cnmf_debug_*, include/cnmf_hip_debug.h) are compiled in unless CNMF_PRODUCT_BUILD=1 is set."""
    out = out or LIB_PATH
    if not force and out == LIB_PATH and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    product = os.environ.get("CNMF_PRODUCT_BUILD", "0") not in ("", "0")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-Wno-unused-result"] + ([] if product else ["-DCNMF_DEBUG_ABI"]) + (
           os.environ.get("CNMF_HIPCC_FLAGS", "").split()) + [          # (A/B builds: tools/, e.g. -DCNMF_SP_PF=1)
           # MFMA accumulators in plain VGPRs (gfx950 has one unified file): no v_accvgpr moves around the
           # elementwise work between chained MFMAs (kernels_mu_mfma.hip.h)
           "-mllvm", "-amdgpu-mfma-vgpr-form=1",
           os.path.join(SRC_DIR, "cnmf_hip.hip"), "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


_lib = None


def load():
    """Load the shared object and declare the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CNMF_LIB_PATH") or LIB_PATH        # (A/B builds of the library: tools/)
    if not os.path.exists(path):
        raise ImportError(
            "cnmf_amd: %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(path)
    vp, i32, i64, f32p, dblp = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_double)
    i32p, u32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    lib.cnmf_device_count.restype = i32
    lib.cnmf_create.restype = vp
    lib.cnmf_create.argtypes = [i32]
    lib.cnmf_destroy.restype = None
    lib.cnmf_destroy.argtypes = [vp]
    lib.cnmf_last_error.restype = C.c_char_p
    lib.cnmf_last_error.argtypes = [vp]
    lib.cnmf_matrix_images.restype = i32
    lib.cnmf_matrix_images.argtypes = [vp, i32p]
    lib.cnmf_reload_env.restype = i32
    lib.cnmf_reload_env.argtypes = [vp]
    lib.cnmf_version.restype = C.c_char_p
    lib.cnmf_set_matrix.restype = i32
    lib.cnmf_set_matrix.argtypes = [vp, f32p, i64, i64]
    lib.cnmf_set_matrix_csr.restype = i32
    lib.cnmf_set_matrix_csr.argtypes = [vp, i32p, i32p, f32p, i64, i64]
    lib.cnmf_set_count_detection.restype = i32
    lib.cnmf_set_count_detection.argtypes = [vp, i32]
    lib.cnmf_get_shape.restype = i32
    lib.cnmf_get_shape.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    lib.cnmf_get_matrix.restype = i32
    lib.cnmf_get_matrix.argtypes = [vp, f32p]
    lib.cnmf_col_moments.restype = i32
    lib.cnmf_col_moments.argtypes = [vp, dblp, dblp]
    lib.cnmf_scale_columns.restype = i32
    lib.cnmf_scale_columns.argtypes = [vp, dblp]
    lib.cnmf_row_sums.restype = i32
    lib.cnmf_row_sums.argtypes = [vp, dblp]
    lib.cnmf_nmf_cd_batch.restype = i32
    lib.cnmf_nmf_cd_batch.argtypes = [vp, i32, i32p, i32, u32p, dblp, f32p, f32p,
                                      C.POINTER(CdParams), f32p, f32p, i32p, dblp,
                                      C.POINTER(BatchStats)]
    lib.cnmf_nmf_cd_batch_resident.restype = i32
    lib.cnmf_nmf_cd_batch_resident.argtypes = [vp, i32, i32p, i32, u32p, dblp, f32p, f32p,
                                               C.POINTER(CdParams), i32p, dblp,
                                               C.POINTER(BatchStats)]
    lib.cnmf_nmf_mu_batch.restype = i32
    lib.cnmf_nmf_mu_batch.argtypes = [vp, i32, i32p, i32, u32p, dblp, f32p, f32p, i32, i32,
                                      C.POINTER(CdParams), f32p, f32p, i32p, dblp]
    lib.cnmf_spectra_fetch_rows.restype = i32
    lib.cnmf_spectra_fetch_rows.argtypes = [vp, i64, i64, f32p]
    lib.cnmf_mu_refit_f64.restype = i32
    lib.cnmf_mu_refit_f64.argtypes = [vp, i32, i32, i32, dblp, dblp, C.c_double, C.POINTER(CdParams), dblp, i32p, dblp]
    lib.cnmf_get_iteration_means.restype = i32
    lib.cnmf_get_iteration_means.argtypes = [vp, dblp]
    lib.cnmf_set_iteration_hints.restype = i32
    lib.cnmf_set_iteration_hints.argtypes = [vp, i32, i32p, dblp]
    lib.cnmf_nnls.restype = i32
    lib.cnmf_nnls.argtypes = [vp, i32, f32p, C.POINTER(CdParams), f32p, i32p, dblp]
    lib.cnmf_pairwise_distances.restype = i32
    lib.cnmf_pairwise_distances.argtypes = [vp, dblp, i32, i32, i32p, i32, dblp, dblp]
    lib.cnmf_consensus.restype = i32
    lib.cnmf_consensus.argtypes = [vp, dblp, i32, i32, C.POINTER(ConsensusParams), dblp,
                                   dblp, i32p, i32p, dblp, dblp, dblp]
    lib.cnmf_prediction_error.restype = i32
    lib.cnmf_prediction_error.argtypes = [vp, i32, dblp, dblp, dblp]
    lib.cnmf_x_matmul.restype = i32
    lib.cnmf_x_matmul.argtypes = [vp, i32, f32p, i32, f32p]
    lib.cnmf_format_rows_f64.restype = i64
    lib.cnmf_format_rows_f64.argtypes = [dblp, i64, i64, C.c_char, C.c_char_p, i64, C.c_void_p, i64]
    lib.cnmf_range_finder.restype = i32
    lib.cnmf_range_finder.argtypes = [vp, i32, i32, i32p, f32p, i32, f32p, f32p]
    lib.cnmf_xt_matmul_f64.restype = i32
    lib.cnmf_xt_matmul_f64.argtypes = [vp, i32, dblp, i32, dblp, dblp, dblp]
    lib.cnmf_nnls_spectra.restype = i32
    lib.cnmf_nnls_spectra.argtypes = [vp, i32, dblp, C.POINTER(CdParams), dblp, i32p, dblp]
    lib.cnmf_nnls_f64.restype = i32
    lib.cnmf_nnls_f64.argtypes = [vp, i32, dblp, dblp, C.POINTER(CdParams), dblp, i32p, dblp]
    lib.cnmf_nnls_gram.restype = i32
    lib.cnmf_nnls_gram.argtypes = [vp, i32, f32p, f32p, C.POINTER(CdParams), f32p, i32p, dblp]
    lib.cnmf_nnls_batch.restype = i32
    lib.cnmf_nnls_batch.argtypes = [vp, i32, i32p, f32p, C.POINTER(CdParams), f32p, i32p, dblp, dblp]
    lib.cnmf_kselect_stats.restype = i32
    lib.cnmf_kselect_stats.argtypes = [vp, i32, i32p, i32p, dblp, C.POINTER(ConsensusParams), dblp,
                                       C.POINTER(CdParams), dblp, dblp, dblp, i32p]
    u8p = C.POINTER(C.c_ubyte)
    lib.cnmf_comm_unique_id.restype = i32
    lib.cnmf_comm_unique_id.argtypes = [u8p]
    lib.cnmf_comm_init.restype = i32
    lib.cnmf_comm_init.argtypes = [vp, u8p, i32, i32]
    lib.cnmf_comm_finalize.restype = i32
    lib.cnmf_comm_finalize.argtypes = [vp]
    lib.cnmf_comm_rank.restype = i32
    lib.cnmf_comm_rank.argtypes = [vp]
    lib.cnmf_comm_world.restype = i32
    lib.cnmf_comm_world.argtypes = [vp]
    lib.cnmf_allgather_bytes.restype = i32
    lib.cnmf_allgather_bytes.argtypes = [vp, vp, i64, vp]
    lib.cnmf_allgather_spectra.restype = i32
    lib.cnmf_allgather_spectra.argtypes = [vp, f32p, i64, i64, i64, f32p]
    lib.cnmf_spectra_rows.restype = i64
    lib.cnmf_spectra_rows.argtypes = [vp]
    lib.cnmf_spectra_append.restype = i32
    lib.cnmf_spectra_append.argtypes = [vp, f32p, i64, i64]
    lib.cnmf_spectra_genes.restype = i64
    lib.cnmf_spectra_genes.argtypes = [vp]
    i64p = C.POINTER(C.c_int64)
    lib.cnmf_consensus_store.restype = i32
    lib.cnmf_consensus_store.argtypes = [vp, i64p, i32, i32, C.POINTER(ConsensusParams), dblp,
                                         dblp, i32p, i32p, dblp, dblp, dblp]
    lib.cnmf_kselect_stats_store.restype = i32
    lib.cnmf_kselect_stats_store.argtypes = [vp, i32, i32p, i32p, i64p, C.POINTER(ConsensusParams), dblp,
                                             C.POINTER(CdParams), dblp, dblp, dblp, i32p]
    lib.cnmf_spectra_reset.restype = i32
    lib.cnmf_spectra_reset.argtypes = [vp]
    lib.cnmf_spectra_fetch.restype = i32
    lib.cnmf_spectra_fetch.argtypes = [vp, f32p]
    lib.has_debug_abi = all(hasattr(lib, n) for n in DEBUG_SYMBOLS)      # (a product build has none of them)
    if lib.has_debug_abi:
        lib.cnmf_debug_stream.restype = i32
        lib.cnmf_debug_stream.argtypes = [vp, i32, C.c_longlong, i32]
        lib.cnmf_debug_gemm.restype = i32
        lib.cnmf_debug_gemm.argtypes = [vp, i32, i32, f32p, f32p, f32p, i32, i32, i32, i32, dblp, i32]
        lib.cnmf_debug_gemm3.restype = i32
        lib.cnmf_debug_gemm3.argtypes = [vp, f32p, f32p, f32p, i32, i32, i32, i32, dblp, i32]
        lib.cnmf_debug_gemm3c.restype = i32
        lib.cnmf_debug_gemm3c.argtypes = [vp, f32p, f32p, f32p, i32, i32, i32, i32, dblp, i32]
        lib.cnmf_debug_gemm2h.restype = i32
        lib.cnmf_debug_gemm2h.argtypes = [vp, f32p, f32p, f32p, i32, i32, i32, i32, i32, dblp, i32]
        lib.cnmf_debug_standard_normal.restype = i32
        lib.cnmf_debug_standard_normal.argtypes = [vp, C.c_uint32, i64, dblp]
    _lib = lib
    return lib
