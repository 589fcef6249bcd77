"""BASELINE config 4's shape as a Kullback-Leibler job handed over SPARSE end to end (round 5): 200 000 cells x 2000 genes at
a real 10x library size (~9 % non-zero), normalised counts and TPM matrix as scipy CSR (the reference's sparse branch),
K = 20, n_iter restarts -- prepare_from_matrix -> factorize -> combine -> consensus(K) with the TPM tail through the mirror
class.  Reports seconds per stage, what the device holds at the end of each stage, and the device memory in use."""
import contextlib, io, json, os, sys, tempfile, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.cnmf import cNMF

n_iter = int(os.environ.get("N_ITER", 20))
K = int(os.environ.get("K", 20))
t = {}
t0 = time.perf_counter()
C, _ = synth.topic_counts(int(os.environ.get("N_CELLS", 200000)), 2000, 20, 5.2, 0.4, 3)
C = C[:, C.sum(axis=0) > 0]
C = C[C.sum(axis=1) > 0]
std = C.std(axis=0, ddof=1, dtype=np.float64)
NC = sp.csr_matrix(C, dtype=np.float64)
NC = NC @ sp.diags(1.0 / std)                                     # counts / std per gene, stays sparse (cnmf.py:537-539)
NC = sp.csr_matrix(NC)
TPM = sp.csr_matrix(sp.diags(1e6 / C.sum(axis=1, dtype=np.float64)) @ sp.csr_matrix(C, dtype=np.float64))
cells = ["c%d" % i for i in range(C.shape[0])]
genes = ["g%d" % j for j in range(C.shape[1])]
del C
t["synthesize_s"] = time.perf_counter() - t0
out = tempfile.mkdtemp(prefix="cnmf_e2e_c4kl_")
obj = cNMF(output_dir=out, name="c4kl", compress_merged=False)
buf = io.StringIO()


def mem_gb():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 1e9


res = {}
t0 = time.perf_counter(); obj.prepare_from_matrix((NC, cells, genes), components=[K], n_iter=n_iter, seed=14, beta_loss="kullback-leibler", tpm=(TPM, genes)); t["prepare_from_matrix_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
with contextlib.redirect_stdout(buf):
    obj.factorize(write_iter_files=False)
t["factorize_s"] = time.perf_counter() - t0
res["after_factorize"] = {"resident": [a for a, b in obj.engine.matrix_images().items() if b], "device_memory_in_use_GB": mem_gb()}
n_it = np.asarray(obj.last_factorize_stats["n_iter"])
t0 = time.perf_counter()
with contextlib.redirect_stdout(buf):
    obj.combine()
t["combine_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
with contextlib.redirect_stdout(buf):
    med, usages = obj.consensus(K, density_threshold=2.0)
t["consensus_with_tpm_tail_s"] = time.perf_counter() - t0
res["after_consensus"] = {"resident": [a for a, b in obj.engine.matrix_images().items() if b], "device_memory_in_use_GB": mem_gb()}
res.update(config="C4 shape as a KL job, sparse end to end: %d x %d, %.1f %% non-zero, K=%d, n_iter=%d, 1x MI355X" % (NC.shape[0], NC.shape[1], 100.0 * NC.nnz / np.prod(NC.shape), K, n_iter),
           stages=t, total_prepare_to_consensus_s=sum(v for k_, v in t.items() if k_ != "synthesize_s"), restarts=int(len(n_it)),
           mean_iterations_per_restart=float(n_it.mean()), us_per_restart_iteration=1e6 * t["factorize_s"] / float(n_it.sum()))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/e2e_c4_kl.json", "w"), indent=1)
print(json.dumps(res))
