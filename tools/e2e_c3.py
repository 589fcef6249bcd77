"""End-to-end wall-clock on one MI355X for the north-star configuration (merged spectra written without
zlib unless COMPRESS_MERGED=1 -- same npz container):
N=50 000 cells x 2000 HVGs, K=5..13, n_iter=100 (900 restarts), through the host mirror of the
reference's cNMF object: prepare_from_matrix -> factorize -> combine -> k_selection_stats -> consensus."""
import json, os, sys, tempfile, time
import numpy as np
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.cnmf import cNMF

n_iter = int(os.environ.get("N_ITER", 100))
beta_loss = os.environ.get("BETA_LOSS", "frobenius")          # kullback-leibler: the multiplicative-update solver
t = {}
t0 = time.perf_counter(); X = synth.make_config("C3", dtype=np.float32); t["synthesize_input_s"] = time.perf_counter() - t0
df = pd.DataFrame(X, index=["c%d" % i for i in range(X.shape[0])], columns=["g%d" % j for j in range(X.shape[1])])
out = tempfile.mkdtemp(prefix="cnmf_e2e_")
obj = cNMF(output_dir=out, name="c3", compress_merged=os.environ.get("COMPRESS_MERGED", "0") == "1")
tpm = None
if os.environ.get("WITH_TPM", "1") == "1":
    # TPM matrix of the same cells (counts / library size x 1e6), handed over SPARSE (CSR) like a sparse tpm.h5ad:
    # the consensus tail (TPM spectra, OLS z-scores, final usage refit) then runs on the device from one CSR upload
    import scipy.sparse as sp
    t0 = time.perf_counter()
    C, _ = synth.topic_counts(*synth.CONFIGS["C3"][:5], seed=synth.CONFIGS["C3"][5])
    C = C[:, C.sum(axis=0) > 0]
    tpm = (sp.csr_matrix((C / C.sum(axis=1, keepdims=True) * 1e6).astype(np.float32)), list(df.columns))
    t["synthesize_tpm_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); obj.prepare_from_matrix(df, components=list(range(5, 14)), n_iter=n_iter, seed=14, beta_loss=beta_loss, tpm=tpm); t["prepare_from_matrix_s"] = time.perf_counter() - t0
import io, contextlib
buf = io.StringIO()
t0 = time.perf_counter()
with contextlib.redirect_stdout(buf):
    obj.factorize(write_iter_files=os.environ.get("WRITE_ITER", "0") == "1")
t["factorize_s"] = time.perf_counter() - t0
st = obj.last_factorize_stats
t0 = time.perf_counter()
with contextlib.redirect_stdout(buf):
    obj.combine()
t["combine_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); stats = obj.k_selection_stats(); t["k_selection_stats_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); obj.k_selection_stats(batched=False); t_loop = time.perf_counter() - t0
t0 = time.perf_counter(); med, usages = obj.consensus(9, density_threshold=0.5); t["consensus_k9_s"] = time.perf_counter() - t0
res = dict(config="C3 north star: 50000 x 2000, K=5..13, n_iter=%d (%d restarts), beta_loss=%s, 1x MI355X" % (n_iter, 9 * n_iter, beta_loss),
           stages=t, total_prepare_to_consensus_s=sum(v for k, v in t.items() if not k.startswith("synthesize")),
           k_selection_per_k_loop_s=t_loop, consensus_includes_tpm_tail=tpm is not None,
           restarts=9 * n_iter, restarts_per_s=9 * n_iter / t["factorize_s"],
           mean_iterations_per_restart=float(np.mean(st["n_iter"])), gpu_ms=st.get("gpu_ms"),
           factorize_host_seconds=st.get("host_seconds"),
           k_selection=stats.to_dict(orient="list"))
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/e2e_c3%s.json" % ("" if beta_loss == "frobenius" else "_kl"), "w"), indent=1)
