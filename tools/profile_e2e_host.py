"""cProfile of the host side of prepare_from_counts and consensus(k) at C3 (where the non-device seconds of the e2e leg go)."""
import cProfile, contextlib, io, os, pstats, shutil, sys, tempfile, time
import numpy as np, pandas as pd, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.cnmf import cNMF
from cnmf_amd.engine import Engine
Ncfg, Gcfg, Kt, mu, sg, dseed = synth.CONFIGS["C3"]
C, _ = synth.topic_counts(Ncfg, Gcfg, Kt, mu, sg, dseed)
keep = C.sum(axis=0) > 0
Ck = C[:, keep]
genes = ["g%d" % j for j in range(Ck.shape[1])]; cells = ["c%d" % i for i in range(Ck.shape[0])]
tpm_csr = sp.csr_matrix((Ck / Ck.sum(axis=1, keepdims=True) * 1e6).astype(np.float32))
eng = Engine(0)
out = tempfile.mkdtemp(prefix="cnmf_prof_")
buf = io.StringIO()
def prof(tag, fn):
    pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
    with contextlib.redirect_stdout(buf):
        r = fn()
    pr.disable(); dt = time.perf_counter() - t
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
    print("==== %s: %.3f s" % (tag, dt)); print("\n".join(s.getvalue().splitlines()[6:40])); return r
try:
    obj = cNMF(output_dir=out, name="c3", engine=eng, compress_merged=False)
    prof("prepare_from_counts", lambda: obj.prepare_from_counts(pd.DataFrame(Ck, index=cells, columns=genes), components=[8, 9], n_iter=20, seed=14, beta_loss="frobenius", tpm=(tpm_csr, genes)))
    prof("factorize", lambda: obj.factorize(write_iter_files=False))
    prof("combine", lambda: obj.combine())
    prof("consensus", lambda: obj.consensus(9, density_threshold=0.5))
finally:
    shutil.rmtree(out, ignore_errors=True)
