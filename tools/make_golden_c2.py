"""tests/golden/ref_c2_consensus.npz: the consensus spectra of BASELINE config 2 computed ENTIRELY by the CPU reference
path -- scikit-learn float64 restarts (the reference's own call, cnmf.py:672, through oracle/sklearn_ref.py) for the
full ledger K = 10, n_iter = 100, seed 14, then the consensus core of cnmf.py:871-916 (oracle/consensus.py, pinned to
sklearn / pandas) at density_threshold 0.5.  tests/test_gpu_configs.py runs the same ledger on the device, pushes the
device's merged spectra through the device's consensus and compares the consensus spectra -- the artefact users
consume -- at the reference's own bar (sum of squared differences < 1e-4, tests/test_reproducibility.py:12).

    python tools/make_golden_c2.py          # ~5 min on 8 cores
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from cnmf_amd import synth
from cnmf_amd.cnmf import ledger_seeds
from oracle import consensus as oc
from oracle import sklearn_ref

_X = None


def _one(job):
    k, seed, f32 = job
    from threadpoolctl import threadpool_limits
    with threadpool_limits(1):
        H, _, n = sklearn_ref.nmf(_X.astype(np.float32) if f32 else _X, k, seed)
    return H.astype(np.float64), n


def _pipeline(pool, K, f32):
    led = ledger_seeds([K], 100, 14)
    t0 = time.time()
    res = pool.map(_one, [(k, int(s), f32) for k, _, s in led], chunksize=1)
    its = np.array([n for _, n in res], dtype=np.int32)
    print("K=%d %s: 100 sklearn restarts in %.0f s, iterations %d..%d (mean %.0f)"
          % (K, "float32" if f32 else "float64", time.time() - t0, its.min(), its.max(), its.mean()), flush=True)
    merged = np.concatenate([H for H, _ in res], axis=0)             # (iter asc, topic asc), cnmf.py:765-770
    core = oc.consensus_core(merged, _X, K, density_threshold=0.5)
    return its, core


def main():
    """K = 10 (= K_true: BASELINE config 2 as written, restarts of ~40 iterations) and K = 13 on the same matrix
    (rank above the data's own: restarts of hundreds of iterations, the regime the bench spends its time in).  For
    both, scikit-learn's OWN float32 pipeline is run beside its float64 one: how far the consensus spectra move when
    only the working precision changes is the calibration the device is held against."""
    global _X
    import multiprocessing as mp
    from oracle import nmf_cd
    _X = synth.make_config("C2", dtype=np.float64)
    out = dict(x_checksum=np.array([_X.sum(), (_X * _X).sum()]), shape=np.array(_X.shape))
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        for K in (10, 13):
            its, core = _pipeline(pool, K, False)
            its32, core32 = _pipeline(pool, K, True)
            ref, m32 = core["median_spectra"], core32["median_spectra"]
            perm, cos = nmf_cd.match_components(ref, m32)
            m32 = m32[perm]
            drift = np.array([((m32 - ref) ** 2).sum(), np.linalg.norm(m32 - ref) / np.linalg.norm(ref),
                              np.abs(m32 - ref).max() / np.abs(ref).max(), cos.min(),
                              abs(int(core32["density_filter"].sum()) - int(core["density_filter"].sum()))])
            print("K=%d kept %d of %d; sklearn float32 pipeline vs float64: sum sq %.3g, rel fro %.3g, rel max %.3g, "
                  "min cos %.6f, |delta kept| %d" % (K, core["density_filter"].sum(), len(core["density_filter"]), *drift),
                  flush=True)
            out.update({"k%d_n_iter" % K: its, "k%d_local_density" % K: core["local_density"],
                        "k%d_density_filter" % K: core["density_filter"], "k%d_median_spectra" % K: ref,
                        "k%d_n_kept" % K: np.array([int(core["density_filter"].sum())]), "k%d_f32_drift" % K: drift})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_c2_consensus.npz"), **out)


if __name__ == "__main__":
    main()
