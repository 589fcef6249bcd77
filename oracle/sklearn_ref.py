"""The REAL third-party calls the reference makes on the hot path (TEST ORACLE / CPU baseline).

The reference tree (/root/reference) cannot travel to the GPU box, but the arithmetic of
its hot path does not live there: ``cNMF._nmf`` (cnmf.py:661-674) is a one-line call into
scikit-learn, which IS installed on the GPU box (same image, 1.7.2).  These wrappers make
exactly those calls with exactly the keyword arguments the reference persists in
``nmf_idvrun_params.yaml`` (cnmf.py:618-631), so they serve as

* the pin for the numpy restatements (tests/test_oracle_*.py), and
* the ``cpu_baseline`` of kind "reference" in bench.py.
"""
import warnings

import numpy as np


def cnmf_nmf_kwargs(beta_loss="frobenius", alpha_usage=0.0, alpha_spectra=0.0, init="random",
                    max_iter=1000):
    """The kwargs dict built by cNMF.get_nmf_iter_params (cnmf.py:618-631)."""
    kw = dict(alpha_W=alpha_usage, alpha_H=alpha_spectra, l1_ratio=0.0, beta_loss=beta_loss,
              solver="mu", tol=1e-4, max_iter=max_iter, init=init)
    if beta_loss == "frobenius":
        kw["solver"] = "cd"
    return kw


def nmf(X, k, seed, **overrides):
    """cNMF._nmf(X, kwargs) (cnmf.py:672): returns (spectra, usages, n_iter)."""
    from sklearn.decomposition import non_negative_factorization
    kw = cnmf_nmf_kwargs()
    kw.update(overrides)
    kw["random_state"] = seed
    kw["n_components"] = k
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        usages, spectra, n_iter = non_negative_factorization(X, **kw)
    return spectra, usages, n_iter


def refit_usage(X, spectra, **overrides):
    """cNMF.refit_usage (cnmf.py:792-798): NNLS with fixed H."""
    from sklearn.decomposition import non_negative_factorization
    kw = cnmf_nmf_kwargs()
    kw.update(overrides)
    kw.update(dict(n_components=spectra.shape[0], H=spectra, update_H=False))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        usages, _, n_iter = non_negative_factorization(X, **kw)
    return usages, n_iter


def ledger(ks, n_iter, seed):
    """Restart ledger of cNMF.get_nmf_iter_params (cnmf.py:593-610): rows (k, iter, nmf_seed).

    NB the seed vector has len(ks)*n_iter entries for the UN-deduplicated ks (cnmf.py:599)
    while rows iterate the sorted de-duplicated list (cnmf.py:597,605)."""
    import itertools
    if isinstance(ks, int):
        ks = [ks]
    k_list = sorted(set(list(ks)))
    n_runs = len(ks) * n_iter
    np.random.seed(seed=seed)
    nmf_seeds = np.random.randint(low=1, high=(2 ** 31) - 1, size=n_runs)
    return [(k, r, int(nmf_seeds[i])) for i, (k, r) in enumerate(itertools.product(k_list, range(n_iter)))]
