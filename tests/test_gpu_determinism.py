"""Run-to-run determinism through the C-ABI: the batched multiplicative-update kernels (workgroup-cooperative, LDS
double buffering, one barrier per step) and the coordinate-descent batcher must return bit-identical results when the
same call is repeated -- a missing barrier or a buffer hazard shows up here as a difference."""
import numpy as np
import pytest

from cnmf_amd import synth

pytestmark = pytest.mark.gpu


def test_repeated_batches_are_bit_identical(engine):
    X = synth.make_config("C3", dtype=np.float32, n_cells=6000)
    engine.set_matrix(X)
    ks = [5, 9, 13, 7, 11, 6, 12, 8, 10, 20, 17, 24, 32, 3, 16, 1, 9, 9]
    seeds = list(range(100, 100 + len(ks)))

    def run():
        H, W, n, err = engine.nmf_mu_batch(ks, seeds=seeds, max_iter=30, return_W=True, warn=False)
        Hi, Wi, ni, _ = engine.nmf_mu_batch(ks[:6], seeds=seeds[:6], beta_loss="itakura-saito", max_iter=20,
                                            return_W=True, warn=False)
        Hc, _, nc, _ = engine.nmf_batch(ks * 4, seeds=list(range(500, 500 + 4 * len(ks))), max_iter=50, warn=False)
        return [np.concatenate([a.ravel() for a in part]) for part in (H, W, Hi, Wi, Hc)] + [n.copy(), ni.copy(), nc.copy(), np.asarray(err)]

    ref = run()
    for _ in range(2):
        for a, b in zip(ref, run()):
            np.testing.assert_array_equal(a, b)
