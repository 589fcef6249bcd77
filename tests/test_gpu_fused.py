"""The W half-step inside the pass-A epilogue (kernels_fusedw.hip.h, opt-in CNMF_FUSE_A=1; DESIGN.md section 8: built as the
round-3 review asked, measured slower, kept as the executable form of the argument) must be BIT-IDENTICAL to the stand-alone
sweep with the same per-tile partials (CNMF_FUSE_A=2): spectra, usages, iteration counts, violations -- on a batch of 1024
packed columns over 55 cell tiles (stream-K cuts some tiles: both kinds occur), with ranks the epilogue does not take
(16 < k) mixed in.  The switch is read once per process, so the two arms are child processes (tools/fused_ab.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mixed", [False, True])
def test_fused_w_half_step_is_bit_identical_to_the_stand_alone_sweep(mixed):
    env = dict(os.environ, AB_CELLS="14000", AB_RESTARTS="120", AB_ITERS="12", AB_W="1")
    if mixed:
        env["AB_MIXED"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fused_ab.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0 and "FUSED_AB_IDENTICAL" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
