"""The streams of known byte counts behind tools/gpu_pmc_calibrate.sh (cnmf_debug_stream): 10 launches each of a 4-B-per-lane
copy, a 16-B-per-lane copy and a read-only LDS-DMA stream over 128 Mi floats (512 MiB read [+ 512 MiB written])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd.engine import Engine
eng = Engine(0)
n = 128 << 20
for width in (1, 4, 0):
    eng._check(eng._lib.cnmf_debug_stream(eng._ctx, width, n, 10))
eng.close()
