"""One timed step of the GENERAL path (count detection off: gemm_mode 5) at the bench's shape, for A/B runs of process-lifetime
knobs (CNMF_G2_XMAP, CNMF_LIB_PATH ...):   CNMF_G2_XMAP=0 python tools/general_ab.py [restarts_per_k]
Prints one JSON line (bench.py::general_path_step: restarts/s, pass A / pass B launch times, roofline fraction)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cnmf_amd import synth  # noqa: E402
from cnmf_amd.cnmf import ledger_seeds  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

rpk = int(sys.argv[1]) if len(sys.argv) > 1 else 20
X = synth.make_config("C3", dtype=np.float32)
ks_all = list(range(5, 14))
led = ledger_seeds(ks_all, 100, 14)
by_k = {k: [int(s) for (kk, _, s) in led if kk == k] for k in ks_all}
out = bench.general_path_step(X, ks_all, by_k, rpk, 64)
out["env"] = {k: v for k, v in os.environ.items() if k.startswith("CNMF_")}
print(json.dumps(out))
