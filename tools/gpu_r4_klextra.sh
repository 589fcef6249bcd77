#!/bin/bash
# the bench line's kl_non_zero_path extra on its own (bench.py::kl_non_zero_path)
cd $GRAFT_REPO_ROOT
timeout 80 python -c "
import json, bench
print(json.dumps(bench.kl_non_zero_path(list(range(5, 14)))))
" 2>&1 | tail -2
