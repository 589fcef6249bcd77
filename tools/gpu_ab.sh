#!/bin/bash
for m in 3 4; do echo "== mode $m"; CNMF_GEMM3=$m timeout 300 python tools/probe_gemm3c.py 2>&1 | grep -v amdgpu.ids | tail -7; done
