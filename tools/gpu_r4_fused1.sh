#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
AB_W=1 timeout 600 python tools/fused_ab.py 2>&1 | tail -12
AB_W=1 AB_MIXED=1 AB_ITERS=25 timeout 600 python tools/fused_ab.py 2>&1 | tail -8
