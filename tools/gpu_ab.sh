#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|rror" | tail -3
python bench.py --no-cpu-baseline --restarts-per-k 10 --steps 1 --warmup 1 2>/dev/null | cut -c1-120
