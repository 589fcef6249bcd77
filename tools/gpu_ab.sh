#!/bin/bash
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|rror\|assert" | tail -6
