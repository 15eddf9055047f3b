#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 )
VARIANTS="head cur" bash tools/ab_variants.sh
