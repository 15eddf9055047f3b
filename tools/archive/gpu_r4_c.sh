#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
