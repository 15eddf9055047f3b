#!/bin/bash
# GPU box, round 4 first contact: GPU parity suite, then the driver-form bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_k20.json 2> gpurun_out/r04/bench_k20.err; tail -c 1500 gpurun_out/r04/bench_k20.json; tail -3 gpurun_out/r04/bench_k20.err
