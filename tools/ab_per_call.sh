#!/bin/bash
# GPU box: the same legs driven call by call from Python (`--per-call`) vs from one lyra_hip_run_steps_dev call
run() { python bench.py "$@" --no-cpu-baseline --latency-steps 0 --no-kernel-table --steps 300 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-62s %10.0f  %.4f ms' % ('$*', r['value'], r['ms_per_step']))"; }
for i in 1 2; do
for m in "" "--rate 48000" "--full-decoder" "--dtx" "--rate 48000 --full-decoder --dtx"; do
run $m
run --per-call $m
done; done
