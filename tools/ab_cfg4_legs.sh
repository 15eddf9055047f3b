#!/bin/bash
run() { python bench.py "$@" --no-cpu-baseline --latency-steps 0 --no-kernel-table --steps 300 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-62s %10.0f  %.4f ms' % ('$*', r['value'], r['ms_per_step']))"; }
for i in 1 2; do
run --config 4
run --config 4 --full-decoder
run --config 4 --rate 48000
run --config 4 --full-decoder --rate 48000
run --config 4 --sub-batches 1
run --config 4 --sub-batches 1 --full-decoder --rate 48000
done
