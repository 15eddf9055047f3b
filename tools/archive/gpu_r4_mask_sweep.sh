#!/bin/bash
# GPU box: CU-masked chains vs default over the batch size (policy threshold)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
H="00ff00ff,ff00ff00,00ff00ff,ff00ff00"
run() { local label=$1 B=$2 bits=$3; shift 3
  env "$@" timeout 300 python bench.py --config 3 --streams $B --bits $bits --no-cpu-baseline --no-verify --no-kernel-table --steps 600 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s B=%5d bits=%3d  %.3f M frames/s  %.4f ms/step' % ('$label', $B, $bits, r['value']/1e6, r['ms_per_step']))"; }
for bits in 64 184; do
for B in 256 512 1024 1536 2048 2560 3072; do
  run default $B $bits A=1
  run masks $B $bits LYRA_HIP_CU_MASKS=$H
done; done 2>&1 | tee gpurun_out/r04/cumask_batch_sweep.txt
