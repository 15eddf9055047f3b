#!/bin/bash
# GPU box: throughput of the per-object plugin path (one host thread per codec, calls combined into device batches)
for n in 64 256 1024; do
python - $n <<'PY'
import sys, numpy as np
n = int(sys.argv[1]); frames = 100
np.random.default_rng(1).integers(-32768, 32768, size=(frames, n, 320)).astype(np.int16).tofile('/tmp/pmt_in.s16')
PY
lyra_amd/plugin_mt_demo lyra_amd/assets /tmp/pmt_in.s16 $n 184 /tmp/pmt_bits.txt /tmp/pmt_out.s16 2>&1 | tail -2
done
