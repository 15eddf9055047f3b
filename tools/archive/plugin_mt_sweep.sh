#!/bin/bash
# GPU box: the per-object plugin path (lyra_amd/plugin_mt_demo) over the call combiner's tunables:  tools/plugin_mt_sweep.sh
python - <<'PY'
import numpy as np
for n in (64, 256, 1024):
    np.random.default_rng(1).integers(-32768, 32768, size=(100, n, 320)).astype(np.int16).tofile('/tmp/pmt_in_%d.s16' % n)
PY
run() { for n in $NS; do echo "$* n=$n: $(env "$@" lyra_amd/plugin_mt_demo lyra_amd/assets /tmp/pmt_in_$n.s16 $n 184 /tmp/pmt_bits.txt /tmp/pmt_out.s16 2>&1 | tail -2 | tr '\n' ' ' | sed -e 's/(extract.*//' -e 's/plugin_calls [0-9]* //')"; done; }
NS="${NS:-256 1024}"
run LYRA_HIP_COMBINER_QUIET_US=0
run LYRA_HIP_COMBINER_QUIET_US=10
run LYRA_HIP_COMBINER_QUIET_US=40
run LYRA_HIP_COMBINER_QUIET_US=150
run LYRA_HIP_COMBINER_QUIET_US=40 LYRA_HIP_COMBINER_FAN=16
run LYRA_HIP_COMBINER_QUIET_US=40 LYRA_HIP_COMBINER_FAN=2
run LYRA_HIP_COMBINER_QUIET_US=400 LYRA_HIP_COMBINER_GATHER_US=1000
