#!/bin/bash
# GPU box: parity of a library variant against the oracle (the smoke test + the golden / oracle parity tests through
# LYRA_HIP_LIB), then the alternating driver-form / sustained A/B against the shipped library:
#   tools/ab_variant.sh lyra_amd/variants/x.so [rounds] [regions]
V=$1; R=${2:-3}; N=${3:-9}
echo "== parity of $V"
LYRA_HIP_LIB=$PWD/$V python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
LYRA_HIP_LIB=$PWD/$V python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
echo "== A/B"
bash tools/ab_k20.sh $R $N default $PWD/$V
