#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in timing te1 te3; do
export LYRA_HIP_LIB=$GRAFT_REPO_ROOT/lyra_amd/variants/$v.so
echo "== $v"
EXIT_AT="d0:54,55,-1" python tools/pipeline_probe.py 2>&1 | grep exit_at | sed 's/full  *wall.step *//; s/enc_s0.*dec_s1/dec_s1/'
done
