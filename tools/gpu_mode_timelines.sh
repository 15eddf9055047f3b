#!/bin/bash
# GPU box: steady-state kernel timelines (rocprofv3 --kernel-trace) of the bench's optional legs:
#   bash tools/gpu_mode_timelines.sh            -> gpurun_out/modes/timeline_<leg>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/modes; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace -d $O/prof_$name -o t -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 300 --no-cpu-baseline --latency-steps 0 --no-kernel-table > $O/bench_$name.json 2> $O/err_$name.txt
  python $GRAFT_REPO_ROOT/tools/timeline.py $(find $O/prof_$name -name "*.db" | head -1) ${N:-44} > $O/timeline_$name.txt 2>&1
  rm -rf $O/prof_$name
  python -c "
import json; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', r['value'], r['ms_per_step'])"
}
run all3 --full-decoder --dtx --rate 48000
run fulldec48k --full-decoder --rate 48000
run fulldec --full-decoder
run dtx --dtx
run r48k --rate 48000
