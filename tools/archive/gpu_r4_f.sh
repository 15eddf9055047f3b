#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O/prof
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof/c2sb2 -o t -- python $GRAFT_REPO_ROOT/bench.py --config 2 --sub-batches 2 --steps 60 --warmup 5 --no-cpu-baseline --no-verify --no-kernel-table --latency-steps 0 > /dev/null 2> $O/rocprof_c2sb2.err; echo rc=$?
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find $O/prof/c2sb2 -name "*.db" | head -1) 60 > $O/timeline_config2_sb2.txt 2>&1
rm -rf $O/prof
cat $O/timeline_config2_sb2.txt
