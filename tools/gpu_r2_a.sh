#!/bin/bash
# round-2 GPU pass A: regression tests + every bench configuration + rocprofv3 of the default bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ) 
tail -3 $O/pytest.log
for c in 3 2 4 5; do
  timeout 400 python bench.py --config $c --steps 50 > $O/bench_c$c.json 2> $O/bench_c$c.err
  echo "config $c rc=$?"; head -c 600 $O/bench_c$c.json; echo
done
timeout 300 python bench.py --with-logmel --steps 50 --no-cpu-baseline > $O/bench_logmel.json 2> $O/bench_logmel.err
echo "logmel rc=$?"; head -c 300 $O/bench_logmel.json; echo
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec rm {} \;
head -20 $O/kernel_stats.csv
ls -R $O/prof | head -20
