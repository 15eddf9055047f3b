#!/bin/bash
# Everything under profiles/ for one round, in one GPU call:  TAG=r02 bash tools/gpu_profile_round.sh
#   bench lines of every BASELINE config, rocprofv3 kernel-trace stats of the default bench command, FETCH_SIZE /
#   WRITE_SIZE passes (separate --pmc runs, MI355X_MICROARCH.md "HBM"), SQ counter passes.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r03}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; rm -rf $O; mkdir -p $O/prof
export TMPDIR=/tmp
for c in 3 2 4 5; do
  timeout 400 python bench.py --config $c > $O/${TAG}_bench_config$c.json 2> $O/bench_c$c.err; echo "config $c rc=$?"
done
timeout 300 python bench.py --with-logmel --no-cpu-baseline > $O/${TAG}_bench_with_logmel.json 2>> $O/bench.err
timeout 300 python bench.py --full-decoder --no-cpu-baseline > $O/${TAG}_bench_full_decoder.json 2>> $O/bench.err
timeout 300 python bench.py --dtx --no-cpu-baseline > $O/${TAG}_bench_dtx.json 2>> $O/bench.err
timeout 300 python bench.py --rate 48000 --no-cpu-baseline > $O/${TAG}_bench_48k.json 2>> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 3 > $O/${TAG}_bench_driver_form_k20.json 2>> $O/bench.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof/trace -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --no-cpu-baseline --latency-steps 0 > $O/${TAG}_bench_under_rocprof.json 2> $O/rocprof_trace.err; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof/fetch -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-table --latency-steps 0 > /dev/null 2> $O/rocprof_fetch.err; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof/write -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-table --latency-steps 0 > /dev/null 2> $O/rocprof_write.err; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $O/prof $TAG $O > $O/summary.txt 2>&1; tail -20 $O/summary.txt
timeout 600 python tools/pmc_probe.py > $O/${TAG}_pmc_sq.txt 2>&1; echo "sq rc=$?"
python tools/timeline.py $(find $O/prof/trace -name "*.db" | head -1) 22 > $O/${TAG}_timeline.txt 2>&1
rm -rf $O/prof
ls -la $O
