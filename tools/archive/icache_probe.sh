#!/bin/bash
# GPU box: instruction-cache counters of the stage kernels (are the once-executed straight-line kernels fetch-bound?)
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQ_INST_LEVEL|SQC_" | head -40
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  rm -rf /tmp/icp; rocprofv3 --pmc $set --kernel-trace -d /tmp/icp -o x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-table --steps 4 --warmup 2 --latency-steps 0 > /dev/null 2> /tmp/icp.err
  python - <<'PY'
import sqlite3, glob
for db in glob.glob('/tmp/icp/**/x_results.db', recursive=True) + glob.glob('/tmp/icp/x_results.db'):
    cur = sqlite3.connect(db).cursor()
    q = "select kernel_name, counter_name, avg(value), avg(duration), count(*) from counters_collection where kernel_name like 'lyra::%' group by kernel_name, counter_name"
    for k, c, v, d, n in cur.execute(q):
        print(f"{k.split('(')[0].replace('lyra::',''):20s} {c:30s} {v:14.1f}  dur_us {d/1e3:7.1f} n {n}")
    break
else:
    print("no db;", open('/tmp/icp.err').read()[-400:])
PY
done
