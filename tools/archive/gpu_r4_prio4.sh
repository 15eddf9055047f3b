cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
run() { env LYRA_HIP_PRIO=$1 timeout 300 python bench.py --no-cpu-baseline --no-verify --no-kernel-table --steps 600 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio $1 bench', r['value'], r['ms_per_step'])"; }
bb() { env LYRA_HIP_PRIO=$1 timeout 300 lyra_amd/batch_bench lyra_amd/assets 4096 16000 9200 0 200 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio $1 batch_bench enc/dec/pipelined', round(r['encode_frames_per_s']/1e6,2), round(r['decode_frames_per_s']/1e6,2), round(r['encode_decode_pipelined_frames_per_s']/1e6,2))"; }
for p in 0,0,2 0,2,0 0,2,2 0,1,2 1,2,2 0,0,0; do
  for i in 1 2 3 4; do run $p; done
  bb $p; bb $p
done | tee gpurun_out/r04/prio_ab4.txt
