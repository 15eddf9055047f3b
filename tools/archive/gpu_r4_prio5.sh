cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
for i in 1 2 3 4 5 6 7 8; do timeout 300 python bench.py --no-cpu-baseline --no-verify --no-kernel-table --steps 1000 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default(0,1,2) bench', r['value'], r['ms_per_step'])"; done
for i in 1 2; do timeout 300 lyra_amd/batch_bench lyra_amd/assets 4096 16000 9200 0 200 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch_bench enc/dec/pipelined', round(r['encode_frames_per_s']/1e6,2), round(r['decode_frames_per_s']/1e6,2), round(r['encode_decode_pipelined_frames_per_s']/1e6,2))"; done
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form', r['value'], r['ms_per_step'], r['verified'])"; done
