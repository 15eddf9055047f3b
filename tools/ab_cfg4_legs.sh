run() { python bench.py "$@" --no-cpu-baseline --latency-steps 0 --no-kernel-table --steps 300 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %10.0f  %.4f ms' % ('$LAB $*', r['value'], r['ms_per_step']))"; }
echo "== tests with the hook"
python -m pytest tests/test_gpu_api_fuzz.py tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do
LAB=off-chain-default run --config 4 --rate 48000
LAB=on-chain-old LYRA_HIP_RS_OUT_ON_CHAIN_SPLIT=1 run --config 4 --rate 48000
LAB=off-chain-default run --config 4 --full-decoder --rate 48000
LAB=on-chain-old LYRA_HIP_RS_OUT_ON_CHAIN_SPLIT=1 run --config 4 --full-decoder --rate 48000
done
