for m in 0xf80 0x008 0x070 0x007; do echo "== delay mask $m"; LYRA_HIP_LIB=$PWD/lyra_amd/variants/stress_$m.so timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
for m in 0xf80 0x008; do echo "== delay mask $m, split 2"; LYRA_HIP_SUBBATCHES=2 LYRA_HIP_LIB=$PWD/lyra_amd/variants/stress_$m.so timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
echo "== shipped, split 2 / 4, hw queues 1 / 8"
LYRA_HIP_SUBBATCHES=2 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1
LYRA_HIP_SUBBATCHES=4 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1
GPU_MAX_HW_QUEUES=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1
GPU_MAX_HW_QUEUES=8 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1
