#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cpp_plugins.py -x -q 2>&1 | tail -4
for i in 1 2; do
echo "== zero-copy"; timeout 300 lyra_amd/plugin_demo --bench lyra_amd/assets 2000 120 2>&1
echo "== copy engine (LYRA_HIP_NO_ZEROCOPY=1)"; LYRA_HIP_NO_ZEROCOPY=1 timeout 300 lyra_amd/plugin_demo --bench lyra_amd/assets 2000 120 2>&1
done | tee gpurun_out/r04/plugin_bench.txt
