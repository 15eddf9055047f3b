#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 )
python bench.py --with-logmel --steps 50 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench+logmel', d['value'], d['ms_per_step'], {k.replace('_kernel',''):v['avg_us'] for k,v in d['kernels'].items()})"
