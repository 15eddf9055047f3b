#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
VARIANTS="prev cur" bash tools/ab_variants.sh
