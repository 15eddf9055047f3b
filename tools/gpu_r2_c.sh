#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 )
