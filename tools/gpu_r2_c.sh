#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in WP WPNH WPSW WPBOTH; do
  s=$(LYRA_HIP_LIB=$GRAFT_REPO_ROOT/lyra_amd/variants/$v.so MODES=full python tools/pipeline_probe.py 2>&1 | grep "^full")
  echo "$v | $s"
done
