#!/bin/bash
# code_sizes.sh obj... -- prints `{"<kernel name>", <machine-code bytes>},` for every gfx950 kernel in the given hipcc
# objects (function symbol sizes in the device code object embedded in .hip_fatbin).  api.hip uses the table to tell
# each stage kernel how much of its own instruction stream to pull into L2 at start (lyra_dev.h code_warm).  If the
# LLVM tools are missing the table is empty and the kernels simply skip that step.
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
fat=$(mktemp); elf=$(mktemp)
for o in "$@"; do
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$fat $o 2>/dev/null || continue
  tgt=$($LLVM/clang-offload-bundler --list --type=o --input=$fat 2>/dev/null | grep gfx950 | head -1)
  [ -z "$tgt" ] && continue
  $LLVM/clang-offload-bundler --type=o --targets=$tgt --input=$fat --output=$elf --unbundle 2>/dev/null || continue
  $LLVM/llvm-readelf -s --wide $elf 2>/dev/null | awk '$4 == "FUNC" && $8 ~ /_kernel/ { print $3, $8 }' | sort -u |
    while read size name; do
      short=$(echo $name | sed -E 's/^_ZN4lyra[0-9]+([a-z0-9_]+_kernel)E.*/\1/')
      echo "{\"$short\", $size},"
    done
done
rm -f $fat $elf
