#!/bin/bash
# code_sizes.sh obj... -- prints `{"<kernel name>", <machine-code bytes>, <offset of its s_getpc_b64>},` for every gfx950 kernel in the given hipcc
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
  # symbol sizes, and for every kernel the offset of its (first) s_getpc_b64 from the start of the function: code_warm
  # warms [pc & ~127, + size - offset), which then ends inside the function whatever the compiler put before the s_getpc
  $LLVM/llvm-objdump -d $elf 2>/dev/null | python3 -c '
import re, sys
start = name = None
for line in sys.stdin:
    m = re.match(r"^([0-9a-f]+) <(\S*_kernel\S*)>:", line)
    if m:
        start, name = int(m.group(1), 16), m.group(2)
        continue
    if name and "s_getpc_b64" in line:
        a = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        if a:
            print(name, int(a.group(1), 16) - start)
        name = None
' > $elf.pc
  $LLVM/llvm-readelf -s --wide $elf 2>/dev/null | awk '$4 == "FUNC" && $8 ~ /_kernel/ { print $3, $8 }' | sort -u |
    while read size name; do
      short=$(echo $name | sed -E 's/^_ZN4lyra[0-9]+([a-z0-9_]+_kernel)E.*/\1/')
      off=$(awk -v n=$name '$1 == n { print $2 }' $elf.pc | head -1)
      echo "{\"$short\", $size, ${off:-$size}},"
    done
  rm -f $elf.pc
done
rm -f $fat $elf
