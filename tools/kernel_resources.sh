#!/bin/bash
# kernel_resources.sh [obj...] -- VGPR / SGPR / spill / LDS / code size of every gfx950 kernel in the hipcc objects
# (default: lyra_amd/csrc/*.o), from the code object's metadata notes.
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
cd "$(dirname "$0")/../lyra_amd/csrc" || exit 1
objs=${@:-enc_kernels.o enc_s2_kernel.o dec_kernels.o misc_kernels.o}
fat=$(mktemp); elf=$(mktemp)
for o in $objs; do
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$fat $o 2>/dev/null || continue
  tgt=$($LLVM/clang-offload-bundler --list --type=o --input=$fat 2>/dev/null | grep gfx950 | head -1)
  $LLVM/clang-offload-bundler --type=o --targets=$tgt --input=$fat --output=$elf --unbundle 2>/dev/null || continue
  $LLVM/llvm-readelf --notes $elf 2>/dev/null | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
    name = re.sub(r"^_ZN4lyra\d+([a-z0-9_]+_kernel)E.*", r"\1", g("name"))
    print("%-22s vgpr %3s  sgpr %3s  spill(v) %2s  scratch %4s B  lds(static) %6s" % (name, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
  $LLVM/llvm-readelf -s --wide $elf 2>/dev/null | awk '$4 == "FUNC" && $8 ~ /_kernel/ { print $3, $8 }' | sort -u | sed -E 's/_ZN4lyra[0-9]+([a-z0-9_]+_kernel)E.*/\1/' | awk '{printf "    code %6d B  %s\n", $1, $2}'
done
rm -f $fat $elf
