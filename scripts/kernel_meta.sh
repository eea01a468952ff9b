#!/bin/bash
# kernel_meta.sh -- registers, spills, scratch, LDS of every kernel in the shipped libraries, from the code objects' metadata,
# and the count of s_nop / scratch / packed instructions in the disassembly of the kernels whose name holds the argument.
#   bash scripts/kernel_meta.sh [kernel-name-substring-to-disassemble]
set -e
cd "$(dirname "$0")/.."
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for lib in mgl_amd/libmgl_sw_hip.so mgl_amd/libmgl_pairhmm_hip.so; do
  echo "== $lib"
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin $lib $T/fat.bin
  # one bundle per object file, laid end to end: split at the magic
  python3 - $T <<'PY'
import sys, os
t = sys.argv[1]; d = open(os.path.join(t, "fat.bin"), "rb").read(); m = b"__CLANG_OFFLOAD_BUNDLE__"
pos = []; i = d.find(m)
while i >= 0:
    pos.append(i); i = d.find(m, i + 1)
for k, p in enumerate(pos):
    open(os.path.join(t, f"b{k}.bin"), "wb").write(d[p:pos[k + 1] if k + 1 < len(pos) else len(d)])
PY
  for b in $T/b*.bin; do
    $B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$b --output=$b.co
    $B/llvm-readelf --notes $b.co | awk '
      /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2}
      /\.sgpr_count:/ {sg=$2} /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {sp=$2; printf "%-72s vgpr %3d  spilled %3d  scratch %4d B  lds %6d B  sgpr %3d\n", name, vg, sp, scr, lds, sg}'
    if [ -n "$1" ]; then
      $B/llvm-objdump -d $b.co | awk -v pat="$1" '/^[0-9a-f]+ <.*>:/ {if (n) flush_(); on = index($0, pat) > 0; if (on) name = $2}
        function flush_() {printf "   %s %d instructions, %d s_nop, %d scratch accesses, %d v_pk_*, %d flat_*\n", name, n, nop, scr, pk, fl; n = nop = scr = pk = fl = 0}
        on && /^\t/ {n++; if ($0 ~ /s_nop/) nop++; if ($0 ~ /scratch_/) scr++; if ($0 ~ /v_pk_/) pk++; if ($0 ~ /flat_/) fl++}
        END {if (n) flush_()}'
    fi
  done
  rm -f $T/b*.bin $T/b*.co
done
rm -rf $T
