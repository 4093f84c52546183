#!/bin/bash
# Per-kernel register / spill / LDS figures of libcgsvmc_hip.so (or any object with gfx950 code):
# extracts the embedded code objects and prints name, VGPRs, AGPRs, spilled VGPRs, scratch, LDS.
# Usage: tools/kernel_resources.sh [file] [name filter regex]
set -euo pipefail
F=${1:-$(dirname "$0")/../cgs_vmc_amd/libcgsvmc_hip.so}
PAT=${2:-.}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
LLVM=/opt/rocm/lib/llvm/bin
# roc-obj-ls/extract are not everywhere: carve the ELF code objects out of the fat binary section
python3 - "$F" "$TMP" <<'PY'
import sys
data = open(sys.argv[1], 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
i, n = 0, 0
while True:
  i = data.find(magic, i)
  if i < 0:
    break
  import struct
  cnt = struct.unpack_from('<Q', data, i + 24)[0]
  off = i + 32
  for _ in range(cnt):
    o, s, tl = struct.unpack_from('<QQQ', data, off)
    triple = data[off + 24:off + 24 + tl].decode()
    off += 24 + tl
    if 'gfx950' in triple and s > 0:
      open('%s/co_%d.elf' % (sys.argv[2], n), 'wb').write(data[i + o:i + o + s])
      n += 1
  i += 24
PY
for co in "$TMP"/co_*.elf; do
  $LLVM/llvm-readelf --notes "$co" | awk '
    /\.name:/ {name=$2}
    /\.vgpr_count:/ {v=$2}
    /\.agpr_count:/ {a=$2}
    /\.vgpr_spill_count:/ {sp=$2}
    /\.private_segment_fixed_size:/ {ps=$2}
    /\.group_segment_fixed_size:/ {g=$2}
    /\.symbol:/ {printf "%s vgpr=%s agpr=%s spill=%s scratch=%s lds=%s\n", name, v, a, sp, ps, g}'
done | c++filt | grep -E "$PAT" | sort -u
