#!/bin/bash
# Register / LDS / scratch use of every kernel in libwfk_hip.so (from the code object's metadata):
#   tools/kernel_regs.sh [pattern]
set -e
cd "$(dirname "$0")/../waveforms_amd/csrc"
tmp=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
# every translation unit carries its own fat binary: walk the object files (WFK_OBJ: one of them)
for o in ${WFK_OBJ:-_obj/*.o}; do
  $B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $o 2>/dev/null || continue
  $B/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.co 2>/dev/null || continue
$B/llvm-readelf --notes $tmp/k.co | python3 -c "
import sys, re, subprocess
pat = sys.argv[1] if len(sys.argv) > 1 else ''
keys = ('name', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'group_segment_fixed_size', 'private_segment_fixed_size', 'agpr_count')
cur, rows = {}, []
for line in sys.stdin:
    m = re.match(r'(\s*)(-?)\s*\.(\w+):\s*(.*)', line)
    if not m: continue
    k, v = m.group(3), m.group(4).strip()
    if m.group(2) == '-' and len(m.group(1)) <= 4 and k != 'address_space' and k != 'offset' and k != 'name' or (m.group(2) == '-' and k == 'agpr_count'):
        if 'vgpr_count' in cur: rows.append(cur)
        cur = {}
    if k in keys and k not in cur: cur[k] = v
if 'vgpr_count' in cur: rows.append(cur)
for r in rows:
    n = r.get('name', '')
    n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
    if pat in n:
        print('%-100s vgpr %3s agpr %3s sgpr %3s spill %s/%s lds %6s scratch %s' % (n[:100], r.get('vgpr_count'), r.get('agpr_count'), r.get('sgpr_count'), r.get('vgpr_spill_count'), r.get('sgpr_spill_count'), r.get('group_segment_fixed_size'), r.get('private_segment_fixed_size')))
" "$1"
done
rm -rf $tmp
