"""Register / LDS / scratch usage per kernel from a -save-temps gfx950 .s file (amdhsa.kernels metadata)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
rows = []
for blk in txt.split('  - .agpr_count:')[1:]:
    get = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk)  # noqa: E731
    name = get('name').group(1)
    if pat and not re.search(pat, name):
        continue
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    rows.append((dem.split('(')[0][:60], blk.split('\n')[0].strip(), get('vgpr_count').group(1), get('sgpr_count').group(1),
                 get('vgpr_spill_count').group(1), get('private_segment_fixed_size').group(1), get('group_segment_fixed_size').group(1)))
print('%-60s %5s %5s %5s %6s %8s %6s' % ('kernel', 'agpr', 'vgpr', 'sgpr', 'spill', 'scratch', 'lds'))
for r in rows:
    print('%-60s %5s %5s %5s %6s %8s %6s' % r)
