"""Aggregate a rocprofv3 kernel_trace.csv by (kernel name, grid size): count, mean/min duration."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'][:70]
    grid = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
    agg[(name, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f'total kernel time {tot / 1e3:.2f} ms over {len(rows)} dispatches')
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:70]:
    print(f'{sum(v) / 1e3:8.2f} ms {100 * sum(v) / tot:5.1f}%  n={len(v):5d} mean={sum(v) / len(v):8.1f}us min={min(v):8.1f}  grid={grid}  {name}')
