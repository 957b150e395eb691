"""rocprofv3 counter_collection.csv -> one row per (kernel, grid, workgroup): mean of every counter per dispatch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    key = (r['Kernel_Name'].split('(')[0][-46:], r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?'))
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
ctrs = sorted({c for v in agg.values() for c in v})
print('%-46s %9s %5s %5s ' % ('kernel', 'grid', 'wg', 'n') + ' '.join('%14s' % c[-14:] for c in ctrs))
for key, v in sorted(agg.items()):
    n = max(len(x) for x in v.values())
    print('%-46s %9s %5s %5d ' % (key + (n,)) + ' '.join('%14.0f' % (sum(v[c]) / max(len(v[c]), 1)) for c in ctrs))
