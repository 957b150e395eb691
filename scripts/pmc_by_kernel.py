"""Aggregate a rocprofv3 counter_collection.csv by kernel: mean counter value per dispatch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:60]:
    print(f'{c:12s} n={len(v):5d} mean={sum(v) / len(v):14.1f} total={sum(v):16.1f}  {k}')
