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

# timeline view: how much of the wall span is covered by kernels, how much is gaps, and what the small kernels cost
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows)
span = (ev[-1][1] - ev[0][0]) / 1e6
busy, cur_end, gaps = 0, ev[0][0], []
for s, e, _ in ev:
    if s > cur_end:
        gaps.append((s - cur_end) / 1e3)
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
small = [(e - s) / 1e3 for s, e, _ in ev if e - s < 10000]
gsm = [g for g in gaps if g < 100]
print(f'\ntimeline: span {span:.2f} ms, busy {busy / 1e6:.2f} ms, {len(gaps)} gaps: {sum(gsm) / 1e3:.2f} ms in {len(gsm)} gaps < 100 us '
      f'(median {sorted(gsm)[len(gsm) // 2] if gsm else 0:.1f} us), {sum(g for g in gaps if g >= 100) / 1e3:.2f} ms in longer ones')
print(f'kernels shorter than 10 us: {len(small)} dispatches, {sum(small) / 1e3:.2f} ms total, mean {sum(small) / max(len(small), 1):.1f} us')
by = collections.Counter()
for s, e, n in ev:
    if e - s < 10000:
        by[n] += 1
for n, c in by.most_common(25):
    print(f'   {c:5d}  {n}')
