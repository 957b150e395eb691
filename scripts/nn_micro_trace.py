"""Per-shape kernel durations of tools/nn_micro.py --small from a rocprofv3 kernel trace: launches in time order, 55 per shape (5 warm-up
+ 50 timed), mean of the last 50."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_gemm_nn_split' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
if '--big' in sys.argv:   # tools/nn_micro.py without --small: M = 64 000, all eight shapes
    shapes = ['mlp 208->208', 'proj 320->624', 'dX 624->208', '640->208', '1024->208', '416->208', 'dS 624->112', '[dX|dS] 624->320']
    Ms = [64000]
else:
    shapes = ['mlp 208->208', 'proj 320->624', 'dX 624->208', 'dS 624->112']
    Ms = [500, 2000, 12800]
i = 0
for M in Ms:
    for sh in shapes:
        chunk = rows[i:i + 55]
        i += 55
        if len(chunk) < 55:
            break
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in chunk[5:]]
        name = chunk[-1]['Kernel_Name']
        nt = name[name.index('<') + 1:name.index(',')]
        print(f'M={M:6d} {sh:16s} NT={nt:>2s} grid={chunk[-1].get("Grid_Size_X") or chunk[-1].get("Grid_Size")}  mean {sum(d) / len(d):7.1f} us  min {min(d):7.1f}')
