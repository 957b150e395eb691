"""HBM-side traffic of the forward edge stage from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as
MI355X_MICROARCH.md prescribes).  rocprofv3 reports KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so
reads are doubled -- the factor is checked in the same run on k_gelu_dropout<false>, which reads and writes exactly N*DP*4 bytes.

usage: pmc_edge_traffic.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> <N> <DP> > pmc_edge_fwd.json"""
import collections
import csv
import json
import sys


def by_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def pick(table, needle):
    hits = {k: v for k, v in table.items() if needle in k}
    assert len(hits) == 1, (needle, list(hits))
    return next(iter(hits.values()))


fetch, write = by_kernel(sys.argv[1]), by_kernel(sys.argv[2])
N, DP = int(sys.argv[3]), int(sys.argv[4])
names = {'k_edge_scores': 'qagnn::k_edge_scores(', 'k_edge_aggregate': 'qagnn::k_edge_aggregate('}
per = {k: {'FETCH_SIZE': round(pick(fetch, n), 1), 'WRITE_SIZE': round(pick(write, n), 1)} for k, n in names.items()}
bwd_names = {'k_edge_bwd_src1': 'qagnn::k_edge_bwd_src1(', 'k_edge_bwd_src2': 'qagnn::k_edge_bwd_src2(', 'k_edge_bwd_tgt': 'qagnn::k_edge_bwd_tgt(',
             'k_edge_bwd_cls': 'qagnn::k_edge_bwd_cls(', 'k_cls_reduce': 'qagnn::k_cls_reduce('}
per_bwd = {k: {'FETCH_SIZE': round(pick(fetch, n), 1), 'WRITE_SIZE': round(pick(write, n), 1)} for k, n in bwd_names.items()}
cal_f, cal_w = pick(fetch, 'k_gelu_dropout<false>'), pick(write, 'k_gelu_dropout<false>')
exact_kib = N * DP * 4 / 1024.0
total = sum(2 * v['FETCH_SIZE'] + v['WRITE_SIZE'] for v in per.values()) * 1024
total_bwd = sum(2 * v['FETCH_SIZE'] + v['WRITE_SIZE'] for v in per_bwd.values()) * 1024
print(json.dumps({
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` (bench.py's own PMC child run: bench.py --pmc)',
    'units': 'KiB per launch (mean over the launches of the run); FETCH_SIZE is doubled in the totals (gfx950 tallies 128-B reads at 64 B)',
    'calibration': {'kernel': 'k_gelu_dropout<false> (reads and writes N*DP*4 bytes)', 'exact_KiB': round(exact_kib, 1),
                    'FETCH_SIZE_KiB': round(cal_f, 1), 'WRITE_SIZE_KiB': round(cal_w, 1),
                    'fetch_factor': round(exact_kib / cal_f, 3), 'write_factor': round(exact_kib / cal_w, 3)},
    'per_launch_KiB': per,
    'traffic_bytes_per_launch': int(total),
    'backward_per_launch_KiB': per_bwd,
    'backward_traffic_bytes_per_launch': int(total_bwd),
}, indent=1))
