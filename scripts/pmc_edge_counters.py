"""rocprofv3 counter_collection.csv files (one per pass) -> one block per edge kernel: mean counter values per dispatch and the ratios
that say what bounds the kernel (texture-address unit busy, L1 / L2 hit rates, issue stalls).  Usage: pmc_edge_counters.py a.csv b.csv ..."""
import collections
import csv
import sys

KERNELS = ('k_edge_scores', 'k_edge_aggregate', 'k_edge_bwd_src1', 'k_edge_bwd_src2', 'k_edge_bwd_tgt', 'k_edge_bwd_cls', 'k_cls_reduce(')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r['Kernel_Name']
            for k in KERNELS:
                if k in name:
                    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in KERNELS:
    if k not in agg:
        continue
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    print(f'== {k.rstrip("(")}  ({len(next(iter(agg[k].values())))} dispatches)')
    print('   ' + '  '.join(f'{c}={v:.4g}' for c, v in sorted(m.items())))
    g = m.get('GRBM_GUI_ACTIVE')
    g = g / 8.0 if g else g  # rocprofv3 sums the counter over the 8 XCDs: cycles of ONE XCD's clock
    out = []
    if g and 'TA_BUSY_avr' in m:
        out.append(f'TA busy (average over the TAs) = {m["TA_BUSY_avr"] / g:.2f} of the kernel\'s cycles')
    if g and 'TA_BUSY_max' in m:
        out.append(f'busiest TA = {m["TA_BUSY_max"] / g:.2f}')
    if g and 'TD_TD_BUSY_sum' in m:
        out.append(f'TD busy (sum / 256 CUs) = {m["TD_TD_BUSY_sum"] / 256 / g:.2f}')
    if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m:
        out.append(f'L2 hit rate = {m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1):.3f}')
    if 'TCP_TOTAL_CACHE_ACCESSES_sum' in m and 'TCP_TCC_READ_REQ_sum' in m:
        out.append(f'L1 (TCP): {m["TCP_TCC_READ_REQ_sum"] / max(m["TCP_TOTAL_CACHE_ACCESSES_sum"], 1):.3f} of the cache accesses go on to L2')
    if 'SQ_WAVE_CYCLES' in m:
        wc = m['SQ_WAVE_CYCLES']
        out.append('wave time: ' + ' '.join(f'{c[3:]}={m[c] / wc:.2f}' for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU',
                                                                                   'SQ_INST_CYCLES_VMEM_RD', 'SQ_ACTIVE_INST_LDS') if c in m))
    if g and 'SQ_INSTS_VMEM_RD' in m:
        out.append(f'{m["SQ_INSTS_VMEM_RD"] / 256 / g:.4f} vector-memory read instructions per CU and cycle')
    for line in out:
        print('   -> ' + line)
