"""rocprofv3 counter_collection.csv -> one line per (kernel, grid) with the SQ counters side by side (means per dispatch) and the
ratios that say where a wave's time goes (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, quad-cycles)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    key = (r['Kernel_Name'][:70], r.get('Grid_Size', ''))
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0]))):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0:
        continue
    n = len(next(iter(c.values())))
    parts = ' '.join(f'{k[3:]}={m[k] / wc:5.3f}' for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU',
                                                        'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_MISC') if k in m)
    extra = ' '.join(f'{k}={m[k]:.3g}' for k in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS',
                                                 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE') if k in m)
    print(f'n={n:4d} grid={key[1]:>9s} WAVE_CYCLES={wc:12.4g} (of it: {parts})  {extra}  {key[0]}')
