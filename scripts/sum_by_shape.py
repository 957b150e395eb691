"""sum the per-step kernel time of kernel families in a trace_by_shape.py table: python scripts/sum_by_shape.py <by_shape.txt> <steps>"""
import re, sys
fam = [('gemm NN', r'k_gemm_nn'), ('gemm TN', r'k_gemm_tn|k_sum_chunks'), ('pack', r'k_pack_b'), ('edge fwd', r'k_edge_scores|k_edge_aggregate'),
       ('edge bwd', r'k_edge_bwd|k_cls_reduce'), ('gelu', r'k_gelu_dropout'), ('amax', r'k_absmax|k_zero_words'), ('bn/colreduce', r'k_bn_|k_colreduce'),
       ('graph prep', r'k_blob|k_scan|k_cls_scatter|k_chunk|k_xcd|k_zero16|k_decode|k_fill|k_rank|k_hist'), ('pool/head', r'k_pool|k_head|k_add_row0'),
       ('gather/optim', r'k_gather_multi|k_radam|k_node_prep|k_sin_basis'), ('rocblas', r'Cijk_'), ('torch', r'at::|rocprim|rocclr|elementwise_kernel')]
steps = float(sys.argv[2])
tot = {k: 0.0 for k, _ in fam}
cnt = {k: 0 for k, _ in fam}
other = 0.0
for line in open(sys.argv[1]):
    m = re.match(r'\s+([\d.]+) ms\s+[\d.]+%\s+n=\s*(\d+)\s+mean=\s*([\d.]+)us', line)
    if not m:
        continue
    if float(m.group(3)) > 900:  # one-off set-up kernels
        continue
    ms, n = float(m.group(1)), int(m.group(2))
    for k, pat in fam:
        if re.search(pat, line):
            tot[k] += ms; cnt[k] += n
            break
    else:
        other += ms
for k, _ in fam:
    print(f'{k:14s} {tot[k] / steps * 1000:8.1f} us/step  {cnt[k] / steps:6.1f} launches/step')
print(f'{"other":14s} {other / steps * 1000:8.1f} us/step')
print(f'{"sum":14s} {(sum(tot.values()) + other) / steps * 1000:8.1f} us/step')
