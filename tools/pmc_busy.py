#!/usr/bin/env python3
"""gpurun_out/pmc_layers_<tag>.csv (tools/pmc_layers.sh, counters SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...) -> matrix-pipe busy fraction per
kernel instantiation, time-weighted:  busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)  (GRBM_GUI_ACTIVE sums the 8 XCDs)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r['kernel'].split('::')[-1].split('(')[0]
    a = agg.setdefault(k, collections.defaultdict(float))
    a['n'] += 1
    for c, v in r.items():
        if c not in ('dispatch', 'kernel', 'grid', 'lds') and v != '':
            a[c] += float(v)
print(f'{"kernel":44s} {"n":>4s} {"busy":>6s} {"coexec":>7s} {"wait_any":>9s} {"lds_conf":>9s}')
for k, a in agg.items():
    simd_cycles = a['GRBM_GUI_ACTIVE'] / 8 * 1024
    busy = a['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles if simd_cycles else float('nan')
    co = a.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0.0) / simd_cycles if simd_cycles else float('nan')
    wait = a.get('SQ_WAIT_ANY', 0.0) / a['SQ_WAVE_CYCLES'] if a.get('SQ_WAVE_CYCLES') else float('nan')
    ldsc = a.get('SQ_LDS_BANK_CONFLICT', 0.0) / a['SQ_LDS_IDX_ACTIVE'] if a.get('SQ_LDS_IDX_ACTIVE') else float('nan')
    print(f'{k:44s} {int(a["n"]):4d} {busy:6.3f} {co:7.3f} {wait:9.3f} {ldsc:9.3f}')
