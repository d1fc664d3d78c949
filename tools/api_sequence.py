#!/usr/bin/env python3
"""Steady-state HIP API sequence of the last train steps of a rocprofv3 --hip-trace run of bench.py: per step, how many kernel
launches, memcpys and memsets the host issues, and which non-pnnp (at::native / rocclr) kernels ran.  The whole-process kernel
statistics also count the one-time uploads of ~330 parameter tensors at start-up, which is not what a step costs.
    python tools/api_sequence.py <hip_api_trace.csv> <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [r['Function'] for r in rows if r['Function'] in ('hipMemcpyWithStream', 'hipLaunchKernel', 'hipMemcpyAsync', 'hipMemsetAsync',
                                                         'hipDeviceSynchronize', 'hipStreamSynchronize', 'hipEventSynchronize', 'hipModuleLaunchKernel')]
# the timed region of bench.py ends with hipDeviceSynchronize; take the events between the last two synchronisations that have > 100 launches between them
idx = [i for i, n in enumerate(ev) if n == 'hipDeviceSynchronize']
segs = [(a, b) for a, b in zip(idx, idx[1:]) if sum(1 for n in ev[a:b] if 'Launch' in n) > 100]
a, b = segs[-1]
seq = ev[a + 1:b]
c = collections.Counter(seq)
out, prev, cnt = [], None, 0
for n in seq:
    if n == prev:
        cnt += 1
    else:
        if prev:
            out.append(f'{prev} x{cnt}')
        prev, cnt = n, 1
out.append(f'{prev} x{cnt}')
print('events between the last two device synchronisations (the timed steps):')
print('  ' + ', '.join(out))
print('totals:', dict(c))
# kernels that STARTED between those two synchronisations (same clock as the API trace)
ia, ib = segs[-1]
names_in_seq = [r for r in rows if r['Function'] in ('hipMemcpyWithStream', 'hipLaunchKernel', 'hipMemcpyAsync', 'hipMemsetAsync', 'hipDeviceSynchronize',
                                                     'hipStreamSynchronize', 'hipEventSynchronize', 'hipModuleLaunchKernel')]
t0, t1 = int(names_in_seq[ia]['End_Timestamp']), int(names_in_seq[ib]['Start_Timestamp'])
k = [r for r in csv.DictReader(open(sys.argv[2])) if t0 <= int(r['Start_Timestamp']) <= t1 + 50_000_000]
names = collections.Counter(r['Kernel_Name'][:100] for r in k)
foreign = {n: v for n, v in names.items() if 'at::' in n or 'rocclr' in n}
print(f'{len(k)} kernels ran in that window; non-pnnp kernels among them:', foreign if foreign else 'none')
