#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS table of the library's HIP sources.

    python tools/isa_resources.py [file.hip ...] [--flags "-DH2S_NSETS=3"] [--all] > profiles/r6/isa_resources.txt

Compiles each source with the flags tools/build.py uses plus -Rpass-analysis=kernel-resource-usage (hipcc cross-compiles: no GPU
needed) and prints one line per kernel instantiation.  `sgpr_spill` are scalar registers parked in vector-register LANES
(v_writelane / v_readlane: no memory traffic); `vgpr_spill` and `scratch` are real scratch memory.  Dynamic LDS (the persistent
convolution kernels allocate theirs at launch) is not in the compiler's number: see SCfg<BN>::LDS_BYTES and friends."""
import argparse
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import build as B  # noqa: E402

HOT = ['conv_h2s.hip', 'wgrad_h2s.hip', 'gemm_h2s.hip', 'wgrad_h2g.hip', 'thin.hip', 'misc.hip', 'pack_jobs.hip', 'noise.hip']


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
        return [re.sub(r'\(anonymous namespace\)::', '', o) for o in out[:len(names)]]
    except Exception:
        return names


def resources(src, extra):
    cmd = [B.HIPCC] + B.COMMON + B.EXTRA.get(os.path.basename(src), []) + extra + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split('\n'):
        m = re.search(r'remark:\s+(.*?)\s+\[-Rpass-analysis', line)
        if not m:
            continue
        k, _, v = m.group(1).partition(':')
        k, v = k.strip(), v.strip()
        if k == 'Function Name':
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('files', nargs='*')
    ap.add_argument('--flags', default='')
    ap.add_argument('--all', action='store_true', help='every .hip of the library (default: the kernels of a default train step)')
    a = ap.parse_args()
    files = a.files or (sorted(f for f in os.listdir(B.SRC) if f.endswith('.hip')) if a.all else HOT)
    print(f"# {'file':<16} {'kernel':<78} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'sgpr_spill':>10} {'vgpr_spill':>10} {'scratch_B':>9} {'static_lds':>10} {'occ':>3}")
    bad = 0
    for f in files:
        rows = resources(os.path.join(B.SRC, os.path.basename(f)), a.flags.split())
        names = demangle([r['name'] for r in rows])
        for r, n in zip(rows, names):
            n = re.sub(r'\(.*$', '', n).replace('void ', '')
            scratch = int(r.get('ScratchSize [bytes/lane]', 0))
            bad += scratch > 0
            print(f"  {os.path.basename(f):<16} {n[:78]:<78} {r.get('VGPRs', '?'):>4} {r.get('AGPRs', '?'):>4} {r.get('TotalSGPRs', '?'):>4} {r.get('SGPRs Spill', '?'):>10} "
                  f"{r.get('VGPRs Spill', '?'):>10} {scratch:>9} {r.get('LDS Size [bytes/block]', '?'):>10} {r.get('Occupancy [waves/SIMD]', '?'):>3}")
    print(f'# kernels with scratch memory: {bad}')


if __name__ == '__main__':
    main()
