#!/usr/bin/env python3
"""Build libpnnp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, 'pnnp_amd', 'csrc')
OBJ = os.path.join(SRC, '_build')
OUT = os.path.join(REPO, 'pnnp_amd', 'libpnnp_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
          '-I', os.path.join(REPO, 'include')] + os.environ.get('PNNP_HIPCC_EXTRA', '').split()     # A/B experiments (-DWINO_FENCED=0 ...)
# per-file extra flags: the sampler keeps every float32 rounding explicit (matches oracle/pnnp_oracle.c)
EXTRA = {'wino.hip': ['-fno-slp-vectorize'], 'conv_igemm.hip': ['-fno-slp-vectorize'], 'conv_x3.hip': ['-fno-slp-vectorize'], 'conv_x3s.hip': ['-fno-slp-vectorize'], 'conv_h2s.hip': ['-fno-slp-vectorize'], 'wgrad_x3.hip': ['-fno-slp-vectorize'], 'wgrad_x3s.hip': ['-fno-slp-vectorize'], 'wgrad_x3g.hip': ['-fno-slp-vectorize'], 'gemm_x3.hip': ['-fno-slp-vectorize'], 'gemm_x3s.hip': ['-fno-slp-vectorize'], 'gemm_h2s.hip': ['-fno-slp-vectorize'], 'wgrad_h2g.hip': ['-fno-slp-vectorize'],      # measured: +0.5 % / +0.7 % on the step each
         # wgrad_h2s: with the SLP vectoriser on, the producers' rolling refill ends in register copies that wait for the loads it has just issued: step -2.1 % where
         # the same source without it gains 1.05 % (profiles/r6/ab_wgrad_rolling.txt, part 4); round 5's source was indifferent to the flag
         'wgrad_h2s.hip': ['-fno-slp-vectorize'],
         'noise.hip': ['-ffp-contract=off'], 'pack.hip': ['-ffp-contract=off'], 'cropaug.hip': ['-ffp-contract=off'],
         # the Winograd backward-weight kernel's source order is its schedule (slots fenced with sched_barrier)
         # (-fno-slp-vectorize: its stride-2 transforms vectorise into packed ops fed by 84 register moves per chunk; scalar is 44 fewer)
         'wino_wgrad.hip': ['-mllvm', '-pre-RA-sched=source'] + os.environ.get('PNNP_WW_FLAGS', '-fno-slp-vectorize').split()}


def newer(a, deps):
    return not os.path.exists(a) or any(os.path.getmtime(d) > os.path.getmtime(a) for d in deps)


def flags_of(cmd):
    """A compile command as the stamp keeps it: independent of where the repository lies (the GPU box sees it under another path and must not rebuild)."""
    return ' '.join(a.replace(REPO, '.') for a in cmd[1:])


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(SRC) if f.endswith('.hip'))
    hdrs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith('.h')] + [os.path.join(REPO, 'include', 'pnnp_hip.h')]
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, s[:-4] + '.o')
        cmd = [HIPCC] + COMMON + EXTRA.get(s, []) + ['-c', os.path.join(SRC, s), '-o', o]
        # an object is stale when its source or a header is newer -- or when it was compiled with OTHER FLAGS (kept beside it in <object>.flags)
        stamp = o + '.flags'
        same_flags = os.path.exists(stamp) and open(stamp).read() == flags_of(cmd)
        if force or not same_flags or newer(o, [os.path.join(SRC, s)] + hdrs):
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        if '-c' in cmd:
            with open(cmd[-1] + '.flags', 'w') as f:
                f.write(flags_of(cmd))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s[:-4] + '.o') for s in srcs]
    if force or jobs or newer(OUT, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs)
    # test / tool helper (not part of the product): the kernel that occupies k CUs beside a persistent grid (tools/squat_test.py,
    # tests/test_gpu_overlap.py) -- built HERE so that no test has to spawn hipcc from a process that has initialised the GPU
    sq_src, sq_so = os.path.join(REPO, 'tools', 'ubench', 'squat.hip'), os.path.join(REPO, 'tools', 'ubench', 'libsquat.so')
    if os.path.exists(sq_src) and (force or newer(sq_so, [sq_src])):
        run([HIPCC, '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', sq_so, sq_src])
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print('built', OUT)
