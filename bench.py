#!/usr/bin/env python3
"""Benchmark of the PNNP hot path on MI355X: one TRAIN STEP of BASELINE.json's config
"PNNP noise-proxy + UNet train step, batch 16x512x512x4":

    per crop: sample_params_max (host scalars) -> physics noise sampler (HIP) -> UNetSeeInDark
    nf=32 forward -> L1(pred.clamp(0,1), hr) -> backward -> [RCCL all-reduce] -> Adam

Metric: packed raw crops (4x512x512 fp32) per second, whole job (all ranks).  Inputs are
synthetic clean crops already resident in HBM; weights are random-init (initialize_weights).
Weak scaling: every rank processes --batch crops per step.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (schema in the task contract) with `roofline` (dominant kernel
class = the 3x3 implicit-GEMM convolutions, fp32 MFMA, timed with HIP events on the launch
stream inside the timed region) and `cpu_baseline` (the torch-fp32 oracle port of the same
step, on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_CROP_TRAIN = {'unet': 289.70, 'resunet': 375.07}   # fwd + dgrad + wgrad, nf=32 @4x512x512
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6     # MI355X_MICROARCH.md (~2.5 PF dense): v_mfma_f32_32x32x16_bf16, 256 CUs x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
# Kernel families of the 3x3 forward / backward-data convolutions: FLOPs the matrix pipe EXECUTES per algorithmic FLOP, its
# dense peak, and what the kernel is.  `roofline.frac` = executed / peak (a hardware fraction, always <= 1).
FAMILY = {
    'x3':     (6.0, PEAK_BF16_MFMA_TFLOPS, ('conv9_fwd_x3', 'conv9_dgrad_x3'),
               "igemm_x3s_kernel (csrc/conv_x3s.hip: conv3x3 forward + backward-data, float32 operands split into three bf16 pieces: the six piece products of a "
               "block as three v_mfma_f32_16x16x32_bf16 with two pieces concatenated along K, i.e. 6 executed bf16 FLOP per algorithmic FLOP, fp32 accumulation; "
               "8 MFMA-only consumer waves + 4 producer waves per workgroup)"),
    'h2':     (28.0 / 9.0, PEAK_BF16_MFMA_TFLOPS, ('conv9_fwd_h2', 'conv9_dgrad_h2'),
               "igemm_h2s_kernel (csrc/conv_h2s.hip: conv3x3 forward + backward-data, float32 operands scaled per tensor and split into two fp16 pieces: "
               "per 16-channel chunk and 16x16 block 14 v_mfma_f32_16x16x32_f16 for the nine taps (hi hi' + lo hi' per tap, hi lo' for two taps at a time), "
               "i.e. 28/9 = 3.11 executed fp16 FLOP per algorithmic FLOP (the fp16 dense peak equals the bf16 one), fp32 accumulation; "
               "8 MFMA-only consumer waves + 4 producer waves per workgroup)"),
    'wino':   (16.0 / 36.0, PEAK_F32_MFMA_TFLOPS, ('conv9_fwd_wino', 'conv9_dgrad_wino'),
               "wino_kernel (conv3x3 forward + backward-data as Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32: 16 multiply-adds where the direct form has 36)"),
    'direct': (1.0, PEAK_F32_MFMA_TFLOPS, ('conv9_fwd', 'conv9_dgrad'),
               "igemm_kernel<9,...> (conv3x3 forward + backward-data, v_mfma_f32_32x32x2_f32)"),
}


def physical_cores():
    """Physical cores of this host (unique (package, core) pairs of /proc/cpuinfo; psutil as a second opinion)."""
    try:
        pairs, phys, core = set(), None, None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return n
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(H, W, budget_s=240.0):
    """The reference's train step restated on torch-CPU fp32 (oracle/net_torch.py + oracle/noise_np.py), timed on this
    box's host cores by the protocol of BASELINE.md section 3: per crop `sample_params_max('SonyA7S2')` ->
    `generate_noisy_torch('pr', clip=2)` -> clamp -> UNetSeeInDark nf=32 forward -> L1(clamp) -> backward -> Adam(lr 1e-4),
    `np.random.seed(1997 + step)`, `torch.manual_seed(1997)`, 2 warm-up + 5 timed steps per leg:
      * all PHYSICAL cores (torch.set_num_threads(physical_cores())) on B = 16 crops of 4 x H x W   -> `value`
      * one thread (the reference exports OMP_NUM_THREADS=1, utils/utils.py:2) on B = 1 crop        -> `one_thread`
    with the four buckets the reference's tqdm line shows (trainer_SID.py:81-123): dataloader (here: synthetic clean crops),
    preprocess (parameters + sampler + clamp), net (forward), bp (loss + backward + Adam).
    Bounded by `budget_s` seconds (--cpu-baseline-seconds; the one-thread leg runs first within 1/4 of it, the all-cores leg gets
    the rest): a leg that runs out of budget stops after the step in progress -- warm-ups first, at least one timed step -- and
    the record says how many warm-up / timed steps it actually did (`protocol_complete` false), instead of silently shrinking
    the sample.  On the pool's 128-core hosts one all-cores step of 16 crops takes ~23 s (torch-CPU convolutions scale poorly
    past ~32 threads: round 2 measured 1.39 crops/s with 32 threads on 4 crops), so the complete protocol needs
    ~175 s; the default budget is 240 s (round 4: the driver's own line must say `protocol_complete: true`)."""
    import numpy as np
    import torch
    from oracle import net_torch as O, noise_np as N
    from pnnp_amd import process as P                      # host-side parameter tables / draws (no device code)
    ncpu = os.cpu_count() or 1
    all_cores = physical_cores()

    def leg(threads, batch, warm, timed, budget):
        torch.set_num_threads(threads)
        torch.manual_seed(1997)
        sd = O.init_state(O.unet_param_shapes(nf=32), seed=0)
        m = {k: torch.zeros_like(v) for k, v in sd.items()}
        v = {k: torch.zeros_like(vv) for k, vv in sd.items()}
        buckets = dict(dataloader=0.0, preprocess=0.0, net=0.0, bp=0.0)
        total, done_warm, done_timed = 0.0, 0, 0
        t_leg = time.perf_counter()
        step = 0
        while True:
            left = budget - (time.perf_counter() - t_leg)
            if done_warm < warm and (left > 0 or done_warm == 0):
                is_warm = True                              # warm-ups first; at least one even when the budget is tiny
            elif done_timed < timed and (left > 0 or done_timed == 0):
                is_warm = False                             # then timed steps while the budget lasts; at least one
            else:
                break
            step += 1
            np.random.seed(1997 + step)
            t = [time.perf_counter()]
            hr = torch.rand(batch, 4, H, W)
            t.append(time.perf_counter())
            noisy = []
            for i in range(batch):                          # trainer_SID.py:451-462, per crop
                p = P.sample_params_max(camera_type='SonyA7S2', ratio=None)
                pt = {k: torch.from_numpy(np.array(val, np.float32)) for k, val in p.items()}
                noisy.append(N.generate_noisy_torch(hr[i], noise_code='pr', param=pt, ori=False, clip=2))
            lr_in = torch.stack(noisy).clamp(max=1.0)       # :481-485 with clip == HALF_CLIP
            tgt = hr.clamp(0, 1)
            t.append(time.perf_counter())
            leaves = {k: w.detach().clone().requires_grad_(True) for k, w in sd.items()}
            pred = O.unet_forward(leaves, lr_in)
            t.append(time.perf_counter())
            loss = O.l1_clamp_loss(pred, tgt)
            loss.backward()
            with torch.no_grad():
                O.adam_step(sd, {k: w.grad for k, w in leaves.items()}, m, v, step, lr=1e-4)
            t.append(time.perf_counter())
            if is_warm:
                done_warm += 1
                continue
            done_timed += 1
            for k, a, b in (('dataloader', 0, 1), ('preprocess', 1, 2), ('net', 2, 3), ('bp', 3, 4)):
                buckets[k] += t[b] - t[a]
            total += t[4] - t[0]
        return dict(value=batch * done_timed / total, timed_s=total, warmup_steps=done_warm, timed_steps=done_timed, batch=batch, threads=threads,
                    split={k: round(val / done_timed, 3) for k, val in buckets.items()})

    t_all = time.perf_counter()
    O1 = leg(1, 1, 2, 5, 0.25 * budget_s)                  # the cheap leg first (~12 s) ...
    A = leg(all_cores, 16, 2, 5, budget_s - (time.perf_counter() - t_all))       # ... the rest of the budget for the all-cores leg
    torch.set_num_threads(all_cores)
    # the dataloader-side sampler (generate_noisy_obs: numpy, one core, as a DataLoader worker runs it), one crop per code
    obs = {}
    y = np.random.rand(4, H, W).astype(np.float32)
    for code in ('pr', 'pgrq'):
        np.random.seed(1997)
        pn = P.sample_params_max(camera_type='SonyA7S2', ratio=None)
        pn['bias'] = np.zeros(4)
        N.generate_noisy_obs(y[:, :8, :8], param=pn, noise_code=code, ori=False, clip=False)       # un-timed: first-call imports
        t0 = time.perf_counter()
        N.generate_noisy_obs(y, param=pn, noise_code=code, ori=False, clip=False)
        obs[code] = round(time.perf_counter() - t0, 3)
    complete = (A['warmup_steps'], A['timed_steps'], O1['warmup_steps'], O1['timed_steps']) == (2, 5, 2, 5)
    return {"value": A['value'], "unit": "crops/s", "cores": all_cores, "kind": "port",
            "sample": (f"UNet nf=32 train step (sample_params_max + generate_noisy_torch 'pr' clip=2 + fwd/L1/bwd/Adam), BASELINE.md section 3 protocol "
                       f"(2 warm-up + 5 timed steps) under a {budget_s:.0f} s budget: {all_cores} threads (all physical cores) on 16 crops of 4x{H}x{W}: "
                       f"{A['warmup_steps']} warm-up + {A['timed_steps']} timed steps, {A['timed_s']:.1f} s timed; 1 thread on 1 crop: "
                       f"{O1['warmup_steps']} + {O1['timed_steps']} steps, {O1['timed_s']:.1f} s timed; "
                       f"torch {torch.__version__} CPU fp32, host has {ncpu} logical CPUs / {all_cores} physical cores"),
            "protocol_complete": complete, "budget_s": budget_s,
            "warmup_steps": A['warmup_steps'], "timed_steps": A['timed_steps'], "batch": 16,
            "split_s_per_step": A['split'],
            "one_thread": {"value": O1['value'], "unit": "crops/s", "cores": 1, "batch": 1, "warmup_steps": O1['warmup_steps'],
                           "timed_steps": O1['timed_steps'], "split_s_per_step": O1['split']},
            "generate_noisy_obs_1crop_1core_s": obs}


PEAK_HBM_TBS = 8.0                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def probe_rates():
    """What a bare register-only loop of v_mfma_f32_16x16x32_f16 sustains on this chip (tools/ubench/h2_probe.hip part (1), random operands,
    2 waves per SIMD: the power-limited clock), read from the newest committed suite output -- the GPU box has no compiler-independent way to
    re-measure it inside this run's time budget.  Returns (dict, source path) or (None, None)."""
    import glob
    import re
    cands = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*', 'final_h2_probe.txt')) + glob.glob(os.path.join(REPO, 'profiles', 'r*', 'h2_probe.txt')),
                   key=lambda f: (int(re.search(r'profiles/r(\d+)/', f).group(1)), os.path.basename(f).startswith('final_')))
    for f in reversed(cands):
        rates = {}
        for line in open(f):
            m = re.match(r'rate\s+(bf16|f16)\s+16x16x32, (\d) per pair\s+[\d.]+ ms\s+([\d.]+) TFLOP/s', line)
            if m:
                rates.setdefault(f'{m.group(1)}_{m.group(2)}_per_pair', []).append(float(m.group(3)))
        if rates:
            return {k: max(v) for k, v in rates.items()}, os.path.relpath(f, REPO)
    return None, None


def lib_sha():
    """Content hash of the library binary this process loaded (an A/B whose two arms print the same hash is not an A/B)."""
    import hashlib
    from pnnp_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def csrc_sha():
    """Content hash of the kernel sources: PMC-derived numbers stored under profiles/ are attached to a bench line only when
    they were measured on exactly these kernels (the GPU box has no .git to ask for HEAD)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, 'pnnp_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='crops per GPU per step')
    ap.add_argument('--strong', action='store_true', help='SURVEY 8(d) C4 strong scaling: --batch is the GLOBAL batch, split over the ranks')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--arch', default='unet', choices=['unet', 'resunet'], help='resunet + --noise noiseflow = BASELINE config 5')
    ap.add_argument('--noise', default='physics', choices=['physics', 'noiseflow'])
    ap.add_argument('--proxy-mode', default='train', choices=['train', 'eval'],
                    help='NoiseFlow proxy BatchNorm mode while sampling: train = trainer_LRID (config 5), eval = trainer_SID')
    ap.add_argument('--family', default='h2', choices=['h2', 'x3', 'wino', 'direct'],
                    help='3x3 kernel family: h2 = fp16x2 split on the fp16 matrix cores (csrc/h2.h; default; where it does not apply: x3), x3 = bf16x3 split on the bf16 matrix cores, wino = Winograd on the fp32 matrix cores, direct = fp32 implicit GEMM')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-seconds', type=float, default=240.0,
                    help='time budget of the CPU baseline legs (BASELINE.md section 3 protocol: B=16, 2 warm-up + 5 timed steps, all physical cores + 1 thread); '
                         'a leg that runs out stops early and says so')
    ap.add_argument('--force-reducer', action='store_true', help='N=1 only: run the bucketed RCCL all-reduce path on a 1-rank group '
                    '(for rocprofv3 traces of the collective kernels on the side stream overlapping the backward pass)')
    ap.add_argument('--overlap-allreduce', action='store_true',
                    help='N > 1: launch the gradient all-reduce in buckets from inside the backward pass (default: one collective after it; '
                         'the persistent convolution kernels need every CU, see pnnp_amd.trainer.BucketedAllReduce)')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip per-launch HIP events (roofline becomes whole-step)')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not os.path.exists(os.path.join(REPO, 'pnnp_amd', 'libpnnp_hip.so')):
        if rank == 0:
            ge.build()
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('PNNP_BENCH_BACKEND', 'nccl')       # 'gloo': rehearsal of the N>1 path on a box with fewer GPUs than ranks
        if backend != 'nccl':
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
        dist.barrier()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if world == 1 and args.force_reducer:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)

    from pnnp_amd import ops
    from pnnp_amd.archs import NoiseFlow, ResUnet, UNetSeeInDark, initialize_weights
    from pnnp_amd.trainer import HipTrainStep

    torch.manual_seed(1997)                     # same init on every rank (utils/utils.py:45-48 seeds 1997)
    np.random.seed(1997)
    net = (UNetSeeInDark if args.arch == 'unet' else ResUnet)(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
    initialize_weights(net)
    proxy = None
    if args.noise == 'noiseflow':               # random-init proxy (no checkpoint here), non-trivial couplings
        proxy = NoiseFlow({'x_shape': (4, args.size, args.size), 'arch': 'sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc'})
        with torch.no_grad():
            for k, v in proxy.state_dict().items():
                if k.endswith('conv2d_3.weight'):
                    v.normal_(0, 0.05)
        # trainer_LRID.py:34-39 never calls .eval() on the proxy it samples from: BASELINE config 5 (IMX686 LRID) samples with
        # BatchNorm on BATCH statistics (training mode); --proxy-mode eval is the SID trainer's variant (trainer_SID.py:41-42)
        proxy = proxy.to(dev)
        proxy = proxy.train() if args.proxy_mode == 'train' else proxy.eval()
    net = net.to(dev)
    net.engine.set_policy(x3=args.family in ('x3', 'h2'), wino=args.family != 'direct', h2=args.family == 'h2')
    B, S = args.batch, args.size
    global_batch = B * world
    if args.strong:                             # one global batch split by shard_crops (remainders to the low ranks)
        from pnnp_amd.trainer import shard_crops
        global_batch = B
        lo, hi = shard_crops(global_batch, rank, world)
        B = hi - lo
        if global_batch < world:
            raise SystemExit(f'--strong: global batch {global_batch} is smaller than {world} ranks')
    ts = HipTrainStep(net, lr=1e-4, camera_type='SonyA7S2', noise_code='pr', ori=False, clip=2, seed=1997,
                      rank=rank, world=world, force_reducer=(world == 1 and args.force_reducer),
                      global_batch=global_batch if args.strong else None, overlap_allreduce=args.overlap_allreduce)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    hr = torch.rand(B, 4, S, S, device=dev, generator=g)          # synthetic clean crops, resident in HBM
    hr_nf = hr * 0.01                                             # dark crops for the NoiseFlow proxy (clean/gain scale)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(step):
        np.random.seed(1997 + step * world + rank)                # per-crop params from sample_params_max
        if proxy is not None:                                     # config 5: NoiseFlow proxy, IMX686 ratios {1,2,4,8,16}
            noisy, _, _ = ts.make_noisy_proxy(hr_nf, proxy, ratio_choices=(1, 2, 4, 8, 16), iso=6400)
            return ts.step(hr_nf, noisy=noisy)
        return ts.step(hr)

    for i in range(args.warmup):
        one_step(i)
    ts._state(dev)                              # (with --warmup 0: the reducer exists before the timed region asks for its wait times)
    barrier()
    # HIP events on the launching stream: in the TIMED region only around the dominant kernel's launches (the `roofline` object);
    # the per-class table comes from a short un-timed pass afterwards, so that the headline number is not taxed by ~200 event
    # records per step (measured: 1.7 %).
    pol = net.engine.policy
    fam0 = ('h2' if getattr(pol, 'h2', False) else 'x3') if pol.x3 else ('wino' if pol.wino else 'direct')
    dom_kinds = set(FAMILY[fam0][2])
    if not args.no_kernel_events:
        ops.PROFILE, ops.PROFILE_KINDS = [], dom_kinds
    if ts.reducer is not None:
        ts.reducer.time_waits = True            # one event pair per step around reducer.finish(): is the all-reduce hidden?
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(args.warmup + i)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0           # this rank's own time to finish its K steps (before the barrier)
    barrier()
    dt = time.perf_counter() - t0
    waits = ts.reducer.wait_ms() if ts.reducer is not None else []
    if ts.reducer is not None:
        ts.reducer.time_waits = False
    prof, ops.PROFILE, ops.PROFILE_KINDS = ops.PROFILE, None, None
    prof_all, extra_steps = None, 3
    if not args.no_kernel_events:
        ops.PROFILE = []
        for i in range(extra_steps):
            one_step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        prof_all, ops.PROFILE = ops.PROFILE, None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # per-rank view for the scaling record: own ms/step and the compute stream's wait for the gradient all-reduce
        mine = torch.tensor([1e3 * dt_own / args.steps, (sum(waits) / len(waits)) if waits else 0.0, max(waits) if waits else 0.0],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu()
    elif waits:
        per_rank = torch.tensor([[1e3 * dt_own / args.steps, sum(waits) / len(waits), max(waits)]], dtype=torch.float64)
    dt = float(tmax.item())
    loss_val = float(loss[0])
    if not np.isfinite(loss_val):
        # a number measured on a diverged network is not a measurement (NaN / inf operands even run FASTER on this power-limited chip)
        raise SystemExit(f'bench: the training loss is {loss_val} after {args.warmup + args.steps} steps: the step is broken, no throughput is reported')
    spread = ts.replica_checksum() if world > 1 else None      # 0.0: every rank holds bit-identical weights after the run

    if rank == 0:
        crops = global_batch * args.steps
        value = crops / dt
        out = {
            "metric": "512x512x4 raw crops/sec (train step)", "value": value, "unit": "crops/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": ("f32 (operands, accumulation and results; 3x3 and ConvTranspose2d / 1x1 / stride-2 forward / backward-data / backward-weight multiply on the fp16 matrix cores through a per-tensor-scaled 2-way split, 22 significand bits per operand: csrc/h2.h)" if getattr(pol, 'h2', False) else "f32 (operands, accumulation and results; 3x3 forward / backward-data multiply on the bf16 matrix cores through an exact 3-way split)") if pol.x3 else "f32", "data": "synthetic",
            "config": {"workload": ("PNNP noise-proxy ('pr' physics sampler, SonyA7S2 params)" if proxy is None else f"NoiseFlow.sample proxy (iso 6400, ratio in {{1,2,4,8,16}}, BatchNorm in {args.proxy_mode} mode)") +
                                   (" + UNetSeeInDark" if args.arch == "unet" else " + ResUnet") + " nf=32 train step (fwd + L1 + bwd + Adam)", "crops_per_gpu": B, "global_batch": global_batch,
                       "crop": f"4x{S}x{S}", "parallelism": f"dp{world}", "optimizer": "Adam lr 1e-4"},
            "final_loss": loss_val, "lib_sha": lib_sha(), "csrc_sha": csrc_sha(),
        }
        if spread is not None:
            out["replica_checksum_spread"] = spread
        if per_rank is not None:
            # What the event pair around reducer.finish() on the compute stream measures depends on the mode, so the field is named after it:
            #   --overlap-allreduce: `allreduce_wait_ms_per_step` = how long the compute stream stood still for buckets that were still in
            #     flight (~0 when RCCL hides behind the remaining backward kernels);
            #   default (one collective behind the backward pass): `allreduce_ms_per_step` = the whole collective, all of it exposed.
            out["per_rank_ms_per_step"] = {"min": float(per_rank[:, 0].min()), "max": float(per_rank[:, 0].max()),
                                           "all": [round(float(v), 4) for v in per_rank[:, 0]]}
            out["allreduce_wait_ms_per_step" if args.overlap_allreduce else "allreduce_ms_per_step"] = {"mean_min_over_ranks": float(per_rank[:, 1].min()), "mean_max_over_ranks": float(per_rank[:, 1].max()),
                                                 "worst_step_any_rank": float(per_rank[:, 2].max()),
                                                 "overlap": bool(args.overlap_allreduce), "bucket_bytes": ts.bucket_bytes, "grad_bytes": int(net.engine.params.grad.numel() * 4)}
        step_tflops = GFLOP_PER_CROP_TRAIN[args.arch] * (S * S / (512 * 512)) * B * 1e-3 / (dt / args.steps)
        classes = {}
        if prof:
            inst = {}                               # (kind, instantiation) of the dominant kernel: its own line against the roof that bounds it
            for kind, fl, by, e0, e1, sub in prof:
                ms_ = e0.elapsed_time(e1) * 1e-3
                c = classes.setdefault(kind, [0, 0.0, 0.0, 0.0])
                c[0] += 1; c[1] += fl; c[2] += by; c[3] += ms_
                if sub:
                    q = inst.setdefault((kind, sub), [0, 0.0, 0.0, 0.0])
                    q[0] += 1; q[1] += fl; q[2] += by; q[3] += ms_
            # dominant kernel = the family of 3x3 forward / backward-data launches timed in the timed region
            fam = fam0
            dom = [k for k in classes if k in FAMILY[fam][2]]
            use_wino = fam == 'wino'
            n = sum(classes[k][0] for k in dom); fl = sum(classes[k][1] for k in dom); sec = sum(classes[k][3] for k in dom)
            # `frac` is a hardware fraction: FLOPs the matrix pipe EXECUTES per second / its dense peak.  The Winograd kernels
            # execute 16 multiply-adds where the direct form has 36 (F(2x2,3x3)): executed = algorithmic x 16/36.
            exec_factor, peak, _, kdesc = FAMILY[fam]
            alg_tflops = fl / sec / 1e12
            traffic, traffic_note = None, None      # HBM bytes per launch from the PMC passes of this command (tools/traffic_from_pmc.py)
            tj = os.path.join(REPO, 'profiles', 'traffic.json')
            if os.path.exists(tj) and args.arch == 'unet' and args.noise == 'physics' and B == 16 and S == 512:
                tdoc = json.load(open(tj))
                if tdoc.get('csrc_sha') == csrc_sha():
                    traffic = tdoc.get({'x3': 'x3', 'h2': 'h2', 'wino': 'wino', 'direct': 'igemm9'}[fam], {}).get('hbm_bytes_per_launch')
                    traffic_note = f"PMC passes at commit {tdoc.get('commit')}, kernel sources {tdoc.get('csrc_sha')}"
                else:
                    traffic_note = f"profiles/traffic.json was measured on other kernel sources ({tdoc.get('csrc_sha')} != {csrc_sha()}): not attached"
            by = sum(classes[k][2] for k in dom)
            # `frac` = FLOPs the matrix pipe EXECUTES / its dense peak (MFMA utilisation: what the hardware does);
            # `frac_algorithmic` = SURVEY 8(d)'s definition: algorithmic FLOPs (2 px Cout Cin 9 per layer) / launch time / the same peak --
            # the fp16x2 scheme executes 28/9 piece products per float32 product, so the second is the first / 3.11.
            out["roofline"] = {"bound": "mfma", "achieved": alg_tflops * exec_factor, "peak": peak, "unit": "TFLOP/s",
                               "frac": alg_tflops * exec_factor / peak, "frac_algorithmic": alg_tflops / peak,
                               "traffic": traffic, "traffic_source": traffic_note,
                               "alg_bytes_per_launch": by / n, "kernel": kdesc,
                               "launches": n, "avg_launch_ms": 1e3 * sec / n,
                               "achieved_algorithmic": alg_tflops, "executed_over_algorithmic": exec_factor,
                               "alg_gflop_per_launch": fl / n / 1e9, "executed_gflop_per_launch": fl / n / 1e9 * exec_factor}
            if inst:
                # the dominant kernel by instantiation: the 64-column tiles against the matrix pipe, the 32-column tiles (the 32-channel layers
                # of the 512 x 512 level) against HBM -- algorithmic bytes = the tensors a layer must read and write once, / launch time
                tab = {}
                for (kind, sub), (n_, fl_, by_, s_) in sorted(inst.items()):
                    hbm = sub == 'bn32'
                    row = {"launches": n_, "ms_per_step": 1e3 * s_ / args.steps, "share_of_kernel_time": s_ / sec,
                           "algorithmic_tflops": fl_ / s_ / 1e12, "algorithmic_tbytes_per_s": by_ / s_ / 1e12}
                    row.update({"bound": "hbm", "peak": PEAK_HBM_TBS, "unit": "TB/s", "frac": by_ / s_ / 1e12 / PEAK_HBM_TBS} if hbm else
                               {"bound": "mfma", "peak": peak, "unit": "TFLOP/s", "frac": fl_ / s_ / 1e12 * exec_factor / peak,
                                "frac_algorithmic": fl_ / s_ / 1e12 / peak})
                    tab[f"{'igemm_h2s_kernel' if fam == 'h2' else kind}<{sub[2:]},{'forward' if 'fwd' in kind else 'backward-data'}>"] = row
                out["roofline"]["instantiations"] = tab
                hb = sum(v[3] for (k_, sub), v in inst.items() if sub == 'bn32')
                out["roofline"]["hbm_bound_share_of_kernel_time"] = hb / sec
            if fam == 'h2':
                # informational: a bare register-only loop of v_mfma_f32_16x16x32_f16 on random operands (the power-limited clock), 2 waves per SIMD
                pr, psrc = probe_rates()
                if pr:
                    out["roofline"]["sustained_bare_mfma_loop"] = {"f16_6_products_per_operand_pair": pr.get('f16_6_per_pair'), "f16_3_products_per_operand_pair": pr.get('f16_3_per_pair'),
                                                                   "unit": "TFLOP/s", "source": f"tools/ubench/h2_probe.hip part (1), {psrc}"}
            if fam == 'x3':
                # informational: what a bare loop of the 32x32x16 MFMA sustains on this chip (the 16x16x32 shape the kernel uses: ~1950; tools/ubench/mfma_shape.hip, 8 waves per CU, operands
                # re-read from LDS): 1790 TFLOP/s on random operands (power-limited clock), 2250 on zero operands; `peak` stays the 2.4 GHz figure
                out["roofline"]["sustained_bare_mfma_loop"] = {"random_operands": 1790.0, "zero_operands": 2250.0, "unit": "TFLOP/s",
                                                               "source": "tools/ubench/mfma_shape.hip (DESIGN 5)"}
            # per-class table: from the un-timed pass with events around every launch (extra_steps steps)
            call = {}
            for kind, fl2, by2, e0, e1, _sub in (prof_all or []):
                c = call.setdefault(kind, [0, 0.0, 0.0, 0.0])
                c[0] += 1; c[1] += fl2; c[2] += by2; c[3] += e0.elapsed_time(e1) * 1e-3
            if call:
                t_all = lambda ks: sum(call[k][3] for k in ks)
                conv9 = [k for k in call if k.startswith('conv9_')]
                out["conv3x3_all"] = {"launches": sum(call[k][0] for k in conv9), "ms_per_step": 1e3 * t_all(conv9) / extra_steps,
                                      "tflops": sum(call[k][1] for k in conv9) / t_all(conv9) / 1e12, "from": f"{extra_steps} un-timed steps with events on every launch"}
                out["kernel_classes"] = {k: {"launches": v[0], "ms_per_step": 1e3 * v[3] / extra_steps,
                                             "tflops": (v[1] / v[3] / 1e12) if v[3] > 0 else None} for k, v in sorted(call.items())}
                out["mfma_time_frac_of_step"] = sum(v[3] for v in call.values()) / extra_steps / (dt / args.steps)
        else:
            out["roofline"] = {"bound": "mfma", "achieved": None, "peak": FAMILY[fam0][1], "unit": "TFLOP/s", "frac": None, "traffic": None,
                               "kernel": "not measured (--no-kernel-events)", "step_algorithmic_tflops": step_tflops}
        out["step_tflops_per_gpu"] = step_tflops
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, S, args.cpu_baseline_seconds)
        # RCCL prints its version banner through C stdio (block-buffered when piped): push that out first so that the JSON line is the
        # last line on stdout, then keep the communicator alive until after the print (its teardown prints nothing)
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
