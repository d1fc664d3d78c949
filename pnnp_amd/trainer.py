"""The training step of the reference (trainer_SID.py:86-102,421-486) as one fused device
pipeline, and its data-parallel form (one process per GPU, RCCL all-reduce over xGMI).

    per crop: sample_params_max (host scalars) -> noise sampler -> clamp -> UNet forward ->
    L1(pred.clamp(0,1), hr) -> backward -> [all-reduce of gradients] -> Adam

Everything between the clean crops and the updated weights runs in libpnnp_hip.so; torch
supplies device memory, streams and the process group only.  The reference's single-process
``nn.DataParallel`` (base_trainer.py:115-118) is replaced by sharding the crops over ranks:
rank r owns global crops [r*B, (r+1)*B); the sampler's counter RNG is keyed by the global
crop index, so the noise does not depend on the number of GPUs.
"""
import math

import os

import numpy as np
import torch

from . import ops, process
from ._lib import PnnpError


def get_cos_lr(step, period=1000, peak=20, lr=1e-4, ratio=0.2):
    """base_trainer.py:140-149 -- SGDR with warm-up on restarts; returns the absolute lr."""
    T = step // period
    s = step % period
    if s <= peak and T > 0:
        mul = s / peak
    else:
        mul = (1 - ratio) * (np.cos((s - peak) / (period - peak) * math.pi) * 0.5 + 0.5) + ratio
    return lr * mul / (2 ** T)


class BucketedAllReduce:
    """Sum-all-reduce of a flat gradient buffer in contiguous buckets, launched from the END
    of the buffer backwards as the backward pass finishes layers (the flat layout is the
    forward parameter order, the backward pass completes it in exactly the reverse order),
    on a side stream so RCCL overlaps the remaining backward kernels.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring moves 2(N-1)/N x bytes over
    one link per direction, ~0.35 ms for the 31 MB of UNet gradients -- a few 8 MB buckets keep
    each collective bandwidth-bound rather than latency-bound while leaving >= 3 to pipeline.

    ``overlap`` (default False).  The convolution kernels are PERSISTENT: one workgroup per CU (150-160 KB of LDS, 2 x ~230
    registers per SIMD lane: nothing else fits beside it) with an equal, static share of the tiles.  A collective kernel that is
    resident on even a few CUs when such a grid is dispatched leaves that many workgroups waiting for a free CU, and they run their
    whole share after the others have finished: the layer takes up to twice as long.  With the gradient traffic this small (31 MB:
    ~0.4-0.5 ms of ring time against a ~23 ms step, i.e. 2 %) hiding it is worth less than that risk, so by default the all-reduce is
    ONE collective over the whole flat buffer, issued when the backward pass has finished; ``overlap=True`` launches the buckets
    from the backward pass as described above (bench.py --overlap-allreduce; measure before relying on it).  Round 4 removed the
    hazard itself for the forward / backward-data kernels: with ``ops.set_persistent_split(4)`` (HipTrainStep sets it around the
    backward pass of a step when ``overlap_allreduce`` is on, and restores the previous value) they launch quarter shares that the hardware dispatcher hands to whichever CU is free -- measured with a
    kernel squatting on 8 / 32 / 64 CUs beside conv2_2: x 1.05 / 1.04 / 1.21 instead of x 1.45 (tests/test_gpu_overlap.py,
    tools/squat_test.py); alone on the chip the split costs 6-14 % of a layer, hence not the default.  Round 5: the 3x3 backward-weight kernels
    follow the same setting (four times the slabs, summed by slab index in a fixed order: deterministic for a given share, another pixel
    partition -- i.e. other rounding -- than with static shares; tests/test_gpu_overlap.py)."""

    def __init__(self, flat, bucket_bytes=8 << 20, group=None, force=False, overlap=False):
        import torch.distributed as dist
        self.dist = dist
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: exercise the collective path at world size 1
        self.overlap = bool(overlap)
        n = flat.numel()
        per = max(1, bucket_bytes // 4)
        edges = list(range(n, 0, -per))[::-1]            # bucket starts, counted from the end
        starts = [max(0, e - per) for e in edges]
        self.buckets = sorted(set(zip(starts, edges)))   # [(start, end)] ascending
        self.pending = len(self.buckets) - 1             # next bucket to launch (from the end)
        self.works = []
        self.cuda = flat.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        # one event per bucket, allocated once: "the backward stream has finished this bucket's gradients"
        self.events = [torch.cuda.Event() for _ in self.buckets] if (self.cuda and self.active) else []
        # measurement (bench.py --gpus N): with `time_waits` set, finish() brackets its waits with an event pair on the compute
        # stream; wait_ms() then says how long the compute stream stood still for the collectives (0 = communication hidden)
        self.time_waits = False
        self.wait_pairs = []

    def reset(self):
        self.pending = len(self.buckets) - 1
        self.works = []

    def ready(self, offset):
        """Everything at flat[offset:] has its final gradient: launch every bucket that lies
        entirely in that range.  (Without ``overlap`` nothing is launched here: finish() reduces the whole buffer at once.)"""
        if not self.active or not self.overlap:
            return
        while self.pending >= 0 and self.buckets[self.pending][0] >= offset:
            s, e = self.buckets[self.pending]
            view = self.flat[s:e]
            if self.cuda:
                ev = self.events[self.pending]
                ev.record(torch.cuda.current_stream())
                self.comm_stream.wait_event(ev)
                with torch.cuda.stream(self.comm_stream):
                    self.works.append(self.dist.all_reduce(view, group=self.group, async_op=True))
            else:
                self.works.append(self.dist.all_reduce(view, group=self.group, async_op=True))
            self.pending -= 1

    def finish(self):
        """Launch what is left and make the current stream wait for every bucket."""
        timed = self.time_waits and self.cuda and self.active
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        if self.active and not self.overlap:
            self.dist.all_reduce(self.flat, group=self.group)          # one collective, in order behind the backward pass
            self.pending = -1
        else:
            self.ready(0)
            for w in self.works:
                w.wait()
        self.works = []
        if timed:
            e1.record(torch.cuda.current_stream())
            self.wait_pairs.append((e0, e1))

    def wait_ms(self):
        """Per finish() since the last call: milliseconds the compute stream spent between reaching finish() and having every
        bucket reduced (waits for the recorded events only).  With ``overlap`` that is the WAIT for collectives still in flight (0 =
        communication hidden; includes launching the buckets backward had not released yet); without it (the default) the pair
        brackets the one collective itself, i.e. the whole, exposed all-reduce time."""
        if not self.wait_pairs:
            return []
        self.wait_pairs[-1][1].synchronize()
        out = [a.elapsed_time(b) for a, b in self.wait_pairs]
        self.wait_pairs = []
        return out


def shard_crops(global_batch, rank, world):
    """Rank r of `world` owns crops [lo, hi) of a global batch (strong scaling); returns
    (lo, hi).  Remainders go to the low ranks."""
    q, r = divmod(global_batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class HipTrainStep:
    """One optimiser step on a batch of clean crops ``hr`` [B,4,H,W] (CUDA):
    noise sampler ('camera_type', 'noise_code', 'ori', 'clip' as in the dst section of the
    YAMLs) -> denoiser -> L1 -> backward -> (all-reduce) -> Adam.  Returns the device tensor
    ``[loss, sse_0 .. sse_{B-1}]`` without synchronising.

    ``ori=True`` (dst.ori in the YAMLs): the sampler leaves the noisy crop un-brightened and the prediction is multiplied
    by the per-crop ratio before the loss (trainer_SID.py:97-99), so loss, PSNR and the gradient carry the ratio.

    Data parallel (world > 1): replicas must start identical.  The reference's ``nn.DataParallel`` re-broadcasts the
    parameters from device 0 on every forward (base_trainer.py:115-118); here rank 0's flat parameter buffer is broadcast
    ONCE, at the first step (``sync_replicas``), after which identical gradients (all-reduce) and the deterministic fused
    Adam keep the replicas bit-identical; ``replica_checksum`` lets a caller verify that at any time."""

    def __init__(self, net, lr=1e-4, camera_type='SonyA7S2', noise_code='pr', ori=False, clip=process.HALF_CLIP,
                 seed=1997, rank=0, world=1, group=None, bucket_bytes=8 << 20, force_reducer=False, tukey=False,
                 proxy_net=None, proxy_ratio_choices=None, proxy_iso=None, global_batch=None, overlap_allreduce=False):
        """``proxy_net`` (a NoiseFlow, `arch_proxy` of the run files): noise comes from ``proxy_net.sample`` instead of the
        physics sampler -- the 'NF_Syn_Dataset' branch of preprocess (trainer_SID.py:463-472: ratio ~ U(100,300) per crop,
        one random legal ISO per batch) or, with ``proxy_ratio_choices`` (dst.ratio_list), the 'IMX686_NF_Syn_Dataset'
        branch (trainer_LRID.py:33,419-427: one ratio of the list per batch, ISO from the data: ``proxy_iso`` or step(iso=)).

        ``global_batch``: None = weak scaling, every rank steps on the same number of crops and rank r owns global crops
        [r B, (r+1) B).  An integer = strong scaling of ONE global batch over the ranks by ``shard_crops`` (remainders to the low
        ranks, so shards may differ by one crop): the rank's crops are [lo, hi) of that batch -- the sampler's counter base is
        ``lo``, and the local gradient (a mean over the rank's crops) is weighted by B_local / global_batch so that the
        all-reduced sum is the mean over the global batch."""
        self.global_batch = global_batch
        self.overlap_allreduce = overlap_allreduce      # BucketedAllReduce(overlap=): buckets from inside the backward pass (see its docstring)
        # collectives resident on some CUs while convolution grids are dispatched: quarter shares handed out by the hardware dispatcher
        # instead of one static share per CU (tools/squat_test.py: x 1.05 instead of x 1.45 beside a kernel on 32 CUs).  The setting is
        # process-wide library state, so it is scoped to the backward pass of a step whose reducer overlaps (step(): set, run, restore) --
        # an eval loop, a layer benchmark or another HipTrainStep in the same process keeps whatever it had.
        self.overlap_split = 4 if (overlap_allreduce and (world > 1 or force_reducer)) else None
        self.proxy_check_every = 50          # steps between reads of the NoiseFlow proxy's `scale >= 0` flag (a host sync each)
        self.net = net
        self.engine = net.engine
        self.lr = lr
        self.camera_type, self.noise_code, self.ori, self.clip = camera_type, noise_code, ori, clip
        self.seed = seed
        self.tukey = tukey          # extension: let noise codes with 'g' run (Tukey-lambda read noise on the device)
        self.proxy_net, self.proxy_ratio_choices, self.proxy_iso = proxy_net, proxy_ratio_choices, proxy_iso
        self.rank, self.world, self.group = rank, world, group
        self.bucket_bytes = bucket_bytes
        self.force_reducer = force_reducer
        self.step_count = 0
        self.m = self.v = None
        self.reducer = None
        self._loss = None
        self._synced = False

    def sync_replicas(self):
        """Broadcast rank 0's parameters (and Adam moments) to every rank of the group: the start-up counterpart of
        DataParallel's per-step replicate() (base_trainer.py:115-118)."""
        import torch.distributed as dist
        if self.world > 1 and dist.is_initialized():
            for t in (self.engine.params.flat, self.m, self.v):
                dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            self.engine.mark_dirty()
        self._synced = True

    def replica_checksum(self):
        """(max - min) over ranks of a parameter checksum: 0.0 when all replicas hold bit-identical weights."""
        import torch.distributed as dist
        flat = self.engine.params.flat
        w = (torch.arange(flat.numel(), device=flat.device) % 977 + 1).double()          # position-dependent weights: catches permutations
        cs = torch.stack([flat.double().sum(), (flat.double() * w).sum()])
        if self.world > 1 and dist.is_initialized():
            hi, lo = cs.clone(), cs.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            return float((hi - lo).abs().max())
        return 0.0

    def _state(self, device):
        self.engine.params.ensure(device)
        flat = self.engine.params.flat
        if self.m is None or self.m.numel() != flat.numel() or self.m.device != device:
            self.m = torch.zeros_like(flat)
            self.v = torch.zeros_like(flat)
        if (self.world > 1 or self.force_reducer) and (self.reducer is None or self.reducer.flat is not self.engine.params.grad):
            self.reducer = BucketedAllReduce(self.engine.params.grad, self.bucket_bytes, self.group, force=self.force_reducer,
                                             overlap=self.overlap_allreduce)
        if not self._synced:
            self.sync_replicas()

    def shard(self, B):
        """(first global crop index of this rank's B local crops, weight of the local mean-gradient in the global mean x world)."""
        if self.global_batch is None:
            return self.rank * B, 1.0
        lo, hi = shard_crops(self.global_batch, self.rank, self.world)
        if hi - lo != B:
            raise PnnpError(f'rank {self.rank} of {self.world} owns crops [{lo}, {hi}) of a global batch of {self.global_batch}, got {B} crops')
        if B == 0:
            raise PnnpError('a rank without crops cannot take part in the step (global batch < world size)')
        return lo, B * self.world / self.global_batch

    def sample_noise_params(self, batch):
        """trainer_SID.py:451-459: one host-side parameter draw per crop."""
        return [process.sample_params_max(camera_type=self.camera_type, ratio=None) for _ in range(batch)]

    def make_noisy(self, hr, plist=None, rows=None):
        """preprocess(): per-crop physics noise on the clean crops, then the trainer's clamp
        (trainer_SID.py:461,481-485) -- fused into the sampler's store."""
        B = hr.shape[0]
        if rows is None:
            rows = process.pack_params(plist if plist is not None else self.sample_noise_params(B), hr.device)
        # `clip: 2` (HALF_CLIP) is truthy: generate_noisy_torch clamps to [0,1] before x ratio
        # (process.py:668), then preprocess clamps the result to (-inf, 1] (trainer_SID.py:483-484)
        code = self.noise_code.lower()
        if 'g' in code and 'b' not in code and not self.tukey:
            raise NotImplementedError            # process.py:654
        flags = process.noise_flags(code if self.tukey else code.replace('g', ''), ori=self.ori, clip=bool(self.clip), torch_mode=True)
        if self.tukey and 'g' in code:
            flags |= process.F_TORCH_TUKEY
        if self.clip:
            flags |= process.F_POST_MAX1 | (0 if self.clip == process.HALF_CLIP else process.F_POST_MIN0)
        return process.noise_sample(hr, rows, flags, seed=self.seed, offset=self.step_count,
                                    crop_base=self.shard(B)[0]), rows

    LEGAL_ISO = (50, 64, 80, 100, 125, 160, 200, 250, 320, 400, 500, 640, 800, 1000, 1250, 1600, 2000, 2500, 3200,
                 4000, 5000, 6400, 8000, 10000, 12800, 16000, 20000, 25600)             # trainer_SID.py:33-34

    def make_noisy_proxy(self, hr, proxy_net, ratio=None, iso=None, ratio_choices=None):
        """preprocess() with a NoiseFlow proxy (dataset 'NF_Syn_Dataset'):
        ratio ~ U(100,300) per crop (trainer_SID.py:464) or one draw from ``ratio_choices`` per batch
        (trainer_LRID.py:33,420: {1,2,4,8,16}); one random legal ISO per batch (:465);
        noisy = hr + proxy.sample(clean=hr/ratio, iso) * ratio (:466-472); then the clamp (:481-485)."""
        B = hr.shape[0]
        if ratio is None:
            if ratio_choices is not None:
                ratio = float(ratio_choices[np.random.randint(len(ratio_choices))])      # one host scalar per batch: no device op at all
            else:
                ratio = torch.rand(B, dtype=torch.float32, device=hr.device).view(-1, 1, 1, 1) * 200 + 100
        if iso is None:
            iso = self.LEGAL_ISO[np.random.randint(len(self.LEGAL_ISO))]
        if hasattr(proxy_net, 'sample_mixed'):
            # one pass: clean = hr / ratio inside the signal-dependent step, hr + noise * ratio and the clamp in the last step's store,
            # the reference's `assert scale >= 0` (signal_dependant.py:50) as a device flag: read on the FIRST proxy step (a bad
            # checkpoint or ISO fails at once, before any Adam step on NaN noise), then every `proxy_check_every` steps, and by
            # check_proxy() -- which the epoch loop calls at every epoch end and before anything is saved or returned
            lo, hi = (-float('inf'), float('inf'))
            if self.clip:
                lo, hi = (-float('inf') if self.clip == process.HALF_CLIP else 0.0), 1.0
            noisy = proxy_net.sample_mixed(hr, ratio, iso, lo, hi)
            self._proxy_steps = getattr(self, '_proxy_steps', 0) + 1
            if self._proxy_steps == 1 or self._proxy_steps % self.proxy_check_every == 0:
                proxy_net.check_scale_flag()
        else:
            rt = ratio if torch.is_tensor(ratio) else torch.full((B, 1, 1, 1), ratio, device=hr.device)
            noise = proxy_net.sample(clean=hr / rt, iso=iso)
            noisy = torch.addcmul(hr, noise, rt)
            if self.clip:
                noisy = noisy.clamp_(max=1.0) if self.clip == process.HALF_CLIP else noisy.clamp_(0.0, 1.0)
        if not torch.is_tensor(ratio):
            ratio = torch.full((B, 1, 1, 1), ratio, device=hr.device) if self.ori else ratio
        return noisy, ratio, iso

    def step(self, hr, plist=None, rows=None, noisy=None, lr=None, ratio=None, iso=None):
        """``ratio`` [B] (or [B,1,1,1]): the per-crop ratio for ``ori=True`` when ``noisy`` comes from outside (with the
        built-in sampler it is the ratio column of the parameter rows).  ``iso``: the batch's ISO for a NoiseFlow proxy."""
        if not hr.is_cuda:
            raise PnnpError('HipTrainStep needs CUDA tensors (no CPU path)')
        dev = hr.device
        self._state(dev)
        e = self.engine
        B, C, H, W = hr.shape
        if noisy is None and self.proxy_net is not None:
            noisy, ratio, _ = self.make_noisy_proxy(hr, self.proxy_net, ratio=ratio, iso=iso if iso is not None else self.proxy_iso,
                                                    ratio_choices=self.proxy_ratio_choices)
        elif noisy is None:
            noisy, rows = self.make_noisy(hr, plist, rows)
        scale = None
        if self.ori:                                         # trainer_SID.py:97-98: pred = pred * ratio
            if ratio is None and rows is not None:
                ratio = rows[:, 6]                           # column order of process.pack_params
            if ratio is None:
                raise PnnpError('ori=True needs the per-crop ratio: pass rows= / plist= or ratio=')
            scale = ratio.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
            if scale.numel() == 1 and B > 1:
                scale = scale.expand(B).contiguous()
        pred = e.forward(noisy, True)
        bufs = e.bufs[(B, H, W, dev)]
        # dL/d(out), NHWC: 4 channels when the 1x1 head's backward runs on the streaming kernel (it reads gcs >= 4), zero-padded to 8 for the GEMM kernels
        gch = e.grad_out_channels(B, H, W) if hasattr(e, 'grad_out_channels') else e.cout_pad
        g8 = bufs.get('g_out8' if gch == e.cout_pad else 'g_out4', (B, H, W, gch), dev)
        loss = bufs.get('loss_out', (1 + B,), dev)
        lws = bufs.get('loss_ws', (128 * B,), dev)
        # trainer_SID.py:485: the target is clamped under dst.clip, in the kernel; uneven shards of a global batch: this rank's mean
        # gradient counts B_local / B_global (x world: Adam divides the all-reduced sum by world) -- folded into the kernel's gradient scale
        ops.l1_clamp_loss(pred, hr, g8, loss, lws, scale=scale, clamp_target=bool(self.clip), grad_weight=self.shard(B)[1])
        if self.reducer is not None:
            self.reducer.reset()
        prev_split = None
        if self.reducer is not None and self.overlap_split:
            prev_split = ops.get_persistent_split()
            ops.set_persistent_split(self.overlap_split)
        try:
            e.backward(g8, on_ready=self.reducer.ready if self.reducer is not None else None)
        finally:
            if prev_split is not None:
                ops.set_persistent_split(prev_split)
        if self.reducer is not None:
            self.reducer.finish()
        self.step_count += 1
        ops.adam_step(e.params.flat, e.params.grad, self.m, self.v, self.lr if lr is None else lr, self.step_count,
                      grad_scale=1.0 / self.world)
        e.mark_dirty()
        self._loss = loss
        return loss

    def check_proxy(self):
        """Raise the reference's ``AssertionError('scale must be non-negative')`` (signal_dependant.py:50) if any proxy sample since the
        last check saw a negative scale.  One host synchronisation: call it at every epoch end and before a checkpoint is written
        (`runfile.main` does); `make_noisy_proxy` itself checks on the first step and every ``proxy_check_every`` steps."""
        if self.proxy_net is not None and hasattr(self.proxy_net, 'check_scale_flag'):
            self.proxy_net.check_scale_flag()

    @staticmethod
    def psnr_from(loss_out, elems_per_crop):
        """PSNR_Loss (losses/__init__.py:4-15) from the SSE the loss kernel returns."""
        sse = loss_out[1:].double()
        return float((-10.0 * torch.log10(sse / elems_per_crop)).mean())


class NoiseFlowFitStep:
    """One NLL fitting step of the NoiseFlow proxy on synthetic pairs (trainer_NF_SID.py:97-126 with the `Raw_Dataset`
    preprocess branch :431-446): physics-sampler noise at a table ISO on the clean crops ``hr`` [B,4,H,W] (CUDA), then

        noise = (lr - hr) / ratio;  clean = hr / ratio;  nll, sd_z = net.loss(noise=, clean=, iso=);  nll.backward();  Adam

    with ``net`` in training mode (BatchNorm batch statistics, :102).  ``step`` returns the device scalars
    ``(nll + mean log ratio, sd_z * mean ratio)`` that the reference logs (:129-133).

    As in the reference's preprocess: the ratio is ALWAYS the sampled parameter's ratio (:435, also with ``ori``); the
    parameters come from ``sample_params_max(camera_type, ratio=None)`` -- the regression table (:429) -- unless
    ``iso_table=True`` asks for the per-ISO table of the step's ISO; with ``clip`` set (run files: ``clip: 2``) the noisy
    crop is clamped to (-inf or 0, 1] and the clean crop to [0, 1] after the sampler (:438-442)."""

    def __init__(self, net, lr=2e-3, camera_type='SonyA7S2', noise_code='pgrq', ori=False, clip=False, seed=1997,
                 rank=0, world=1, group=None, tukey=True, iso_table=False):
        self.net = net
        self.lr = lr
        self.camera_type, self.noise_code, self.ori, self.clip = camera_type, noise_code, ori, clip
        self.seed, self.rank, self.world, self.group = seed, rank, world, group
        self.tukey = tukey                       # 'g' in the run files' noise_code: drawn on the device (row f3)
        self.iso_table = iso_table
        self.step_count = 0
        self._flat_grad = None
        self.optimizer = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=lr)      # trainer_NF_SID.py:34

    def make_pair(self, hr, iso):
        B = hr.shape[0]
        plist = [process.sample_params_max(camera_type=self.camera_type, ratio=None, iso=iso if self.iso_table else None) for _ in range(B)]
        rows = process.pack_params(plist, hr.device)
        code = self.noise_code.lower()
        if 'g' in code and 'b' not in code and not self.tukey:
            raise NotImplementedError            # process.py:654
        flags = process.noise_flags(code if self.tukey else code.replace('g', ''), ori=self.ori, clip=bool(self.clip), torch_mode=True)
        if self.tukey and 'g' in code:
            flags |= process.F_TORCH_TUKEY
        if self.clip:                            # trainer_NF_SID.py:438-441, fused into the sampler's store
            flags |= process.F_POST_MAX1 | (0 if self.clip == process.HALF_CLIP else process.F_POST_MIN0)
        lr_img = process.noise_sample(hr, rows, flags, seed=self.seed, offset=self.step_count, crop_base=self.rank * B)
        ratio = rows[:, 6].reshape(B, 1, 1, 1)   # :435 (column order of process.pack_params: K sigGs sigTL lam sigR q ratio wp bl)
        return lr_img, ratio

    def step(self, hr, iso=1600, lr=None):
        if not hr.is_cuda:
            raise PnnpError('NoiseFlowFitStep needs CUDA tensors (no CPU path)')
        if lr is not None:
            for g in self.optimizer.param_groups:
                g['lr'] = lr
        self.net.train()
        imgs_lr, ratio = self.make_pair(hr, iso)
        if self.clip:
            hr = hr.clamp(0, 1)                  # :442
        self.optimizer.zero_grad(set_to_none=True)
        nll, sd_z = self.net.loss(noise=(imgs_lr - hr) / ratio, clean=hr / ratio, iso=float(iso))
        nll.backward()
        if self.world > 1:                       # replicas average their gradients (BatchNorm statistics stay per replica,
            import torch.distributed as dist     # as under the reference's nn.DataParallel): ONE all-reduce of the few
            ps = [p for p in self.net.parameters() if p.grad is not None]           # hundred parameters, flattened
            flat = torch.cat([p.grad.reshape(-1) for p in ps])
            dist.all_reduce(flat, group=self.group)
            flat.div_(self.world)
            o = 0
            for p in ps:
                p.grad.copy_(flat[o:o + p.numel()].view_as(p.grad)); o += p.numel()
        self.optimizer.step()
        self.step_count += 1
        return nll.detach() + torch.log(ratio).mean(), sd_z * ratio.mean()
