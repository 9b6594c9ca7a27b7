"""Training step of the SR generator on the HIP path (reference: interfaces/super_resolution_eng.py:214-279,
restricted to BASELINE config 2: one SR forward/backward, ImageLoss.mean()*100, clip_grad_norm_ 0.25, Adam).

Parameters and gradients live in two flat fp32 buffers laid out in the order the backward pass produces the
gradients (tail -> SRBs -> TP interpreter -> block1/STN), so that data-parallel training all-reduces contiguous
ranges over RCCL (two collectives per step: tail + SRBs while the rest of the backward pass is still running, then
TP interpreter + block1/STN), and the fused clip+Adam kernel touches every parameter once.  Every nn.Parameter of the module stays a view of the flat
buffer (state_dict / checkpoints are unaffected).

Data parallel + hipGraph: the step is captured as TWO graphs cut where the SRBs' gradients are complete
(forward + tail + SRBs | TP interpreter + block1/STN) plus one for clip+Adam; between two replays the finished
buckets' all-reduce is issued on RCCL's own stream, so it is in flight while the next segment runs -- the same
overlap the eager path gets from Engine.backward(on_stage=...), and bit-identical to it.  How much can overlap is
bounded by the network: the recurrent positional encoding's weights (19.6 MB of the 30.4 MB) and the STN head
(7.4 MB) are the LAST gradients the backward pass produces, so only the tail and SRB buckets (3.3 MB) travel
under cover; the rest is exposed (reported by bench.py as allreduce_ms_exposed).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .lib import lib

import os

LOSS_ONE_LAUNCH = os.environ.get("TATT_LOSS_ONE_LAUNCH", "1") != "0"   # 0: ImageLoss forward and backward as two launches (A/B runs; same numbers)
BUCKETS = ["tail", "srbs", "infogen", "head"]
# Collectives of a data-parallel step: the backward pass announces a finished bucket (Engine.backward(on_stage=...)); at these two
# announcements ONE all-reduce covers the adjacent buckets named (the flat layout keeps them contiguous).  Few, large collectives: the
# tail bucket is 0.8 MB -- an all-reduce and a graph cut of its own cost more than its early start returned (one rank, B=64: the
# four-segment step took 6.84 ms) -- and the two buckets that the pass finishes last go out as one 27 MB call.
REDUCE_AT = {"srbs": ("tail", "srbs"), "infogen": ("infogen", "head")}


def _bucket_of(name: str, srb_nums: int) -> int:
    blk = name.split(".")[0]
    if blk in (f"block{srb_nums + 2}", f"block{srb_nums + 3}"):
        return 0
    if blk.startswith("block") and blk != "block1":
        return 1
    if blk == "infoGen":
        return 2
    return 3                                            # block1, stn_head


def plan_flat_layout(named_shapes, srb_nums, align=64):
    """Flat-buffer layout in gradient-production order.  named_shapes: [(name, numel)] in module order.
    Returns (offsets {name: offset}, total floats, bucket_ranges [[begin, end]] per BUCKETS entry)."""
    idx = list(range(len(named_shapes)))
    bk = [_bucket_of(n, srb_nums) for n, _ in named_shapes]
    order = sorted(idx, key=lambda i: (bk[i], -i if bk[i] == 1 else i))      # SRBs finish last-block-first
    offs, off, ranges = {}, 0, [[None, None] for _ in BUCKETS]
    for i in order:
        n, numel = named_shapes[i]
        if ranges[bk[i]][0] is None:
            ranges[bk[i]][0] = off
        offs[n] = off
        off += (numel + align - 1) // align * align          # 256-byte aligned slots (vector loads in the kernels)
        ranges[bk[i]][1] = off
    return offs, off, ranges


def allreduce_mean(t, async_op=False):
    """Mean over ranks of a gradient bucket: AVG on RCCL; SUM + scale where AVG is not implemented (gloo on CPU)."""
    if dist.get_backend() == "nccl":
        return dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=async_op)
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=False)
    t.div_(dist.get_world_size())
    return None


def broadcast_buffers(bufs, src=0):
    """Rank `src`'s module buffers on every rank: one collective per dtype over a packed copy (a CRNN + the SR net have a hundred
    BatchNorm buffers of a few hundred bytes each; one broadcast per buffer is a hundred launches -- and, on a backend that stages
    device tensors through the host, a hundred synchronisations)."""
    groups = {}
    for b in bufs:
        groups.setdefault(b.dtype, []).append(b)
    for same in groups.values():
        flat = torch.cat([b.detach().reshape(-1) for b in same])
        dist.broadcast(flat, src)
        off = 0
        for b in same:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()


def replica_fingerprint(t):
    """(sum, sum of squares, two bit-pattern checksums) of a flat fp32 buffer as four float64 values: equal on every rank iff the
    replicas hold bit-identical parameters (up to a 2^-53-ish chance for the sums).  The checksums are position-weighted sums of
    the low and of the high 16 bits of every word (sign, exponent and the top mantissa bits are in the high half), computed in
    chunks so that the int64 temporaries stay small."""
    flat = t.detach().contiguous().view(-1)
    bits = flat.view(torch.int32)
    lo = hi = 0
    step = 1 << 20
    for a in range(0, bits.numel(), step):
        b = bits[a:a + step].to(torch.int64)
        w = (torch.arange(a + 1, a + 1 + b.numel(), device=t.device, dtype=torch.int64) % 8191) + 1   # position-dependent weights
        lo += int(((b & 0xFFFF) * w).sum().item())
        hi += int((((b >> 16) & 0xFFFF) * w).sum().item())
    d = flat.double()
    return torch.tensor([d.sum().item(), (d * d).sum().item(), float(lo % (1 << 52)), float(hi % (1 << 52))], dtype=torch.float64,
                        device=t.device)


def assert_replicas_identical(t, what="parameters"):
    """Data-parallel self-check: every rank must hold the same `t`.  All-gathers replica_fingerprint(t) and raises on EVERY rank
    if any differs (the reference precedent, DistributedDataParallel, verifies parameter shapes only; here the values are
    checked because the replicas are kept in step by construction, not by a framework).  Returns the number of ranks seen."""
    fp = replica_fingerprint(t)
    world = dist.get_world_size()
    allfp = [torch.zeros_like(fp) for _ in range(world)]
    dist.all_gather(allfp, fp)
    bad = [r for r in range(world) if not torch.equal(allfp[r], allfp[0])]
    if bad:
        raise RuntimeError(f"data-parallel replicas diverged: {what} on ranks {bad} differ from rank 0 "
                           f"(fingerprints {[f.tolist() for f in allfp]})")
    return world


class SRTrainer:
    def __init__(self, model, lr=1e-3, betas=(0.5, 0.999), eps=1e-8, max_norm=0.25, loss_weight=(1.0, 1e-4),
                 use_graph=False, distributed=None, tsc=False, tssim_rotation_degree=10.0, training_stablize=False,
                 strict_graph=False):
        self.model, self.L = model, lib()
        if getattr(model, "in_planes", 4) != 4:
            raise NotImplementedError("SRTrainer: the fused training step is wired for mask=True (four channels); a mask=False model "
                                      "trains through its autograd interface (model(lr, prior) -> loss.backward() -> torch.optim)")
        self.lr, self.betas, self.eps, self.max_norm, self.loss_weight = lr, betas, eps, max_norm, loss_weight
        # data parallel whenever a process group exists (a 1-rank group takes the same broadcast / all-reduce path, which is
        # how the RCCL plumbing is tested on a single GPU)
        self.dp = bool(distributed if distributed is not None else dist.is_initialized())
        self.world = dist.get_world_size() if self.dp else 1
        self.eng = model.engine()
        # TSC term of the reference step (interfaces/super_resolution_eng.py:249-262): second SR forward on the rotated LR
        # input, three rotations, (1 - TRI_SSIM) * 10; `--training_stablize` masks the per-sample image loss (:241-242)
        self.tsc, self.tssim_rotation_degree, self.training_stablize = tsc, tssim_rotation_degree, training_stablize
        self.tssim = None                                     # device scalar: TRI_SSIM of the last step
        import os
        self.tsc_batched = os.environ.get("TATT_TSC_BATCH", "1") != "0"   # both SR passes of a TSC step as one 2B-row batch (0: one after the other)
        named = list(model.named_parameters())
        dev = named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("SRTrainer needs the model on a GPU")
        offs, off, self.bucket_range = plan_flat_layout([(n, p.numel()) for n, p in named], model.srb_nums)
        self.n = off
        self.flat_p = torch.zeros(off, device=dev)
        self.flat_g = torch.zeros(off, device=dev)
        self.m = torch.zeros(off, device=dev)
        self.v = torch.zeros(off, device=dev)
        self.state = torch.zeros(8, device=dev)
        self.ws = torch.empty(1024, device=dev)
        self.G = {}
        for n, p in named:
            sl = slice(offs[n], offs[n] + p.numel())
            self.flat_p[sl].copy_(p.data.reshape(-1))
            p.data = self.flat_p[sl].view(p.shape)
            self.G[n] = self.flat_g[sl].view(p.shape)
            p.grad = self.G[n]
        if self.dp:                                           # identical replicas: rank 0's initial weights + buffers
            dist.broadcast(self.flat_p, 0)
            broadcast_buffers(list(model.buffers()))
        self.seed_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.eng.seed_dev = self.seed_dev
        self.use_graph = use_graph
        self.strict_graph = strict_graph                      # True (bench.py): a failed hipGraph capture raises instead of going eager
        self.graph = None                                     # list of captured segments (one entry without a process group)
        self.graph_update = None                              # data parallel: clip+Adam, replayed after the all-reduces
        self.static = None
        self.per = None
        self.steps = 0
        self.time_comm = False                                # bench.py: bracket the exposed all-reduce wait with events
        self.comm_events = []

    # ------------------------------------------------------------------------------------------------------
    @staticmethod
    def loss_stablizing_mask(per, keep_proportion=0.7):
        """0/1 mask of TextSR.loss_stablizing (interfaces/super_resolution_eng.py:38-48), quirk included: everything
        from the ORIGINAL position sorted_ind[int(keep*B)] onward is dropped.  Device-side, no host sync."""
        B = per.shape[0]
        start = torch.argsort(per)[int(keep_proportion * B)]
        return (torch.arange(B, device=per.device) < start).to(per.dtype)

    def _fwd_bwd(self, lr_img, hr_img, tp, overlap, arc=None, tp_ready=None, on_stage=None):
        """overlap: issue each gradient bucket's all-reduce from inside the backward pass (eager data parallel); on_stage: called
        instead at the same points (graph capture cuts its segments there)."""
        L, B = self.L, lr_img.shape[0]
        self.flat_g.zero_()
        L.seed_bump(self.seed_dev)
        batched = self.tsc and self.tsc_batched and self.eng.dec_fused
        if self.tsc:
            arc = arc.to(device=lr_img.device, dtype=torch.float32)
        if batched:
            # The two SR passes of the step (interfaces/super_resolution_eng.py:233 on the LR image, :259 on the rotated LR image: the
            # second reads the rotated INPUT, not the first one's output) as ONE batch of 2B rows through every launch; BatchNorm
            # statistics, dropout masks (seeds 0 / 1), the STN head and the text branch stay per pass (Engine.forward(x2=...)).
            d_lr = torch.empty_like(lr_img)
            L.rotate_fwd(lr_img, arc, d_lr, B, lr_img.shape[1], lr_img.shape[2], lr_img.shape[3])
            sr_all, ctx = self.eng.forward(lr_img, tp, True, 0, tp_ready=tp_ready, x2=d_lr, seed2=1)
            sr, sr2 = sr_all[:B], sr_all[B:]
        else:
            sr, ctx = self.eng.forward(lr_img, tp, True, 0, tp_ready=tp_ready)
        C, H, W = sr.shape[1:]
        per = torch.empty(B, device=sr.device)
        dsr_all = torch.empty(2 * B, C, H, W, device=sr.device) if batched else None
        dsr = dsr_all[:B] if batched else torch.empty_like(sr)
        if self.training_stablize:
            L.image_loss_fwd(sr, hr_img, per, B, C, H, W, self.loss_weight[0], self.loss_weight[1])
            mask = self.loss_stablizing_mask(per)
            per = per * mask
            L.image_loss_bwd(sr, hr_img, mask * (100.0 / B), 0.0, dsr, B, C, H, W, self.loss_weight[0], self.loss_weight[1])
        elif LOSS_ONE_LAUNCH:                                 # losses (reported only) and gradient in one launch
            L.image_loss_fwd_bwd(sr, hr_img, per, 100.0 / B, dsr, B, C, H, W, self.loss_weight[0], self.loss_weight[1])
        else:
            L.image_loss_fwd(sr, hr_img, per, B, C, H, W, self.loss_weight[0], self.loss_weight[1])
            L.image_loss_bwd(sr, hr_img, None, 100.0 / B, dsr, B, C, H, W, self.loss_weight[0], self.loss_weight[1])
        if self.tsc:
            # D(F(Y)), F(D(Y)), D(X): the rotated SR output, the SR of the rotated input, the rotated HR image
            d_sr, d_hr = torch.empty_like(sr), torch.empty_like(hr_img)
            L.rotate_fwd(sr, arc, d_sr, B, C, H, W)
            L.rotate_fwd(hr_img, arc, d_hr, B, C, H, W)
            if not batched:
                d_lr = torch.empty_like(lr_img)
                L.rotate_fwd(lr_img, arc, d_lr, B, lr_img.shape[1], lr_img.shape[2], lr_img.shape[3])
                # seed 1: its own dropout masks; packed weights and the recurrent positional encoding (a function of the weights alone)
                # are those of the first pass, whose backward pass below also runs the one BPTT chain for both
                sr2, ctx2 = self.eng.forward(d_lr, tp, True, 1, wpack=ctx["wpack"], rpe_from=ctx)
            self.tssim = torch.empty(1, device=sr.device)
            L.tri_ssim_fwd(d_sr, sr2, d_hr, self.tssim, B, C, H, W, self.ws_ssim(B * C))
            g_rot = torch.empty_like(sr)
            g_sr2 = dsr_all[B:] if batched else torch.empty_like(sr)
            L.tri_ssim_bwd(d_sr, sr2, d_hr, -10.0, g_rot, g_sr2, 0, B, C, H, W)      # loss_tssim = (1 - mean) * 10
            L.rotate_bwd(arc, g_rot, dsr, B, C, H, W)                                # += onto the image-loss gradient
            if not batched:
                self.eng.backward(ctx2, g_sr2, self.G)
        if batched:
            dsr = dsr_all
        works = []
        cut = on_stage

        def on_stage(name):
            if overlap:
                if name in REDUCE_AT:
                    works.extend(allreduce_mean(t, async_op=True) for t in self._bucket_views(name))
            elif cut is not None:
                cut(name)
        on_stage.stages = frozenset(REDUCE_AT)

        self.eng.backward(ctx, dsr, self.G, on_stage=on_stage if (overlap or cut is not None) else None)
        return per, works

    def ws_ssim(self, n):
        if getattr(self, "_ws_ssim", None) is None or self._ws_ssim.numel() < n:
            self._ws_ssim = torch.empty(n, device=self.flat_p.device)
        return self._ws_ssim

    def sample_arc(self, B):
        """angle_batch = U(-deg, deg) per sample, in radians (interfaces/super_resolution_eng.py:254-255)."""
        import math
        return (torch.rand(B) * 2 - 1) * self.tssim_rotation_degree / 180.0 * math.pi

    def extra_grad_buffers(self):
        """Gradient buffers besides flat_g that a data-parallel step must all-reduce (subclasses)."""
        return []

    def _update(self):
        self.L.clip_adam(self.flat_p, self.flat_g, self.m, self.v, self.n, self.lr, self.betas[0], self.betas[1], self.eps,
                         self.max_norm, self.state, self.ws)

    def check_chains(self):
        """Raise if a single-launch recurrence (RPE chains; LSTM chains in TATTTrainer) ever reported a timed-out wait.  Synchronises."""
        self.eng.check_rpe_chain()

    def verify_replicas(self):
        """Raise unless every rank holds bit-identical parameters and optimiser moments (and the chain kernels
        completed without a timed-out wait); returns the rank count RCCL reports through a collective (1 without a process group)."""
        self.check_chains()
        if not self.dp:
            return 1
        n = assert_replicas_identical(self.flat_p, "SR parameters")
        assert_replicas_identical(self.m, "Adam first moments")
        return n

    def stateful_buffers(self):
        """Buffers a forward pass updates in place (BatchNorm running statistics and counters)."""
        return [b for n, b in self.model.named_buffers() if "running_" in n or n.endswith("num_batches_tracked")]

    def _bucket_views(self, stage):
        """The slice of the flat gradient buffer that the collective issued at announcement `stage` reduces (REDUCE_AT)."""
        first, last = REDUCE_AT[stage]
        a, b = self.bucket_range[BUCKETS.index(first)][0], self.bucket_range[BUCKETS.index(last)][1]
        return [self.flat_g[a:b]]

    def _wait_all(self, works):
        """Make the current stream wait for the collectives (no host block); with time_comm the wait is bracketed by events:
        what it takes is the part of the all-reduce time that the backward pass did not cover."""
        e0 = e1 = None
        if self.time_comm:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in works:
            if w is not None:
                w.wait()
        if self.time_comm:
            e1.record()
            self.comm_events.append((e0, e1))

    def exposed_comm_ms(self):
        """Mean over the recorded steps of the un-overlapped all-reduce wait (call after a device synchronisation)."""
        ms = [a.elapsed_time(b) for a, b in self.comm_events]
        self.comm_events = []
        return sum(ms) / len(ms) if ms else 0.0

    @staticmethod
    def _is_capture_error(e):
        msg = str(e).lower()
        return isinstance(e, RuntimeError) and ("captur" in msg or "graph" in msg) and "fault" not in msg and "illegal" not in msg

    def _capture(self, lr_img, hr_img, tp, arc):
        """Warm up once outside capture (allocator, lazy code-object loads) with every side effect undone, then capture the step:
        one graph without a process group; with one, segments cut where a gradient bucket is complete (module docstring)."""
        self.static = [lr_img.clone(), hr_img.clone(), tp.clone()] + ([arc.clone()] if self.tsc else [])
        args = dict(arc=self.static[3] if self.tsc else None)
        bufs = self.stateful_buffers()
        snap = [b.clone() for b in bufs]                  # the warm-up pass must not count as a training step: BatchNorm running
        s = torch.cuda.Stream()                           # statistics / num_batches_tracked are put back, like the dropout seed
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._fwd_bwd(*self.static[:3], overlap=False, **args)
        torch.cuda.current_stream().wait_stream(s)
        for b, v in zip(bufs, snap):
            b.copy_(v)
        self.seed_dev.zero_()
        # thread_local: RCCL's watchdog thread polls events of earlier collectives (the parameter broadcast, the previous
        # step's all-reduce); under the default global mode such a poll from ANOTHER thread during capture aborts the
        # process (seen once in tests/test_gpu_dp.py when the capture began right after an eager data-parallel step)
        torch.cuda.synchronize()
        graphs, pool, cur = [], torch.cuda.graph_pool_handle(), [None]

        def begin():
            g = torch.cuda.CUDAGraph()
            cm = torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local")
            cm.__enter__()
            cur[0] = (g, cm)

        def end(buckets):
            g, cm = cur[0]
            cm.__exit__(None, None, None)
            graphs.append((g, buckets))
            cur[0] = None

        def cut(name):                                    # Engine.backward has joined its weight-gradient streams here
            if name == "srbs":
                end(["srbs"])
                begin()
        try:
            begin()
            self.per, _ = self._fwd_bwd(*self.static[:3], overlap=False, on_stage=cut if self.dp else None, **args)
            if self.dp:
                end(["infogen", "extra"])
                begin()
            self._update()
            end([])
        except Exception:
            if cur[0] is not None:
                try:
                    cur[0][1].__exit__(None, None, None)
                except Exception:
                    pass
            raise
        self.graph = graphs[:-1] if self.dp else graphs
        self.graph_update = graphs[-1][0] if self.dp else None

    def _reset_streams(self):
        self.eng.reset_streams()

    def step(self, lr_img, hr_img, tp, arc=None):
        """One optimisation step; returns the per-sample ImageLoss [B] (device tensor, no host sync).  With tsc=True
        `arc` [B] are the rotation angles in radians (drawn like the reference when omitted); the TRI_SSIM value of
        the step is left in `self.tssim` (device scalar)."""
        self.model.train()
        # the kernels read raw fp32 memory: normalise every input once, for the eager and the captured path alike (a sliced,
        # channels_last or half-precision batch would otherwise be misread by the TSC rotation / the student's input transform)
        lr_img = lr_img.detach().contiguous().float()
        hr_img = hr_img.detach().contiguous().float()
        if tp is not None:
            tp = tp.detach().float()                          # the [B,K,1,26] view of the recogniser's softmax: read through its strides
        if self.tsc:
            arc = (self.sample_arc(lr_img.shape[0]) if arc is None else arc).to(device=lr_img.device, dtype=torch.float32)
        if not self.use_graph:
            per, works = self._fwd_bwd(lr_img, hr_img, tp, overlap=self.dp, arc=arc)
            self._wait_all(works)
            self._update()
            self.steps += 1
            return per
        # hipGraph path: the step captured once, replayed per step
        if self.graph is None:
            try:
                self._capture(lr_img, hr_img, tp, arc)
            except Exception as e:
                if self.strict_graph or not self._is_capture_error(e):   # a kernel fault or an argument error is not a capture problem
                    raise
                import sys
                print(f"tatt_amd: hipGraph capture of the training step failed ({type(e).__name__}: {e}); continuing with eager "
                      "launches on fresh streams", file=sys.stderr, flush=True)
                self.use_graph, self.graph, self.graph_update = False, None, None
                torch.cuda.synchronize()
                self._reset_streams()
                return self.step(lr_img, hr_img, tp, arc=arc)
        for dst, src in zip(self.static, (lr_img, hr_img, tp) + ((arc,) if self.tsc else ())):
            dst.copy_(src)
        if not self.dp:
            self.graph[0][0].replay()
        else:
            works = []
            for g, buckets in self.graph:                     # segment k+1 runs while bucket k is on the wire
                g.replay()
                for name in buckets:
                    views = self.extra_grad_buffers() if name == "extra" else self._bucket_views(name)
                    works += [allreduce_mean(t, async_op=True) for t in views]
            self._wait_all(works)
            self.graph_update.replay()
        self.steps += 1
        return self.per


class TATTTrainer(SRTrainer):
    """The reference's whole training iteration (interfaces/super_resolution_eng.py:214-279) as one fused step:

        teacher prior   label_vecs_hr = softmax(CRNN_fixed(parse_crnn_data(HR)))            (:220-222, no gradient)
        student prior   label_vecs    = softmax(CRNN_student(parse_crnn_data(LR)))          (:226-228)
        L_SR (+ L_TSC)  SR(LR, label_vecs.detach()) -> ImageLoss*100 (+ TRI_SSIM term)      (:233-262)   [SRTrainer]
        L_TP            SemanticLoss(label_vecs, label_vecs_hr) * 100                       (:240)
        clip_grad_norm_(SR parameters, 0.25); Adam over two groups: SR lr, student lr_tpg   (:277-279, base.py:217-237)

    The student's gradient comes from L_TP only (the SR net takes its prior detached), so the two backward passes are
    independent; its Adam step is not clipped (the reference clips `model_list` = the SR net only)."""

    def __init__(self, model, teacher, student, lr=1e-3, lr_tpg=1e-4, **kw):
        super().__init__(model, lr=lr, **kw)
        self.teacher, self.student, self.lr_tpg = teacher.eval(), student, lr_tpg
        named = list(student.named_parameters())
        dev = self.flat_p.device
        n, offs = 0, {}
        for k, p in named:
            offs[k] = n
            n += (p.numel() + 63) // 64 * 64
        self.n_s = n
        self.s_p, self.s_g = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.s_m, self.s_v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.s_state, self.s_ws = torch.zeros(8, device=dev), torch.empty(1024, device=dev)
        self.G_s = {}
        for k, p in named:
            sl = slice(offs[k], offs[k] + p.numel())
            self.s_p[sl].copy_(p.data.reshape(-1).float())
            p.data = self.s_p[sl].view(p.shape)
            self.G_s[k] = self.s_g[sl].view(p.shape)
            p.grad = self.G_s[k]
        if self.dp:
            dist.broadcast(self.s_p, 0)
            broadcast_buffers(list(student.buffers()))
        self.loss_tp = None                                    # device scalar: SemanticLoss of the last step (L_TP = 100 x)

    def stateful_buffers(self):
        return super().stateful_buffers() + [b for n, b in self.student.named_buffers()
                                             if "running_" in n or n.endswith("num_batches_tracked")]

    def _reset_streams(self):
        super()._reset_streams()
        self._tp_stream = self._st_stream = None

    def _fwd_bwd(self, lr_img, hr_img, tp, overlap, arc=None, on_stage=None):
        """`tp` is ignored: the prior comes from the student CRNN."""
        L, B = self.L, lr_img.shape[0]
        K = self.student.nclass
        # Three independent strands: teacher prior (HR image only), the SR step (needs the student's prior), and L_TP with the
        # student's backward pass (needs both sets of logits, feeds nothing but the student's Adam step).  The teacher and the
        # L_TP strand run on a second stream -- a parallel branch of the captured graph -- beside the SR step.
        main = torch.cuda.current_stream()
        if getattr(self, "_tp_stream", None) is None:
            self._tp_stream = torch.cuda.Stream(device=lr_img.device)
        side = self._tp_stream
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            logits_t = self.teacher.forward_from_images(hr_img)                          # [26, B, K]
        # the student's forward pass runs on a stream of its own as well: the SR step only needs its prior at the TP interpreter,
        # after the STN head and block1 (Engine.forward(tp_ready=...))
        if getattr(self, "_st_stream", None) is None:
            self._st_stream = torch.cuda.Stream(device=lr_img.device)
        st = self._st_stream
        st.wait_stream(main)
        plan = self.student.plan()
        with torch.cuda.stream(st):
            x4 = torch.empty(B, 32, 100, 4, device=lr_img.device)
            L.crnn_prep(lr_img, x4, B, lr_img.shape[1], lr_img.shape[2], lr_img.shape[3], 32, 100)
            logits_s, ctx_s = plan.forward(x4, B, 100, True, want_ctx=True)
            T = logits_s.shape[0]
            probs = torch.empty(T, B, K, device=lr_img.device)
            L.softmax_rows(logits_s, probs, T * B, K)
            tp_ready = torch.cuda.Event()
            tp_ready.record(st)
        prior = probs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)                 # label_vecs_final: the [B,K,1,26] view
        # L_TP and the student's backward pass (everything they read was produced above and stays alive until the join)
        side.wait_stream(st)
        with torch.cuda.stream(side):
            self.s_g.zero_()
            self.loss_tp = torch.empty(1, device=lr_img.device)
            dz = torch.empty(T * B, K, device=lr_img.device)
            ws = torch.empty(L.semantic_loss_workspace(T * B), device=lr_img.device)
            L.semantic_loss(logits_s, logits_t, self.loss_tp, dz, 100.0, T * B, K, ws)
            plan.backward(ctx_s, dz.view(T, B, K), self.G_s)
        if on_stage is not None:
            # segmented capture (data parallel + hipGraph): every forked stream must be back on the capturing stream when a
            # segment ends, so the L_TP / student strand is joined before the first cut instead of at the end of the step
            # (once: in a later segment `side` is not part of the capture, and a capturing stream may not wait on outside work)
            cut, joined = on_stage, []

            def on_stage(name):
                if not joined:
                    main.wait_stream(side)
                    joined.append(True)
                cut(name)
        per, works = super()._fwd_bwd(lr_img, hr_img, prior, overlap, arc=arc, tp_ready=tp_ready, on_stage=on_stage)
        if on_stage is None:
            main.wait_stream(side)
        if overlap:
            works.append(allreduce_mean(self.s_g, async_op=True))
        return per, works

    def extra_grad_buffers(self):
        return [self.s_g]

    def check_chains(self):
        super().check_chains()
        self.teacher.plan().check_lstm_chain()
        self.student.plan().check_lstm_chain()

    def verify_replicas(self):
        n = super().verify_replicas()
        if self.dp:
            assert_replicas_identical(self.s_p, "student CRNN parameters")
        return n

    def _update(self):
        super()._update()
        self.L.clip_adam(self.s_p, self.s_g, self.s_m, self.s_v, self.n_s, self.lr_tpg, self.betas[0], self.betas[1], self.eps,
                         1e30, self.s_state, self.s_ws)                                  # max_norm = inf: the student is not clipped
