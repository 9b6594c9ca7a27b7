"""Training step of the SR generator on the HIP path (reference: interfaces/super_resolution_eng.py:214-279,
restricted to BASELINE config 2: one SR forward/backward, ImageLoss.mean()*100, clip_grad_norm_ 0.25, Adam).

Parameters and gradients live in two flat fp32 buffers laid out in the order the backward pass produces the
gradients (tail -> SRBs -> TP interpreter -> block1/STN), so that data-parallel training all-reduces four
contiguous buckets over RCCL while the rest of the backward pass is still running, and the fused
clip+Adam kernel touches every parameter once.  Every nn.Parameter of the module stays a view of the flat
buffer (state_dict / checkpoints are unaffected).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .lib import lib

BUCKETS = ["tail", "srbs", "infogen", "head"]


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


class SRTrainer:
    def __init__(self, model, lr=1e-3, betas=(0.5, 0.999), eps=1e-8, max_norm=0.25, loss_weight=(1.0, 1e-4),
                 use_graph=False, distributed=None, tsc=False, tssim_rotation_degree=10.0, training_stablize=False):
        self.model, self.L = model, lib()
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
            for b in model.buffers():
                dist.broadcast(b, 0)
        self.seed_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.eng.seed_dev = self.seed_dev
        self.use_graph = use_graph
        self.graph = None
        self.static = None
        self.per = None
        self.steps = 0

    # ------------------------------------------------------------------------------------------------------
    @staticmethod
    def loss_stablizing_mask(per, keep_proportion=0.7):
        """0/1 mask of TextSR.loss_stablizing (interfaces/super_resolution_eng.py:38-48), quirk included: everything
        from the ORIGINAL position sorted_ind[int(keep*B)] onward is dropped.  Device-side, no host sync."""
        B = per.shape[0]
        start = torch.argsort(per)[int(keep_proportion * B)]
        return (torch.arange(B, device=per.device) < start).to(per.dtype)

    def _fwd_bwd(self, lr_img, hr_img, tp, overlap, arc=None, tp_ready=None):
        L, B = self.L, lr_img.shape[0]
        self.flat_g.zero_()
        L.seed_bump(self.seed_dev)
        sr, ctx = self.eng.forward(lr_img, tp, True, 0, tp_ready=tp_ready)
        C, H, W = sr.shape[1:]
        per = torch.empty(B, device=sr.device)
        L.image_loss_fwd(sr, hr_img, per, B, C, H, W, self.loss_weight[0], self.loss_weight[1])
        dsr = torch.empty_like(sr)
        if self.training_stablize:
            mask = self.loss_stablizing_mask(per)
            per = per * mask
            L.image_loss_bwd(sr, hr_img, mask * (100.0 / B), 0.0, dsr, B, C, H, W, self.loss_weight[0], self.loss_weight[1])
        else:
            L.image_loss_bwd(sr, hr_img, None, 100.0 / B, dsr, B, C, H, W, self.loss_weight[0], self.loss_weight[1])
        if self.tsc:
            # D(F(Y)), F(D(Y)), D(X): the rotated SR output, the SR of the rotated input, the rotated HR image
            arc = arc.to(device=sr.device, dtype=torch.float32)
            d_sr, d_hr, d_lr = torch.empty_like(sr), torch.empty_like(hr_img), torch.empty_like(lr_img)
            L.rotate_fwd(sr, arc, d_sr, B, C, H, W)
            L.rotate_fwd(hr_img, arc, d_hr, B, C, H, W)
            L.rotate_fwd(lr_img, arc, d_lr, B, lr_img.shape[1], lr_img.shape[2], lr_img.shape[3])
            # seed 1: its own dropout masks; packed weights and the recurrent positional encoding (a function of the weights alone)
            # are those of the first pass, whose backward pass below also runs the one BPTT chain for both
            sr2, ctx2 = self.eng.forward(d_lr, tp, True, 1, wpack=ctx["wpack"], rpe_from=ctx)
            self.tssim = torch.empty(1, device=sr.device)
            L.tri_ssim_fwd(d_sr, sr2, d_hr, self.tssim, B, C, H, W, self.ws_ssim(B * C))
            g_rot, g_sr2 = torch.empty_like(sr), torch.empty_like(sr)
            L.tri_ssim_bwd(d_sr, sr2, d_hr, -10.0, g_rot, g_sr2, 0, B, C, H, W)      # loss_tssim = (1 - mean) * 10
            L.rotate_bwd(arc, g_rot, dsr, B, C, H, W)                                # += onto the image-loss gradient
            self.eng.backward(ctx2, g_sr2, self.G)
        works = []

        def on_stage(name):
            if overlap:
                a, b = self.bucket_range[BUCKETS.index(name)]
                works.append(allreduce_mean(self.flat_g[a:b], async_op=True))

        self.eng.backward(ctx, dsr, self.G, on_stage=on_stage)
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

    def step(self, lr_img, hr_img, tp, arc=None):
        """One optimisation step; returns the per-sample ImageLoss [B] (device tensor, no host sync).  With tsc=True
        `arc` [B] are the rotation angles in radians (drawn like the reference when omitted); the TRI_SSIM value of
        the step is left in `self.tssim` (device scalar)."""
        self.model.train()
        hr_img = hr_img.contiguous().float()
        if self.tsc:
            arc = (self.sample_arc(lr_img.shape[0]) if arc is None else arc).to(device=lr_img.device, dtype=torch.float32)
        if not self.use_graph:
            per, works = self._fwd_bwd(lr_img, hr_img, tp, overlap=self.dp, arc=arc)
            for w in works:
                if w is not None:
                    w.wait()
            self._update()
            self.steps += 1
            return per
        # hipGraph path: forward+backward (+ optimiser when single-GPU) captured once, replayed per step
        if self.graph is None:
            self.static = [lr_img.clone().contiguous(), hr_img.clone(), tp.clone()] + ([arc.clone()] if self.tsc else [])
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                       # warm-up outside capture (allocator, lazy module loads)
                self._fwd_bwd(*self.static[:3], overlap=False, arc=self.static[3] if self.tsc else None)
            torch.cuda.current_stream().wait_stream(s)
            self.seed_dev.zero_()
            # the warm-up touched BatchNorm running stats: harmless for a benchmark, noted for training use
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls events of earlier collectives (the parameter broadcast, the previous
            # step's all-reduce); under the default global mode such a poll from ANOTHER thread during capture aborts the
            # process (seen once in tests/test_gpu_dp.py when the capture began right after an eager data-parallel step)
            if self.dp:
                torch.cuda.synchronize()
            try:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.per, _ = self._fwd_bwd(*self.static[:3], overlap=False, arc=self.static[3] if self.tsc else None)
                    if not self.dp:
                        self._update()
            except Exception as e:                            # keep training: same step, launched eagerly (and say so loudly)
                import sys
                print(f"tatt_amd: hipGraph capture of the training step failed ({type(e).__name__}: {e}); continuing with eager "
                      "launches", file=sys.stderr, flush=True)
                self.use_graph, self.graph = False, None
                torch.cuda.synchronize()
                return self.step(lr_img, hr_img, tp, arc=arc)
        for dst, src in zip(self.static, (lr_img, hr_img, tp) + ((arc,) if self.tsc else ())):
            dst.copy_(src)
        self.graph.replay()
        if self.dp:
            works = [allreduce_mean(self.flat_g[a:b], async_op=True) for a, b in self.bucket_range]
            works += [allreduce_mean(t, async_op=True) for t in self.extra_grad_buffers()]
            for w in works:
                if w is not None:
                    w.wait()
            self._update()
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
            for b in student.buffers():
                dist.broadcast(b, 0)
        self.loss_tp = None                                    # device scalar: SemanticLoss of the last step (L_TP = 100 x)

    def _fwd_bwd(self, lr_img, hr_img, tp, overlap, arc=None):
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
        per, works = super()._fwd_bwd(lr_img, hr_img, prior, overlap, arc=arc, tp_ready=tp_ready)
        main.wait_stream(side)
        if overlap:
            works.append(allreduce_mean(self.s_g, async_op=True))
        return per, works

    def extra_grad_buffers(self):
        return [self.s_g]

    def _update(self):
        super()._update()
        self.L.clip_adam(self.s_p, self.s_g, self.s_m, self.s_v, self.n_s, self.lr_tpg, self.betas[0], self.betas[1], self.eps,
                         1e30, self.s_state, self.s_ws)                                  # max_norm = inf: the student is not clipped
