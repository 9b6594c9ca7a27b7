"""Launch plan of the TATT SR network on the HIP kernels: explicit forward and hand-derived backward.

Host side only: this file decides which C-ABI entry point (include/tatt_hip.h) runs on which buffer; torch is
used for device memory and the current stream.  Activations are NHWC fp32 ("rows x channels", row = (b, y, x)),
which makes the image tokens of the TP interpreter ([B*1024, 64]), the GRU sequences and the conv GEMMs all
views of the same buffers.  Reference call stack: SURVEY.md §3.2 (model/tsrn.py:648-684).
"""
from __future__ import annotations

import ctypes
import os

import torch

from .lib import lib

DROP_SITES = {
    "pe": 1,
    "enc_attn": 2, "enc_drop1": 3, "enc_ffn": 4, "enc_drop2": 5,
    "dec0_attn": 6, "dec0_drop2": 7, "dec0_ffn": 8, "dec0_drop3": 9,
    "dec1_attn": 10, "dec1_drop2": 11, "dec1_ffn": 12, "dec1_drop3": 13,
}
STN_STAGES = [(0, 4, 32, (2, 2)), (2, 32, 64, (2, 2)), (4, 64, 128, (2, 2)), (6, 128, 256, (2, 2)),
              (8, 256, 256, (1, 2)), (10, 256, 256, None)]
BN_GROUPED = os.environ.get("TATT_BN_GROUPED", "1") != "0"   # 0: the two passes of a batched TSC step get a BatchNorm final + apply launch each (A/B)
RPE_CHAIN = os.environ.get("TATT_RPE_CHAIN", "1") != "0"   # 0: one launch per step of the RPE recurrence (A/B runs)
DEC_FUSED = os.environ.get("TATT_DEC_FUSED", "1") != "0"   # 0: the decoder as one launch per projection / attention / LayerNorm (A/B runs)
WS_FLOATS = 48 * 1024 * 1024
RPE_FORK_EARLY = os.environ.get("TATT_RPE_FORK_EARLY", "1") != "0"   # 0: the RPE chain forks after the decoder's reductions (A/B runs)
RPE_TAIL_SPLIT = os.environ.get("TATT_RPE_TAIL_SPLIT", "1") != "0"   # 0: the whole tail of the RPE chain on the chain's stream (A/B runs)
RPE_TAIL_ON_SIDE = os.environ.get("TATT_RPE_TAIL_ON_SIDE", "1") != "0"   # 0: the main stream joins the RPE chain before the STN head's backward pass (A/B runs)
ENC_BWD_FORK = os.environ.get("TATT_ENC_BWD_FORK", "1") != "0"   # 0: the text branch's backward pass on the main stream (A/B runs)
RPE_FINISH_EARLY = os.environ.get("TATT_RPE_FINISH_EARLY", "1") != "0"   # 0: join the RPE chain after the STN head's backward pass (A/B runs)
STN_FC_SPLITK = int(os.environ.get("TATT_STN_FC_SPLITK", "8"))   # split-K factor of the STN head's fc products (1: none; A/B runs)
STN_SMALL = os.environ.get("TATT_STN_SMALL", "1") != "0"   # 0: the STN head's small maps through the generic BatchNorm / pool launches (A/B runs)
GRU_RES = os.environ.get("TATT_GRU_RES", "1") != "0"       # 0: "x + residual" of an SRB as an element-wise launch of its own (A/B runs)
BF16_STORE = os.environ.get("TATT_BF16_STORE", "1") != "0"   # 0: the bf16 step keeps every tensor fp32 in memory (A/B runs; bit-identical results)
WGRAD192_WGS = int(os.environ.get("TATT_WGRAD192_WGS", "255"))
WGRAD_WGS = int(os.environ.get("TATT_WGRAD_WGS", "480"))   # workgroups of a convolution weight-gradient launch (split-K factor = this / tiles)
NODROP = (0, 0, 1.0, None)


class Ctx:
    """Saved tensors of one forward call."""

    def __init__(self):
        self.t = {}

    def __getitem__(self, k):
        return self.t[k]

    def __setitem__(self, k, v):
        self.t[k] = v

    def get(self, k, default=None):
        return self.t.get(k, default)


class Engine:
    def __init__(self, module):
        self.m = module
        self.L = lib()
        self.dev = None
        self.ws = None
        self._names = [n for n, _ in module.named_parameters()]
        self._ptrs = None
        self.P = {}
        self.debug = None         # set to a dict to stash intermediate gradients (tests/debug_grads.py)
        self.seed_dev = None      # optional uint32 device tensor added to the dropout seed (graph-captured training)
        self.wpack = {}
        self._side = None         # side stream of the RPE recurrence (see forward)
        self._enc = None          # stream of the TP interpreter's text branch (forward)
        self._nbt = []            # BatchNorm num_batches_tracked buffers touched by the running forward pass
        # Weight-gradient kernels are leaves of the backward pass: nothing downstream reads dW before the optimiser.  In "batch"
        # mode they are collected and issued in groups on their own stream (a parallel branch of the captured hipGraph), so that
        # their ramp-up and under-filled tails overlap with the dgrad chain; see _wgrad().  "off" = in line on the main stream.
        # Measured at B=64 (ms/step): off 12.82, one fork per kernel 12.92 (each cross-stream dependency costs more than it
        # hides), one fork per half SRB 12.23.  Upper bound if the 3.8 ms of weight-gradient kernels were free: 8.99.
        # Matrix-core precision of THIS engine's launches (a per-call flag of the C-ABI, no process state): 0 = exact fp32 operands,
        # TATT_BF16 = operands rounded to bf16 into the MFMA, fp32 accumulate.  Set per forward call from module.matmul_precision
        # and kept in the call's context for its backward pass.  self.pf: the 3x3 convolutions of the trunk (SRBs, block7, the
        # up-sampling conv; forward, input and weight gradients) -- 85 % of the step's FLOPs, the launches that are matrix-core bound.
        # TATT_F32X3 (matmul_precision "f32x3", opt-in) = fp32 products from three bf16 pieces per operand: same accuracy as fp32 MFMA,
        # so it goes on every launch that has such a kernel (pf and pl; `fp32=True` call sites stay on fp32 MFMA all the same).
        # self.pl: Linear / 1x1 / GRU-projection GEMMs: never TATT_BF16 (0 in fp32 mode, TATT_F32X3 in the other two).  With K = 64..192 they are bound by moving their fp32 tensors, so
        # bf16 operands buy no time there, and measured at B=4 they carry most of the mode's error (output max-abs 1.4e-2 with them
        # in bf16, 6e-3 without).  The STN head, the TPS matrices, the recurrent positional encoding and the output convolution are
        # fp32 in both modes too: control points are scaled x64 into pixel coordinates by grid_sample, the RPE feeds every query,
        # the output conv feeds the sigmoid directly -- their rounding error is amplified instead of averaged out.
        # self.px: what a call site pinned with fp32=True gets -- 0 (fp32 MFMA) unless the mode itself has fp32 accuracy (f32x3).
        self.pf, self.pl, self.px = 0, 0, 0
        # bf16 STORAGE (bf16 mode, training): a tensor whose every reader is a bf16 matrix-core launch is written ONCE as bf16 by its
        # producer and read as bf16 (flags TATT_X_BF16 / TATT_DY_BF16) -- per SRB mish(bn1(conv1)) (read by conv2 and conv2's weight
        # gradient) and the gradients d conv2-out / d conv1-out (read by that convolution's dgrad and wgrad), block7's output + skip (read
        # by the up-sampling convolution and its weight gradient), d block7-conv-out and the [R,256] gradient of the up-sampling
        # convolution's output.  The readers rounded exactly these values to bf16 on their way into LDS before, so the step's numbers
        # do not change by a bit (tests/test_gpu_bf16.py); what changes is 0.5 GB less HBM traffic per step at B = 64.
        self.store16 = False
        self._rpe_chain_ok = None
        self.dec_fused = DEC_FUSED   # the TP interpreter's decoder as tatt_dec_fwd / tatt_dec_bwd (False: one launch per op; tests compare the two)
        self._rpe_sync = self._rpe_sync_b = None
        self.ws_side = self.ws_enc = self.ws_side_b = None
        self._strand, self._wgrad_ws = False, None
        self.wgrad_mode = os.environ.get("TATT_WGRAD_STREAM", "batch")
        if self.wgrad_mode not in ("batch", "off"):
            raise RuntimeError("TATT_WGRAD_STREAM must be 'batch' or 'off'")
        self._wside, self.ws_w, self._w_used = None, None, set()
        self._w_keep, self._w_pending, self._w_defer = [], set(), []

    # ---- parameters --------------------------------------------------------------------------------
    def param_list(self):
        return [p for _, p in self.m.named_parameters()]

    def refresh(self):
        ps = dict(self.m.named_parameters())
        ptrs = tuple(p.data_ptr() for p in ps.values())
        if ptrs != self._ptrs:
            self.P = {k: v.data for k, v in ps.items()}
            for k, v in self.P.items():
                if not v.is_contiguous() or v.dtype != torch.float32:
                    raise RuntimeError(f"parameter {k} must be contiguous fp32")
            self._ptrs = ptrs
        for k, v in self.m.named_buffers():
            if v.is_floating_point() and v.dtype != torch.float32:
                raise RuntimeError(f"buffer {k} must be fp32")
            if not v.is_contiguous():
                if "running_" in k:
                    raise RuntimeError(f"buffer {k} must be contiguous (it is updated in place)")
                v = v.contiguous()      # constant buffers (e.g. a column-major tps.inverse_kernel from an old checkpoint)
            self.P[k] = v
        dev = next(iter(ps.values())).device
        if self.ws is None or self.dev != dev:
            self.dev = dev
            self.ws = torch.empty(WS_FLOATS, device=dev, dtype=torch.float32)

    def side_stream(self):
        if os.environ.get("TATT_NO_FORK") == "1":             # experiment knob: the RPE / text branches on the main stream
            return torch.cuda.current_stream()
        if self._side is None or self._side.device != self.dev:
            self._side = torch.cuda.Stream(device=self.dev)
        return self._side

    # ---- weight-gradient stream ------------------------------------------------------------------------------
    def _wgrad(self, fn, dys, xs=()):
        """Queue fn(workspace) -- weight-gradient kernels reading the tensors `dys` (gradients) and `xs` (saved activations) --
        for the weight-gradient stream; _w_flush() launches the queue there, ordered after everything the main stream has
        issued by then.  The operands are kept alive until the next join (the caching allocator must not hand their memory
        to a later main-stream tensor), and the gradients are remembered by storage so that _guard() can order a later
        in-place update of one of them after the queued reads."""
        if self._wgrad_ws is not None:                        # a side strand launching its own weight gradients in place (RPE tail)
            return fn(self._wgrad_ws)
        if self.wgrad_mode != "batch":
            return fn(self.ws)
        self._w_defer.append(fn)                              # launched on the weight-gradient stream at the next _w_flush()
        self._w_keep.extend(dys)
        self._w_keep.extend(xs)
        self._w_pending.update(t.untyped_storage().data_ptr() for t in dys)

    def _w_flush(self):
        """One fork for everything queued so far (a cross-stream dependency per kernel costs more than it hides)."""
        if not self._w_defer:
            return
        main = torch.cuda.current_stream()
        if self._wside is None or self._wside[0].device != self.dev:
            # 1 or 2: with the main chain and the RPE chain that is at most four concurrent branches, the number of internal
            # queues the HIP runtime runs a graph on (forcing more queues crashed hipStreamEndCapture on this ROCm)
            n = int(os.environ.get("TATT_WGRAD_STREAMS", "2"))
            if n not in (1, 2):
                raise RuntimeError(f"TATT_WGRAD_STREAMS must be 1 or 2 (got {n})")
            self._wside = [torch.cuda.Stream(device=self.dev) for _ in range(n)]
            self.ws_w = [torch.empty(WS_FLOATS, device=self.dev, dtype=torch.float32) for _ in range(n)]
            self._w_rr = 0
        # batches alternate between the weight-gradient streams (no two kernels of a pass write the same gradient tensor, so
        # the batches are independent of one another); each stream has its own split-K / slab workspace
        i = self._w_rr % len(self._wside)
        self._w_rr += 1
        self._wside[i].wait_stream(main)
        with torch.cuda.stream(self._wside[i]):
            for fn in self._w_defer:
                fn(self.ws_w[i])
        self._w_defer = []
        self._w_used.add(i)

    def _w_join(self):
        """Main stream waits for the weight-gradient stream (before a bucket's all-reduce, the optimiser, or a guarded write)."""
        self._w_flush()
        if self._wside is not None and (self._w_keep or self._w_pending):
            for i in sorted(self._w_used):
                torch.cuda.current_stream().wait_stream(self._wside[i])
            self._w_used = set()
            self._w_keep, self._w_pending, self._w_defer = [], set(), []

    def _guard(self, t):
        """Call before the main stream writes `t` in place: if a pending weight-gradient kernel still reads it, join first."""
        if self._w_pending and t is not None and t.untyped_storage().data_ptr() in self._w_pending:
            if self._strand:
                raise RuntimeError("a side strand of the backward pass would have to join the weight-gradient streams (guarded in-place write)")
            self._w_join()

    def rpe_chain(self):
        """Single-launch RPE chains: on unless TATT_RPE_CHAIN=0 or the device cannot hold them (tatt_rpe_chain_supported: another
        partition mode, fewer CUs per XCD) -- then the recurrence runs as one launch per step."""
        if self._rpe_chain_ok is None:
            self._rpe_chain_ok = RPE_CHAIN and bool(self.L.rpe_chain_supported())
        return self._rpe_chain_ok

    def check_rpe_chain(self):
        """Raise if a single-launch RPE chain (tatt_rpe_chain_fwd / _bwd) of ANY pass since the engine was built reported a reader
        that gave up waiting (sync word 1 is sticky: the calls never clear it; that pass's numbers were invalid).  Synchronises;
        call it at check points, not per step."""
        for name in ("_rpe_sync", "_rpe_sync_b"):
            t = getattr(self, name, None)
            if t is not None and int(t[1].item()) != 0:
                raise RuntimeError(f"RPE chain kernel ({name}) gave up waiting for its predecessor step (flag {int(t[1].item())}): results "
                                   "are invalid; set TATT_RPE_CHAIN=0 to run the recurrence as one launch per step")

    def reset_streams(self):
        """Drop every forked stream and queued weight-gradient launch (after a failed hipGraph capture the old streams may be
        left in an invalidated capture state; fresh ones are created on demand)."""
        self._side = self._enc = self._wside = self.ws_w = None
        self._w_used = set()
        self._w_keep, self._w_pending, self._w_defer = [], set(), []

    def enc_stream(self):
        if os.environ.get("TATT_NO_FORK") == "1":
            return torch.cuda.current_stream()
        if self._enc is None or self._enc.device != self.dev:
            self._enc = torch.cuda.Stream(device=self.dev)
        return self._enc

    def new(self, *shape):
        return torch.empty(shape, device=self.dev, dtype=torch.float32)

    def new16(self, *shape):
        return torch.empty(shape, device=self.dev, dtype=torch.bfloat16)

    @staticmethod
    def _x16(t):
        return 8 if t.dtype == torch.bfloat16 else 0          # TATT_X_BF16

    def zeros(self, *shape):
        return torch.zeros(shape, device=self.dev, dtype=torch.float32)

    # ---- thin op wrappers --------------------------------------------------------------------------------
    def _splitk(self, M, N, K):
        tiles = -(-M // 128) * -(-N // 64)
        s = max(1, min(K // 256, -(-768 // tiles)))
        while s > 1 and M * N * s > WS_FLOATS:
            s //= 2
        return max(1, s)

    def linear(self, x, W, b, out, R, N, K, ldx=None, ldw=None, ldc=None, acc=0, relu=0, splitk=1, fp32=False, ws=None):
        """out[R,N] (+)= x[R,K] @ W[N,K]^T + b   (splitk > 1 for the few-row, long-K products of the RPE; ws: the split-K workspace
        when the call runs on a stream beside the main one, which owns self.ws)"""
        self.L.gemm(x, W, out, b, R, N, K, ldx or K, 1, ldw or K, 1, ldc or N, 1, 0, 0, 0, acc, relu | (self.px if fp32 else self.pl),
                    splitk, (ws if ws is not None else self.ws) if splitk > 1 else None)

    def linear_dx(self, dy, W, dx, R, N, K, lddy=None, ldw=None, lddx=None, acc=0, splitk=1, fp32=False):
        """dx[R,K] (+)= dy[R,N] @ W[N,K]"""
        self._guard(dx)
        self.L.gemm(dy, W, dx, None, R, K, N, lddy or N, 1, 1, ldw or K, lddx or K, 1, 0, 0, 0, acc, self.px if fp32 else self.pl, splitk,
                    self.ws if splitk > 1 else None)

    def _tps_matrix(self, H, W):
        """target_coordinate_repr @ inverse_kernel[:, :20] [H*W, 20] of the TPS grid (model/tps_spatial_transformer.py:99-102): a
        product of two registered buffers, so it is formed once per engine and buffer storage, not once per forward pass."""
        P = self.P
        a, b = P["tps.target_coordinate_repr"], P["tps.inverse_kernel"]
        key = (a.data_ptr(), a._version, b.data_ptr(), b._version, H, W)
        if getattr(self, "_tps_key", None) != key:
            Tm = self.new(H * W, 20)
            self.L.gemm(a, b, Tm, None, H * W, 20, 23, 23, 1, 1, 23, 20, 1, 0, 0, 0, 0, 0, 1, None)
            if torch.cuda.is_current_stream_capturing():        # a captured launch has not run yet: nothing to remember
                return Tm
            self._tps_key, self._tps_Tm = key, Tm
        return self._tps_Tm

    def linear_dw(self, dy, x, dW, R, N, K, lddy=None, ldx=None, lddw=None, fp32=False):
        """dW[N,K] += dy[R,N]^T @ x[R,K]  (always accumulates into the gradient buffer)"""
        sk = self._splitk(N, K, R)
        fl = self.px if fp32 else self.pl
        self._wgrad(lambda ws: self.L.gemm(dy, x, dW, None, N, K, R, 1, lddy or N, 1, ldx or K, lddw or K, 1, 0, 0, 0, 1, fl, sk,
                                           ws if sk > 1 else None), [dy], [x])

    def linear_dwb(self, dy, x, dW, db, R, N, K, lddy=None, ldx=None, lddw=None):
        """dW[N,K] += dy^T x and db[N] += colsum(dy): one streaming pass (tatt_tn_small) when the output is small."""
        if N <= 96 and K <= 64:
            self._wgrad(lambda ws: self.L.tn_small(dy, lddy or N, x, ldx or K, dW, lddw or K, db, N, K, R, ws), [dy], [x])
            return
        self.linear_dw(dy, x, dW, R, N, K, lddy=lddy, ldx=ldx, lddw=lddw)
        if db is not None:
            self.colsum(dy, R, N, db, ld=lddy)

    def linear_dwb_group(self, probs):
        """Several linear_dwb problems (dy, x, dW, db, R, N, K, lddy, ldx, lddw[, dy2, lddy2, nsplit]) of the small kind (N <= 96,
        K <= 64) in one launch pair (tatt_tn_small_group): fewer, longer row ranges per workgroup and one slab reduction for the
        lot.  dy2: columns >= nsplit of dy come from that second matrix."""
        assert all(p[5] <= 96 and p[6] <= 64 for p in probs)
        for i in range(0, len(probs), 8):
            ch = [tuple(p) + (None, None, None) * (len(p) == 10) for p in probs[i:i + 8]]
            n = len(ch)
            VP, LG, IN = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
            two = any(p[10] is not None for p in ch)
            args = (n, VP(*[p[0].data_ptr() for p in ch]), LG(*[p[7] or p[5] for p in ch]),
                    VP(*[p[1].data_ptr() for p in ch]), LG(*[p[8] or p[6] for p in ch]),
                    VP(*[p[2].data_ptr() for p in ch]), LG(*[p[9] or p[6] for p in ch]),
                    VP(*[None if p[3] is None else p[3].data_ptr() for p in ch]),
                    IN(*[p[5] for p in ch]), IN(*[p[6] for p in ch]), IN(*[p[4] for p in ch]),
                    VP(*[None if p[10] is None else p[10].data_ptr() for p in ch]) if two else None,
                    LG(*[p[11] or 0 for p in ch]) if two else None, IN(*[p[12] or 0 for p in ch]) if two else None)
            self._wgrad(lambda ws, args=args: self.L.tn_small_group(*args, ws), [p[0] for p in ch] + [p[10] for p in ch if p[10] is not None],
                        [p[1] for p in ch])

    def colsum(self, x, R, C, out, ld=None):
        if x.dtype != torch.float32:
            raise RuntimeError("colsum: fp32 input (a tensor the bf16 step stores as bf16 has no fp32 reader)")
        self._wgrad(lambda ws: self.L.colsum(x, ld or C, R, C, out, 1, ws), [x])

    def _skinny(self, x, W, b, out, B, H, Wd, Cin, Cout, k, wstr, acc, fp32=False):
        wsz = self.L.conv_skinny_workspace(B, H, Wd, Cin, Cout, k, k)
        z = self.new(wsz)                                      # 9x9, Cout = 4 (direct kernel): the packed weights; else planar Z [(tap,co)][pixel]
        self.L.conv_skinny_fwd(x, W, b, out, B, H, Wd, Cin, Cout, k, k, Cin, Cout, *wstr, acc | (self.px if fp32 else self.pf), z)

    def _prepack(self, use_stn, training=False):
        """Pack every convolution weight of the network for this forward call with ONE launch (tatt_conv_pack_w_group)."""
        m, P = self.m, self.P
        names = ["block1.0.weight"]
        for i in range(m.srb_nums):
            names += [f"block{i + 2}.conv1.weight", f"block{i + 2}.conv2.weight"]
        names += [f"block{m.srb_nums + 2}.0.weight", f"block{m.srb_nums + 3}.0.conv.weight"]
        if training:   # the 9x9 output convolution's input gradient runs on the tiled kernel: its pack would otherwise be a launch of
            names.append(f"block{m.srb_nums + 3}.1.weight")   # its own at the head of the backward pass, alone on the GPU
        if use_stn:
            names += [f"stn_head.stn_convnet.{idx}.0.weight" for idx, *_ in STN_STAGES]
        Ws = [P[n] for n in names if P[n].data_ptr() not in self.wpack]
        if not Ws:
            return
        tot = sum(w.numel() for w in Ws)
        buf = self.new(2 * tot)
        outs, o = [], 0
        for w in Ws:
            outs.append((buf[o:o + w.numel()], buf[tot + o:tot + o + w.numel()]))
            o += w.numel()
        for c0 in range(0, len(Ws), 32):
            ws_, os_ = Ws[c0:c0 + 32], outs[c0:c0 + 32]
            n = len(ws_)
            VP, IN = ctypes.c_void_p * n, ctypes.c_int * n
            self.L.conv_pack_w_group(n, VP(*[w.data_ptr() for w in ws_]), VP(*[a.data_ptr() for a, _ in os_]),
                                     VP(*[b.data_ptr() for _, b in os_]), IN(*[w.shape[0] for w in ws_]), IN(*[w.shape[1] for w in ws_]),
                                     IN(*[w.shape[2] * w.shape[3] for w in ws_]))
        for w, pk in zip(Ws, outs):
            self.wpack[w.data_ptr()] = pk

    def _packed(self, W, Cout, Cin, k):
        """(Wf [co][tap][ci], Wd [ci][flipped tap][co]) of an OIHW weight, packed once per forward call (tatt_conv_pack_w)."""
        key = W.data_ptr()
        pk = self.wpack.get(key)
        if pk is None:
            pk = (self.new(W.numel()), self.new(W.numel()))
            self.L.conv_pack_w(W, pk[0], pk[1], Cout, Cin, k, k)
            self.wpack[key] = pk
        return pk

    @staticmethod
    def _conv_splitk(M, N, K):
        """Split the conv's K loop when it has too few output tiles to fill 256 CUs (the STN head's small maps)."""
        tiles = -(-M // 128) * -(-N // 64)
        if tiles >= 128 or K < 256:
            return 1
        sk = min(K // 64, -(-384 // tiles))
        while sk > 1 and M * N * sk > WS_FLOATS:
            sk //= 2
        return max(1, sk)

    def conv(self, x, W, b, out, B, H, Wd, Cin, Cout, k, acc=0, relu=0, fp32=False):
        T = k * k
        if Cout in (4, 8, 16) and not relu:
            return self._skinny(x, W, b, out, B, H, Wd, Cin, Cout, k, (Cin * T, T, 0, 1), acc, fp32)
        self.L.conv_fwd_splitk(x, self._packed(W, Cout, Cin, k)[0], b, out, B, H, Wd, Cin, Cout, k, k, Cin, Cout,
                               T * Cin, 1, 0, Cin, acc, relu | (self.px if fp32 else self.pf) | self._x16(x), self._conv_splitk(B * H * Wd, Cout, T * Cin),
                               self.ws)

    def conv_dgrad(self, dy, W, dx, B, H, Wd, Cin, Cout, k, acc=0, fp32=False):
        """dx[B,H,W,Cin] (+)= conv^T(dy[B,H,W,Cout]);  W is the forward OIHW weight [Cout,Cin,k,k]."""
        T = k * k
        self._guard(dx)
        if Cin in (4, 8, 16):
            return self._skinny(dy, W, None, dx, B, H, Wd, Cout, Cin, k, (T, Cin * T, T - 1, -1), acc, fp32)
        self.L.conv_fwd_splitk(dy, self._packed(W, Cout, Cin, k)[1], None, dx, B, H, Wd, Cout, Cin, k, k, Cout, Cin,
                               T * Cout, 1, 0, Cout, acc, (self.px if fp32 else self.pf) | self._x16(dy), self._conv_splitk(B * H * Wd, Cin, T * Cout), self.ws)

    def conv_wgrad(self, x, dy, dW, db, B, H, Wd, Cin, Cout, k, fp32=False, bias_grad=True):
        """bias_grad=False: the convolution feeds a training-mode BatchNorm, whose input gradient sums to zero over the rows for
        every channel (sum dx = -gamma*invstd*mean(g*xhat)*sum(xhat), and sum(xhat) = 0): the bias gradient is exactly 0 and the
        16.8 MB column-sum pass that would produce rounding noise around it is not launched (db is left as it is: += 0)."""
        T = k * k
        if Cout <= 16 and Cin <= 64:                           # M = T*Cout (im2col of dY), N = Cin (see tatt_conv_wgrad)
            tiles = -(-(T * Cout) // 128)
        elif Cout <= 16:                                       # roles swapped: M = Cin, N = T*Cout
            tiles = -(-Cin // 128) * -(-(T * Cout) // 64)
        elif Cout <= 64:                                       # M = T*Cin, N = Cout
            tiles = -(-(T * Cin) // 128)
        else:                                                  # M = Cout, N = T*Cin
            tiles = -(-Cout // 128) * -(-(T * Cin) // 64)
        sk = -(-WGRAD_WGS // tiles)
        if WGRAD192_WGS and 16 < Cout <= 64 and (T * Cin) % 192 == 0 and Wd % 32 == 0 and (self.px if fp32 else self.pf) == 0:
            sk = WGRAD192_WGS // (T * Cin // 192)              # wgrad192_kernel (fp32 MFMA): one twelve-wave workgroup per CU, never 257
        sk = max(1, min((B * H * Wd) // 256, sk))        # 3x3 64->64: 96 slabs; in the step 7.10 ms against 7.20 with 154 (standalone 57.7 / 61.0 us, and a shorter slab sum)
        while sk > 1 and Cout * T * Cin * sk > WS_FLOATS:
            sk //= 2
        fl = 1 | (self.px if fp32 else self.pf) | self._x16(x) | 2 * self._x16(dy)   # TATT_X_BF16 = 8, TATT_DY_BF16 = 16
        self._wgrad(lambda ws: self.L.conv_wgrad(x, dy, dW, B, H, Wd, Cin, Cout, k, k, Cin, Cout, sk, ws, fl), [dy], [x])
        if bias_grad:
            self.colsum(dy, B * H * Wd, Cout, db)

    def bn(self, pre, x, R, C, train, act, out=None, res=None):
        P = self.P
        stat = self.new(2, C)
        if train:
            self.L.bn_stats(x, C, R, C, 1e-5, 0.1, stat, P[pre + "running_mean"], P[pre + "running_var"], self.ws)
            self._nbt.append(P[pre + "num_batches_tracked"])    # bumped together at the end of forward (one launch, not 18)
        else:
            self.L.bn_stat_from_running(P[pre + "running_mean"], P[pre + "running_var"], C, 1e-5, stat)
        y = out if out is not None else self.new(R, C)
        self.L.bn_apply(x, C, stat, P[pre + "weight"], P[pre + "bias"], res, C, y, C, R, C, act)
        return y, stat

    # ---- BatchNorm fused into the producing launch (tatt_conv_fwd_bn / tatt_gemm_cat_bnbwd): the row reductions of both passes run
    # in the epilogue of the convolution / GEMM that writes the tensor, one (mean, M2) or (sum g, sum g xhat) pair per 128-row tile
    def conv_bn(self, pre_bn, x, W, b, B, H, Wd, Cin, Cout, k, train, act, res=None, ng=1, out16=False):
        """y = act(BatchNorm(conv(x))) (+ res).  Training mode: batch statistics from the convolution's epilogue.
        Returns (conv output, y, stat).  ng = 2: the B images are two passes of B/2 (the batched TSC step): ONE convolution launch,
        BatchNorm statistics / running-statistic updates / apply per pass, in pass order; stat is then [2][2][Cout].
        out16: y is read by bf16 matrix-core launches only -- stored as bf16 when the step keeps such tensors in bf16 (self.store16)."""
        P, R, T = self.P, B * H * Wd, k * k
        c = self.new(R, Cout)
        if not train:
            if ng != 1:
                raise RuntimeError("two-pass batches are a training-mode construct")
            self.conv(x, W, b, c, B, H, Wd, Cin, Cout, k)
            y, stat = self.bn(pre_bn, c, R, Cout, False, act, res=res)
            return c, y, stat
        Rg = R // ng
        if Rg % 128:
            raise RuntimeError("conv_bn: rows per pass must be a multiple of 128")
        nblk = Rg // 128
        part = self.new(ng * nblk * 2 * Cout)
        self.L.conv_fwd_bn(x, self._packed(W, Cout, Cin, k)[0], b, c, B, H, Wd, Cin, Cout, k, k, Cin, Cout, T * Cin, 1, 0, Cin, self.pf | self._x16(x), 1,
                           None, 0, None, None, None, 0, part)
        stat = self.new(ng, 2, Cout)
        y = self.new16(R, Cout) if (out16 and self.store16) else self.new(R, Cout)
        apply = self.L.bn_apply_bf16 if y.dtype == torch.bfloat16 else self.L.bn_apply
        if ng > 1 and BN_GROUPED and Rg % (8192 // Cout if Cout <= 1024 else 8) == 0:
            # both passes' finals in one launch (running statistics updated pass after pass), both applies in one launch
            self.L.bn_stats_from_partial_groups(part, nblk, 128, Rg, Cout, ng, 1e-5, 0.1, stat, P[pre_bn + "running_mean"], P[pre_bn + "running_var"])
            self._nbt += [P[pre_bn + "num_batches_tracked"]] * ng
            (self.L.bn_apply_bf16_groups if y.dtype == torch.bfloat16 else self.L.bn_apply_groups)(
                c, Cout, stat, P[pre_bn + "weight"], P[pre_bn + "bias"], res, Cout, y, Cout, R, Cout, act, Rg)
            return c, y, stat
        for g in range(ng):
            self.L.bn_stats_from_partial(part[g * nblk * 2 * Cout:], nblk, 128, Rg, Cout, 1e-5, 0.1, stat[g], P[pre_bn + "running_mean"],
                                         P[pre_bn + "running_var"])
            self._nbt.append(P[pre_bn + "num_batches_tracked"])
            apply(c[g * Rg:], Cout, stat[g], P[pre_bn + "weight"], P[pre_bn + "bias"], None if res is None else res[g * Rg:], Cout,
                  y[g * Rg:], Cout, Rg, Cout, act)
        return c, y, (stat[0] if ng == 1 else stat)

    def conv_dgrad_bn(self, dy, W, B, H, Wd, Cin, Cout, k, pre_bn, bn_x, stat, act, train, G, ng=1):
        """dx through conv^T (forward weight W [Cout,Cin,k,k]) and the act(BatchNorm) in front of the convolution's input:
        returns (g = conv^T(dy) * act'(bn(bn_x)) -- for act 0 the plain input gradient --, d bn_x).  ng = 2: one launch per pass
        (the epilogue reads ONE set of batch statistics)."""
        P, R, T = self.P, B * H * Wd, k * k
        g = self.new(R, Cin)
        Rg, Bg = R // ng, B // ng
        nblk = -(-Rg // 128)
        dx = self.new16(R, Cin) if self.store16 else self.new(R, Cin)   # read by the convolution's dgrad / wgrad launches only
        for q in range(ng):
            part = self.new(nblk * 2 * Cin)
            st = stat if ng == 1 else stat[q]
            self.L.conv_fwd_bn(dy[q * Rg:], self._packed(W, Cout, Cin, k)[1], None, g[q * Rg:], Bg, H, Wd, Cout, Cin, k, k, Cout, Cin, T * Cout, 1, 0,
                               Cout, self.pf | self._x16(dy), 2, bn_x[q * Rg:], Cin, st, P[pre_bn + "weight"], P[pre_bn + "bias"], act, part)
            self.bn_bwd_part(pre_bn, bn_x[q * Rg:], g[q * Rg:], st, part, nblk, Rg, Cin, train, G, dx=dx[q * Rg:])
        return g, dx

    def bn_bwd_part(self, pre, x, g, stat, part, nblk, R, C, train, G, dx=None, out16=False):
        if dx is None:
            dx = self.new16(R, C) if (out16 and self.store16) else self.new(R, C)
        if dx.dtype == torch.bfloat16:
            self.L.bn_bwd_from_partial_bf16(x, C, g, C, stat, self.P[pre + "weight"], self.P[pre + "bias"], part, nblk, dx, C, G[pre + "weight"],
                                            G[pre + "bias"], R, C, 1 if train else 0, 1, self.new(2 * C))
            return dx
        self.L.bn_bwd_from_partial(x, C, g, C, stat, self.P[pre + "weight"], self.P[pre + "bias"], part, nblk, dx, C, G[pre + "weight"],
                                   G[pre + "bias"], R, C, 1 if train else 0, 1, 0, self.new(2 * C))
        return dx

    def bn_bwd(self, pre, x, dy, stat, R, C, act, train, G):
        dx = self.new(R, C)
        self.L.bn_bwd(x, C, dy, C, stat, self.P[pre + "weight"], self.P[pre + "bias"], dx, C, G[pre + "weight"],
                      G[pre + "bias"], R, C, act, 1 if train else 0, 1, 0, self.ws)
        return dx

    def dropcfg(self, site, seed, p):
        if p <= 0.0:
            return NODROP
        ss = (seed * 0x85EBCA77 + DROP_SITES[site] * 0xC2B2AE3D) & 0xFFFFFFFF
        return (ss, int(p * (1 << 24)), 1.0 / (1.0 - p), self.seed_dev)

    # ---- GRU block (1x1 conv + BiGRU) ------------------------------------------------------------------------
    def _seqmap(self, B, H, W, vertical):
        # (nseq, T, inner, outer_stride, inner_stride, tstride) in pixels of the NHWC [B,H,W,*] buffer
        if vertical:
            return (B * W, H, W, H * W, 1, W)
        return (B * H, W, 1, W, 0, 1)

    def _gru_fold(self):
        """Per forward call, ONE launch (tatt_gru_fold_w) composes every GruBlock's 1x1 conv with the stacked forward / reverse GRU
        input projection: Wf = W_ih W_c [192][Cin], bf = W_ih b_c + b_ih (model/tsrn.py:1083,:1088 have no non-linearity between
        them).  Returns {block prefix: (Wf, bf)}; the [R,64] conv output and its GEMMs (forward, input gradient, weight gradient)
        never exist."""
        pk = self.wpack.get("gru")
        if pk is None:
            P = self.P
            pres = [f"block{i + 2}.gru{j}." for i in range(self.m.srb_nums) for j in (1, 2)]
            n = len(pres)
            cin = [P[pre + "conv1.weight"].shape[1] for pre in pres]
            buf = self.new(sum(256 * c + 192 for c in cin))      # per block: Wf [192][Cin], WcT [Cin][64], bf [192]
            pk, o = {}, 0
            for pre, c in zip(pres, cin):
                pk[pre] = (buf[o:o + 192 * c].view(192, c), buf[o + 256 * c:o + 256 * c + 192], buf[o + 192 * c:o + 256 * c])
                o += 256 * c + 192
            VP = ctypes.c_void_p
            ptr = lambda ts: [t.data_ptr() for t in ts]
            self.L.gru_fold_w(n, (VP * n)(*ptr([P[pre + "conv1.weight"] for pre in pres])), (VP * n)(*ptr([P[pre + "conv1.bias"] for pre in pres])),
                              (VP * (2 * n))(*ptr([P[pre + "gru.weight_ih_l0" + sfx] for pre in pres for sfx in ("", "_reverse")])),
                              (VP * (2 * n))(*ptr([P[pre + "gru.bias_ih_l0" + sfx] for pre in pres for sfx in ("", "_reverse")])),
                              (VP * n)(*ptr([pk[pre][0] for pre in pres])), (VP * n)(*ptr([pk[pre][1] for pre in pres])),
                              (ctypes.c_int * n)(*cin))
            self.wpack["gru"] = pk
        return pk

    def gru_block_fwd(self, pre, inputs, B, H, W, vertical, save, ctx, key, res=None):
        """inputs: one or two [R,64] tensors whose channel concatenation is the block's input.  Returns out [R,64]; with res
        [R,64] returns (out, res + out), the sum written by the recurrence kernel itself."""
        P, R = self.P, B * H * W
        Wf, bf, _ = self._gru_fold()[pre]
        Cin = Wf.shape[1]
        assert Cin == 64 * len(inputs)
        out = self.new(R, 64)
        sv = self.new(R, 2, 4, 32) if save else None
        gi = self.new(R, 192)
        self.L.gemm_cat(inputs[0], inputs[1] if len(inputs) > 1 else None, 64, Wf, gi, None, 0, bf, R, 192, Cin, 64, Cin, 1, 192, 0,
                        0, 0, self.pl)
        whh = (P[pre + "gru.weight_hh_l0"], P[pre + "gru.weight_hh_l0_reverse"], P[pre + "gru.bias_hh_l0"], P[pre + "gru.bias_hh_l0_reverse"])
        if save:
            ctx[key] = (out, sv)
        if res is None:
            self.L.gru32_fwd(gi, *whh, out, sv, *self._seqmap(B, H, W, vertical))
            return out
        out2 = self.new(R, 64)
        self.L.gru32_fwd_res(gi, *whh, out, sv, res, out2, *self._seqmap(B, H, W, vertical))
        return out, out2

    def gru_block_bwd(self, pre, dout, inputs, B, H, W, vertical, ctx, key, G, bn=None, ng=1):
        """inputs: list of (x [R,64], dx [R,64], acc) per concatenated input.  Writes parameter grads, input grads into dx.
        bn = (pre_bn, bn_x, stat): the first input is the output of that (activation-free) BatchNorm; the GEMM's epilogue then
        also reduces (sum g, sum g xhat) over its rows and the partial sums are returned (part, nblk) for bn_bwd_part()."""
        P, R, L = self.P, B * H * W, self.L
        out, sv = ctx[key]
        dgi, dghn, hprev = self.new(R, 192), self.new(R, 64), self.new(R, 64)
        L.gru32_bwd(dout, out, sv, P[pre + "gru.weight_hh_l0"], P[pre + "gru.weight_hh_l0_reverse"], dgi, dghn, hprev,
                    *self._seqmap(B, H, W, vertical))
        Wf, _, WcT = self._gru_fold()[pre]
        Cin = Wf.shape[1]
        tmp = self._gru_gtmp[pre]                             # (zeroed) gradient of the composed (Wf, bf); unfolded onto W_ih, W_c, biases
        dWf, dbf = tmp[:192 * Cin], tmp[192 * Cin:]           # by ONE grouped launch for all GruBlocks at the end of the SRB stage
        wg = []                                               # all streaming weight gradients of the block: one grouped launch
        for d, sfx in enumerate(["", "_reverse"]):
            gi_d, ghn_d, hp_d = dgi.view(-1)[96 * d:], dghn.view(-1)[32 * d:], hprev.view(-1)[32 * d:]
            # dgh = (dr, dz | dn*r): the r / z columns are dgi's, only the n gate has a block of its own
            wg.append((gi_d, hp_d, G[pre + "gru.weight_hh_l0" + sfx], G[pre + "gru.bias_hh_l0" + sfx], R, 96, 32, 192, 64, None, ghn_d, 64, 64))
            for s, (x, _, _) in enumerate(inputs):
                wg.append((gi_d, x, dWf[96 * d * Cin + 64 * s:], dbf[96 * d:] if s == 0 else None, R, 96, 64, 192, None, Cin))
        self.linear_dwb_group(wg)                             # queued before the input gradients below may update a dx in place
        self._gru_unfold.append((pre, dWf, dbf, WcT, Cin))
        for _, dx, _ in inputs:
            self._guard(dx)
        two = len(inputs) > 1
        if bn is None:
            L.gemm_cat(dgi, None, 0, Wf, inputs[0][1], inputs[1][1] if two else None, 64, None, R, Cin, 192, 192, 1, Cin, 64, 64,
                       inputs[0][2], inputs[1][2] if two else 0, self.pl)
            return None
        pre_bn, bn_x, stat = bn
        assert inputs[0][2] == 0
        Rg = R // ng                                          # ng = 2: one launch per pass (the epilogue reads one set of batch statistics)
        nblk = -(-Rg // 128)
        parts = []
        for q in range(ng):
            part = self.new(nblk * 2 * 64)
            L.gemm_cat_bnbwd(dgi[q * Rg:], None, 0, Wf, inputs[0][1][q * Rg:], inputs[1][1][q * Rg:] if two else None, 64, None, Rg, Cin, 192, 192, 1,
                             Cin, 64, 64, 0, inputs[1][2] if two else 0, self.pl, bn_x[q * Rg:], 64, stat if ng == 1 else stat[q],
                             P[pre_bn + "weight"], P[pre_bn + "bias"], 0, 64, part)
            parts.append(part)
        return (parts[0] if ng == 1 else parts), nblk

    # ---- attention block pieces ------------------------------------------------------------------------------------
    def mha_fwd(self, pre, q_in, k_in, v_in, B, Lq, cfg):
        P = self.P
        W, b = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
        Rq, Rk = B * Lq, B * 26
        Q, K, V = self.new(Rq, 64), self.new(Rk, 64), self.new(Rk, 64)
        self.linear(q_in, W, b, Q, Rq, 64, 64)
        self.linear(k_in, W[64:], b[64:], K, Rk, 64, 64)
        self.linear(v_in, W[128:], b[128:], V, Rk, 64, 64)
        c = self.new(Rq, 64)
        self.L.attn_fwd(Q, K, V, c, None, B, Lq, 26, *cfg)
        a = self.new(Rq, 64)
        self.linear(c, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"], a, Rq, 64, 64)
        return a, (Q, K, V, c)

    def mha_bwd(self, pre, da, saved, q_in, k_in, v_in, B, Lq, cfg, G, dq_in, dk_in, dv_in, acc_q, acc_k, acc_v):
        """da [Rq,64] -> grads of projections; input grads (+)= into dq_in/dk_in/dv_in."""
        P = self.P
        Q, K, V, c = saved
        Rq, Rk = B * Lq, B * 26
        W = P[pre + "in_proj_weight"]
        dW, db = G[pre + "in_proj_weight"], G[pre + "in_proj_bias"]
        self.linear_dwb(da, c, G[pre + "out_proj.weight"], G[pre + "out_proj.bias"], Rq, 64, 64)
        dc = self.new(Rq, 64)
        self.linear_dx(da, P[pre + "out_proj.weight"], dc, Rq, 64, 64)
        dQ, dK, dV = self.new(Rq, 64), self.new(Rk, 64), self.new(Rk, 64)
        self.L.attn_bwd(Q, K, V, dc, dQ, dK, dV, B, Lq, 26, *cfg, 0, self.new(self.L.attn_bwd_workspace(B, Lq)))
        trip = [(dQ, q_in, Rq, dq_in, acc_q), (dK, k_in, Rk, dk_in, acc_k), (dV, v_in, Rk, dv_in, acc_v)]
        self.linear_dwb_group([(d, x, dW[64 * i:], db[64 * i:], R, 64, 64, None, None, None) for i, (d, x, R, _, _) in enumerate(trip)])
        for i, (d, x, R, dx, acc) in enumerate(trip):
            if dx is not None:
                self.linear_dx(d, W[64 * i:], dx, R, 64, 64, acc=acc)

    def ffn_fwd(self, pre, x, R, cfg):
        P = self.P
        h = self.new(R, 64)
        self.linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"], h, R, 64, 64, relu=1)
        hd = h
        if cfg[1]:
            hd = self.new(R, 64)
            self.L.dropout_mul(h, hd, R * 64, *cfg)
        f = self.new(R, 64)
        self.linear(hd, P[pre + "linear2.weight"], P[pre + "linear2.bias"], f, R, 64, 64)
        return f, (h, hd)

    def ffn_bwd(self, pre, df, saved, x, R, cfg, G, dx):
        """dx += d/dx of linear2(drop(relu(linear1 x)))"""
        P = self.P
        h, hd = saved
        self.linear_dwb(df, hd, G[pre + "linear2.weight"], G[pre + "linear2.bias"], R, 64, 64)
        dh = self.new(R, 64)
        self.linear_dx(df, P[pre + "linear2.weight"], dh, R, 64, 64)
        self._guard(dh)
        self.L.relu_drop_bwd(h, dh, dh, R * 64, *cfg)
        self.linear_dwb(dh, x, G[pre + "linear1.weight"], G[pre + "linear1.bias"], R, 64, 64)
        self.linear_dx(dh, P[pre + "linear1.weight"], dx, R, 64, 64, acc=1)

    # ---- fused decoder (tatt_dec_fwd / tatt_dec_bwd): argument marshalling shared by both passes ---------------------------------
    def _dec_params(self, li):
        P, d = self.P, f"infoGen.transformer.decoder.layers.{li}."
        return [P[d + "multihead_attn.in_proj_bias"], P[d + "multihead_attn.out_proj.bias"], P[d + "linear1.bias"], P[d + "linear2.bias"],
                P[d + "norm2.weight"], P[d + "norm2.bias"], P[d + "norm3.weight"], P[d + "norm3.bias"]]

    def _dec_sites(self, li, cfgs):
        """cfgs: one dropout-config function per pass.  -> (site words [pass][4], thresh, scale, seed_dev)"""
        c = [[cfg(f"dec{li}_{k}") for k in ("attn", "drop2", "ffn", "drop3")] for cfg in cfgs]
        return [x[0] for cp in c for x in cp], c[0][0][1], c[0][0][2], c[0][0][3]

    def _dec_call(self, fn, layer, tgt, qp, dec_w, dec_kv, saved, tp_map, B, Lq, cfgs, extra=None):
        """B: batch elements of the launch (all passes); cfgs: dropout-config function per pass."""
        VP, U32 = ctypes.c_void_p, ctypes.c_uint32
        nw, nb = self.P["infoGen.transformer.decoder.norm.weight"], self.P["infoGen.transformer.decoder.norm.bias"]
        ptr = lambda ts: [t.data_ptr() for t in ts]
        npass = len(cfgs)
        if layer is None:                                                  # forward: both layers; sites [pass][layer][4]
            s0, th, sc, sdev = self._dec_sites(0, cfgs)
            s1 = self._dec_sites(1, cfgs)[0]
            sites = [w for ps in range(npass) for sl in (s0, s1) for w in sl[4 * ps:4 * ps + 4]]
            fn(tgt, qp, dec_w, (VP * 2)(*ptr([kv[2] for kv in dec_kv])), (VP * 16)(*ptr(self._dec_params(0) + self._dec_params(1))), nw, nb,
               (VP * 14)(*ptr(saved[0] + saved[1])), tp_map, B, Lq, npass, (U32 * len(sites))(*sites), th, sc, sdev)
            return
        dt3_in, dtp, dtgt, dtgt_acc, dqp, dqp_acc, dY, kvpart, lnpart = extra
        st, th, sc, sdev = self._dec_sites(layer, cfgs)
        fn(layer, dt3_in, dtp, dec_w, dec_kv[layer][2], (VP * 8)(*ptr(self._dec_params(layer))), nw, (VP * 7)(*ptr(saved[layer])), dtgt, dtgt_acc,
           dqp, dqp_acc, (VP * 4)(*ptr(dY)), kvpart, lnpart, B, Lq, npass, (U32 * len(st))(*st), th, sc, sdev)

    def _dec_bwd_fused(self, dec, dtp, db1, dqp, dkins, dmems, kins, mems, B, Lq, cfgs, G, after_dqp=None):
        """Backward of both decoder layers: two tatt_dec_bwd launches (layer 1, then layer 0 -- which adds its input gradient onto
        db1 and its query_pos gradient onto layer 1's), the reductions of their partials, and ONE grouped launch for the eight
        [64,64] weight gradients.  B: batch elements of all passes; kins / mems / dkins / dmems: per pass (keys = memory + pos)."""
        L, P, tr = self.L, self.P, "infoGen.transformer."
        _, dec_w, dec_kv, saved = dec
        ng = len(cfgs)
        R0, Rt, Rg = B * Lq, B * 26, (B // ng) * 26
        VP = ctypes.c_void_p
        dtgt1 = self.new(R0, 64)
        wg, outs = [], {}
        for li in (1, 0):                                                  # the two launches back to back: layer 0 needs layer 1's dtgt only
            dY = [self.new(R0, 64) for _ in range(4)]
            kvpart, lnpart = self.new(L.dec_bwd_kvpart_floats(B, Lq)), self.new(L.dec_bwd_lnpart_floats(B, Lq))
            if li == 0:
                self._guard(db1)
            self._dec_call(L.dec_bwd, li, None, None, dec_w, dec_kv, saved, None, B, Lq, cfgs,
                           extra=(None if li == 1 else dtgt1, dtp, dtgt1 if li == 1 else db1, 0 if li == 1 else 1, dqp, 0 if li == 1 else 1,
                                  dY, kvpart, lnpart))
            outs[li] = (dY, kvpart, lnpart)
        if after_dqp is not None:                                          # dqp is complete: the RPE's BPTT chain -- the longest strand of
            after_dqp()                                                    # what is left of the pass -- forks here, ahead of the reductions
        for li in (1, 0):
            d = f"{tr}decoder.layers.{li}."
            dY, kvpart, lnpart = outs[li]
            L.dec_ln_final(lnpart, B, Lq, (VP * 6)(*[G[n].data_ptr() for n in (d + "norm2.weight", d + "norm2.bias", d + "norm3.weight", d + "norm3.bias",
                                                                            tr + "decoder.norm.weight", tr + "decoder.norm.bias")]))
            dKp, dVp = self.new(Rt, 64), self.new(Rt, 64)
            L.attn_kv_final(kvpart, Lq // 128, B, dKp, dVp, 0)
            W = P[d + "multihead_attn.in_proj_weight"]
            dW, db = G[d + "multihead_attn.in_proj_weight"], G[d + "multihead_attn.in_proj_bias"]
            qin, _, c, _, t2, hd, _ = saved[li]
            wg += [(dY[0], qin, dW, db, R0, 64, 64, None, None, None),
                   (dY[1], c, G[d + "multihead_attn.out_proj.weight"], G[d + "multihead_attn.out_proj.bias"], R0, 64, 64, None, None, None),
                   (dY[2], t2, G[d + "linear1.weight"], G[d + "linear1.bias"], R0, 64, 64, None, None, None),
                   (dY[3], hd, G[d + "linear2.weight"], G[d + "linear2.bias"], R0, 64, 64, None, None, None)]
            for q in range(ng):                                            # key / value projections: per pass (their inputs differ by pass)
                dK, dV = dKp[q * Rg:], dVp[q * Rg:]
                self.linear_dwb_group([(dK, kins[q], dW[64:], db[64:], Rg, 64, 64, None, None, None),
                                       (dV, mems[q], dW[128:], db[128:], Rg, 64, 64, None, None, None)])
                self.linear_dx(dK, W[64:], dkins[q], Rg, 64, 64, acc=0 if li == 1 else 1)
                self.linear_dx(dV, W[128:], dmems[q], Rg, 64, 64, acc=0 if li == 1 else 1)
        self.linear_dwb_group(wg)

    def ln_fwd(self, pre, x, res, cfg, R, save_v=True):
        v = self.new(R, 64) if (res is not None and save_v) else None
        y = self.new(R, 64)
        self.L.ln_fwd(x, res, *cfg, self.P[pre + "weight"], self.P[pre + "bias"], v, y, R, 1e-5)
        return y, (v if v is not None else x)

    def ln_bwd(self, pre, v, dy, dyscale, dv, dv_acc, cfg, R, G, want_dres):
        """dv (+)= grad wrt the LN input v;  returns dres (grad wrt the dropped residual branch)."""
        dres = self.new(R, 64) if (want_dres and cfg[1]) else None
        self._guard(dv)
        self.L.ln_bwd(v, dy, dyscale, self.P[pre + "weight"], dv, dv_acc, dres, *cfg, G[pre + "weight"], G[pre + "bias"], 1, R,
                      1e-5, self.ws)
        return dres

    # ================================================================================================================
    # ---- mask=False (three image channels, reference --mask off: model/tsrn.py:586, interfaces/base.py:179-181) ------------------
    # The kernels are built for four channels.  A three-channel network is the four-channel one whose extra input channel is zero
    # and whose weights touching that channel are zero: block1 and the STN head's first convolution get a zero input-channel slice,
    # the output convolution a zero fourth filter + bias (its output, sigmoid(0), is dropped).  The padded tensors are made per call
    # from the module's parameters; their gradients are computed into padded temporaries and the live slices added to G.
    def _pad3_names(self):
        k8 = f"block{self.m.srb_nums + 3}.1."
        names = {"block1.0.weight": 1, k8 + "weight": 0, k8 + "bias": 0}
        if self.m.stn:
            names["stn_head.stn_convnet.0.0.weight"] = 1
        return names

    def forward(self, x, text_emb, training, seed, p_drop=None, want_ctx=True, wpack=None, tp_ready=None, rpe_from=None, x2=None, seed2=1):
        if getattr(self.m, "in_planes", 4) == 4:
            return self._forward(x, text_emb, training, seed, p_drop, want_ctx, wpack, tp_ready, rpe_from, x2, seed2)
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 16, 64):
            raise RuntimeError(f"expected LR input [B,3,16,64] (mask=False), got {tuple(x.shape)}")
        if wpack is not None or rpe_from is not None or x2 is not None:
            raise NotImplementedError("mask=False: the two-pass (TSC) step is not wired up")
        self.refresh()
        orig = self.P
        padded = dict(orig)
        for n, dim in self._pad3_names().items():
            t = orig[n]
            shape = list(t.shape)
            shape[dim] = 1
            padded[n] = torch.cat([t, t.new_zeros(shape)], dim).contiguous()
        x4 = torch.cat([x.detach().float(), x.new_zeros(x.shape[0], 1, 16, 64, dtype=torch.float32)], 1).contiguous()
        self.P = padded
        try:
            sr4, ctx = self._forward(x4, text_emb, training, seed, p_drop, want_ctx, None, tp_ready, None)
        finally:
            self.P = orig
        ctx["pad3"] = padded
        return sr4[:, :3].contiguous(), ctx

    def backward(self, ctx, dsr, G, on_stage=None):
        padded = ctx.get("pad3")
        if padded is None:
            return self._backward(ctx, dsr, G, on_stage)
        if on_stage is not None:
            raise NotImplementedError("mask=False: bucketed data-parallel training is not wired up")
        orig = self.P
        G4 = dict(G)
        names = self._pad3_names()
        for n in names:
            G4[n] = torch.zeros_like(padded[n])
        dsr4 = torch.cat([dsr.float(), dsr.new_zeros(dsr.shape[0], 1, *dsr.shape[2:], dtype=torch.float32)], 1).contiguous()
        self.P = padded
        try:
            self._backward(ctx, dsr4, G4, None)
        finally:
            self.P = orig
        for n, dim in names.items():
            G[n] += G4[n].narrow(dim, 0, 3)

    def _forward(self, x, text_emb, training, seed, p_drop=None, want_ctx=True, wpack=None, tp_ready=None, rpe_from=None, x2=None, seed2=1):
        """tp_ready: optional torch.cuda.Event after which `text_emb` may be read (the prior generator may still be running on
        another stream while the STN head, block1 and the RPE chain -- which do not need the prior -- are under way).
        wpack: the packed conv weights of an earlier forward call on the SAME weights (the TSC step runs the network
        twice per optimiser step) — skips the repack.
        x2 (with seed2): a SECOND input batch pushed through the network in the same launches -- the two SR passes of the TSC step
        (interfaces/super_resolution_eng.py:233 and :259: same weights, same prior, pass 2 on the rotated LR image) as one batch of 2B
        rows.  Per pass stay: BatchNorm batch statistics and running-statistic updates (pass 1 first, num_batches_tracked += 2), dropout
        masks (seed / seed2, indices count within a pass), the STN head (BatchNorm over tiny maps: per-pass launches) and the text
        branch (26 tokens per image: per-pass launches); the recurrent positional encoding is a function of the weights alone and is
        computed once.  Returns sr [2B,4,32,128] (pass 1 first)."""
        L, m = self.L, self.m
        self.refresh()
        P = self.P
        self._nbt = []
        B, C, H, W = x.shape
        if (C, H, W) != (4, 16, 64):
            raise RuntimeError(f"expected LR input [B,4,16,64], got {tuple(x.shape)}")
        K = m.text_embedding
        if tuple(text_emb.shape) != (B, K, 1, 26):
            raise RuntimeError(f"expected text prior [B,{K},1,26], got {tuple(text_emb.shape)}")
        x = x.detach().contiguous().float()
        text_emb = text_emb.detach().float()
        ng = 1
        xin = [x]
        if x2 is not None:
            if tuple(x2.shape) != tuple(x.shape) or not training or rpe_from is not None:
                raise RuntimeError("x2: a second training-mode pass of the same shape (and no rpe_from)")
            if not self.dec_fused or (B * H * W) % 128:
                raise RuntimeError("two-pass batches need the fused decoder (TATT_DEC_FUSED) and B*H*W % 128 == 0")
            ng, xin = 2, [x, x2.detach().contiguous().float()]
        p = (m.dropout_p if p_drop is None else p_drop) if training else 0.0
        seeds = [seed, seed2][:ng]
        cfgs = [lambda site, s_=s_: self.dropcfg(site, s_, p) for s_ in seeds]
        ctx = Ctx()
        prec = getattr(m, "matmul_precision", "fp32")
        if prec not in ("fp32", "bf16", "f32x3"):
            raise RuntimeError(f"matmul_precision must be 'fp32', 'bf16' or 'f32x3', got {prec!r}")
        self.pf = ctx["pf"] = {"fp32": 0, "bf16": 2, "f32x3": 4}[prec]
        # Linear / GRU-projection GEMMs and the sites bf16 mode pins to fp32: split products (fp32 accuracy) in both non-default modes
        # (bf16 mode: 6.77 -> 6.62 ms against fp32 MFMA on those launches, same error as before: 6.1e-3 max-abs at B=4)
        self.pl = self.px = ctx["pl"] = 0 if prec == "fp32" else 4
        self.store16 = ctx["s16"] = bool(prec == "bf16" and training and BF16_STORE and getattr(m, "bf16_storage", True))
        ctx["meta"] = (B, training, seed, p)
        ctx["ng"], ctx["seeds"] = ng, seeds
        self.wpack = ctx["wpack"] = {} if wpack is None else wpack   # weights change every step: repacked per forward call
        Bt = ng * B                                                  # images in the batched launches
        R0, Ra = B * H * W, Bt * H * W                               # rows of one pass / of the batch
        use_stn = bool(m.stn and training)
        ctx["x"] = xin

        # ---- recurrent positional encoding (model/transformer_v2.py:215-221): GRU over the batch axis, identical input at
        # every step.  It depends on weights only and is a chain of B dependent, latency-bound launches, so it is forked
        # onto a side stream here and joined where the decoder first needs query_pos; the STN head, block1 and the encoder
        # run beside it (inside a captured hipGraph this becomes a parallel branch).
        ig, tr = "infoGen.", "infoGen.transformer."
        g = tr + "gru_encoding."
        main, side = torch.cuda.current_stream(), self.side_stream()
        if rpe_from is not None:
            # A second forward pass of the same optimisation step (the TSC term's SR pass over the rotated input): the encoding is a
            # function of the weights alone, so the first pass's chain is reused; backward() of this pass then hands its dL/dquery_pos
            # to that pass (ctx["rpe_owner"]) instead of running a BPTT chain of its own -- the chain is linear in that gradient.
            if rpe_from["meta"][0] != B or rpe_from.get("rpe_owner") is not None:
                raise RuntimeError("rpe_from must be the context of a forward pass of this step with the same batch size that ran the RPE itself")
            xg, gi, hstep, gh_all, qp = rpe_from["rpe"]
            ctx["rpe_owner"] = rpe_from
            rpe_from["rpe_lent"] = True
        else:
            # The whole strand -- input gather, the two time-invariant input projections (split-K into a workspace of the strand's
            # own), h_0 = 0 and the chain -- is forked before anything else of the pass is issued: it reads weights only, and with
            # its 100 us of small launches on the main stream the STN head / block1 started that much later.
            xg = self.new(64, 1024)
            gi = self.new(2, 64, 1536)
            gh_all = self.new(2, B, 64, 1536)
            qp = self.new(R0, 64)
            whh_f, whh_b = P[g + "weight_hh_l0"], P[g + "weight_hh_l0_reverse"]
            dsh, dsg = (B + 1) * 64 * 512, B * 64 * 1536
            if self.ws_side is None or self.ws_side.device != self.dev:
                self.ws_side = torch.empty(8 * 64 * 1536, device=self.dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                L.rpe_gather_x(P[ig + "init_factor.weight"], xg)
                self.linear(xg, P[g + "weight_ih_l0"], P[g + "bias_ih_l0"], gi[0], 64, 1536, 1024, splitk=8, fp32=True, ws=self.ws_side)
                self.linear(xg, P[g + "weight_ih_l0_reverse"], P[g + "bias_ih_l0_reverse"], gi[1], 64, 1536, 1024, splitk=8, fp32=True,
                            ws=self.ws_side)
                hstep = self.zeros(2, B + 1, 64, 512)
                if self.rpe_chain():   # the whole chain as one launch of persistent workgroups (tatt_rpe_chain_fwd)
                    if self._rpe_sync is None or self._rpe_sync.device != self.dev:   # once per engine: word 1 is a sticky error flag
                        self._rpe_sync = torch.zeros(L.rpe_chain_sync_words(B), device=self.dev, dtype=torch.int32)
                    L.rpe_chain_fwd(gi, whh_f, whh_b, P[g + "bias_hh_l0"], P[g + "bias_hh_l0_reverse"], hstep, gh_all, qp, B, dsg, dsh,
                                    self._rpe_sync)
                else:
                    for s in range(B):  # one fused launch per step: h_s W_hh^T, gates, h_{s+1}, query_pos (tatt_rpe_step_fwd)
                        L.rpe_step_fwd(gi, whh_f, whh_b, P[g + "bias_hh_l0"], P[g + "bias_hh_l0_reverse"], hstep[0, s], hstep[0, s + 1],
                                       gh_all[0, s], qp, s, B, dsg, dsh)
        ctx["rpe"] = (xg, gi, hstep, gh_all, qp)

        # ---- text branch of the TP interpreter (fc_in, PReLU, positional encoding, encoder layer): it reads the prior and weights
        # only, so it runs on a stream of its own beside the STN head / block1 (and the RPE chain) until the decoder needs it
        enc = self.enc_stream() if tp_ready is None else main    # (with a prior still in flight there are enough strands already:
        enc.wait_stream(main)                                    #  the whole-iteration step got slower with one more)
        fused = self.dec_fused and (H * W) % 128 == 0
        with torch.cuda.stream(enc):
            Rt = B * 26
            tpr = self.new(Rt, K)
            if tp_ready is not None:
                enc.wait_event(tp_ready)
            L.tp_rows(text_emb, text_emb.stride(0), text_emb.stride(1), text_emb.stride(3), tpr, B, K, 26)
            t0 = self.new(Rt, 64)
            self.linear(tpr, P[ig + "fc_in.weight"], P[ig + "fc_in.bias"], t0, Rt, 64, K)
            t = self.new(Rt, 64)
            L.prelu_fwd(t0, P[ig + "activation.weight"], t, Rt * 64)
            src2 = self.new(Rt, 64)
            L.axpby(t, 2.0, None, 0.0, src2, Rt * 64, 0)             # encoder input doubled (model/transformer_v2.py:273)
            e = tr + "encoder.layers.0."
            text = []                                                # per pass: the dropout masks differ
            for cfg in cfgs:
                pos = self.new(Rt, 64)
                L.pe_dropout(P[ig + "pe.pe"], pos, B, 26, *cfg("pe"))
                qk = self.new(Rt, 64)
                L.axpby(src2, 1.0, pos, 1.0, qk, Rt * 64, 0)
                a_e, att_e = self.mha_fwd(e + "self_attn.", qk, qk, src2, B, 26, cfg("enc_attn"))
                s1, v1e = self.ln_fwd(e + "norm1.", src2, a_e, cfg("enc_drop1"), Rt)
                f_e, ffn_e = self.ffn_fwd(e, s1, Rt, cfg("enc_ffn"))
                mem, v2e = self.ln_fwd(e + "norm2.", s1, f_e, cfg("enc_drop2"), Rt)
                kin = self.new(Rt, 64)
                L.axpby(mem, 1.0, pos, 1.0, kin, Rt * 64, 0)
                text.append((pos, qk, att_e, v1e, s1, ffn_e, v2e, mem, kin))
            if fused:
                # fused decoder (tatt_dec_fwd): the projected keys / values of both layers and the MFMA-order operand packs belong to
                # the text branch -- they need the encoder memory and weights only
                dec_w = self.new(L.dec_packed_w_floats())
                VP = ctypes.c_void_p
                dl = [f"{tr}decoder.layers.{li}." for li in range(2)]
                L.dec_pack_w((VP * 8)(*[t_.data_ptr() for d_ in dl for t_ in (P[d_ + "multihead_attn.in_proj_weight"], P[d_ + "multihead_attn.out_proj.weight"],
                                                                             P[d_ + "linear1.weight"], P[d_ + "linear2.weight"])]), dec_w)
                dec_kv = []
                for d_ in dl:
                    Wi, bi = P[d_ + "multihead_attn.in_proj_weight"], P[d_ + "multihead_attn.in_proj_bias"]
                    Kp, Vp = self.new(ng * Rt, 64), self.new(ng * Rt, 64)
                    for q_, tx in enumerate(text):
                        self.linear(tx[8], Wi[64:], bi[64:], Kp[q_ * Rt:], Rt, 64, 64)
                        self.linear(tx[7], Wi[128:], bi[128:], Vp[q_ * Rt:], Rt, 64, 64)
                    tab = self.new(L.dec_kv_table_floats(Bt))
                    L.dec_pack_kv(Kp, Vp, tab, Bt)
                    dec_kv.append((Kp, Vp, tab))

        # per-step weight packs of the main strand (conv weights in MFMA order, 1x1 conv folded into the GRU projections): issued
        # after the two forks, which need neither
        self._prepack(use_stn, training)
        self._gru_fold()

        # ---- STN head + TPS rectification (model/tsrn.py:655-657), per pass --------------------------------
        xs = self.new(Ra, 4)
        stn_ctx = []
        for q_, xq in enumerate(xin):
            xs_q = xs[q_ * R0:]
            if not use_stn:
                L.nchw_to_nhwc(xq, xs_q, B, 4, H * W)
                continue
            xn = self.new(R0, 4)
            L.nchw_to_nhwc(xq, xn, B, 4, H * W)
            h, hh, ww = xn, H, W
            stn_saved = []
            for (idx, cin, cout, pool) in STN_STAGES:
                pre = f"stn_head.stn_convnet.{idx}."
                Rr = B * hh * ww
                c = self.new(Rr, cout)
                self.conv(h, P[pre + "0.weight"], P[pre + "0.bias"], c, B, hh, ww, cin, cout, 3, fp32=True)
                kh, kw = pool or (1, 1)
                if STN_SMALL and L.bn_pool_small_supported(B, hh, ww, cout, kh, kw):
                    # the small maps: batch statistics, apply + ReLU and the max-pool as one launch (tatt_bn_pool_small_fwd)
                    stat, a = self.new(2, cout), self.new(Rr, cout)
                    hp = self.new(B * (hh // kh) * (ww // kw), cout) if pool is not None else None
                    L.bn_pool_small_fwd(c, stat, P[pre + "1.weight"], P[pre + "1.bias"], P[pre + "1.running_mean"], P[pre + "1.running_var"],
                                        a, hp, B, hh, ww, cout, kh, kw, 1e-5, 0.1)
                    self._nbt.append(P[pre + "1.num_batches_tracked"])
                    rec = [h, c, stat, a, hh, ww]
                    h = a if pool is None else hp
                    hh, ww = hh // kh, ww // kw
                    stn_saved.append(rec)
                    continue
                a, stat = self.bn(pre + "1.", c, Rr, cout, True, 1)
                rec = [h, c, stat, a, hh, ww]
                if pool is not None:
                    hp = self.new(B * (hh // pool[0]) * (ww // pool[1]), cout)
                    L.maxpool_fwd(a, hp, B, hh, ww, cout, pool[0], pool[1])
                    h, hh, ww = hp, hh // pool[0], ww // pool[1]
                else:
                    h = a
                stn_saved.append(rec)
            f = self.new(B, 512)
            L.nhwc_to_nchw(h, f, B, 256, 2)                                   # flatten order of x.view(B,-1) on NCHW
            l1 = self.new(B, 512)
            # few rows, K = 512: split-K (one workgroup per column tile walked sixteen k tiles on its own, 20 us)
            self.linear(f, P["stn_head.stn_fc1.0.weight"], P["stn_head.stn_fc1.0.bias"], l1, B, 512, 512, fp32=True, splitk=STN_FC_SPLITK)
            if STN_SMALL and L.bn_pool_small_supported(B, 1, 1, 512, 1, 1):
                stat1, feat = self.new(2, 512), self.new(B, 512)
                L.bn_pool_small_fwd(l1, stat1, P["stn_head.stn_fc1.1.weight"], P["stn_head.stn_fc1.1.bias"], P["stn_head.stn_fc1.1.running_mean"],
                                    P["stn_head.stn_fc1.1.running_var"], feat, None, B, 1, 1, 512, 1, 1, 1e-5, 0.1)
                self._nbt.append(P["stn_head.stn_fc1.1.num_batches_tracked"])
            else:
                feat, stat1 = self.bn("stn_head.stn_fc1.1.", l1, B, 512, True, 1)
            feat01 = self.new(B, 512)
            L.axpby(feat, 0.1, None, 0.0, feat01, B * 512, 0)
            ctrl = self.new(B, 40)
            self.linear(feat01, P["stn_head.stn_fc2.weight"], P["stn_head.stn_fc2.bias"], ctrl, B, 40, 512, fp32=True, splitk=2 * STN_FC_SPLITK)
            Tm = self._tps_matrix(H, W)
            L.tps_sample_fwd(xq, ctrl, Tm, xs_q, None, B, 4, H, W)
            stn_ctx.append((stn_saved, f, l1, stat1, feat01, ctrl, Tm))
        if use_stn:
            ctx["stn_all"] = stn_ctx
            ctx["stn"] = stn_ctx[0]
        ctx["xs"] = xs

        # ---- block1: conv9x9 + PReLU -----------------------------------------------------------------------------
        c1 = self.new(Ra, 64)
        self.conv(xs, P["block1.0.weight"], P["block1.0.bias"], c1, Bt, H, W, 4, 64, 9)
        b1 = self.new(Ra, 64)
        L.prelu_fwd(c1, P["block1.1.weight"], b1, Ra * 64)
        ctx["c1"], ctx["b1"] = c1, b1

        # The first SRB's convolutions (conv-BN-mish-conv-BN, model/tsrn.py:333-338) need block1's output only: they go here,
        # ahead of the decoder, where the main chain would otherwise sit waiting for the RPE chain's query_pos.
        def srb_convs(pre, blk):
            ca, ma, st1 = self.conv_bn(pre + "bn1.", blk, P[pre + "conv1.weight"], P[pre + "conv1.bias"], Bt, H, W, 64, 64, 3, training, 2, ng=ng, out16=True)
            cb, r2, st2 = self.conv_bn(pre + "bn2.", ma, P[pre + "conv2.weight"], P[pre + "conv2.bias"], Bt, H, W, 64, 64, 3, training, 0, ng=ng)
            return ca, ma, st1, cb, r2, st2
        first_convs = srb_convs("block2.", b1) if m.srb_nums > 0 else None

        # ---- TP interpreter (model/tsrn.py:194-226, model/transformer_v2.py:198-244) -----------------------------------
        # decoder layers (cross attention only, model/transformer_v2.py:805-832)
        main.wait_stream(enc)                                                 # join the text branch: memory / keys
        main.wait_stream(side)                                                # join the RPE chain: query_pos is needed now
        tgt = b1
        dec = []
        tp_map = self.new(Ra, 64)
        if fused:
            if want_ctx:
                saved = [[self.new(Ra, 64) for _ in range(7)] for _ in range(2)]
            else:                                                         # nobody reads them: one scratch tensor takes all the stores
                saved = [[self.new(Ra, 64)] * 7] * 2
            self._dec_call(L.dec_fwd, None, b1, qp, dec_w, dec_kv, saved, tp_map, Bt, H * W, cfgs)
            dec = ("fused", dec_w, dec_kv, saved)
        pos, qk, att_e, v1e, s1, ffn_e, v2e, mem, kin = text[0]
        cfg = cfgs[0]
        for li in range(0 if fused else 2):
            d = f"{tr}decoder.layers.{li}."
            qin = self.new(R0, 64)
            L.axpby(tgt, 1.0, qp, 1.0, qin, R0 * 64, 0)
            a, att = self.mha_fwd(d + "multihead_attn.", qin, kin, mem, B, H * W, cfg(f"dec{li}_attn"))
            t2, v2 = self.ln_fwd(d + "norm2.", tgt, a, cfg(f"dec{li}_drop2"), R0)
            f_, ffn = self.ffn_fwd(d, t2, R0, cfg(f"dec{li}_ffn"))
            t3, v3 = self.ln_fwd(d + "norm3.", t2, f_, cfg(f"dec{li}_drop3"), R0)
            inter, _ = self.ln_fwd(tr + "decoder.norm.", t3, None, NODROP, R0)
            L.axpby(inter, 0.5, None, 0.0, tp_map, R0 * 64, int(li > 0))
            dec.append((tgt, qin, att, v2, t2, ffn, v3, t3))
            tgt = t3
        ctx["tp"] = (tpr, t0, t, pos, xg, gi, hstep, gh_all, qp, src2, qk, att_e, v1e, s1, ffn_e, v2e, mem, kin, dec)
        ctx["text"] = text
        ctx["tp_map"] = tp_map

        # ---- SRBs (model/tsrn.py:883-901) ----------------------------------------------------------------------------
        blk = b1
        srbs = []
        for i in range(m.srb_nums):
            pre = f"block{i + 2}."
            ca, ma, st1, cb, r2, st2 = first_convs if i == 0 else srb_convs(pre, blk)
            if GRU_RES:
                r, s = self.gru_block_fwd(pre + "gru1.", [r2, tp_map], Bt, H, W, True, want_ctx, ctx, pre + "gru1", res=blk)   # s = blk + r
            else:
                r = self.gru_block_fwd(pre + "gru1.", [r2, tp_map], Bt, H, W, True, want_ctx, ctx, pre + "gru1")
                s = self.new(Ra, 64)
                L.axpby(blk, 1.0, r, 1.0, s, Ra * 64, 0)
            out = self.gru_block_fwd(pre + "gru2.", [s], Bt, H, W, False, want_ctx, ctx, pre + "gru2")
            srbs.append((blk, ca, st1, ma, cb, st2, r2, s))
            blk = out
        ctx["srbs"] = srbs

        # ---- block7 (conv+BN) + global residual, block8 (upsample + 9x9) + sigmoid ---------------------------------------
        k7 = f"block{m.srb_nums + 2}."
        c7, u_in, st7 = self.conv_bn(k7 + "1.", blk, P[k7 + "0.weight"], P[k7 + "0.bias"], Bt, H, W, 64, 64, 3, training, 0, res=b1, ng=ng, out16=True)
        k8 = f"block{m.srb_nums + 3}."
        cu = self.new(Ra, 256)
        self.conv(u_in, P[k8 + "0.conv.weight"], P[k8 + "0.conv.bias"], cu, Bt, H, W, 64, 256, 3)
        up = self.new(4 * Ra, 64)
        L.pixshuf_mish_fwd(cu, up, Bt, H, W, 64)
        pre_s = self.new(4 * Ra, 4)
        # the output convolution feeds the sigmoid directly (no normalisation behind it to absorb operand rounding): fp32 in both modes
        self.conv(up, P[k8 + "1.weight"], P[k8 + "1.bias"], pre_s, Bt, 2 * H, 2 * W, 64, 4, 9, fp32=True)
        sr = self.new(Bt, 4, 2 * H, 2 * W)
        L.sigmoid_to_nchw(pre_s, sr, Bt, 4, 4 * H * W)
        ctx["tail"] = (blk, c7, st7, u_in, cu, up, pre_s)
        ctx["sr"] = sr
        ctx["use_stn"] = use_stn
        if self._nbt:                                         # one fused launch; a two-pass batch counts every BatchNorm twice
            uniq = {}
            for t_ in self._nbt:
                uniq.setdefault(t_.data_ptr(), [t_, 0])[1] += 1
            for n_ in set(c_ for _, c_ in uniq.values()):
                torch._foreach_add_([t_ for t_, c_ in uniq.values() if c_ == n_], n_)
            self._nbt = []
        return sr, ctx

    # ================================================================================================================
    def _backward(self, ctx, dsr, G, on_stage=None):
        """Accumulates parameter gradients of one forward call into G (dict name -> fp32 tensor).
        on_stage(name) is called when a gradient bucket is complete ("tail", "srbs", then "head" and "infogen" together at the end):
        the data-parallel trainer starts all-reduces there, overlapping the rest of the pass.  A callback with a `stages` attribute is
        called for those names only (the weight-gradient streams are joined only where the caller acts on the announcement)."""
        user_stage = on_stage
        self._w_keep, self._w_pending, self._w_defer = [], set(), []   # nothing may survive from a pass that raised half way

        def finish_unfold():
            """The gradients of the composed GRU input projections (Wf, bf) of ALL GruBlocks -> W_ih, W_c, biases, one launch
            (tatt_gru_unfold_grad_group) on the main stream, after a join: it reads what the grouped weight-gradient kernels of
            both weight-gradient streams accumulated."""
            un, n_ = self._gru_unfold, len(self._gru_unfold)
            if not n_:
                return
            VP, sfxs = ctypes.c_void_p, ("", "_reverse")
            arr = lambda ts: (VP * len(ts))(*[t.data_ptr() for t in ts])
            L.gru_unfold_grad_group(n_, arr([u[1] for u in un]), arr([u[2] for u in un]), arr([u[3] for u in un]), arr([P[u[0] + "conv1.bias"] for u in un]),
                                    arr([P[u[0] + "gru.weight_ih_l0" + x] for u in un for x in sfxs]),
                                    arr([G[u[0] + "gru.weight_ih_l0" + x] for u in un for x in sfxs]),
                                    arr([G[u[0] + "gru.bias_ih_l0" + x] for u in un for x in sfxs]), arr([G[u[0] + "conv1.weight"] for u in un]),
                                    arr([G[u[0] + "conv1.bias"] for u in un]), (ctypes.c_int * n_)(*[u[4] for u in un]))
            self._gru_unfold = []

        wanted = getattr(user_stage, "stages", ("tail", "srbs", "head", "infogen"))   # announcements the caller acts on

        def on_stage(name):                                   # a bucket is complete only once its weight gradients have landed
            if user_stage is not None and name in wanted:
                self._w_join()
                if name == "srbs":
                    finish_unfold()
                user_stage(name)
            elif name == "infogen":                           # end of the pass: the optimiser reads every gradient
                self._w_join()
                finish_unfold()
            else:
                self._w_flush()

        def axpby(x, a, y, b, out, n, accumulate):            # every in-place update of a gradient tensor goes through the guard
            self._guard(out)
            L.axpby(x, a, y, b, out, n, accumulate)
        L, m, P = self.L, self.m, self.P
        B, training, seed, p = ctx["meta"]
        ng, seeds = ctx.get("ng", 1), ctx.get("seeds", [seed])    # ng = 2: a two-pass batch (forward(x2=...)); B is the size of ONE pass
        self.pf, self.pl = ctx["pf"], ctx.get("pl", 0)
        self.px = self.pl
        self.store16 = ctx.get("s16", False)
        self.wpack = ctx["wpack"]
        cfgs = [lambda site, s_=s_: self.dropcfg(site, s_, p) for s_ in seeds]
        cfg = cfgs[0]
        H, W = 16, 64
        Bt = ng * B
        R0, Ra = B * H * W, Bt * H * W
        dsr = dsr.contiguous().float()
        if dsr.shape[0] != Bt:
            raise RuntimeError(f"backward: gradient of {dsr.shape[0]} images for a forward pass of {Bt}")
        blk_last, c7, st7, u_in, cu, up, pre_s = ctx["tail"]
        b1, c1, tp_map = ctx["b1"], ctx["c1"], ctx["tp_map"]
        k7, k8 = f"block{m.srb_nums + 2}.", f"block{m.srb_nums + 3}."

        # ---- tail ------------------------------------------------------------------------------------------------
        dpre = self.new(4 * Ra, 4)
        L.sigmoid_bwd_to_nhwc(ctx["sr"], dsr, dpre, Bt, 4, 4 * H * W)
        self.conv_wgrad(up, dpre, G[k8 + "1.weight"], G[k8 + "1.bias"], Bt, 2 * H, 2 * W, 64, 4, 9, fp32=True)
        dup = self.new(4 * Ra, 64)
        self.conv_dgrad(dpre, P[k8 + "1.weight"], dup, Bt, 2 * H, 2 * W, 64, 4, 9, fp32=True)
        if self.store16:
            # read by the up-sampling convolution's dgrad / wgrad launches only: stored as bf16; the convolution's bias gradient (the
            # column sum of the unrounded values) comes from per-block fp32 sums the same launch writes on the side
            dcu, colpart = self.new16(Ra, 256), self.new(L.pixshuf_colpart_rows(Bt, H, W), 256)
            L.pixshuf_mish_bwd_bf16(cu, dup, dcu, colpart, Bt, H, W, 64)
            self.conv_wgrad(u_in, dcu, G[k8 + "0.conv.weight"], None, Bt, H, W, 64, 256, 3, bias_grad=False)
            self.colsum(colpart, colpart.shape[0], 256, G[k8 + "0.conv.bias"])
        else:
            dcu = self.new(Ra, 256)
            L.pixshuf_mish_bwd(cu, dup, dcu, Bt, H, W, 64)
            self.conv_wgrad(u_in, dcu, G[k8 + "0.conv.weight"], G[k8 + "0.conv.bias"], Bt, H, W, 64, 256, 3)
        # grad of block1's output (accumulated below) = grad of u_in = bn(c7) + b1: the BatchNorm's reduction runs in the dgrad epilogue
        db1, dc7 = self.conv_dgrad_bn(dcu, P[k8 + "0.conv.weight"], Bt, H, W, 64, 256, 3, k7 + "1.", c7, st7, 0, training, G, ng=ng)
        self.conv_wgrad(blk_last, dc7, G[k7 + "0.weight"], G[k7 + "0.bias"], Bt, H, W, 64, 64, 3, bias_grad=not training)
        dout = self.new(Ra, 64)
        self.conv_dgrad(dc7, P[k7 + "0.weight"], dout, Bt, H, W, 64, 64, 3)
        on_stage("tail")

        # ---- SRBs, last to first ----------------------------------------------------------------------------------
        fold = self._gru_fold()
        sizes = {pre: 192 * fold[pre][0].shape[1] + 192 for pre in fold}
        gbuf, o = self.zeros(sum(sizes.values())), 0          # one memset for the (Wf, bf) gradients of all GruBlocks
        self._gru_gtmp, self._gru_unfold = {}, []
        for pre, n_ in sizes.items():
            self._gru_gtmp[pre] = gbuf[o:o + n_]
            o += n_
        dtp = self.new(Ra, 64)
        first_tp = True
        for i in reversed(range(m.srb_nums)):
            pre = f"block{i + 2}."
            blk, ca, st1, ma, cb, st2, r2, s = ctx["srbs"][i]
            ds = self.new(Ra, 64)
            self.gru_block_bwd(pre + "gru2.", dout, [(s, ds, 0)], Bt, H, W, False, ctx, pre + "gru2", G)
            dr2 = self.new(Ra, 64)
            parts, nblk = self.gru_block_bwd(pre + "gru1.", ds, [(r2, dr2, 0), (tp_map, dtp, 0 if first_tp else 1)], Bt, H, W,
                                             True, ctx, pre + "gru1", G, bn=(pre + "bn2.", cb, st2), ng=ng)
            first_tp = False
            self._w_flush()
            if ng == 1:
                dcb = self.bn_bwd_part(pre + "bn2.", cb, dr2, st2, parts, nblk, R0, 64, training, G, out16=True)
            else:
                dcb = self.new16(Ra, 64) if self.store16 else self.new(Ra, 64)
                for q_ in range(ng):
                    self.bn_bwd_part(pre + "bn2.", cb[q_ * R0:], dr2[q_ * R0:], st2[q_], parts[q_], nblk, R0, 64, training, G, dx=dcb[q_ * R0:])
            self.conv_wgrad(ma, dcb, G[pre + "conv2.weight"], G[pre + "conv2.bias"], Bt, H, W, 64, 64, 3, bias_grad=not training)
            _, dca = self.conv_dgrad_bn(dcb, P[pre + "conv2.weight"], Bt, H, W, 64, 64, 3, pre + "bn1.", ca, st1, 2, training, G, ng=ng)
            self.conv_wgrad(blk, dca, G[pre + "conv1.weight"], G[pre + "conv1.bias"], Bt, H, W, 64, 64, 3, bias_grad=not training)
            self.conv_dgrad(dca, P[pre + "conv1.weight"], ds, Bt, H, W, 64, 64, 3, acc=1)     # dx = ds + conv1^T(dca)
            dout = ds                                            # (no flush here: the conv weight gradients wait for the next SRB's
        axpby(dout, 1.0, None, 0.0, db1, Ra * 64, 1)             #  flush, past its latency-bound GRU kernels: 9.48 -> 9.41 ms)
        on_stage("srbs")

        # ---- TP interpreter --------------------------------------------------------------------------------------
        ig, tr = "infoGen.", "infoGen.transformer."
        (tpr, t0, t, _, xg, gi, hstep, gh_all, qp, src2, _, _, _, _, _, _, _, _, dec) = ctx["tp"]
        text = ctx["text"]                                                     # per pass: (pos, qk, att_e, v1e, s1, ffn_e, v2e, mem, kin)
        Rt, K = B * 26, m.text_embedding
        dqp = self.new(Ra, 64)
        dmems, dkins = [self.new(Rt, 64) for _ in range(ng)], [self.new(Rt, 64) for _ in range(ng)]
        dnext = None                                                           # grad flowing into layer li's output from above
        fused = len(dec) > 0 and isinstance(dec[0], str)
        # The recurrent positional encoding's BPTT over the batch axis -- a chain of B latency-bound steps, the longest strand of what is
        # left of the pass -- runs on the side stream beside the encoder / block1 / STN backward.  What it needs besides dqp (zeroed
        # carries, W_hh transposed) is made now, ahead of the decoder's backward kernels; the chain itself forks the
        # moment dqp is complete (start_rpe_chain, called by _dec_bwd_fused between its launches and their reductions).
        g = tr + "gru_encoding."
        whh_f, whh_b = P[g + "weight_hh_l0"], P[g + "weight_hh_l0_reverse"]
        dsh, dsg = (B + 1) * 64 * 512, B * 64 * 1536
        main, side = torch.cuda.current_stream(), self.side_stream()
        owner = ctx.get("rpe_owner")
        if owner is None:
            # (on the main stream, ahead of the decoder's kernels: made on the side stream -- one more early fork in the captured graph --
            #  the step was 0.2 ms slower, whichever point the chain then forked at)
            carry = self.zeros(2, 64, 512)
            dgi_sum = self.zeros(2, 64, 1536)
            dgh_all = self.new(2, B, 64, 1536)
            whhT = self.new(2, 512, 1536)
            L.rpe_whh_transpose(whh_f, whh_b, whhT)
            if self.rpe_chain() and (self._rpe_sync_b is None or self._rpe_sync_b.device != self.dev):
                self._rpe_sync_b = torch.zeros(L.rpe_chain_sync_words(B), device=self.dev, dtype=torch.int32)

        def start_rpe_chain():
            if ng == 2:                                                        # both passes read the same query_pos
                axpby(dqp[R0:], 1.0, None, 0.0, dqp, R0 * 64, 1)
            if owner is not None:                                   # the chain belongs to another pass of this step (forward(rpe_from=...))
                owner["dqp_extra"] = dqp
                return
            if ctx.get("rpe_lent"):                             # gradient handed over by the pass that reused this chain
                if ctx.get("dqp_extra") is None:
                    raise RuntimeError("the pass that reused this pass's RPE chain (forward(rpe_from=...)) must run its backward first")
                axpby(ctx["dqp_extra"], 1.0, None, 0.0, dqp, R0 * 64, 1)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if self.rpe_chain():   # the whole BPTT chain as one launch of persistent workgroups (tatt_rpe_chain_bwd)
                    L.rpe_chain_bwd(gi, gh_all, P[g + "bias_hh_l0"], P[g + "bias_hh_l0_reverse"], hstep, dqp, carry, dgh_all, dgi_sum, whhT,
                                    B, dsg, dsh, self._rpe_sync_b)
                else:
                    for s_ in reversed(range(B)):
                        L.rpe_step_bwd(gi, gh_all[0, s_], P[g + "bias_hh_l0"], P[g + "bias_hh_l0_reverse"], hstep[0, s_], dqp, carry,
                                       dgh_all[0, s_ + 1] if s_ + 1 < B else None, dgh_all[0, s_], dgi_sum, whhT, s_, B, dsg, dsh)

        if fused:
            self._dec_bwd_fused(dec, dtp, db1, dqp, dkins, dmems, [tx[8] for tx in text], [tx[7] for tx in text], Bt, H * W, cfgs, G,
                                after_dqp=start_rpe_chain if RPE_FORK_EARLY else None)
            if not RPE_FORK_EARLY:
                start_rpe_chain()
        kin, mem, dkin, dmem = text[0][8], text[0][7], dkins[0], dmems[0]
        for li in (() if fused else (1, 0)):
            d = f"{tr}decoder.layers.{li}."
            tgt, qin, att, v2, t2, ffn, v3, t3 = dec[li]
            dt3 = self.new(R0, 64) if dnext is None else dnext
            self.ln_bwd(tr + "decoder.norm.", t3, dtp, 0.5, dt3, 0 if dnext is None else 1, NODROP, R0, G, False)
            dt2 = self.new(R0, 64)
            df = self.ln_bwd(d + "norm3.", v3, dt3, 1.0, dt2, 0, cfg(f"dec{li}_drop3"), R0, G, True)
            self.ffn_bwd(d, df if df is not None else dt2, ffn, t2, R0, cfg(f"dec{li}_ffn"), G, dt2)
            dtgt = self.new(R0, 64)
            da = self.ln_bwd(d + "norm2.", v2, dt2, 1.0, dtgt, 0, cfg(f"dec{li}_drop2"), R0, G, True)
            dqin = self.new(R0, 64)
            last = li == 1
            self.mha_bwd(d + "multihead_attn.", da if da is not None else dtgt, att, qin, kin, mem, B, H * W,
                         cfg(f"dec{li}_attn"), G, dqin, dkin, dmem, 0, 0 if last else 1, 0 if last else 1)
            axpby(dqin, 1.0, None, 0.0, dtgt, R0 * 64, 1)
            axpby(dqin, 1.0, None, 0.0, dqp, R0 * 64, 0 if last else 1)
            dnext = dtgt
        if not fused:
            axpby(dnext, 1.0, None, 0.0, db1, R0 * 64, 1)
        for q_ in range(ng):
            axpby(dkins[q_], 1.0, None, 0.0, dmems[q_], Rt * 64, 1)           # kin = mem + pos
        # (no flush here: the decoder's and the encoder's weight gradients stay queued until rpe_finish() -- bulk kernels beside the
        #  RPE chain slow its 64 latency-bound steps down by more than the overlap returns: 9.51 -> 9.45 ms)
        if not fused:
            start_rpe_chain()

        def rpe_tail(dirs=(0, 1), dx=True):
            """What follows the chain: its weight gradients (of the directions `dirs`), and with dx the gradient of the
            (time-invariant) input and of init_factor."""
            for dd in dirs:
                sfx = ["", "_reverse"][dd]
                self.linear_dw(dgh_all[dd], hstep[dd], G[g + "weight_hh_l0" + sfx], 64 * B, 1536, 512, fp32=True)
                self.colsum(dgh_all[dd], 64 * B, 1536, G[g + "bias_hh_l0" + sfx])
                self.linear_dw(dgi_sum[dd], xg, G[g + "weight_ih_l0" + sfx], 64, 1536, 1024, fp32=True)
                self.colsum(dgi_sum[dd], 64, 1536, G[g + "bias_ih_l0" + sfx])
            if dx:
                dxg = self.new(64, 1024)
                for dd, sfx in enumerate(["", "_reverse"]):
                    self.linear_dx(dgi_sum[dd], P[g + "weight_ih_l0" + sfx], dxg, 64, 1536, 1024, acc=dd, splitk=8, fp32=True)
                L.rpe_scatter_dx(dxg, G[ig + "init_factor.weight"], 1)

        rpe_on_side = RPE_TAIL_ON_SIDE and owner is None and side != main and self.wgrad_mode == "batch"
        chain_done = None
        if rpe_on_side:
            chain_done = torch.cuda.Event()
            chain_done.record(side)

        def rpe_finish():
            if owner is not None or rpe_on_side:
                return
            main.wait_stream(side)
            rpe_tail()

        def rpe_join():
            if rpe_on_side:
                main.wait_stream(side)
                if forked:
                    main.wait_stream(enc)
        # encoder, per pass (the text branch runs once per pass: its dropout masks differ); the passes' gradients wrt the doubled
        # encoder input (shared) are summed.  Nothing downstream on the image chain reads what this produces (parameter gradients
        # only), so the two dozen small launches run on the text branch's stream beside block1's and the STN head's backward pass,
        # with a workspace of their own; the main stream joins before the weight-gradient queue they fed is flushed.
        enc = self.enc_stream() if ENC_BWD_FORK else main
        enc.wait_stream(main)
        # (sized once, for the RPE tail that may follow on this stream: replacing it later would free memory that kernels of this
        #  strand still use to the main stream's allocator)
        need = max(1536 * 512 * self._splitk(1536, 512, 64 * B), 8 * 64 * 1024, 1 << 20)   # split-K slabs of the RPE W_hh gradient / of dxg
        if self.ws_enc is None or self.ws_enc.device != self.dev or self.ws_enc.numel() < need:
            torch.cuda.current_stream().wait_stream(enc)      # nothing of an earlier pass may still be using the old one
            self.ws_enc = torch.empty(need, device=self.dev)
        forked = enc != main
        ws_main, self.ws = self.ws, (self.ws_enc if forked else self.ws)
        self._strand = forked                                 # a guarded write must not join the weight-gradient streams from here
        with torch.cuda.stream(enc):
            e = tr + "encoder.layers.0."
            dsrc2 = self.new(Rt, 64)
            for q_, (pos, qk, att_e, v1e, s1, ffn_e, v2e, mem, kin) in enumerate(text):
                cf = cfgs[q_]
                ds1 = self.new(Rt, 64)
                dfe = self.ln_bwd(e + "norm2.", v2e, dmems[q_], 1.0, ds1, 0, cf("enc_drop2"), Rt, G, True)
                if dfe is None:
                    # p = 0: the residual branch's gradient IS ds1, which the queued weight-gradient kernels of the FFN read while its
                    # input gradient is added onto ds1 -- a guarded write, i.e. a join of the weight-gradient streams in mid-chain.
                    # The branch gets a copy instead (one 6.6 k-element launch).
                    dfe = self.new(Rt, 64)
                    L.axpby(ds1, 1.0, None, 0.0, dfe, Rt * 64, 0)
                self.ffn_bwd(e, dfe, ffn_e, s1, Rt, cf("enc_ffn"), G, ds1)
                dsq = dsrc2 if q_ == 0 else self.new(Rt, 64)
                dae = self.ln_bwd(e + "norm1.", v1e, ds1, 1.0, dsq, 0, cf("enc_drop1"), Rt, G, True)
                if dae is None:                                 # likewise for the attention branch
                    dae = self.new(Rt, 64)
                    L.axpby(dsq, 1.0, None, 0.0, dae, Rt * 64, 0)
                dqk = self.new(Rt, 64)
                self.mha_bwd(e + "self_attn.", dae, att_e, qk, qk, src2, B, 26, cf("enc_attn"), G, dqk, dqk, dsq, 0, 1, 1)
                axpby(dqk, 1.0, None, 0.0, dsq, Rt * 64, 1)
                if q_ > 0:
                    axpby(dsq, 1.0, None, 0.0, dsrc2, Rt * 64, 1)
            dt = self.new(Rt, 64)
            axpby(dsrc2, 2.0, None, 0.0, dt, Rt * 64, 0)
            dt0 = self.new(Rt, 64)
            L.prelu_bwd(t0, P[ig + "activation.weight"], dt, dt0, G[ig + "activation.weight"], 1, Rt * 64, self.ws)
            self.linear_dwb(dt0, tpr, G[ig + "fc_in.weight"], G[ig + "fc_in.bias"], Rt, 64, K)
            enc_done = torch.cuda.Event()
            enc_done.record(enc)
        self.ws, self._strand = ws_main, False
        if rpe_on_side:
            # The chain's tail stays off the main stream, which used to wait for the chain here before the STN head's backward pass
            # and now joins at the end of the pass: the input gradient and one direction's weight gradients follow the chain on its
            # own stream, the other direction's go behind the text branch (a third stream that is idle by then); the weight-gradient
            # kernels are launched in place, with workspaces of the strands' own, not through the queue of the weight-gradient streams.
            if self.ws_side_b is None or self.ws_side_b.device != self.dev or self.ws_side_b.numel() < need:
                self.ws_side_b = torch.empty(need, device=self.dev)   # (first use by this pass: nothing in flight reads an old one)
            two = forked and RPE_TAIL_SPLIT
            ws_keep, self.ws, self._wgrad_ws = self.ws, self.ws_side_b, self.ws_side_b
            with torch.cuda.stream(side):
                rpe_tail(dirs=(0,) if two else (0, 1))
            if two:
                self.ws = self._wgrad_ws = self.ws_enc
                enc.wait_event(chain_done)
                with torch.cuda.stream(enc):
                    rpe_tail(dirs=(1,), dx=False)
            self.ws, self._wgrad_ws = ws_keep, None

        def enc_join():                                           # the text branch's backward pass (not what may follow it on that stream)
            if forked:
                main.wait_event(enc_done)

        # ---- block1 ------------------------------------------------------------------------------------------------
        dc1 = self.new(Ra, 64)
        L.prelu_bwd(c1, P["block1.1.weight"], db1, dc1, G["block1.1.weight"], 1, Ra * 64, self.ws)
        xs = ctx["xs"]
        self.conv_wgrad(xs, dc1, G["block1.0.weight"], G["block1.0.bias"], Bt, H, W, 4, 64, 9)
        if not ctx["use_stn"]:
            enc_join()
            rpe_finish()
            rpe_join()
            on_stage("head")
            on_stage("infogen")
            return
        # ---- TPS + STN head, per pass ------------------------------------------------------------------------------
        dxs = self.new(Ra, 4)
        self.conv_dgrad(dc1, P["block1.0.weight"], dxs, Bt, H, W, 4, 64, 9, fp32=True)
        for q_, (stn_saved, f, l1, stat1, feat01, ctrl, Tm) in enumerate(ctx["stn_all"]):
            dctrl = self.new(B, 40)
            L.tps_sample_bwd(ctx["x"][q_], ctrl, Tm, dxs[q_ * R0:], dctrl, B, 4, H, W, 0)
            if q_ == 0:
                enc_join()                                      # the flushes below launch weight gradients queued by the text branch
            if q_ == 0 and RPE_FINISH_EARLY:
                # The RPE chain (forked after the decoder) has ended by now at the benchmark shapes: its input-gradient GEMMs go here and
                # its weight gradients join the queue, so that they overlap with the STN head's backward pass instead of trailing it.
                rpe_finish()
                self._w_flush()
            elif q_ == 0:
                self._w_flush()
            self.linear_dw(dctrl, feat01, G["stn_head.stn_fc2.weight"], B, 40, 512, fp32=True)
            self.colsum(dctrl, B, 40, G["stn_head.stn_fc2.bias"])
            dfeat = self.new(B, 512)
            self.linear_dx(dctrl, P["stn_head.stn_fc2.weight"], dfeat, B, 40, 512, fp32=True)
            axpby(dfeat, 0.1, None, 0.0, dfeat, B * 512, 0)
            if STN_SMALL and L.bn_pool_small_supported(B, 1, 1, 512, 1, 1):   # feat01 = 0.1 relu(bn(l1)): the same sign as the activation
                dl1 = self.new(B, 512)
                L.bn_pool_small_bwd(l1, feat01, dfeat, stat1, P["stn_head.stn_fc1.1.weight"], dl1, G["stn_head.stn_fc1.1.weight"],
                                    G["stn_head.stn_fc1.1.bias"], B, 1, 1, 512, 1, 1)
            else:
                dl1 = self.bn_bwd("stn_head.stn_fc1.1.", l1, dfeat, stat1, B, 512, 1, True, G)
            self.linear_dw(dl1, f, G["stn_head.stn_fc1.0.weight"], B, 512, 512, fp32=True)
            self.colsum(dl1, B, 512, G["stn_head.stn_fc1.0.bias"])
            df = self.new(B, 512)
            self.linear_dx(dl1, P["stn_head.stn_fc1.0.weight"], df, B, 512, 512, fp32=True, splitk=STN_FC_SPLITK)
            dh = self.new(B * 2, 256)
            L.nchw_to_nhwc(df, dh, B, 256, 2)
            for (idx, cin, cout, pool), rec in zip(reversed(STN_STAGES), reversed(stn_saved)):
                pre = f"stn_head.stn_convnet.{idx}."
                hin, c, stat, a, hh, ww = rec
                Rr = B * hh * ww
                kh, kw = pool or (1, 1)
                if STN_SMALL and L.bn_pool_small_supported(B, hh, ww, cout, kh, kw):   # max-pool, ReLU and BatchNorm backward: one launch
                    da, dc = dh, self.new(Rr, cout)
                    L.bn_pool_small_bwd(c, a, dh, stat, P[pre + "1.weight"], dc, G[pre + "1.weight"], G[pre + "1.bias"], B, hh, ww, cout, kh, kw)
                else:
                    if pool is not None:
                        da = self.new(Rr, cout)
                        L.maxpool_bwd(a, dh, da, B, hh, ww, cout, pool[0], pool[1])
                    else:
                        da = dh
                    dc = self.bn_bwd(pre + "1.", c, da, stat, Rr, cout, 1, True, G)
                if self.debug is not None:
                    self.debug[f"stn{idx}"] = (dh.clone(), da.clone(), dc.clone())
                self.conv_wgrad(hin, dc, G[pre + "0.weight"], G[pre + "0.bias"], B, hh, ww, cin, cout, 3, fp32=True, bias_grad=False)  # BN in training mode
                if idx > 0:
                    dh = self.new(Rr, cin)
                    self.conv_dgrad(dc, P[pre + "0.weight"], dh, B, hh, ww, cin, cout, 3, fp32=True)
                if idx in (8, 4):
                    self._w_flush()
        if not RPE_FINISH_EARLY:
            rpe_finish()
        rpe_join()
        on_stage("head")
        on_stage("infogen")


# parameters that never receive a gradient in the reference: autograd must see None for them
def unused_param(name: str) -> bool:
    return (name.startswith("infoGen.fc_feature_in.") or
            (".decoder.layers." in name and (".self_attn." in name or ".norm1." in name)))


class TSRNFunction(torch.autograd.Function):
    """Autograd boundary of the module: one node for the whole network."""

    @staticmethod
    def forward(fctx, eng, x, text_emb, training, seed, *params):
        need = any(fctx.needs_input_grad)      # grad mode is always off inside Function.forward
        sr, ctx = eng.forward(x, text_emb, training, seed, want_ctx=True)
        fctx.eng, fctx.c, fctx.need = eng, ctx if need else None, need
        fctx.names = eng._names
        fctx.req = [p.requires_grad for p in params]
        return sr

    @staticmethod
    def backward(fctx, dsr):
        eng = fctx.eng
        if fctx.c is None:
            raise RuntimeError("backward through a forward that saved no context")
        shapes = {n: eng.P[n].shape for n in fctx.names}
        total = sum(s.numel() for s in shapes.values())
        flat = torch.zeros(total, device=dsr.device, dtype=torch.float32)
        G, off = {}, 0
        for n in fctx.names:
            k = shapes[n].numel()
            G[n] = flat[off:off + k].view(shapes[n])
            off += k
        eng.backward(fctx.c, dsr, G)
        grads = [G[n] if (req and not unused_param(n)) else None for n, req in zip(fctx.names, fctx.req)]
        return (None, None, None, None, None, *grads)
