"""CRNN text-prior generator on the HIP kernels — forward and backward.

Reference: model/crnn/crnn.py:5-42 (BidirectionalLSTM) and :45-119 (CRNN); `parse_crnn_data` interfaces/base.py:441-459;
the prior handed to the SR network is `softmax(logits).permute(1,0,2).unsqueeze(1).permute(0,3,1,2)`
(interfaces/super_resolution_eng.py:227-231).  Same class names, constructor arguments, `state_dict` keys and shapes
(`cnn.conv{0..6}`, `cnn.batchnorm{2,4,6}`, `rnn.{0,1}.rnn.*`, `rnn.{0,1}.embedding.*`) and call
`CRNN(...)(gray[B,1,32,100]) -> logits[26,B,nclass]`, attached to autograd through one `torch.autograd.Function`
(teacher: call under `torch.no_grad()` / `.eval()`; student: `.train()` and backpropagate).

The torch modules below (`nn.Conv2d`, `nn.BatchNorm2d`, `nn.LSTM`, `nn.Linear`) are parameter containers that give the
reference's names, shapes and default initialisation; their `forward` is never called.  torch is used for device memory
and pure data movement (slicing, transposed copies); every arithmetic step is a kernel of libtatt_hip.so — the trunk on
the SR path's conv / BatchNorm / GEMM kernels, the rest in csrc/crnn.hip.  There is no CPU fallback.
"""
import os

import torch
from torch import nn

from ..lib import lib

WS_FLOATS = 24 * 1024 * 1024
NODROP = (0, 0, 1.0, None)
# (index, Cin, Cout, batchnorm?, pool after) of the 3x3 layers 1..5; layer 0 (Cin = 1) and 6 (2x2, no padding) are special
TRUNK = [(1, 64, 128, False, (2, 2, 2, 2, 0, 0)), (2, 128, 256, True, None), (3, 256, 256, False, (2, 2, 2, 1, 0, 1)),
         (4, 256, 512, True, None), (5, 512, 512, False, (2, 2, 2, 1, 0, 1))]


class BidirectionalLSTM(nn.Module):
    def __init__(self, nIn, nHidden, nOut):
        super().__init__()
        self.rnn = nn.LSTM(nIn, nHidden, bidirectional=True)
        self.embedding = nn.Linear(nHidden * 2, nOut)


def _pool_out(H, W, k):
    KH, KW, SH, SW, PH, PW = k
    return (H + 2 * PH - KH) // SH + 1, (W + 2 * PW - KW) // SW + 1


class _Plan:
    """Launch plan of one CRNN call: explicit forward (optionally saving what the backward needs) and hand-derived backward."""

    def __init__(self, mod):
        self.m, self.L = mod, lib()
        self.ws = None
        # the T steps of each nn.LSTM as one launch of persistent workgroups (tatt_lstm256_chain_fwd / _bwd) where the device can hold
        # the chains; TATT_LSTM_CHAIN=0 (A/B runs), other devices and B > 128: one launch per step
        self.lstm_chain = os.environ.get("TATT_LSTM_CHAIN", "1") != "0"
        self._sync_f = self._sync_b = None

    def _chain(self, B):
        return self.lstm_chain and B <= 128 and bool(self.L.rpe_chain_supported())

    def _sync(self, which):
        """The chain launches' flag buffer (zeroed once: word 1 is a sticky error flag), one per direction of the pass; the two
        LSTM layers run one after the other on one stream and share it."""
        t = getattr(self, which)
        if t is None or t.device != self.dev:
            t = torch.zeros(self.L.lstm256_chain_sync_words(), device=self.dev, dtype=torch.int32)
            setattr(self, which, t)
        return t

    def check_lstm_chain(self):
        """Raise if a chain launch of this plan ever reported a timed-out wait (synchronises; call at check points)."""
        for which in ("_sync_f", "_sync_b"):
            t = getattr(self, which)
            if t is not None and int(t[1].item()) != 0:
                raise RuntimeError(f"LSTM chain kernel ({which}) gave up waiting for its predecessor step: results are invalid; "
                                   "set TATT_LSTM_CHAIN=0 to run the recurrence as one launch per step")

    def new(self, *shape):
        return torch.empty(shape, device=self.dev, dtype=torch.float32)

    def params(self):
        P = dict(self.m.named_parameters())
        P.update(dict(self.m.named_buffers()))
        out = {}
        for k, v in P.items():
            if not v.is_cuda:
                raise RuntimeError("tatt_amd CRNN runs on the GPU only (no CPU fallback)")
            v = v.detach()
            if v.is_floating_point() and (v.dtype != torch.float32 or not v.is_contiguous()):
                # the reference leaves `embedding` in float64 after its first call (crnn.py:31); the values are what matter
                if "running_" in k:
                    raise RuntimeError(f"buffer {k} must be contiguous fp32 (it is updated in place)")
                v = v.float().contiguous()
            out[k] = v
        prec = getattr(self.m, "matmul_precision", "fp32")
        if prec not in ("fp32", "bf16", "f32x3"):
            raise RuntimeError(f"matmul_precision must be 'fp32', 'bf16' or 'f32x3', got {prec!r}")
        self.pf = {"fp32": 0, "bf16": 2, "f32x3": 4}[prec]   # TATT_BF16 / TATT_F32X3 flag of the GEMM / convolution entry points (per call)
        self.dev = next(iter(out.values())).device
        if self.ws is None or self.ws.device != self.dev:
            self.ws = torch.empty(WS_FLOATS, device=self.dev)
        return out

    # ---- forward pieces ---------------------------------------------------------------------------------------------
    def conv3(self, P, i, x, B, H, W, Cin, Cout, relu, packs):
        L = self.L
        y = self.new(B * H * W, Cout)
        wf, wd = self.new(Cout * 9 * Cin), self.new(Cout * 9 * Cin)
        L.conv_pack_w(P[f"cnn.conv{i}.weight"], wf, wd, Cout, Cin, 3, 3)
        packs[i] = wd
        L.conv_fwd(x, wf, P[f"cnn.conv{i}.bias"], y, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 9 * Cin, 1, 0, Cin, 0, relu | self.pf)
        return y

    def bn_relu(self, P, i, x, R, C, training):
        L, pre = self.L, f"cnn.batchnorm{i}."
        stat = self.new(2, C)
        if training:
            L.bn_stats(x, C, R, C, 1e-5, 0.1, stat, P[pre + "running_mean"], P[pre + "running_var"], self.ws)
            P[pre + "num_batches_tracked"] += 1
        else:
            L.bn_stat_from_running(P[pre + "running_mean"], P[pre + "running_var"], C, 1e-5, stat)
        y = self.new(R, C)
        L.bn_apply(x, C, stat, P[pre + "weight"], P[pre + "bias"], None, C, y, C, R, C, 1)
        return y, stat

    def bilstm(self, P, pre, x, I, T, B, save):
        """x [T*B, I] time-major -> (rec [T*B, 512], saved gates or None)."""
        L, r = self.L, pre + "rnn."
        gi = self.new(T * B, 2048)
        for d, sfx in enumerate(["", "_reverse"]):
            L.gemm(x, P[r + "weight_ih_l0" + sfx], gi[:, d * 1024:], P[r + "bias_ih_l0" + sfx], T * B, 1024, I, I, 1, I, 1, 2048, 1, 0, 0, 0,
                   0, self.pf, 1, None)
        rec = self.new(T * B, 512)
        sv = self.new(T * B, 2, 5, 256) if save else None
        if self._chain(B):
            L.lstm256_chain_fwd(gi, P[r + "weight_hh_l0"], P[r + "weight_hh_l0_reverse"], P[r + "bias_hh_l0"], P[r + "bias_hh_l0_reverse"],
                                rec, sv, T, B, self._sync("_sync_f"))
            return rec, sv
        h = torch.zeros(2, 2, B, 256, device=self.dev)               # ping-pong [buf][dir][B][256]
        c = torch.zeros(2, 2, B, 256, device=self.dev)
        for s in range(T):
            a, b = s & 1, (s + 1) & 1
            L.lstm256_step_fwd(gi, P[r + "weight_hh_l0"], P[r + "weight_hh_l0_reverse"], P[r + "bias_hh_l0"],
                               P[r + "bias_hh_l0_reverse"], h[a], c[a], h[b], c[b], rec, sv, s, T, B)
        return rec, sv

    def forward(self, x4, B, W, training, want_ctx):
        L, P = self.L, self.params()
        dev, H = self.dev, 32
        ctx = {"B": B, "W": W, "training": training, "x4": x4, "packs": {}} if want_ctx else {"packs": {}}
        packs = ctx["packs"]
        # conv0: Cin = 1, weights zero-padded to 4 input channels ([co][tap][4])
        w0 = torch.zeros(64, 9, 4, device=dev)
        w0[:, :, 0] = P["cnn.conv0.weight"].view(64, 9)
        y = self.new(B * H * W, 64)
        L.conv_fwd(x4, w0, P["cnn.conv0.bias"], y, B, H, W, 4, 64, 3, 3, 4, 64, 36, 1, 0, 4, 0, 1)
        p = self.new(B * (H // 2) * (W // 2), 64)
        L.maxpool_fwd(y, p, B, H, W, 64, 2, 2)
        acts = [("relu_pool", y, (2, 2, 2, 2, 0, 0), H, W)]
        H, W = H // 2, W // 2                                                        # 16 x 50
        x = p
        for i, Cin, Cout, has_bn, pool in TRUNK:
            cpre = self.conv3(P, i, x, B, H, W, Cin, Cout, 0 if has_bn else 1, packs)
            if has_bn:
                a, stat = self.bn_relu(P, i, cpre, B * H * W, Cout, training)
                acts.append(("bn", x, cpre, stat, H, W))
                x = a
            else:
                if pool is not None:
                    H2, W2 = _pool_out(H, W, pool)
                    pp = self.new(B * H2 * W2, Cout)
                    if pool == (2, 2, 2, 2, 0, 0):
                        L.maxpool_fwd(cpre, pp, B, H, W, Cout, 2, 2)
                    else:
                        L.maxpool_gen_fwd(cpre, pp, B, H, W, Cout, *pool)
                    acts.append(("conv_relu_pool", x, cpre, pool, H, W))
                    x, H, W = pp, H2, W2
        if H != 2:
            raise RuntimeError("the height of conv must be 1")
        # conv6 (2x2, no padding) as one GEMM over an im2col copy whose k order (ci, dy, dx) is the OIHW weight's own
        T = W - 1
        p5 = x.view(B, 2, W, 512)
        cols = torch.stack([p5[:, dy, dx:dx + T, :] for dy in (0, 1) for dx in (0, 1)], dim=-1).reshape(B * T, 2048)
        c6 = self.new(B * T, 512)
        L.gemm(cols, P["cnn.conv6.weight"], c6, P["cnn.conv6.bias"], B * T, 512, 2048, 2048, 1, 2048, 1, 512, 1, 0, 0, 0, 0, self.pf, 1, None)
        feat, stat6 = self.bn_relu(P, 6, c6, B * T, 512, training)                  # rows (b, t)
        featT = feat.view(B, T, 512).transpose(0, 1).contiguous().view(T * B, 512)   # time-major rows (t, b)
        rec0, sv0 = self.bilstm(P, "rnn.0.", featT, 512, T, B, want_ctx)
        mid = self.new(T * B, 256)
        L.linear_f64(rec0, 512, P["rnn.0.embedding.weight"], P["rnn.0.embedding.bias"], mid, 256, T * B, 256, 512)
        rec1, sv1 = self.bilstm(P, "rnn.1.", mid, 256, T, B, want_ctx)
        K = self.m.nclass
        logits = self.new(T * B, K)
        L.linear_f64(rec1, 512, P["rnn.1.embedding.weight"], P["rnn.1.embedding.bias"], logits, K, T * B, K, 512)
        if want_ctx:
            ctx.update(acts=acts, cols=cols, c6=c6, stat6=stat6, W5=W, T=T, featT=featT, rec0=rec0, sv0=sv0, mid=mid, rec1=rec1, sv1=sv1)
        return logits.view(T, B, K), (ctx if want_ctx else None)

    # ---- backward pieces ----------------------------------------------------------------------------------------------
    def f64_linear_bwd(self, P, pre, dY, X, R, N, K, G):
        """embedding Linear, float64 arithmetic like the forward: returns dX [R,K]; G[weight] += dY^T X, G[bias] += colsum(dY)."""
        L, Wt = self.L, P[pre + "embedding.weight"]
        WT, dYT, XT = self.new(K, N), self.new(N, R), self.new(K, R)
        L.transpose(Wt, WT, N, K)
        L.transpose(dY, dYT, R, N)
        L.transpose(X, XT, R, K)
        dX = self.new(R, K)
        L.linear_f64(dY, N, WT, None, dX, K, R, K, N)
        dW, db = self.new(N, K), self.new(N)
        L.linear_f64(dYT, R, XT, None, dW, K, N, K, R)
        L.linear_f64(dYT, R, torch.ones(R, device=self.dev), None, db, 1, N, 1, R)
        L.axpby(dW, 1.0, None, 0.0, G[pre + "embedding.weight"], N * K, 1)
        L.axpby(db, 1.0, None, 0.0, G[pre + "embedding.bias"], N, 1)
        return dX

    def splitk(self, M, N, K):
        tiles = -(-M // 128) * -(-N // 64)
        s = max(1, min(K // 256, -(-768 // tiles)))
        while s > 1 and M * N * s > WS_FLOATS:
            s //= 2
        return s

    def dw(self, dy, lddy, x, ldx, dW, R, N, K):
        """dW[N,K] += dy[R,N]^T x[R,K]"""
        sk = self.splitk(N, K, R)
        self.L.gemm(dy, x, dW, None, N, K, R, 1, lddy, 1, ldx, K, 1, 0, 0, 0, 1, self.pf, sk, self.ws if sk > 1 else None)

    def bilstm_bwd(self, P, pre, drec, sv, X, I, rec, T, B, G):
        """BPTT of one BidirectionalLSTM's nn.LSTM: returns dX [T*B, I] (time-major); accumulates the eight parameter grads."""
        L, r = self.L, pre + "rnn."
        whhT = self.new(2, 256, 1024)
        for d, sfx in enumerate(["", "_reverse"]):
            L.transpose(P[r + "weight_hh_l0" + sfx], whhT[d], 1024, 256)
        dgates = self.new(T * B, 2048)
        if self._chain(B):
            L.lstm256_chain_bwd(sv, whhT, drec, dgates, T, B, self._sync("_sync_b"))
        else:
            dcc = torch.zeros(2, B, 256, device=self.dev)
            for s in reversed(range(T)):
                L.lstm256_step_bwd(sv, whhT, drec, dgates, dcc, s, T, B)
        dX = self.new(T * B, I)
        for d, sfx in enumerate(["", "_reverse"]):
            g = dgates[:, d * 1024:]
            # h_{prev}: the recurrent output one step earlier in the direction's own order (zero at its first step)
            if d == 0:
                self.dw(g[B:], 2048, rec[:(T - 1) * B, :256], 512, G[r + "weight_hh_l0"], (T - 1) * B, 1024, 256)
            else:
                self.dw(g[:(T - 1) * B], 2048, rec[B:, 256:], 512, G[r + "weight_hh_l0_reverse"], (T - 1) * B, 1024, 256)
            self.dw(g, 2048, X, I, G[r + "weight_ih_l0" + sfx], T * B, 1024, I)
            L.colsum(g, 2048, T * B, 1024, G[r + "bias_ih_l0" + sfx], 1, self.ws)
            L.colsum(g, 2048, T * B, 1024, G[r + "bias_hh_l0" + sfx], 1, self.ws)
            L.gemm(g, P[r + "weight_ih_l0" + sfx], dX, None, T * B, I, 1024, 2048, 1, 1, I, I, 1, 0, 0, 0, d, self.pf, 1, None)
        return dX

    def conv3_bwd(self, P, i, x, dc, B, H, W, Cin, Cout, wd, G, need_dx=True):
        """dc = gradient at the conv output (pre-activation): weight / bias grads, returns dx."""
        L = self.L
        T9 = 9
        if Cout <= 16:
            tiles = -(-Cin // 128) * -(-(T9 * Cout) // 64)
        elif Cout <= 64:
            tiles = -(-(T9 * Cin) // 128)
        else:
            tiles = -(-Cout // 128) * -(-(T9 * Cin) // 64)
        sk = max(1, min((B * H * W) // 256, -(-1024 // tiles)))
        while sk > 1 and Cout * T9 * Cin * sk > WS_FLOATS:
            sk //= 2
        L.conv_wgrad(x, dc, G[f"cnn.conv{i}.weight"], B, H, W, Cin, Cout, 3, 3, Cin, Cout, sk, self.ws, 1 | self.pf)
        L.colsum(dc, Cout, B * H * W, Cout, G[f"cnn.conv{i}.bias"], 1, self.ws)
        if not need_dx:
            return None
        dx = self.new(B * H * W, Cin)
        L.conv_fwd(dc, wd, None, dx, B, H, W, Cout, Cin, 3, 3, Cout, Cin, 9 * Cout, 1, 0, Cout, 0, self.pf)
        return dx

    def backward(self, ctx, dlogits, G):
        """Accumulates parameter gradients of one forward call into G (dict name -> fp32 tensor shaped like the parameter)."""
        L, P = self.L, self.params()
        B, T, K, training = ctx["B"], ctx["T"], self.m.nclass, ctx["training"]
        dlog = dlogits.contiguous().float().view(T * B, K)
        drec1 = self.f64_linear_bwd(P, "rnn.1.", dlog, ctx["rec1"], T * B, K, 512, G)
        dmid = self.bilstm_bwd(P, "rnn.1.", drec1, ctx["sv1"], ctx["mid"], 256, ctx["rec1"], T, B, G)
        drec0 = self.f64_linear_bwd(P, "rnn.0.", dmid, ctx["rec0"], T * B, 256, 512, G)
        dfeatT = self.bilstm_bwd(P, "rnn.0.", drec0, ctx["sv0"], ctx["featT"], 512, ctx["rec0"], T, B, G)
        dfeat = dfeatT.view(T, B, 512).transpose(0, 1).contiguous().view(B * T, 512)     # back to rows (b, t)
        # batchnorm6 + relu, conv6
        dc6 = self.new(B * T, 512)
        pre = "cnn.batchnorm6."
        L.bn_bwd(ctx["c6"], 512, dfeat, 512, ctx["stat6"], P[pre + "weight"], P[pre + "bias"], dc6, 512, G[pre + "weight"], G[pre + "bias"],
                 B * T, 512, 1, 1 if training else 0, 1, 0, self.ws)
        L.colsum(dc6, 512, B * T, 512, G["cnn.conv6.bias"], 1, self.ws)
        self.dw(dc6, 512, ctx["cols"], 2048, G["cnn.conv6.weight"].view(512, 2048), B * T, 512, 2048)
        W5 = ctx["W5"]
        dx = torch.zeros(B, 2, W5, 512, device=self.dev)
        w6 = P["cnn.conv6.weight"].view(-1)
        for dy in (0, 1):
            for dxx in (0, 1):                       # dP[b, dy, t+dx, ci] += sum_co dc6[b, t, co] W[co, ci, dy, dx]
                L.gemm(dc6, w6[dy * 2 + dxx:], dx.view(-1)[(dy * W5 + dxx) * 512:], None, T, 512, 512, 512, 1, 4, 2048, 512, B,
                       T * 512, 0, 2 * W5 * 512, 1, self.pf, 1, None)
        dx = dx.view(B * 2 * W5, 512)
        # the 3x3 trunk, last layer first
        acts = ctx["acts"]
        H, W = 2, W5
        for (i, Cin, Cout, has_bn, pool), act in zip(reversed(TRUNK), reversed(acts[1:])):
            if has_bn:
                _, xin, cpre, stat, H, W = act
                pre = f"cnn.batchnorm{i}."
                dc = self.new(B * H * W, Cout)
                L.bn_bwd(cpre, Cout, dx, Cout, stat, P[pre + "weight"], P[pre + "bias"], dc, Cout, G[pre + "weight"], G[pre + "bias"],
                         B * H * W, Cout, 1, 1 if training else 0, 1, 0, self.ws)
            else:
                _, xin, y, pool, H, W = act                      # y = relu(conv) before the pool
                dyp = self.new(B * H * W, Cout)
                if pool == (2, 2, 2, 2, 0, 0):
                    L.maxpool_bwd(y, dx, dyp, B, H, W, Cout, 2, 2)
                else:
                    L.maxpool_gen_bwd(y, dx, dyp, B, H, W, Cout, *pool)
                dc = self.new(B * H * W, Cout)
                L.relu_drop_bwd(y, dyp, dc, B * H * W * Cout, *NODROP)
            dx = self.conv3_bwd(P, i, xin, dc, B, H, W, Cin, Cout, ctx["packs"][i], G)
        # conv0 (+relu, 2x2 pool): weight gradient through the 4-channel padded view, no input gradient
        _, y0, _, H, W = acts[0]
        dyp = self.new(B * H * W, 64)
        L.maxpool_bwd(y0, dx, dyp, B, H, W, 64, 2, 2)
        dc0 = self.new(B * H * W, 64)
        L.relu_drop_bwd(y0, dyp, dc0, B * H * W * 64, *NODROP)
        dW4 = torch.zeros(64, 4, 3, 3, device=self.dev)
        sk = max(1, min((B * H * W) // 256, 256))
        L.conv_wgrad(ctx["x4"], dc0, dW4, B, H, W, 4, 64, 3, 3, 4, 64, sk, self.ws, 1)
        L.axpby(dW4[:, :1].contiguous(), 1.0, None, 0.0, G["cnn.conv0.weight"], 64 * 9, 1)
        L.colsum(dc0, 64, B * H * W, 64, G["cnn.conv0.bias"], 1, self.ws)


class _CRNNFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, plan, x4, B, W, training, *params):
        need = any(fctx.needs_input_grad)
        logits, ctx = plan.forward(x4, B, W, training, want_ctx=need)
        fctx.plan, fctx.c = plan, ctx
        fctx.names = [n for n, _ in plan.m.named_parameters()]
        fctx.req = [p.requires_grad for p in params]
        return logits

    @staticmethod
    def backward(fctx, dlogits):
        plan = fctx.plan
        if fctx.c is None:
            raise RuntimeError("backward through a CRNN forward that saved no context")
        P = dict(plan.m.named_parameters())
        G = {n: torch.zeros(P[n].shape, device=dlogits.device, dtype=torch.float32) for n in fctx.names}
        plan.backward(fctx.c, dlogits, G)
        grads = [G[n].to(P[n].dtype) if req else None for n, req in zip(fctx.names, fctx.req)]
        return (None, None, None, None, None, *grads)


class CRNN(nn.Module):
    def __init__(self, imgH=32, nc=1, nclass=37, nh=256, n_rnn=2, leakyRelu=False):
        super().__init__()
        assert imgH % 16 == 0, 'imgH has to be a multiple of 16'
        if (imgH, nc, nh, leakyRelu) != (32, 1, 256, False):
            raise NotImplementedError("tatt_amd CRNN: imgH=32, nc=1, nh=256, ReLU (the configuration the reference builds)")
        ks, ps, nm = [3, 3, 3, 3, 3, 3, 2], [1, 1, 1, 1, 1, 1, 0], [64, 128, 256, 256, 512, 512, 512]
        cnn = nn.Sequential()
        for i in range(7):
            cnn.add_module(f"conv{i}", nn.Conv2d(nc if i == 0 else nm[i - 1], nm[i], ks[i], 1, ps[i]))
            if i in (2, 4, 6):
                cnn.add_module(f"batchnorm{i}", nn.BatchNorm2d(nm[i]))
        self.cnn = cnn
        self.rnn = nn.Sequential(BidirectionalLSTM(512, nh, nh), BidirectionalLSTM(nh, nh, nclass))
        self.nclass = nclass
        self.matmul_precision = "fp32"               # or "bf16" / "f32x3": see tatt_amd.model.tsrn.TSRN_TL_TRANS.matmul_precision
        self._plan = None

    def plan(self):
        if self._plan is None:
            self._plan = _Plan(self)
        return self._plan

    def _run(self, x4, B, W):
        params = [p for _, p in self.named_parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _CRNNFn.apply(self.plan(), x4, B, W, self.training, *params)
        return self.plan().forward(x4, B, W, self.training, want_ctx=False)[0]

    def forward(self, input):
        """input: gray [B,1,32,W] (what parse_crnn_data returns) -> logits [W/4+1, B, nclass]."""
        if not input.is_cuda:
            raise RuntimeError("tatt_amd CRNN runs on the GPU only (no CPU fallback)")
        B, C, H, W = input.shape
        if C != 1 or H != 32 or W % 4 != 0:
            raise RuntimeError(f"expected gray input [B,1,32,W] with W % 4 == 0, got {tuple(input.shape)}")
        x4 = torch.zeros(B, H, W, 4, device=input.device)
        x4[..., 0] = input.detach().float()[:, 0]                # NCHW with C=1 is NHWC; pad C to 4 for the conv kernel
        return self._run(x4, B, W)

    def forward_from_images(self, imgs):
        """parse_crnn_data + forward in one go: imgs [B,>=3,H,W] in [0,1] (LR 16x64 or HR 32x128)."""
        if not imgs.is_cuda:
            raise RuntimeError("tatt_amd CRNN runs on the GPU only (no CPU fallback)")
        imgs = imgs.detach().contiguous().float()
        B, C, H, W = imgs.shape
        x4 = torch.empty(B, 32, 100, 4, device=imgs.device)
        lib().crnn_prep(imgs, x4, B, C, H, W, 32, 100)
        return self._run(x4, B, 100)


def parse_crnn_data(imgs_input_, ratio_keep=False):
    """TextBase.parse_crnn_data (interfaces/base.py:441-459): bicubic resize to 32x100 + gray -> [B,1,32,100]."""
    if ratio_keep:
        raise NotImplementedError("tatt_amd parse_crnn_data: ratio_keep is not used by the TATT trainer and is not built")
    if not imgs_input_.is_cuda:
        raise RuntimeError("tatt_amd parse_crnn_data runs on the GPU only")
    imgs = imgs_input_.detach().contiguous().float()
    B, C, H, W = imgs.shape
    x4 = torch.empty(B, 32, 100, 4, device=imgs.device)
    lib().crnn_prep(imgs, x4, B, C, H, W, 32, 100)
    return x4[..., 0].unsqueeze(1).contiguous()


def text_prior(logits):
    """label_vecs_final (interfaces/super_resolution_eng.py:227-231): [T,B,K] logits -> the [B,K,1,T] view the SR net takes.
    The softmax over 37 classes of a [26,B,37] tensor is torch plumbing here; the SR engine reads the view by strides."""
    return torch.softmax(logits, -1).permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
