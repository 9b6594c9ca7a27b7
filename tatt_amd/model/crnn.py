"""CRNN text-prior generator on the HIP kernels — forward pass.

Reference: model/crnn/crnn.py:5-42 (BidirectionalLSTM) and :45-119 (CRNN); `parse_crnn_data` interfaces/base.py:441-459;
the prior handed to the SR network is `softmax(logits).permute(1,0,2).unsqueeze(1).permute(0,3,1,2)`
(interfaces/super_resolution_eng.py:227-231).  Same class names, constructor arguments, `state_dict` keys and shapes
(`cnn.conv{0..6}`, `cnn.batchnorm{2,4,6}`, `rnn.{0,1}.rnn.*`, `rnn.{0,1}.embedding.*`) and call
`CRNN(...)(gray[B,1,32,100]) -> logits[26,B,nclass]`.

Scope of this round: the FORWARD pass (teacher / frozen prior generator, eval or train-mode BatchNorm).  The returned
tensor carries no autograd graph; the student's backward pass is not built yet and asking for it raises.
The torch modules below (`nn.Conv2d`, `nn.BatchNorm2d`, `nn.LSTM`, `nn.Linear`) are parameter containers that give the
reference's names, shapes and default initialisation; their `forward` is never called.
"""
import torch
from torch import nn

from ..lib import lib


class BidirectionalLSTM(nn.Module):
    def __init__(self, nIn, nHidden, nOut):
        super().__init__()
        self.rnn = nn.LSTM(nIn, nHidden, bidirectional=True)
        self.embedding = nn.Linear(nHidden * 2, nOut)


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
        self._ws = None

    # ---- helpers -------------------------------------------------------------------------------------------------
    def _P(self):
        P = dict(self.named_parameters())
        P.update(dict(self.named_buffers()))
        for k, v in P.items():
            if not v.is_cuda:
                raise RuntimeError("tatt_amd CRNN runs on the GPU only (no CPU fallback)")
            if v.is_floating_point() and (v.dtype != torch.float32 or not v.is_contiguous()):
                # the reference leaves `embedding` in float64 after its first call (crnn.py:31); values are what matter
                P[k] = v.detach().float().contiguous()
            else:
                P[k] = v.detach()
        return P

    def _conv3(self, L, P, i, x, B, H, W, Cin, Cout, relu, ws):
        y = torch.empty(B * H * W, Cout, device=x.device)
        wf, wd = torch.empty(Cout * 9 * Cin, device=x.device), torch.empty(Cout * 9 * Cin, device=x.device)
        L.conv_pack_w(P[f"cnn.conv{i}.weight"], wf, wd, Cout, Cin, 3, 3)
        L.conv_fwd(x, wf, P[f"cnn.conv{i}.bias"], y, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 9 * Cin, 1, 0, Cin, 0, relu)
        return y

    def _bn_relu(self, L, P, i, x, R, C, ws):
        pre = f"cnn.batchnorm{i}."
        stat = torch.empty(2, C, device=x.device)
        if self.training:
            L.bn_stats(x, C, R, C, 1e-5, 0.1, stat, P[pre + "running_mean"], P[pre + "running_var"], ws)
            P[pre + "num_batches_tracked"] += 1
        else:
            L.bn_stat_from_running(P[pre + "running_mean"], P[pre + "running_var"], C, 1e-5, stat)
        y = torch.empty(R, C, device=x.device)
        L.bn_apply(x, C, stat, P[pre + "weight"], P[pre + "bias"], None, C, y, C, R, C, 1)
        return y

    def _bilstm(self, L, P, pre, gi, T, B):
        """gi [T*B, 2048] (both directions' W_ih x + b_ih) -> recurrent outputs [T*B, 512]."""
        dev = gi.device
        h = torch.zeros(2, 2, B, 256, device=dev)               # ping-pong [buf][dir][B][256]
        c = torch.zeros(2, 2, B, 256, device=dev)
        out = torch.empty(T * B, 512, device=dev)
        r = pre + "rnn."
        for s in range(T):
            a, b = s & 1, (s + 1) & 1
            L.lstm256_step_fwd(gi, P[r + "weight_hh_l0"], P[r + "weight_hh_l0_reverse"], P[r + "bias_hh_l0"],
                               P[r + "bias_hh_l0_reverse"], h[a], c[a], h[b], c[b], out, s, T, B)
        return out

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, input):
        """input: gray [B,1,32,W] (what parse_crnn_data returns) -> logits [W/4+1, B, nclass]."""
        if not input.is_cuda:
            raise RuntimeError("tatt_amd CRNN runs on the GPU only (no CPU fallback)")
        if torch.is_grad_enabled() and input.requires_grad:
            raise NotImplementedError("tatt_amd CRNN: the backward pass (student training) is not built yet")
        B, C, H, W = input.shape
        if C != 1 or H != 32:
            raise RuntimeError(f"expected gray input [B,1,32,W], got {tuple(input.shape)}")
        x4 = torch.zeros(B, H, W, 4, device=input.device)
        x4[..., 0] = input.detach().float()[:, 0]                # NCHW with C=1 is NHWC; pad C to 4 for the conv kernel
        return self._forward_nhwc4(x4, B, W)

    def forward_from_images(self, imgs):
        """parse_crnn_data + forward in one go: imgs [B,>=3,H,W] in [0,1] (LR 16x64 or HR 32x128)."""
        if not imgs.is_cuda:
            raise RuntimeError("tatt_amd CRNN runs on the GPU only (no CPU fallback)")
        imgs = imgs.detach().contiguous().float()
        B, C, H, W = imgs.shape
        x4 = torch.empty(B, 32, 100, 4, device=imgs.device)
        lib().crnn_prep(imgs, x4, B, C, H, W, 32, 100)
        return self._forward_nhwc4(x4, B, 100)

    def _forward_nhwc4(self, x4, B, W):
        L, P, dev = lib(), self._P(), x4.device
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(1 << 20, device=dev)
        ws = self._ws
        # conv0: Cin = 1, weights zero-padded to 4 input channels ([co][tap][4])
        w0 = torch.zeros(64, 9, 4, device=dev)
        w0[:, :, 0] = P["cnn.conv0.weight"].view(64, 9)
        y = torch.empty(B * 32 * W, 64, device=dev)
        L.conv_fwd(x4, w0, P["cnn.conv0.bias"], y, B, 32, W, 4, 64, 3, 3, 4, 64, 36, 1, 0, 4, 0, 1)
        H = 32
        p = torch.empty(B * (H // 2) * (W // 2), 64, device=dev)
        L.maxpool_fwd(y, p, B, H, W, 64, 2, 2)
        H, W = H // 2, W // 2                                                        # 16 x 50
        y = self._conv3(L, P, 1, p, B, H, W, 64, 128, 1, ws)
        p = torch.empty(B * (H // 2) * (W // 2), 128, device=dev)
        L.maxpool_fwd(y, p, B, H, W, 128, 2, 2)
        H, W = H // 2, W // 2                                                        # 8 x 25
        y = self._bn_relu(L, P, 2, self._conv3(L, P, 2, p, B, H, W, 128, 256, 0, ws), B * H * W, 256, ws)
        y = self._conv3(L, P, 3, y, B, H, W, 256, 256, 1, ws)
        H2, W2 = (H - 2) // 2 + 1, (W + 2 - 2) // 1 + 1                              # MaxPool2d((2,2),(2,1),(0,1)): 4 x 26
        p = torch.empty(B * H2 * W2, 256, device=dev)
        L.maxpool_gen_fwd(y, p, B, H, W, 256, 2, 2, 2, 1, 0, 1)
        H, W = H2, W2
        y = self._bn_relu(L, P, 4, self._conv3(L, P, 4, p, B, H, W, 256, 512, 0, ws), B * H * W, 512, ws)
        y = self._conv3(L, P, 5, y, B, H, W, 512, 512, 1, ws)
        H2, W2 = (H - 2) // 2 + 1, W + 1                                             # 2 x 27
        p = torch.empty(B * H2 * W2, 512, device=dev)
        L.maxpool_gen_fwd(y, p, B, H, W, 512, 2, 2, 2, 1, 0, 1)
        H, W = H2, W2
        if H != 2:
            raise RuntimeError("the height of conv must be 1")
        # conv6: 2x2, no padding -> [B, 1, W-1, 512]; two batched GEMMs (one per kernel row) over 1024 contiguous inputs
        T = W - 1
        wf, wd = torch.empty(512 * 4 * 512, device=dev), torch.empty(512 * 4 * 512, device=dev)
        L.conv_pack_w(P["cnn.conv6.weight"], wf, wd, 512, 512, 2, 2)                 # wf[co][dy*2+dx][ci]
        c6 = torch.empty(B, T, 512, device=dev)
        for dy in range(2):
            L.gemm(p.view(-1)[dy * W * 512:], wf[dy * 1024:], c6, P["cnn.conv6.bias"] if dy == 0 else None, T, 512, 1024,
                   512, 1, 2048, 1, 512, B, 2 * W * 512, 0, T * 512, dy, 0, 1, None)
        feat = self._bn_relu(L, P, 6, c6, B * T, 512, ws)                            # rows (b, t)
        # BidirectionalLSTM 0: time-major input projections through a batched GEMM (batch = t, rows = b)
        gi = torch.empty(T * B, 2048, device=dev)
        for d, sfx in enumerate(["", "_reverse"]):
            L.gemm(feat, P["rnn.0.rnn.weight_ih_l0" + sfx], gi[:, d * 1024:], P["rnn.0.rnn.bias_ih_l0" + sfx], B, 1024, 512,
                   T * 512, 1, 512, 1, 2048, T, 512, 0, B * 2048, 0, 0, 1, None)
        rec = self._bilstm(L, P, "rnn.0.", gi, T, B)
        mid = torch.empty(T * B, 256, device=dev)
        L.linear_f64(rec, 512, P["rnn.0.embedding.weight"], P["rnn.0.embedding.bias"], mid, 256, T * B, 256, 512)
        # BidirectionalLSTM 1
        for d, sfx in enumerate(["", "_reverse"]):
            L.gemm(mid, P["rnn.1.rnn.weight_ih_l0" + sfx], gi[:, d * 1024:], P["rnn.1.rnn.bias_ih_l0" + sfx], T * B, 1024, 256,
                   256, 1, 256, 1, 2048, 1, 0, 0, 0, 0, 0, 1, None)
        rec = self._bilstm(L, P, "rnn.1.", gi, T, B)
        logits = torch.empty(T * B, self.nclass, device=dev)
        L.linear_f64(rec, 512, P["rnn.1.embedding.weight"], P["rnn.1.embedding.bias"], logits, self.nclass, T * B, self.nclass, 512)
        return logits.view(T, B, self.nclass)


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
