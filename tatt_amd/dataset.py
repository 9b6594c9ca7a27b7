"""The reference's input transform on the device.  Reference: `resizeNormalize` (dataset/dataset.py:272-325)
and `alignCollate` (:677-693, the collate function of the CRNN-prior configuration): every sample's HR / LR image is resized
with PIL's bicubic resampler to (imgW, imgH) / (imgW, imgH) / down_sample_scale, scaled to [0, 1] and given a fourth mask
channel (1 where the gray value is <= its mean).

Here the collate function only packs the decoded uint8 pixels of the batch into one pinned buffer; the resampling, scaling
and mask run in one kernel launch per batch (`tatt_resize_normalize`), bit-exact with Pillow.  The reference's `aug` hook
(imgaug, host-side) and `ratio_keep` are not part of the TATT training path and are not built."""
import numpy as np
import torch

from .lib import lib


def _as_rgb_array(img):
    a = np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise RuntimeError(f"expected uint8 RGB images [h, w, 3], got {a.dtype} {a.shape}")
    return np.ascontiguousarray(a)


def pack_images(images):
    """List of PIL images / uint8 [h,w,3] arrays -> (pinned uint8 buffer, offsets int64 [B], heights int32 [B], widths int32 [B])."""
    arrs = [_as_rgb_array(i) for i in images]
    sizes = np.array([a.size for a in arrs], np.int64)
    offs = np.concatenate([[0], np.cumsum((sizes + 15) // 16 * 16)[:-1]]).astype(np.int64)
    buf = torch.empty(int(offs[-1] + sizes[-1]), dtype=torch.uint8)
    if torch.cuda.is_available():
        buf = buf.pin_memory()
    nb = buf.numpy()
    for a, o in zip(arrs, offs):
        nb[o:o + a.size] = a.reshape(-1)
    return buf, torch.from_numpy(offs), torch.tensor([a.shape[0] for a in arrs], dtype=torch.int32), \
        torch.tensor([a.shape[1] for a in arrs], dtype=torch.int32)


class resizeNormalize(object):
    """dataset/dataset.py:272-325, batched: `size` = (W, H) as in the reference; calling it on a LIST of images returns the
    stacked [B, 3 + mask, H, W] float tensor on the GPU; calling it on one image returns that image's [3 + mask, H, W]."""

    def __init__(self, size, mask=False, interpolation=3, aug=None, blur=False, device="cuda"):
        if interpolation != 3:
            raise NotImplementedError("tatt_amd resizeNormalize: Image.BICUBIC (3) only, the reference's default")
        if aug is not None:
            raise NotImplementedError("tatt_amd resizeNormalize: the imgaug hook is host-side and outside the built path")
        self.size, self.mask, self.device = size, mask, device        # `blur` is a no-op in the reference too (:300-309)

    def __call__(self, imgs, ratio_keep=False):
        if ratio_keep:
            raise NotImplementedError("tatt_amd resizeNormalize: ratio_keep is not used by the TATT trainer and is not built")
        single = not isinstance(imgs, (list, tuple))
        buf, offs, hs, ws = pack_images([imgs] if single else imgs)
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("tatt_amd resizeNormalize runs on the GPU only (no CPU fallback)")
        B, (W, H) = len(hs), self.size
        out = torch.empty(B, 4 if self.mask else 3, H, W, device=dev)
        lib().resize_normalize(buf.to(dev, non_blocking=True), offs.to(dev), hs.to(dev), ws.to(dev), out, B, H, W, int(self.mask),
                               int(hs.max()), int(ws.max()))
        return out[0] if single else out


class alignCollate(object):
    """dataset/dataset.py:677-693 (with alignCollate_syn.__init__ :495-533): batch of (HR image, LR image, label) ->
    (images_HR [B,C,imgH,imgW], images_lr [B,C,imgH/s,imgW/s], label_strs), both tensors on the GPU."""

    def __init__(self, imgH=64, imgW=256, down_sample_scale=4, keep_ratio=False, min_ratio=1, mask=False, alphabet=53, train=True,
                 y_domain=False, device="cuda"):
        self.imgH, self.imgW, self.down_sample_scale, self.mask, self.train = imgH, imgW, down_sample_scale, mask, train
        self.transform = resizeNormalize((imgW, imgH), mask, device=device)
        self.transform2 = resizeNormalize((imgW // down_sample_scale, imgH // down_sample_scale), mask, blur=True, device=device)

    def __call__(self, batch):
        images_HR, images_lr, label_strs = zip(*batch)
        return self.transform(list(images_HR)), self.transform2(list(images_lr)), label_strs
