"""The reference's evaluation pass on the HIP kernels.  Reference: TextSR.model_inference
(interfaces/super_resolution_eng.py:84-103) and TextSR.eval (:375-552) for `test_model == "crnn"`: student prior on the LR
image -> SR -> recogniser on LR / HR / SR -> greedy CTC strings -> str_filt('lower') accuracy, plus PSNR / SSIM of SR and of the
bicubic-upsampled LR against HR, averaged per batch exactly as the reference does (mean of per-batch values)."""
import time

import torch

from .lib import lib
from .model.crnn import text_prior
from .utils.metrics import get_string_crnn
from .utils.ssim_psnr import SSIM, calculate_psnr
from .utils.util import str_filt


def bicubic_resize(x, size):
    """F.interpolate(x, size, mode="bicubic") for NCHW float tensors (super_resolution_eng.py:486)."""
    if not x.is_cuda:
        raise RuntimeError("tatt_amd bicubic_resize runs on the GPU only")
    x = x.detach().contiguous().float()
    B, C, H, W = x.shape
    y = torch.empty(B, C, size[0], size[1], device=x.device)
    lib().bicubic_resize(x, y, B * C, H, W, size[0], size[1])
    return y


def model_inference(images_lr, model, stu_model):
    """TextSR.model_inference (:84-103): {'images_sr', 'duration'}; `stu_model` = the student CRNN, `model` = the SR net."""
    ret = {"duration": 0.0}
    before = time.time()
    logits = stu_model.forward_from_images(images_lr)           # parse_crnn_data + CRNN in one go
    prior = text_prior(logits)
    ret["duration"] += time.time() - before
    before = time.time()
    ret["images_sr"] = model(images_lr, prior.detach())
    ret["duration"] += time.time() - before
    return ret


@torch.no_grad()
def evaluate(model, stu_model, test_rec_model, val_loader, stu_iter=1, filter_mode="lower", is_crnn_voc=True, voc_root="."):
    """TextSR.eval (:375-552) with the CRNN recogniser: returns the reference's metric_dict (per-batch 'psnr', 'ssim', 'psnr_lr',
    'ssim_lr' lists; 'accuracy', 'psnr_avg', 'ssim_avg') plus 'accuracy_lr' / 'accuracy_hr' (which the reference prints) and the
    decoded strings.  `val_loader` yields (images_hr, images_lr, label_strs).  The reference's `str_filt(…, 'lower')` opens the
    40-character vocabulary file of CDistNet; `is_crnn_voc=True` uses the identical built-in digits+lowercase set instead
    (pass False and `voc_root` to read the file as the reference does)."""
    model.eval(), stu_model.eval(), test_rec_model.eval()
    cal_ssim = SSIM()
    md = {"psnr_lr": [], "ssim_lr": [], "psnr": [], "ssim": [], "accuracy": 0.0, "psnr_avg": 0.0, "ssim_avg": 0.0,
          "strings_sr": [], "strings_lr": [], "strings_hr": []}
    counters = {i: 0 for i in range(stu_iter)}
    n_lr = n_hr = n_img = 0
    filt = lambda s: str_filt(s, filter_mode, is_crnn_voc, voc_root)
    for images_hr, images_lr, label_strs in val_loader:
        images_lr, images_hr = images_lr.cuda(), images_hr.cuda()
        images_sr = model_inference(images_lr, model, stu_model)["images_sr"]
        pred = {}
        for tag, img in (("lr", images_lr), ("hr", images_hr), ("sr", images_sr)):
            pred[tag] = get_string_crnn(test_rec_model.forward_from_images(img[:, :3]), False)
            md["strings_" + tag] += pred[tag]
        img_lr = bicubic_resize(images_lr, images_sr.shape[-2:])
        md["psnr"].append(calculate_psnr(images_sr[:, :3], images_hr[:, :3]))
        md["ssim"].append(cal_ssim(images_sr[:, :3], images_hr[:, :3]))
        md["psnr_lr"].append(calculate_psnr(img_lr[:, :3], images_hr[:, :3]))
        md["ssim_lr"].append(cal_ssim(img_lr[:, :3], images_hr[:, :3]))
        for b in range(images_lr.shape[0]):
            label = filt(label_strs[b])
            for k in range(stu_iter):
                counters[k] += filt(pred["sr"][b]) == label
            n_lr += filt(pred["lr"][b]) == label
            n_hr += filt(pred["hr"][b]) == label
        n_img += images_lr.shape[0]
    psnr_avg = sum(md["psnr"]) / (len(md["psnr"]) + 1e-10)
    ssim_avg = sum(md["ssim"]) / (len(md["psnr"]) + 1e-10)
    md["accuracy"] = round(counters[stu_iter - 1] / n_img, 4)
    md["accuracy_lr"], md["accuracy_hr"] = round(n_lr / n_img, 4), round(n_hr / n_img, 4)
    md["psnr_avg"], md["ssim_avg"] = round(psnr_avg.item(), 6), round(ssim_avg.item(), 6)
    md["psnr_lr_avg"] = round((sum(md["psnr_lr"]) / (len(md["psnr_lr"]) + 1e-10)).item(), 6)
    md["ssim_lr_avg"] = round((sum(md["ssim_lr"]) / (len(md["ssim_lr"]) + 1e-10)).item(), 6)
    eng = getattr(model, "_engine", None)
    if eng is not None:
        eng.check_rpe_chain()          # the metrics above already synchronised; a timed-out RPE chain invalidates them
    return md
