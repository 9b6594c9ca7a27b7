"""Checkpoint I/O in the reference's on-disk format.  Reference: TextBase.save_checkpoint
(interfaces/base.py:239-276), TextBase.generator_init (:163-215), TextBase.CRNN_init (:317-349).

The files are plain `torch.save` dicts: `model_<iters>.pth` / `model_best.pth` = {'state_dict_G', 'info': {'arch', 'iters',
'epochs', 'batch_size', 'voc_type', 'up_scale_factor'}, 'best_history_res', 'best_model_info', 'param_num', 'converge'},
`recognizer_<iters>.pth` / `recognizer_best.pth` = the student CRNN's state_dict.  Parameter names and shapes of the tatt_amd
modules equal the reference's, so the files move in both directions unchanged.  Loading uses `weights_only=True`: nothing in
a checkpoint is executed (the reference's "whole pickled model" branch of CRNN_init, :344-346, is refused for that reason)."""
import os
from collections import OrderedDict

import torch

from .loss.image_loss import ImageLoss
from .model.crnn import CRNN
from .model.tsrn import TSRN_TL_TRANS


def text_embedding_for(voc_type="eng", tpg="crnn"):
    """generator_init :170-177."""
    if voc_type == "kor":
        return 2354
    if voc_type == "eng" and tpg == "cdistnet_eng":
        return 40
    if voc_type == "all":
        return 2448
    return 37


def _load(path):
    return torch.load(path, map_location="cpu", weights_only=True)


def generator_init(resume="", scale_factor=2, width=128, height=32, STN=True, mask=True, srb=5, hd_u=32, voc_type="eng",
                   tpg="crnn", gradient=True, device="cuda"):
    """generator_init (:163-215): build the SR net + ImageLoss; `resume` = '' (fresh), a directory (its model_best.pth, loaded
    strict=False as the reference does) or a file (strict).  Returns {'model', 'crit', 'epoch', 'iter'}."""
    model = TSRN_TL_TRANS(scale_factor=scale_factor, width=width, height=height, STN=STN, mask=mask, srb_nums=srb,
                          hidden_units=hd_u, text_embedding=text_embedding_for(voc_type, tpg))
    crit = ImageLoss(gradient=gradient, loss_weight=[1, 1e-4])
    epoch = iters = 0
    if resume != "":
        if os.path.isdir(resume):
            ckpt = _load(os.path.join(resume, "model_best.pth"))
            model.load_state_dict(ckpt["state_dict_G"], strict=False)
        else:
            ckpt = _load(resume)
            model.load_state_dict(ckpt["state_dict_G"])
        epoch, iters = ckpt["info"]["epochs"], ckpt["info"]["iters"]
    return {"model": model.to(device), "crit": crit.to(device), "epoch": epoch, "iter": iters}


def CRNN_init(recognizer_path, device="cuda"):
    """CRNN_init (:317-349): CRNN(32, 1, 37, 256) with the state dict stored at `recognizer_path`."""
    model = CRNN(imgH=32, nc=1, nclass=37, nh=256)
    sd = _load(recognizer_path)
    if not isinstance(sd, (dict, OrderedDict)):
        raise RuntimeError("tatt_amd CRNN_init loads state dicts only (whole pickled modules are not executed)")
    model.load_state_dict(sd)
    return model.to(device)


def save_checkpoint(netG_list, epoch, iters, best_acc_dict, best_model_info, is_best, converge_list, recognizer=None,
                    prefix="acc", global_model=None, *, ckpt_path, batch_size, voc_type="eng", scale_factor=2):
    """save_checkpoint (:239-276), same positional arguments; what the reference reads from `self` is keyword-only here."""
    os.makedirs(ckpt_path, exist_ok=True)
    netG = netG_list[0]
    netG = getattr(netG, "module", netG)
    eng = getattr(netG, "_engine", None)
    if eng is not None:
        eng.check_rpe_chain()          # a checkpoint must not hold weights trained through an RPE chain that reported a timed-out wait
    cpu = lambda sd: OrderedDict((k, v.detach().cpu().clone()) for k, v in sd.items())
    save_dict = {
        "state_dict_G": cpu(netG.state_dict()),
        "info": {"arch": "tatt", "iters": iters, "epochs": epoch, "batch_size": batch_size, "voc_type": voc_type,
                 "up_scale_factor": scale_factor},
        "best_history_res": best_acc_dict,
        "best_model_info": best_model_info,
        "param_num": sum(p.nelement() for p in netG.parameters()),
        "converge": converge_list,
    }
    names = ["model_%s.pth" % iters] + (["model_best.pth"] if is_best else [])
    for n in names:
        torch.save(save_dict, os.path.join(ckpt_path, n))
    if recognizer is not None:
        rec = cpu(recognizer[0].state_dict())
        for n in ["recognizer_%s.pth" % iters] + (["recognizer_best.pth"] if is_best else []):
            torch.save(rec, os.path.join(ckpt_path, n))
    if global_model is not None:
        g = cpu(global_model.state_dict())
        for n in ["global_model_%s.pth" % iters] + (["global_model_best.pth"] if is_best else []):
            torch.save(g, os.path.join(ckpt_path, n))
    return save_dict
