"""Recogniser-output decoding on the device.  Reference: utils/metrics.py:128-146 (`get_string_crnn`)."""
import torch

from ..lib import lib


def ctc_greedy(outputs_):
    """Device part of get_string_crnn: logits [T,B,K] -> (arg [B,T] per-step argmax, codes [B,T], lens [B]) int32 tensors;
    codes[b, :lens[b]] are the class indices left after merging runs and dropping the blank (class 0)."""
    if not outputs_.is_cuda:
        raise RuntimeError("tatt_amd get_string_crnn runs on the GPU only (no CPU fallback)")
    x = outputs_.detach().contiguous().float()
    T, B, K = x.shape
    arg = torch.empty(B, T, dtype=torch.int32, device=x.device)
    codes = torch.empty(B, T, dtype=torch.int32, device=x.device)
    lens = torch.empty(B, dtype=torch.int32, device=x.device)
    lib().ctc_greedy(x, arg, codes, lens, T, B, K)
    return arg, codes, lens


def get_string_crnn(outputs_, use_chinese=False, alphabet='-0123456789abcdefghijklmnopqrstuvwxyz'):
    """utils/metrics.py:128-146: greedy CTC decode of [T,B,K] recogniser logits into B strings (same arguments; `use_chinese`
    is unused there too).  Argmax and the run/blank collapse happen in one kernel; one [B,T] int copy comes back."""
    arg, codes, lens = ctc_greedy(outputs_)
    if len(set(alphabet)) == len(alphabet) and outputs_.shape[2] <= len(alphabet):
        codes, lens = codes.cpu().tolist(), lens.cpu().tolist()
        return ["".join(alphabet[i] for i in c[:n]) for c, n in zip(codes, lens)]
    # an alphabet with repeated characters: the reference merges runs by CHARACTER, so collapse on the host from the argmax
    predict_result = []
    for max_index in arg.cpu().tolist():
        out_str, last = "", ""
        for i in max_index:
            if alphabet[i] != last:
                if i != 0:
                    out_str += alphabet[i]
                    last = alphabet[i]
                else:
                    last = ""
        predict_result.append(out_str)
    return predict_result
