"""String helpers of the evaluation loop.  Reference: utils/util.py:14-42 (`str_filt`).  Host-side text processing: there
is nothing here for the GPU to do."""
import os
import string

_VOC_FILES = {                                   # vocabulary files the reference opens relative to its checkout root
    "korean": "CDistNet/cdistnet/utils/dict_2354_kor.txt",
    "lower": "CDistNet/cdistnet/utils/dict_40_num_eng.txt",
    "all": "CDistNet/cdistnet/utils/dict_2448_num_eng_Eng_spe_kor.txt",
    "all_test": "CDistNet/cdistnet/utils/dict_2390_num_eng_kor.txt",
}


def str_filt(str_, voc_type, is_crnn=False, voc_root="."):
    """utils/util.py:14-42: lower-case `str_` and drop every character that is not in the vocabulary of `voc_type`
    ('digit', 'lower', 'upper' built in; 'korean', 'all', 'all_test' and non-CRNN 'lower' read the reference's vocabulary
    files, found under `voc_root` = the reference checkout, default the working directory as in the reference)."""
    alpha_dict = {
        "digit": string.digits,
        "lower": string.digits + string.ascii_lowercase,
        "upper": string.digits + string.ascii_letters,
    }
    if voc_type in ("korean", "all", "all_test") or (voc_type == "lower" and not is_crnn):
        with open(os.path.join(voc_root, _VOC_FILES[voc_type]), "r") as f:
            alpha_dict[voc_type] = [x.replace("\n", "") for x in f.readlines()]
    str_ = str_.lower()
    for char in str_:
        if char not in alpha_dict[voc_type]:
            str_ = str_.replace(char, "")
    return str_
