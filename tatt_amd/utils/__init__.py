from . import metrics, ssim_psnr, util  # noqa: F401
