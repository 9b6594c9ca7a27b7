from . import ssim_psnr  # noqa: F401
