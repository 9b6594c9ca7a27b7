"""tatt_tri_ssim_fwd_static: the allocation-free variant of tatt_tri_ssim_fwd (its per-plane partial sums live in a static device
array of the library instead of a caller-provided workspace).  Through the C-ABI, bit-exact against the existing entry point, on two
calls in a row -- the second with a smaller input, so stale partial sums of the first call sit in the array -- and refused beyond
the array's stated maximum."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_tri_ssim_fwd_static_matches_workspace_variant():
    from tatt_amd.lib import lib
    L = lib()
    g = torch.Generator().manual_seed(17)
    for B, C, H, W in [(64, 4, 32, 128), (3, 4, 32, 128), (5, 3, 16, 64)]:          # second and third calls are smaller than the first
        x1, x2, x3 = (torch.rand(B, C, H, W, generator=g).cuda() for _ in range(3))
        ref, got = torch.full((1,), -1.0, device="cuda"), torch.full((1,), -2.0, device="cuda")
        ws = torch.empty(L.ssim_workspace(B, C), device="cuda")
        L.tri_ssim_fwd(x1, x2, x3, ref, B, C, H, W, ws)
        L.tri_ssim_fwd_static(x1, x2, x3, got, B, C, H, W)
        torch.cuda.synchronize()
        assert torch.equal(ref, got), (B, C, H, W, float(ref), float(got))
        assert 0.0 < float(got) < 1.0
    with pytest.raises(RuntimeError, match="exceed the static scratch"):
        L.tri_ssim_fwd_static(x1, x2, x3, got, 4097, 4, 16, 64)                      # 16388 planes > 16384: refused before any launch
    torch.cuda.synchronize()
