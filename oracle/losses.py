"""CPU restatement of the image-quality scores of the eval harness (oracle: test infrastructure only).

The reference scores denoised images with ``pt_helpers.get_losses``
(/root/reference/src/nind_denoise/common/libs/pt_helpers.py:40-48): ``mse`` = ``F.mse_loss``, ``ssim`` =
``1 - piqa.SSIM(reduction=None)``, ``msssim`` = ``1 - piqa.MS_SSIM(reduction=None)``
(/root/reference/src/nind_denoise/common/libs/pt_losses.py:6-18), and trains with the same two classes
(/root/reference/src/nind_denoise/nn_common.py:170-177).

PARITY UNPINNED: the SSIM / MS-SSIM arithmetic lives in the third-party package **piqa** (pinned ``piqa~=1.3.2`` in the
reference's pyproject.toml:35), which is not vendored in the reference and not installed here, and the reference holds no
fixture or test of these scores.  What follows restates piqa 1.3's published algorithm (``piqa/ssim.py``: functions
``ssim`` and ``ms_ssim`` with the defaults the reference uses -- it passes no arguments):

  * window: 1-D Gaussian, 11 taps, sigma 1.5, normalised to sum 1, applied separably per channel, VALID (no padding);
  * value_range 1, k1 = 0.01, k2 = 0.03, c1 = k1^2, c2 = k2^2;
  * mu = G*x, sigma_xx = G*(x^2) - mu_x^2 (same for yy, xy);
    cs = (2 sigma_xy + c2) / (sigma_xx + sigma_yy + c2);  ss = (2 mu_x mu_y + c1) / (mu_x^2 + mu_y^2 + c1) * cs;
  * SSIM  = mean of ss over (C, H', W') per sample;
  * MS-SSIM: 5 scales, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333); between scales ``avg_pool2d(kernel 2,
    ceil_mode=True)``; per (sample, channel): prod_i relu(cs_i)^w_i for the first four scales times relu(ss_5)^w_5;
    mean over channels.  Needs min(H, W) >= 161 (the fifth scale must still hold an 11-tap window; the reference's own
    probe reports 162 for square inputs, pt_losses.py:20-28).

Properties that do not depend on piqa pin the restatement in tests/: SSIM(x, x) = MS-SSIM(x, x) = 1, symmetry, the closed
form for constant images, and the 162-pixel limit above.
"""
import torch
import torch.nn.functional as F

WINDOW, SIGMA = 11, 1.5
K1, K2 = 0.01, 0.03
MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gaussian_window(size=WINDOW, sigma=SIGMA, dtype=torch.float32):
    k = torch.arange(size, dtype=dtype) - (size - 1) / 2
    k = torch.exp(-(k ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def _filter(x, g):
    """separable valid Gaussian filter, per channel.  x: [N,C,H,W]"""
    c = x.size(1)
    kh = g.view(1, 1, -1, 1).repeat(c, 1, 1, 1)
    kw = g.view(1, 1, 1, -1).repeat(c, 1, 1, 1)
    return F.conv2d(F.conv2d(x, kh, groups=c), kw, groups=c)


def ssim_maps(x, y, value_range=1.0):
    g = gaussian_window(dtype=x.dtype)
    c1, c2 = (K1 * value_range) ** 2, (K2 * value_range) ** 2
    mu_x, mu_y = _filter(x, g), _filter(y, g)
    mu_xx, mu_yy, mu_xy = mu_x ** 2, mu_y ** 2, mu_x * mu_y
    s_xx = _filter(x * x, g) - mu_xx
    s_yy = _filter(y * y, g) - mu_yy
    s_xy = _filter(x * y, g) - mu_xy
    cs = (2 * s_xy + c2) / (s_xx + s_yy + c2)
    ss = (2 * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs
    return ss, cs


def ssim(x, y):
    """piqa.SSIM(reduction=None)(x, y): [N]"""
    if min(x.shape[-2:]) < WINDOW:
        raise RuntimeError("SSIM: the image is smaller than the 11-tap window")
    ss, _ = ssim_maps(x, y)
    return ss.flatten(1).mean(-1)


def ms_ssim(x, y):
    """piqa.MS_SSIM(reduction=None)(x, y): [N]"""
    w = torch.tensor(MS_WEIGHTS, dtype=x.dtype)
    vals = []
    for i in range(len(MS_WEIGHTS)):
        if i > 0:
            x = F.avg_pool2d(x, kernel_size=2, ceil_mode=True)
            y = F.avg_pool2d(y, kernel_size=2, ceil_mode=True)
        if min(x.shape[-2:]) < WINDOW:
            raise RuntimeError("MS-SSIM: the image is too small for five scales of an 11-tap window (needs >= 161 pixels)")
        ss, cs = ssim_maps(x, y)
        ss, cs = ss.flatten(2).mean(-1), cs.flatten(2).mean(-1)     # [N, C]
        vals.append(torch.relu(cs) if i + 1 < len(MS_WEIGHTS) else torch.relu(ss))
    ms = torch.stack(vals, dim=-1) ** w
    return ms.prod(dim=-1).mean(dim=-1)


def get_losses(img1, img2):
    """pt_helpers.get_losses on two [1,3,H,W] tensors (the file reading is the caller's)."""
    assert img1.shape == img2.shape, f'{img1.shape=}, {img2.shape=}'
    return {"mse": F.mse_loss(img1, img2).item(), "ssim": (1 - ssim(img1, img2)).item(),
            "msssim": (1 - ms_ssim(img1, img2)).item()}
