"""SSIM / MS-SSIM scores on the HIP path.
Interface of /root/reference/src/nind_denoise/common/libs/pt_losses.py:6-18: ``SSIM_loss()(input, target)`` and
``MS_SSIM_loss()(input, target)`` return ``1 - piqa.SSIM / piqa.MS_SSIM`` per sample (``reduction=None``: shape [N]).
The arithmetic is piqa's published algorithm with piqa's defaults (piqa itself is not installed: parity unpinned, see
oracle/losses.py); it runs in ``libnind_hip.so`` (csrc/ssim.hip).  Differentiable with respect to ``input`` (the generated
batch), like the reference's classes when they are used as training criterions (nn_common.py:170-177): the backward pass is
``nd_ssim_loss_grad``.  ``target`` is treated as a constant."""
import torch

from ... import _lib


def _prep(input, target):
    if input.shape != target.shape or input.dim() != 4:
        raise ValueError(f"expected two [N,C,H,W] tensors of one shape, got {tuple(input.shape)} and {tuple(target.shape)}")
    if input.device.type != "cuda" or target.device != input.device:
        raise RuntimeError("SSIM / MS-SSIM run on the GPU only (no CPU fallback): move both images to the device")
    return input.detach().to(torch.float32).contiguous(), target.detach().to(torch.float32).contiguous()


def _score(fn_name, input, target):
    x, y = _prep(input, target)
    n, c, h, w = x.shape
    lib = _lib.load()
    wsb = lib.nd_ssim_workspace_bytes(n, c, h, w)
    ws = torch.empty(max(wsb, 4096), dtype=torch.uint8, device=x.device)
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(getattr(lib, fn_name)(x.data_ptr(), y.data_ptr(), n, c, h, w, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                         _lib.stream_ptr(x.device)), fn_name)
    return out


def mse(input, target):
    """F.mse_loss(input, target) (mean over every element) as a 0-dim tensor on the device."""
    x, y = input.detach().to(torch.float32).contiguous(), target.detach().to(torch.float32).contiguous()
    if x.shape != y.shape:
        raise ValueError(f"shapes differ: {tuple(x.shape)} vs {tuple(y.shape)}")
    if x.device.type != "cuda" or y.device != x.device:
        raise RuntimeError("mse runs on the GPU only (no CPU fallback)")
    ws = torch.empty(4096, dtype=torch.uint8, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().nd_mse(x.data_ptr(), y.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr(x.device)), "nd_mse")
    return out[0]


class _LossFn(torch.autograd.Function):
    """1 - score per sample; backward through nd_ssim_loss_grad (one call per sample weight vector)."""

    @staticmethod
    def forward(ctx, input, target, multiscale):
        ctx.multiscale = multiscale
        ctx.save_for_backward(input, target)
        return 1 - _score("nd_ms_ssim" if multiscale else "nd_ssim", input, target)

    @staticmethod
    def backward(ctx, grad_out):
        input, target = ctx.saved_tensors
        x, y = _prep(input, target)
        n, c, h, w = x.shape
        lib = _lib.load()
        wsb = lib.nd_ssim_loss_workspace_bytes(1, c, h, w)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        gx = torch.empty_like(x)
        scratch = torch.zeros(1, dtype=torch.float32, device=x.device)
        go = grad_out.detach().to(torch.float32).cpu()
        with torch.cuda.device(x.device):
            for i in range(n):      # d(1 - score_i)/dx_i scaled by the incoming gradient of sample i
                _lib.check(lib.nd_ssim_loss_grad(x[i:i + 1].data_ptr(), y[i:i + 1].data_ptr(), 1, c, h, w,
                                                 1 if ctx.multiscale else 0, float(go[i]), scratch.data_ptr(),
                                                 gx[i:i + 1].data_ptr(), 0, ws.data_ptr(), wsb, _lib.stream_ptr(x.device)),
                           "nd_ssim_loss_grad")
        return gx.to(input.dtype), None, None


class SSIM_loss(torch.nn.Module):
    def forward(self, input, target):
        return _LossFn.apply(input, target, False)


class MS_SSIM_loss(torch.nn.Module):
    def forward(self, input, target):
        return _LossFn.apply(input, target, True)
