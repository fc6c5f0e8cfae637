"""UNet on MI355X: constructor, parameter/buffer names and call contract of the reference module
(/root/reference/src/nind_denoise/networks/ThirdPartyNets.py:62-169), forward executed by libnind_hip.so
(eval mode: BatchNorm2d running statistics are folded into the convolutions when the weights are packed).

The torch layers below are parameter containers only; ``forward`` never calls them.  No CPU path.
"""
import ctypes

import torch
from torch import nn

from .. import _lib


class _Box(nn.Module):
    """Named container: gives the state-dict the reference's nesting (inc.conv.conv.0.weight, ...)."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            self.add_module(k, v)


def _double_conv(in_ch, out_ch):
    return _Box(conv=nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True),
                                   nn.Conv2d(out_ch, out_ch, 3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True)))


class UNet(nn.Module):
    def __init__(self, n_channels=3, n_classes=3, funit=64, find_noise=False):
        super().__init__()
        if int(n_channels) != 3 or int(n_classes) != 3:
            raise NotImplementedError("the HIP UNet path is built for RGB in / RGB out (the reference's defaults)")
        self.inc = _Box(conv=_double_conv(3, 64))
        for n, (ci, co) in enumerate([(64, 128), (128, 256), (256, 512), (512, 512)], start=1):
            self.add_module(f"down{n}", _Box(mpconv=nn.Sequential(nn.MaxPool2d(2), _double_conv(ci, co))))
        for n, (ci, co) in enumerate([(1024, 256), (512, 128), (256, 64), (128, 64)], start=1):
            self.add_module(f"up{n}", _Box(up=nn.ConvTranspose2d(ci // 2, ci // 2, 2, stride=2), conv=_double_conv(ci, co)))
        self.outc = _Box(conv=nn.Conv2d(64, 3, 1))
        self.find_noise = find_noise in (True, "True", "true", "1")
        self._packed = None
        self._workspaces = {}

    def _weights_key(self, device):
        return (str(device),) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def packed_weights(self, device):
        key = self._weights_key(device)
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        lib = _lib.load()
        sd = self.state_dict()
        n = lib.nd_unet_num_tensors()
        host, ptrs = [], (ctypes.c_void_p * n)()
        for i in range(n):
            t = sd[lib.nd_unet_tensor_name(i).decode()].detach().to(device="cpu", dtype=torch.float32).contiguous()
            host.append(t)
            ptrs[i] = t.data_ptr()
        nbytes = lib.nd_unet_packed_bytes(_lib.ND_F32)
        blob = torch.empty(nbytes // 4, dtype=torch.float32)
        _lib.check(lib.nd_unet_pack_weights(_lib.ND_F32, ptrs, n, blob.data_ptr(), nbytes), "nd_unet_pack_weights")
        dev_blob = blob.to(device)
        self._packed = (key, dev_blob)
        return dev_blob

    def workspace(self, h, w, batch, device):
        key = (str(device), h, w, batch)
        ws = self._workspaces.get(key)
        if ws is None:
            lib = _lib.load()
            nbytes = lib.nd_unet_workspace_bytes(h, w, batch, _lib.ND_F32)
            if nbytes == 0:
                _lib.check(lib.nd_unet_workspace_init(None, 0, h, w, batch, _lib.ND_F32, None), "UNet")
            self._workspaces.clear()
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            _lib.check(lib.nd_unet_workspace_init(ws.data_ptr(), nbytes, h, w, batch, _lib.ND_F32, _lib.stream_ptr(device)),
                       "nd_unet_workspace_init")
            self._workspaces[key] = ws
        return ws

    def forward(self, x):
        if x.device.type != "cuda":
            raise RuntimeError("nind_denoise_amd.UNet runs on the MI355X HIP path only (no CPU fallback)")
        if self.training:
            raise RuntimeError("nind_denoise_amd.UNet implements eval mode (BatchNorm running statistics) only; call .eval()")
        if x.dim() != 4 or x.size(1) != 3:
            raise ValueError(f"UNet expects [B,3,H,W], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        b, _, h, w = x.shape
        lib = _lib.load()
        with torch.cuda.device(x.device):
            blob = self.packed_weights(x.device)
            ws = self.workspace(h, w, b, x.device)
            y = torch.empty_like(x)
            _lib.check(lib.nd_unet_forward(_lib.ND_F32, blob.data_ptr(), x.data_ptr(), y.data_ptr(), b, h, w, ws.data_ptr(),
                                           ws.numel(), _lib.stream_ptr(x.device)), "nd_unet_forward")
        return x - y if self.find_noise else y
