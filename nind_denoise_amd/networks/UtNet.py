"""UtNet on MI355X: same constructor, parameter names and call contract as the reference module
(/root/reference/src/nind_denoise/networks/UtNet.py:13-109), forward executed by libnind_hip.so.

The torch layers created here are parameter CONTAINERS only (they give the module the reference's
state-dict layout, initialisation and ``load_state_dict`` behaviour); ``forward`` never calls them.
It repacks the weights once into MFMA fragment order, keeps the packed blob and a per-(cs, batch)
activation workspace resident in HBM, and enqueues the whole conv stack on the current stream
through ``nd_utnet_forward``.  There is no CPU path: a CPU tensor raises.
"""
import ctypes

import torch
from torch import nn

from .. import _lib
from ..synth import utnet_layer_table, utnet_prelu_keys

_ACTIVATIONS = {"PReLU": nn.PReLU, "ELU": nn.ELU, "Hardswish": nn.Hardswish}


def valid_cs(cs):
    """The network only accepts cs = 16k + 56 (the reference raises in torch.cat otherwise)."""
    return cs >= 104 and (cs - 56) % 16 == 0


def nearest_valid_cs(cs):
    k = (cs - 56 + 8) // 16          # ties go up: 256 -> 264, 512 -> 520, 128 -> 136
    return max(104, 16 * k + 56)


class _TrainState:
    """Flat parameter / gradient buffers, packed-weight blobs and the training workspace of one module on one device
    (nd_utnet_train_forward / nd_utnet_train_backward of include/nind_hip.h)."""

    def __init__(self, model, device):
        lib = _lib.load()
        self.device = device
        n = lib.nd_utnet_param_count(model.funit)
        if n == 0:
            raise ValueError(f"UtNet: funit={model.funit} is not supported by the HIP training path (multiple of 8)")
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=device)
        self.blobs = torch.empty(lib.nd_utnet_train_blob_bytes(model.funit), dtype=torch.uint8, device=device)
        self.ranges = {}
        for i, name in enumerate(_lib.utnet_tensor_names()):
            off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(lib.nd_utnet_param_range(model.funit, i, off, cnt))
            self.ranges[name] = (off.value, cnt.value)
        self.ws, self.ws_key = None, None
        self.generation = 0          # bumped by every forward: a backward must follow ITS forward

    def workspace(self, model, cs, batch):
        if self.ws_key != (cs, batch):
            lib = _lib.load()
            nbytes = lib.nd_utnet_train_workspace_bytes(model.funit, cs, batch)
            if nbytes == 0:
                _lib.check(lib.nd_utnet_train_workspace_init(None, 0, model.funit, cs, batch, None), "UtNet training")
            self.ws = None
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(lib.nd_utnet_train_workspace_init(self.ws.data_ptr(), nbytes, model.funit, cs, batch, _lib.stream_ptr(self.device)),
                       "nd_utnet_train_workspace_init")
            self.ws_key = (cs, batch)
        return self.ws


class _UtNetFunction(torch.autograd.Function):
    """UtNet.forward under autograd: forward = device-side weight packing + the conv stack with the pre-activations kept
    (nd_utnet_train_forward), backward = activation / bias / weight / data gradients of every layer (nd_utnet_train_backward),
    the kernels of the fused training step (csrc/utnet_train.hip).  Gradients are returned for the parameters; the input image
    gets none (the reference never trains through it)."""

    @staticmethod
    def forward(ctx, model, names, x, *params):
        st = model._train_state(x.device)
        with torch.no_grad():
            for n, p in zip(names, params):
                off, cnt = st.ranges[n]
                st.flat[off:off + cnt].copy_(p.detach().reshape(-1))
        batch, cs = x.size(0), x.size(2)
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            ws = st.workspace(model, cs, batch)
            _lib.check(_lib.load().nd_utnet_train_forward(model.funit, _lib.ACT[model.activation], model.flags, st.flat.data_ptr(),
                                                          st.blobs.data_ptr(), x.data_ptr(), y.data_ptr(), batch, cs, ws.data_ptr(),
                                                          ws.numel(), _lib.stream_ptr(x.device)), "nd_utnet_train_forward")
        st.generation += 1
        ctx.model, ctx.names, ctx.geom, ctx.generation = model, names, (batch, cs), st.generation
        ctx.shapes = [p.shape for p in params]
        return y

    @staticmethod
    def backward(ctx, gy):
        model = ctx.model
        st = model._train_state(gy.device)
        if st.generation != ctx.generation:
            raise RuntimeError("UtNet backward: another forward of this module ran under autograd since this graph was built; its "
                               "activations are gone (one forward/backward in flight per module -- as the reference's training loop runs)")
        batch, cs = ctx.geom
        gy = gy.to(torch.float32).contiguous()
        with torch.cuda.device(gy.device):
            _lib.check(_lib.load().nd_utnet_train_backward(model.funit, _lib.ACT[model.activation], model.flags, st.flat.data_ptr(),
                                                           st.grads.data_ptr(), st.blobs.data_ptr(), gy.data_ptr(), batch, cs,
                                                           st.ws.data_ptr(), st.ws.numel(), _lib.stream_ptr(gy.device), None, 0),
                       "nd_utnet_train_backward")
        grads = []
        for n, shape in zip(ctx.names, ctx.shapes):
            off, cnt = st.ranges[n]
            grads.append(st.grads[off:off + cnt].view(shape).clone())
        return (None, None, None) + tuple(grads)


class UtNet(nn.Module):
    # per-call arithmetic flags (nd_flags of include/nind_hip.h), overridable per instance:
    #   split_k = False  -> every output tile whole: a tile's bits do not depend on the batch grouping
    #   winograd = False -> direct convolution on every 3x3 layer
    #   w1d_regs = True  -> A/B switch: the fused 1-D Winograd layers through conv_w1d (transform in registers) instead of conv_w2d
    #   useful_only = False -> the fused denoise loop computes whole tiles in every layer (as forward() always does) instead of
    #                          only what the useful centre of a tile depends on in the last decoder levels (same canvas)
    #   fused_pool = False -> A/B switch: every MaxPool2d(2) as its own kernel instead of from the producing layer's epilogue (same values)
    split_k = True
    winograd = True
    w1d_regs = False
    useful_only = True
    fused_pool = True

    def __init__(self, funit=64, activation='PReLU'):
        super().__init__()
        funit = int(funit)
        if activation not in _ACTIVATIONS:
            exit(f'UtNet: unknown activation function: {activation}')
        self.funit, self.activation = funit, activation
        groups = {}
        for key, kind, cin, cout, k in utnet_layer_table(funit):
            if kind == "conv":
                layer = nn.Conv2d(cin, cout, k)
            elif kind == "convT":
                layer = nn.ConvTranspose2d(cin, cout, k)
            else:
                layer = nn.ConvTranspose2d(cin, cout, k, stride=2)
            if "." in key:
                seq, idx = key.split(".")
                groups.setdefault(seq, {})[int(idx)] = layer
            else:
                groups[key] = layer
        for key in utnet_prelu_keys():
            seq, idx = key.split(".")
            groups[seq][int(idx)] = _ACTIVATIONS[activation]()
        # registration order = the reference's state-dict order (convs1..4, bottom, up1, tconvs1, up2, ...)
        order = ["convs1", "convs2", "convs3", "convs4", "bottom"]
        for n in range(1, 5):
            order += [f"up{n}", f"tconvs{n}"]
        for name in order:
            g = groups[name]
            self.add_module(name, nn.Sequential(*[g[i] for i in sorted(g)]) if isinstance(g, dict) else g)
        self._packed = {}         # dtype -> (key, device blob)
        self.pack_on_device = True   # fp32: build the packed blob in HBM (nd_utnet_pack_weights_device); False: host packer
        self._workspaces = {}     # (device, h, w, batch, dtype) -> uint8 tensor
        self.max_cached_workspaces = 2
        self.compute_dtype = "f32"   # storage of activations + weights inside the conv stack: "f32" | "bf16" | "f16"
        self.weights_generation = 0   # bumped by whoever rewrites the parameters through raw pointers (train.UtNetTrainer)

    def set_compute_dtype(self, name):
        """"f32": fp32 storage, exact-fp32 MFMA (the reference's arithmetic).  "bf16" / "f16": 16-bit storage of
        activations and weights with fp32 accumulation (BASELINE configs 3 / 4); inputs and outputs stay float32."""
        if name not in ("f32", "bf16", "f16"):
            raise ValueError(f"unknown compute dtype {name!r}")
        self.compute_dtype = name
        return self

    @property
    def _dt(self):
        return _lib.DTYPE[self.compute_dtype]

    @property
    def flags(self):
        return ((0 if self.split_k else _lib.FLAG_NO_SPLITK) | (0 if self.winograd else _lib.FLAG_DIRECT_CONV) |
                (_lib.FLAG_W1D_REGS if self.w1d_regs else 0) | (0 if self.useful_only else _lib.FLAG_FULL_TILES) |
                (0 if self.fused_pool else _lib.FLAG_UNFUSED_POOL))

    # ------------------------------------------------------------------ weights
    def _weights_key(self, device):
        return (str(device), self.weights_generation) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed_weights(self, device):
        """Packed blob in HBM (re-packed when a parameter changed or moved)."""
        key = self._weights_key(device)
        hit = self._packed.get(self.compute_dtype)
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = _lib.load()
        sd = self.state_dict()
        names = _lib.utnet_tensor_names()
        nbytes = lib.nd_utnet_packed_bytes(self.funit, self._dt)
        if nbytes == 0:
            raise ValueError(f"UtNet: funit={self.funit} is not supported by the HIP path for {self.compute_dtype} "
                             "(multiple of 8 for f32, of 16 for bf16 / f16)")
        on_device = self.compute_dtype == "f32" and self.pack_on_device
        where = device if on_device else "cpu"     # fp32: pack in HBM (device-side packers); 16-bit: on the host
        keep = []
        ptrs = (ctypes.c_void_p * len(names))()
        for i, n in enumerate(names):
            if n in sd:
                t = sd[n].detach().to(device=where, dtype=torch.float32).contiguous()
                keep.append(t)
                ptrs[i] = t.data_ptr()
            else:
                ptrs[i] = None  # activation without parameters (ELU / Hardswish)
        if on_device:
            dev_blob = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
            with torch.cuda.device(device):
                _lib.check(lib.nd_utnet_pack_weights_device(self.funit, self._dt, ptrs, len(names), dev_blob.data_ptr(), nbytes,
                                                            _lib.stream_ptr(device)), "nd_utnet_pack_weights_device")
                torch.cuda.current_stream(device).synchronize()     # `keep` may be freed after this
        else:
            blob = torch.empty(nbytes // 4, dtype=torch.float32)
            _lib.check(lib.nd_utnet_pack_weights(self.funit, self._dt, ptrs, len(names), blob.data_ptr(), nbytes),
                       "nd_utnet_pack_weights")
            dev_blob = blob.to(device)
        self._packed[self.compute_dtype] = (key, dev_blob)
        return dev_blob

    def workspace(self, cs, batch, device, width=None):
        """Activation workspace for [batch,3,cs,width or cs] inputs (zero borders initialised once, then cached)."""
        h, w = int(cs), int(cs if width is None else width)
        key = (str(device), h, w, int(batch), self.compute_dtype)
        ws = self._workspaces.get(key)
        if ws is None:
            lib = _lib.load()
            nbytes = lib.nd_utnet_workspace_bytes_hw(self.funit, h, w, batch, self._dt)
            if nbytes == 0:
                _lib.check(lib.nd_utnet_workspace_init_hw(None, 0, self.funit, h, w, batch, self._dt, None), "UtNet")
            while len(self._workspaces) >= self.max_cached_workspaces:
                self._workspaces.pop(next(iter(self._workspaces)))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            _lib.check(lib.nd_utnet_workspace_init_hw(ws.data_ptr(), nbytes, self.funit, h, w, batch, self._dt,
                                                      _lib.stream_ptr(device)), "nd_utnet_workspace_init")
            self._workspaces[key] = ws
        return ws

    # ------------------------------------------------------------------ forward
    def forward(self, l):
        if l.device.type != "cuda":
            raise RuntimeError("nind_denoise_amd.UtNet runs on the MI355X HIP path only (no CPU fallback); "
                               "move the module and its input to the GPU")
        if l.dim() != 4 or l.size(1) != 3:
            raise ValueError(f"UtNet expects [B,3,H,W], got {tuple(l.shape)}")
        batch, h, w = l.size(0), l.size(2), l.size(3)
        for cs in (h, w):
            if not valid_cs(cs):
                raise ValueError(f"UtNet: tile size {cs} is not of the form 16k+56 (e.g. {nearest_valid_cs(cs)}); "
                                 "the reference network fails on it too")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # the differentiable path: the reference trains through plain autograd (nn_common.py:146 `self.model.train()`,
            # :198-218 forward / backward); an eval() module (nn_common.py:142, denoise_image.py:229) takes the inference path
            # below, whose output carries no graph
            if l.requires_grad:
                raise NotImplementedError("UtNet: the gradient with respect to the input image is not produced by the HIP path")
            if h != w:
                raise ValueError(f"UtNet under autograd expects square crops, got {h}x{w}")
            if self.compute_dtype != "f32":
                raise NotImplementedError("UtNet under autograd runs in fp32 (the training step's arithmetic)")
            named = [(n, p) for n, p in self.named_parameters()]
            return _UtNetFunction.apply(self, tuple(n for n, _ in named), l.detach().to(torch.float32).contiguous(),
                                        *[p for _, p in named])
        x = l.detach().to(torch.float32).contiguous()
        lib = _lib.load()
        with torch.cuda.device(x.device):
            blob = self.packed_weights(x.device)
            ws = self.workspace(h, batch, x.device, width=w)
            y = torch.empty_like(x)
            _lib.check(lib.nd_utnet_forward_hw(self.funit, _lib.ACT[self.activation], self._dt, self.flags, blob.data_ptr(),
                                               x.data_ptr(), y.data_ptr(), batch, h, w, ws.data_ptr(), ws.numel(),
                                               _lib.stream_ptr(x.device)), "nd_utnet_forward")
        return y

    def _train_state(self, device):
        st = getattr(self, "_tstate", None)
        if st is None or st.device != device:
            st = self._tstate = _TrainState(self, device)
        return st

    def flops_per_tile(self, cs):
        return _lib.load().nd_utnet_flops(self.funit, cs)
