"""Synthetic weights and frames (there is no network: no pretrained ``generator_650.pt``, no NIND images).

The state-dicts produced here have exactly the key names / shapes of the reference networks
(UtNet: /root/reference/src/nind_denoise/networks/UtNet.py:27-88, UNet: ThirdPartyNets.py:62-169),
so they load into the reference modules with ``load_state_dict`` (that is how tests/golden/make_golden.py
pins the oracle) and into this package's modules alike.  Values are drawn per tensor from a generator
seeded by (seed, key), so the result does not depend on module construction order.
"""
import hashlib
import zlib

import numpy as np
import torch


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def _uniform(shape, bound, seed, key):
    return (torch.rand(shape, generator=_gen(seed, key), dtype=torch.float32) * 2 - 1) * bound


def utnet_layer_table(funit=64):
    """(key, kind, cin, cout, k) for every weighted UtNet layer, in forward order."""
    f = funit
    t = []
    for n, (ci, co) in enumerate([(3, f), (f, 2 * f), (2 * f, 4 * f), (4 * f, 8 * f)], start=1):
        t.append((f"convs{n}.0", "conv", ci, co, 3))
        t.append((f"convs{n}.2", "conv", co, co, 3))
    t.append(("bottom.0", "conv", 8 * f, 16 * f, 3))
    t.append(("bottom.2", "convT", 16 * f, 16 * f, 3))
    c = 16 * f
    for n in range(1, 5):
        t.append((f"up{n}", "up", c, c // 2, 2))
        t.append((f"tconvs{n}.0", "convT", c, c // 2, 3))
        t.append((f"tconvs{n}.2", "convT", c // 2, c // 2, 3))
        c //= 2
    t.append(("tconvs4.4", "conv", f, 3, 1))
    return t


def utnet_prelu_keys():
    keys = []
    for n in range(1, 5):
        keys += [f"convs{n}.1", f"convs{n}.3"]
    keys += ["bottom.1", "bottom.3"]
    for n in range(1, 5):
        keys += [f"tconvs{n}.1", f"tconvs{n}.3"]
    return keys


def make_utnet_state_dict(funit=64, seed=123, activation="PReLU", gain=1.0):
    """Random UtNet weights: U(+-gain/sqrt(cin*k*k)) like torch's default conv init; PReLU slopes U(0.05,0.4)
    (the default 0.25 everywhere would hide slope-plumbing bugs)."""
    sd = {}
    for key, kind, ci, co, k in utnet_layer_table(funit):
        bound = gain / float(np.sqrt(ci * k * k))
        shape = (co, ci, k, k) if kind == "conv" else (ci, co, k, k)
        sd[key + ".weight"] = _uniform(shape, bound, seed, key + ".weight")
        sd[key + ".bias"] = _uniform((co,), bound, seed, key + ".bias")
    if activation == "PReLU":
        for key in utnet_prelu_keys():
            sd[key + ".weight"] = 0.225 + _uniform((1,), 0.175, seed, key + ".weight")
    return sd


def make_unet_state_dict(seed=0):
    """Random weights for the reference UNet(3,3) with non-trivial BatchNorm running stats."""
    sd = {}

    def dconv(p, ci, co):
        for k, (a, b) in zip((0, 3), ((ci, co), (co, co))):
            bound = 1.0 / float(np.sqrt(a * 9))
            sd[f"{p}.{k}.weight"] = _uniform((b, a, 3, 3), bound * 1.7, seed, f"{p}.{k}.weight")
            sd[f"{p}.{k}.bias"] = _uniform((b,), bound, seed, f"{p}.{k}.bias")
            sd[f"{p}.{k + 1}.weight"] = 1.0 + _uniform((b,), 0.3, seed, f"{p}.{k + 1}.weight")
            sd[f"{p}.{k + 1}.bias"] = _uniform((b,), 0.1, seed, f"{p}.{k + 1}.bias")
            sd[f"{p}.{k + 1}.running_mean"] = _uniform((b,), 0.1, seed, f"{p}.{k + 1}.running_mean")
            sd[f"{p}.{k + 1}.running_var"] = 0.3 + _uniform((b,), 0.2, seed, f"{p}.{k + 1}.running_var").abs()
            sd[f"{p}.{k + 1}.num_batches_tracked"] = torch.tensor(7, dtype=torch.long)

    dconv("inc.conv.conv", 3, 64)
    for n, (ci, co) in enumerate([(64, 128), (128, 256), (256, 512), (512, 512)], start=1):
        dconv(f"down{n}.mpconv.1.conv", ci, co)
    for n, (ci, co) in enumerate([(1024, 256), (512, 128), (256, 64), (128, 64)], start=1):
        h = ci // 2
        bound = 1.0 / float(np.sqrt(h * 4))
        sd[f"up{n}.up.weight"] = _uniform((h, h, 2, 2), bound, seed, f"up{n}.up.weight")
        sd[f"up{n}.up.bias"] = _uniform((h,), bound, seed, f"up{n}.up.bias")
        dconv(f"up{n}.conv.conv", ci, co)
    sd["outc.conv.weight"] = _uniform((3, 64, 1, 1), 0.125, seed, "outc.conv.weight")
    sd["outc.conv.bias"] = _uniform((3,), 0.125, seed, "outc.conv.bias")
    return sd


def state_dict_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def make_frame(width, height, seed=24, gradient=True):
    """float32 CHW frame in [0,1): default_rng(seed) noise, optionally on a smooth ramp so that
    mirrored borders are distinguishable from shifted copies."""
    rng = np.random.default_rng(seed)
    img = rng.random((3, height, width), dtype=np.float32)
    if gradient:
        yy = np.linspace(0, 1, height, dtype=np.float32)[None, :, None]
        xx = np.linspace(0, 1, width, dtype=np.float32)[None, None, :]
        img = (np.float32(0.5) * img + np.float32(0.25) * yy + np.float32(0.25) * xx).astype(np.float32)
    return img
