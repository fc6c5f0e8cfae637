"""torch.nn.functional (CPU, fp32) restatement of the reference networks (oracle: test infrastructure only).

``utnet_forward`` follows ``UtNet.forward`` (/root/reference/src/nind_denoise/networks/UtNet.py:97-109)
with the layer definitions of UtNet.py:27-88; ``unet_forward`` follows ``UNet.forward``
(/root/reference/src/nind_denoise/networks/ThirdPartyNets.py:138-169) with its
building blocks (:62-135).  Both are driven by a plain ``state_dict`` (name -> tensor)
with the reference's key names, so they consume the reference's ``.pt`` files as is.

The arithmetic (conv / conv-transpose / max-pool / PReLU) lives in PyTorch's CPU
backend exactly as it does for the reference; the reference's own tests do not pin
it, the fixtures in tests/golden/ do.
"""
import torch
import torch.nn.functional as F

UTNET_ENC = ("convs1", "convs2", "convs3", "convs4")
UTNET_DEC = ("tconvs1", "tconvs2", "tconvs3", "tconvs4")


def _act(sd, key, x, activation):
    if activation == "PReLU":
        return F.prelu(x, sd[key + ".weight"])
    if activation == "ELU":
        return F.elu(x)
    if activation == "Hardswish":
        return F.hardswish(x)
    raise ValueError(f"UtNet: unknown activation function: {activation}")


def utnet_valid_cs(cs):
    """UtNet only accepts cs = 16k + 56 (every pooled size must be even; SURVEY.md headline fact 3)."""
    return cs >= 104 and (cs - 56) % 16 == 0


def utnet_forward(sd, x, activation="PReLU", taps=None):
    """x: [B,3,S,S] fp32 CPU tensor.  ``taps`` (dict) optionally receives every intermediate."""
    def rec(name, t):
        if taps is not None:
            taps[name] = t
        return t

    def enc(name, t):
        t = F.conv2d(t, sd[f"{name}.0.weight"], sd[f"{name}.0.bias"])
        t = rec(f"{name}.1", _act(sd, f"{name}.1", t, activation))
        t = F.conv2d(t, sd[f"{name}.2.weight"], sd[f"{name}.2.bias"])
        return rec(f"{name}.3", _act(sd, f"{name}.3", t, activation))

    def dec(name, t):
        t = F.conv_transpose2d(t, sd[f"{name}.0.weight"], sd[f"{name}.0.bias"])
        t = rec(f"{name}.1", _act(sd, f"{name}.1", t, activation))
        t = F.conv_transpose2d(t, sd[f"{name}.2.weight"], sd[f"{name}.2.bias"])
        return rec(f"{name}.3", _act(sd, f"{name}.3", t, activation))

    def up(name, t):
        return rec(name, F.conv_transpose2d(t, sd[f"{name}.weight"], sd[f"{name}.bias"], stride=2))

    l = F.pad(x, (2, 2, 2, 2), mode="reflect")
    l1 = enc("convs1", l)
    l2 = enc("convs2", F.max_pool2d(l1, 2))
    l3 = enc("convs3", F.max_pool2d(l2, 2))
    l4 = enc("convs4", F.max_pool2d(l3, 2))
    b = F.max_pool2d(l4, 2)
    b = F.conv2d(b, sd["bottom.0.weight"], sd["bottom.0.bias"])
    b = rec("bottom.1", _act(sd, "bottom.1", b, activation))
    b = F.conv_transpose2d(b, sd["bottom.2.weight"], sd["bottom.2.bias"])
    b = rec("bottom.3", _act(sd, "bottom.3", b, activation))
    l = torch.cat([up("up1", b), l4], dim=1)
    l = torch.cat([up("up2", dec("tconvs1", l)), l3], dim=1)
    l = torch.cat([up("up3", dec("tconvs2", l)), l2], dim=1)
    l = torch.cat([up("up4", dec("tconvs3", l)), l1], dim=1)
    l = dec("tconvs4", l)
    l = rec("tconvs4.4", F.conv2d(l, sd["tconvs4.4.weight"], sd["tconvs4.4.bias"]))
    return l[:, :, 2:-2, 2:-2]


def utnet_flops(cs, funit=64):
    """Algorithmic FLOP (2*MAC) of one UtNet tile; conv Hout^2*Cin*Cout*k^2, convT Hin^2*Cin*Cout*k^2
    (equals torch FlopCounterMode on the reference)."""
    f = funit
    mac = 0
    h = cs + 4
    chans = [(3, f), (f, 2 * f), (2 * f, 4 * f), (4 * f, 8 * f)]
    for ci, co in chans:
        mac += (h - 2) ** 2 * ci * co * 9
        mac += (h - 4) ** 2 * co * co * 9
        h = (h - 4) // 2
    mac += (h - 2) ** 2 * 8 * f * 16 * f * 9          # bottom.0
    mac += (h - 2) ** 2 * 16 * f * 16 * f * 9         # bottom.2 (convT: Hin^2)
    c = 16 * f
    for _ in range(4):
        mac += h * h * c * (c // 2) * 4                # upN (convT 2x2 s2)
        h *= 2
        mac += h * h * c * (c // 2) * 9                # tconvsN.0
        mac += (h + 2) ** 2 * (c // 2) * (c // 2) * 9  # tconvsN.2
        h += 4
        c //= 2
    mac += h * h * f * 3                               # tconvs4.4 (1x1)
    return 2 * mac


# ----------------------------------------------------------------------------- UNet

def _double_conv(sd, p, x):
    for k in (0, 3):
        x = F.conv2d(x, sd[f"{p}.{k}.weight"], sd[f"{p}.{k}.bias"], padding=1)
        x = F.batch_norm(x, sd[f"{p}.{k + 1}.running_mean"], sd[f"{p}.{k + 1}.running_var"],
                         sd[f"{p}.{k + 1}.weight"], sd[f"{p}.{k + 1}.bias"], training=False, eps=1e-5)
        x = F.relu(x)
    return x


def unet_forward(sd, x, find_noise=False):
    """ThirdPartyNets.py:138-169 in eval mode: skip FIRST, up-sampled second in the cat (:124)."""
    x1 = _double_conv(sd, "inc.conv.conv", x)
    skips = [x1]
    t = x1
    for n in (1, 2, 3, 4):
        t = _double_conv(sd, f"down{n}.mpconv.1.conv", F.max_pool2d(t, 2))
        skips.append(t)
    t = skips[4]
    for n, skip in zip((1, 2, 3, 4), (skips[3], skips[2], skips[1], skips[0])):
        u = F.conv_transpose2d(t, sd[f"up{n}.up.weight"], sd[f"up{n}.up.bias"], stride=2)
        dy, dx = skip.size(2) - u.size(2), skip.size(3) - u.size(3)
        u = F.pad(u, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        t = _double_conv(sd, f"up{n}.conv.conv", torch.cat([skip, u], dim=1))
    t = F.conv2d(t, sd["outc.conv.weight"], sd["outc.conv.bias"])
    if find_noise:
        return x - torch.sigmoid(t)
    return torch.sigmoid(t)
