"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Tolerance for fp32 layers / networks: max|y_hip - y_ref| <= 1e-3 absolute per pixel (BASELINE.json north_star)
AND, because seeded-random-weight outputs are O(0.1), max|diff| / max|y_ref| <= 1e-3.
Tile geometry, gathered bytes and stitching of identical tile outputs: bit-exact.
"""
import ctypes
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from nind_denoise_amd import _lib, synth

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-3
REL_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need a real MI355X")
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture
def whole_k(dev, monkeypatch):
    """Plumbing tests that compare two groupings of the same tiles bit for bit run with the split-K tail off (the per-call flag
    ND_FLAG_NO_SPLITK, which UtNet.split_k = False sets): with it on, a tile's fp32 sums may be re-associated depending on
    where the tile sits in its batch."""
    from nind_denoise_amd.networks.UtNet import UtNet
    monkeypatch.setattr(UtNet, "split_k", False)
    yield


def assert_close(y, ref, what=""):
    y = y.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert y.shape == ref.shape, (what, y.shape, ref.shape)
    assert torch.isfinite(y).all(), what
    err = (y - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= ABS_TOL, f"{what}: max abs err {err:.3e} > {ABS_TOL}"
    assert err <= REL_TOL * max(scale, 1e-6), f"{what}: max abs err {err:.3e} vs max|ref| {scale:.3e}"
    return err


# ---------------------------------------------------------------------------- single layers

def layer_forward(dev, kind, x, w, b, act="none", slope=0.25, variant=-1, dtype="f32", flags=0):
    lib = _lib.load()
    k = _lib.KIND[kind]
    dt = _lib.DTYPE[dtype]
    B, cin, H, W = x.shape
    cout = w.shape[0] if kind in ("conv3", "conv1", "conv2s2") else w.shape[1]
    nbytes = lib.nd_layer_packed_bytes(k, cin, cout, dt)
    packed = torch.empty(nbytes // 4, dtype=torch.float32)
    wc, bc = w.contiguous(), b.contiguous()
    _lib.check(lib.nd_layer_pack(k, cin, cout, dt, wc.data_ptr(), bc.data_ptr(), packed.data_ptr(), nbytes))
    packed = packed.to(dev)
    oh, ow = {"conv3": (H - 2, W - 2), "convT3": (H + 2, W + 2), "convT2s2": (2 * H, 2 * W), "conv1": (H, W),
              "conv2s2": (H // 2, W // 2)}[kind]
    y = torch.full((B, cout, oh, ow), float("nan"), dtype=torch.float32, device=dev)
    wsb = lib.nd_layer_workspace_bytes(k, B, cin, cout, H, W, dt)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    xd = x.to(dev).contiguous()
    _lib.check(lib.nd_layer_forward(k, _lib.ACT[act], slope, dt, packed.data_ptr(), xd.data_ptr(), B, cin, H, W,
                                    cout, y.data_ptr(), ws.data_ptr(), wsb, variant, flags, _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
    return y


def ref_layer(kind, x, w, b, act, slope):
    if kind == "conv3" or kind == "conv1":
        y = F.conv2d(x, w, b)
    elif kind == "convT3":
        y = F.conv_transpose2d(x, w, b)
    else:
        y = F.conv_transpose2d(x, w, b, stride=2)
    if act == "PReLU":
        y = F.prelu(y, torch.tensor([slope]))
    elif act == "ELU":
        y = F.elu(y)
    elif act == "Hardswish":
        y = F.hardswish(y)
    return y


def rnd(shape, seed, scale=1.0):
    return (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1) * scale


LAYER_CASES = [
    # kind, B, cin, cout, H, W, act
    ("conv3", 2, 8, 32, 20, 24, "PReLU"),
    ("conv3", 1, 3, 64, 40, 40, "PReLU"),        # first layer: 3 channels padded to 8
    ("conv3", 3, 64, 64, 34, 30, "PReLU"),
    ("conv3", 1, 64, 128, 66, 66, "none"),
    ("conv3", 2, 24, 40, 17, 19, "PReLU"),       # odd sizes, Cout not a multiple of 32
    ("conv3", 1, 128, 256, 13, 13, "PReLU"),
    ("convT3", 2, 16, 32, 11, 13, "PReLU"),
    ("convT3", 1, 128, 64, 40, 36, "PReLU"),
    ("convT3", 1, 256, 256, 11, 11, "PReLU"),
    ("convT2s2", 2, 64, 32, 13, 13, "none"),
    ("convT2s2", 1, 128, 64, 30, 26, "none"),
    ("convT2s2", 1, 16, 8, 9, 7, "none"),        # 4*Cout = 32: one M tile holds all four sub-positions
    ("conv1", 2, 64, 32, 21, 23, "none"),
    ("conv3", 1, 16, 16, 30, 30, "ELU"),
    ("conv3", 1, 16, 16, 30, 30, "Hardswish"),
    ("conv3", 1, 64, 64, 270, 270, "PReLU"),     # full-width rows (Wb = 270 -> largest LDS halo image at cs=264)
]


@pytest.mark.parametrize("case", LAYER_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_layer_parity(dev, case):
    kind, B, cin, cout, H, W, act = case
    k = {"conv3": 3, "convT3": 3, "convT2s2": 2, "conv1": 1}[kind]
    x = rnd((B, cin, H, W), 1)
    bound = 1.0 / np.sqrt(cin * k * k)
    wshape = (cout, cin, k, k) if kind in ("conv3", "conv1") else (cin, cout, k, k)
    w = rnd(wshape, 2, bound * 1.7)
    b = rnd((cout,), 3, 0.2)
    slope = 0.13
    y = layer_forward(dev, kind, x, w, b, act, slope)
    assert_close(y, ref_layer(kind, x, w, b, act, slope), str(case))


SPLIT_CASES = [
    # the deep, few-pixel layers of a training batch (UtNet(64), crop 136, 30 crops): fewer tiles than CUs -> K is split
    ("conv3", 30, 512, 1024, 5, 5, "PReLU"),     # bottom.0: 9 valid pixels per image, tiles run across the whole batch
    ("convT3", 30, 1024, 1024, 3, 3, "PReLU"),   # bottom.2
    ("convT3", 6, 1024, 512, 10, 10, "PReLU"),   # tconvs1.0
    ("conv3", 4, 256, 512, 12, 12, "ELU"),       # generic activation through the finish kernel
    ("convT2s2", 8, 256, 128, 5, 5, "none"),     # pixel-shuffle store in the finish kernel
    ("conv1", 3, 512, 64, 9, 9, "none"),
    ("conv3", 70, 64, 64, 66, 66, "PReLU"),      # > 1 round of tiles: whole tiles first, only the tail round is split
]


@pytest.mark.parametrize("case", SPLIT_CASES, ids=lambda c: "-".join(str(v) for v in c))
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_layer_split_k(dev, case, dtype):
    """Split-K tail (conv_f32.hip: plan_split / k_split_finish): same result as the unsplit launch up to fp32 re-association,
    and both match torch."""
    kind, B, cin, cout, H, W, act = case
    k = {"conv3": 3, "convT3": 3, "convT2s2": 2, "conv1": 1}[kind]
    x = rnd((B, cin, H, W), 1)
    bound = 1.0 / np.sqrt(cin * k * k)
    wshape = (cout, cin, k, k) if kind in ("conv3", "conv1") else (cin, cout, k, k)
    w = rnd(wshape, 2, bound * 1.7)
    b = rnd((cout,), 3, 0.2)
    if dtype == "bf16":
        x, w = x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
    y_on = layer_forward(dev, kind, x, w, b, act, 0.13, dtype=dtype)
    y_off = layer_forward(dev, kind, x, w, b, act, 0.13, dtype=dtype, flags=_lib.FLAG_NO_SPLITK)
    ref = ref_layer(kind, x, w, b, act, 0.13)
    if dtype == "f32":
        assert_close(y_on, ref, f"{case} split")
        assert_close(y_off, ref, f"{case} unsplit")
        assert (y_on - y_off).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    else:
        # 16-bit storage of the result: one bf16 rounding of the same fp32 sums (up to their re-association)
        tol = 2.0 ** -7 * max(1.0, ref.abs().max().item())
        assert (y_on.cpu() - ref).abs().max().item() <= tol and (y_off.cpu() - ref).abs().max().item() <= tol
        assert (y_on - y_off).abs().max().item() <= tol


WINO_CASES = [
    # kind, B, cin, cout, H, W, act
    ("conv3", 1, 3, 64, 40, 40, "PReLU"),        # first layer: 3 channels padded to one K block (1-D form only: Cin % 16)
    ("conv3", 3, 64, 64, 70, 66, "PReLU"),       # several tiles, rows shorter than a tile
    ("convT3", 2, 64, 64, 266, 266, "PReLU"),    # full-width rows of a 264-pixel tile (largest LDS halo image)
    ("conv3", 2, 16, 32, 10, 12, "PReLU"),       # Cout < 128: the generic 1-tap variant
    ("conv3", 3, 64, 128, 21, 19, "PReLU"),      # odd output sizes: partial tiles masked
    ("convT3", 2, 32, 256, 11, 11, "PReLU"),     # transposed layer (flipped taps, zero border), 256-row GEMM tile
    ("conv3", 2, 256, 256, 30, 30, "none"),
    ("convT3", 1, 512, 256, 28, 26, "ELU"),
    ("conv3", 5, 512, 512, 13, 13, "Hardswish"),
]


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "-".join(str(v) for v in c))
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6])
def test_layer_winograd(dev, case, tile):
    """Winograd F(t x t, 3 x 3) form of a 3x3 layer (csrc/winograd.hip: input transform, 16 / 36 batched GEMMs in one launch,
    output transform) against torch and against the direct kernel."""
    kind, B, cin, cout, H, W, act = case
    # tile 1 | 3 = the 1-D F(2,3) | F(4,3) form fused into the implicit-GEMM kernel; 2 | 4 = the three-pass F(2x2) | F(4x4) form;
    # 5 = the F(4,3) form with the input transform shared by the workgroup through LDS (conv_w2d); 6 = three-pass F(6x6)
    x = rnd((B, cin, H, W), 1)
    bound = 1.0 / np.sqrt(cin * 9)
    wshape = (cout, cin, 3, 3) if kind == "conv3" else (cin, cout, 3, 3)
    w = rnd(wshape, 2, bound * 1.7)
    b = rnd((cout,), 3, 0.2)
    if tile % 2 == 0 and cin % 16:
        pytest.skip("the three-pass form needs Cin % 16 == 0")
    lib = _lib.load()
    k = _lib.KIND[kind]
    nbytes = lib.nd_winograd_packed_bytes(tile, cin, cout)
    packed = torch.empty(nbytes // 4, dtype=torch.float32)
    wc, bc = w.contiguous(), b.contiguous()
    _lib.check(lib.nd_winograd_pack(tile, k, cin, cout, wc.data_ptr(), bc.data_ptr(), packed.data_ptr(), nbytes))
    packed = packed.to(dev)
    oh, ow = (H - 2, W - 2) if kind == "conv3" else (H + 2, W + 2)
    y = torch.full((B, cout, oh, ow), float("nan"), dtype=torch.float32, device=dev)
    wsb = lib.nd_layer_winograd_workspace_bytes(tile, k, B, cin, cout, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    xd = x.to(dev).contiguous()
    _lib.check(lib.nd_layer_forward_winograd(tile, k, _lib.ACT[act], 0.13, packed.data_ptr(), xd.data_ptr(), B, cin, H, W, cout,
                                             y.data_ptr(), ws.data_ptr(), wsb, 0, _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
    ref = ref_layer(kind, x, w, b, act, 0.13)
    err = assert_close(y, ref, f"winograd F({tile},3) {case}")
    direct = layer_forward(dev, kind, x, w, b, act, 0.13)
    scale = max(1.0, ref.abs().max().item())
    assert (y - direct).abs().max().item() <= (2e-5 if tile <= 2 else 1e-4) * scale
    print(f"winograd tile code {tile} {case}: max abs err {err:.2e}")


def test_layer_asymmetric_identity(dev):
    # exact-integer check of the fragment maps: an asymmetric integer kernel and integer inputs give integer
    # results that must match bit for bit (a transposed tap or swapped channel pairing cannot hide)
    cin, cout, H, W = 16, 32, 12, 14
    x = torch.arange(cin * H * W, dtype=torch.float32).reshape(1, cin, H, W) % 17 - 8
    w = (torch.arange(cout * cin * 9, dtype=torch.float32).reshape(cout, cin, 3, 3) % 7) - 3
    b = torch.arange(cout, dtype=torch.float32)
    y = layer_forward(dev, "conv3", x, w, b, "none")
    assert torch.equal(y.cpu(), F.conv2d(x, w, b))
    wt = (torch.arange(cout * cin * 9, dtype=torch.float32).reshape(cin, cout, 3, 3) % 5) - 2
    y = layer_forward(dev, "convT3", x, wt, b, "none")
    assert torch.equal(y.cpu(), F.conv_transpose2d(x, wt, b))
    wu = (torch.arange(cout * cin * 4, dtype=torch.float32).reshape(cin, cout, 2, 2) % 5) - 2
    y = layer_forward(dev, "convT2s2", x, wu, b, "none")
    assert torch.equal(y.cpu(), F.conv_transpose2d(x, wu, b, stride=2))


def test_every_conv_variant(dev):
    lib = _lib.load()
    x = rnd((2, 64, 36, 40), 5)
    cases = {9: ("conv3", rnd((128, 64, 3, 3), 6, 0.07)), 1: ("conv1", rnd((128, 64, 1, 1), 7, 0.2)),
             4: ("conv2s2", rnd((128, 64, 2, 2), 10, 0.1))}
    up_w = rnd((64, 32, 2, 2), 8, 0.1)
    b = rnd((128,), 9, 0.1)
    tested = 0
    for v in range(lib.nd_num_conv_variants()):
        name = lib.nd_conv_variant_name(v).decode()
        dtype = name.split("_")[0]
        tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
        q = (lambda t: t.to(tdt).float())       # what the 16-bit paths store
        if "_uptrue" in name:
            y = layer_forward(dev, "convT2s2", x, up_w, b[:32], "none", variant=v, dtype=dtype)
            ref = F.conv_transpose2d(q(x), q(up_w), b[:32], stride=2)
        else:
            taps = 9 if "_t9_" in name else (4 if "_t4_" in name else 1)
            kind, w = cases[taps]
            y = layer_forward(dev, kind, x, w, b, "PReLU", 0.2, variant=v, dtype=dtype)
            ref = F.prelu(F.conv2d(q(x), q(w), b, stride=2 if taps == 4 else 1), torch.tensor([0.2]))
        if dtype == "f32":
            assert_close(y, ref, name)
        else:   # same 16-bit operands, fp32 accumulation; the stored result is rounded once to 16 bits
            ulp = 2.0 ** (-8 if dtype == "bf16" else -11)
            err = (y.cpu() - ref).abs()
            assert (err <= ulp * ref.abs() + 1e-6).all(), (name, err.max().item())
        tested += 1
    assert tested == lib.nd_num_conv_variants()


def test_maxpool(dev):
    lib = _lib.load()
    x = rnd((2, 24, 26, 30), 4)
    y = torch.empty((2, 24, 13, 15), dtype=torch.float32, device=dev)
    ws = torch.empty(1 << 22, dtype=torch.uint8, device=dev)
    xd = x.to(dev)
    _lib.check(lib.nd_maxpool2_forward(xd.data_ptr(), 2, 24, 26, 30, y.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), F.max_pool2d(x, 2))


# ---------------------------------------------------------------------------- tiler

def _geoms():
    with open(os.path.join(os.path.dirname(__file__), "golden", "tiler_geoms.json")) as f:
        return json.load(f)


def test_tile_gather_bitexact_vs_oracle_and_golden(dev):
    import hashlib
    from oracle import tiler as otiler
    from nind_denoise_amd import pipeline
    for g in _geoms():
        frame = synth.make_frame(g["W"], g["H"], seed=g["seed"])
        img = torch.from_numpy(frame).to(dev)
        grid = otiler.TileGrid(g["W"], g["H"], g["cs"], g["ucs"], g["ol"])
        ids = sorted(int(k) for k in g["tile_sha"])
        for i in ids:
            t = pipeline.gather_tiles(img, g["cs"], g["ucs"], g["ol"], i, 1)[0].cpu().numpy()
            assert hashlib.sha256(t.tobytes()).hexdigest() == g["tile_sha"][str(i)], (g["W"], g["H"], i)
        # a contiguous batch against the oracle
        n = min(grid.size, 7)
        t = pipeline.gather_tiles(img, g["cs"], g["ucs"], g["ol"], grid.size - n, n).cpu().numpy()
        for k in range(n):
            assert np.array_equal(t[k], otiler.gather_tile(frame, grid, grid.size - n + k))


@pytest.mark.parametrize("batch", [1, 5, 64])
def test_identity_roundtrip_bitexact(dev, batch):
    from nind_denoise_amd import pipeline
    for g in _geoms()[:8]:
        if (g["cs"] - g["ucs"]) % 2:
            continue
        frame = synth.make_frame(g["W"], g["H"], seed=g["seed"])
        img = torch.from_numpy(frame).to(dev)
        out = pipeline.denoise_frame(lambda x: x, img, g["cs"], g["ucs"], g["ol"], batch=batch)
        assert torch.equal(out, img), (g["W"], g["H"], g["cs"], g["ucs"], g["ol"])


def test_stitch_bitexact_vs_oracle_nonidentity(dev):
    # stitch of identical (random) tile outputs, including the cs-ucs odd quirk geometry
    from oracle import tiler as otiler
    from nind_denoise_amd import pipeline
    for g in _geoms()[:8]:
        grid = otiler.TileGrid(g["W"], g["H"], g["cs"], g["ucs"], g["ol"])
        rng = np.random.default_rng(g["seed"])
        tiles = rng.standard_normal((grid.size, 3, g["cs"], g["cs"]), dtype=np.float32)
        ref = np.zeros((3, g["H"], g["W"]), dtype=np.float32)
        for i in range(grid.size):
            otiler.stitch_add(ref, tiles[i], grid, i)
        canvas = torch.zeros((3, g["H"], g["W"]), dtype=torch.float32, device=dev)
        td = torch.from_numpy(tiles).to(dev)
        for b0 in range(0, grid.size, 5):
            pipeline.stitch_tiles(canvas, td[b0:b0 + 5], g["cs"], g["ucs"], g["ol"], b0)
        assert np.array_equal(canvas.cpu().numpy(), ref), (g["W"], g["H"], g["cs"], g["ucs"], g["ol"])


def test_g24_frame_identity_full_size(dev):
    # BASELINE config 2 geometry at full size: size-independent property (identity model reproduces the frame)
    from nind_denoise_amd import pipeline
    frame = synth.make_frame(6000, 4000, seed=24)
    img = torch.from_numpy(frame).to(dev)
    out = pipeline.denoise_frame(lambda x: x, img, 264, 200, 64, batch=128)
    assert torch.equal(out, img)


def test_g61_frame_identity_full_size(dev):
    # BASELINE config 4 geometry at full size (9504x6336, cs=520 ucs=456 ol=64 -> 25x16 tiles): same property
    from nind_denoise_amd import pipeline
    assert pipeline.tile_count(9504, 6336, 520, 456, 64) == 400
    g = torch.Generator(device=dev).manual_seed(61)
    img = torch.rand((3, 6336, 9504), generator=g, device=dev)
    out = pipeline.denoise_frame(lambda x: x, img, 520, 456, 64, batch=50)
    assert torch.equal(out, img)


# ---------------------------------------------------------------------------- networks

def test_utnet_f8_golden(dev, golden_dir):
    from nind_denoise_amd.networks.UtNet import UtNet
    d = np.load(os.path.join(golden_dir, "utnet_f8.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd/")}
    net = UtNet(funit=8)
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    for cs in (104, 120):
        y = net(torch.from_numpy(d[f"x{cs}"]).to(dev))
        assert_close(y, torch.from_numpy(d[f"y{cs}"]), f"utnet f8 cs{cs}")


def test_utnet_activation_variants_golden(dev, golden_dir):
    from nind_denoise_amd.networks.UtNet import UtNet
    d = np.load(os.path.join(golden_dir, "utnet_act_variants.npz"))
    for act in ("ELU", "Hardswish"):
        net = UtNet(funit=8, activation=act)
        net.load_state_dict(synth.make_utnet_state_dict(funit=8, seed=11, activation=act))
        y = net.eval().to(dev)(torch.from_numpy(d["x"]).to(dev))
        assert_close(y, torch.from_numpy(d[f"y_{act}"]), act)


def test_utnet_f64_cs264_golden(dev, golden_dir):
    from nind_denoise_amd.networks.UtNet import UtNet
    d = np.load(os.path.join(golden_dir, "utnet_f64_cs264.npz"))
    sd = synth.make_utnet_state_dict(funit=64, seed=123)
    assert synth.state_dict_digest(sd) == str(d["sd_digest"])
    net = UtNet()
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    x = torch.from_numpy(d["x"]).to(dev)
    err = assert_close(net(x), torch.from_numpy(d["y"]), "utnet f64 cs264")
    # run-to-run determinism: the same batch gives the same bits.  Across batch compositions the split-K tail of a launch
    # may re-associate a tile's fp32 sums, so that comparison carries a tolerance; with ND_FLAG_NO_SPLITK on the call
    # (net.split_k = False) a tile's bits do not depend on the batch around it
    xb = torch.cat([x, x.flip(3), x])
    yb = net(xb)
    assert torch.equal(yb, net(xb))
    assert (yb[0] - net(x)[0]).abs().max().item() < 1e-5 and (yb[0] - yb[2]).abs().max().item() < 1e-5
    net.split_k = False
    yb = net(xb)
    assert torch.equal(yb[0], yb[2]) and torch.equal(yb[0], net(x)[0])
    assert_close(yb[:1], torch.from_numpy(d["y"]), "utnet f64 cs264, split-K off")
    net.split_k = True
    # the default path runs the >= 128-channel 3x3 layers in Winograd F(4x4,3x3) form and the narrower ones in the fused 1-D
    # form; the direct form of every layer (ND_FLAG_DIRECT_CONV) must meet the same bar, and the two agree far inside it
    y_w = net(x)
    assert net.winograd and net.flags == 0
    net.winograd = False
    y_d = net(x)
    err_d = assert_close(y_d, torch.from_numpy(d["y"]), "utnet f64 cs264, direct convolution everywhere")
    net.winograd = True
    assert (y_w - y_d).abs().max().item() < 1e-5 and not torch.equal(y_w, y_d)
    print(f"utnet f64 cs264 direct-only max abs err {err_d:.3e}, winograd vs direct {(y_w - y_d).abs().max().item():.3e}")
    print(f"utnet f64 cs264 max abs err {err:.3e}")


def test_utnet_vs_oracle_other_sizes(dev):
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    sd = synth.make_utnet_state_dict(funit=16, seed=5)
    net = UtNet(funit=16)
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    for cs, B in ((104, 3), (136, 2), (184, 1)):
        x = torch.rand(B, 3, cs, cs, generator=torch.Generator().manual_seed(cs))
        with torch.no_grad():
            ref = onet.utnet_forward(sd, x)
        assert_close(net(x.to(dev)), ref, f"f16 cs{cs}")


def test_utnet_f64_smallest_tile_and_many_tiles(dev):
    # production width at the smallest valid tile (1x1 pixels at the bottom level: Winograd tiles mostly padding) and more tiles
    # than one three-pass Winograd pass takes (kWinoChunk = 256): every tile must equal the same tile run alone
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    sd = synth.make_utnet_state_dict(funit=64, seed=123)
    net = UtNet()
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    x = torch.rand(3, 3, 104, 104, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        ref = onet.utnet_forward(sd, x)
    assert_close(net(x.to(dev)), ref, "f64 cs104")
    xb = x.repeat(90, 1, 1, 1)[:262].to(dev)           # 262 tiles: one full pass of 256 and a tail of 6
    yb = net(xb)
    assert_close(yb[:3], ref, "f64 cs104, first tiles of 262")
    assert_close(yb[-3:], net(xb[-3:]), "f64 cs104, tail tiles of 262")
    assert (yb[255] - yb[0]).abs().max().item() < 1e-5 and (yb[258] - yb[0]).abs().max().item() < 1e-5   # 255, 258 = tile 0 again


def test_utnet_f64_winograd_tile_remainders(dev):
    # production width at tile sizes whose three-pass layers leave every remainder of the 6 x 6 Winograd output tile (and partial
    # 8-row strips of the fused 1-D form), square and non-square, odd batch sizes
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    sd = synth.make_utnet_state_dict(funit=64, seed=7)
    net = UtNet()
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    for (h, w), B in (((120, 120), 3), ((136, 136), 1), ((152, 152), 5), ((200, 200), 2), ((152, 104), 2), ((104, 216), 1)):
        x = torch.rand(B, 3, h, w, generator=torch.Generator().manual_seed(h + w))
        with torch.no_grad():
            ref = onet.utnet_forward(sd, x)
        err = assert_close(net(x.to(dev)), ref, f"f64 {h}x{w} batch {B}")
        assert err <= 2e-6 * max(1.0, ref.abs().max().item()), (h, w, err)     # measured ~1e-7: the fp32 path, not just the 1e-3 bar


def test_device_packed_weights_equal_host_packed(dev):
    # model load packs the fp32 blob (direct + both Winograd forms) in HBM; the host packer is the reference layout
    from nind_denoise_amd.networks.UtNet import UtNet
    for funit in (16, 64):
        sd = synth.make_utnet_state_dict(funit=funit, seed=3)
        net = UtNet(funit=funit)
        net.load_state_dict(sd)
        net = net.eval().to(dev)
        b_dev = net.packed_weights(dev).cpu()
        net.pack_on_device = False
        net._packed.clear()
        b_host = net.packed_weights(dev).cpu()
        assert b_dev.shape == b_host.shape
        scale = b_host.abs().max().item()
        err = (b_dev - b_host).abs().max().item()                            # (plain floats below: never let pytest render a
        pad_leak = b_dev[b_host == 0].abs().max().item()                     #  150 M-element tensor into a failure message)
        assert err <= 2e-6 * scale, err                                      # Winograd weights: fp32 vs double transform
        assert pad_leak <= 2e-6 * scale, pad_leak                            # padding rows / channels / gaps stay (numerically) zero
        x = torch.rand(1, 3, 120, 120, generator=torch.Generator().manual_seed(1)).to(dev)
        y_host = net(x)
        net.pack_on_device = True
        net._packed.clear()
        assert (net(x) - y_host).abs().max().item() < 1e-6


def test_utnet_rejects_invalid_cs_and_cpu(dev):
    from nind_denoise_amd.networks.UtNet import UtNet
    net = UtNet(funit=8).to(dev)
    for cs in (128, 256, 512):
        with pytest.raises(ValueError):
            net(torch.zeros(1, 3, cs, cs, device=dev))
    with pytest.raises(RuntimeError):
        UtNet(funit=8)(torch.zeros(1, 3, 104, 104))


def test_frame_end_to_end_vs_oracle(dev, whole_k):
    # crop -> UtNet -> stitch on a small frame: fused device loop vs the oracle loop, and fused == unfused bit for bit
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    from oracle import tiler as otiler
    sd = synth.make_utnet_state_dict(funit=16, seed=9)
    net = UtNet(funit=16)
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    W, H, cs, ucs, ol = 333, 290, 120, 88, 16
    frame = synth.make_frame(W, H, seed=3)

    def model_fn(x):
        with torch.no_grad():
            return onet.utnet_forward(sd, torch.from_numpy(x)).numpy()

    ref = otiler.denoise_frame(frame, cs, ucs, ol, model_fn, batch=4)
    img = torch.from_numpy(frame).to(dev)
    out_roi = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=5)    # default: only what the useful tile centres depend on
    assert_close(out_roi, torch.from_numpy(ref), "frame e2e")
    net.useful_only = False                                             # whole tiles in every layer, as forward() computes them
    out = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=5)
    assert_close(out, torch.from_numpy(ref), "frame e2e, whole tiles")
    assert (out - out_roi).abs().max().item() <= 2e-6 * max(1.0, out.abs().max().item())
    out2 = pipeline.denoise_frame(lambda x: net(x), img, cs, ucs, ol, batch=5)
    assert torch.equal(out, out2), "fused gather/stitch path differs from the unfused one"
    # a different batch size only changes how tiles are grouped, never the result
    out3 = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=3)
    assert torch.equal(out, out3)
    # split-K tail on (the default): same frame up to fp32 re-association of a few tiles' sums
    net.split_k = True
    out4 = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=5)
    assert_close(out4, torch.from_numpy(ref), "frame e2e, split-K on")
    assert (out4 - out).abs().max().item() < 1e-5


def test_cli_end_to_end(dev, tmp_path):
    # denoise_image CLI: float TIFF in -> float TIFF out (the contract denoise.py:430-439 relies on), vs the oracle loop
    from nind_denoise_amd import denoise_image as di
    from nind_denoise_amd.common.libs import imgcodec, np_imgops
    from oracle import networks as onet
    from oracle import tiler as otiler
    sd = synth.make_utnet_state_dict(funit=8, seed=21)
    mdir = tmp_path / "2021-06-14T20_27_nn_train"
    mdir.mkdir()
    torch.save(sd, mdir / "generator_650.pt")
    frame = synth.make_frame(310, 275, seed=8)
    inp, outp = str(tmp_path / "x_s1.tif"), str(tmp_path / "x_s1_denoised.tiff")
    imgcodec.write_tiff(inp, np.ascontiguousarray(frame.transpose(1, 2, 0)))
    rc = di.main(["--network", "UtNet", "--model_path", str(mdir / "generator_650.pt"), "--model_parameters", "funit=8",
                  "--input", inp, "--output", outp, "--cs", "120", "--ucs", "88", "-ol", "16", "--exif_method", "noexif"])
    assert rc == 0 and os.path.isfile(outp)
    got = np_imgops.img_path_to_np_flt(outp)

    def model_fn(x):
        with torch.no_grad():
            return onet.utnet_forward(sd, torch.from_numpy(x)).numpy()

    ref = otiler.denoise_frame(frame, 120, 88, 16, model_fn, batch=8)
    assert_close(torch.from_numpy(got), torch.from_numpy(ref), "cli")
    # the reference's triple from the dataset object
    ds = di.OneImageDS(frame, 120, 88, 16, device=dev)
    grid = otiler.TileGrid(310, 275, 120, 88, 16)
    assert len(ds) == grid.size
    t, ud, us = ds[len(ds) - 1]
    assert np.array_equal(t.cpu().numpy(), otiler.gather_tile(frame, grid, grid.size - 1))
    assert tuple(ud.tolist()) == grid.geom(grid.size - 1)[2] and tuple(us.tolist()) == grid.geom(grid.size - 1)[3]
    assert ud.dtype == torch.int32


def test_device_side_sample_conversion_is_bit_identical(dev, tmp_path):
    """CLI I/O with the sample conversions on the GPU (np_imgops.img_path_to_device_flt, pt_helpers.tensor_to_imgfile on a CUDA
    tensor): the same IEEE float32 operations as the host path of np_imgops.py:12-29 / pt_helpers.py:22-34 -- frames and files
    must be byte-identical."""
    from nind_denoise_amd.common.libs import imgcodec, np_imgops, pt_helpers
    rng = np.random.default_rng(5)
    u16 = rng.integers(0, 65536, size=(37, 53, 3), dtype=np.uint16)
    u16[0, :8, 0] = [0, 1, 32767, 32768, 65534, 65535, 255, 256]
    imgcodec.write_tiff(str(tmp_path / "a.tif"), u16)
    imgcodec.write_png(str(tmp_path / "a.png"), u16)
    imgcodec.write_tiff(str(tmp_path / "b.tif"), (u16 >> 8).astype(np.uint8))
    imgcodec.write_tiff(str(tmp_path / "c.tiff"), rng.standard_normal((37, 53, 3)).astype(np.float32))
    for name in ("a.tif", "a.png", "b.tif", "c.tiff", "NIND_bananapi_ISO50_20_30_104.png"):
        path = str(tmp_path / name) if not name.startswith("NIND") else os.path.join(os.path.dirname(__file__), "golden", name)
        host = np_imgops.img_path_to_np_flt(path)
        got = np_imgops.img_path_to_device_flt(path, dev)
        assert got.dtype == torch.float32 and got.is_contiguous() and np.array_equal(got.cpu().numpy(), host), name
    # writer: values below 0, above 1, exact .5 ties of x * 65535, denormals
    t = torch.from_numpy(rng.uniform(-0.2, 1.2, size=(3, 41, 29)).astype(np.float32))
    t[0, 0, :6] = torch.tensor([0.5 / 65535, 1.5 / 65535, 2.5 / 65535, 1.0, 0.0, 1e-40])
    for ext in (".tif", ".png", ".tiff"):
        pt_helpers.tensor_to_imgfile(t.clone(), str(tmp_path / ("h" + ext)))
        pt_helpers.tensor_to_imgfile(t.to(dev), str(tmp_path / ("d" + ext)))
        with open(tmp_path / ("h" + ext), "rb") as f1, open(tmp_path / ("d" + ext), "rb") as f2:
            assert f1.read() == f2.read(), ext


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_pool_fused_into_conv_epilogue(dev, dtype):
    """16-bit storage: the MaxPool2d(2) that follows convsN.2 is written from that layer's epilogue (conv_qp walks the pixels in
    2-row bands, so a 2x2 block is a quad of adjacent lanes: maximum by two DPP permutes, then the same rounding and 16-byte store).
    The pooled values are those of the separate kernel bit for bit (ND_FLAG_UNFUSED_POOL), on narrow and production-width nets,
    tiles that do and do not divide into whole bands, and the fused path really runs (the pool steps take no time of their own)."""
    import bench
    from nind_denoise_amd.networks.UtNet import UtNet
    for funit, cs, batch, seed in ((16, 104, 3, 7), (16, 136, 5, 8), (64, 264, 4, 9)):
        net = UtNet(funit=funit)
        net.load_state_dict(synth.make_utnet_state_dict(funit=funit, seed=seed))
        net = net.eval().to(dev).set_compute_dtype(dtype)
        net.split_k = False      # (a pooling layer keeps its tiles whole; the separate-kernel run must not re-associate them either)
        x = torch.rand(batch, 3, cs, cs, generator=torch.Generator().manual_seed(seed)).to(dev)
        with torch.no_grad():
            y_fused = net(x).clone()
            net.fused_pool = False
            y_sep = net(x).clone()
            net.fused_pool = True
        assert torch.isfinite(y_fused).all() and torch.equal(y_fused, y_sep), (funit, cs)
    steps = bench.conv_stack_profile(net, 264, 4, dev, reps=1)
    pools = [s_ for s_ in steps if s_["form"] == "pool"]   # (fused on the two large levels; the small ones keep the separate kernel)
    assert len(pools) == 4 and all(s_["ms"] < 0.02 for s_ in pools[:2]), [s_["ms"] for s_ in pools]


def test_resident_worker_serves_cli_clients(dev, tmp_path):
    """Row f2: `python -m nind_denoise_amd.serve --socket PATH` started once, three images through
    `python -m nind_denoise_amd.denoise_image ... --server PATH` clients (fresh light processes that load neither torch nor the
    HIP library), relative paths resolved in the client's directory, the reference's printed lines relayed, the second and
    third request served by the resident model; outputs == the oracle loop, as in test_cli_end_to_end.  A failing request
    returns its status and message and leaves the worker up."""
    import subprocess
    import sys
    import time
    from nind_denoise_amd.common.libs import imgcodec, np_imgops
    from oracle import networks as onet
    from oracle import tiler as otiler
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sd = synth.make_utnet_state_dict(funit=8, seed=21)
    torch.save(sd, tmp_path / "generator_650.pt")
    frames = [synth.make_frame(310, 275, seed=30 + k) for k in range(3)]
    for k, fr in enumerate(frames):
        imgcodec.write_tiff(str(tmp_path / f"in{k}.tif"), np.ascontiguousarray(fr.transpose(1, 2, 0)))
    sock = str(tmp_path / "w.sock")
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    worker = subprocess.Popen([sys.executable, "-m", "nind_denoise_amd.serve", "--socket", sock], env=env, cwd=root,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        t0 = time.time()
        while not os.path.exists(sock):
            assert worker.poll() is None, worker.stdout.read()
            assert time.time() - t0 < 180, "worker did not come up"
            time.sleep(0.1)
        # the client must not import torch or the HIP library: -X importtime lists every import of the process
        base = [sys.executable, "-X", "importtime", "-m", "nind_denoise_amd.denoise_image", "--network", "UtNet", "--model_path", "generator_650.pt",
                "--model_parameters", "funit=8", "--cs", "120", "--ucs", "88", "-ol", "16", "--exif_method", "noexif"]
        for k in range(3):
            r = subprocess.run(base + ["--input", f"in{k}.tif", "--output", f"out{k}.tiff", "--server", sock], env=env, cwd=tmp_path,
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout + r.stderr
            assert "Elapsed time: " in r.stdout and f"Wrote denoised image to {tmp_path}/out{k}.tiff" in r.stdout
            assert "| torch" not in r.stderr and "numpy" not in r.stderr and "ctypes" not in r.stderr, "client imported heavy modules"
        # three clients at once: one worker thread per connection, the device sections serialised behind the lock
        ps = [subprocess.Popen(base[:1] + base[3:] + ["--input", f"in{k}.tif", "--output", f"par{k}.tiff", "--server", sock], env=env, cwd=tmp_path,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(3)]
        for k, p_ in enumerate(ps):
            o, e = p_.communicate(timeout=300)
            assert p_.returncode == 0 and f"par{k}.tiff" in o, o + e
        for k in range(3):
            with open(tmp_path / f"out{k}.tiff", "rb") as f1, open(tmp_path / f"par{k}.tiff", "rb") as f2:
                assert f1.read() == f2.read()
        # environment variable instead of the flag; an invalid tile size comes back as the CLI's own message and status
        r = subprocess.run(base[:1] + base[3:-8] + ["--cs", "128", "--ucs", "88", "--input", "in0.tif", "--output", "bad.tiff"],
                           env=dict(env, NIND_DENOISE_SERVER=sock), cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and "not a valid UtNet tile size" in r.stderr and not os.path.exists(tmp_path / "bad.tiff")
        r = subprocess.run([sys.executable, "-m", "nind_denoise_amd.client", "--server", sock, "--ping"], env=env, cwd=tmp_path,
                           capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "7 request(s) served, 1 model(s) resident" in r.stdout, r.stdout + r.stderr
        r = subprocess.run([sys.executable, "-m", "nind_denoise_amd.client", "--server", sock, "--shutdown"], env=env, cwd=tmp_path,
                           capture_output=True, text=True, timeout=60)
        assert r.returncode == 0
        assert worker.wait(timeout=60) == 0
    finally:
        if worker.poll() is None:
            worker.kill()

    def model_fn(x):
        with torch.no_grad():
            return onet.utnet_forward(sd, torch.from_numpy(x)).numpy()

    for k, fr in enumerate(frames):
        got = np_imgops.img_path_to_np_flt(str(tmp_path / f"out{k}.tiff"))
        ref = otiler.denoise_frame(fr, 120, 88, 16, model_fn, batch=8)
        assert_close(torch.from_numpy(got), torch.from_numpy(ref), f"worker image {k}")


def test_cli_debug_crop_dumps(dev, tmp_path, monkeypatch):
    # --debug (denoise_image.py:149-150, 260-269): per-tile crop dumps in ./dbg and the last tile with borders; the canvas is the
    # same as without the flag
    from PIL import Image
    from nind_denoise_amd import denoise_image as di
    from nind_denoise_amd.common.libs import imgcodec, np_imgops
    from oracle import tiler as otiler
    sd = synth.make_utnet_state_dict(funit=8, seed=21)
    torch.save(sd, tmp_path / "generator_1.pt")
    frame = synth.make_frame(200, 180, seed=9)
    inp = str(tmp_path / "in.tif")
    imgcodec.write_tiff(inp, np.ascontiguousarray(frame.transpose(1, 2, 0)))
    monkeypatch.chdir(tmp_path)
    common = ["--network", "UtNet", "--model_path", str(tmp_path / "generator_1.pt"), "--model_parameters", "funit=8", "--input", inp,
              "--cs", "120", "--ucs", "88", "-ol", "16", "--exif_method", "noexif", "-b", "3"]
    assert di.main(common + ["--output", str(tmp_path / "a.tiff")]) == 0
    assert di.main(common + ["--output", str(tmp_path / "b.tiff"), "--debug"]) == 0
    a, b = np_imgops.img_path_to_np_flt(str(tmp_path / "a.tiff")), np_imgops.img_path_to_np_flt(str(tmp_path / "b.tiff"))
    assert np.abs(a - b).max() <= 1e-6
    grid = otiler.TileGrid(200, 180, 120, 88, 16)
    dumps = sorted(os.listdir(tmp_path / "dbg"))
    assert len(dumps) == 3 * grid.size
    n_last, i_last = (grid.size - 1) // 3, (grid.size - 1) % 3
    noisy = np.asarray(Image.open(tmp_path / "dbg" / f"crop{n_last}_{i_last}_noisy.jpg"))
    assert noisy.shape == (120, 120, 3)
    want = (otiler.gather_tile(frame, grid, grid.size - 1).transpose(1, 2, 0) * 255 + 0.5).clip(0, 255).astype(np.uint8)
    # JPEG of the mirrored input tile: the synthetic frame is pixel noise on a gradient (chroma subsampling smears the noise), so
    # compare 8x8 block means of the luminance
    def blocks(a):
        y = a.astype(np.float64) @ np.array([0.299, 0.587, 0.114])
        return y.reshape(15, 8, 15, 8).mean(axis=(1, 3))
    assert np.abs(blocks(noisy) - blocks(want)).max() < 4
    _, _, ud, _ = grid.geom(grid.size - 1)
    tens = Image.open(tmp_path / "dbg" / f"crop{n_last}_{i_last}_tensimg.jpg")
    assert tens.size == (ud[2] - ud[0], ud[3] - ud[1])
    assert os.path.isfile(str(tmp_path / "b.tiff") + "dbg_inclborders.tif")


@pytest.mark.parametrize("cs", [504, 520])
def test_utnet_f64_wide_tiles_vs_oracle(dev, cs):
    # the shipped default tile (cs=504, denoise_image.py:41) and BASELINE config 4's cs=520: rows too wide for three
    # LDS stage images, the conv kernel falls back to its 2-stage variants
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    sd = synth.make_utnet_state_dict(funit=64, seed=123)
    net = UtNet()
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    x = torch.rand(1, 3, cs, cs, generator=torch.Generator().manual_seed(cs))
    with torch.no_grad():
        ref = onet.utnet_forward(sd, x)
    assert_close(net(x.to(dev)), ref, f"f64 cs{cs}")
    # BASELINE configs 3 / 4 at these tile sizes: 16-bit storage, fp32 accumulate (1e-3 applies to fp32 only -> PSNR bars)
    # (bars a few dB under the measured values: 89.7 dB fp16 / 70.6 dB bf16 at cs = 264, peak = the reference's range)
    for dtype, min_psnr in (("f16", 85.0), ("bf16", 65.0)):
        y = net.set_compute_dtype(dtype)(x.to(dev)).float().cpu()
        mse = ((y - ref) ** 2).mean().item()
        psnr = 10 * np.log10((ref.max() - ref.min()).item() ** 2 / max(mse, 1e-30))
        print(f"UtNet(64) cs={cs} {dtype}: PSNR {psnr:.1f} dB, max abs err {(y - ref).abs().max().item():.3e}")
        assert torch.isfinite(y).all() and psnr >= min_psnr, (dtype, cs, psnr)
    net.set_compute_dtype("f32")


def test_utnet_non_square_and_whole_image(dev):
    # --whole_image branch (denoise_image.py:110-128): one item = the whole frame with a symmetric mirror border;
    # the network then sees a non-square input whose sides are each of the form 16k+56
    from nind_denoise_amd import denoise_image as di
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    sd = synth.make_utnet_state_dict(funit=16, seed=4)
    net = UtNet(funit=16)
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    x = torch.rand(2, 3, 104, 152, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = onet.utnet_forward(sd, x)
    assert_close(net(x.to(dev)), ref, "non-square 104x152")
    frame = synth.make_frame(136, 88, seed=5)              # W=136, H=88 -> padded by 8: 152 x 104
    ds = di.OneImageDS(frame, None, None, None, whole_image=True, pad=8, device=dev)
    assert len(ds) == 1
    t, ud, us = ds[0]
    assert tuple(t.shape) == (3, 104, 152) and tuple(ud.tolist()) == (8, 8, 144, 96) and tuple(us.tolist()) == (8, 8)
    from oracle import tiler as otiler
    want = otiler.whole_image_item(frame, 8)[0]       # sides mirrored, corners zero (pinned by tests/golden/whole_image.json)
    assert np.array_equal(t.cpu().numpy(), want)
    y = net(t[None])[0][:, ud[1]:ud[3], ud[0]:ud[2]]
    with torch.no_grad():
        ref = onet.utnet_forward(sd, torch.from_numpy(want)[None])[0][:, 8:96, 8:144]
    assert_close(y, ref, "whole image")


def test_unet_golden(dev, golden_dir):
    # BASELINE config 1 (one 256x256 RGB tile through UNet) on the HIP path + an odd size through the F.pad fix-up
    from nind_denoise_amd.networks.ThirdPartyNets import UNet
    d = np.load(os.path.join(golden_dir, "unet_256.npz"))
    sd = synth.make_unet_state_dict(seed=0)
    assert synth.state_dict_digest(sd) == str(d["sd_digest"])
    net = UNet()
    net.load_state_dict(sd, strict=True)
    net = net.eval().to(dev)
    assert_close(net(torch.from_numpy(d["x"]).to(dev)), torch.from_numpy(d["y"]), "unet 256")
    assert_close(net(torch.from_numpy(d["x2"]).to(dev)), torch.from_numpy(d["y2"]), "unet 100x92")
    with pytest.raises(RuntimeError):
        net.train()(torch.from_numpy(d["x"]).to(dev))


def test_unet_tiled_frame_vs_oracle(dev):
    # the generic (non-UtNet) path of the device loop: nd_tile_gather -> model -> nd_stitch_add with UNet
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.nn_common import Model
    from oracle import networks as onet
    from oracle import tiler as otiler
    sd = synth.make_unet_state_dict(seed=0)
    net = Model.instantiate_model(network="UNet", device=dev)
    net.load_state_dict(sd)
    net.eval()
    W, H, cs, ucs, ol = 200, 170, 96, 64, 8
    frame = synth.make_frame(W, H, seed=6)

    def model_fn(x):
        with torch.no_grad():
            return onet.unet_forward(sd, torch.from_numpy(x)).numpy()

    ref = otiler.denoise_frame(frame, cs, ucs, ol, model_fn, batch=4)
    out = pipeline.denoise_frame(net, torch.from_numpy(frame).to(dev), cs, ucs, ol, batch=5)
    assert_close(out, torch.from_numpy(ref), "unet frame")


# ---------------------------------------------------------------------------- bf16 / fp16 storage (BASELINE configs 3, 4)

@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_half_layers_exact_on_integer_data(dev, dtype):
    # small integers are exact in bf16 / fp16 and fp32 accumulation of their products is exact: the only rounding is
    # the final store, so the result must equal the 16-bit rounding of the exact answer bit for bit (pins the
    # 32x32x16 fragment maps, the 8-channel plane layout, tap flips and the half-plane epilogue stores)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    cin, cout, H, W = 32, 32, 12, 14
    x = torch.arange(cin * H * W, dtype=torch.float32).reshape(1, cin, H, W) % 7 - 3
    w = (torch.arange(cout * cin * 9, dtype=torch.float32).reshape(cout, cin, 3, 3) % 5) - 2
    b = torch.arange(cout, dtype=torch.float32) - 16
    y = layer_forward(dev, "conv3", x, w, b, "none", dtype=dtype)
    assert torch.equal(y.cpu(), F.conv2d(x, w, b).to(tdt).float())
    wt = (torch.arange(cout * cin * 9, dtype=torch.float32).reshape(cin, cout, 3, 3) % 5) - 2
    y = layer_forward(dev, "convT3", x, wt, b, "PReLU", 0.5, dtype=dtype)
    assert torch.equal(y.cpu(), F.prelu(F.conv_transpose2d(x, wt, b), torch.tensor([0.5])).to(tdt).float())
    wu = (torch.arange(cout * cin * 4, dtype=torch.float32).reshape(cin, cout, 2, 2) % 5) - 2
    y = layer_forward(dev, "convT2s2", x, wu, b, "none", dtype=dtype)
    assert torch.equal(y.cpu(), F.conv_transpose2d(x, wu, b, stride=2).to(tdt).float())
    x3 = torch.arange(3 * H * W, dtype=torch.float32).reshape(1, 3, H, W) % 5 - 2     # first layer: 3 -> one K block
    w3 = (torch.arange(cout * 3 * 9, dtype=torch.float32).reshape(cout, 3, 3, 3) % 7) - 3
    y = layer_forward(dev, "conv3", x3, w3, b, "none", dtype=dtype)
    assert torch.equal(y.cpu(), F.conv2d(x3, w3, b).to(tdt).float())


@pytest.mark.parametrize("dtype,min_psnr", [("bf16", 65.0), ("f16", 85.0)])   # measured 70.6 / 89.7 dB
def test_utnet_half_storage_vs_fp32_oracle(dev, golden_dir, dtype, min_psnr):
    # configs 3 / 4: 16-bit storage, fp32 accumulate.  The 1e-3 bar is for fp32 only; here parity is reported as
    # max-abs and PSNR against the fp32 reference output (peak = the reference's own range)
    from nind_denoise_amd.networks.UtNet import UtNet
    d = np.load(os.path.join(golden_dir, "utnet_f64_cs264.npz"))
    sd = synth.make_utnet_state_dict(funit=64, seed=123)
    net = UtNet()
    net.load_state_dict(sd)
    net = net.eval().to(dev).set_compute_dtype(dtype)
    y = net(torch.from_numpy(d["x"]).to(dev)).cpu().numpy()
    ref = d["y"]
    assert np.isfinite(y).all()
    err = np.abs(y - ref)
    peak = float(ref.max() - ref.min())
    psnr = 10 * np.log10(peak ** 2 / float(np.mean((y - ref) ** 2)))
    print(f"UtNet(64) cs=264 {dtype}: max abs err {err.max():.3e}, PSNR {psnr:.1f} dB (peak {peak:.3f})")
    assert psnr >= min_psnr, psnr
    # and the fp32 path is untouched by switching back
    net.set_compute_dtype("f32")
    assert_close(net(torch.from_numpy(d["x"]).to(dev)), torch.from_numpy(ref), "fp32 after half")


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_frame_half_storage_fused_equals_unfused(dev, dtype, whole_k):
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    net = UtNet(funit=16)
    net.load_state_dict(synth.make_utnet_state_dict(funit=16, seed=9))
    net = net.eval().to(dev).set_compute_dtype(dtype)
    img = torch.from_numpy(synth.make_frame(333, 290, seed=3)).to(dev)
    a_roi = pipeline.denoise_frame(net, img, 120, 88, 16, batch=5)
    net.useful_only = False        # bit-for-bit equality holds between the two WHOLE-tile paths (fused / forward + stitch)
    a = pipeline.denoise_frame(net, img, 120, 88, 16, batch=5)
    b = pipeline.denoise_frame(lambda x: net(x), img, 120, 88, 16, batch=3)
    assert torch.equal(a, b)
    assert (a - a_roi).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())
    net.set_compute_dtype("f32")
    c = pipeline.denoise_frame(net, img, 120, 88, 16, batch=5)
    rel = ((a - c).abs().max() / c.abs().max()).item()
    assert rel < (0.01 if dtype == "bf16" else 0.002), rel


def test_frame_engine_streams_frames_in_order(dev):
    # resident multi-frame engine: overlapped H2D / compute / D2H must not change a single bit and keeps the order
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.serve import FrameEngine
    net = UtNet(funit=16)
    net.load_state_dict(synth.make_utnet_state_dict(funit=16, seed=2))
    net = net.eval().to(dev)
    W, H, cs, ucs, ol = 300, 260, 120, 88, 16
    frames = [synth.make_frame(W, H, seed=s) for s in range(7)]
    eng = FrameEngine(net, W, H, cs, ucs, ol, batch=6, slots=3, device=dev)
    outs = list(eng.run(frames))
    assert len(outs) == len(frames)
    for f, o in zip(frames, outs):
        ref = pipeline.denoise_frame(net, torch.from_numpy(f).to(dev), cs, ucs, ol, batch=6).cpu().numpy()
        assert np.array_equal(o, ref)
    with pytest.raises(ValueError):
        eng.submit(np.zeros((3, 10, 10), dtype=np.float32))


def test_reference_style_main_loop_with_dataloader(dev, whole_k):
    # the reference's own loop shape (denoise_image.py:232-267): DataLoader over OneImageDS, model(ybatch), crop by
    # usefuldim, make_seamless_edges, canvas += in tile order -- driven here with this package's drop-in classes only;
    # must give the same bits as the fused device loop (row a4 / a8 of SURVEY.md section 8)
    from torch.utils.data import DataLoader
    from nind_denoise_amd import denoise_image as di
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    net = UtNet(funit=16)
    net.load_state_dict(synth.make_utnet_state_dict(funit=16, seed=12))
    net = net.eval().to(dev)
    W, H, cs, ucs, ol = 333, 290, 120, 88, 16
    frame = synth.make_frame(W, H, seed=4)
    ds = di.OneImageDS(frame, cs, ucs, ol, device=dev)
    loader = DataLoader(dataset=ds, num_workers=0, drop_last=False, batch_size=4, shuffle=False)
    newimg = torch.zeros(3, H, W, dtype=torch.float32, device=dev)

    def make_seamless_edges(tcrop, x0, y0):
        if x0 != 0:
            tcrop[:, :, 0:ol] = tcrop[:, :, 0:ol].div(2)
        if y0 != 0:
            tcrop[:, 0:ol, :] = tcrop[:, 0:ol, :].div(2)
        if x0 + ucs < W and ol:
            tcrop[:, :, -ol:] = tcrop[:, :, -ol:].div(2)
        if y0 + ucs < H and ol:
            tcrop[:, -ol:, :] = tcrop[:, -ol:, :].div(2)
        return tcrop

    n = 0
    for ybatch, usefuldims, usefulstarts in loader:
        assert ybatch.is_cuda and usefuldims.dtype == torch.int32
        xbatch = net(ybatch)
        for i in range(ybatch.size(0)):
            ud = usefuldims[i]
            t = xbatch[i][:, ud[1]:ud[3], ud[0]:ud[2]].clone()
            ax, ay = tuple(usefulstarts[i].tolist())
            t = make_seamless_edges(t, ax, ay)
            newimg[:, ay:ay + t.shape[1], ax:ax + t.shape[2]] += t
            n += 1
    assert n == len(ds)
    net.useful_only = False        # the reference loop calls forward(): whole tiles; the fused loop matches it bit for bit in that mode
    fused = pipeline.denoise_frame(net, torch.from_numpy(frame).to(dev), cs, ucs, ol, batch=7)
    assert torch.equal(newimg, fused)
    net.useful_only = True         # default: the last decoder levels restricted to what the useful centres need -- same canvas
    fused_roi = pipeline.denoise_frame(net, torch.from_numpy(frame).to(dev), cs, ucs, ol, batch=7)
    assert (fused_roi - fused).abs().max().item() <= 2e-6 * max(1.0, fused.abs().max().item())


def test_api_error_paths(dev):
    lib = _lib.load()
    from nind_denoise_amd import pipeline
    img = torch.zeros(3, 300, 300, device=dev)
    with pytest.raises(ValueError):
        pipeline.gather_tiles(img, 264, 200, 64, 3, 5)          # tiles outside the grid
    assert pipeline.gather_tiles(img, 264, 200, 64, 0, 0).shape[0] == 0   # empty range is a no-op
    with pytest.raises(ValueError):
        pipeline.tile_count(100, 100, 264, 200, 64)            # mirror padding would reach outside the frame
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    blob = torch.zeros(1 << 10, device=dev)
    x = torch.zeros(1, 3, 104, 104, device=dev)
    rc = lib.nd_utnet_forward(16, 1, 0, 0, blob.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 104, ws.data_ptr(), ws.numel(),
                              _lib.stream_ptr(dev))
    assert rc == -2 and b"workspace" in lib.nd_last_error()      # ND_ENOMEM: workspace too small, nothing launched
    rc = lib.nd_utnet_forward(12, 1, 0, 0, blob.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 104, ws.data_ptr(), ws.numel(),
                              _lib.stream_ptr(dev))
    assert rc == -1                                              # funit not a multiple of 8
    rc = lib.nd_utnet_forward(16, 7, 0, 0, blob.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 104, ws.data_ptr(), ws.numel(),
                              _lib.stream_ptr(dev))
    assert rc == -1                                              # unknown activation
    rc = lib.nd_utnet_forward(16, 1, 0, 64, blob.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 104, ws.data_ptr(), ws.numel(),
                              _lib.stream_ptr(dev))
    assert rc == -1 and b"flag" in lib.nd_last_error()           # unknown flag bit
    assert lib.nd_utnet_workspace_bytes(16, 104, 1, 1) > 0 and lib.nd_utnet_workspace_bytes(8, 104, 1, 1) == 0   # bf16 needs funit%16
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------- training step building blocks

def test_backward_data_through_the_forward_kernels(dev):
    # data gradients of the three layer types are forward launches of the SAME conv kernel with the weight tensor
    # re-read in its transposed role (no copy): dgrad(Conv2d) = ConvTranspose2d, dgrad(ConvTranspose2d) = Conv2d,
    # dgrad(ConvTranspose2d(2, s=2)) = Conv2d(2, stride=2).  Checked against torch autograd.
    zero = lambda n: torch.zeros(n)
    x = rnd((2, 24, 20, 18), 1).requires_grad_()
    w = rnd((40, 24, 3, 3), 2, 0.1)
    dy = rnd((2, 40, 18, 16), 3)
    F.conv2d(x, w).backward(dy)
    assert_close(layer_forward(dev, "convT3", dy, w, zero(24)), x.grad, "dgrad conv3")

    x = rnd((2, 16, 11, 13), 4).requires_grad_()
    wt = rnd((16, 32, 3, 3), 5, 0.1)
    dy = rnd((2, 32, 13, 15), 6)
    F.conv_transpose2d(x, wt).backward(dy)
    assert_close(layer_forward(dev, "conv3", dy, wt, zero(16)), x.grad, "dgrad convT3")

    for (ci, co, h, w_) in ((64, 32, 13, 13), (16, 8, 9, 7), (128, 64, 30, 26)):
        x = rnd((2, ci, h, w_), 7).requires_grad_()
        wu = rnd((ci, co, 2, 2), 8, 0.2)
        dy = rnd((2, co, 2 * h, 2 * w_), 9)
        F.conv_transpose2d(x, wu, stride=2).backward(dy)
        assert_close(layer_forward(dev, "conv2s2", dy, wu, zero(ci)), x.grad, f"dgrad convT2s2 {ci}->{co}")
    # exact-integer pin of the stride-2 tap / lane maps
    xi = (torch.arange(16 * 12 * 10, dtype=torch.float32).reshape(1, 16, 12, 10) % 7) - 3
    wi = (torch.arange(8 * 16 * 4, dtype=torch.float32).reshape(8, 16, 2, 2) % 5) - 2
    y = layer_forward(dev, "conv2s2", xi, wi, torch.arange(8, dtype=torch.float32))
    assert torch.equal(y.cpu(), F.conv2d(xi, wi, torch.arange(8, dtype=torch.float32), stride=2))


WGRAD_CASES = [
    # kind, B, cin, cout, H, W
    ("conv3", 2, 24, 40, 20, 18),
    ("conv3", 3, 64, 64, 34, 30),
    ("conv3", 2, 3, 64, 40, 36),          # first layer: 3 input channels
    ("conv3", 1, 128, 72, 13, 13),
    ("convT3", 2, 16, 32, 11, 13),
    ("convT3", 1, 128, 64, 26, 22),
    ("convT2s2", 2, 64, 32, 13, 13),
    ("convT2s2", 1, 16, 8, 9, 7),
    ("conv1", 2, 64, 4, 21, 23),
    ("conv3", 4, 64, 64, 138, 138),       # many K slices (training-size rows)
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_weight_gradient_vs_autograd(dev, case):
    kind, B, cin, cout, H, W = case
    lib = _lib.load()
    k = {"conv3": 3, "convT3": 3, "convT2s2": 2, "conv1": 1}[kind]
    x = rnd((B, cin, H, W), 1)
    wshape = (cout, cin, k, k) if kind in ("conv3", "conv1") else (cin, cout, k, k)
    w = rnd(wshape, 2, 0.1).requires_grad_()
    b = rnd((cout,), 3, 0.1).requires_grad_()
    if kind in ("conv3", "conv1"):
        y = F.conv2d(x, w, b)
    else:
        y = F.conv_transpose2d(x, w, b, stride=2 if kind == "convT2s2" else 1)
    dy = rnd(tuple(y.shape), 4)
    y.backward(dy)
    wsb = lib.nd_layer_wgrad_workspace_bytes(_lib.KIND[kind], B, cin, cout, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dw = torch.full(wshape, float("nan"), device=dev)
    db = torch.full((cout,), float("nan"), device=dev)
    xd, dyd = x.to(dev).contiguous(), dy.to(dev).contiguous()
    _lib.check(lib.nd_layer_wgrad(_lib.KIND[kind], xd.data_ptr(), dyd.data_ptr(), B, cin, H, W, cout, dw.data_ptr(),
                                  db.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
    # sums of up to ~1e5 products of O(1) values: compare relative to the gradient's own scale
    for got, ref, what in ((dw, w.grad, "dW"), (db, b.grad, "db")):
        got = got.cpu()
        assert torch.isfinite(got).all(), what
        err = (got - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1.0) + 1e-5, (case, what, err, ref.abs().max().item())


def _autograd_reference(sd, x, t, w_l1, w_mse, dtype=torch.float32, w_ssim=0.0, w_msssim=0.0, loss_cs=None):
    from oracle import losses as olosses
    from oracle import networks as onet
    params = {k: v.clone().to(dtype).requires_grad_() for k, v in sd.items()}
    x, t = x.to(dtype), t.to(dtype)
    y = onet.utnet_forward(params, x)
    g = y.clip(0, 1)
    if loss_cs is not None:     # pt_ops.pt_crop_batch (nn_train.py:319-323): the criteria see the centre crop only
        o = (g.shape[2] - loss_cs) // 2
        g, t = g[:, :, o:o + loss_cs, o:o + loss_cs], t[:, :, o:o + loss_cs, o:o + loss_cs]
    loss = w_l1 * F.l1_loss(g, t) + w_mse * F.mse_loss(g, t)
    if w_ssim:
        loss = loss + w_ssim * (1 - olosses.ssim(g, t)).mean()
    if w_msssim:
        loss = loss + w_msssim * (1 - olosses.ms_ssim(g, t)).mean()
    loss.backward()
    return y.detach(), loss.detach(), params


@pytest.mark.parametrize("funit,cs,B,w_l1,w_mse", [(8, 104, 2, 0.0, 1.0), (16, 120, 3, 0.3, 0.7), (64, 136, 2, 0.5, 0.5)])
def test_training_step_gradients_vs_autograd(dev, funit, cs, B, w_l1, w_mse):
    # BASELINE config 5 building block: forward + loss + backward against torch autograd on the oracle (CPU)
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.train import UtNetTrainer
    sd = synth.make_utnet_state_dict(funit=funit, seed=31, gain=1.8)
    net = UtNet(funit=funit)
    net.load_state_dict(sd)
    tr = UtNetTrainer(net, device=dev, weights={"L1": w_l1, "MSE": w_mse})
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, cs, cs, generator=g)
    t = (x * 0.9 + 0.05 * torch.rand(B, 3, cs, cs, generator=g)).clip(0, 1)
    y, loss = tr.forward_backward(x, t)
    torch.cuda.synchronize()
    # the production width: 23 layers of up to 9216-term fp32 sums and gradients spanning 1e-1 .. 1e-7 (these synthetic
    # weights make them vanish towards the bottom).  There fp32 autograd itself is 3e-4 .. 5e-4 off float64 autograd on the
    # smallest tensors, and the MFMA's sequential fmaf chains ~3-10x that; so that case is checked against float64 with a bar
    # tied to torch's own fp32 error
    wide = funit == 64
    y_ref, loss_ref, params = _autograd_reference(sd, x, t, w_l1, w_mse, torch.float64 if wide else torch.float32)
    params32 = _autograd_reference(sd, x, t, w_l1, w_mse)[2] if wide else None
    y_ref, loss_ref = y_ref.float(), loss_ref.float()
    assert_close(y, y_ref, "training forward")
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    worst = 0.0
    for name, p in params.items():
        got = tr.grad_of(name).cpu()
        ref = p.grad.float()
        scale = max(ref.abs().max().item(), 1e-8)
        err = (got - ref).abs().max().item() / scale
        worst = max(worst, err)
        # fixed bars, ~3x / ~1.5x what was measured (narrow nets vs fp32 autograd: 8e-6 .. 9.4e-5 with the fused 1-D Winograd
        # kernel in the forward and data-gradient passes; funit 64 vs float64: 3.5e-3)
        bar = 5e-3 if wide else 3e-4
        assert torch.isfinite(got).all() and err <= bar, (name, err, bar, scale)
    print(f"training step f{funit} cs{cs}: worst relative gradient error {worst:.2e}")


def test_autograd_training_matches_fused_step(dev):
    """Row f3 behind plain autograd: the reference's own statements (nn_common.py:198-218) -- model(x).clip(0, 1), a torch
    criterion, loss.backward(), torch.optim.Adam(amsgrad) -- on the module, three updates, against UtNetTrainer (fused step +
    nd_adam_step) from the same start: same outputs, same losses, same parameters."""
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.train import UtNetTrainer
    funit, cs, B, lr = 8, 104, 2, 3e-3
    sd = synth.make_utnet_state_dict(funit=funit, seed=31, gain=1.8)
    ref_net = UtNet(funit=funit)
    ref_net.load_state_dict(sd)
    tr = UtNetTrainer(ref_net, device=dev, lr=lr, beta1=0.75, weights={"L1": 0.0, "MSE": 1.0})
    net = UtNet(funit=funit)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=lr, betas=(0.75, 0.999), amsgrad=True)
    g = torch.Generator().manual_seed(3)
    for step in range(3):
        x = torch.rand(B, 3, cs, cs, generator=g)
        t = (x * 0.9 + 0.05 * torch.rand(B, 3, cs, cs, generator=g)).clip(0, 1)
        y_ref, loss_ref = tr.forward_backward(x, t)
        loss_ref = loss_ref.item()
        tr.optimizer_step()
        opt.zero_grad()
        out = net(x.to(dev))
        assert out.requires_grad
        loss = F.mse_loss(out.clip(0, 1), t.to(dev))
        loss.backward()
        opt.step()
        if step == 0:
            assert torch.equal(out.detach(), y_ref)            # the same kernels on the same weights
        else:                                                  # (torch's Adam and nd_adam_step round differently in the last bits)
            assert (out.detach() - y_ref).abs().max().item() <= 1e-5, step
        assert abs(loss.item() - loss_ref) <= 1e-6 * max(1.0, abs(loss_ref)), (step, loss.item(), loss_ref)
    worst = 0.0
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref_net.named_parameters()):
        assert n1 == n2
        worst = max(worst, (p1.detach() - p2.detach()).abs().max().item())
    assert worst <= 1e-6, worst
    # the updated weights are what inference sees (torch bumped the parameters' versions: the packed blob is rebuilt)
    net.eval()
    with torch.no_grad():
        xi = torch.rand(1, 3, cs, cs, generator=g).to(dev)
        assert_close(net(xi), ref_net.eval()(xi), "inference after autograd training")
    # one graph per forward: a second forward invalidates the first one's saved activations, loudly
    net.train()
    o1 = net(xi)
    net(xi)
    with pytest.raises(RuntimeError, match="another forward"):
        o1.sum().backward()


@pytest.mark.parametrize("activation", ["ELU", "Hardswish", "PReLU"])
def test_autograd_gradients_all_activations(dev, activation):
    """loss.backward() through the module for the three activations the reference constructor takes (networks/UtNet.py:17-26)
    against torch autograd on the oracle (CPU)."""
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    funit, cs, B = 8, 104, 2
    # (seeds and the smooth MSE criterion of test_training_step_gradients_vs_autograd's first case: with an L1 term the sign of
    #  g - t flips on pixels where the two forwards differ in the last bits, and the vanishing gradients of the deep layers of
    #  these synthetic weights -- 1e-6 .. 1e-5 -- move by more than the bar)
    sd = synth.make_utnet_state_dict(funit=funit, seed=31, activation=activation, gain=1.8)
    net = UtNet(funit=funit, activation=activation)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, cs, cs, generator=g)
    t = (x * 0.9 + 0.05 * torch.rand(B, 3, cs, cs, generator=g)).clip(0, 1)
    out = net(x.to(dev))
    loss = F.mse_loss(out.clip(0, 1), t.to(dev))
    loss.backward()
    params = {k: v.clone().requires_grad_() for k, v in sd.items()}
    y = onet.utnet_forward(params, x, activation=activation)
    lref = F.mse_loss(y.clip(0, 1), t)
    lref.backward()
    assert_close(out.detach(), y.detach(), f"autograd forward {activation}")
    assert abs(loss.item() - lref.item()) <= 1e-5 * max(1.0, abs(lref.item()))
    worst = 0.0
    for name, p in net.named_parameters():
        ref = params[name].grad
        scale = max(ref.abs().max().item(), 1e-8)
        err = (p.grad.cpu() - ref).abs().max().item() / scale
        worst = max(worst, err)
        assert torch.isfinite(p.grad).all() and err <= 1e-3, (activation, name, err, scale)   # (the fused step's own bar: 3e-4)
    print(f"autograd {activation}: worst relative gradient error {worst:.2e}")


def _ddp_worker(port, outq):
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.train import UtNetTrainer
    d = torch.device("cuda", 0)
    torch.cuda.set_device(d)
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
    try:
        res = {}
        for overlapped in (False, True):
            net = UtNet(funit=8)
            net.load_state_dict(synth.make_utnet_state_dict(funit=8, seed=31, gain=1.8))
            tr = UtNetTrainer(net, device=d, weights={"L1": 0.2, "MSE": 0.8})
            tr.averager.active = overlapped      # world size 1: force the bucket-by-bucket RCCL path (sum over one rank, / 1)
            g = torch.Generator().manual_seed(3)
            x = torch.rand(2, 3, 104, 104, generator=g)
            t = (x * 0.9 + 0.05 * torch.rand(2, 3, 104, 104, generator=g)).clip(0, 1)
            for _ in range(2):
                tr.learn(x, t)
            torch.cuda.synchronize()
            res[overlapped] = (tr.flat.cpu().numpy(), tr.grads.cpu().numpy())
        outq.put(res)
    finally:
        tdist.destroy_process_group()


def test_bucketed_gradient_reduce_over_rccl(dev):
    """BASELINE configs[4] building block: the training step records one event per level bucket while the backward pass runs
    (nd_utnet_train_step_ev) and the reducer all-reduces each bucket behind its event on a side stream (RCCL, world size 1 here:
    the transport and the event / stream ordering run, the sum is over one rank): two updates give the same parameters and
    gradients as without the reducer, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_ddp_worker, args=(port, q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert np.array_equal(res[False][0], res[True][0]) and np.array_equal(res[False][1], res[True][1])
    assert np.isfinite(res[True][1]).all() and np.abs(res[True][1]).max() > 0


def test_training_step_w2d_forward_with_preactivation_copy(dev):
    # From 512 workgroup tiles up the training forward runs a 3x3 layer through conv_w2d, which then writes the pre-activation copy
    # itself (9 crops of 184 pixels: 595 tiles on the first level), and every data gradient may take that kernel: the whole step
    # must agree with the conv_w1d path (flag ND_FLAG_W1D_REGS), which the autograd tests above pin.
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.train import UtNetTrainer
    funit, cs, B = 8, 184, 9
    sd = synth.make_utnet_state_dict(funit=funit, seed=17, gain=1.8)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, cs, cs, generator=g)
    t = torch.rand(B, 3, cs, cs, generator=g)
    res = {}
    for regs in (False, True):
        net = UtNet(funit=funit)
        net.load_state_dict(sd)
        net.w1d_regs = regs
        tr = UtNetTrainer(net, device=dev, weights={"L1": 0.4, "MSE": 0.6})
        y, loss = tr.forward_backward(x, t)
        torch.cuda.synchronize()
        res[regs] = ({k: tr.grad_of(k).cpu().clone() for k, _ in net.named_parameters()}, y.cpu().clone(), loss.item())
    assert abs(res[False][2] - res[True][2]) <= 1e-6 * max(1.0, abs(res[True][2]))
    assert (res[False][1] - res[True][1]).abs().max().item() <= 3e-6
    for k, ref in res[True][0].items():
        got = res[False][0][k]
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() <= 5e-5 * max(ref.abs().max().item(), 1e-12), k


@pytest.mark.parametrize("cs,weights", [(120, {"SSIM": 1.0}), (168, {"MSSSIM": 1.0}),
                                        (184, {"L1": 0.2, "MSE": 0.2, "SSIM": 0.2, "MSSSIM": 0.4})])
def test_training_step_ssim_losses_vs_autograd(dev, cs, weights):
    # the reference's default generator loss is MS-SSIM (weights {'MSSSIM': 1}): network gradients through the HIP SSIM /
    # MS-SSIM backward against torch autograd through the oracle network + oracle scores
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.train import UtNetTrainer
    funit, B = 8, 2
    sd = synth.make_utnet_state_dict(funit=funit, seed=17, gain=1.8)
    net = UtNet(funit=funit)
    net.load_state_dict(sd)
    tr = UtNetTrainer(net, device=dev, weights=weights)
    g = torch.Generator().manual_seed(5)
    # image-like input; the target is the (random-weight) network's own output plus noise, so that generated and target
    # batches are correlated: on unrelated images some coarse-scale cs is <= 0 and MS-SSIM's relu kills the whole gradient
    from oracle import networks as onet
    x = F.interpolate(torch.rand(B, 3, cs // 8, cs // 8, generator=g), size=(cs, cs), mode="bilinear", align_corners=False)
    x = (0.8 * x + 0.1 + 0.05 * torch.randn(B, 3, cs, cs, generator=g)).clip(0, 1)
    with torch.no_grad():
        y0 = onet.utnet_forward(sd, x)
    y0 = (y0 - y0.mean()) / (4 * y0.std()) + 0.5            # bring the output into [0, 1] territory: scale the last layer
    sd = dict(sd)
    with torch.no_grad():
        k = 1.0 / (4 * onet.utnet_forward(sd, x).std())
        sd["tconvs4.4.bias"] = (sd["tconvs4.4.bias"] - onet.utnet_forward(sd, x).mean()) * k + 0.5
        sd["tconvs4.4.weight"] = sd["tconvs4.4.weight"] * k
    net.load_state_dict(sd)
    tr = UtNetTrainer(net, device=dev, weights=weights)
    # (+0.02: a brightness offset, else the last bias' gradient -- the response to a global shift -- hinges on the sample
    # mean of the noise and is ill-conditioned against 1e-6 differences of the two forward passes)
    t = (y0 + 0.02 + 0.03 * torch.randn(B, 3, cs, cs, generator=g)).clip(0, 1)
    y, loss = tr.forward_backward(x, t)
    torch.cuda.synchronize()
    kw = dict(w_ssim=weights.get("SSIM", 0.0), w_msssim=weights.get("MSSSIM", 0.0))
    y_ref, loss_ref, params = _autograd_reference(sd, x, t, weights.get("L1", 0.0), weights.get("MSE", 0.0), torch.float64, **kw)
    params32 = _autograd_reference(sd, x, t, weights.get("L1", 0.0), weights.get("MSE", 0.0), **kw)[2]
    assert_close(y, y_ref.float(), "training forward")
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    worst, worst_of = 0.0, (0.0, 0.0, "")
    gmax = max(p.grad.abs().max().item() for p in params.values())
    assert gmax > 1e-4                                                        # a real gradient, not the relu-dead case
    for name, p in params.items():
        got, ref = tr.grad_of(name).cpu(), p.grad.float()
        scale = max(ref.abs().max().item(), 1e-8)
        err = (got - ref).abs().max().item() / scale
        worst = max(worst, err)
        # reference = float64 autograd, fixed bars at ~2x what was measured (mixed loss 7e-4, MS-SSIM alone 7.5e-3: the SSIM
        # gradient oscillates in sign from pixel to pixel, so whole-image sums -- PReLU slopes, biases -- keep few digits in
        # fp32; fp32 autograd itself is params32 away from float64 on the same tensors.  Measured on the worst parameter of
        # the MS-SSIM case, the PReLU slope tconvs4.3.weight: HIP 6.2e-3, torch fp32 autograd 6.3e-3 -- the error is the
        # conditioning of the sum in fp32, not the order of the five-scale product in nd_ssim_loss_grad)
        bar = 1.5e-2 if set(weights) == {"MSSSIM"} else 2e-3
        err32 = (params32[name].grad - ref).abs().max().item() / scale       # torch's own fp32 autograd against float64
        if err > worst_of[0]:
            worst_of = (err, err32, name)
        assert torch.isfinite(got).all() and err <= bar, (name, err, bar, scale, err32)
    print(f"training step {weights} cs{cs}: worst relative gradient error {worst:.2e} ({worst_of[2]}; torch fp32 autograd on the "
          f"same parameter: {worst_of[1]:.2e})")
    if "MSSSIM" in weights:
        with pytest.raises(ValueError, match="161"):     # the 128 / 136-pixel crops of BASELINE config 5 cannot use MS-SSIM
            tr.forward_backward(x[..., :136, :136].contiguous(), t[..., :136, :136].contiguous())


def test_adam_amsgrad_two_steps_vs_torch(dev):
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.train import UtNetTrainer
    from oracle import networks as onet
    funit, cs, B = 8, 104, 2
    sd = synth.make_utnet_state_dict(funit=funit, seed=5, gain=1.8)
    net = UtNet(funit=funit)
    net.load_state_dict(sd)
    tr = UtNetTrainer(net, lr=1e-3, beta1=0.75, device=dev, weights={"L1": 1.0, "MSE": 0.0})
    params = {k: v.clone().requires_grad_() for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, betas=(0.75, 0.999), amsgrad=True)
    g = torch.Generator().manual_seed(9)
    for step in range(2):
        x = torch.rand(B, 3, cs, cs, generator=g)
        t = torch.rand(B, 3, cs, cs, generator=g)
        loss = tr.learn(x, t)
        opt.zero_grad()
        ref = F.l1_loss(onet.utnet_forward(params, x).clip(0, 1), t)
        ref.backward()
        opt.step()
        assert abs(loss.item() - ref.item()) <= 1e-4 * max(1.0, abs(ref.item())), (step, loss.item(), ref.item())
    torch.cuda.synchronize()
    new = dict(net.named_parameters())
    for name, p in params.items():
        # Adam's sign-like first steps amplify tiny gradient differences where the gradient is ~0: compare updates
        upd_ref = (p.detach() - sd[name]).abs().max().item()
        err = (new[name].detach().cpu() - p.detach()).abs().max().item()
        assert err <= 0.05 * max(upd_ref, 1e-6) + 2e-6, (name, err, upd_ref)


def test_forward_after_optimizer_step_uses_new_weights(dev):
    # nd_adam_step rewrites the parameters through raw pointers (torch's version counters do not move): the inference blob the
    # module packed before the step must not be reused (validation between epochs, nn_train.py, runs the same model object)
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.train import UtNetTrainer
    from oracle import networks as onet
    funit, cs, B = 8, 104, 2
    sd = synth.make_utnet_state_dict(funit=funit, seed=5, gain=1.8)
    net = UtNet(funit=funit)
    net.load_state_dict(sd)
    tr = UtNetTrainer(net, lr=1e-2, beta1=0.75, device=dev, weights={"L1": 1.0, "MSE": 0.0})
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 3, cs, cs, generator=g)
    t = torch.rand(B, 3, cs, cs, generator=g)
    y_before = tr.model(x.to(dev)).clone()          # primes the packed-weights cache
    tr.learn(x, t)
    y_after = tr.model(x.to(dev))
    new_sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    with torch.no_grad():
        ref = onet.utnet_forward(new_sd, x)
    assert_close(y_after, ref, "forward after learn()")
    assert (y_after - y_before).abs().max().item() > 1e-4       # the step really moved the output


@pytest.mark.parametrize("weights,cs,loss_cs", [({"L1": 0.5, "MSE": 0.5}, 120, 88), ({"SSIM": 1.0}, 136, 101),
                                                ({"MSE": 1.0}, 104, 104)])
def test_training_step_loss_center_crop(dev, weights, cs, loss_cs):
    # nn_train.py:319-323 computes the loss on pt_crop_batch(., loss_cs) (train_conf_defaults.yaml: loss_cs 161 < cs): the
    # gradient with respect to the output is zero outside the centre crop
    from nind_denoise_amd.networks.UtNet import UtNet
    from nind_denoise_amd.train import UtNetTrainer
    funit, B = 8, 2
    sd = synth.make_utnet_state_dict(funit=funit, seed=19, gain=1.8)
    net = UtNet(funit=funit)
    net.load_state_dict(sd)
    tr = UtNetTrainer(net, device=dev, weights=weights, loss_cs=loss_cs)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, cs, cs, generator=g)
    t = (x * 0.9 + 0.05 * torch.rand(B, 3, cs, cs, generator=g)).clip(0, 1)
    y, loss = tr.forward_backward(x, t)
    torch.cuda.synchronize()
    kw = dict(w_ssim=weights.get("SSIM", 0.0), w_msssim=weights.get("MSSSIM", 0.0), loss_cs=loss_cs)
    y_ref, loss_ref, params = _autograd_reference(sd, x, t, weights.get("L1", 0.0), weights.get("MSE", 0.0), torch.float64, **kw)
    assert_close(y, y_ref.float(), "training forward")
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    for name, p in params.items():
        got, ref = tr.grad_of(name).cpu(), p.grad.float()
        scale = max(ref.abs().max().item(), 1e-8)
        err = (got - ref).abs().max().item() / scale
        assert torch.isfinite(got).all() and err <= 2e-3, (name, err, scale)
    with pytest.raises(ValueError):
        UtNetTrainer(net, device=dev, weights=weights, loss_cs=cs + 8).forward_backward(x, t)


# ---------------------------------------------------------------------------- the benchmark's own launch shapes

def _psnr(y, ref):
    return 10 * np.log10(float(ref.max() - ref.min()) ** 2 / max(float(np.mean((y - ref) ** 2)), 1e-30))


def test_bench_frame_g24_fp32_launch_shapes_vs_oracle(dev):
    """bench.py's timed configuration (BASELINE configs[1], G24): one 6000x4000 frame, UtNet(64) fp32, cs=264, 256 tiles per
    conv-stack launch -- three-pass Winograd over a 256-tile chunk, split-K tail planning at that size, long plane offsets
    (256 x 178 MB of activations) -- and the frame's last launch of 252 tiles (1276 = 4 x 256 + 252) in the 256-tile workspace.
    The fused loop bench.py times == gather -> nd_utnet_forward -> stitch of the same five launches; sampled tiles of the first
    and the last launch (first / last of the launch, some past index 128) against the oracle."""
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    W, H, cs, ucs, ol, batch = 6000, 4000, 264, 200, 64, 256
    sd = synth.make_utnet_state_dict(funit=64, seed=123)
    net = UtNet()
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    img = torch.from_numpy(synth.make_frame(W, H, seed=24)).to(dev)
    total = pipeline.tile_count(W, H, cs, ucs, ol)
    assert total == 1276
    fused = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch)       # exactly bench.py's step
    unfused = torch.zeros_like(img)
    worst = 0.0
    for first in range(0, total, batch):
        count = min(batch, total - first)
        x = pipeline.gather_tiles(img, cs, ucs, ol, first, count)
        y = net(x)
        assert torch.isfinite(y).all()
        pipeline.stitch_tiles(unfused, y, cs, ucs, ol, first)
        if first == 0 or first + count == total:
            picks = sorted({0, 1, 129, count // 2 + 3, count - 2, count - 1})
            with torch.no_grad():
                ref = onet.utnet_forward(sd, x[picks].cpu())
            for k, i in enumerate(picks):
                worst = max(worst, assert_close(y[i:i + 1], ref[k:k + 1], f"G24 launch [{first},{first + count}) tile {i}"))
        del x, y
    print(f"G24 frame, 256 tiles per launch: worst max abs err over the sampled tiles {worst:.3e}")
    err = (fused - unfused).abs().max().item()
    scale = unfused.abs().max().item()
    # same tiles, same kernels; the last launch runs 252 tiles in a 256-tile workspace on the fused side (split-K tail may differ)
    assert err <= 1e-5 and scale > 0.01, (err, scale)


def test_bench_launch_shape_g61_fp16_vs_oracle(dev):
    """BASELINE configs[3] (G61): cs=520 / ucs=456 / ol=64, fp16 storage, several tiles per launch (the wide-row 2-stage conv
    variants with a multi-image batch, tiles gathered from a frame with mirrored edges) against the fp32 oracle."""
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    from oracle import networks as onet
    W, H, cs, ucs, ol = 2400, 1500, 520, 456, 64
    sd = synth.make_utnet_state_dict(funit=64, seed=123)
    net = UtNet()
    net.load_state_dict(sd)
    net = net.eval().to(dev)
    img = torch.from_numpy(synth.make_frame(W, H, seed=61)).to(dev)
    total = pipeline.tile_count(W, H, cs, ucs, ol)
    n = 10
    assert total >= n
    x = pipeline.gather_tiles(img, cs, ucs, ol, total - n, n)        # the frame's last tiles: right / bottom mirrors
    picks = [0, 4, n - 1]
    with torch.no_grad():
        ref = onet.utnet_forward(sd, x[picks].cpu()).numpy()
    y32 = net(x)
    for k, i in enumerate(picks):
        assert_close(y32[i:i + 1], torch.from_numpy(ref[k:k + 1]), f"cs520 fp32 tile {i} of a {n}-tile launch")
    for dtype, bar in (("f16", 85.0), ("bf16", 65.0)):
        y = net.set_compute_dtype(dtype)(x).float().cpu().numpy()
        assert np.isfinite(y).all()
        worst = min(_psnr(y[i], ref[k]) for k, i in enumerate(picks))
        print(f"G61 geometry, {n} tiles per launch, {dtype}: worst PSNR over sampled tiles {worst:.1f} dB")
        assert worst >= bar, (dtype, worst)
    # fused loop in fp16 == unfused loop, and both within the 16-bit bar of the fp32 canvas
    net.set_compute_dtype("f16")
    a = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=n)
    net.set_compute_dtype("f32")
    c = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=n)
    rel = ((a - c).abs().max() / c.abs().max()).item()
    assert rel < 0.002, rel


# ---------------------------------------------------------------------------- multi-GPU entry with the HIP loop as compute

def _sharded_worker(rank, world, port, geom, funit, outq):
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from nind_denoise_amd import dist as ndist
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    # one GPU per rank over RCCL when there are enough GPUs; else the ranks share GPU 0 and talk over gloo (RCCL refuses two
    # ranks on one device): the same exchange logic and the same device loop, host-staged messages
    shared = torch.cuda.device_count() < world
    d = torch.device("cuda", 0 if shared else rank)
    torch.cuda.set_device(d)
    if shared:
        tdist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=d)
    try:
        W, H, cs, ucs, ol, seed = geom
        net = UtNet(funit=funit)
        net.load_state_dict(synth.make_utnet_state_dict(funit=funit, seed=9))
        net = net.eval().to(d)
        net.split_k = False
        geo = ndist.Geo(W, H, cs, ucs, ol)
        frame = torch.from_numpy(synth.make_frame(W, H, seed=seed)).to(d) if rank == 0 else \
            torch.full((3, H, W), float("nan"), device=d)
        canvas = torch.full((3, H, W), 7.0, device=d)

        def compute(fr, cv, lo, hi):
            pipeline.denoise_frame(net, fr, cs, ucs, ol, batch=7, tile_range=(lo, hi), canvas=cv)

        ndist.denoise_frame_sharded(compute, frame, canvas, geo)
        torch.cuda.synchronize()
        if rank == 0:
            outq.put(canvas.cpu().numpy())
    finally:
        tdist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_frame_with_hip_compute(dev, world):
    """dist.denoise_frame_sharded with the device loop as the per-rank compute: world size 1 over RCCL (backend "nccl"); world
    size 2 over RCCL when two GPUs are visible, else both ranks on GPU 0 over gloo (host-staged messages).  Result == the
    single-GPU canvas (seam rows re-associated: <= 1 ulp)."""
    import socket
    import torch.multiprocessing as mp
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    geom = (500, 430, 120, 88, 16, 3)
    funit = 16
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, geom, funit, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    W, H, cs, ucs, ol, seed = geom
    net = UtNet(funit=funit)
    net.load_state_dict(synth.make_utnet_state_dict(funit=funit, seed=9))
    net = net.eval().to(dev)
    net.split_k = False
    ref = pipeline.denoise_frame(net, torch.from_numpy(synth.make_frame(W, H, seed=seed)).to(dev), cs, ucs, ol, batch=7).cpu().numpy()
    assert np.isfinite(got).all()
    if world == 1:
        assert np.array_equal(got, ref)
    else:
        assert np.abs(got - ref).max() <= 1e-6 and (got == ref).mean() > 0.7


def _stream_worker(rank, world, port, geom, funit, n_frames, outq):
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from nind_denoise_amd import dist as ndist
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    shared = torch.cuda.device_count() < world
    d = torch.device("cuda", 0 if shared else rank)
    torch.cuda.set_device(d)
    if shared:
        tdist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=d)
    try:
        W, H, cs, ucs, ol, seed = geom
        # rank 0 owns the model: the other ranks start from other weights and get the raw parameters by broadcast
        net = UtNet(funit=funit)
        net.load_state_dict(synth.make_utnet_state_dict(funit=funit, seed=9 if rank == 0 else 77 + rank))
        net = net.eval().to(d)
        net.split_k = False
        ndist.broadcast_parameters(net, src=0)
        geo = ndist.Geo(W, H, cs, ucs, ol)

        def compute(fr, cv, lo, hi):
            pipeline.denoise_frame(net, fr, cs, ucs, ol, batch=7, tile_range=(lo, hi), canvas=cv)

        stream = ndist.ShardedFrameStream(compute, geo, d)
        if stream.frames is not None:
            for f in stream.frames:
                f.fill_(float("nan"))       # rows that are never received must never be read
        for c in stream.canvas:
            c.fill_(7.0)
        frames = (torch.from_numpy(synth.make_frame(W, H, seed=seed + k)).to(d) for k in range(n_frames)) if rank == 0 else None
        got = {}
        for k, cv in stream.run(frames, n_frames):
            if rank == 0:
                got[k] = cv.cpu().numpy()       # (stream-ordered copy of a ring slot)
        torch.cuda.synchronize()
        if rank == 0:
            outq.put(got)
    finally:
        tdist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_pipelined_frame_stream_with_hip_compute(dev, world):
    """dist.ShardedFrameStream (the default N > 1 mode of bench.py) with the device loop as the per-rank compute, 5 frames through
    the two-slot rings (>= 3 in flight), weights broadcast from rank 0 as raw parameters: world size 1 over RCCL, world size 2
    over RCCL when two GPUs are visible, else both ranks on GPU 0 over gloo.  Every canvas == the single-GPU canvas of its frame."""
    import socket
    import torch.multiprocessing as mp
    from nind_denoise_amd import pipeline
    from nind_denoise_amd.networks.UtNet import UtNet
    geom = (500, 430, 120, 88, 16, 3)
    funit, n_frames = 16, 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, geom, funit, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    W, H, cs, ucs, ol, seed = geom
    net = UtNet(funit=funit)
    net.load_state_dict(synth.make_utnet_state_dict(funit=funit, seed=9))
    net = net.eval().to(dev)
    net.split_k = False
    assert sorted(got) == list(range(n_frames))
    for k in range(n_frames):
        ref = pipeline.denoise_frame(net, torch.from_numpy(synth.make_frame(W, H, seed=seed + k)).to(dev), cs, ucs, ol, batch=7).cpu().numpy()
        assert np.isfinite(got[k]).all()
        if world == 1:
            assert np.array_equal(got[k], ref)
        else:
            assert np.abs(got[k] - ref).max() <= 1e-6 and (got[k] == ref).mean() > 0.7


def test_bench_n2_rehearsal_line(dev):
    """`bench.py --gpus 2` end to end on the one-GPU box (ND_BENCH_REHEARSAL=1: both ranks on GPU 0, messages over gloo): the launcher
    spawns the ranks, the default N > 1 mode is the pipelined tile-shard stream, the parameters are broadcast from rank 0, rank 0
    prints ONE JSON line that names the partition and carries the pipeline / replicas keys."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ND_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "1500",
                        "--height", "1100", "--funit", "16", "--batch", "32", "--no-roofline"], env=env, cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3 and d["value"] > 0
    assert "tile-shard x2" in d["config"]["parallelism"] and "pipelined" in d["config"]["parallelism"]
    assert d["pipeline"]["compute_only_ms_largest_shard"] > 0 and 0 < d["pipeline"]["exchange_hidden_frac"] <= 1
    assert d["frame_shard"]["frames_per_step"] == 2 and "REPLICAS" in d["frame_shard"]["note"]
    assert d["weight_broadcast"]["bytes"] > 0 and "REHEARSAL" in d["data"]


# ---------------------------------------------------------------------------- fixtures executed by the reference itself

def test_whole_image_item_vs_reference_fixture(dev, golden_dir):
    # row a3: OneImageDS(whole_image=True, pad=p) as the reference builds it on square frames (sides mirrored, corners zero)
    import hashlib
    from nind_denoise_amd import denoise_image as di
    with open(os.path.join(golden_dir, "whole_image.json")) as f:
        cases = json.load(f)
    for c in cases:
        frame = synth.make_frame(c["side"], c["side"], seed=c["seed"])
        ds = di.OneImageDS(frame, None, None, None, whole_image=True, pad=c["pad"], device=dev)
        assert len(ds) == 1
        t, ud, us = ds[0]
        assert list(t.shape) == c["shape"] and ud.tolist() == c["usefuldim"] and us.tolist() == c["usefulstart"]
        got = hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()
        assert got == c["item_sha"], c


def test_stitch_vs_reference_main_loop_fixture(dev, golden_dir):
    # rows a8 / a9: the canvas the reference's own make_seamless_edges + main loop build (executed in the build container,
    # tests/golden/make_golden.py) with a stand-in network of exact fp32 operations; device gather -> model -> device stitch
    # must give the same bytes
    import hashlib
    from nind_denoise_amd import pipeline
    with open(os.path.join(golden_dir, "stitch_main_loop.json")) as f:
        cases = json.load(f)

    def stand_in(x):
        return 0.5 * x + 0.25 * torch.roll(x, 1, dims=1) + 0.01

    for c in cases:
        img = torch.from_numpy(synth.make_frame(c["W"], c["H"], seed=c["seed"])).to(dev)
        for batch in (1, 7):
            out = pipeline.denoise_frame(stand_in, img, c["cs"], c["ucs"], c["ol"], batch=batch)
            got = hashlib.sha256(np.ascontiguousarray(out.cpu().numpy()).tobytes()).hexdigest()
            assert got == c["canvas_sha"], (c["W"], c["H"], c["cs"], c["ucs"], c["ol"], batch)
