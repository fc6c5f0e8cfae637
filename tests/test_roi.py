"""The fused denoise loop computes, in the last decoder levels, only what the useful centre of a tile depends on
(csrc/utnet_net.h: plan_rois; flag ND_FLAG_FULL_TILES restores whole tiles): same canvas as whole-tile computation and as the oracle."""
import numpy as np
import pytest
import torch

from nind_denoise_amd import _lib, pipeline, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _lib.load()
    return torch.device("cuda:0")


def _net(funit, dev, dtype="f32", seed=11):
    from nind_denoise_amd.networks.UtNet import UtNet
    sd = synth.make_utnet_state_dict(funit=funit, seed=seed)
    net = UtNet(funit=funit)
    net.load_state_dict(sd)
    return net.eval().to(dev).set_compute_dtype(dtype), sd


@pytest.mark.parametrize("funit,W,H,cs,ucs,ol,batch", [
    (16, 700, 500, 264, 200, 64, 7),      # fused 1-D Winograd + direct up layers
    (64, 640, 420, 264, 200, 64, 12),     # production width: three-pass F(6x6) layers with regions
    (64, 900, 600, 504, 480, 6, 4),       # the shipped default tiling (margin 12)
    (16, 520, 400, 248, 201, 32, 6),      # odd cs - ucs: pad = 23, useful width ucs + 1
    (16, 150, 110, 136, 16, 4, 9),        # a useful centre of 16 pixels in a 136-pixel tile: regions shrink to a few pixels at every level
    (64, 300, 280, 104, 96, 8, 5),        # margin 4 at the smallest tile: regions touch the borders
])
def test_useful_region_canvas_equals_whole_tiles_and_oracle(dev, funit, W, H, cs, ucs, ol, batch):
    from oracle import networks as onet
    from oracle import tiler as otiler
    net, sd = _net(funit, dev)
    frame = synth.make_frame(W, H, seed=3)
    img = torch.from_numpy(frame).to(dev)
    net.useful_only = True
    roi = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch).cpu().numpy()
    net.useful_only = False
    full = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch).cpu().numpy()
    scale = max(1.0, float(np.abs(full).max()))
    assert np.isfinite(roi).all()
    assert np.abs(roi - full).max() <= 3e-6 * scale, np.abs(roi - full).max()     # same math; Winograd tile grids start elsewhere
    grid = otiler.TileGrid(W, H, cs, ucs, ol)
    canvas = np.zeros_like(frame)
    with torch.no_grad():
        for i in range(grid.size):
            y = onet.utnet_forward(sd, torch.from_numpy(otiler.gather_tile(frame, grid, i))[None]).numpy()[0]
            otiler.stitch_add(canvas, y, grid, i)
    err = np.abs(roi - canvas).max()
    assert err <= 1e-3 and err <= 1e-3 * max(np.abs(canvas).max(), 1e-6), err
    print(f"useful-region loop f{funit} {W}x{H} cs{cs}/ucs{ucs}: |roi - whole| {np.abs(roi - full).max():.2e}, |roi - oracle| {err:.2e}")


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_useful_region_half_storage(dev, dtype):
    net, _ = _net(16, dev, dtype)
    frame = synth.make_frame(600, 450, seed=5)
    img = torch.from_numpy(frame).to(dev)
    net.useful_only = True
    roi = pipeline.denoise_frame(net, img, 264, 200, 64, batch=6).cpu().numpy()
    net.useful_only = False
    full = pipeline.denoise_frame(net, img, 264, 200, 64, batch=6).cpu().numpy()
    # direct kernels on both paths: a pixel's K loop is the same, only its tile / lane differs
    assert np.abs(roi - full).max() <= 1e-6 * max(1.0, float(np.abs(full).max())), np.abs(roi - full).max()


def test_useful_region_random_geometries(dev):
    # seeded sweep: tile sizes of the form 16k + 56, useful sizes / overlaps / frame sizes / batch sizes at random (incl. frames barely
    # larger than a tile and margins from 1 pixel to a third of the tile): the useful-region canvas must equal the whole-tile canvas
    rng = np.random.default_rng(20260105)
    nets = {(16, "f32"): _net(16, dev)[0], (64, "f32"): _net(64, dev)[0], (16, "bf16"): _net(16, dev, "bf16")[0]}
    for case in range(14):
        funit, dtype = [(16, "f32"), (64, "f32"), (16, "bf16")][case % 3]
        cs = 56 + 16 * int(rng.integers(3, 11 if funit == 16 else 9))
        margin = int(rng.integers(1, cs // 3))
        ucs = cs - 2 * margin - int(rng.integers(0, 2))
        ol = int(rng.integers(0, max(1, min(ucs - 1, 2 * margin + 1))))
        W = cs + int(rng.integers(1, 2 * cs))        # (the tiler refuses frames smaller than its mirror padding)
        H = cs + int(rng.integers(1, cs))
        batch = int(rng.integers(1, 12))
        net = nets[(funit, dtype)]
        # 16-bit storage: with the split-K tail on, a re-associated fp32 sum may round to the neighbouring bf16 value and the two
        # canvases drift by ~1e-4; with whole-K tiles a pixel's sum does not depend on the launch and the canvases must be identical
        net.split_k = dtype == "f32"
        img = torch.from_numpy(synth.make_frame(W, H, seed=case)).to(dev)
        net.useful_only = True
        roi = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch)
        net.useful_only = False
        full = pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch)
        net.useful_only = True
        scale = max(1.0, float(full.abs().max().item()))
        err = float((roi - full).abs().max().item())
        assert torch.isfinite(roi).all() and err <= (3e-6 * scale if dtype == "f32" else 0.0), (case, funit, dtype, W, H, cs, ucs, ol, batch, err)
