"""The oracle (oracle/) against the fixtures generated from the reference itself (tests/golden/make_golden.py).

CPU only.  Tile geometry / gathered bytes / identity stitch are bit-exact; network outputs are compared
at 1e-5 (same torch CPU kernels as the reference, different graph construction only).
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from nind_denoise_amd import synth
from oracle import networks as onet
from oracle import tiler as otiler


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _geoms(golden_dir):
    with open(os.path.join(golden_dir, "tiler_geoms.json")) as f:
        return json.load(f)


def test_tiler_geometry_and_gather_bitexact(golden_dir):
    for g in _geoms(golden_dir):
        frame = synth.make_frame(g["W"], g["H"], seed=g["seed"])
        assert sha(frame) == g["frame_sha"], "synthetic frame generator drifted"
        grid = otiler.TileGrid(g["W"], g["H"], g["cs"], g["ucs"], g["ol"])
        assert grid.size == g["size"] and grid.iperhl == g["iperhl"] and grid.pad == g["pad"]
        for row in g["table"]:
            i = row[0]
            _, _, ud, us = grid.geom(i)
            assert list(ud) + list(us) == row[1:], (g, i)
            assert sha(otiler.gather_tile(frame, grid, i)) == g["tile_sha"][str(i)], (g["W"], g["H"], i)


def test_integer_tile_index_equals_reference_float_formula():
    # denoise_image.py:131 uses ceil((i+1)/(iperhl+1) - 1) in floating point; i // cols is the same integer
    import math
    for cols in (1, 2, 3, 7, 13, 44, 45, 101, 1000):
        for i in range(0, cols * 60):
            assert int(math.ceil((i + 1) / cols - 1)) == i // cols


@pytest.mark.parametrize("batch", [1, 5])
def test_identity_model_roundtrip_bitexact(golden_dir, batch):
    # tiler + seamless stitch with an identity "network" must reproduce the frame exactly
    for g in _geoms(golden_dir)[:8]:
        if (g["cs"] - g["ucs"]) % 2:
            # reference quirk: pad = int((cs-ucs)/2) truncates, the useful crop becomes ucs+1 wide and one
            # column/row per seam is summed at full weight twice -- the reference itself is not an identity there
            continue
        frame = synth.make_frame(g["W"], g["H"], seed=g["seed"])
        out = otiler.denoise_frame(frame, g["cs"], g["ucs"], g["ol"], lambda x: x, batch=batch)
        assert np.array_equal(out, frame), (g["W"], g["H"], g["cs"], g["ucs"], g["ol"])


def test_utnet_f8_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "utnet_f8.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd/")}
    for cs in (104, 120):
        taps = {}
        y = onet.utnet_forward(sd, torch.from_numpy(d[f"x{cs}"]), taps=taps)
        assert y.shape == d[f"y{cs}"].shape
        assert np.abs(y.numpy() - d[f"y{cs}"]).max() <= 1e-5
        if cs == 104:
            for k in d.files:
                if k.startswith("tap104/"):
                    assert np.abs(taps[k[7:]].numpy() - d[k]).max() <= 1e-5, k


def test_utnet_activation_variants(golden_dir):
    d = np.load(os.path.join(golden_dir, "utnet_act_variants.npz"))
    for act in ("ELU", "Hardswish"):
        sd = synth.make_utnet_state_dict(funit=8, seed=11, activation=act)
        y = onet.utnet_forward(sd, torch.from_numpy(d["x"]), activation=act)
        assert np.abs(y.numpy() - d[f"y_{act}"]).max() <= 1e-5


def test_utnet_f64_cs264_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "utnet_f64_cs264.npz"))
    sd = synth.make_utnet_state_dict(funit=64, seed=123)
    assert synth.state_dict_digest(sd) == str(d["sd_digest"]), "synthetic weight generator drifted"
    with torch.no_grad():
        y = onet.utnet_forward(sd, torch.from_numpy(d["x"]))
    assert np.abs(y.numpy() - d["y"]).max() <= 1e-5


def test_utnet_cs_validity(golden_dir):
    with open(os.path.join(golden_dir, "utnet_cs_validity.json")) as f:
        v = json.load(f)
    for cs, res in v.items():
        assert onet.utnet_valid_cs(int(cs)) == (res == "ok")


def test_utnet_flop_table():
    # SURVEY.md section 2a (= torch FlopCounterMode on the reference)
    assert onet.utnet_flops(264) == 84_830_297_600
    assert onet.utnet_flops(104) == 10_142_113_280
    assert onet.utnet_flops(504) == 338_002_542_080
    assert onet.utnet_flops(520) == 360_902_663_680


def test_unet_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "unet_256.npz"))
    sd = synth.make_unet_state_dict(seed=0)
    assert synth.state_dict_digest(sd) == str(d["sd_digest"])
    with torch.no_grad():
        y = onet.unet_forward(sd, torch.from_numpy(d["x"]))
        y2 = onet.unet_forward(sd, torch.from_numpy(d["x2"]))
    assert np.abs(y.numpy() - d["y"]).max() <= 1e-5
    assert np.abs(y2.numpy() - d["y2"]).max() <= 1e-5


def test_whole_image_item_vs_reference(golden_dir):
    # OneImageDS(whole_image=True, pad=p) run by the reference itself on square frames (tests/golden/make_golden.py):
    # sides mirrored, corners zero
    with open(os.path.join(golden_dir, "whole_image.json")) as f:
        cases = json.load(f)
    assert any(c["corner_is_zero"] for c in cases)
    for c in cases:
        frame = synth.make_frame(c["side"], c["side"], seed=c["seed"])
        assert sha(frame) == c["frame_sha"]
        item, ud, us = otiler.whole_image_item(frame, c["pad"])
        assert list(item.shape) == c["shape"] and list(ud) == c["usefuldim"] and list(us) == c["usefulstart"]
        assert sha(item) == c["item_sha"], c


def _stand_in(x):
    # the stand-in network of make_golden.py (exact fp32 operations, one rounding per add)
    return np.float32(0.5) * x + np.float32(0.25) * np.roll(x, 1, axis=1) + np.float32(0.01)


@pytest.mark.parametrize("batch", [1, 4])
def test_stitch_vs_reference_main_loop(golden_dir, batch):
    # canvas hashes produced by the reference's own make_seamless_edges + main loop (denoise_image.py:204-213, 240-267),
    # executed in the build container over the reference's OneImageDS with the stand-in network
    with open(os.path.join(golden_dir, "stitch_main_loop.json")) as f:
        cases = json.load(f)
    assert len(cases) == 8
    for c in cases:
        frame = synth.make_frame(c["W"], c["H"], seed=c["seed"])
        assert sha(frame) == c["frame_sha"]
        out = otiler.denoise_frame(frame, c["cs"], c["ucs"], c["ol"], _stand_in, batch=batch)
        assert sha(out) == c["canvas_sha"], (c["W"], c["H"], c["cs"], c["ucs"], c["ol"])
