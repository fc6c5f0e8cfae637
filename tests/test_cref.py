"""The plain-C oracle restatement (oracle/cref) against the golden tables, the numpy oracle and torch (CPU only)."""
import ctypes
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from nind_denoise_amd import synth
from oracle import tiler as otiler

CREF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "cref")
FP = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def cref():
    so = os.path.join(CREF_DIR, "libnd_cref.so")
    if not os.path.isfile(so):
        subprocess.check_call(["make", "-C", CREF_DIR])
    return ctypes.CDLL(so)


def fp(a):
    return a.ctypes.data_as(FP)


def test_c_tiler_matches_golden_and_numpy_oracle(cref, golden_dir):
    with open(os.path.join(golden_dir, "tiler_geoms.json")) as f:
        geoms = json.load(f)
    for g in geoms[:8]:
        W, H, cs, ucs, ol = g["W"], g["H"], g["cs"], g["ucs"], g["ol"]
        frame = synth.make_frame(W, H, seed=g["seed"])
        grid = otiler.TileGrid(W, H, cs, ucs, ol)
        tile = np.empty((3, cs, cs), dtype=np.float32)
        x0, y0 = ctypes.c_int(), ctypes.c_int()
        ud, us = (ctypes.c_int * 4)(), (ctypes.c_int * 2)()
        for row in g["table"]:
            i = row[0]
            cref.cref_tile_geom(i, W, H, cs, ucs, ol, ctypes.byref(x0), ctypes.byref(y0), ud, us)
            assert list(ud) + list(us) == row[1:]
            cref.cref_gather_tile(fp(frame), W, H, cs, ucs, ol, i, fp(tile))
            assert hashlib.sha256(tile.tobytes()).hexdigest() == g["tile_sha"][str(i)]
        rng = np.random.default_rng(1)
        tiles = rng.standard_normal((grid.size, 3, cs, cs), dtype=np.float32)
        a = np.zeros((3, H, W), dtype=np.float32)
        b = np.zeros((3, H, W), dtype=np.float32)
        for i in range(grid.size):
            otiler.stitch_add(a, tiles[i], grid, i)
            cref.cref_stitch_add(fp(b), W, H, cs, ucs, ol, i, fp(np.ascontiguousarray(tiles[i])))
        assert np.array_equal(a, b)


def test_c_layers_match_torch(cref):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 5, 9, 11, generator=g) - 0.5
    w = torch.rand(7, 5, 3, 3, generator=g) - 0.5
    b = torch.rand(7, generator=g)
    y = np.empty((2, 7, 7, 9), dtype=np.float32)
    cref.cref_conv2d(fp(x.numpy()), 2, 5, 9, 11, fp(w.numpy()), fp(b.numpy()), 7, 3, fp(y))
    assert np.abs(y - F.conv2d(x, w, b).numpy()).max() < 2e-6
    for k, s in ((3, 1), (2, 2)):
        wt = torch.rand(5, 6, k, k, generator=g) - 0.5
        bt = torch.rand(6, generator=g)
        ref = F.conv_transpose2d(x, wt, bt, stride=s).numpy()
        y = np.empty(ref.shape, dtype=np.float32)
        scratch = np.empty(ref.size, dtype=np.float64)
        cref.cref_conv_transpose2d(fp(x.numpy()), 2, 5, 9, 11, fp(wt.numpy()), fp(bt.numpy()), 6, k, s,
                                   scratch.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), fp(y))
        assert np.abs(y - ref).max() < 2e-6
    xp = torch.rand(3, 4, 10, 14, generator=g)
    yp = np.empty((3, 4, 5, 7), dtype=np.float32)
    cref.cref_maxpool2(fp(xp.numpy()), 12, 10, 14, fp(yp))
    assert np.array_equal(yp, F.max_pool2d(xp, 2).numpy())
    z = (torch.rand(100, generator=g) - 0.5).numpy().copy()
    ref = F.prelu(torch.from_numpy(z.copy()), torch.tensor([0.3])).numpy()
    cref.cref_prelu(fp(z), ctypes.c_size_t(100), ctypes.c_float(0.3))
    assert np.array_equal(z, ref)
