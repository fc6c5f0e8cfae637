"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol the header declares, the
pure-integer tile geometry entry points match the golden tables, weight packing, model-path resolution, CLI
defaults and the image codecs.  No compute call needs a GPU here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from nind_denoise_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nind_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.load().nd_version() >= 100


def test_tile_geometry_matches_golden_tables(golden_dir):
    with open(os.path.join(golden_dir, "tiler_geoms.json")) as f:
        geoms = json.load(f)
    for g in geoms:
        cols, rows, pad = _lib.tile_grid(g["W"], g["H"], g["cs"], g["ucs"], g["ol"])
        assert cols * rows == g["size"] and cols == g["iperhl"] + 1 and pad == g["pad"]
        for row in g["table"]:
            _, _, ud, us = _lib.tile_geom(row[0], g["W"], g["H"], g["cs"], g["ucs"], g["ol"])
            assert list(ud) + list(us) == row[1:]


def test_tile_geometry_equals_oracle_everywhere():
    from oracle import tiler as otiler
    for (W, H, cs, ucs, ol) in [(6000, 4000, 264, 200, 64), (9504, 6336, 520, 456, 64), (6000, 4000, 504, 480, 6),
                                (401, 333, 120, 87, 9), (200, 200, 264, 200, 64)]:
        grid = otiler.TileGrid(W, H, cs, ucs, ol)
        cols, rows, pad = _lib.tile_grid(W, H, cs, ucs, ol)
        assert (cols, rows, pad) == (grid.cols, grid.rows, grid.pad)
        for i in range(grid.size):
            x0, y0, ud, us = _lib.tile_geom(i, W, H, cs, ucs, ol)
            assert (x0, y0, ud, us) == grid.geom(i)
    assert _lib.tile_grid(6000, 4000, 264, 200, 64)[:2] == (44, 29)       # G24: 1276 tiles
    assert _lib.tile_grid(9504, 6336, 520, 456, 64)[:2] == (25, 16)       # G61: 400 tiles
    assert _lib.tile_grid(6000, 4000, 504, 480, 6)[:2] == (13, 9)         # G24d: 117 tiles


def test_tile_geometry_errors():
    with pytest.raises(ValueError):
        _lib.tile_grid(100, 100, 264, 200, 64)     # frame smaller than ucs: undefined in the reference
    with pytest.raises(ValueError):
        _lib.tile_grid(500, 500, 264, 200, 200)    # ucs must exceed the overlap
    with pytest.raises(ValueError):
        _lib.tile_geom(99, 500, 700, 264, 200, 64)


def _up_row(m, cout, cpp=4):
    """GEMM row m of a 2x2 stride-2 transpose -> (a, b, co) (nd_up_row of csrc/nd_common.h: the two lane halves of an MFMA
    accumulator group are the output columns 2x, 2x + 1 of one channel group, so that a wave stores contiguous runs):
    m = 2 cpp (a * cout / cpp + group) + cpp * b + e, co = cpp * group + e; cpp = 4 channels per plane element in fp32."""
    e, b, g = m % cpp, (m // cpp) & 1, m // (2 * cpp)
    return g // (cout // cpp), b, cpp * (g % (cout // cpp)) + e


def _ref_pack(kind, cin, cout, w):
    taps = 9 if kind in ("conv3", "convT3") else 1
    M = 4 * cout if kind == "convT2s2" else cout
    MT, KB = ((M + 255) // 256 * 8 if kind == "convT2s2" else (M + 127) // 128 * 4), (cin + 7) // 8
    out = np.zeros((MT, KB, taps, 64, 4), dtype=np.float32)
    for mt in range(MT):
        for lane in range(64):
            i, h = lane & 31, lane >> 5
            m = 32 * mt + i
            if m >= M:
                continue
            for kb in range(KB):
                for s in range(4):
                    ci = 8 * kb + 4 * h + s
                    if ci >= cin:
                        continue
                    for t in range(taps):
                        if kind == "conv3":
                            v = w[m, ci, t // 3, t % 3]
                        elif kind == "convT3":
                            v = w[ci, m, 2 - t // 3, 2 - t % 3]
                        elif kind == "convT2s2":
                            a, b, co = _up_row(m, cout)
                            v = w[ci, co, a, b]
                        else:
                            v = w[m, ci, 0, 0]
                        out[mt, kb, t, lane, s] = v
    return out.reshape(-1)


@pytest.mark.parametrize("kind,cin,cout", [("conv3", 3, 8), ("conv3", 16, 40), ("convT3", 24, 16), ("convT2s2", 16, 8),
                                           ("conv1", 8, 12)])
def test_weight_packing(kind, cin, cout):
    lib = _lib.load()
    k = {"conv3": 3, "convT3": 3, "convT2s2": 2, "conv1": 1}[kind]
    shape = (cout, cin, k, k) if kind in ("conv3", "conv1") else (cin, cout, k, k)
    w = torch.randn(shape, generator=torch.Generator().manual_seed(1))
    b = torch.randn(cout, generator=torch.Generator().manual_seed(2))
    nbytes = lib.nd_layer_packed_bytes(_lib.KIND[kind], cin, cout, _lib.ND_F32)
    packed = torch.empty(nbytes // 4)
    _lib.check(lib.nd_layer_pack(_lib.KIND[kind], cin, cout, _lib.ND_F32, w.data_ptr(), b.data_ptr(), packed.data_ptr(), nbytes))
    ref = _ref_pack(kind, cin, cout, w.numpy())
    assert np.array_equal(packed.numpy()[:ref.size], ref)
    bias = packed.numpy()[ref.size:]
    M = 4 * cout if kind == "convT2s2" else cout
    want = b.numpy()[[_up_row(m, cout)[2] for m in range(M)]] if kind == "convT2s2" else b.numpy()
    assert np.array_equal(bias[:M], want)
    if kind == "convT2s2":   # every (a, b, co) appears exactly once
        assert sorted(_up_row(m, cout) for m in range(M)) == [(a, b_, c) for a in range(2) for b_ in range(2) for c in range(cout)]
    assert not bias[M:].any()
    with pytest.raises(MemoryError):
        _lib.check(lib.nd_layer_pack(_lib.KIND[kind], cin, cout, _lib.ND_F32, w.data_ptr(), b.data_ptr(), packed.data_ptr(), 16))


def test_utnet_module_has_reference_state_dict_layout(golden_dir):
    from nind_denoise_amd.networks.UtNet import UtNet
    net = UtNet()
    sd = net.state_dict()
    assert list(sd.keys()) == _lib.utnet_tensor_names()
    assert len(sd) == 64 and sum(p.numel() for p in net.parameters()) == 31_031_893
    assert tuple(sd["bottom.2.weight"].shape) == (1024, 1024, 3, 3) and tuple(sd["up1.weight"].shape) == (1024, 512, 2, 2)
    assert tuple(sd["tconvs4.4.weight"].shape) == (3, 64, 1, 1) and tuple(sd["convs1.1.weight"].shape) == (1,)
    # the fixture state-dict produced for the REFERENCE module loads strictly
    d = np.load(os.path.join(golden_dir, "utnet_f8.npz"))
    UtNet(funit=8).load_state_dict({k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd/")}, strict=True)
    UtNet(funit=8, activation="ELU").load_state_dict(synth.make_utnet_state_dict(8, 1, "ELU"), strict=True)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 104, 104))  # CPU tensor: no fallback


def test_utnet_pack_whole_net_and_sizes():
    lib = _lib.load()
    assert lib.nd_utnet_flops(64, 264) == 84_830_297_600.0
    assert lib.nd_utnet_flops(64, 256) == 0.0
    assert lib.nd_utnet_workspace_bytes(64, 256, 1, 0) == 0
    assert lib.nd_utnet_workspace_bytes(64, 264, 4, 0) > 4 * 150e6
    from nind_denoise_amd.networks.UtNet import UtNet, nearest_valid_cs, valid_cs
    assert [nearest_valid_cs(c) for c in (128, 256, 512, 50)] == [136, 264, 520, 104]
    assert valid_cs(264) and valid_cs(504) and not valid_cs(256)


def test_model_path_resolution(tmp_path):
    from nind_denoise_amd.nn_common import Model
    d = tmp_path / "2021-run"
    d.mkdir()
    for e in (3, 12, 7):
        (d / f"generator_{e}.pt").write_bytes(b"")
    (d / "discriminator_99.pt").write_bytes(b"")
    assert Model.complete_path(str(d), None, keyword="generator").endswith("generator_12.pt")
    (d / "trainres.json").write_text(json.dumps({"best_epoch": {"validation_loss": 7}}))
    assert Model.complete_path(str(d), None, keyword="generator").endswith("generator_7.pt")
    assert Model.complete_path("2021-run", str(tmp_path), keyword="generator").endswith("generator_7.pt")
    f = d / "generator_3.pt"
    assert Model.complete_path(str(f), None) == str(f)
    with pytest.raises(FileNotFoundError):
        Model.complete_path(str(tmp_path / "nope"), str(tmp_path))
    # state-dict round trip through instantiate_model (stays on CPU: no forward)
    from nind_denoise_amd.networks.UtNet import UtNet
    sd = synth.make_utnet_state_dict(8, 3)
    torch.save(sd, d / "generator_7.pt")
    m = Model.instantiate_model(model_path=str(d), network="UtNet", device="cpu", strparameters="funit=8", keyword="generator")
    assert isinstance(m, UtNet) and torch.equal(m.state_dict()["up2.bias"], sd["up2.bias"])


def test_cli_defaults_and_autodetect():
    from nind_denoise_amd import denoise_image as di
    a = di.parse_args(["--model_path", "m/x_utnet_y/generator_650.pt"])
    assert (a.overlap, a.input, a.exif_method, a.batch_size) == (6, "in.jpg", "piexif", None)
    di.autodetect_network_cs_ucs(a)
    assert (a.g_network, a.cs, a.ucs) == ("UtNet", 504, 480)
    a = di.parse_args(["--network", "UNet", "--model_path", "g.pt", "--cs", "300"])   # ucs missing -> both replaced
    di.autodetect_network_cs_ucs(a)
    assert (a.cs, a.ucs) == (440, 320)
    a = di.parse_args(["--arch", "UtNet", "--model_path", "g.pt", "--cs", "264", "--ucs", "200", "-ol", "64", "-b", "8"])
    di.autodetect_network_cs_ucs(a)
    assert (a.g_network, a.cs, a.ucs, a.overlap, a.batch_size) == ("UtNet", 264, 200, 64, 8)
    with pytest.raises(SystemExit):
        di.autodetect_network_cs_ucs(di.parse_args(["--model_path", "nothing_recognisable.pt"]))


def test_image_codecs_roundtrip(tmp_path):
    from PIL import Image
    from nind_denoise_amd.common.libs import imgcodec, np_imgops, pt_helpers
    rng = np.random.default_rng(0)
    t = torch.from_numpy(rng.random((3, 41, 57), dtype=np.float32) * 1.2 - 0.1)   # values outside [0,1]
    p32 = str(tmp_path / "o.tiff")
    pt_helpers.tensor_to_imgfile(t, p32)                        # 'tiff' -> float32, NOT clipped
    assert np.array_equal(np_imgops.img_path_to_np_flt(p32), t.numpy())
    for ext in (".tif", ".png"):                                # 16-bit: round(clip(x)*65535)
        p16 = str(tmp_path / ("o" + ext))
        pt_helpers.tensor_to_imgfile(t, p16)
        want = (t.clip(0, 1) * 65535).round().numpy().astype(np.uint16).astype(np.float32) / 65535
        assert np.array_equal(np_imgops.img_path_to_np_flt(p16), want)
    pj = str(tmp_path / "o.jpg")
    pt_helpers.tensor_to_imgfile(t, pj)
    assert np_imgops.img_path_to_np_flt(pj).shape == (3, 41, 57)
    x8 = (rng.random((33, 47, 3)) * 255).astype(np.uint8)
    for comp in (None, "tiff_lzw", "tiff_adobe_deflate", "packbits"):
        Image.fromarray(x8).save(str(tmp_path / "p.tif"), compression=comp)
        assert np.array_equal(imgcodec.read_tiff(str(tmp_path / "p.tif")), x8), comp
    Image.fromarray(x8).save(str(tmp_path / "p.png"))           # PIL picks per-row filters, Paeth included
    assert np.array_equal(imgcodec.read_png(str(tmp_path / "p.png")), x8)
    with pytest.raises(FileNotFoundError):
        np_imgops.img_path_to_np_flt(str(tmp_path / "missing.tif"))


def test_unet_module_state_dict_layout():
    from nind_denoise_amd.networks.ThirdPartyNets import UNet
    from nind_denoise_amd.nn_common import NETWORKS
    assert set(NETWORKS) >= {"UtNet", "UNet"}       # the reference only registers UtNet (nn_common.py:12)
    net = UNet()
    sd = net.state_dict()
    ref = synth.make_unet_state_dict(seed=0)
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    assert sum(p.numel() for p in net.parameters()) == 14_789_059
    lib = _lib.load()
    names = [lib.nd_unet_tensor_name(i).decode() for i in range(lib.nd_unet_num_tensors())]
    assert set(names) <= set(sd)


def test_read_16bit_png_of_the_reference(golden_dir):
    """The reference's own 16-bit test image (unittest_resources/NIND_bananapi_ISO50_20_30_104.png, a data file its
    dataset_torch_3 tests hold; kept under tests/golden/) through the PNG reader and the reference's conversion rule
    (np_imgops.py:12-29: RGB, CHW, uint16 / 65535; the alpha channel is dropped as cv2.IMREAD_COLOR does).  cv2 is absent, so the
    decode is cross-checked against PIL, which reads the same file at 8 bits (the high byte of every sample)."""
    from PIL import Image
    from nind_denoise_amd.common.libs import imgcodec, np_imgops
    path = os.path.join(golden_dir, "NIND_bananapi_ISO50_20_30_104.png")
    raw = imgcodec.read_png(path)
    assert raw.dtype == np.uint16 and raw.shape == (144, 144, 4)
    assert raw.max() > 255 and (raw & 0xFF).std() > 10          # really 16 bits of signal, not 8 bits shifted
    pil = np.asarray(Image.open(path))
    assert pil.shape == raw.shape and np.array_equal((raw >> 8).astype(np.uint8), pil)
    img = np_imgops.img_path_to_np_flt(path)
    assert img.dtype == np.float32 and img.shape == (3, 144, 144)
    assert np.array_equal(img, raw[:, :, :3].transpose(2, 0, 1).astype(np.float32) / 65535)
    assert 0.0 <= img.min() and img.max() <= 1.0


@pytest.mark.parametrize("cs,ucs", [(264, 200), (504, 480), (520, 456), (248, 201), (136, 16), (104, 96)])
def test_useful_regions_equal_receptive_field_backprojection(cs, ucs):
    """nd_utnet_useful_region (the regions the fused loop restricts the decoder layers to) against an independent restatement:
    project the kept centre [pad, cs - pad) backwards through the layer list of UtNet.py:97-109 (ZeroPad2d(-2) + 1x1, transposed 3x3
    layers grow the needed interval by 2 on the low side, 2x2 stride-2 transposes halve it)."""
    import ctypes
    lib = _lib.load()
    crop = int((cs - ucs) / 2)
    names = [lib.nd_utnet_step_name(i).decode() for i in range(26)]
    # sizes of the decoder tensors for a cs x cs tile (valid convolutions / transposed convolutions of UtNet)
    l1 = cs
    l2 = l1 // 2 - 4
    l3 = l2 // 2 - 4
    l4 = l3 // 2 - 4
    out_size = {"tconvs4.2": l1 + 4, "tconvs4.0": l1 + 2, "tconvs3.2": l2 + 4, "tconvs3.0": l2 + 2, "tconvs2.2": l3 + 4, "tconvs2.0": l3 + 2,
                "tconvs1.2": l4 + 4, "tconvs1.0": l4 + 2}
    up_in = {"up4": l2 + 4, "up3": l3 + 4, "up2": l4 + 4, "up1": l4 // 2}
    want = {}
    lo, hi = crop + 2, cs - crop + 2                       # rows of tconvs4.2's output that the final 1x1 + crop reads
    for lvl in (4, 3, 2, 1):
        for name in (f"tconvs{lvl}.2", f"tconvs{lvl}.0"):
            size = out_size[name]
            a, b = max(lo, 0), min(hi, size)
            want[name] = None if (a, b) == (0, size) else (a, b - a)
            lo, hi = max(a - 2, 0), min(b, size - 2)       # input rows (interior) a transposed 3x3 layer needs
        size = up_in[f"up{lvl}"]
        a, b = lo >> 1, min((hi + 1) >> 1, size)
        want[f"up{lvl}"] = None if (a, b) == (0, size) else (a, b - a)
        lo, hi = a, b
    rect = (ctypes.c_int * 4)()
    restricted = 0
    for i, name in enumerate(names):
        n = lib.nd_utnet_useful_region(64, cs, crop, i, rect)
        assert n >= 0, lib.nd_last_error()
        got = tuple(rect)
        if name in want and want[name] is not None:
            r0, rows = want[name]
            assert got == (r0, r0, rows, rows), (name, got, want[name])
            restricted += 1
        else:
            assert got == (0, 0, 0, 0), (name, got)           # encoder, pools and fully needed decoder layers: whole tensors
    assert n == restricted


def test_thin_client_protocol(tmp_path):
    """nind_denoise_amd.client against a stand-in worker (no GPU): argument list and working directory go over the Unix socket,
    streamed output is relayed to the right stream, the exit status is the worker's; the client process imports neither torch
    nor the HIP library; no worker -> a clear message and a non-zero status."""
    import json
    import socket
    import subprocess
    import sys
    import threading
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = str(tmp_path / "s.sock")
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(sock)
    srv.listen(4)
    seen = []

    def serve():
        conn, _ = srv.accept()
        with conn, conn.makefile("rwb") as f:
            seen.append(json.loads(f.readline()))
            for m in ({"stream": "stdout", "data": "0/2\n"}, {"stream": "stderr", "data": "warn\n"}, {"stream": "stdout", "data": "done\n"}, {"exit": 7}):
                f.write((json.dumps(m) + "\n").encode())
            f.flush()
    t = threading.Thread(target=serve, daemon=True)
    t.start()
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-X", "importtime", "-m", "nind_denoise_amd.denoise_image", "-i", "a.tif", "--server", sock, "-o", "b.tiff"],
                       env=env, cwd=tmp_path, capture_output=True, text=True, timeout=60)
    t.join(timeout=10)
    srv.close()
    assert r.returncode == 7 and r.stdout == "0/2\ndone\n" and "warn" in r.stderr
    assert seen == [{"argv": ["-i", "a.tif", "-o", "b.tiff"], "cwd": str(tmp_path)}]
    assert "| torch" not in r.stderr and "numpy" not in r.stderr and "ctypes" not in r.stderr
    r = subprocess.run([sys.executable, "-m", "nind_denoise_amd.denoise_image", "-i", "a.tif"], env=dict(env, NIND_DENOISE_SERVER=str(tmp_path / "none.sock")),
                       cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 111 and "no worker at" in r.stderr
