#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself (build container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)

What is imported from /root/reference (read-only, executed here, never copied):
  * networks/UtNet.py            -> ``UtNet``        (pure torch; imported by file path)
  * networks/ThirdPartyNets.py   -> ``UNet``         (module-level ``import torchvision`` is absent here)
  * denoise_image.py             -> ``OneImageDS``   (module-level cv2 / exiv2 / configargparse / torchvision /
                                                      imageio / piqa are absent here)
The absent third-party packages are given inert ``sys.modules`` placeholders so that the reference's
module-level ``import`` statements succeed; none of them is executed by the code under test except
``cv2.imread`` inside ``np_imgops.img_path_to_np_flt``, which is replaced by a function that returns the
in-memory frame (the procedure SURVEY.md section 8c recorded as working, with nothing denied).

The stitch loop of denoise_image.py (:204-213 ``make_seamless_edges``, :240-267 the loop) is ``__main__`` script code
and cannot be imported: the two statements are located in the parsed source (``ast``) and executed here, in the
build container, exactly as written (the loop runs over the reference's own ``OneImageDS`` through a
``torch.utils.data.DataLoader``, with a cheap deterministic stand-in for the network); only the SHA-256 of the
stitched canvas reaches the fixture -- no text of the reference is stored under tests/.

Outputs (data only -- inputs / expected outputs / hashes):
  tiler_geoms.json          per geometry: grid, per-tile (x0,y0,ud,us) table hash + first/last rows,
                            sha256 of every gathered tile, sha256 of the frame
  utnet_f8.npz              UtNet(funit=8) state-dict, inputs at cs 104 / 120, outputs, a few intermediates
  utnet_f64_cs264.npz       UtNet(64,'PReLU') output for synth weights seed 123 at cs=264 (+ digest of the weights)
  utnet_act_variants.npz    UtNet(funit=8, ELU / Hardswish) outputs at cs=104
  unet_256.npz              UNet() output for synth weights seed 0 on a 1x3x256x256 input
  whole_image.json          OneImageDS(whole_image=True, pad=p) on square frames: sha256 of the item, usefuldim, usefulstart
  stitch_main_loop.json     sha256 of the canvas the reference's own main loop builds (batch sizes 1 and 3) per geometry
"""
import ast
import hashlib
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/nind_denoise"

from nind_denoise_amd import synth  # noqa: E402


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _placeholder(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    ref_utnet = _load_by_path("ref_utnet", os.path.join(REF, "networks", "UtNet.py"))
    # inert placeholders for packages that are not installed in this image
    for n in ("cv2", "exiv2", "configargparse", "imageio"):
        try:
            importlib.import_module(n)
        except ModuleNotFoundError:
            _placeholder(n)
    try:
        importlib.import_module("torchvision")
    except ModuleNotFoundError:
        tv = _placeholder("torchvision")
        tv.models = _placeholder("torchvision.models")
        tv.transforms = _placeholder("torchvision.transforms")
        tv.utils = _placeholder("torchvision.utils")
    try:
        importlib.import_module("piqa")
    except ModuleNotFoundError:
        _placeholder("piqa", SSIM=torch.nn.Module, MS_SSIM=torch.nn.Module)
    ref_tpn = _load_by_path("ref_thirdparty", os.path.join(REF, "networks", "ThirdPartyNets.py"))
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        ref_di = importlib.import_module("denoise_image")
    finally:
        os.chdir(cwd)
    return ref_utnet, ref_tpn, ref_di


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


GEOMS = [
    # (W, H, cs, ucs, ol, seed)
    (500, 700, 264, 200, 64, 1),
    (608, 472, 264, 200, 64, 2),     # stride divides W-ucs
    (600, 400, 264, 200, 6, 3),
    (1500, 1000, 504, 480, 6, 4),    # shipped default cs/ucs/ol
    (777, 333, 104, 72, 10, 5),
    (300, 300, 264, 200, 0, 6),      # no overlap
    (401, 333, 120, 87, 9, 7),       # cs-ucs odd -> pad truncation
    (264, 264, 264, 200, 64, 8),     # one tile column/row pair, everything mirrored
    (6000, 4000, 264, 200, 64, 24),  # G24 (hashes of 12 sampled tiles only)
]


def tiler_fixtures(ref_di):
    import common.libs.np_imgops as ref_np_imgops  # the module object denoise_image imported
    out = []
    for (W, H, cs, ucs, ol, seed) in GEOMS:
        frame = synth.make_frame(W, H, seed=seed)
        ref_np_imgops.img_path_to_np_flt = lambda fpath, _f=frame: _f
        ds = ref_di.OneImageDS("in-memory", cs, ucs, ol)
        n = len(ds)
        big = n > 400
        sample = sorted(set([0, 1, n // 2, n - 1, n - 2] + [int(x) for x in np.linspace(0, n - 1, 8)])) if big else list(range(n))
        table, hashes = [], {}
        for i in range(n):
            if big and i not in sample:
                # geometry only: computed by the reference's formulas through a light call
                pass
            t, ud, us = ds[i] if (not big or i in sample) else (None, None, None)
            if t is not None:
                hashes[str(i)] = sha(t.numpy())
                table.append([i] + [int(v) for v in ud.tolist()] + [int(v) for v in us.tolist()])
        out.append(dict(W=W, H=H, cs=cs, ucs=ucs, ol=ol, seed=seed, size=n,
                        iperhl=int(ds.iperhl), pad=int(ds.pad), frame_sha=sha(frame),
                        table=table, tile_sha=hashes))
        print(f"tiler {W}x{H} cs{cs} ucs{ucs} ol{ol}: {n} tiles, {len(hashes)} hashed")
    with open(os.path.join(HERE, "tiler_geoms.json"), "w") as f:
        json.dump(out, f)


def utnet_fixtures(ref_utnet):
    torch.manual_seed(0)
    # funit=8, PReLU, own weights stored in the fixture
    sd = synth.make_utnet_state_dict(funit=8, seed=7)
    net = ref_utnet.UtNet(funit=8).eval()
    net.load_state_dict(sd)
    arrs = {f"sd/{k}": v.numpy() for k, v in sd.items()}
    taps = {}
    hooks = []
    for name in ("convs1.1", "convs1.3", "convs2.3", "bottom.1", "bottom.3", "up1", "tconvs1.3", "up4", "tconvs4.1", "tconvs4.3"):
        mod = net.get_submodule(name)
        hooks.append(mod.register_forward_hook(lambda m, i, o, _n=name: taps.__setitem__(_n, o.detach().clone())))
    for cs in (104, 120):
        x = torch.rand(2, 3, cs, cs, generator=torch.Generator().manual_seed(cs))
        with torch.no_grad():
            y = net(x)
        arrs[f"x{cs}"] = x.numpy()
        arrs[f"y{cs}"] = y.numpy()
        if cs == 104:
            for k, v in taps.items():
                arrs[f"tap104/{k}"] = v.numpy()
    for h in hooks:
        h.remove()
    np.savez(os.path.join(HERE, "utnet_f8.npz"), **arrs)
    print("utnet_f8 done")

    # activation variants
    arrs = {}
    for act in ("ELU", "Hardswish"):
        sd = synth.make_utnet_state_dict(funit=8, seed=11, activation=act)
        net = ref_utnet.UtNet(funit=8, activation=act).eval()
        net.load_state_dict(sd)
        x = torch.rand(1, 3, 104, 104, generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            arrs[f"y_{act}"] = net(x).numpy()
        arrs["x"] = x.numpy()
    np.savez(os.path.join(HERE, "utnet_act_variants.npz"), **arrs)

    # funit=64 (production net), synthetic weights regenerated from the seed at test time
    sd = synth.make_utnet_state_dict(funit=64, seed=123)
    net = ref_utnet.UtNet().eval()
    net.load_state_dict(sd)
    x = torch.rand(1, 3, 264, 264, generator=torch.Generator().manual_seed(264))
    with torch.no_grad():
        y = net(x)
    np.savez(os.path.join(HERE, "utnet_f64_cs264.npz"), x=x.numpy(), y=y.numpy(),
             sd_digest=np.array(synth.state_dict_digest(sd)), torch_version=np.array(torch.__version__))
    print("utnet_f64 done", float(y.abs().max()))
    # the reference rejects cs = 128/256/512 (SURVEY.md headline fact 3): record it
    bad = {}
    for cs in (128, 256, 512, 104, 264):
        try:
            with torch.no_grad():
                net(torch.zeros(1, 3, cs, cs))
            bad[str(cs)] = "ok"
        except RuntimeError as e:
            bad[str(cs)] = "RuntimeError"
    with open(os.path.join(HERE, "utnet_cs_validity.json"), "w") as f:
        json.dump(bad, f)


def unet_fixture(ref_tpn):
    sd = synth.make_unet_state_dict(seed=0)
    net = ref_tpn.UNet().eval()
    missing = net.load_state_dict(sd)
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        y = net(x)
    # odd size exercises the F.pad fix-up (ThirdPartyNets.py:110-118)
    x2 = torch.rand(1, 3, 100, 92, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y2 = net(x2)
    np.savez(os.path.join(HERE, "unet_256.npz"), x=x.numpy(), y=y.numpy(), x2=x2.numpy(), y2=y2.numpy(),
             sd_digest=np.array(synth.state_dict_digest(sd)), torch_version=np.array(torch.__version__))
    print("unet done", missing, float(y.mean()))


WHOLE = [
    # (side, pad, seed): square frames only -- the reference allocates (3, W+2p, H+2p) (denoise_image.py:113), which is
    # only consistent for W == H
    (88, 8, 31),       # 88 + 16 = 104: a valid UtNet size, so the whole-image item can go through the network
    (136, 8, 32),      # 152
    (120, 0, 33),      # pad 0: the item is the frame itself
    (97, 5, 34),       # odd everything
]


def whole_image_fixtures(ref_di):
    import common.libs.np_imgops as ref_np_imgops
    out = []
    for side, pad, seed in WHOLE:
        frame = synth.make_frame(side, side, seed=seed)
        ref_np_imgops.img_path_to_np_flt = lambda fpath, _f=frame: _f
        ds = ref_di.OneImageDS("in-memory", None, None, None, whole_image=True, pad=pad)
        assert len(ds) == 1
        t, ud, us = ds[0]
        out.append(dict(side=side, pad=pad, seed=seed, shape=list(t.shape), item_sha=sha(t.numpy()), frame_sha=sha(frame),
                        usefuldim=[int(v) for v in ud.tolist()], usefulstart=[int(v) for v in us.tolist()],
                        corner_is_zero=bool(pad) and bool((t[:, :pad, :pad] == 0).all())))
        print(f"whole image {side}x{side} pad {pad}: item {tuple(t.shape)}, corners zero: {out[-1]['corner_is_zero']}")
    with open(os.path.join(HERE, "whole_image.json"), "w") as f:
        json.dump(out, f)


def stand_in_model(x):
    """cheap deterministic stand-in for the network: [B,3,cs,cs] -> same (exact fp32 operations, one rounding per add)"""
    return 0.5 * x + 0.25 * torch.roll(x, 1, dims=1) + 0.01


def lift_main_loop():
    """compile the reference's own ``make_seamless_edges`` and main ``for`` loop out of denoise_image.py's __main__ block"""
    with open(os.path.join(REF, "denoise_image.py")) as f:
        tree = ast.parse(f.read())
    main_if = [n for n in tree.body if isinstance(n, ast.If) and "__main__" in ast.unparse(n.test)][0]
    fdef = [n for n in main_if.body if isinstance(n, ast.FunctionDef) and n.name == "make_seamless_edges"][0]
    loop = [n for n in main_if.body if isinstance(n, ast.For) and "DLoader" in ast.unparse(n.iter)][0]
    mod_f = ast.Module(body=[fdef], type_ignores=[])
    mod_l = ast.Module(body=[loop], type_ignores=[])
    return compile(mod_f, "<reference make_seamless_edges>", "exec"), compile(mod_l, "<reference main loop>", "exec")


def stitch_fixtures(ref_di):
    import common.libs.np_imgops as ref_np_imgops
    from torch.utils.data import DataLoader
    code_f, code_l = lift_main_loop()
    out = []
    for (W, H, cs, ucs, ol, seed) in GEOMS[:8]:
        frame = synth.make_frame(W, H, seed=seed)
        ref_np_imgops.img_path_to_np_flt = lambda fpath, _f=frame: _f
        shas = {}
        for bs in (1, 3):
            ds = ref_di.OneImageDS("in-memory", cs, ucs, ol)
            ns = dict(ref_di.__dict__)          # the names the script body sees (torch, math, os, sys, ...)
            ns.update(args=types.SimpleNamespace(overlap=ol, ucs=ucs, cs=cs, batch_size=bs, max_subpixels=None, whole_image=False,
                                                 debug=False, input="in-memory"),
                      fswidth=W, fsheight=H, ds=ds, device=torch.device("cpu"), model=stand_in_model,
                      DLoader=DataLoader(dataset=ds, num_workers=0, drop_last=False, batch_size=bs, shuffle=False),
                      newimg=torch.zeros(3, H, W, dtype=torch.float32), print=lambda *a, **k: None)
            exec(code_f, ns)
            exec(code_l, ns)
            shas[str(bs)] = sha(ns["newimg"].numpy())
        assert shas["1"] == shas["3"]
        out.append(dict(W=W, H=H, cs=cs, ucs=ucs, ol=ol, seed=seed, canvas_sha=shas["1"], frame_sha=sha(frame)))
        print(f"stitch {W}x{H} cs{cs} ucs{ucs} ol{ol}: canvas {shas['1'][:16]}")
    with open(os.path.join(HERE, "stitch_main_loop.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref_utnet, ref_tpn, ref_di = import_reference()
    only = set(sys.argv[1:])
    if not only or "tiler" in only:
        tiler_fixtures(ref_di)
    if not only or "whole" in only:
        whole_image_fixtures(ref_di)
    if not only or "stitch" in only:
        stitch_fixtures(ref_di)
    if not only or "utnet" in only:
        utnet_fixtures(ref_utnet)
    if not only or "unet" in only:
        unet_fixture(ref_tpn)
