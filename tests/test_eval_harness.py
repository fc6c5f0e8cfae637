"""Eval harness: score oracle properties, host helpers; GPU: HIP scores vs the oracle and
denoise_dir end to end.  The SSIM / MS-SSIM arithmetic is piqa's (not installed, not vendored by the reference): PARITY
UNPINNED against piqa itself; the oracle restates its published algorithm and is pinned here by properties that hold for
any correct SSIM (identity, symmetry, the closed form on constant images, the 161-pixel limit of five scales)."""
import json
import os

import numpy as np
import pytest
import torch

from nind_denoise_amd import dataset_torch_3, synth
from nind_denoise_amd.common.libs import json_saver, utilities
from oracle import losses as olosses


def _pair(n, c, h, w, seed, noise=0.1):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, c, h, w, generator=g)
    y = (x + noise * torch.randn(n, c, h, w, generator=g)).clip(0, 1)
    return x, y


# ---------------------------------------------------------------------------- oracle properties (CPU)

def test_oracle_window_and_identity():
    g = olosses.gaussian_window()
    assert g.numel() == 11 and abs(g.sum().item() - 1) < 1e-6 and torch.equal(g, g.flip(0)) and g.argmax().item() == 5
    x, y = _pair(2, 3, 170, 180, 0)
    assert torch.allclose(olosses.ssim(x, x), torch.ones(2), atol=1e-6)
    assert torch.allclose(olosses.ms_ssim(x, x), torch.ones(2), atol=1e-6)
    assert torch.allclose(olosses.ssim(x, y), olosses.ssim(y, x), atol=1e-6)
    s, m = olosses.ssim(x, y), olosses.ms_ssim(x, y)
    assert ((s > 0) & (s < 1)).all() and ((m > 0) & (m < 1)).all()


def test_oracle_constant_images_closed_form():
    # constant images: variances vanish, cs = 1, ss = (2ab + c1) / (a^2 + b^2 + c1).  fp32 evaluates E[x^2] - mu^2 with
    # ~1e-7 of cancellation noise against c2 = 9e-4, hence the 1e-4 tolerance (inherent to the formula, piqa included)
    a, b = 0.3, 0.8
    x, y = torch.full((1, 3, 40, 40), a), torch.full((1, 3, 40, 40), b)
    want = (2 * a * b + 1e-4) / (a * a + b * b + 1e-4)
    assert abs(olosses.ssim(x, y).item() - want) < 1e-4
    x, y = torch.full((1, 3, 176, 176), a), torch.full((1, 3, 176, 176), b)
    assert abs(olosses.ms_ssim(x, y).item() - want ** 0.1333) < 1e-4     # only the last scale carries luminance


def test_oracle_size_limits():
    x = torch.rand(1, 3, 161, 161)
    olosses.ms_ssim(x, x)                       # 161 -> 81 -> 41 -> 21 -> 11
    with pytest.raises(RuntimeError):
        olosses.ms_ssim(x[..., :160, :], x[..., :160, :])
    with pytest.raises(RuntimeError):
        olosses.ssim(x[..., :10, :], x[..., :10, :])


# ---------------------------------------------------------------------------- host helpers (CPU)

def test_sort_isos_and_baseline(tmp_path):
    b, o = dataset_torch_3.sortISOs(["ISO6400", "ISO200", "ISOH1", "ISO800", "ISO200-1"])
    assert b == ["ISO200", "ISO200-1"] and o == ["ISO800", "ISO6400", "ISOH1"]
    b, o = dataset_torch_3.sortISOs(["ISO200-2", "ISO200", "ISO3200-1", "ISO3200", "ISO200-1"])
    assert b[0] == "ISO200" and sorted(b[1:]) == ["ISO200-1", "ISO200-2"] and sorted(o) == ["ISO3200", "ISO3200-1"]
    assert dataset_torch_3.sortISOs(["GT1", "noisy2", "noisy1"]) == (["GT1"], ["noisy1", "noisy2"])
    assert dataset_torch_3.sortISOs(["b", "a", "c"]) == (["a"], ["b", "c"])
    assert dataset_torch_3.sortISOs(["ISO100"]) == (["ISO100"], [])
    d = tmp_path / "banana"
    d.mkdir()
    for iso in ("ISO200", "ISO6400", "ISOH1"):
        (d / f"NIND_banana_{iso}.png").write_bytes(b"")
    assert dataset_torch_3.get_baseline_fpath(str(d)) == str(d / "NIND_banana_ISO200.png")


def test_json_saver_and_averages(tmp_path):
    assert utilities.avg_listofdicts([{"a": 1.0, "b": 2.0}, {"a": 3.0, "b": 6.0}]) == {"a": 2.0, "b": 4.0}
    fp = str(tmp_path / "res.json")
    js = json_saver.JSONSaver(fp, step_type="epoch")
    js.add_res(step=3, res={"mse": 0.5, "ssim": 0.2}, key_prefix="test_")
    js.add_res(step=4, res={"mse": 0.4, "ssim": 0.3}, key_prefix="test_")
    d = json.load(open(fp))
    assert d["3"] == {"test_mse": 0.5, "test_ssim": 0.2} and d["best_epoch"] == {"test_mse": 4, "test_ssim": 3}
    assert d["best_val"] == {"test_mse": 0.4, "test_ssim": 0.2}
    js2 = json_saver.JSONSaver(fp, step_type="epoch")        # reload: digit keys come back as ints
    assert 3 in js2.results and js2.get_best_steps() == {3, 4}
    with pytest.raises(ValueError):
        js2.add_res(step=None, res={})


# ---------------------------------------------------------------------------- HIP scores vs the oracle (GPU)

@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need a real MI355X")
    from nind_denoise_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


SCORE_TOL = 2e-5   # fp32 scores in [0, 1]; the summation orders differ (tile partials vs torch's mean)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 11, 11), (2, 3, 64, 75), (1, 1, 43, 200), (3, 3, 177, 161), (1, 3, 1000, 1500)])
def test_ssim_matches_oracle(dev, shape):
    from nind_denoise_amd.common.libs import pt_losses
    x, y = _pair(*shape, seed=shape[2])
    got = pt_losses.SSIM_loss()(x.to(dev), y.to(dev)).cpu()
    want = 1 - olosses.ssim(x, y)
    assert got.shape == want.shape and (got - want).abs().max().item() < SCORE_TOL, (got, want)
    assert pt_losses.SSIM_loss()(x.to(dev), x.to(dev)).abs().max().item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 161, 161), (2, 3, 184, 184), (1, 3, 163, 301), (1, 1, 700, 525), (1, 3, 1000, 1500)])
def test_ms_ssim_matches_oracle(dev, shape):
    from nind_denoise_amd.common.libs import pt_losses
    x, y = _pair(*shape, seed=shape[3], noise=0.2)
    got = pt_losses.MS_SSIM_loss()(x.to(dev), y.to(dev)).cpu()
    want = 1 - olosses.ms_ssim(x, y)
    assert got.shape == want.shape and (got - want).abs().max().item() < SCORE_TOL, (got, want)
    assert pt_losses.MS_SSIM_loss()(y.to(dev), y.to(dev)).abs().max().item() < 1e-5


@pytest.mark.gpu
def test_mse_and_error_paths(dev):
    from nind_denoise_amd.common.libs import pt_losses
    x, y = _pair(1, 3, 333, 517, 5)
    assert abs(pt_losses.mse(x.to(dev), y.to(dev)).item() - torch.nn.functional.mse_loss(x, y).item()) < 1e-7
    with pytest.raises(ValueError, match="too small"):     # piqa raises from its convolution on the same inputs
        pt_losses.MS_SSIM_loss()(x[..., :160, :].to(dev), y[..., :160, :].to(dev))     # e.g. the 128 / 136-pixel crops
    with pytest.raises(ValueError, match="too small"):
        pt_losses.SSIM_loss()(x[..., :10].to(dev), y[..., :10].to(dev))
    with pytest.raises(RuntimeError, match="GPU only"):
        pt_losses.SSIM_loss()(x, y)
    with pytest.raises(ValueError):
        pt_losses.SSIM_loss()(x.to(dev), y[..., :100].to(dev))


@pytest.mark.gpu
def test_denoise_dir_end_to_end(dev, tmp_path):
    """A two-set test directory through denoise_dir: files written, per-image scores equal the oracle's on the same files,
    results recorded under the 'test_' prefix next to the model."""
    from nind_denoise_amd import denoise_dir
    from nind_denoise_amd.common.libs import imgcodec, np_imgops
    from oracle import networks as onet
    from oracle import tiler as otiler
    sd = synth.make_utnet_state_dict(funit=16, seed=4)
    mdir = tmp_path / "models" / "run_utnet"
    mdir.mkdir(parents=True)
    torch.save(sd, str(mdir / "generator_7.pt"))
    json.dump({"best_epoch": {"validation_loss": 7}, "best_val": {}}, open(mdir / "trainres.json", "w"))
    noisy = tmp_path / "ds" / "NIND_120_88"
    rng = np.random.default_rng(0)
    for aset, (w, h) in (("bike", (230, 200)), ("tree", (250, 190))):
        (noisy / aset).mkdir(parents=True)
        clean = synth.make_frame(w, h, seed=len(aset))
        for iso, sigma in (("ISO200", 0.0), ("ISO3200", 0.05), ("ISOH1", 0.1)):
            img = np.clip(clean + sigma * rng.standard_normal(clean.shape).astype(np.float32), 0, 1)
            imgcodec.write_png(str(noisy / aset / f"NIND_{aset}_{iso}.png"), (img * 65535).round().astype(np.uint16).transpose(1, 2, 0))
    res = denoise_dir.main(["--model_path", str(mdir / "generator_7.pt"), "--network", "UtNet", "--model_parameters", "funit=16",
                            "--cs", "120", "--ucs", "88", "-ol", "16", "--noisy_dir", str(noisy),
                            "--result_dir", str(tmp_path / "results"), "--config", "/nonexistent.yaml"])
    outdir = tmp_path / "results" / "run_utnet"
    outs = sorted(os.listdir(outdir))
    assert outs == ["NIND_bike_ISO3200.png", "NIND_bike_ISOH1.png", "NIND_tree_ISO3200.png", "NIND_tree_ISOH1.png"]
    # oracle pass over the same files: CPU tiler + CPU network + CPU scores
    per_set = []
    for aset in ("bike", "tree"):
        per_img = []
        base = torch.from_numpy(np_imgops.img_path_to_np_flt(str(noisy / aset / f"NIND_{aset}_ISO200.png")))[None]
        for iso in ("ISO3200", "ISOH1"):
            frame = np_imgops.img_path_to_np_flt(str(noisy / aset / f"NIND_{aset}_{iso}.png"))
            with torch.no_grad():
                ref = otiler.denoise_frame(frame, 120, 88, 16, lambda t: onet.utnet_forward(sd, torch.from_numpy(t)).numpy(), batch=4)
            got = np_imgops.img_path_to_np_flt(str(outdir / f"NIND_{aset}_{iso}.png"))
            assert np.abs(got - np.clip(ref, 0, 1)).max() <= 1.0 / 65535 + 1e-3     # 16-bit file of the denoised frame
            per_img.append(olosses.get_losses(base, torch.from_numpy(got)[None]))
        per_set.append(utilities.avg_listofdicts(per_img))
    want = utilities.avg_listofdicts(per_set)
    for k in ("mse", "ssim", "msssim"):
        assert abs(res[k] - want[k]) < SCORE_TOL, (k, res, want)
    for fn in ("trainres.json", "testres.json"):
        d = json.load(open(mdir / fn))
        assert abs(d["7"]["test_msssim"] - want["msssim"]) < SCORE_TOL and d["best_epoch"]["test_mse"] == 7


@pytest.mark.gpu
@pytest.mark.parametrize("multiscale,shape", [(False, (2, 3, 40, 57)), (False, (1, 3, 184, 184)), (True, (2, 3, 184, 184)),
                                                (True, (1, 3, 163, 201)), (True, (1, 1, 330, 169))])
def test_ssim_losses_backward_vs_oracle_autograd(dev, multiscale, shape):
    """The scores as training criterions (nn_common.py:170-177): gradient with respect to the generated batch against torch
    autograd through the oracle, per-sample loss vector weighted like `(loss * weight).mean().backward()`."""
    from nind_denoise_amd.common.libs import pt_losses
    x, y = _pair(*shape, seed=shape[3], noise=0.15)
    wvec = torch.linspace(0.5, 1.5, shape[0])
    xr = x.clone().requires_grad_()
    lref = 1 - (olosses.ms_ssim(xr, y) if multiscale else olosses.ssim(xr, y))
    (lref * wvec).mean().backward()
    xd = x.to(dev).requires_grad_()
    crit = pt_losses.MS_SSIM_loss() if multiscale else pt_losses.SSIM_loss()
    l = crit(xd, y.to(dev))
    (l * wvec.to(dev)).mean().backward()
    assert (l.detach().cpu() - lref.detach()).abs().max().item() < SCORE_TOL
    g, gref = xd.grad.cpu(), xr.grad
    scale = gref.abs().max().item()
    err = (g - gref).abs().max().item() / scale
    assert torch.isfinite(g).all() and err < 2e-4, (err, scale)
