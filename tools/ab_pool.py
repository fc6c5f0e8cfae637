import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from nind_denoise_amd import pipeline, synth
from nind_denoise_amd.networks.UtNet import UtNet
dev = torch.device("cuda:0")
for dtype, geom, batch in (("bf16", (6000, 4000, 264, 200, 64), 160), ("f16", (9504, 6336, 520, 456, 64), 40)):
    W, H, cs, ucs, ol = geom
    net = UtNet(64); net.load_state_dict(synth.make_utnet_state_dict(64, 123)); net = net.eval().to(dev).set_compute_dtype(dtype)
    img = torch.from_numpy(synth.make_frame(W, H, seed=24)).to(dev); cv = torch.zeros_like(img)
    def run(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            cv.zero_(); pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch, canvas=cv)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    run(2)
    for rep in range(2):
        for fused in (True, False):
            net.fused_pool = fused
            run(1)
            print(dtype, "fused" if fused else "separate", f"{run(4):.2f} ms/frame", flush=True)
    del net, img, cv
    torch.cuda.empty_cache()
