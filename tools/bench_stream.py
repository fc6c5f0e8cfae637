#!/usr/bin/env python3
"""PCIe-inclusive throughput of the resident multi-frame engine (nind_denoise_amd.serve.FrameEngine): frames start in
pageable host memory and end in host memory; host->HBM, compute and HBM->host overlap across frames.

    python tools/bench_stream.py [--frames 8] [--dtype f32|bf16|f16] [--batch 160]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nind_denoise_amd import synth  # noqa: E402
from nind_denoise_amd.networks.UtNet import UtNet  # noqa: E402
from nind_denoise_amd.serve import FrameEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=6000)
    ap.add_argument("--height", type=int, default=4000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    net = UtNet()
    net.load_state_dict(synth.make_utnet_state_dict(64, 123))
    net = net.eval().to(dev).set_compute_dtype(args.dtype)
    W, H = args.width, args.height
    base = synth.make_frame(W, H, seed=24)
    frames = [np.roll(base, k * 17, axis=2) for k in range(3)]   # a few distinct frames, reused
    eng = FrameEngine(net, W, H, 264, 200, 64, batch=args.batch, slots=3, device=dev)
    list(eng.run(frames[:2]))   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for out in eng.run(frames[i % 3] for i in range(args.frames)):
        n += 1
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": "host->host stream, PCIe inclusive", "dtype": args.dtype, "frames": n,
                      "MP_per_s": round(W * H / 1e6 * n / dt, 3), "ms_per_frame": round(1e3 * dt / n, 2)}))


if __name__ == "__main__":
    main()
