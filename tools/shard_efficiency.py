#!/usr/bin/env python3
"""Per-rank compute time of the tile-sharded frame at N = 1, 2, 4, 8 ranks, measured on ONE GPU (GPU box only): the shard loop of
every rank of an N-rank run, timed alone.  max over shards x N / single-GPU time = the scaling efficiency the partition allows
before any exchange cost."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nind_denoise_amd import dist as ndist, pipeline, synth  # noqa: E402
from nind_denoise_amd.networks.UtNet import UtNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    W, H, cs, ucs, ol = 6000, 4000, 264, 200, 64
    net = UtNet(64)
    net.load_state_dict(synth.make_utnet_state_dict(64, 123))
    net = net.eval().to(dev).set_compute_dtype(a.dtype)
    img = torch.from_numpy(synth.make_frame(W, H, seed=24)).to(dev)
    cv = torch.zeros_like(img)
    geo = ndist.Geo(W, H, cs, ucs, ol)
    out = {}
    for world in (1, 2, 4, 8):
        times = []
        for rank in range(world):
            lo, hi = geo.shard(rank, world)
            def run():
                pipeline.denoise_frame(net, img, cs, ucs, ol, batch=a.batch, tile_range=(lo, hi), canvas=cv)
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / 3 * 1e3)
        out[world] = {"shard_ms": [round(t, 2) for t in times], "max_ms": round(max(times), 2)}
    base = out[1]["max_ms"]
    for world, v in out.items():
        v["speedup_bound"] = round(base / v["max_ms"], 3)
        v["efficiency_bound"] = round(base / v["max_ms"] / world, 4)
    print(json.dumps({"what": f"G24 {a.dtype}, tiles per launch <= {a.batch}: per-rank shard loop timed alone on one GPU", "worlds": out}, indent=1))


if __name__ == "__main__":
    main()
