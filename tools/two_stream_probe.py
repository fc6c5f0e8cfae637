#!/usr/bin/env python3
"""Probe: one conv-stack launch of N tiles against two launches of N/2 tiles in flight on two streams (separate workspaces and
canvases), whole G24 frame.  GPU box only.  If the HBM-bound transform passes of one half run under the persistent MFMA-bound
GEMMs of the other, two streams finish a frame sooner."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nind_denoise_amd import _lib, pipeline, synth  # noqa: E402
from nind_denoise_amd.networks.UtNet import UtNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--frames", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    W, H, cs, ucs, ol = 6000, 4000, 264, 200, 64
    sd = synth.make_utnet_state_dict(64, 123)
    nets = []
    for _ in range(2):      # two module instances = two workspaces; the packed blob is per instance too
        n = UtNet(64)
        n.load_state_dict(sd)
        nets.append(n.eval().to(dev).set_compute_dtype(a.dtype))
    img = torch.from_numpy(synth.make_frame(W, H, seed=24)).to(dev)
    total = pipeline.tile_count(W, H, cs, ucs, ol)
    cv = [torch.zeros_like(img) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    lib = _lib.load()

    def one_stream(batch):
        cv[0].zero_()
        pipeline.denoise_frame(nets[0], img, cs, ucs, ol, batch=batch, canvas=cv[0])

    def two_streams(batch):
        half = batch // 2
        for c in cv:
            c.zero_()
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        ranges = [(t0, min(total, t0 + half)) for t0 in range(0, total, half)]
        for k, (lo, hi) in enumerate(ranges):
            with torch.cuda.stream(streams[k % 2]):
                pipeline.denoise_frame(nets[k % 2], img, cs, ucs, ol, batch=half, tile_range=(lo, hi), canvas=cv[k % 2])
        for s in streams:
            cur.wait_stream(s)

    for name, fn in (("one stream", one_stream), ("two streams", two_streams)):
        fn(a.batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.frames):
            fn(a.batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.frames
        print(f"{name:12s} {a.batch} tiles in flight: {dt * 1e3:8.2f} ms per frame = {W * H / 1e6 / dt:7.2f} MP/s", flush=True)
    ref = cv[0] + cv[1]
    one_stream(a.batch)
    torch.cuda.synchronize()
    print("max |two-stream canvas sum - one-stream canvas| =", float((ref - cv[0]).abs().max()))


if __name__ == "__main__":
    main()
