#!/usr/bin/env python3
"""Per-layer table of one conv-stack launch (GPU box only): kernel form, ms, the three-pass split (input transform / GEMMs /
output transform) with the transform passes' achieved TB/s, executed and algorithmic TFLOP/s.

    python tools/stack_table.py [--cs 264] [--batch 256] [--dtype f32] [--flags 0] [--reps 3]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cs", type=int, default=264)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--crop", type=int, default=0, help="margin of the useful tile centre ((cs - ucs) / 2 of the denoise loop); 0: whole tiles")
    a = ap.parse_args()
    from nind_denoise_amd.networks.UtNet import UtNet
    torch.manual_seed(123)
    dev = torch.device("cuda:0")
    net = UtNet(64, "PReLU").to(dev)
    net.compute_dtype = a.dtype
    net.split_k = not (a.flags & 1)
    net.winograd = not (a.flags & 2)
    net.w1d_regs = bool(a.flags & 4)
    net.fused_pool = not (a.flags & 16)
    steps = bench.conv_stack_profile(net, a.cs, a.batch, dev, reps=a.reps, crop=a.crop)
    tot = 0.0
    print(f"{'layer':12s} {'form':12s} {'ms':>8s} {'xf_in':>7s} {'TB/s':>5s} {'gemm':>7s} {'xf_out':>7s} {'TB/s':>5s} {'exec TF':>8s} {'alg TF':>8s}")
    for s in steps:
        tot += s["ms"]
        bi = s["xform_bytes_in"] / s["ms_xform_in"] / 1e9 if s["ms_xform_in"] > 0 else 0.0
        bo = s["xform_bytes_out"] / s["ms_xform_out"] / 1e9 if s["ms_xform_out"] > 0 else 0.0
        gemm_ms = s["ms_gemm"] if s["ms_gemm"] > 0 else s["ms"]
        print(f"{s['name']:12s} {s['form']:12s} {s['ms']:8.3f} {s['ms_xform_in']:7.3f} {bi:5.2f} {s['ms_gemm']:7.3f} {s['ms_xform_out']:7.3f} {bo:5.2f} "
              f"{s['mfma_flop'] / gemm_ms / 1e9:8.1f} {s['flop'] / s['ms'] / 1e9:8.1f}")
    print(f"total {tot:.3f} ms per launch of {a.batch} tiles")


if __name__ == "__main__":
    main()
