#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernel variants on the UtNet layer shapes (GPU box only).

    python tools/bench_layers.py [--batch 32] [--cs 264] [--variants 0,1,2,3] [--layers tconvs4.0,...]
Prints one line per (layer, variant): mean launch ms and algorithmic TFLOP/s (SURVEY.md 2a convention)."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nind_denoise_amd import _lib  # noqa: E402


def utnet_shapes(cs, f=64):
    """(name, kind, cin, cout, h_in, w_in) with h_in the UNBORDERED input size of the reference layer."""
    out = []
    h = cs + 4
    for n, (ci, co) in enumerate([(3, f), (f, 2 * f), (2 * f, 4 * f), (4 * f, 8 * f)], start=1):
        out.append((f"convs{n}.0", "conv3", ci, co, h))
        out.append((f"convs{n}.2", "conv3", co, co, h - 2))
        h = (h - 4) // 2
    out.append(("bottom.0", "conv3", 8 * f, 16 * f, h))
    out.append(("bottom.2", "convT3", 16 * f, 16 * f, h - 2))
    c = 16 * f
    for n in range(1, 5):
        out.append((f"up{n}", "convT2s2", c, c // 2, h))
        h *= 2
        out.append((f"tconvs{n}.0", "convT3", c, c // 2, h))
        out.append((f"tconvs{n}.2", "convT3", c // 2, c // 2, h + 2))
        h += 4
        c //= 2
    return out


def flops(kind, cin, cout, h, batch):
    if kind == "conv3":
        return 2.0 * (h - 2) ** 2 * cin * cout * 9 * batch
    if kind == "convT3":
        return 2.0 * h * h * cin * cout * 9 * batch
    return 2.0 * h * h * cin * cout * 4 * batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cs", type=int, default=264)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--variants", default="auto")
    ap.add_argument("--layers", default="")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--winograd", action="store_true", help="also time the Winograd F(2,3) / F(4,3) form of every eligible 3x3 layer")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    names = [lib.nd_conv_variant_name(v).decode() for v in range(lib.nd_num_conv_variants())]
    ws = torch.empty(int(40e9 if args.winograd else 6e9), dtype=torch.uint8, device=dev)
    want = [s for s in args.layers.split(",") if s]
    tot = {}
    for (name, kind, cin, cout, h) in utnet_shapes(args.cs):
        if want and name not in want:
            continue
        k = _lib.KIND[kind]
        cands = [-1] if args.variants == "auto" else [int(v) for v in args.variants.split(",")]
        for v in cands:
            if v >= 0:
                nm = names[v]
                if not nm.startswith(args.dtype + "_"):
                    continue
                if "_t4_" in nm:
                    continue
                taps = 9 if "_t9_" in nm else 1
                if (taps == 9) != (kind in ("conv3", "convT3")) or ("_uptrue" in nm) != (kind == "convT2s2"):
                    continue
            ms = ctypes.c_float()
            rc = lib.nd_conv_bench(k, _lib.DTYPE[args.dtype], args.batch, cin, cout, h, h, v, args.iters, ws.data_ptr(), ws.numel(),
                                   _lib.stream_ptr(dev), ms)
            if rc != 0:
                print(f"{name:10s} v{v:<2d} skipped: {lib.nd_last_error().decode()}")
                continue
            tf = flops(kind, cin, cout, h, args.batch) / (ms.value * 1e-3) / 1e12
            print(f"{name:10s} {kind:8s} {cin:4d}->{cout:4d} {h:3d}^2  v{v:<2d} {names[v] if v >= 0 else 'auto':34s} {ms.value:8.4f} ms {tf:7.2f} TF", flush=True)
            tot.setdefault(v, 0.0)
            tot[v] += ms.value
            if v == -1 and args.winograd and kind in ("conv3", "convT3") and args.dtype == "f32":
                for tile in (3, 5, 4, 6):   # 3 = 1-D F(4,3) in registers (conv_w1d), 5 = the same with the LDS-shared transform (conv_w2d), 4 | 6 = three-pass F(4x4,3x3) | F(6x6,3x3)
                    if tile % 2 == 0 and cin % 16:
                        continue
                    rc = lib.nd_winograd_bench(tile, k, args.batch, cin, cout, h, h, args.iters, ws.data_ptr(), ws.numel(),
                                               _lib.stream_ptr(dev), ms)
                    if rc != 0:
                        print(f"{name:10s} F({tile},3) skipped: {lib.nd_last_error().decode()}")
                        continue
                    tf = flops(kind, cin, cout, h, args.batch) / (ms.value * 1e-3) / 1e12
                    print(f"{name:10s} {kind:8s} {cin:4d}->{cout:4d} {h:3d}^2  winograd F({tile},3){'':20s} {ms.value:8.4f} ms {tf:7.2f} TF (algorithmic)", flush=True)
                    tot.setdefault(f"F{tile}", 0.0)
                    tot[f"F{tile}"] += ms.value
    print("sum ms per variant:", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
