#!/usr/bin/env python3
"""Probe (GPU box): two three-pass Winograd layers of the same shape in flight on two streams (two host threads, each inside
nd_winograd_bench) against the same work back to back on one stream.  Tells whether the HBM-bound transform passes of one
launch run under the MFMA-bound GEMMs of the other."""
import ctypes
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nind_denoise_amd import _lib  # noqa: E402

SHAPES = [("convs3.2", "conv3", 256, 256, 62), ("tconvs2.0", "convT3", 512, 256, 60), ("tconvs3.0", "convT3", 256, 128, 128),
          ("tconvs1.0", "convT3", 1024, 512, 28)]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ws = [torch.empty(int(30e9), dtype=torch.uint8, device=dev) for _ in range(2)]
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    batch, iters = 128, 10
    for name, kind, cin, cout, h in SHAPES:
        def run(k, out):
            torch.cuda.set_device(dev)
            ms = ctypes.c_float()
            rc = lib.nd_winograd_bench(6, _lib.KIND[kind], batch, cin, cout, h, h, iters, ws[k].data_ptr(), ws[k].numel(),
                                       ctypes.c_void_p(st[k].cuda_stream), ms)
            out[k] = (rc, ms.value)
        res = {}
        run(0, res)                      # warm-up + single-stream time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(0, res)
        run(1, res)
        torch.cuda.synchronize()
        serial = (time.perf_counter() - t0) * 1e3
        one = res[0][1]
        th = [threading.Thread(target=run, args=(k, res)) for k in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        both = (time.perf_counter() - t0) * 1e3
        print(f"{name:10s} {cin:4d}->{cout:4d} {h:3d}^2 x {batch}: one launch {one:.3f} ms; 2 x {iters} launches back to back {serial:.1f} ms, "
              f"on two streams {both:.1f} ms ({100 * (1 - both / serial):+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
