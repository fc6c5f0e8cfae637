#!/usr/bin/env python3
"""Can an HBM-bound kernel run UNDER the persistent fp32-MFMA GEMM of a three-pass Winograd layer (GPU box only)?
Stream A: the 1-tap conv_qp GEMM (variant: M256 x N256, 4 K blocks per step) on a Cin=Cout=512 problem, back to back.
Stream B: a torch device-to-device copy of a buffer far beyond the caches (few registers, no LDS).
Prints the time of each alone and of both together."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nind_denoise_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    names = [lib.nd_conv_variant_name(v).decode() for v in range(lib.nd_num_conv_variants())]
    var = [i for i, n in enumerate(names) if n == "f32_m2x4_n4x2_t1_k4_upfalse_s2"][0]
    ws = torch.empty(int(8e9), dtype=torch.uint8, device=dev)
    src = torch.empty(int(2e9) // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ms = ctypes.c_float()

    def gemm(iters):
        with torch.cuda.stream(sa):
            _lib.check(lib.nd_conv_bench(_lib.KIND["conv1"], 0, 64, 512, 512, 132, 132, var, iters, ws.data_ptr(), ws.numel(),
                                         ctypes.c_void_p(sa.cuda_stream), ms))
        return ms.value * iters

    def copies(n):
        with torch.cuda.stream(sb):
            for _ in range(n):
                dst.copy_(src)

    gemm(2)
    copies(2)
    torch.cuda.synchronize()
    t = gemm(10)
    print(f"GEMM alone: {t / 10:.3f} ms per launch ({2.0 * 64 * 132 * 132 * 512 * 512 / (t / 10 * 1e-3) / 1e12:.1f} TFLOP/s)")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    copies(20)
    torch.cuda.synchronize()
    tc = (time.perf_counter() - t0) * 1e3
    print(f"copy alone: {tc / 20:.3f} ms per 2 GB copy ({4.0 / (tc / 20 * 1e-3) / 1e3:.2f} TB/s read+write)")
    # together: nd_conv_bench synchronises its stream at the end, so the copies are enqueued first and the GEMM call returns
    # when the GEMMs are done; then wait for the copies
    ng = 10
    ncp = max(1, int(round(t / (tc / 20))))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    copies(ncp)
    tg = gemm(ng)
    t_g_done = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    both = (time.perf_counter() - t0) * 1e3
    print(f"together: {ng} GEMMs ({t:.1f} ms alone) + {ncp} copies ({ncp * tc / 20:.1f} ms alone): GEMM stream done after {t_g_done:.1f} ms "
          f"(its own events: {tg:.1f} ms), everything after {both:.1f} ms; serial sum {t + ncp * tc / 20:.1f} ms")


if __name__ == "__main__":
    main()
