import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from nind_denoise_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
names = [lib.nd_conv_variant_name(v).decode() for v in range(lib.nd_num_conv_variants())]
ws = torch.empty(int(60e9), dtype=torch.uint8, device=dev)
ms = ctypes.c_float()
def run(dt, cin, cout, var, batch=64, h=132):
    rc = lib.nd_conv_bench(_lib.KIND["conv1"], _lib.DTYPE[dt], batch, cin, cout, h, h, var, 5, ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev), ms)
    return ms.value if rc == 0 else float("nan")
f32v = names.index("f32_m2x4_n4x2_t1_k4_upfalse_s2")
for cin, cout in ((256, 256), (512, 512), (1024, 512)):
    t32 = run("f32", cin, cout, f32v)
    outs = []
    for v, n in enumerate(names):
        if n.startswith("bf16") and "_t1_" in n and "upfalse" in n:
            outs.append((n, run("bf16", 6 * cin, cout, v)))
    best = min(outs, key=lambda x: x[1])
    flop = 2.0 * 64 * 132 * 132 * cin * cout
    print(f"K={cin} M={cout}: fp32 GEMM {t32:.3f} ms ({flop / t32 / 1e9:.1f} TF); bf16 with 6x the K: best {best[0]} {best[1]:.3f} ms ({flop / best[1] / 1e9:.1f} fp32-equivalent TF) -> x{t32 / best[1]:.2f}", flush=True)
