#!/usr/bin/env python3
"""Headline benchmark: megapixels/s denoised, UtNet(64,'PReLU') fp32, 24 MP synthetic frames, cs=264/ucs=200/ol=64.

    python bench.py --gpus N --steps K --warmup W            (N > 1: this process spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (also fine: ranks from the env)

A "step" is one 6000x4000 frame through the device-resident crop -> UtNet -> stitch loop (BASELINE.json configs[1],
geometry "G24" of SURVEY.md section 8: cs=264 is the valid tile size nearest to the named 256, which the reference
network itself rejects).  The frame is resident in HBM (on rank 0) when the timed region starts and the stitched canvas is in
HBM (on rank 0) when it ends.

N > 1 (default): north_star's partition -- the TILE loop of every frame is sharded over the ranks (contiguous tile-index
shards, denoise_image.py:240-267).  Rank 0 owns the model and the frames: it broadcasts the raw parameters once (124 MB; every
rank packs them into MFMA fragment order on its own device), and per frame sends each rank the image rows its shard reads and
receives + adds the canvas rows it wrote (RCCL point-to-point over xGMI).  The stream of frames is software-pipelined
(nind_denoise_amd/dist.py: ShardedFrameStream): the scatter of frame n+1 and the gather of frame n-1 run on a side stream under
the compute of frame n.  Total work per step is fixed as N grows -> "scaling": "strong"; `value` = megapixels of the frames that
left rank 0 / max-over-ranks time.  The same line carries, under `frame_shard`, the rate of N independent REPLICAS (one whole
frame per rank per step, no data-path traffic) -- labelled as such, not the metric.
`--frames F` times BASELINE configs[2]'s shape instead (a step = F frames dealt round-robin to the ranks, no per-frame exchange);
`--frame-per-rank` makes the replicas mode the timed one.

The timed loop is the product's default: every tile's whole cs x cs crop goes in, every layer runs, and the layers of the last
decoder levels compute only the outputs that the tile's useful centre (what the stitch keeps) depends on; `--whole-tiles` times
the loop with whole-tile layers instead, and the default run reports that rate and the difference of the two canvases under
`whole_tiles`.

One JSON line on rank 0:
  roofline      the dominant kernel by time (conv_w2d, the fused 1-D Winograd 3x3 kernel on the fp32 path): MFMA-EXECUTED
                FLOP / HIP-event time on the launch stream / MFMA peak (<= 1 by construction); the algorithmic
                (direct-convolution) rate is reported beside it, and `families` prices every kernel family against its own
                bound (transform passes against HBM)
  parity        the TIMED path (pipeline.denoise_frame: fused gather -> useful-region conv stack -> stitch, whole launches of
                `batch` tiles) against the oracle's stitched canvas of the tiles the cpu_baseline leg computes anyway
  cpu_baseline  the oracle (torch CPU fp32, the primitives the reference runs) on a bounded sample of the same frame's tiles,
                in grad mode (as the reference runs, denoise_image.py:246) and under no_grad
  other_configs (N = 1) short legs of the other single-GPU configurations BASELINE.json names -- bf16 G24 (configs[2]'s
                per-GPU work), fp16 G61 at cs=520 (configs[3]), fp32 at the shipped default tiling (G24d) -- each with its
                rate, its dominant kernel's roofline, HBM traffic from the matching stored PMC summary, and parity against
                the oracle on tiles of its timed launch shape
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)" 157.3 TFLOP/s; "Peak BF16/FP16 MFMA ~2.5 PF dense"; HBM 8 TB/s
PEAK_MFMA_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}
PEAK_HBM_TBS = 8.0
METRIC = "megapixels/sec denoised, UtNet cs=256 on 24 MP frames, 1/2/4/8 MI355X"
# ND_BENCH_REHEARSAL=1: rehearse the N > 1 code path on a box with fewer GPUs -- every rank on GPU 0, messages over gloo
# (host-staged; RCCL refuses two ranks on one device).  Exercises the launcher, the shard loop and the exchange logic, not
# the transport; its numbers are not benchmark results and the JSON line says so.
REHEARSAL = os.environ.get("ND_BENCH_REHEARSAL", "") == "1"


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="tiles per launch of the conv stack")
    ap.add_argument("--width", type=int, default=6000)
    ap.add_argument("--height", type=int, default=4000)
    ap.add_argument("--cs", type=int, default=264)
    ap.add_argument("--ucs", type=int, default=200)
    ap.add_argument("--ol", type=int, default=64)
    ap.add_argument("--funit", type=int, default=64)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"],
                    help="storage inside the conv stack; f32 is the headline configuration (exact-fp32 MFMA), bf16 / f16 are "
                         "BASELINE configs 2 / 3 (16-bit storage, fp32 accumulate)")
    ap.add_argument("--frames", type=int, default=0,
                    help="> 0: BASELINE configs[2]'s shape -- a step is this many frames, dealt round-robin to the ranks "
                         "(frame-level sharding, no per-frame exchange)")
    ap.add_argument("--tile-shard", action="store_true", help="(the default for N > 1; kept for older command lines)")
    ap.add_argument("--frame-per-rank", action="store_true",
                    help="N > 1: time N independent replicas (one whole frame per rank per step, no exchange) instead of the tile-sharded stream")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the bf16 G24 / fp16 G61 / fp32 G24d legs of the default single-GPU run")
    ap.add_argument("--cpu-sample-tiles", type=int, default=0,
                    help="0: sized for ~10 s of CPU work per mode; -1: the WHOLE frame, un-extrapolated (minutes: for a kept record)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch CPU threads of the baseline leg (0: all the cgroup allows, max 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-winograd", action="store_true", help="direct convolution on every layer (A/B switch)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host->host (PCIe-inclusive) measurement")
    ap.add_argument("--no-whole-leg", action="store_true", help="skip the comparison leg that runs the same frame with whole-tile layers")
    ap.add_argument("--whole-tiles", action="store_true",
                    help="compute every layer on the whole tile, as UtNet.forward does (ND_FLAG_FULL_TILES); default: the last decoder "
                         "levels compute only what the useful centre [pad, cs - pad) of a tile depends on -- same canvas")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ rank launcher

def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start N fresh rank processes BEFORE this process touches the GPU, relay rank 0's
    JSON line, fail if any rank fails.  (Never re-exec a process that has initialised the GPU.)"""
    import torch   # importing torch and counting devices does not initialise the GPU
    have = torch.cuda.device_count()
    if have < n and not REHEARSAL:
        sys.exit(f"bench.py --gpus {n}: only {have} GPU(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # relay rank 0's stdout; if any rank dies, end the others (by their exact pids) instead of leaving them in a collective
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            time.sleep(5)      # let the failing rank's message reach stderr, and the others fail on their own if they will
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    # rank 0's stdout carries the one JSON line; anything else a library printed there (gloo's connection banner) goes to stderr
    for ln in b"".join(c for c in chunks if c).decode(errors="replace").splitlines():
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    if failed or any(rcs):
        sys.exit(f"bench.py: rank exit codes {rcs}")


# ------------------------------------------------------------------------------------------------ roofline

def conv_stack_profile(net, cs, batch, dev, reps=3, crop=0):
    """HIP-event timing of every launch of the conv stack for one batch (median over reps); the library reports which kernel
    family ran each layer and the FLOP its matrix cores executed (nd_step_profile, include/nind_hip.h)."""
    import numpy as np
    from nind_denoise_amd import _lib
    lib = _lib.load()
    blob = net.packed_weights(dev)
    ws = net.workspace(cs, batch, dev)
    n = 26
    arr = (_lib.StepProfile * n)()
    runs = []
    for _ in range(reps + 1):
        _lib.check(lib.nd_utnet_profile_stack(net.funit, _lib.ACT[net.activation], _lib.DTYPE[net.compute_dtype], net.flags,
                                              blob.data_ptr(), batch, cs, crop, ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev),
                                              ctypes.cast(arr, ctypes.c_void_p), n))
        runs.append([(a.ms, a.ms_xform_in, a.ms_gemm, a.ms_xform_out) for a in arr])
    med = np.median(np.array(runs[1:]), axis=0)
    steps = []
    for i, a in enumerate(arr):
        steps.append(dict(name=lib.nd_utnet_step_name(i).decode(), form=_lib.FORM_NAMES[a.form], kind=a.kind, ms=float(med[i][0]),
                          ms_xform_in=float(med[i][1]), ms_gemm=float(med[i][2]), ms_xform_out=float(med[i][3]), flop=a.flops,
                          mfma_flop=a.mfma_flops, bytes=a.bytes, xform_bytes_in=a.xform_bytes_in, xform_bytes_out=a.xform_bytes_out))
    return steps


def pmc_traffic(kernel_regex, cs, batch, funit, dtype):
    """HBM bytes per launch of one kernel family from the newest committed rocprofv3 PMC summary of the same configuration
    (FETCH_SIZE and WRITE_SIZE collected in separate passes, gfx950 read correction applied -- profiles/*_pmc_summary.json).
    Counters cannot be read inside the timed run, so the figure comes from a stored profile and says so; None otherwise."""
    import glob
    import re
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        c = d.get("config", {})
        if (c.get("cs"), c.get("tiles_per_launch"), c.get("funit"), c.get("dtype", "f32")) != (cs, batch, funit, dtype):
            continue
        rd = wr = n = 0
        for name, v in d.get("kernels", {}).items():
            if re.match(kernel_regex, name) and "hbm_read_bytes_mean" in v and "hbm_write_bytes_mean" in v:
                k = v["FETCH_SIZE"]["dispatches"]
                rd += v["hbm_read_bytes_mean"] * k
                wr += v["hbm_write_bytes_mean"] * k
                n += k
        if n:
            return {"bytes_per_launch": round((rd + wr) / n), "read": round(rd / n), "write": round(wr / n),
                    "launches_profiled": n, "source": os.path.relpath(f, ROOT) + " (stored profile of this configuration, not this run)"}
    return None


def roofline_report(steps, dtype, cs, batch, funit):
    peak = PEAK_MFMA_TFLOPS[dtype]
    conv = [s for s in steps if s["form"] != "pool"]
    fam = {}

    def add(key, bound, ms, flop=0.0, mfma=0.0, byts=0.0, launches=1):
        f = fam.setdefault(key, dict(bound=bound, ms=0.0, algorithmic_flop=0.0, mfma_flop=0.0, bytes=0.0, launches=0))
        f["ms"] += ms
        f["algorithmic_flop"] += flop
        f["mfma_flop"] += mfma
        f["bytes"] += byts
        f["launches"] += launches

    for i, s in enumerate(steps):
        # a direct 3x3 layer whose pool step took no time of its own wrote the pooled tensor from its epilogue (16-bit storage)
        s["pools"] = s["form"] == "direct" and i + 1 < len(steps) and steps[i + 1]["form"] == "pool" and steps[i + 1]["ms"] <= 0.02
    for s in steps:
        if s["form"] == "pool":
            # (fp32 default path: MaxPool2d(2) is fused into the producing layer's epilogue -- no launch, no time of its own)
            if s["ms"] > 0.02:
                add("k_maxpool2", "hbm", s["ms"], byts=s["bytes"])
        elif s["form"] in ("w1d_f43", "w1d_f23"):
            add("conv_w2d (3x3, 1-D Winograd F(4,3), input transform shared through LDS)", "mfma", s["ms"], s["flop"], s["mfma_flop"], s["bytes"])
        elif s["form"] == "wino3p_f6x6":
            if s["ms_gemm"] > 0:
                add("conv_qp 1-tap (64 GEMMs of a three-pass Winograd F(6x6,3x3) layer)", "mfma", s["ms_gemm"], s["flop"], s["mfma_flop"],
                    s["xform_bytes_in"] + s["xform_bytes_out"] - s["bytes"])
                add("k_wino_input + k_wino_output (transform passes)", "hbm", s["ms_xform_in"] + s["ms_xform_out"],
                    byts=s["xform_bytes_in"] + s["xform_bytes_out"], launches=2)
            else:   # batches above one Winograd chunk: no split of the layer's time
                add("three-pass Winograd layers (transforms + GEMMs)", "mfma", s["ms"], s["flop"], s["mfma_flop"], s["bytes"], launches=3)
        elif s["kind"] == 2:
            add("conv_qp up (ConvTranspose2d 2x2 s2)", "mfma", s["ms"], s["flop"], s["mfma_flop"], s["bytes"])
        elif s["pools"]:
            add("conv_qp direct 3x3 + fused 2x2 max pool (2-row band pixel order, quad maxima in the epilogue)", "mfma", s["ms"], s["flop"], s["mfma_flop"], s["bytes"])
        else:
            add("conv_qp direct 3x3", "mfma", s["ms"], s["flop"], s["mfma_flop"], s["bytes"])
    total_ms = sum(s["ms"] for s in steps)
    families = []
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        row = dict(kernel=k, bound=f["bound"], ms_per_batch=round(f["ms"], 4), share_of_stack=round(f["ms"] / total_ms, 4), launches=f["launches"])
        if f["bound"] == "mfma":
            ach = f["mfma_flop"] / (f["ms"] * 1e-3) / 1e12
            row.update(achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                       algorithmic_tflops=round(f["algorithmic_flop"] / (f["ms"] * 1e-3) / 1e12, 2))
        else:
            ach = f["bytes"] / (f["ms"] * 1e-3) / 1e12
            row.update(achieved=round(ach, 3), peak=PEAK_HBM_TBS, unit="TB/s", frac=round(ach / PEAK_HBM_TBS, 4),
                       algorithmic_bytes_per_batch=round(f["bytes"]))
        families.append(row)
    # the dominant kernel by time among the MFMA families
    dom = max((r for r in families if r["bound"] == "mfma"), key=lambda r: r["ms_per_batch"])
    f = fam[dom["kernel"]]
    dt_idx = {"f32": 0, "bf16": 1, "f16": 2}[dtype]
    if dom["kernel"].startswith("conv_w2d"):
        rx = r"conv_w2d<"
    elif dom["kernel"].startswith("conv_qp direct 3x3"):
        rx = rf"conv_qp<{dt_idx}, \d, \d, \d, \d, 9, "
    elif dom["kernel"].startswith("conv_qp 1-tap"):
        rx = rf"conv_qp<{dt_idx}, \d, \d, \d, \d, 1, \d, false"
    else:
        rx = r"$^"
    traffic = pmc_traffic(rx, cs, batch, funit, dtype)
    stack_alg = sum(s["flop"] for s in conv) / (sum(s["ms"] for s in conv) * 1e-3) / 1e12
    stack_exe = sum(s["mfma_flop"] for s in conv) / (sum(s["ms"] for s in conv) * 1e-3) / 1e12
    return {
        "bound": "mfma",
        "kernel": dom["kernel"],
        "achieved": dom["achieved"],
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": dom["frac"],
        "definition": "MFMA-executed FLOP of the dominant kernel's launches (MFMA instructions x 4096) / their HIP-event time on the "
                      "launch stream / peak; the algorithmic (direct-convolution, SURVEY.md 8d) rate is algorithmic_tflops",
        "launches": f["launches"],
        "avg_launch_ms": round(f["ms"] / f["launches"], 4),
        "mfma_flop_per_launch": f["mfma_flop"] / f["launches"],
        "algorithmic_flop_per_launch": f["algorithmic_flop"] / f["launches"],
        "algorithmic_bytes_per_launch": round(f["bytes"] / f["launches"]),
        "algorithmic_tflops": dom["algorithmic_tflops"],
        "algorithmic_speedup": round(f["algorithmic_flop"] / f["mfma_flop"], 4),
        "traffic": traffic["bytes_per_launch"] if traffic else None,
        "traffic_detail": traffic,
        "share_of_stack_time": dom["share_of_stack"],
        "tiles_per_launch": batch,
        "stack": {"ms_per_batch": round(total_ms, 4), "algorithmic_tflops": round(stack_alg, 2), "mfma_executed_tflops": round(stack_exe, 2),
                  "mfma_executed_frac": round(stack_exe / peak, 4), "algorithmic_speedup": round(stack_alg / stack_exe, 4)},
        "families": families,
        "per_layer": [dict(name=s["name"], form=s["form"], ms=round(s["ms"], 4), algorithmic_tflops=round(s["flop"] / max(s["ms"], 1e-9) / 1e9, 2),
                           mfma_tflops=round(s["mfma_flop"] / max(s["ms"], 1e-9) / 1e9, 2)) for s in steps],
    }


# ------------------------------------------------------------------------------------------------ CPU baseline + parity

def sample_runs(total, n, run=8):
    """Tile ids of the CPU leg: runs of `run` consecutive tiles spread over the frame (the first run starts at tile 0, the last
    ends at the frame's last tile), ~n tiles in all.  Runs, not isolated tiles: a run's stitched canvas has pixels that no tile
    outside the run touches, which is what lets the parity leg compare CANVASES of the timed path (fused stitch included)."""
    run = max(1, min(run, total))
    nruns = max(1, n // run)
    starts = [0] if nruns == 1 else [round(k * (total - run) / (nruns - 1)) for k in range(nruns)]
    return sorted({i for s0 in starts for i in range(s0, s0 + run)})


def cpu_baseline(frame, sd, cs, ucs, ol, ids, threads, grad_mode):
    """The oracle's crop -> UtNet -> stitch on the tiles `ids` of the frame (torch CPU fp32).  Returns (seconds, outputs, tiles)."""
    import numpy as np
    import torch
    from oracle import networks as onet
    from oracle import tiler as otiler
    torch.set_num_threads(threads)
    grid = otiler.TileGrid(frame.shape[2], frame.shape[1], cs, ucs, ol)
    canvas = np.zeros_like(frame)
    outs = []
    with torch.set_grad_enabled(grad_mode):
        if grad_mode:   # the reference's parameters are nn.Parameters: autograd records the forward (denoise_image.py:246)
            sd = {k: v.clone().requires_grad_() for k, v in sd.items()}
        onet.utnet_forward(sd, torch.from_numpy(otiler.gather_tile(frame, grid, 0))[None])  # warm-up
        t0 = time.perf_counter()
        for i in ids:
            x = torch.from_numpy(otiler.gather_tile(frame, grid, i))[None]
            y = onet.utnet_forward(sd, x).detach().numpy()
            otiler.stitch_add(canvas, y[0], grid, i)
            outs.append(y[0])
        dt = time.perf_counter() - t0
    return dt, outs, grid.size


def canvas_parity(net, img, frame_shape, cs, ucs, ol, batch, ids, outs):
    """The TIMED path against the oracle.  For every launch of `batch` tiles that holds sampled tiles, pipeline.denoise_frame
    (fused gather -> conv stack with useful-region layers -> fused stitch, exactly the timed call) stitches the WHOLE launch
    into a zero canvas; the oracle stitches its outputs of the sampled tiles; the two canvases are compared on the pixels that
    sampled tiles cover and no other tile of the launch touches."""
    import numpy as np
    import torch
    from nind_denoise_amd import pipeline
    from oracle import tiler as otiler
    _, H, W = frame_shape
    grid = otiler.TileGrid(W, H, cs, ucs, ol)
    total = grid.size
    by_launch = {}
    for k, i in enumerate(ids):
        by_launch.setdefault(i // batch, []).append((k, i))
    worst = scale = 0.0
    sse = 0.0
    npix = 0
    lo_v, hi_v = float("inf"), float("-inf")
    for launch, members in sorted(by_launch.items()):
        a, b = launch * batch, min(total, (launch + 1) * batch)
        cv = torch.zeros((3, H, W), dtype=torch.float32, device=img.device)
        pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch, tile_range=(a, b), canvas=cv)
        ref = np.zeros((3, H, W), dtype=np.float32)
        mine = np.zeros((H, W), dtype=bool)
        other = np.zeros((H, W), dtype=bool)
        sampled = {i for _, i in members}
        for k, i in members:
            otiler.stitch_add(ref, outs[k], grid, i)
        for t in range(a, b):
            _, _, ud, us = grid.geom(t)
            (mine if t in sampled else other)[us[1]:us[1] + ud[3] - ud[1], us[0]:us[0] + ud[2] - ud[0]] = True
        m = mine & ~other
        ys, xs = np.nonzero(m.any(axis=1))[0], np.nonzero(m.any(axis=0))[0]
        if len(ys) == 0:
            continue
        y0, y1, x0, x1 = ys[0], ys[-1] + 1, xs[0], xs[-1] + 1
        got = cv[:, y0:y1, x0:x1].cpu().numpy()
        r = ref[:, y0:y1, x0:x1]
        mm = np.broadcast_to(m[y0:y1, x0:x1], got.shape)
        d = np.abs(got - r)[mm]
        worst = max(worst, float(d.max()))
        scale = max(scale, float(np.abs(r[mm]).max()))
        sse += float((d.astype(np.float64) ** 2).sum())
        npix += int(d.size)
        lo_v, hi_v = min(lo_v, float(r[mm].min())), max(hi_v, float(r[mm].max()))
        del cv
    mse = sse / max(npix, 1)
    psnr = 10 * math.log10((hi_v - lo_v) ** 2 / max(mse, 1e-30)) if npix else float("nan")
    return {"max_abs": worst, "max_abs_ref": scale, "psnr_db": round(psnr, 2), "tiles": len(ids), "canvas_values_compared": npix,
            "launches": len(by_launch)}


# ------------------------------------------------------------------------------------------------ single-GPU legs

def time_frames(net, img, canvas, cs, ucs, ol, batch, steps, warmup):
    import torch
    from nind_denoise_amd import pipeline
    for _ in range(warmup):
        canvas.zero_()
        pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch, canvas=canvas)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        canvas.zero_()
        pipeline.denoise_frame(net, img, cs, ucs, ol, batch=batch, canvas=canvas)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


OTHER_CONFIGS = [
    # key, BASELINE config, (W, H, cs, ucs, ol), dtype, tiles per launch, parity bar (PSNR dB against the fp32 oracle; None: the fp32 bar)
    ("bf16_g24", "configs[2] per-GPU work: 24 MP frame, bf16 storage", (6000, 4000, 264, 200, 64), "bf16", 320, 65.0),
    ("f16_g61", "configs[3]: 61 MP frame (9504x6336), cs=520 (nearest valid to 512), fp16 storage", (9504, 6336, 520, 456, 64), "f16", 80, 85.0),
    ("f32_g24d", "configs[1] at the shipped default tiling (CS_UTNET, UCS_UTNET = 504, 480, overlap 6)", (6000, 4000, 504, 480, 6), "f32", 64, None),
]


def other_config_leg(key, what, geom, dtype, batch, bar, sd, funit, dev, threads, steps):
    """One short leg of another single-GPU configuration: rate, dominant-kernel roofline, parity of the timed path."""
    import torch
    from nind_denoise_amd import synth
    from nind_denoise_amd.networks.UtNet import UtNet
    W, H, cs, ucs, ol = geom
    net = UtNet(funit=funit)
    net.load_state_dict(sd)
    net = net.eval().to(dev).set_compute_dtype(dtype)
    frame_np = synth.make_frame(W, H, seed=61 if W != 6000 else 24)
    img = torch.from_numpy(frame_np).to(dev)
    canvas = torch.zeros_like(img)
    dt = time_frames(net, img, canvas, cs, ucs, ol, batch, steps, 1)
    mp = W * H / 1e6
    from nind_denoise_amd import pipeline
    total = pipeline.tile_count(W, H, cs, ucs, ol)
    row = {"what": what, "value": round(mp * steps / dt, 4), "unit": "MP/s", "dtype": dtype, "ms_per_frame": round(1e3 * dt / steps, 3),
           "frames": steps, "geometry": {"width": W, "height": H, "cs": cs, "ucs": ucs, "ol": ol, "tiles_per_frame": total,
                                         "tiles_per_launch": batch}}
    b = min(batch, total)
    prof = conv_stack_profile(net, cs, b, dev, crop=(cs - ucs) // 2)
    rf = roofline_report(prof, dtype, cs, b, funit)
    for k in ("per_layer", "definition"):
        rf.pop(k, None)
    row["roofline"] = rf
    # parity: two runs of 8 tiles (first tiles of the first launch, last tiles of the frame) through the oracle
    ids = sample_runs(total, 16, run=8)
    cdt, outs, _ = cpu_baseline(frame_np, sd, cs, ucs, ol, ids, threads, False)
    par = canvas_parity(net, img, frame_np.shape, cs, ucs, ol, batch, ids, outs)
    if bar is None:
        par["bar"] = "fp32: max_abs <= 1e-3 and <= 1e-3 * max_abs_ref"
        par["ok"] = bool(par["max_abs"] <= 1e-3 and par["max_abs"] <= 1e-3 * par["max_abs_ref"])
    else:
        par["bar"] = f"16-bit storage against the fp32 oracle: PSNR >= {bar} dB"
        par["ok"] = bool(par["psnr_db"] >= bar)
    par["note"] = (f"timed path (pipeline.denoise_frame, {batch} tiles per launch, useful-region layers, fused stitch) vs the oracle's "
                   f"stitched canvas of {len(ids)} tiles ({cdt:.1f} s of torch CPU)")
    row["parity"] = par
    del net, img, canvas
    torch.cuda.empty_cache()
    return row


# ------------------------------------------------------------------------------------------------ main

def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
        return
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist
    from nind_denoise_amd import _lib, pipeline, synth
    from nind_denoise_amd import dist as ndist
    from nind_denoise_amd.networks.UtNet import UtNet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if REHEARSAL:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (a rank that dies takes the others out of their collectives within this timeout instead of hanging them)
        tmo = datetime.timedelta(seconds=300)
        if REHEARSAL:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
    _lib.load()

    W, H, cs, ucs, ol = args.width, args.height, args.cs, args.ucs, args.ol
    # rank 0 owns the model (seeded synthetic weights with the reference's state-dict layout); the other ranks start from
    # different values and receive the raw parameters by broadcast, then pack them on their own device
    sd = synth.make_utnet_state_dict(funit=args.funit, seed=123 if rank == 0 else 1000 + rank)
    net = UtNet(funit=args.funit)
    net.load_state_dict(sd)
    net = net.eval().to(dev).set_compute_dtype(args.dtype)
    net.winograd = not args.no_winograd
    net.useful_only = not args.whole_tiles
    bcast = None
    if world > 1:
        t0 = time.perf_counter()
        nbytes = ndist.broadcast_parameters(net, src=0)
        torch.cuda.synchronize()
        bcast = {"bytes": nbytes, "seconds": round(time.perf_counter() - t0, 4),
                 "what": "raw state-dict tensors as one flat fp32 buffer from rank 0; every rank packs them on its own device"}
    net.packed_weights(dev)

    geo = ndist.Geo(W, H, cs, ucs, ol)
    total = geo.total
    mp = W * H / 1e6
    frame_np = None

    def compute(fr, cv, a, b):
        pipeline.denoise_frame(net, fr, cs, ucs, ol, batch=args.batch, tile_range=(a, b), canvas=cv)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    mode = "single" if world == 1 and args.frames == 0 else ("frames" if args.frames > 0 else ("replicas" if args.frame_per_rank else "tile-stream"))
    lo, hi = geo.shard(rank, world) if mode == "tile-stream" else (0, total)
    # a few distinct synthetic frames resident in HBM are cycled (the arithmetic does not depend on the pixel values)
    if rank == 0 or mode in ("frames", "replicas"):
        frame_np = synth.make_frame(W, H, seed=24)
        pool = [torch.from_numpy(frame_np).to(dev)] + [torch.from_numpy(synth.make_frame(W, H, seed=s_)).to(dev) for s_ in (25, 26)]
    else:
        pool = []
    canvas = torch.zeros((3, H, W), dtype=torch.float32, device=dev)

    def replica_steps(n, frames_in_step):
        mine = ndist.frame_shard(frames_in_step, rank, world)
        for _ in range(n):
            for k, _f in enumerate(mine):
                canvas.zero_()
                compute(pool[k % len(pool)], canvas, 0, total)

    stream = None
    if mode == "tile-stream":
        stream = ndist.ShardedFrameStream(compute, geo, dev)

        def run_steps(n):
            src = (pool[k % len(pool)] for k in range(n)) if rank == 0 else None
            for _k, _cv in stream.run(src, n):
                pass
        frames_per_step = 1
    elif mode == "single":
        def run_steps(n):
            for _ in range(n):
                canvas.zero_()
                compute(pool[0], canvas, 0, total)
        frames_per_step = 1
    else:
        frames_per_step = args.frames if mode == "frames" else world

        def run_steps(n):
            replica_steps(n, frames_per_step)

    log(f"setup done: {total} tiles/frame, mode {mode}, rank tile shard [{lo},{hi}), batch {args.batch}, frames/step {frames_per_step}")
    if args.warmup > 0:
        run_steps(args.warmup)
        torch.cuda.synchronize()
        log(f"{args.warmup} warmup step(s) done")
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed {args.steps} steps in {dt:.3f} s")

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if REHEARSAL else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = max_over_ranks(dt)
    extra = {}
    if mode == "tile-stream":
        # compute-only time of this rank's shard (no exchange): what a step would cost if the exchange were free
        compute(pool[0] if rank == 0 else stream.frames[0], stream.canvas[0], lo, hi)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nc = max(2, min(4, args.steps))
        for _ in range(nc):
            compute(pool[0] if rank == 0 else stream.frames[0], stream.canvas[0], lo, hi)
        torch.cuda.synchronize()
        tc = max_over_ranks((time.perf_counter() - t1) / nc)
        extra["pipeline"] = {"ms_per_step": round(1e3 * dt / args.steps, 3), "compute_only_ms_largest_shard": round(1e3 * tc, 3),
                             "exchange_hidden_frac": round(min(1.0, tc / (dt / args.steps)), 4),
                             "note": "compute_only = this mode's per-rank shard loop with no exchange, max over ranks; a step of the pipelined "
                                     "stream costs max(compute, exchange) + fill/drain of two exchange steps per timed region"}
        # secondary key: N independent replicas (one whole frame per rank per step; no data-path traffic) -- NOT the metric
        if rank != 0:
            frame_np = synth.make_frame(W, H, seed=24)
            pool = [torch.from_numpy(frame_np).to(dev)]
        nr = max(1, min(3, args.steps))
        replica_steps(1, world)
        barrier()
        t2 = time.perf_counter()
        replica_steps(nr, world)
        barrier()
        tr = max_over_ranks(time.perf_counter() - t2)
        extra["frame_shard"] = {"value": round(mp * world * nr / tr, 4), "unit": "MP/s", "frames_per_step": world, "steps": nr,
                                "note": "REPLICAS: one whole frame per rank per step, no data-path collective -- reported for reference, not the metric"}
    if world > 1 and rank != 0:
        # the post-timing legs are rank 0's alone and contain no collective: the other ranks are done
        dist.barrier()
        dist.destroy_process_group()
        return
    if world > 1:
        dist.barrier()

    value = mp * frames_per_step * args.steps / dt
    cfg_idx = 1 if args.dtype == "f32" else 2 if args.dtype == "bf16" else 3
    if mode == "frames":
        par = f"frame-shard x{world} (frames dealt round-robin, no per-frame exchange)" if world > 1 else "single GPU"
        what = f"{args.frames} {W}x{H} ({mp:.1f} MP) fp32 frames per step"
        scaling = "strong" if world > 1 else "none"
    elif mode == "replicas":
        par = f"replicas x{world} (one whole frame per rank per step, no data-path collective)"
        what = f"{world} {W}x{H} ({mp:.1f} MP) fp32 frames per step, one per rank"
        scaling = "weak"
    elif mode == "tile-stream":
        par = (f"tile-shard x{world}: every frame's tile loop split into {world} contiguous tile-index shards; rank 0 scatters input row bands and "
               "gathers + adds canvas row bands point-to-point (RCCL over xGMI), pipelined across frames (scatter of n+1 / gather of n-1 under "
               "the compute of n); weights broadcast once as raw parameters")
        what = f"one {W}x{H} ({mp:.1f} MP) fp32 frame per step, entering and leaving through rank 0"
        scaling = "strong"
    else:
        par = "single GPU"
        what = f"one {W}x{H} ({mp:.1f} MP) fp32 frame per step"
        scaling = "none"
    out = {
        "metric": METRIC,
        "value": round(value, 4),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic" + (" -- REHEARSAL: all ranks on one GPU over gloo, not a benchmark result" if REHEARSAL else ""),
        "config": {
            "workload": f"configs[{cfg_idx}]: {what}, {args.dtype} storage in the conv stack (fp32 accumulate), UtNet(funit={args.funit},PReLU) "
                        f"random-init (seed 123), cs={cs} (nearest valid to the named size; the reference rejects 256 / 512) ucs={ucs} ol={ol} -> "
                        f"{total} tiles/frame, tiles per conv-stack launch {args.batch}, crop->infer->stitch device resident (frame and canvas in HBM)"
                        + ("" if args.whole_tiles else "; every tile's input is the whole cs x cs crop, and the layers of the last decoder levels compute "
                           "only the outputs that the tile's useful centre (the part the stitch keeps, denoise_image.py:249-258) depends on -- the canvas "
                           "is the same as with whole-tile layers (key `whole_tiles`)"),
            "tiles_per_frame": total,
            "frames_per_step": frames_per_step,
            "flop_per_frame": net.flops_per_tile(cs) * total,
            "parallelism": par,
        },
    }
    out.update(extra)
    if bcast:
        out["weight_broadcast"] = bcast

    flop_frame = net.flops_per_tile(cs) * total
    out["end_to_end_algorithmic_tflops"] = round(flop_frame * frames_per_step * args.steps / dt / 1e12, 3)
    if not args.no_roofline:
        # (N > 1: rank 0 profiles the conv stack at the launch size of its own tile shard)
        b = min(args.batch, hi - lo)
        steps = conv_stack_profile(net, cs, b, dev, crop=(cs - ucs) // 2)
        log("conv stack profile done")
        out["roofline"] = roofline_report(steps, args.dtype, cs, b, args.funit)
        out["config"]["computed_flop_per_frame"] = sum(s_["flop"] for s_ in steps) / b * total   # conv stack, regions counted as computed
    if mode == "single" and not args.whole_tiles and not args.no_whole_leg:
        # the same frame with every layer on whole tiles (what UtNet.forward computes): its rate, and the two canvases compared
        cv_roi = canvas.clone()
        net.useful_only = False
        nw = max(2, min(4, args.steps))
        tw = time_frames(net, pool[0], canvas, cs, ucs, ol, args.batch, nw, 1)
        diff = float((canvas - cv_roi).abs().max().item())
        out["whole_tiles"] = {"value": round(mp * nw / tw, 4), "unit": "MP/s", "frames": nw,
                              "canvas_max_abs_diff": diff, "canvas_max_abs": float(cv_roi.abs().max().item()),
                              "note": "ND_FLAG_FULL_TILES: every layer computes its whole tile; canvas_max_abs_diff = max |canvas - canvas of the "
                                      "timed (useful-region) loop| over the whole frame"}
        net.useful_only = True
        del cv_roi
        log(f"whole-tile leg: {out['whole_tiles']['value']} MP/s, canvases differ by {diff:.2e}")
    if not args.no_host_leg and mode == "single":
        # SURVEY.md 8(d)'s end-to-end definition: decoded fp32 frame in pinned host memory -> stitched frame in host memory,
        # through the resident engine (H2D / compute / D2H overlapped over a ring of 3 slots).  Reported beside `value`.
        from nind_denoise_amd.serve import FrameEngine
        eng = FrameEngine(net, W, H, cs, ucs, ol, batch=args.batch, slots=3, device=dev)
        src = torch.from_numpy(frame_np).pin_memory()
        n_host = max(4, min(12, args.steps))
        for _ in eng.run([src] * 2):
            pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cnt = 0
        for _ in eng.run([src] * n_host, copy=False):
            cnt += 1
        dth = time.perf_counter() - t1
        out["host_to_host"] = {"value": round(mp * cnt / dth, 4), "unit": "MP/s", "frames": cnt,
                               "note": "pinned host frame -> HBM -> crop/UtNet/stitch -> pinned host canvas, PCIe legs overlapped "
                                       "with compute (serve.FrameEngine); `value` above is the HBM-resident rate"}
        del eng
        log("host-to-host leg done")
    threads = args.cpu_threads or min(host_cores(), 64)
    if not args.no_cpu_baseline and world == 1:
        n = args.cpu_sample_tiles or max(8, min(200, threads * 6))   # ~10 s of CPU work per mode at ~0.1 s/tile
        ids = list(range(total)) if n < 0 else sample_runs(total, n, run=8)
        n = len(ids)
        log(f"cpu baseline: {n} tiles on {threads} threads, grad mode then no_grad")
        cdt_g, _, tot = cpu_baseline(frame_np, sd, cs, ucs, ol, ids, threads, True)
        cdt_n, outs, _ = cpu_baseline(frame_np, sd, cs, ucs, ol, ids, threads, False)
        log(f"cpu baseline done in {cdt_g:.1f} + {cdt_n:.1f} s")
        out["cpu_baseline"] = {
            "value": round(mp * (n / tot) / cdt_g, 5),
            "unit": "MP/s",
            "cores": threads,
            "kind": "port",
            "sample": (f"the whole frame, un-extrapolated: all {tot} tiles" if n == tot else f"{n} of {tot} tiles of the same frame (runs of 8 consecutive tiles)")
                      + f" through oracle gather -> UtNet (torch CPU fp32, {threads} threads) -> stitch" + ("" if n == tot else ", scaled by tiles")
                      + f"; grad mode as the reference runs (denoise_image.py:246, no no_grad) in {cdt_g:.2f} s",
            "no_grad": {"value": round(mp * (n / tot) / cdt_n, 5), "seconds": round(cdt_n, 2)},
        }
        par_ = canvas_parity(net, pool[0], frame_np.shape, cs, ucs, ol, args.batch, ids, outs)
        par_["bar"] = "fp32: max_abs <= 1e-3 and <= 1e-3 * max_abs_ref" if args.dtype == "f32" else "16-bit storage: PSNR >= 60 dB against the fp32 oracle"
        par_["ok"] = bool(par_["max_abs"] <= 1e-3 and par_["max_abs"] <= 1e-3 * par_["max_abs_ref"]) if args.dtype == "f32" else bool(par_["psnr_db"] >= 60.0)
        par_["note"] = (f"the TIMED path -- pipeline.denoise_frame: fused gather -> conv stack ("
                        + ("whole-tile layers" if args.whole_tiles else "useful-region layers") + f") -> fused stitch, whole launches of {args.batch} tiles "
                        "-- stitched into a zero canvas, against the oracle's stitched canvas of the cpu_baseline tiles, on the pixels only those tiles touch")
        out["parity"] = par_
        log(f"parity on {n} tiles ({par_['canvas_values_compared']} canvas values): max abs {par_['max_abs']:.3e}")
    if mode == "single" and not args.no_other_configs and args.dtype == "f32" and (W, H, cs, ucs, ol) == (6000, 4000, 264, 200, 64):
        del pool, canvas
        net._workspaces.clear()
        torch.cuda.empty_cache()
        out["other_configs"] = {}
        sd0 = synth.make_utnet_state_dict(funit=args.funit, seed=123)
        for key, what_, geom, dtype_, batch_, bar in OTHER_CONFIGS:
            out["other_configs"][key] = other_config_leg(key, what_, geom, dtype_, batch_, bar, sd0, args.funit, dev, threads, steps=3)
            log(f"other config {key}: {out['other_configs'][key]['value']} MP/s, dominant kernel frac "
                f"{out['other_configs'][key]['roofline']['frac']}, parity {out['other_configs'][key]['parity']}")
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
