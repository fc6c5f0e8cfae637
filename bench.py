#!/usr/bin/env python3
"""Headline benchmark: megapixels/s denoised, UtNet(64,'PReLU') fp32, 24 MP synthetic frames, cs=264/ucs=200/ol=64.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one 6000x4000 frame through the device-resident crop -> UtNet -> stitch loop (BASELINE.json configs[1],
geometry "G24" of SURVEY.md section 8: cs=264 is the valid tile size nearest to the named 256, which the reference
network itself rejects).  The frame is resident in HBM when the timed region starts.  With N > 1 the tile index range
of every frame is split into N contiguous shards (one rank per GPU); inside the timed region rank 0 sends every rank
the image rows its shard reads (RCCL point-to-point over xGMI), every rank denoises its shard into its own canvas and
rank 0 receives and adds the canvas row bands (nind_denoise_amd/dist.py) -> total work per step is fixed: "strong".

One JSON line on rank 0.  `roofline` prices the conv_qp_f32 kernel family (all 22 MFMA conv launches of the stack)
against the fp32 MFMA peak with HIP events recorded on the launch stream; `cpu_baseline` times the oracle (torch CPU
fp32, the same primitives the reference runs) on a bounded sample of the same frame's tiles.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)" 157.3 TFLOP/s; "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_MFMA_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}


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
                         "BASELINE configs 3 / 4 (16-bit storage, fp32 accumulate)")
    ap.add_argument("--cpu-sample-tiles", type=int, default=0, help="0: sized for ~15 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-winograd", action="store_true", help="direct convolution on every layer (A/B switch)")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def lib_winograd_on():
    from nind_denoise_amd import _lib
    lib = _lib.load()
    was = lib.nd_conv_winograd_enable(1)
    lib.nd_conv_winograd_enable(was)
    return bool(was)


def conv_stack_profile(net, cs, batch, dev, reps=3):
    """HIP-event timing of every launch of the conv stack for one batch (median over reps)."""
    import numpy as np
    import torch
    from nind_denoise_amd import _lib
    lib = _lib.load()
    blob = net.packed_weights(dev)
    ws = net.workspace(cs, batch, dev)
    n = 26
    ms = (ctypes.c_float * n)()
    fl = (ctypes.c_double * n)()
    isc = (ctypes.c_int * n)()
    runs = []
    for _ in range(reps + 1):
        _lib.check(lib.nd_utnet_profile_stack(net.funit, _lib.ACT[net.activation], _lib.DTYPE[net.compute_dtype], blob.data_ptr(), batch, cs,
                                              ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev), ms, fl, isc, n))
        runs.append(list(ms))
    med = np.median(np.array(runs[1:]), axis=0)
    steps = [dict(name=lib.nd_utnet_step_name(i).decode(), ms=float(med[i]), flop=float(fl[i]), conv=bool(isc[i]))
             for i in range(n)]
    return steps


def executed_flop(name, flop, cs, batch, funit, dtype):
    """MFMA-executed FLOP of one conv-stack step (csrc/utnet_net.h: wino_layer): 36 GEMMs of Cout x Cin x tiles for the 3x3 layers in
    three-pass Winograd F(4x4,3x3) form, 18 weight planes per group of 4 pixels for the ones in fused 1-D F(4,3) form, the
    algorithmic FLOP for everything else."""
    import math
    f, h = funit, cs + 4
    shapes = {}
    for n, (ci, co) in enumerate([(3, f), (f, 2 * f), (2 * f, 4 * f), (4 * f, 8 * f)], start=1):
        shapes[f"convs{n}.0"] = (ci, co, h - 2)
        shapes[f"convs{n}.2"] = (co, co, h - 4)
        h = (h - 4) // 2
    shapes["bottom.0"] = (8 * f, 16 * f, h - 2)
    shapes["bottom.2"] = (16 * f, 16 * f, h)
    c = 16 * f
    for n in range(1, 5):
        h *= 2
        shapes[f"tconvs{n}.0"] = (c, c // 2, h + 2)
        shapes[f"tconvs{n}.2"] = (c // 2, c // 2, h + 4)
        h += 4
        c //= 2
    if name not in shapes or dtype != "f32":
        return flop
    ci, co, hout = shapes[name]
    if ci >= 128 and co >= 128 and ci * co >= 128 * 256:
        return 36 * 2.0 * ci * co * math.ceil(hout / 4) ** 2 * batch            # three-pass F(4x4,3x3): 36 GEMMs
    cip = (ci + 7) // 8 * 8
    return 3 * 6 * 2.0 * cip * co * hout * math.ceil(hout / 4) * batch           # 1-D F(4,3) in the implicit-GEMM kernel: 18 planes per 4 pixels


def pmc_traffic(cs, batch, funit):
    """HBM bytes per launch of the dominant kernel (conv_w1d, the fused 1-D Winograd 3x3 kernel) from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE collected in separate runs, gfx950 read correction applied -- profiles/*_pmc_summary.json).
    Counters cannot be read inside the timed run, so this is null unless a profile of the same configuration exists."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        c = d.get("config", {})
        if (c.get("cs"), c.get("tiles_per_launch"), c.get("funit")) != (cs, batch, funit):
            continue
        rd = wr = n = 0
        dominant = [n for n in d.get("kernels", {}) if n.startswith("conv_w1d<")] or \
                   [n for n in d.get("kernels", {}) if n.startswith("conv_qp<0,") and ", 9, 1, false" in n]
        for name, v in d.get("kernels", {}).items():   # the dominant kernel (fused 1-D Winograd 3x3; else the direct 3x3 variants)
            if name in dominant and "hbm_read_bytes_mean" in v:
                k = v["FETCH_SIZE"]["dispatches"]
                rd += v["hbm_read_bytes_mean"] * k
                wr += v["hbm_write_bytes_mean"] * k
                n += k
        if n:
            return {"bytes_per_launch": round((rd + wr) / n), "read": round(rd / n), "write": round(wr / n),
                    "launches_profiled": n, "source": os.path.relpath(f, ROOT)}
    return None


def cpu_baseline(frame, sd, cs, ucs, ol, n_tiles, threads):
    """The oracle's crop -> UtNet -> stitch on `n_tiles` tiles of the frame (torch CPU fp32, grad mode off)."""
    import numpy as np
    import torch
    from oracle import networks as onet
    from oracle import tiler as otiler
    torch.set_num_threads(threads)
    grid = otiler.TileGrid(frame.shape[2], frame.shape[1], cs, ucs, ol)
    canvas = np.zeros_like(frame)
    ids = [int(i) for i in np.linspace(0, grid.size - 1, n_tiles)]
    with torch.no_grad():
        onet.utnet_forward(sd, torch.from_numpy(otiler.gather_tile(frame, grid, 0))[None])  # warm-up
        t0 = time.perf_counter()
        for i in ids:
            x = torch.from_numpy(otiler.gather_tile(frame, grid, i))[None]
            y = onet.utnet_forward(sd, x).numpy()
            otiler.stitch_add(canvas, y[0], grid, i)
        dt = time.perf_counter() - t0
    return dt, grid.size


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from nind_denoise_amd import _lib, pipeline, synth
    from nind_denoise_amd.networks.UtNet import UtNet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    _lib.load()
    if args.no_winograd:
        _lib.load().nd_conv_winograd_enable(0)

    W, H, cs, ucs, ol = args.width, args.height, args.cs, args.ucs, args.ol
    sd = synth.make_utnet_state_dict(funit=args.funit, seed=123)
    net = UtNet(funit=args.funit)
    net.load_state_dict(sd)
    net = net.eval().to(dev).set_compute_dtype(args.dtype)
    blob = net.packed_weights(dev)
    if world > 1:
        dist.broadcast(blob, src=0)  # one-time weight broadcast (rank 0 is the model owner)

    frame_np = synth.make_frame(W, H, seed=24) if rank == 0 else None
    frame = torch.from_numpy(frame_np).to(dev) if rank == 0 else torch.empty((3, H, W), dtype=torch.float32, device=dev)
    from nind_denoise_amd import dist as ndist
    geo = ndist.Geo(W, H, cs, ucs, ol)
    total = geo.total
    lo, hi = geo.shard(rank, world)
    canvas = torch.zeros((3, H, W), dtype=torch.float32, device=dev)

    def compute(fr, cv, a, b):
        pipeline.denoise_frame(net, fr, cs, ucs, ol, batch=args.batch, tile_range=(a, b), canvas=cv)

    def step():
        if world > 1:
            # rank 0 scatters input row bands, every rank denoises its tile shard, rank 0 gathers + adds the bands
            ndist.denoise_frame_sharded(compute, frame, canvas, geo)
        else:
            canvas.zero_()
            compute(frame, canvas, 0, total)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"setup done: {total} tiles/frame, rank shard [{lo},{hi}), batch {args.batch}")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed {args.steps} steps in {dt:.3f} s")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    mp = W * H / 1e6
    value = mp * args.steps / dt
    out = {
        "metric": "megapixels/sec denoised, UtNet cs=256 on 24 MP frames, 1/2/4/8 MI355X",
        "value": round(value, 4),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"configs[{1 if args.dtype == 'f32' else 2 if args.dtype == 'bf16' else 3}]: one {W}x{H} ({mp:.1f} MP) fp32 frame per step, "
                        f"{args.dtype} storage in the conv stack (fp32 accumulate), UtNet(funit={args.funit},PReLU) random-init "
                        f"(seed 123), cs={cs} (nearest valid to 256; the reference rejects 256) ucs={ucs} ol={ol} -> {total} tiles, "
                        f"tiles per conv-stack launch {args.batch}, crop->infer->stitch device resident",
            "tiles_per_frame": total,
            "flop_per_frame": net.flops_per_tile(cs) * total,
            "parallelism": f"tile-shard x{world}" if world > 1 else "single GPU",
        },
    }

    if rank == 0:
        flop_frame = net.flops_per_tile(cs) * total
        out["end_to_end_tflops"] = round(flop_frame * args.steps / dt / 1e12, 3)
        if not args.no_roofline:
            # (N > 1: rank 0 profiles the conv stack at the size of its own tile shard; the other ranks wait at the end)
            b = min(args.batch, hi - lo)
            steps = conv_stack_profile(net, cs, b, dev)
            log("conv stack profile done")
            conv_ms = sum(s["ms"] for s in steps if s["conv"])
            conv_flop = sum(s["flop"] for s in steps if s["conv"])
            achieved = conv_flop / (conv_ms * 1e-3) / 1e12
            wino = args.dtype == "f32" and lib_winograd_on()
            exe_flop = sum(executed_flop(s["name"], s["flop"], cs, b, args.funit, args.dtype if wino else "") for s in steps if s["conv"])
            out["roofline"] = {
                "bound": "mfma",
                "kernel": f"conv_qp<{args.dtype}>: the 22 weighted layers of the UtNet conv stack per tile batch"
                          + (" -- 1-D Winograd F(4,3) inside the 3x3 implicit-GEMM kernel (7 layers), 36 batched 1-tap GEMMs per three-pass "
                             "Winograd F(4x4,3x3) layer (11 layers, layer time includes the two transform passes), 2x2-s2 transposed "
                             "(4 layers)" if wino else ""),
                "achieved": round(achieved, 3),
                "peak": PEAK_MFMA_TFLOPS[args.dtype],
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_MFMA_TFLOPS[args.dtype], 4),
                "note": "achieved = ALGORITHMIC FLOP (direct-convolution count, SURVEY.md 8d) / time; with Winograd layers the matrix "
                        "cores execute fewer FLOP than that, so the fraction can exceed 1 -- mfma_executed_* is what the MFMA pipe ran",
                "mfma_executed_tflops": round(exe_flop / (conv_ms * 1e-3) / 1e12, 3),
                "mfma_executed_frac": round(exe_flop / (conv_ms * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS[args.dtype], 4),
                "traffic": (pmc_traffic(cs, b, args.funit) or {}).get("bytes_per_launch") if args.dtype == "f32" else None,
                "traffic_unit": "HBM bytes per launch of the dominant kernel, conv_w1d (PMC FETCH_SIZE*2 + WRITE_SIZE)",
                "traffic_detail": pmc_traffic(cs, b, args.funit) if args.dtype == "f32" else None,
                "launches": len([s for s in steps if s["conv"]]),
                "avg_launch_ms": round(conv_ms / max(1, len([s for s in steps if s["conv"]])), 4),
                "algorithmic_flop_per_launch_avg": conv_flop / max(1, len([s for s in steps if s["conv"]])),
                "tiles_per_launch": b,
                "stack_ms_per_batch": round(sum(s["ms"] for s in steps), 4),
                "per_layer": [dict(name=s["name"], ms=round(s["ms"], 4),
                                   tflops=round(s["flop"] / max(s["ms"], 1e-9) / 1e9, 2)) for s in steps],
            }
        if not args.no_cpu_baseline and world == 1:
            threads = min(host_cores(), 64)
            n = args.cpu_sample_tiles or max(8, min(320, threads * 16))   # ~15 s of CPU work at ~0.05 s/tile
            log(f"cpu baseline: {n} tiles on {threads} threads")
            cdt, tot = cpu_baseline(frame_np, sd, cs, ucs, ol, n, threads)
            log(f"cpu baseline done in {cdt:.1f} s")
            out["cpu_baseline"] = {
                "value": round(mp * (n / tot) / cdt, 5),
                "unit": "MP/s",
                "cores": threads,
                "kind": "port",
                "sample": f"{n} of {tot} tiles of the same frame through oracle gather -> UtNet (torch CPU fp32, no_grad, "
                          f"{threads} threads) -> stitch in {cdt:.2f} s, scaled by tiles",
            }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
