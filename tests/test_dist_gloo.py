"""Multi-rank tile sharding on gloo/CPU, world_size 2 and 3: the exchange logic of nind_denoise_amd.dist (scatter of
input row bands, per-rank shard loop, gather + ordered band add) with the oracle as the per-rank compute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nind_denoise_amd import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(x):  # cheap deterministic stand-in for the network: [B,3,cs,cs] -> same
    return 0.5 * x + 0.25 * np.roll(x, 1, axis=1) + np.float32(0.01)


def _worker(rank, world, port, geom, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nind_denoise_amd import dist as ndist
        from oracle import tiler as otiler
        W, H, cs, ucs, ol, seed = geom
        geo = ndist.Geo(W, H, cs, ucs, ol)
        grid = otiler.TileGrid(W, H, cs, ucs, ol)
        if rank == 0:
            frame = torch.from_numpy(synth.make_frame(W, H, seed=seed))
        else:
            frame = torch.full((3, H, W), float("nan"))  # rows that are never received must never be read
        canvas = torch.full((3, H, W), 7.0)  # stale content must not leak into the result

        def compute(fr, cv, lo, hi):
            f, c = fr.numpy(), cv.numpy()
            for i in range(lo, hi):
                otiler.stitch_add(c, _model(otiler.gather_tile(f, grid, i)[None])[0], grid, i)

        ndist.denoise_frame_sharded(compute, frame, canvas, geo)
        if rank == 0:
            outq.put(canvas.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,geom", [(2, (500, 700, 264, 200, 64, 1)), (2, (333, 290, 120, 88, 16, 3)),
                                        (3, (777, 333, 104, 72, 10, 5))])
def test_sharded_frame_matches_single_rank(world, geom):
    from oracle import tiler as otiler
    W, H, cs, ucs, ol, seed = geom
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, geom, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = otiler.denoise_frame(synth.make_frame(W, H, seed=seed), cs, ucs, ol, _model, batch=3)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-6   # seam pixels are re-associated (<= 1 ulp), the rest is identical
    seam_free = np.abs(got - ref) == 0
    assert seam_free.mean() > 0.7


def test_shard_geometry_covers_every_tile_once():
    from nind_denoise_amd import dist as ndist
    geo = ndist.Geo(6000, 4000, 264, 200, 64)
    for world in (1, 2, 4, 8):
        edges = [geo.shard(r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == geo.total == 1276
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1
        for a, b in edges:
            y0, y1 = geo.rows_in(a, b)
            o0, o1 = geo.rows_out(a, b)
            assert 0 <= y0 <= o0 < o1 <= y1 <= 4000


def _grad_worker(rank, world, port, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nind_denoise_amd import dist as ndist
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        ndist.average_gradients(flat)
        outq.put((rank, flat.numpy()))
    finally:
        dist.destroy_process_group()


def test_flat_gradient_average_world_3():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(1000, dtype=np.float32) * 2.0      # mean of 1x, 2x, 3x
    for r in range(3):
        assert np.array_equal(got[r], want)


def _frames_worker(rank, world, port, geom, n_frames, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nind_denoise_amd import dist as ndist
        from oracle import tiler as otiler
        W, H, cs, ucs, ol = geom
        # every rank can produce the frames it owns; the others' stay NaN and must never be read
        frames = [torch.from_numpy(synth.make_frame(W, H, seed=100 + f)) if f % world == rank else torch.full((3, H, W), float("nan"))
                  for f in range(n_frames)]

        def denoise(fr):
            return torch.from_numpy(otiler.denoise_frame(fr.numpy(), cs, ucs, ol, _model, batch=4))

        out = ndist.denoise_frames_sharded(denoise, frames)
        assert sorted(ndist.frame_shard(n_frames, rank, world)) == sorted(f for f in range(n_frames) if f % world == rank)
        if rank == 0:
            outq.put({f: c.numpy() for f, c in out.items()})
    finally:
        dist.destroy_process_group()


def test_frame_level_sharding_world_2():
    """BASELINE configs[2]'s shape (a batch of frames over the ranks): frames dealt round-robin, no per-frame exchange, results
    collected on rank 0 -- bit-identical to denoising every frame on one rank."""
    from oracle import tiler as otiler
    geom, n_frames, world = (333, 290, 120, 88, 16), 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frames_worker, args=(r, world, port, geom, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    W, H, cs, ucs, ol = geom
    assert sorted(got) == list(range(n_frames))
    for f in range(n_frames):
        ref = otiler.denoise_frame(synth.make_frame(W, H, seed=100 + f), cs, ucs, ol, _model, batch=4)
        assert np.array_equal(got[f], ref)


def _stream_worker(rank, world, port, geom, n_frames, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nind_denoise_amd import dist as ndist
        from oracle import tiler as otiler
        W, H, cs, ucs, ol = geom
        geo = ndist.Geo(W, H, cs, ucs, ol)
        grid = otiler.TileGrid(W, H, cs, ucs, ol)
        calls = []

        def compute(fr, cv, lo, hi):
            calls.append((lo, hi))
            f, c = fr.numpy(), cv.numpy()
            for i in range(lo, hi):
                otiler.stitch_add(c, _model(otiler.gather_tile(f, grid, i)[None])[0], grid, i)

        stream = ndist.ShardedFrameStream(compute, geo, "cpu")
        # non-root ring buffers start as NaN: rows that are never received must never be read
        if stream.frames is not None:
            for f in stream.frames:
                f.fill_(float("nan"))
        for c in stream.canvas:
            c.fill_(7.0)                # stale content must not leak into a result
        frames = (torch.from_numpy(synth.make_frame(W, H, seed=200 + k)) for k in range(n_frames)) if rank == 0 else None
        got = {}
        for k, cv in stream.run(frames, n_frames):
            if rank == 0:
                got[k] = cv.numpy().copy()      # (a ring slot: copy before two more frames pass)
            else:
                assert cv is None
        assert calls == [geo.shard(rank, world)] * n_frames
        if rank == 0:
            outq.put(got)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 5), (3, 4), (2, 1)])
def test_pipelined_frame_stream_matches_single_rank(world, n_frames):
    """ShardedFrameStream: >= 3 frames in flight through the two-slot rings (scatter of n+1 and gather of n-1 in one grouped
    exchange around the compute of n); every frame's canvas == the single-rank canvas of that frame."""
    from oracle import tiler as otiler
    geom = (333, 290, 120, 88, 16)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, geom, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    W, H, cs, ucs, ol = geom
    assert sorted(got) == list(range(n_frames))
    for k in range(n_frames):
        ref = otiler.denoise_frame(synth.make_frame(W, H, seed=200 + k), cs, ucs, ol, _model, batch=3)
        assert np.isfinite(got[k]).all()
        assert np.abs(got[k] - ref).max() <= 1e-6      # seam rows re-associated (<= 1 ulp), the rest identical
        assert (got[k] == ref).mean() > 0.7


def _bcast_worker(rank, world, port, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nind_denoise_amd import dist as ndist
        torch.manual_seed(rank)          # every rank starts from different values
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.PReLU(), torch.nn.ConvTranspose2d(8, 3, 2, stride=2))
        n = ndist.broadcast_parameters(m, src=0)
        outq.put((rank, n, {k: v.numpy().copy() for k, v in m.state_dict().items()}))
    finally:
        dist.destroy_process_group()


def test_parameter_broadcast_world_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, sd0), (r1, n1, sd1) = sorted(got, key=lambda t: t[0])
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.PReLU(), torch.nn.ConvTranspose2d(8, 3, 2, stride=2)).state_dict()
    assert n0 == n1 == 4 * sum(v.numel() for v in ref.values())
    for k, v in ref.items():
        assert np.array_equal(sd0[k], v.numpy()) and np.array_equal(sd1[k], v.numpy())


def _bucket_worker(rank, world, port, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nind_denoise_amd import _lib
        from nind_denoise_amd import dist as ndist
        n = _lib.load().nd_utnet_param_count(8)
        flat = torch.arange(n, dtype=torch.float32) % 1000 * (rank + 1)
        av = ndist.BucketedGradientAverager(8, flat)
        av.reduce()
        outq.put((rank, flat.numpy(), av.buckets))
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_average_world_2():
    """The level buckets of the flat gradient buffer (nd_utnet_grad_buckets) tile it in the backward pass's completion order, and
    reducing them one by one gives the plain mean."""
    from nind_denoise_amd import _lib
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = _lib.load().nd_utnet_param_count(8)
    want = (np.arange(n, dtype=np.float32) % 1000) * 1.5
    for _, flat, buckets in got:
        assert np.array_equal(flat, want)
        assert len(buckets) == 9 and sum(c for _, c in buckets) == n
        ends = sorted((o, o + c) for o, c in buckets)
        assert ends[0][0] == 0 and all(ends[i][1] == ends[i + 1][0] for i in range(8)) and ends[-1][1] == n
        # completion order of the backward pass: decoder level 4 first (the end of the state dict), convs1 (offset 0) last
        assert buckets[-1][0] == 0 and buckets[0][0] == max(o for o, _ in buckets)
