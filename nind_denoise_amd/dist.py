"""Multi-GPU sharding of the tile loop: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The reference has no multi-device code; tiles are independent, so the outer loop of
denoise_image.py:240-267 is partitioned into contiguous tile-index shards, one per rank.  Per frame:

  1. scatter : rank 0 sends every rank only the image rows its tiles read (shard rows + the cs-ucs halo),
               point-to-point, all peers at once (xGMI is a full mesh: one link per peer, no ring);
  2. compute : every rank runs the device-resident crop -> UtNet -> stitch loop on its shard into its own canvas;
  3. gather  : every rank sends the canvas rows its tiles touched; rank 0 adds the bands in rank order.  A pixel on a
               shard seam receives contributions from two ranks, so its fp32 sum is re-associated with respect to the
               single-GPU tile order (<= 1 ulp); everything else is bit-identical to one GPU.

The exchange logic is backend-agnostic (tests run it on gloo/CPU with world_size 2).
"""
import torch
import torch.distributed as dist

from . import _lib


class Geo:
    """Row geometry of a tile shard (pure integer, from nd_tile_grid)."""

    def __init__(self, width, height, cs, ucs, ol):
        self.W, self.H, self.cs, self.ucs, self.ol = width, height, cs, ucs, ol
        self.cols, self.rows, self.pad = _lib.tile_grid(width, height, cs, ucs, ol)
        self.stride = ucs - ol
        self.total = self.cols * self.rows

    def shard(self, rank, world):
        return (self.total * rank) // world, (self.total * (rank + 1)) // world

    def rows_out(self, lo, hi):
        """canvas rows [y0, y1) written by tiles [lo, hi)"""
        if hi <= lo:
            return 0, 0
        yi0, yi1 = lo // self.cols, (hi - 1) // self.cols
        return yi0 * self.stride, min(self.H, yi1 * self.stride + self.cs - 2 * self.pad)

    def rows_in(self, lo, hi):
        """image rows [y0, y1) read by tiles [lo, hi) (mirror padding reflects inside this range)"""
        if hi <= lo:
            return 0, 0
        yi0, yi1 = lo // self.cols, (hi - 1) // self.cols
        y0 = yi0 * self.stride - self.pad
        y1 = yi1 * self.stride - self.pad + self.cs
        a, b = max(0, y0), min(self.H, y1)
        if y0 < 0:
            b = max(b, min(self.H, -y0))           # rows mirrored above the top edge
        if y1 > self.H:
            a = min(a, max(0, 2 * self.H - y1))    # rows mirrored below the bottom edge
        return a, b


def _p2p(ops, group):
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def _host_staged(group):
    """gloo moves CPU tensors only: a GPU run over gloo (the two-ranks-on-one-GPU rehearsal of tests / bench.py; RCCL refuses
    two ranks on one device) stages every message through host memory.  RCCL ("nccl") sends straight from HBM over xGMI."""
    return dist.get_backend(group) == "gloo"


def _wire(t, group):
    """the tensor that goes on the wire for `t` (contiguous; on the host when the backend needs it)"""
    t = t.contiguous()
    return t.cpu() if (t.is_cuda and _host_staged(group)) else t


def _landing(shape, like, group):
    dev = "cpu" if (like.is_cuda and _host_staged(group)) else like.device
    return torch.empty(shape, dtype=like.dtype, device=dev)


def scatter_frame(frame, geo, group=None, src=0):
    """frame: full [3,H,W] buffer on every rank; valid on `src`.  After the call each rank holds the rows it needs."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return frame
    if rank == src:
        ops, keep = [], []
        for r in range(world):
            if r == src:
                continue
            a, b = geo.rows_in(*geo.shard(r, world))
            if b > a:
                buf = _wire(frame[:, a:b, :], group)
                keep.append(buf)
                ops.append(dist.P2POp(dist.isend, buf, r, group))
        _p2p(ops, group)
    else:
        a, b = geo.rows_in(*geo.shard(rank, world))
        if b > a:
            buf = _landing((3, b - a, geo.W), frame, group)
            _p2p([dist.P2POp(dist.irecv, buf, src, group)], group)
            frame[:, a:b, :] = buf.to(frame.device)
    return frame


def gather_canvas(canvas, geo, group=None, dst=0):
    """Sum every rank's canvas band onto rank `dst` (bands added in rank order)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return canvas
    if rank == dst:
        ops, bufs = [], []
        for r in range(world):
            if r == dst:
                continue
            a, b = geo.rows_out(*geo.shard(r, world))
            if b > a:
                buf = _landing((3, b - a, geo.W), canvas, group)
                bufs.append((a, b, buf))
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
        _p2p(ops, group)
        for a, b, buf in bufs:
            canvas[:, a:b, :] += buf.to(canvas.device)
    else:
        a, b = geo.rows_out(*geo.shard(rank, world))
        if b > a:
            _p2p([dist.P2POp(dist.isend, _wire(canvas[:, a:b, :], group), dst, group)], group)
    return canvas


def denoise_frame_sharded(compute, frame, canvas, geo, group=None, root=0):
    """One frame across the ranks of `group`.

    compute(frame, canvas, lo, hi) must add the contributions of tiles [lo, hi) to `canvas` (pipeline.denoise_frame
    with tile_range / canvas does).  `frame` is valid on `root` on entry; the stitched result is in `canvas` on `root`.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    scatter_frame(frame, geo, group, root)
    lo, hi = geo.shard(rank, world)
    a, b = geo.rows_out(lo, hi)
    if b > a:
        canvas[:, a:b, :].zero_()
    if rank == root:
        canvas.zero_()
    compute(frame, canvas, lo, hi)
    gather_canvas(canvas, geo, group, root)
    return canvas


def frame_shard(n_frames, rank, world):
    """Frame-level sharding of a multi-frame batch (BASELINE configs[2]: 100 frames over 8 GPUs): frame f goes to rank
    f % world.  Whole frames are independent, so there is no per-frame exchange at all -- only the one-time weight broadcast."""
    return list(range(rank, n_frames, world))


def denoise_frames_sharded(denoise, frames, group=None, collect=True):
    """A batch of frames across the ranks of `group`, frame-level sharding.

    frames: list of [3,H,W] tensors (every rank holds, or can produce, the frames it owns: only frames[f] with
    f % world == rank are touched).  denoise(frame) -> stitched canvas (pipeline.denoise_frame / serve.FrameEngine).
    Returns {frame index: canvas} for the frames this rank owns; with collect=True rank 0 receives every other rank's
    canvases too (point-to-point, one message per frame) and returns the full dict -- the serving case, where results
    leave through one process."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = frame_shard(len(frames), rank, world)
    out = {f: denoise(frames[f]) for f in mine}
    if not collect or world == 1:
        return out
    if rank == 0:
        for f in range(len(frames)):
            src = f % world
            if src != 0:
                like = out[mine[0]] if mine else frames[f]
                buf = _landing(tuple(like.shape), like, group)
                _p2p([dist.P2POp(dist.irecv, buf, src, group)], group)
                out[f] = buf.to(like.device)
    else:
        for f in mine:
            _p2p([dist.P2POp(dist.isend, _wire(out[f], group), 0, group)], group)
    return out


def average_gradients(flat, group=None):
    """Data-parallel training (BASELINE config 5): ONE all-reduce of the flat state-dict-order gradient buffer
    (124 MB fp32 for UtNet(64)), then the mean.  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world > 1:
        if flat.is_cuda and _host_staged(group):
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat
