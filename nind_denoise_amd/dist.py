"""Multi-GPU sharding of the tile loop: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The reference has no multi-device code; tiles are independent, so the outer loop of
denoise_image.py:240-267 is partitioned into contiguous tile-index shards, one per rank.  Per frame:

  1. scatter : rank 0 sends every rank only the image rows its tiles read (shard rows + the cs-ucs halo),
               point-to-point, all peers at once (xGMI is a full mesh: one link per peer, no ring);
  2. compute : every rank runs the device-resident crop -> UtNet -> stitch loop on its shard into its own canvas;
  3. gather  : every rank sends the canvas rows its tiles touched; rank 0 adds the bands in rank order.  A pixel on a
               shard seam receives contributions from two ranks, so its fp32 sum is re-associated with respect to the
               single-GPU tile order (<= 1 ulp); everything else is bit-identical to one GPU.

A STREAM of frames runs the same partition software-pipelined (`ShardedFrameStream`): the scatter of frame n+1 and the gather +
band adds of frame n-1 travel on a side stream under the compute of frame n, so the per-frame cost at N ranks is the compute
of the largest shard, not compute + two exchanges.

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


class ShardedFrameStream:
    """Tile-shard of a STREAM of equally sized frames over the ranks of `group`, software-pipelined across frames.

    The partition is `denoise_frame_sharded`'s (contiguous tile-index shards of the loop of denoise_image.py:240-267; frames
    enter and leave through `root`).  Step n of the pipeline overlaps, on every rank,

        K(n)   crop -> UtNet -> stitch of this rank's shard of frame n                      (caller's stream)
        C(n)   ONE grouped point-to-point batch: root -> peers the input row bands of frame n+1,
               peers -> root the canvas row bands of frame n-1; then root adds those bands       (side stream)

    over rings of two frame / canvas / landing buffers per rank (whole [3,H,W] buffers: a band is received straight into its
    rows, per channel, so nothing is repacked; 288 MB each at 24 MP -- HBM is not the constraint).  K(n) is enqueued before C(n)
    is issued, C(n) waits for K(n-1) only, K(n+1) waits for C(n): in steady state a frame costs max(compute of the largest
    shard, exchange) instead of their sum, and the pipeline holds a frame for two extra steps.  Both sides issue the scatter and
    the gather of a step in one batch_isend_irecv group, so no send can wait behind an unmatched receive.

    compute(frame, canvas, lo, hi) adds the contributions of tiles [lo, hi) to `canvas` (pipeline.denoise_frame with tile_range /
    canvas does).  On a gloo group with CUDA tensors (the one-GPU rehearsal) messages are staged through host memory; the
    schedule is the same.  With CPU tensors (tests with the oracle as compute) everything is synchronous.
    """

    def __init__(self, compute, geo, device, group=None, root=0):
        self.compute, self.geo, self.group, self.root = compute, geo, group, root
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.staged = self.cuda and _host_staged(group)
        self.lo, self.hi = geo.shard(self.rank, self.world)
        self.bands_in = [geo.rows_in(*geo.shard(r, self.world)) for r in range(self.world)]
        self.bands_out = [geo.rows_out(*geo.shard(r, self.world)) for r in range(self.world)]
        shape = (3, geo.H, geo.W)
        self.is_root = self.rank == root
        self.canvas = [torch.zeros(shape, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.frames = None if self.is_root else [torch.zeros(shape, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.land = None
        if self.is_root:
            self.land = [{r: torch.empty((3, b - a, geo.W), dtype=torch.float32, device=self.device)
                          for r, (a, b) in enumerate(self.bands_out) if r != root and b > a} for _ in range(2)]
        if self.cuda:
            with torch.cuda.device(self.device):
                self.s_comm = torch.cuda.Stream()
                self.ev_k = [torch.cuda.Event() for _ in range(2)]
                self.ev_c = torch.cuda.Event()
                self.ev_ready = torch.cuda.Event()

    # -- one exchange step: scatter of frame n+1 (if any), gather + adds of frame n-1 (if any)
    def _exchange(self, n, n_frames, src_next):
        geo, root = self.geo, self.root
        do_scatter, do_gather = n + 1 < n_frames, n - 1 >= 0
        ops, after = [], []
        if self.is_root:
            if do_scatter:
                for r, (a, b) in enumerate(self.bands_in):
                    if r != root and b > a:
                        for c in range(3):
                            ops.append((dist.isend, src_next[c, a:b, :], r))
            if do_gather:
                slot = (n - 1) % 2
                for r, buf in self.land[slot].items():
                    a, b = self.bands_out[r]
                    for c in range(3):
                        ops.append((dist.irecv, buf[c], r))
                    after.append((self.canvas[slot], a, b, buf))
        else:
            a, b = self.bands_in[self.rank]
            if do_scatter and b > a:
                for c in range(3):
                    ops.append((dist.irecv, self.frames[(n + 1) % 2][c, a:b, :], root))
            a, b = self.bands_out[self.rank]
            if do_gather and b > a:
                for c in range(3):
                    ops.append((dist.isend, self.canvas[(n - 1) % 2][c, a:b, :], root))
        if self.staged:
            # gloo moves host memory: device -> host for what is sent, host -> device for what arrived (on the side stream)
            host = [(fn, t.cpu() if fn is dist.isend else torch.empty(t.shape, dtype=t.dtype), peer, t) for fn, t, peer in ops]
            _p2p([dist.P2POp(fn, h, peer, self.group) for fn, h, peer, _ in host], self.group)
            for fn, h, _, t in host:
                if fn is dist.irecv:
                    t.copy_(h, non_blocking=False)
        else:
            _p2p([dist.P2POp(fn, t, peer, self.group) for fn, t, peer in ops], self.group)
        for cv, a, b, buf in after:      # bands added in rank order (deterministic)
            cv[:, a:b, :] += buf

    def run(self, frames, n_frames):
        """Generator.  `frames`: iterable of [3,H,W] tensors on `root` (ignored elsewhere); every rank passes the same `n_frames`.
        Yields (index, canvas) on root as frames complete, in order -- the canvas is a ring slot: use it (stream-ordered work on
        the current stream is enough, e.g. a copy) BEFORE advancing the generator again, which enqueues the compute that reuses
        the slot; yields (index, None) on the other ranks."""
        geo = self.geo
        it = iter(frames) if self.is_root else None

        def take():
            f = next(it)
            if tuple(f.shape) != (3, geo.H, geo.W):
                raise ValueError(f"ShardedFrameStream: frame shape {tuple(f.shape)} != {(3, geo.H, geo.W)}")
            return f

        cur_src = take() if (self.is_root and n_frames > 0) else None
        for n in range(-1, n_frames + 1):
            nxt_src = take() if (self.is_root and n + 1 < n_frames and n >= 0) else (cur_src if n == -1 else None)
            if self.cuda:
                cur = torch.cuda.current_stream(self.device)
                self.ev_ready.record(cur)          # the caller's frames are ready at this point of its stream
            if 0 <= n < n_frames:
                slot = n % 2
                cv = self.canvas[slot]
                a, b = self.bands_out[self.rank]
                if self.is_root:
                    cv.zero_()
                elif b > a:
                    cv[:, a:b, :].zero_()
                self.compute(cur_src if self.is_root else self.frames[slot], cv, self.lo, self.hi)
                if self.cuda:
                    self.ev_k[slot].record(cur)
            if self.cuda:
                with torch.cuda.stream(self.s_comm):
                    self.s_comm.wait_event(self.ev_ready)
                    if n - 1 >= 0:
                        self.s_comm.wait_event(self.ev_k[(n - 1) % 2])
                    self._exchange(n, n_frames, nxt_src)
                    self.ev_c.record(self.s_comm)
                cur.wait_event(self.ev_c)          # K(n+1) needs the bands of C(n); the canvas of frame n-1 is complete behind it
            else:
                self._exchange(n, n_frames, nxt_src)
            if n >= 1:
                yield n - 1, (self.canvas[(n - 1) % 2] if self.is_root else None)
            if n >= 0:
                cur_src = nxt_src


def frame_shard(n_frames, rank, world):
    """Frame-level sharding of a multi-frame batch (BASELINE configs[2]: 100 frames over 8 GPUs): frame f goes to rank
    f % world.  Whole frames are independent, so there is no per-frame exchange at all -- only the one-time weight broadcast."""
    return list(range(rank, n_frames, world))


def denoise_frames_sharded(denoise, frames, group=None, collect=True):
    """A batch of frames across the ranks of `group`, frame-level sharding.

    frames: list of equally shaped [3,H,W] tensors (checked).  Every rank holds, or can produce, the frames it owns: only the
    CONTENT of frames[f] with f % world == rank is read; of the others only the shape (rank 0 sizes its landing buffers from it).
    denoise(frame) -> stitched canvas (pipeline.denoise_frame / serve.FrameEngine).
    Returns {frame index: canvas} for the frames this rank owns; with collect=True rank 0 receives every other rank's
    canvases too (point-to-point, one message per frame) and returns the full dict -- the serving case, where results
    leave through one process."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shapes = {tuple(f.shape) for f in frames}
    if len(shapes) > 1:
        # the landing buffers on rank 0 are sized from frames[f].shape, which every rank must therefore know and agree on
        raise ValueError(f"denoise_frames_sharded: all frames of a batch must have one shape, got {sorted(shapes)}")
    mine = frame_shard(len(frames), rank, world)
    out = {f: denoise(frames[f]) for f in mine}
    if not collect or world == 1:
        return out
    if rank == 0:
        # every receive posted in ONE batch (the peers send in frame order); only the SHAPE of a foreign frame is looked at
        like = out[mine[0]] if mine else None
        ops, bufs = [], {}
        for f in range(len(frames)):
            src = f % world
            if src != 0:
                ref = like if like is not None else frames[f]
                bufs[f] = _landing(tuple(frames[f].shape), ref, group)
                ops.append(dist.P2POp(dist.irecv, bufs[f], src, group))
        _p2p(ops, group)
        for f, buf in bufs.items():
            out[f] = buf.to(like.device) if like is not None else buf
    else:
        wires = [_wire(out[f], group) for f in mine]
        _p2p([dist.P2POp(dist.isend, w, 0, group) for w in wires], group)
    return out


def broadcast_parameters(module, src=0, group=None):
    """Rank `src` owns the model: ONE broadcast of its raw parameters as a flat fp32 buffer in state-dict order (124 MB for
    UtNet(64)); every rank then packs them into MFMA fragment order on its own device (the packed fp32 blob with its Winograd
    forms is 926 MB: 7x the wire traffic for bytes each rank rebuilds in milliseconds).  Returns the bytes broadcast."""
    params = list(module.parameters())
    if not params:
        return 0
    flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in params])
    if dist.get_world_size(group) > 1:
        if flat.is_cuda and _host_staged(group):
            h = flat.cpu()
            dist.broadcast(h, src, group)
            flat = h.to(flat.device)
        else:
            dist.broadcast(flat, src, group)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))     # (bumps the parameter's version: UtNet re-packs on next use)
            off += n
    return flat.numel() * 4


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


class BucketedGradientAverager:
    """Data-parallel gradient mean that overlaps the backward pass (BASELINE configs[4]).

    The flat state-dict-order gradient buffer is cut into the library's level buckets (nd_utnet_grad_buckets: up4 + tconvs4, ...,
    convs1 -- the order in which the backward pass completes them).  The training step records one HIP event per bucket on the
    compute stream as soon as the bucket is final (nd_utnet_train_step_ev / nd_utnet_train_backward); `reduce()` all-reduces
    bucket k on a side stream behind event k, i.e. under the backward of the shallower levels, and the compute stream only waits
    for the last (smallest) buckets.  On a gloo group (tests, the one-GPU rehearsal) the buckets are reduced one by one through
    host memory after the step; with CPU tensors there are no events at all.  No-op without an initialised process group."""

    def __init__(self, funit, flat, group=None):
        import ctypes
        lib = _lib.load()
        n = lib.nd_utnet_grad_buckets(funit, None, None, 0)
        if n <= 0:
            _lib.check(n, "nd_utnet_grad_buckets")
        off, cnt = (ctypes.c_size_t * n)(), (ctypes.c_size_t * n)()
        _lib.check(min(0, lib.nd_utnet_grad_buckets(funit, off, cnt, n)), "nd_utnet_grad_buckets")
        self.buckets = [(int(off[k]), int(cnt[k])) for k in range(n)]
        assert sum(c for _, c in self.buckets) == flat.numel(), "gradient buckets do not tile the flat buffer"
        self.flat, self.group = flat, group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.events, self.event_ptrs, self.stream = None, None, None
        if flat.is_cuda:
            with torch.cuda.device(flat.device):
                self.stream = torch.cuda.Stream()
                self.events = [torch.cuda.Event() for _ in range(n)]
                for e in self.events:
                    e.record()                    # (a torch event owns its hipEvent_t from the first record on)
                self.event_ptrs = (ctypes.c_void_p * n)(*[e.cuda_event for e in self.events])

    def reduce(self):
        """Call right after the step was enqueued: averages self.flat over the group, bucket by bucket."""
        if not self.active:
            return self.flat
        world = dist.get_world_size(self.group)
        if self.flat.is_cuda and not _host_staged(self.group):
            cur = torch.cuda.current_stream(self.flat.device)
            with torch.cuda.stream(self.stream):
                for k, (off, cnt) in enumerate(self.buckets):
                    self.stream.wait_event(self.events[k])
                    part = self.flat[off:off + cnt]
                    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
                    part.div_(world)
            cur.wait_stream(self.stream)
        else:
            for off, cnt in self.buckets:
                part = self.flat[off:off + cnt]
                if part.is_cuda:
                    h = part.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                    part.copy_(h)
                else:
                    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
                part.div_(world)
        return self.flat
