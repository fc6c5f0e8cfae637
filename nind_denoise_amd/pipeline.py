"""Device-resident crop -> infer -> stitch loop (the hot loop of the reference's denoise_image.py:239-267).

The frame stays in HBM as float32 CHW; tiles are gathered, denoised and stitched on the GPU in batches,
in ascending tile index order (the reference's fp32 summation order), with no host synchronisation
inside the loop.  For ``UtNet`` the three stages are fused by ``nd_utnet_denoise_tiles`` (no NCHW tile
batch is ever materialised); any other callable model goes through ``nd_tile_gather`` -> model ->
``nd_stitch_add``.
"""
import torch

from . import _lib
from .networks.UtNet import UtNet


def tile_count(width, height, cs, ucs, ol):
    cols, rows, _ = _lib.tile_grid(width, height, cs, ucs, ol)
    return cols * rows


def gather_tiles(img, cs, ucs, ol, tile_begin, count):
    """img: [3,H,W] float32 cuda -> [count,3,cs,cs] (OneImageDS.__getitem__, denoise_image.py:129-174)."""
    assert img.is_cuda and img.dtype == torch.float32 and img.dim() == 3 and img.size(0) == 3
    img = img.contiguous()
    out = torch.empty((count, 3, cs, cs), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.load().nd_tile_gather(img.data_ptr(), img.size(2), img.size(1), cs, ucs, ol, tile_begin, count,
                                              out.data_ptr(), _lib.stream_ptr(img.device)), "nd_tile_gather")
    return out


def stitch_tiles(canvas, tiles, cs, ucs, ol, tile_begin):
    """canvas [3,H,W] += seamless useful crops of tiles [n,3,cs,cs] (denoise_image.py:204-213,249-267)."""
    assert canvas.is_cuda and canvas.is_contiguous() and tiles.is_cuda
    tiles = tiles.to(torch.float32).contiguous()
    with torch.cuda.device(canvas.device):
        _lib.check(_lib.load().nd_stitch_add(canvas.data_ptr(), canvas.size(2), canvas.size(1), cs, ucs, ol,
                                             tiles.data_ptr(), tile_begin, tiles.size(0),
                                             _lib.stream_ptr(canvas.device)), "nd_stitch_add")
    return canvas


def denoise_frame(model, img, cs, ucs, ol, batch=16, tile_range=None, canvas=None, progress=None):
    """Denoise one frame.  img: [3,H,W] float32 on the GPU.  Returns the stitched [3,H,W] canvas (on the GPU).

    tile_range=(begin, end) restricts the loop to a contiguous range of tile indices (multi-GPU sharding);
    the canvas then only holds those tiles' contributions.
    """
    if img.device.type != "cuda":
        raise RuntimeError("denoise_frame: the frame must be resident on the GPU (no CPU fallback)")
    img = img.to(torch.float32).contiguous()
    height, width = img.size(1), img.size(2)
    total = tile_count(width, height, cs, ucs, ol)
    begin, end = (0, total) if tile_range is None else tile_range
    if canvas is None:
        canvas = torch.zeros_like(img)
    lib = _lib.load()
    fused = isinstance(model, UtNet)
    with torch.cuda.device(img.device):
        if fused:
            batch = max(1, min(batch, end - begin)) if end > begin else 1
            blob = model.packed_weights(img.device)
            ws = model.workspace(cs, batch, img.device)
            stream = _lib.stream_ptr(img.device)
        for n, t0 in enumerate(range(begin, end, batch)):
            cnt = min(batch, end - t0)
            if progress is not None:
                progress(n, t0, cnt)
            if fused:
                _lib.check(lib.nd_utnet_denoise_tiles(model.funit, _lib.ACT[model.activation], _lib.DTYPE[model.compute_dtype],
                                                      model.flags, blob.data_ptr(), img.data_ptr(), canvas.data_ptr(), width, height,
                                                      cs, ucs, ol, t0, cnt, batch, ws.data_ptr(), ws.numel(), stream),
                           "nd_utnet_denoise_tiles")
            else:
                x = gather_tiles(img, cs, ucs, ol, t0, cnt)
                y = model(x)
                stitch_tiles(canvas, y, cs, ucs, ol, t0)
    return canvas
