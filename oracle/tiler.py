"""numpy restatement of the reference tiler and stitch loop (oracle: test infrastructure only).

Follows, line by line in meaning (not in text):
  * ``OneImageDS.__init__``      /root/reference/src/nind_denoise/denoise_image.py:88-107
  * ``OneImageDS.__getitem__``   denoise_image.py:129-174   (tiled branch), :110-128 (whole-image branch)
  * ``make_seamless_edges``      denoise_image.py:204-213
  * main-loop body (crop, +=)    denoise_image.py:249-267

Everything here is integer index arithmetic plus fp32 copies / halvings / adds,
so results are compared BIT-EXACTLY with the reference and with the HIP path.
"""
import math

import numpy as np


class TileGrid:
    """Grid constants of ``OneImageDS`` (denoise_image.py:100-104)."""

    def __init__(self, width, height, cs, ucs, ol):
        if ucs - ol <= 0:
            raise ValueError("ucs must exceed overlap")
        self.width, self.height = int(width), int(height)
        self.cs, self.ucs, self.ol = int(cs), int(ucs), int(ol)
        # the reference uses true division + ceil on python ints/floats
        self.iperhl = math.ceil((self.width - self.ucs) / (self.ucs - self.ol))
        self.ipervl = math.ceil((self.height - self.ucs) / (self.ucs - self.ol))
        self.pad = int((self.cs - self.ucs) / 2)
        self.cols = self.iperhl + 1
        self.rows = self.ipervl + 1
        self.size = self.cols * self.rows

    def geom(self, i):
        """(x0, y0, usefuldim[4], usefulstart[2]) of tile ``i`` (denoise_image.py:131-143,172-173)."""
        yi = int(math.ceil((i + 1) / (self.iperhl + 1) - 1))
        xi = i - yi * (self.iperhl + 1)
        x0 = self.ucs * xi - self.ol * xi - self.pad
        y0 = self.ucs * yi - self.ol * yi - self.pad
        x1, y1 = x0 + self.cs, y0 + self.cs
        x1pad = max(0, x1 - self.width)
        y1pad = max(0, y1 - self.height)
        ud = (self.pad, self.pad, self.cs - max(self.pad, x1pad), self.cs - max(self.pad, y1pad))
        us = (x0 + self.pad, y0 + self.pad)
        return x0, y0, ud, us


def gather_tile(inimg, grid, i):
    """Tile ``i`` as float32 [3, cs, cs] with the reference's symmetric mirror padding.

    Restates denoise_image.py:138-170 with explicit slices; the mirrored strips
    repeat the edge pixel (np.flip of the adjacent band), and the four corners
    come from the image's own corner blocks flipped on both axes.
    """
    cs = grid.cs
    W, H = grid.width, grid.height
    x0, y0, _, _ = grid.geom(i)
    x1, y1 = x0 + cs, y0 + cs
    x0pad, x1pad = -min(0, x0), max(0, x1 - W)
    y0pad, y1pad = -min(0, y0), max(0, y1 - H)
    ret = np.empty((3, cs, cs), dtype=np.float32)
    ys, ye = y0 + y0pad, y1 - y1pad
    xs, xe = x0 + x0pad, x1 - x1pad
    ret[:, y0pad:cs - y1pad, x0pad:cs - x1pad] = inimg[:, ys:ye, xs:xe]
    if x0pad > 0:
        ret[:, y0pad:cs - y1pad, :x0pad] = inimg[:, ys:ye, xs:xs + x0pad][:, :, ::-1]
        if y0pad > 0:
            ret[:, :y0pad, :x0pad] = inimg[:, :y0pad, :x0pad][:, ::-1, ::-1]
        if y1pad > 0:
            ret[:, cs - y1pad:, :x0pad] = inimg[:, H - y1pad:, :x0pad][:, ::-1, ::-1]
    if x1pad > 0:
        ret[:, y0pad:cs - y1pad, cs - x1pad:] = inimg[:, ys:ye, xe - x1pad:xe][:, :, ::-1]
        if y0pad > 0:
            ret[:, :y0pad, cs - x1pad:] = inimg[:, :y0pad, W - x1pad:][:, ::-1, ::-1]
        if y1pad > 0:
            ret[:, cs - y1pad:, cs - x1pad:] = inimg[:, H - y1pad:, W - x1pad:][:, ::-1, ::-1]
    if y0pad > 0:
        ret[:, :y0pad, x0pad:cs - x1pad] = inimg[:, ys:ys + y0pad, xs:xe][:, ::-1, :]
    if y1pad > 0:
        ret[:, cs - y1pad:, x0pad:cs - x1pad] = inimg[:, ye - y1pad:ye, xs:xe][:, ::-1, :]
    return ret


def whole_image_item(inimg, pad):
    """The single item of ``OneImageDS(whole_image=True, pad=pad)`` (denoise_image.py:110-128): the frame in the centre,
    the four SIDE bands mirrored (edge pixel repeated), the four pad x pad corners left at ZERO -- the reference mirrors
    the sides only.  Returns (item [3, H+2p, W+2p], usefuldim, usefulstart).  (The reference allocates (W+2p, H+2p), which
    only works for square frames; H and W are used in their proper places here.)"""
    p = int(pad or 0)
    H, W = inimg.shape[1], inimg.shape[2]
    ret = np.zeros((3, H + 2 * p, W + 2 * p), dtype=np.float32)
    ret[:, p:H + p, p:W + p] = inimg
    if p:
        ret[:, p:-p, :p] = inimg[:, :, :p][:, :, ::-1]
        ret[:, p:-p, W + p:] = inimg[:, :, W - p:][:, :, ::-1]
        ret[:, :p, p:-p] = inimg[:, :p, :][:, ::-1, :]
        ret[:, H + p:, p:-p] = inimg[:, H - p:, :][:, ::-1, :]
    return ret, (p, p, W + p, H + p), (p, p)


def make_seamless_edges(tcrop, x0, y0, grid):
    """Halve the overlap strips of an already-cropped tile (denoise_image.py:204-213)."""
    ol, ucs = grid.ol, grid.ucs
    if x0 != 0:
        tcrop[:, :, 0:ol] = tcrop[:, :, 0:ol] / np.float32(2)
    if y0 != 0:
        tcrop[:, 0:ol, :] = tcrop[:, 0:ol, :] / np.float32(2)
    if x0 + ucs < grid.width and ol:
        tcrop[:, :, -ol:] = tcrop[:, :, -ol:] / np.float32(2)
    if y0 + ucs < grid.height and ol:
        tcrop[:, -ol:, :] = tcrop[:, -ol:, :] / np.float32(2)
    return tcrop


def stitch_add(canvas, tile_out, grid, i):
    """canvas += seamless(useful crop of tile i's network output) (denoise_image.py:249-267)."""
    _, _, ud, us = grid.geom(i)
    t = np.array(tile_out[:, ud[1]:ud[3], ud[0]:ud[2]], dtype=np.float32, copy=True)
    ax, ay = us
    t = make_seamless_edges(t, ax, ay, grid)
    h, w = t.shape[1], t.shape[2]
    canvas[:, ay:ay + h, ax:ax + w] = canvas[:, ay:ay + h, ax:ax + w] + t
    return canvas


def denoise_frame(inimg, cs, ucs, ol, model_fn, batch=1):
    """Whole crop -> infer -> stitch loop on the CPU; ``model_fn`` maps [B,3,cs,cs] -> same."""
    grid = TileGrid(inimg.shape[2], inimg.shape[1], cs, ucs, ol)
    canvas = np.zeros((3, grid.height, grid.width), dtype=np.float32)
    for b0 in range(0, grid.size, batch):
        idx = list(range(b0, min(grid.size, b0 + batch)))
        x = np.stack([gather_tile(inimg, grid, i) for i in idx])
        y = model_fn(x)
        for k, i in enumerate(idx):
            stitch_add(canvas, y[k], grid, i)
    return canvas
