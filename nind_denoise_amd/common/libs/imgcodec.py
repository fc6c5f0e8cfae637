"""Minimal TIFF / PNG codecs for the denoise CLI (cv2, imageio and tifffile are absent in this image; PIL only
handles 8-bit RGB).  Covers what the reference's I/O path produces and consumes
(/root/reference/src/nind_denoise/common/libs/np_imgops.py:12-29, pt_helpers.py:22-40):

  read : baseline TIFF, strips or tiles, chunky or planar, u8 / u16 / f32 samples (f16/f64 converted), 1/3/4 samples,
         compression none / deflate (8, 32946) / LZW (5) / PackBits (32773), predictor 1 / 2 / 3, both byte orders;
         PNG 8/16-bit gray / RGB / RGBA (non-interlaced);  anything else 8-bit through PIL.
  write: uncompressed little-endian RGB TIFF with u8 / u16 / f32 samples; 8/16-bit RGB PNG.
All arrays are HWC here; callers transpose to the reference's CHW.
"""
import struct
import zlib

import numpy as np

_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d", 16: "Q"}


# ------------------------------------------------------------------------------------------------ TIFF read

def _lzw_decode(data):
    """TIFF LZW (MSB-first codes, early change)."""
    out = bytearray()
    table = [bytes([i]) for i in range(256)] + [b"", b""]
    bits, nbits, width, prev = 0, 0, 9, None
    for byte in data:
        bits = (bits << 8) | byte
        nbits += 8
        while nbits >= width:
            code = (bits >> (nbits - width)) & ((1 << width) - 1)
            nbits -= width
            if code == 256:
                table = table[:258]
                width, prev = 9, None
                continue
            if code == 257:
                return bytes(out)
            if prev is None:
                entry = table[code]
            elif code < len(table):
                entry = table[code]
                table.append(prev + entry[:1])
            else:
                entry = prev + prev[:1]
                table.append(entry)
            out += entry
            prev = entry
            if len(table) >= (1 << width) - 1 and width < 12:
                width += 1
    return bytes(out)


def _packbits_decode(data):
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        h = data[i]
        i += 1
        if h < 128:
            out += data[i:i + h + 1]
            i += h + 1
        elif h > 128:
            out += data[i:i + 1] * (257 - h)
            i += 1
    return bytes(out)


def _decompress(raw, compression):
    if compression == 1:
        return raw
    if compression in (8, 32946):
        return zlib.decompress(raw)
    if compression == 5:
        return _lzw_decode(raw)
    if compression == 32773:
        return _packbits_decode(raw)
    raise NotImplementedError(f"TIFF compression {compression}")


def read_tiff(path):
    with open(path, "rb") as f:
        buf = f.read()
    bo = {b"II": "<", b"MM": ">"}.get(buf[:2])
    if bo is None:
        raise ValueError(f"{path}: not a TIFF file")
    magic = struct.unpack(bo + "H", buf[2:4])[0]
    if magic != 42:
        raise NotImplementedError(f"{path}: BigTIFF / unknown magic {magic}")
    ifd = struct.unpack(bo + "I", buf[4:8])[0]
    n = struct.unpack(bo + "H", buf[ifd:ifd + 2])[0]
    tags = {}
    for k in range(n):
        e = ifd + 2 + 12 * k
        tag, typ, cnt = struct.unpack(bo + "HHI", buf[e:e + 8])
        fmt = _TYPES.get(typ)
        if fmt is None:
            continue
        size = struct.calcsize("=" + fmt) * cnt
        off = e + 8 if size <= 4 else struct.unpack(bo + "I", buf[e + 8:e + 12])[0]
        if typ == 2:
            tags[tag] = buf[off:off + cnt]
        else:
            tags[tag] = struct.unpack(bo + fmt * cnt, buf[off:off + size])
    W, H = tags[256][0], tags[257][0]
    spp = tags.get(277, (1,))[0]
    bps = tags.get(258, (1,) * spp)[0]
    comp = tags.get(259, (1,))[0]
    planar = tags.get(284, (1,))[0]
    pred = tags.get(317, (1,))[0]
    sfmt = tags.get(339, (1,))[0]
    photometric = tags.get(262, (2,))[0]
    if photometric not in (0, 1, 2):
        raise NotImplementedError(f"{path}: photometric {photometric}")
    kind = {1: "u", 2: "i", 3: "f"}.get(sfmt, "u")
    dt = np.dtype(f"{bo}{kind}{bps // 8}")
    if bps % 8:
        raise NotImplementedError(f"{path}: {bps} bits per sample")
    chans = spp if planar == 1 else 1
    nplanes = 1 if planar == 1 else spp

    def decode_block(raw, rows, cols):
        data = _decompress(raw, comp)
        if pred == 3:
            # floating-point predictor (TIFF TechNote 3): per row, sample bytes are split into byte planes (most
            # significant first) and the whole row is byte-differenced with a stride of `chans`
            rowbytes = cols * chans * dt.itemsize
            b = np.frombuffer(data, dtype=np.uint8, count=rows * rowbytes).reshape(rows, rowbytes // chans, chans)
            b = np.cumsum(b, axis=1, dtype=np.uint8).reshape(rows, dt.itemsize, cols * chans).transpose(0, 2, 1)
            be = np.ascontiguousarray(b).view(np.dtype(f">{kind}{dt.itemsize}"))
            return be.reshape(rows, cols, chans)
        a = np.frombuffer(data, dtype=dt, count=rows * cols * chans).reshape(rows, cols, chans)
        if pred == 2:
            a = np.cumsum(a.astype(dt.newbyteorder("=")), axis=1, dtype=dt.newbyteorder("="))
        return a

    img = np.empty((H, W, spp), dtype=dt.newbyteorder("="))
    if 322 in tags:  # tiles
        tw, th = tags[322][0], tags[323][0]
        offs, cnts = tags[324], tags[325]
        tx, ty = (W + tw - 1) // tw, (H + th - 1) // th
        for p in range(nplanes):
            for j in range(ty):
                for i in range(tx):
                    k = (p * ty + j) * tx + i
                    blk = decode_block(buf[offs[k]:offs[k] + cnts[k]], th, tw)
                    h, w = min(th, H - j * th), min(tw, W - i * tw)
                    if planar == 1:
                        img[j * th:j * th + h, i * tw:i * tw + w, :] = blk[:h, :w, :]
                    else:
                        img[j * th:j * th + h, i * tw:i * tw + w, p] = blk[:h, :w, 0]
    else:
        rps = min(tags.get(278, (H,))[0], H)
        offs, cnts = tags[273], tags[279]
        spi = (H + rps - 1) // rps
        for p in range(nplanes):
            for s in range(spi):
                k = p * spi + s
                rows = min(rps, H - s * rps)
                blk = decode_block(buf[offs[k]:offs[k] + cnts[k]], rows, W)
                if planar == 1:
                    img[s * rps:s * rps + rows] = blk
                else:
                    img[s * rps:s * rps + rows, :, p] = blk[:, :, 0]
    if photometric == 0:
        img = (np.iinfo(img.dtype).max - img) if img.dtype.kind == "u" else -img
    if img.dtype.kind == "f" and img.dtype.itemsize != 4:
        img = img.astype(np.float32)
    return img


# ------------------------------------------------------------------------------------------------ TIFF write

def write_tiff(path, img):
    """img: HWC (3 channels) uint8 / uint16 / float32 -> uncompressed little-endian baseline RGB TIFF."""
    img = np.ascontiguousarray(img)
    if img.ndim != 3 or img.shape[2] != 3 or img.dtype not in (np.uint8, np.uint16, np.float32):
        raise NotImplementedError(f"write_tiff: shape {img.shape} dtype {img.dtype}")
    H, W, _ = img.shape
    bps = img.dtype.itemsize * 8
    sfmt = 3 if img.dtype == np.float32 else 1
    data = memoryview(np.ascontiguousarray(img.astype(img.dtype.newbyteorder("<"), copy=False))).cast("B")   # (no copy of the samples)
    rps = max(1, min(H, (1 << 20) // max(1, W * 3 * img.dtype.itemsize)))
    nstrips = (H + rps - 1) // rps
    row_bytes = W * 3 * img.dtype.itemsize
    entries = []
    extra = bytearray()
    header_len = 8
    n_tags = 12
    ifd_len = 2 + 12 * n_tags + 4
    extra_base = header_len + ifd_len

    def put(tag, typ, values):
        fmt = _TYPES[typ]
        cnt = len(values)
        payload = struct.pack("<" + fmt * cnt, *values)
        if len(payload) <= 4:
            entries.append(struct.pack("<HHI", tag, typ, cnt) + payload.ljust(4, b"\0"))
        else:
            off = extra_base + len(extra)
            extra.extend(payload)
            if len(extra) % 2:
                extra.append(0)
            entries.append(struct.pack("<HHII", tag, typ, cnt, off))

    # strip offsets depend on the size of `extra`: two passes
    strip_counts = [min(rps, H - s * rps) * row_bytes for s in range(nstrips)]
    for data_base in (0, None):
        entries.clear()
        del extra[:]
        base = data_base if data_base is not None else extra_base + extra_len
        offsets = [base + s * rps * row_bytes for s in range(nstrips)]
        put(256, 4, [W])
        put(257, 4, [H])
        put(258, 3, [bps] * 3)
        put(259, 3, [1])
        put(262, 3, [2])
        put(273, 4, offsets)
        put(277, 3, [3])
        put(278, 4, [rps])
        put(279, 4, strip_counts)
        put(284, 3, [1])
        put(339, 3, [sfmt] * 3)
        put(305, 2, [bytes([c]) for c in b"nind_denoise_amd\0"])
        extra_len = len(extra)
    assert len(entries) == n_tags
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, 8))
        f.write(struct.pack("<H", n_tags) + b"".join(entries) + struct.pack("<I", 0))
        f.write(bytes(extra))
        f.write(data)


# ------------------------------------------------------------------------------------------------ PNG

def _png_unfilter(raw, rows, stride, bpp):
    """Undo PNG row filters.  Filters 0-2 are vectorised; Average / Paeth rows run through libnind_hip's host helper
    when it is present, else a per-pixel loop."""
    a = np.frombuffer(raw, dtype=np.uint8, count=rows * (stride + 1)).reshape(rows, stride + 1)
    ft = a[:, 0].copy()
    out = a[:, 1:].copy()
    prev = np.zeros(stride, dtype=np.uint8)
    for r in range(rows):
        row = out[r]
        t = ft[r]
        if t == 1:
            v = row.reshape(-1, bpp)
            np.cumsum(v, axis=0, dtype=np.uint8, out=v)
        elif t == 2:
            row += prev
        elif t in (3, 4):
            p = prev.astype(np.int32)
            x = row.astype(np.int32)
            for i in range(stride):
                left = x[i - bpp] if i >= bpp else 0
                up = p[i]
                if t == 3:
                    x[i] = (x[i] + ((left + up) >> 1)) & 255
                else:
                    ul = p[i - bpp] if i >= bpp else 0
                    pa, pb, pc = abs(up - ul), abs(left - ul), abs(left + up - 2 * ul)
                    pr = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
                    x[i] = (x[i] + pr) & 255
            row[:] = x.astype(np.uint8)
        elif t != 0:
            raise ValueError(f"PNG filter type {t}")
        prev = row
    return out


def read_png(path):
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(buf):
        ln, typ = struct.unpack(">I4s", buf[pos:pos + 8])
        body = buf[pos + 8:pos + 8 + ln]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + ln
    W, H, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 4, 6):
        raise NotImplementedError(f"{path}: PNG depth {depth} colour type {ctype} interlace {interlace}")
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    bpp = ch * depth // 8
    data = _png_unfilter(zlib.decompress(b"".join(idat)), H, W * bpp, bpp)
    dt = np.uint8 if depth == 8 else np.dtype(">u2")
    return np.ascontiguousarray(data).view(dt).reshape(H, W, ch).astype(np.uint8 if depth == 8 else np.uint16)


def write_png(path, img):
    img = np.ascontiguousarray(img)
    if img.ndim != 3 or img.shape[2] != 3 or img.dtype not in (np.uint8, np.uint16):
        raise NotImplementedError(f"write_png: shape {img.shape} dtype {img.dtype}")
    H, W, _ = img.shape
    depth = img.dtype.itemsize * 8
    rows = img.astype(img.dtype.newbyteorder(">"), copy=False).reshape(H, -1).view(np.uint8)
    raw = np.concatenate([np.zeros((H, 1), dtype=np.uint8), rows], axis=1).tobytes()

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 3)))
        f.write(chunk(b"IEND", b""))
