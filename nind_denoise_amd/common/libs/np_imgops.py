"""Image reader of the denoise path: file -> float32 CHW RGB in [0,1].
Interface of /root/reference/src/nind_denoise/common/libs/np_imgops.py:12-29 (cv2 replaced by imgcodec / PIL)."""
import os

import numpy as np

from . import imgcodec


def _read_hwc(fpath):
    ext = os.path.splitext(fpath)[1].lower()
    if ext in ('.tif', '.tiff'):
        return imgcodec.read_tiff(fpath)
    if ext == '.png':
        return imgcodec.read_png(fpath)
    from PIL import Image
    return np.asarray(Image.open(fpath).convert('RGB'))


def img_path_to_np_flt(fpath):
    '''returns a numpy float32 array from RGB image path (8-16 bits per component or float32)
    shape: c, y, x'''
    if not os.path.isfile(fpath):
        raise FileNotFoundError(fpath)
    img = _read_hwc(fpath)
    if img.ndim == 2:
        img = img[:, :, None]
    if img.shape[2] == 1:                      # IMREAD_COLOR semantics: gray is replicated
        img = np.repeat(img, 3, axis=2)
    elif img.shape[2] == 2:
        img = np.repeat(img[:, :, :1], 3, axis=2)
    elif img.shape[2] > 3:                     # alpha is dropped
        img = img[:, :, :3]
    rgb_img = np.ascontiguousarray(img.transpose(2, 0, 1))
    if rgb_img.dtype == np.float32:
        return rgb_img
    if rgb_img.dtype == np.ubyte:
        return rgb_img.astype(np.single) / 255
    if rgb_img.dtype == np.ushort:
        return rgb_img.astype(np.single) / 65535
    raise TypeError(f'img_path_to_np_flt: Error: fpath={fpath} has unknown format ({rgb_img.dtype})')


def img_path_to_device_flt(fpath, device):
    '''img_path_to_np_flt with the sample conversion on the GPU: the decoded samples are uploaded as they are stored (a 16-bit
    24 MP frame is 144 MB instead of 288 MB of float32) and transposed / converted there; the frame is bit-identical to
    torch.from_numpy(img_path_to_np_flt(fpath)).to(device).
    Returns a float32 CHW tensor on `device`.'''
    import torch
    if not os.path.isfile(fpath):
        raise FileNotFoundError(fpath)
    img = _read_hwc(fpath)
    if img.ndim == 2:
        img = img[:, :, None]
    # integer samples: the float value of every possible sample comes from a table built on the host with the reference's own
    # expression (astype(np.single) / 65535 resp. / 255), so the GPU only gathers -- bit-identical by construction (torch's GPU
    # division by a scalar multiplies by the reciprocal, which is not)
    if img.dtype == np.ushort:
        idx = torch.from_numpy(np.ascontiguousarray(img).view(np.int16)).to(device).to(torch.int32).bitwise_and_(0xFFFF)
        lut = torch.from_numpy(np.arange(65536, dtype=np.uint16).astype(np.single) / 65535).to(device)
    elif img.dtype == np.ubyte:
        idx = torch.from_numpy(np.ascontiguousarray(img)).to(device).to(torch.int32)
        lut = torch.from_numpy(np.arange(256, dtype=np.uint8).astype(np.single) / 255).to(device)
    elif img.dtype == np.float32:
        idx, lut = torch.from_numpy(np.ascontiguousarray(img)).to(device), None
    else:
        raise TypeError(f'img_path_to_np_flt: Error: fpath={fpath} has unknown format ({img.dtype})')
    c = idx.shape[2]
    if c == 1 or c == 2:                       # IMREAD_COLOR semantics: gray is replicated
        idx = idx[:, :, :1].expand(-1, -1, 3)
    elif c > 3:                                # alpha is dropped
        idx = idx[:, :, :3]
    idx = idx.permute(2, 0, 1).contiguous()
    t = idx if lut is None else lut[idx]
    return t.contiguous()
