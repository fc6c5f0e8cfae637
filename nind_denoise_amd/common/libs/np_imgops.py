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
