"""Image writer / device helper of the denoise path.
Interface of /root/reference/src/nind_denoise/common/libs/pt_helpers.py:13-56 (cv2 / imageio replaced by imgcodec)."""
import numpy as np
import torch

from . import imgcodec, np_imgops, pt_losses


def fpath_to_tensor(img_fpath, device=torch.device(type='cpu'), batch=False):
    tensor = torch.tensor(np_imgops.img_path_to_np_flt(img_fpath), device=device)
    if batch:
        tensor = tensor.unsqueeze(0)
    return tensor


def tensor_to_imgfile(tensor, path):
    """float32 CHW: '.jpg'/'jpeg' -> 8-bit clipped; '.png'/'.tif' -> clip, *65535, round, 16-bit;
    'tiff' -> raw float32, NOT clipped (pt_helpers.py:22-34).  uint8 CHW -> PIL."""
    ext = path[-4:].lower()
    if tensor.dtype == torch.float32:
        if ext in ['.jpg', 'jpeg']:
            from PIL import Image
            arr = tensor.clip(0, 1).mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8).numpy()
            Image.fromarray(arr).save(path)
        elif ext in ['.png', '.tif']:
            if tensor.is_cuda:
                # same IEEE float32 multiply and round-half-even as on the host, done where the canvas lives; the 16-bit HWC image
                # (half the bytes of the float canvas) comes down already in file order
                q = (tensor.clip(0, 1) * 65535).round().to(torch.int32).permute(1, 2, 0).to(torch.int16).contiguous()
                nptensor = q.cpu().numpy().view(np.uint16)
            else:
                nptensor = (tensor.clip(0, 1) * 65535).round().cpu().numpy().astype(np.uint16).transpose(1, 2, 0)
            (imgcodec.write_png if ext == '.png' else imgcodec.write_tiff)(path, nptensor)
        elif ext in ['tiff']:
            if tensor.is_cuda:
                nptensor = tensor.permute(1, 2, 0).contiguous().cpu().numpy()   # (the CHW -> HWC transpose on the GPU)
            else:
                nptensor = tensor.cpu().numpy().astype(np.float32).transpose(1, 2, 0)
            imgcodec.write_tiff(path, nptensor)
        else:
            raise NotImplementedError(f'Extension in {path}')
    elif tensor.dtype == torch.uint8:
        from PIL import Image
        Image.fromarray(tensor.permute(1, 2, 0).cpu().numpy()).save(path)
    else:
        raise NotImplementedError(tensor.dtype)


def get_losses(img1_fpath, img2_fpath, device=None):
    """{'mse', 'ssim', 'msssim'} of two image files (pt_helpers.py:40-48); 'ssim' / 'msssim' are 1 - score.
    The reference scores on the CPU through piqa; here the images go to the GPU and the scores come from libnind_hip.so."""
    device = get_device() if device is None else torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('get_losses needs a GPU (no CPU fallback)')
    img1 = fpath_to_tensor(img1_fpath, device=device).unsqueeze(0)
    img2 = fpath_to_tensor(img2_fpath, device=device).unsqueeze(0)
    assert img1.shape == img2.shape, f'{img1.shape=}, {img2.shape=}'
    res = dict()
    res['mse'] = pt_losses.mse(img1, img2).item()
    res['ssim'] = pt_losses.SSIM_loss()(img1, img2).item()
    res['msssim'] = pt_losses.MS_SSIM_loss()(img1, img2).item()
    return res


def get_device(device_n=None):
    if torch.cuda.is_available():
        return torch.device('cuda', torch.cuda.current_device() if device_n is None else int(device_n))
    print('Accelerator (gpu/xpu/etc.) device not available; defaulting to cpu.')
    return torch.device('cpu')
