'''
Image cropper with overlap -- MI355X native.

Denoise an image: crop it into overlapping tiles with mirror padding, run every tile through the network, stitch the
useful centres back with seamless overlap blending.  Command line, defaults, printed lines and exit behaviour follow
/root/reference/src/nind_denoise/denoise_image.py:181-283; the crop -> infer -> stitch loop itself runs device resident
through libnind_hip.so (pipeline.denoise_frame).

egrun:
    python -m nind_denoise_amd.denoise_image --network UtNet --model_path generator_650.pt -i in.tif -o out.tiff
'''
import os
import sys

if __name__ == '__main__':
    # thin-client mode (--server PATH or NIND_DENOISE_SERVER): hand the arguments to the resident worker BEFORE anything heavy is
    # imported -- neither torch nor libnind_hip.so is loaded in this process (nind_denoise_amd/client.py, serve.py)
    from nind_denoise_amd import client as _client
    _server, _rest = _client.split_server_arg(sys.argv[1:])
    if _server is not None:
        sys.exit(_client.request(_server, {'argv': _rest, 'cwd': os.getcwd()}))

import argparse
import math
import time

import torch
import yaml

from . import _lib, nn_common, pipeline
from .common.libs import np_imgops, pt_helpers, utilities
from .networks.UtNet import nearest_valid_cs, valid_cs

CS_UNET, UCS_UNET = 440, 320
CS_UTNET, UCS_UTNET = 504, 480
CS_UNK, UCS_UNK = 512, 448


def make_output_fpath(input_fpath, model_fpath):
    model_dpath = utilities.get_root(model_fpath)
    model_fn = utilities.get_leaf(model_fpath)
    img_fn = utilities.get_leaf(input_fpath)
    os.makedirs(os.path.join(model_dpath, 'test', 'denoised_images'), exist_ok=True)
    return os.path.join(model_dpath, 'test', 'denoised_images', f'{img_fn}_{model_fn}.tif')


def autodetect_network_cs_ucs(args) -> None:
    '''Reference rule (denoise_image.py:59-79): if EITHER cs or ucs is missing BOTH are replaced by the defaults.'''
    if args.g_network is None:
        print('network parameter not specified')
        if 'unet' in args.model_path.lower():
            args.g_network = 'UNet'
        elif 'utnet' in args.model_path.lower():
            args.g_network = 'UtNet'
        else:
            sys.exit('Could not determine network architecture from path. Please specify a "--network" type (typically UNet or UtNet)')
        print(f'Assuming {args.g_network} from path')
    if args.cs is None or args.ucs is None:
        print('cs and/or ucs not set, using defaults ...')
        if args.g_network == 'UNet':
            args.cs, args.ucs = CS_UNET, UCS_UNET
        elif args.g_network == 'UtNet':
            args.cs, args.ucs = CS_UTNET, UCS_UTNET
        else:
            print('Warning: cs and ucs not known for this architecture; values may be sub-optimal')
            args.cs, args.ucs = CS_UNK, UCS_UNK
        print(f'cs={args.cs}, ucs={args.ucs}')


class OneImageDS(torch.utils.data.Dataset):
    '''
    Single-image dataset which crops an image into equal pieces with overlap and adds mirror padding to the edges
    (denoise_image.py:81-177).  The decoded frame is kept in HBM; items are gathered on the GPU by nd_tile_gather and
    returned as the reference's triple (tile [3,cs,cs] float32, usefuldim IntTensor[4], usefulstart IntTensor[2]).
    `inimg_fpath` may also be an in-memory float32 CHW array / tensor.
    '''
    def __init__(self, inimg_fpath, cs, ucs, ol, whole_image=False, pad=None, device=None):
        if isinstance(inimg_fpath, (str, os.PathLike)):
            inimg = torch.from_numpy(np_imgops.img_path_to_np_flt(inimg_fpath))
        else:
            inimg = torch.as_tensor(inimg_fpath, dtype=torch.float32)
        self.device = torch.device(device) if device is not None else nn_common.default_device()
        if self.device is None or self.device.type != 'cuda':
            raise RuntimeError('OneImageDS: no GPU; nind_denoise_amd has no CPU fallback')
        self.inimg = inimg.to(self.device).contiguous()
        self.width, self.height = self.inimg.shape[2], self.inimg.shape[1]
        if whole_image:
            self.pad = pad
            if self.pad is None or self.pad == 0:
                self.pad = 0
                print('OneImageDS: Warning: you should really consider (pad>0)')
            self.whole_image = True
            self.size = 1
        else:
            self.whole_image = False
            self.cs, self.ucs, self.ol = cs, ucs, ol
            cols, rows, self.pad = _lib.tile_grid(self.width, self.height, cs, ucs, ol)
            self.iperhl = cols - 1
            self.size = cols * rows
            if self.pad == 0:
                print('OneImageDS: Warning: you should really consider padding (cs>ucs)')

    def __getitem__(self, i):
        if i < 0 or i >= self.size:
            raise IndexError(i)
        if self.whole_image:
            # one item: the whole frame with its four SIDE bands mirrored (edge pixel repeated) and the four pad x pad
            # corners left at zero, exactly as the reference builds it (denoise_image.py:110-128 mirrors the sides only);
            # its (W+2p, H+2p) allocation is only right for square frames -- H, W are used in their places here
            p = self.pad
            if p:
                def sym(n):   # edge pixel repeated, like np.flip of the adjacent band
                    i = torch.arange(-p, n + p, device=self.inimg.device)
                    return torch.where(i < 0, -1 - i, torch.where(i >= n, 2 * n - 1 - i, i))
                ret = self.inimg[:, sym(self.height)][:, :, sym(self.width)].contiguous()
                for ys in (slice(0, p), slice(self.height + p, None)):
                    for xs in (slice(0, p), slice(self.width + p, None)):
                        ret[:, ys, xs] = 0
            else:
                ret = self.inimg
            usefuldim = (p, p, self.width + p, self.height + p)
            usefulstart = (p, p)
            return ret, torch.IntTensor(usefuldim), torch.IntTensor(usefulstart)
        tile = pipeline.gather_tiles(self.inimg, self.cs, self.ucs, self.ol, i, 1)[0]
        _, _, usefuldim, usefulstart = _lib.tile_geom(i, self.width, self.height, self.cs, self.ucs, self.ol)
        return tile, torch.IntTensor(usefuldim), torch.IntTensor(usefulstart)

    def __len__(self):
        return self.size


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('--config', default=nn_common.COMMON_CONFIG_FPATH, help='YAML file with default values (models_dpath, ...)')
    parser.add_argument('--cs', type=int, help='Tile size (UtNet: 16k+56, e.g. 264 or 504)')
    parser.add_argument('--ucs', type=int, help='Useful tile size (should be <=.75*cs for U-Net, a smaller value may result in less grid artifacts but costs computation time')
    parser.add_argument('-ol', '--overlap', default=6, type=int, help='Merge crops with this much overlap (Reduces grid artifacts, may reduce sharpness between crops, costs computation time)')
    parser.add_argument('-i', '--input', default='in.jpg', type=str, help='Input image file')
    parser.add_argument('-o', '--output', type=str, help='Output file with extension (default: model_dpath/test/denoised_images/fn.tif)')
    parser.add_argument('-b', '--batch_size', type=int, default=None, help='Tiles per launch of the conv stack (results do not depend on it; default 64)')
    parser.add_argument('--debug', action='store_true', help='Debug (display useful messages)')
    parser.add_argument('--exif_method', default='piexif', type=str, help='How is exif data copied over? (piexif, exiftool, noexif)')
    parser.add_argument('--g_network', '--network', '--arch', type=str, help='Generator network (typically UNet or UtNet)')
    parser.add_argument('--model_path', help='Generator pretrained model path (.pt for dictionary), required')
    parser.add_argument('--model_parameters', type=str, help='Model parameters with format "parameter1=value1,parameter2=value2"')
    parser.add_argument('--max_subpixels', type=int, help='Max. number of sub-pixels, abort if exceeded.')
    parser.add_argument('--whole_image', action='store_true', help='Ignore cs and ucs, denoise whole image')
    parser.add_argument('--pad', type=int, help='Padding amt per side, only used for whole image (otherwise (cs-ucs)/2')
    parser.add_argument('--models_dpath', help='Directory where all models are saved (used when a model name is provided as model_path)')
    return parser


def parse_args(argv=None, cwd=None):
    parser = build_parser()
    args, _ = parser.parse_known_args(argv)
    if cwd is not None and args.config and not os.path.isabs(args.config):
        args.config = os.path.join(cwd, args.config)    # (a worker's request: the client's directory)
    # configargparse behaviour of the reference: values of the default YAML file fill options left unset
    if args.config and os.path.isfile(args.config):
        with open(args.config, 'r') as f:
            conf = yaml.safe_load(f) or {}
        for k, v in conf.items():
            if hasattr(args, k) and getattr(args, k) is None:
                setattr(args, k, v)
    return args


def copy_exif(args):
    '''EXIF transplant (denoise_image.py:272-279): needs exiv2 / piexif, neither ships with this image.'''
    if args.exif_method == 'noexif':
        return
    try:
        import exiv2
    except ImportError:
        print(f'exif_method={args.exif_method}: exiv2 is not installed, metadata not copied (use --exif_method noexif to silence)')
        return
    exiv_src = exiv2.ImageFactory.open(args.input)
    exiv_src.readMetadata()
    exiv_dst = exiv2.ImageFactory.open(args.output)
    exiv_dst.setExifData(exiv_src.exifData())
    exiv_dst.writeMetadata()


def _save_dbg_jpg(t, path):
    """torchvision.utils.save_image semantics for one [3,h,w] float tensor: clamp to [0,1], * 255 + 0.5, uint8, PIL."""
    from PIL import Image
    arr = t.detach().float().clamp(0, 1).mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8).numpy()
    Image.fromarray(arr).save(path)


def _denoise_frame_debug(model, ds, cs, ucs, overlap, batch, outpath, dbg_dir='dbg'):
    """--debug (denoise_image.py:149-150, 260-269): the loop tile by tile with the reference's crop dumps in ./dbg --
    crop<batch>_<i>_denoised.jpg (the network's whole output tile), _tensimg.jpg (useful crop with halved overlap strips),
    _noisy.jpg (the input tile) -- and the last output tile with its borders as <output>dbg_inclborders.tif.  The canvas
    is built by the same device stitch kernel as the fast path (identical result, just not fused)."""
    os.makedirs(dbg_dir, exist_ok=True)
    img = ds.inimg
    H, W = img.size(1), img.size(2)
    canvas = torch.zeros_like(img)
    last = None
    for n_count, t0 in enumerate(range(0, len(ds), batch)):
        cnt = min(batch, len(ds) - t0)
        print(str(n_count) + '/' + str(int(len(ds) / batch)))
        ybatch = pipeline.gather_tiles(img, cs, ucs, overlap, t0, cnt)
        xbatch = model(ybatch)
        pipeline.stitch_tiles(canvas, xbatch, cs, ucs, overlap, t0)
        for i in range(cnt):
            _, _, ud, us = _lib.tile_geom(t0 + i, W, H, cs, ucs, overlap)
            absx0, absy0 = us
            tensimg = xbatch[i][:, ud[1]:ud[3], ud[0]:ud[2]].clone()
            if absx0 != 0:
                tensimg[:, :, 0:overlap] /= 2
            if absy0 != 0:
                tensimg[:, 0:overlap, :] /= 2
            if absx0 + ucs < W and overlap:
                tensimg[:, :, -overlap:] /= 2
            if absy0 + ucs < H and overlap:
                tensimg[:, -overlap:, :] /= 2
            _save_dbg_jpg(xbatch[i], dbg_dir + '/crop' + str(n_count) + '_' + str(i) + '_denoised.jpg')
            _save_dbg_jpg(tensimg, dbg_dir + '/crop' + str(n_count) + '_' + str(i) + '_tensimg.jpg')
            _save_dbg_jpg(ybatch[i], dbg_dir + '/crop' + str(n_count) + '_' + str(i) + '_noisy.jpg')
            print(tensimg.shape)
            print((absx0, absy0, ud))
            last = xbatch[i]
    if last is not None:
        pt_helpers.tensor_to_imgfile(last.clip(0, 1).cpu(), outpath + 'dbg_inclborders.tif')
    return canvas


def denoise_file(model, inpath, outpath, cs, ucs, overlap, batch=32, whole_image=False, pad=None, max_subpixels=None,
                 device=None, verbose=True, debug=False, gpu_lock=None, dbg_dir='dbg'):
    '''One image file through the device-resident crop -> infer -> stitch loop (the body of the reference's __main__,
    denoise_image.py:228-270); also what denoise_dir runs per image, in process, instead of spawning this script.
    gpu_lock (resident worker): held for the device section only, so that the decode of one request and the encode of another
    overlap a third one's GPU time.'''
    import contextlib
    device = torch.device(device) if device is not None else nn_common.default_device()
    if device is None or device.type != 'cuda':
        raise RuntimeError('denoise_file: no GPU; nind_denoise_amd has no CPU fallback')
    raw_path = isinstance(inpath, (str, os.PathLike))
    if raw_path and not os.path.isfile(inpath):
        raise FileNotFoundError(inpath)
    with (gpu_lock if gpu_lock is not None else contextlib.nullcontext()):
        # (decoded samples are uploaded as stored and converted to float32 on the GPU: np_imgops.img_path_to_device_flt)
        frame = np_imgops.img_path_to_device_flt(inpath, device) if raw_path else inpath
        ds = OneImageDS(frame, cs, ucs, overlap, whole_image=whole_image, pad=pad, device=device)
        if whole_image:
            ybatch, usefuldims, _ = ds[0]
            ybatch = ybatch[None]
            if max_subpixels is not None and math.prod(ybatch.shape) > max_subpixels:
                sys.exit(f'denoise_image.py: {ybatch.shape=}, {math.prod(ybatch.shape)=} > {max_subpixels=} for {inpath=}; aborting')
            ud = usefuldims.tolist()
            newimg = model(ybatch)[0][:, ud[1]:ud[3], ud[0]:ud[2]]
        else:
            if max_subpixels is not None and batch * 3 * cs * cs > max_subpixels:
                batch = max(1, max_subpixels // (3 * cs * cs))
                if 3 * cs * cs > max_subpixels:
                    sys.exit(f'denoise_image.py: tile of {3 * cs * cs} sub-pixels > {max_subpixels=} for {inpath=}; aborting')
            nbatches = int(math.ceil(len(ds) / batch))

            def progress(n, t0, cnt):
                if verbose:
                    print(str(n) + '/' + str(nbatches))
            if debug:
                newimg = _denoise_frame_debug(model, ds, cs, ucs, overlap, batch, outpath, dbg_dir)
            else:
                newimg = pipeline.denoise_frame(model, ds.inimg, cs, ucs, overlap, batch=batch, progress=progress)
        # (sample conversion / HWC transpose on the GPU, then the download: the device section ends inside this call)
        pt_helpers.tensor_to_imgfile(newimg, outpath)
    return newimg


def _absolutize(args, cwd):
    '''A worker runs requests of clients that live in other working directories: resolve their relative paths there.'''
    for name in ('input', 'output', 'config', 'models_dpath'):
        v = getattr(args, name, None)
        if v and not os.path.isabs(v):
            setattr(args, name, os.path.join(cwd, v))
    v = args.model_path
    if v and not os.path.isabs(v) and os.path.exists(os.path.join(cwd, v)):   # (else: a model NAME under models_dpath)
        args.model_path = os.path.join(cwd, v)


def main(argv=None, cwd=None, model_cache=None, gpu_lock=None):
    '''The reference's __main__ (denoise_image.py:215-283).  cwd / model_cache / gpu_lock are the resident worker's
    (serve.py): relative paths resolve against the client's directory, a model loaded for one request stays resident (packed
    weights and workspace with it) for the next, and the device section of concurrent requests is serialised while their
    file I/O overlaps.'''
    args = parse_args(argv, cwd)
    if cwd is not None:
        _absolutize(args, cwd)
    assert args.model_path is not None
    autodetect_network_cs_ucs(args)
    if not torch.cuda.is_available():
        sys.exit('denoise_image: no GPU visible; nind_denoise_amd has no CPU fallback')
    torch.manual_seed(123)
    device = nn_common.default_device()
    if args.output is None:
        args.output = make_output_fpath(args.input, args.model_path)
    if args.model_parameters is None and 'activation' in args.model_path:
        args.model_parameters = f"activation={args.model_path.split('activation')[-1].split('_')[1].split('_')[0]}"
        print(f'set model_parameters to {args.model_parameters} based on model_path')
    if args.g_network == 'UtNet' and not args.whole_image and not valid_cs(args.cs):
        sys.exit(f'denoise_image: --cs {args.cs} is not a valid UtNet tile size (16k+56, e.g. {nearest_valid_cs(args.cs)}); '
                 'the reference network fails on it too')

    def load():
        m = nn_common.Model.instantiate_model(network=args.g_network, model_path=args.model_path,
                                              strparameters=args.model_parameters, keyword='generator',
                                              device=device, models_dpath=args.models_dpath)
        m.eval()
        return m.to(device)
    if model_cache is None:
        model = load()
    else:
        path = nn_common.Model.complete_path(path=args.model_path, keyword='generator', models_dpath=args.models_dpath)
        st = os.stat(path)
        key = (args.g_network, os.path.realpath(path), st.st_mtime_ns, st.st_size, args.model_parameters or '', str(device))
        import contextlib
        with (gpu_lock if gpu_lock is not None else contextlib.nullcontext()):   # (two first requests must not both load the model)
            model = model_cache.get(key)
            if model is None:
                model = model_cache[key] = load()
    start_time = time.time()
    denoise_file(model, args.input, args.output, args.cs, args.ucs, args.overlap, batch=args.batch_size or 64,
                 whole_image=args.whole_image, pad=args.pad, max_subpixels=args.max_subpixels, device=device, debug=args.debug,
                 gpu_lock=gpu_lock, dbg_dir=os.path.join(cwd, 'dbg') if cwd is not None else 'dbg')
    print(f'Denoised image written to {args.output}')
    copy_exif(args)
    print(f'Wrote denoised image to {args.output}')
    print('Elapsed time: ' + str(time.time() - start_time) + ' seconds')
    return 0


if __name__ == '__main__':
    sys.exit(main())
