'''Test a model: denoise a directory of image sets and score every result against the set's lowest-ISO ground truth.

Command line of the reference's denoise_dir.py (/root/reference/src/nind_denoise/denoise_dir.py:23-46) -- same flags --
e.g.
    python -m nind_denoise_amd.denoise_dir --model_path .../generator_650.pt --network UtNet --cs 504 --ucs 480 \\
        --noisy_dir ../../datasets/test/NIND_504_480
The reference spawns one `python denoise_image.py` process per image (denoise_dir.py:89-98: model load + device init per
image) and scores on the CPU through piqa.  Here the model is loaded once, every image goes through the device-resident
crop -> infer -> stitch loop in this process (denoise_image.denoise_file), and MSE / SSIM / MS-SSIM are computed on the GPU
(common/libs/pt_losses.py -> csrc/ssim.hip).  Results: per-image dicts printed as in the reference, the average written
to <model dir>/trainres.json and testres.json under the 'test_' prefix (json_saver.JSONSaver).

Known reference defects not reproduced (they make its script crash before any result is written): utilities.avg_listofdicts
returns None; JSONSaver.add_res(key_prefix=...) iterates res.values().  The obsolete loss.gen_score tail (--no_scoring,
needs the pytorch_ssim package) is accepted as a flag and skipped with a message.
'''
import argparse
import os
import sys

import torch
import yaml

from . import dataset_torch_3, denoise_image, nn_common
from .common.libs import json_saver, pt_helpers, utilities
from .networks.UtNet import valid_cs


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('--config', default=nn_common.COMMON_CONFIG_FPATH, help='YAML file with default values (models_dpath, ...)')
    parser.add_argument('--noisy_dir', type=str, help='directory of test dataset (or any directory containing images to be denoised), must end with [CROPSIZE]_[USEFULCROPSIZE]')
    parser.add_argument('--g_network', '--network', type=str, help='Generator network architecture (typically UtNet or UNet)')
    parser.add_argument('--model_path', '--model_fpath', help='Generator pretrained model path (.pt for dictionary)')
    parser.add_argument('--model_parameters', default="", type=str, help='Model parameters with format "parameter1=value1,parameter2=value2"')
    parser.add_argument('--result_dir', default='../../results/NIND/test', type=str, help='directory where results are saved. Can also be set to "make_subdirs" to make a denoised/<model_directory_name> subdirectory')
    parser.add_argument('--no_scoring', action='store_true', help='Skip the obsolete res.txt scoring pass')
    parser.add_argument('--cs', type=str)
    parser.add_argument('--ucs', type=str)
    parser.add_argument('-ol', '--overlap', default=6, type=int, help='Merge crops with this much overlap (denoise_image default)')
    parser.add_argument('-b', '--batch_size', type=int, default=None, help='Tiles per launch of the conv stack')
    parser.add_argument('--skip_existing', action='store_true', help='Skip existing files')
    parser.add_argument('--whole_image', action='store_true', help='Ignore cs and ucs, denoise whole image')
    parser.add_argument('--pad', type=int, help='Padding amt per side, only used for whole image (otherwise (cs-ucs)/2')
    parser.add_argument('--max_subpixels', type=int, help='Max number of pixels, otherwise abort.')
    parser.add_argument('--test_reserve', nargs='*', help='Space separated list of image sets reserved for testing, or yaml file path containing a list. Can be used like in training in place of noisy_dir argument.')
    parser.add_argument('--orig_data', help='Location of the originally downloaded train data (before cropping); used with test_reserve')
    parser.add_argument('--models_dpath', help='Directory where all models are saved')
    return parser


def parse_args(argv=None):
    args, _ = build_parser().parse_known_args(argv)
    if args.config and os.path.isfile(args.config):
        with open(args.config, 'r') as f:
            conf = yaml.safe_load(f) or {}
        for k, v in conf.items():
            if hasattr(args, k) and getattr(args, k) is None:
                setattr(args, k, v)
    for k in ('cs', 'ucs'):          # the reference declares them as strings and forwards them to denoise_image.py
        if getattr(args, k) is not None:
            setattr(args, k, int(getattr(args, k)))
    return args


def get_test_reserve_list(test_reserve):
    '''test_reserve argument (list, or one yaml path, or "0") -> list of set names (nn_common.py:149-160)'''
    if len(test_reserve) == 1:
        if test_reserve[0].endswith('.yaml'):
            with open(test_reserve[0], 'r') as fp:
                return yaml.safe_load(fp)
        elif test_reserve[0] == '0':
            return []
    return test_reserve


def main(argv=None):
    args = parse_args(argv)
    assert args.model_path is not None
    if not torch.cuda.is_available():
        sys.exit('denoise_dir: no GPU visible; nind_denoise_amd has no CPU fallback')
    denoise_image.autodetect_network_cs_ucs(args)
    device = nn_common.default_device()
    model_path = nn_common.Model.complete_path(args.model_path, keyword='generator', models_dpath=args.models_dpath)
    if args.noisy_dir is not None:
        sets_to_denoise = sorted(os.listdir(args.noisy_dir))
        if os.path.isfile(os.path.join(args.noisy_dir, sets_to_denoise[0])):
            sets_to_denoise = ['.']   # just a directory containing images
        if args.result_dir == 'make_subdirs':
            denoised_save_dir = os.path.join(args.noisy_dir, '..', 'denoised', utilities.get_file_dname(args.model_path),
                                             utilities.get_leaf(args.noisy_dir))
        else:
            denoised_save_dir = os.path.join(args.result_dir, model_path.split('/')[-2])
    else:
        sets_to_denoise = get_test_reserve_list(args.test_reserve)
        args.noisy_dir = args.orig_data
        if len(args.test_reserve) == 1 and os.path.isfile(args.test_reserve[0]):
            test_set_str = utilities.get_leaf(args.test_reserve[0])
        else:
            test_set_str = str(args.test_reserve)
        denoised_save_dir = os.path.join(utilities.get_root(args.model_path), 'test', utilities.get_leaf(args.model_path), test_set_str)
    os.makedirs(denoised_save_dir, exist_ok=True)

    if args.g_network == 'UtNet' and not args.whole_image and not valid_cs(args.cs):
        sys.exit(f'denoise_dir: --cs {args.cs} is not a valid UtNet tile size (16k+56); the reference network fails on it too')
    model = nn_common.Model.instantiate_model(network=args.g_network, model_path=model_path,
                                              strparameters=args.model_parameters or None, keyword='generator',
                                              device=device, models_dpath=args.models_dpath)
    model = model.eval().to(device)

    losses_per_set = list()
    for aset in sets_to_denoise:
        losses_per_img = list()
        aset_indir = os.path.join(args.noisy_dir, aset)
        baseline_fpath = dataset_torch_3.get_baseline_fpath(aset_indir)
        for animg in sorted(os.listdir(aset_indir)):
            inimg_path = os.path.join(aset_indir, animg)
            if baseline_fpath == inimg_path or not os.path.isfile(inimg_path):
                continue
            outimg_path = os.path.join(denoised_save_dir, animg)
            if outimg_path.endswith('jpg'):
                outimg_path = outimg_path + '.tif'
            if not (os.path.isfile(outimg_path) and args.skip_existing):
                denoise_image.denoise_file(model, inimg_path, outimg_path, args.cs, args.ucs, args.overlap,
                                           batch=args.batch_size or 64, whole_image=args.whole_image,
                                           pad=128 if args.whole_image else args.pad, max_subpixels=args.max_subpixels,
                                           device=device, verbose=False)
            cur_losses = pt_helpers.get_losses(baseline_fpath, outimg_path, device=device)
            print(f'in: {inimg_path}, out: {outimg_path}, clean: {baseline_fpath}')
            print(cur_losses)
            losses_per_img.append(cur_losses)
        if losses_per_img:
            losses_per_set.append(utilities.avg_listofdicts(losses_per_img))
    if not losses_per_set:
        sys.exit('denoise_dir: nothing to score (every set holds only its baseline image)')
    losses_per_set = utilities.avg_listofdicts(losses_per_set)
    print(losses_per_set)

    # results next to the model: trainres.json (best effort: training may be rewriting it) and testres.json
    try:
        epoch = int(utilities.get_leaf(args.model_path).split('_')[1].split('.')[0])
    except (ValueError, IndexError) as e:
        print(f'Cannot determine epoch from model_path {args.model_path} ({e})')
        epoch = None
    model_root = utilities.get_root(model_path)
    if epoch is not None:
        for fn in ('trainres.json', 'testres.json'):
            json_res_fpath = os.path.join(model_root, fn)
            if fn == 'trainres.json' and not os.path.isfile(json_res_fpath):
                print(f'Model results json file not found ({json_res_fpath})')
                continue
            jsonsaver = json_saver.JSONSaver(json_res_fpath, step_type='epoch')
            jsonsaver.add_res(step=epoch, res=losses_per_set, key_prefix='test_')
    else:
        json_res_fpath = os.path.join(model_root, 'testres.json')
        print(f'results will be dumped to {json_res_fpath}.')
        utilities.dict_to_json(losses_per_set, json_res_fpath)
    if not args.no_scoring:
        print('denoise_dir: the obsolete res.txt scoring pass (loss.gen_score, needs pytorch_ssim) is not run; '
              'the scores above come from pt_helpers.get_losses')
    return losses_per_set


if __name__ == '__main__':
    main()
    sys.exit(0)
