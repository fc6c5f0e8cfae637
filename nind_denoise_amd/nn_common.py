"""Model-load surface of the denoise path: ``Model.complete_path`` / ``Model.instantiate_model`` with the reference's
signatures and resolution rules (/root/reference/src/nind_denoise/nn_common.py:75-138).

Differences, all supersets:
  * the network registry holds ``UtNet`` (HIP path) and ``UNet``; the reference only imports UtNet into globals()
    (nn_common.py:12) so ``--network UNet`` raises KeyError there;
  * state-dict files are read with ``torch.load(..., weights_only=True)``;
  * a missing model path raises FileNotFoundError after printing the reference's message (the reference exit(0)s).
Weights are repacked for the MFMA kernels lazily on the first forward on a device (networks/UtNet.py).
"""
import json
import os

import torch

from .networks.ThirdPartyNets import UNet
from .networks.UtNet import UtNet

COMMON_CONFIG_FPATH = os.path.join('configs', 'common_conf_default.yaml')

NETWORKS = {'UtNet': UtNet, 'UNet': UNet}


def register_network(name, cls):
    NETWORKS[name] = cls


def default_device():
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None


class Model:
    @staticmethod
    def complete_path(path, models_dpath, keyword=''):
        '''File path for instantiate_model when model_path is a directory (a path, or the name of a directory under
        models_dpath): the best epoch recorded in trainres.json for generators, else the highest epoch present.'''
        def find_highest(paths, model_t):
            best = [None, 0]
            for apath in paths:
                try:
                    curval = int(apath.split('_')[-1].split('.')[0])
                except ValueError:
                    continue
                if curval > best[1] and model_t in apath:
                    best = [apath, curval]
            return best[0]

        def find_best(dpath, model_t):
            if model_t != 'generator':
                return None
            resdpath = os.path.join(dpath, 'trainres.json')
            if not os.path.isfile(resdpath):
                print(f'find_best did not find {resdpath}')
                return None
            with open(resdpath, 'r') as fp:
                best_epoch = json.load(fp)['best_epoch']['validation_loss']
            return os.path.join(dpath, f'generator_{best_epoch}.pt')

        if os.path.isfile(path):
            return path
        if os.path.isdir(path):
            best_model_path = find_best(path, model_t=keyword)
            if best_model_path is not None:
                return best_model_path
            highest = find_highest(os.listdir(path), keyword)
            if highest is None:
                raise FileNotFoundError(f'no {keyword} checkpoint in {path}')
            return os.path.join(path, highest)
        if models_dpath is not None and os.path.isdir(os.path.join(models_dpath, path)):
            return Model.complete_path(os.path.join(models_dpath, path), models_dpath, keyword)
        print("Model path not found: %s" % path)
        raise FileNotFoundError(path)

    @staticmethod
    def instantiate_model(models_dpath=None, model_path=None, network=None, device=None, strparameters=None,
                          pfun=print, keyword='', **parameters):
        '''instantiate the network (and load its weights when model_path is given); returns it on `device`'''
        device = (default_device() or torch.device('cpu')) if device is None else torch.device(device)
        if strparameters is not None and strparameters != "":
            parameters.update(dict([parameter.split('=') for parameter in strparameters.split(',')]))
        if model_path is not None:
            path = Model.complete_path(path=model_path, keyword=keyword, models_dpath=models_dpath)
            if path.endswith('.pth'):
                raise NotImplementedError(
                    f'{path}: whole-module pickles (.pth) are not loaded (unpickling executes code); '
                    'export a state-dict (.pt) instead')
            elif path.endswith('pt'):
                assert network is not None
                model = NETWORKS[network](**parameters)
                model.load_state_dict(torch.load(path, map_location='cpu', weights_only=True))
            else:
                pfun('Error: unable to load invalid model path: %s' % path)
                raise ValueError(path)
        else:
            model = NETWORKS[network](**parameters)
        return model.to(device)
