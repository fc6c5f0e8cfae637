"""Result log of training / testing: ``JSONSaver`` with the reference's file format
(/root/reference/src/nind_denoise/common/libs/json_saver.py:9-58): {step: {key: value}, 'best_val': {key: value},
'best_<step_type>': {key: step}}.  One deliberate difference: ``key_prefix`` iterates ``res.items()`` -- the reference
iterates ``res.values()`` there and raises on the first float, so ``denoise_dir`` can never record its 'test_' results."""
from typing import Set

from . import utilities


class JSONSaver:
    def __init__(self, jsonfpath, step_type: str = 'step', default=None):
        self.best_key_str = 'best_{}'.format(step_type)
        self.jsonfpath = jsonfpath
        self.results = utilities.jsonfpath_load(jsonfpath, default={'best_val': dict()} if default is None else default)
        if self.best_key_str not in self.results:
            self.results[self.best_key_str] = dict()
        if 'best_val' not in self.results:
            self.results['best_val'] = dict()

    def add_res(self, step: int, res: dict, minimize=True, write=True, val_type=float, epoch=None, rm_none=False,
                key_prefix=''):
        '''epoch is an alias for step.  Set rm_none to ignore zero values.'''
        if epoch is not None and step is None:
            step = epoch
        elif step is None or epoch is not None:
            raise ValueError('JSONSaver.add_res: Must specify either step or epoch')
        if step not in self.results:
            self.results[step] = dict()
        if key_prefix != '':
            res = {key_prefix + akey: aval for akey, aval in res.items()}
        best_step, best_val = self.results[self.best_key_str], self.results['best_val']
        for akey, aval in res.items():
            if val_type is not None:
                aval = val_type(aval)
            self.results[step][akey] = aval
            if isinstance(aval, list) or (rm_none and aval == 0):
                continue
            if akey not in best_val and akey in best_step:   # best_val was removed but best_step exists
                best_val[akey] = self.results[best_step[akey]][akey]
            if (akey not in best_step or akey not in best_val or (best_val[akey] > aval and minimize)
                    or (best_val[akey] < aval and not minimize)):
                best_step[akey] = step
                best_val[akey] = aval
        if write:
            self.write()

    def write(self):
        utilities.dict_to_json(self.results, self.jsonfpath)

    def get_best_steps(self) -> Set[int]:
        return set(self.results[self.best_key_str].values())
