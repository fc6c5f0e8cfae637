"""The two dataset helpers the eval harness needs: which file of an image set is the clean baseline.
Interface of /root/reference/src/nind_denoise/dataset_torch_3.py:37-96 (``sortISOs``, ``get_baseline_fpath``); the training
dataset classes of that file are out of scope (they need the NIND dataset)."""
import os
from typing import List


def sortISOs(rawISOs: List[str]) -> tuple:
    '''(base ISOs, other ISOs), both sorted.  Names are ISO<NUM>[-REPEAT] or ISOH<NUM>; the lowest ISO number (and its
    -REPEAT duplicates) is the base; ISOH* sort last.  With any other naming, names containing "GT" are the base, else the
    alphabetically first one.'''
    rawISOs = list(rawISOs)
    if any(iso[:3] != 'ISO' for iso in rawISOs):
        bisos = [iso for iso in rawISOs if 'GT' in iso]
        isos = sorted(iso for iso in rawISOs if 'GT' not in iso)
        if not bisos:
            bisos.append(isos.pop(0))
        return bisos, isos
    hisos, nums, dups = [], [], {}
    for iso in rawISOs:
        if 'H' in iso:
            hisos.append(iso)
        elif '-' in iso:
            isoval, _, repid = iso[3:].partition('-')
            nums.append(int(isoval))
            dups.setdefault(isoval, []).append(repid)
        else:
            nums.append(int(iso[3:]))
    base, *rest = sorted(nums)
    bisos = [base]
    while rest and rest[0] == base:          # repeats of the base ISO are bases too
        bisos.append(str(rest.pop(0)) + '-' + dups[str(base)].pop())
    for isoval, repids in dups.items():
        for repid in repids:
            rest[rest.index(int(isoval))] = isoval + '-' + repid
    return ['ISO' + str(i) for i in bisos], ['ISO' + str(i) for i in rest] + sorted(hisos)


def get_baseline_fpath(dpath: str) -> str:
    '''directory of one image set (e.g. NIND/banana with NIND_banana_ISO<value>.png files) -> its baseline image'''
    iso_fn_dict = {fn.split('_')[-1].split('.')[0]: fn for fn in os.listdir(dpath)}
    bisos, _ = sortISOs(iso_fn_dict.keys())
    return os.path.join(dpath, iso_fn_dict[bisos[0]])
