"""Path helpers used by the denoise CLI (interface of the reference's common/libs/utilities.py:47-58)."""
import os


def get_leaf(path: str) -> str:
    """Leaf of a path, whether it is a file or a directory followed by / or not."""
    return os.path.basename(os.path.relpath(path))


def get_root(fpath: str) -> str:
    """Directory a file is located in."""
    while fpath.endswith(os.pathsep):
        fpath = fpath[:-1]
    return os.path.dirname(fpath)


def get_file_dname(fpath: str) -> str:
    """Name of the directory a file is located in."""
    return os.path.basename(os.path.dirname(fpath))


def jsonfpath_load(fpath, default_type=dict, default=None):
    """JSON file -> dict with digit keys turned into ints (utilities.py:30-42); the default when the file is missing."""
    import json
    if not os.path.isfile(fpath):
        print('jsonfpath_load: warning: {} does not exist, returning default'.format(fpath))
        return default_type() if default is None else default

    def keys2int(x):
        if isinstance(x, dict):
            return {k if not k.isdigit() else int(k): v for k, v in x.items()}
        return x
    with open(fpath, 'r') as f:
        return json.load(f, object_hook=keys2int)


def dict_to_json(adict, fpath):
    import json
    with open(fpath, "w") as f:
        json.dump(adict, f, indent=2)


def avg_listofdicts(listofdicts):
    """Key-wise mean of a list of dicts (utilities.py:61-69).  The reference's function builds the result and then falls
    off the end without returning it, so denoise_dir.py:113-115 crashes on the None; this one returns the dict."""
    import statistics
    res = {akey: [] for akey in listofdicts[0].keys()}
    for adict in listofdicts:
        for akey, aval in adict.items():
            res[akey].append(aval)
    return {akey: statistics.mean(vals) for akey, vals in res.items()}
