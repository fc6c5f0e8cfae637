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
