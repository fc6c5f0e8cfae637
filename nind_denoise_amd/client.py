"""Thin client of the resident denoise worker (`python -m nind_denoise_amd.serve --socket PATH`).

The reference pays interpreter start + `import torch` + device init + model load PER IMAGE: `denoise.py:430-436` and
`denoise_dir.py:89-98` spawn one `denoise_image.py` process per frame.  With a worker running,

    python -m nind_denoise_amd.denoise_image <the usual arguments> --server PATH        (or NIND_DENOISE_SERVER=PATH)

becomes this client: it sends its argument list and working directory over a Unix socket, relays the worker's printed lines
(the reference's own output) and exits with the worker's status.  Standard library only -- it imports neither torch nor
libnind_hip.so, so the per-image cost is a ~30 ms Python start plus the image's own I/O and GPU time inside the worker.

Wire format: newline-delimited JSON.  Request {"argv": [...], "cwd": "..."} or {"cmd": "ping" | "shutdown"}; replies
{"stream": "stdout" | "stderr", "data": "..."} any number of times, then {"exit": status}.
"""
import json
import os
import socket
import sys

ENV = "NIND_DENOISE_SERVER"


def split_server_arg(argv):
    """(socket path or None, argv without the --server option)."""
    out, path, i = [], os.environ.get(ENV) or None, 0
    while i < len(argv):
        a = argv[i]
        if a == "--server":
            if i + 1 >= len(argv):
                sys.exit("--server needs the worker's socket path")
            path = argv[i + 1]
            i += 2
            continue
        if a.startswith("--server="):
            path = a.split("=", 1)[1]
            i += 1
            continue
        out.append(a)
        i += 1
    return path, out


def request(path, msg, out=None, err=None, timeout=None):
    """Send one request, relay the streamed output, return the exit status."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.settimeout(timeout)
    try:
        s.connect(path)
    except OSError as e:
        err.write(f"nind_denoise_amd.client: no worker at {path} ({e}); start one with "
                  f"`python -m nind_denoise_amd.serve --socket {path}`\n")
        return 111
    with s, s.makefile("rwb") as f:
        f.write((json.dumps(msg) + "\n").encode())
        f.flush()
        for line in f:
            m = json.loads(line)
            if "exit" in m:
                out.flush()
                err.flush()
                return int(m["exit"])
            (err if m.get("stream") == "stderr" else out).write(m.get("data", ""))
            out.flush()
    err.write("nind_denoise_amd.client: the worker closed the connection without a status\n")
    return 112


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    path, rest = split_server_arg(argv)
    if path is None:
        sys.exit(f"nind_denoise_amd.client: no worker socket given (--server PATH or {ENV}=PATH)")
    if rest == ["--shutdown"]:
        return request(path, {"cmd": "shutdown"})
    if rest == ["--ping"]:
        return request(path, {"cmd": "ping"})
    return request(path, {"argv": rest, "cwd": os.getcwd()})


if __name__ == "__main__":
    sys.exit(main())
