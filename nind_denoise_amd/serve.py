"""Resident serving ("next" row f2 of SURVEY.md section 8): the multi-frame engine and the worker process around it.

The reference pays model load + device init PER IMAGE: `denoise.py:430-436` and `denoise_dir.py:89-98` spawn one
`denoise_image.py` process per frame.  `FrameEngine` keeps the packed weights and the activation workspace resident
and streams frames through a small ring of slots so that, in steady state,

    host->HBM copy of frame n+1   (copy stream, pinned staging)
    crop -> UtNet -> stitch of n  (compute stream, device resident loop of pipeline.denoise_frame)
    HBM->host copy of frame n-1   (copy-back stream, pinned staging)

overlap.  Results come back in submission order.  Nothing here changes the arithmetic: a frame's canvas is bit-identical
to `pipeline.denoise_frame` on the same frame.

The WORKER (`python -m nind_denoise_amd.serve --socket PATH`) is what a per-image caller reaches: it is started once, owns
the GPU context, the loaded models with their packed weights and the activation workspaces, and serves requests that carry
the argument list of `denoise_image.py` (sent by `python -m nind_denoise_amd.denoise_image ... --server PATH`, a client that
imports neither torch nor the HIP library: client.py).  One thread per connection; the device section of a request is
serialised by a lock, so that one request's TIFF decode and another's encode overlap a third one's GPU time.  The worker is
never re-exec'ed and forks nothing after it has touched the GPU.
"""
import collections
import io
import json
import os
import socket
import sys
import threading
import traceback

import numpy as np
import torch

from . import pipeline


class FrameEngine:
    def __init__(self, model, width, height, cs, ucs, ol, batch=160, slots=3, device=None):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise RuntimeError("FrameEngine needs a GPU (no CPU fallback)")
        self.model, self.geom, self.batch = model, (cs, ucs, ol), batch
        self.shape = (3, int(height), int(width))
        self.total = pipeline.tile_count(width, height, cs, ucs, ol)
        with torch.cuda.device(self.device):
            self.compute = torch.cuda.Stream()
            self.h2d = torch.cuda.Stream()
            self.d2h = torch.cuda.Stream()
            self.slots = []
            for _ in range(max(2, slots)):
                self.slots.append(dict(
                    pin_in=torch.empty(self.shape, dtype=torch.float32).pin_memory(),
                    pin_out=torch.empty(self.shape, dtype=torch.float32).pin_memory(),
                    dev_in=torch.empty(self.shape, dtype=torch.float32, device=self.device),
                    dev_out=torch.empty(self.shape, dtype=torch.float32, device=self.device),
                    ev_in=torch.cuda.Event(), ev_done=torch.cuda.Event(), ev_out=torch.cuda.Event(), busy=False))
        self.next = 0
        self.inflight = collections.deque()
        # allocate workspace / pack weights before the first frame (on the compute stream)
        if hasattr(model, "packed_weights"):
            with torch.cuda.device(self.device), torch.cuda.stream(self.compute):
                model.packed_weights(self.device)
                model.workspace(cs, min(batch, self.total), self.device)

    def submit(self, frame):
        """frame: float32 CHW numpy array / CPU tensor (or a CUDA tensor: then no host copy).  Non-blocking unless every
        slot is still in flight."""
        if len(self.inflight) == len(self.slots):
            raise RuntimeError("all slots in flight: call collect() first")
        slot = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        if slot["busy"]:
            slot["ev_out"].synchronize()
        slot["busy"] = True
        cs, ucs, ol = self.geom
        with torch.cuda.device(self.device):
            if isinstance(frame, torch.Tensor) and frame.is_cuda:
                self.compute.wait_stream(torch.cuda.current_stream())
                src = frame
            else:
                t = frame if isinstance(frame, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frame, dtype=np.float32))
                if tuple(t.shape) != self.shape:
                    raise ValueError(f"frame shape {tuple(t.shape)} != engine shape {self.shape}")
                slot["pin_in"].copy_(t)
                with torch.cuda.stream(self.h2d):
                    slot["dev_in"].copy_(slot["pin_in"], non_blocking=True)
                    slot["ev_in"].record()
                self.compute.wait_event(slot["ev_in"])
                src = slot["dev_in"]
            with torch.cuda.stream(self.compute):
                slot["dev_out"].zero_()
                pipeline.denoise_frame(self.model, src, cs, ucs, ol, batch=self.batch, canvas=slot["dev_out"])
                slot["ev_done"].record()
            with torch.cuda.stream(self.d2h):
                self.d2h.wait_event(slot["ev_done"])
                slot["pin_out"].copy_(slot["dev_out"], non_blocking=True)
                slot["ev_out"].record()
        self.inflight.append(slot)

    def collect(self, copy=True):
        """Oldest submitted frame's result as a numpy array (blocks until its copy-back has finished)."""
        slot = self.inflight.popleft()
        slot["ev_out"].synchronize()
        slot["busy"] = False
        out = slot["pin_out"].numpy()
        return out.copy() if copy else out

    def run(self, frames, copy=True):
        """Generator: denoise an iterable of frames, keeping the ring full; yields results in order.  copy=False yields views of
        the pinned output slots (valid until the slot is reused, `slots` frames later)."""
        for f in frames:
            if len(self.inflight) == len(self.slots):
                yield self.collect(copy)
            self.submit(f)
        while self.inflight:
            yield self.collect(copy)


# ---------------------------------------------------------------------------------------------------------------- worker

class _ThreadRouter(io.TextIOBase):
    """sys.stdout / sys.stderr of the worker: text printed by the thread that serves a request goes to that request's client
    (its `sink`), everything else to the worker's own stream."""

    def __init__(self, fallback):
        self.fallback = fallback
        self.local = threading.local()

    def write(self, text):
        sink = getattr(self.local, "sink", None)
        if sink is None:
            return self.fallback.write(text)
        sink(text)
        return len(text)

    def flush(self):
        if getattr(self.local, "sink", None) is None:
            self.fallback.flush()


class Worker:
    def __init__(self, path, log=None):
        self.path = path
        self.models = {}                    # denoise_image.main's model cache: (network, file, mtime, size, parameters, device) -> module
        self.gpu_lock = threading.Lock()
        self.count_lock = threading.Lock()
        self.stop = threading.Event()
        self.served = 0
        self.log = log or (lambda m: None)

    def _serve_one(self, conn, out_router, err_router):
        from . import denoise_image
        with conn, conn.makefile("rwb") as f:
            wlock = threading.Lock()

            def send(obj):
                with wlock:
                    f.write((json.dumps(obj) + "\n").encode())
                    f.flush()
            try:
                line = f.readline()
                if not line:
                    return
                req = json.loads(line)
                if req.get("cmd") == "ping":
                    send({"stream": "stdout", "data": f"nind_denoise_amd worker: {self.served} request(s) served, {len(self.models)} model(s) resident\n"})
                    send({"exit": 0})
                    return
                if req.get("cmd") == "shutdown":
                    send({"exit": 0})
                    self.stop.set()
                    return
                out_router.local.sink = lambda t: send({"stream": "stdout", "data": t})
                err_router.local.sink = lambda t: send({"stream": "stderr", "data": t})
                status = 0
                try:
                    status = denoise_image.main(list(req.get("argv", [])), cwd=req.get("cwd") or os.getcwd(),
                                                model_cache=self.models, gpu_lock=self.gpu_lock) or 0
                except SystemExit as e:       # sys.exit(message) of the CLI: message to stderr, status 1 (as the interpreter does)
                    if isinstance(e.code, int) or e.code is None:
                        status = e.code or 0
                    else:
                        print(e.code, file=sys.stderr)
                        status = 1
                except BaseException:         # a failed request must not take the worker down
                    traceback.print_exc()
                    status = 1
                finally:
                    out_router.local.sink = None
                    err_router.local.sink = None
                with self.count_lock:
                    self.served += 1
                send({"exit": int(status)})
            except (OSError, ValueError) as e:      # client went away / malformed request
                self.log(f"request dropped: {e}")

    def run(self, ready=None):
        import torch
        if not torch.cuda.is_available():
            sys.exit("nind_denoise_amd.serve: no GPU visible; there is no CPU fallback")
        from . import _lib
        _lib.load()
        torch.cuda.init()
        if os.path.exists(self.path):
            os.unlink(self.path)
        out_router, err_router = _ThreadRouter(sys.stdout), _ThreadRouter(sys.stderr)
        sys.stdout, sys.stderr = out_router, err_router
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            srv.bind(self.path)
            srv.listen(64)
            srv.settimeout(0.25)
            print(f"nind_denoise_amd worker ready on {self.path} (pid {os.getpid()})", flush=True)
            if ready is not None:
                ready.set()
            threads = []
            while not self.stop.is_set():
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                conn.settimeout(None)
                t = threading.Thread(target=self._serve_one, args=(conn, out_router, err_router), daemon=True)
                t.start()
                threads = [x for x in threads if x.is_alive()] + [t]
            for t in threads:
                t.join(timeout=60)
        finally:
            srv.close()
            sys.stdout, sys.stderr = out_router.fallback, err_router.fallback
            if os.path.exists(self.path):
                os.unlink(self.path)
        return 0


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Resident denoise worker: owns the GPU, the loaded models and their workspaces; serves "
                                             "`python -m nind_denoise_amd.denoise_image ... --server PATH` clients.")
    ap.add_argument("--socket", required=True, help="Unix socket path to listen on")
    a = ap.parse_args(argv)
    return Worker(a.socket, log=lambda m: print(f"[worker] {m}", file=sys.stderr)).run()


if __name__ == "__main__":
    sys.exit(main())
