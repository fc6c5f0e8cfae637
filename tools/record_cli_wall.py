#!/usr/bin/env python3
"""Wall time per image as a per-image caller (denoise.py:430-436) sees it: one 24 MP 16-bit TIFF through
  (a) python -m nind_denoise_amd.denoise_image ...                 (process per image: interpreter, torch, HIP init, weight packing)
  (b) python -m nind_denoise_amd.denoise_image ... --server PATH   (thin client of the resident worker, serve.py)
Writes one JSON document to stdout.  GPU box only."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from nind_denoise_amd import synth
    from nind_denoise_amd.common.libs import imgcodec
    d = "/tmp/cli"
    os.makedirs(d, exist_ok=True)
    torch.save(synth.make_utnet_state_dict(64, 123), f"{d}/generator_650.pt")
    fr = synth.make_frame(6000, 4000, seed=24)
    imgcodec.write_tiff(f"{d}/in.tif", (fr.transpose(1, 2, 0) * 65535).round().astype(np.uint16))
    env = dict(os.environ, PYTHONPATH=ROOT)
    args = ["--network", "UtNet", "--model_path", f"{d}/generator_650.pt", "-i", f"{d}/in.tif", "--cs", "264", "--ucs", "200", "-ol", "64",
            "-b", "256", "--exif_method", "noexif"]

    def run(extra, out):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "nind_denoise_amd.denoise_image"] + args + ["-o", out] + extra, env=env, cwd=d,
                           capture_output=True, text=True)
        wall = time.time() - t0
        inner = [ln for ln in r.stdout.splitlines() if ln.startswith("Elapsed time")]
        return {"returncode": r.returncode, "wall_s": round(wall, 3), "reference_timer_line": inner[-1] if inner else None,
                "stderr_tail": r.stderr[-300:] if r.returncode else ""}

    doc = {"what": "one 6000x4000 16-bit TIFF, cs 264 / ucs 200 / ol 64, UtNet(64) fp32, wall time of the caller's subprocess",
           "process_per_image": {"float_tiff_out": run([], f"{d}/out_a.tiff")}}
    sock = f"{d}/w.sock"
    worker = subprocess.Popen([sys.executable, "-m", "nind_denoise_amd.serve", "--socket", sock], env=env, cwd=d,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0 = time.time()
        while not os.path.exists(sock):
            if worker.poll() is not None or time.time() - t0 > 300:
                raise RuntimeError("worker did not come up")
            time.sleep(0.05)
        doc["worker_start_s"] = round(time.time() - t0, 3)
        w = {"first_request_float_tiff": run(["--server", sock], f"{d}/out_b.tiff")}      # loads the model, packs, allocates
        w["float_tiff_out"] = [run(["--server", sock], f"{d}/out_b.tiff") for _ in range(4)]
        w["uint16_tif_out"] = [run(["--server", sock], f"{d}/out_c.tif") for _ in range(4)]
        # four clients at once: decode / encode of one overlaps the GPU section of another
        t1 = time.time()
        ps = [subprocess.Popen([sys.executable, "-m", "nind_denoise_amd.denoise_image"] + args + ["-o", f"{d}/out_p{k}.tif", "--server", sock],
                               env=env, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for k in range(4)]
        rcs = [p.wait() for p in ps]
        w["four_concurrent_clients_uint16_tif"] = {"returncodes": rcs, "wall_s_total": round(time.time() - t1, 3),
                                                   "wall_s_per_image": round((time.time() - t1) / 4, 3)}
        doc["resident_worker"] = w
        a = imgcodec.read_tiff(f"{d}/out_a.tiff")
        b = imgcodec.read_tiff(f"{d}/out_b.tiff")
        doc["worker_output_equals_process_output"] = bool(np.array_equal(a, b))
        subprocess.run([sys.executable, "-m", "nind_denoise_amd.client", "--server", sock, "--shutdown"], env=env, cwd=d)
        worker.wait(timeout=60)
    finally:
        if worker.poll() is None:
            worker.kill()
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
