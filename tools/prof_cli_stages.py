#!/usr/bin/env python3
"""Where the time of one 24 MP image goes inside denoise_file (GPU box): decode, upload, device loop, download, encode."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nind_denoise_amd import pipeline, synth  # noqa: E402
from nind_denoise_amd.common.libs import imgcodec, np_imgops, pt_helpers  # noqa: E402
from nind_denoise_amd.networks.UtNet import UtNet  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    net = UtNet(64)
    net.load_state_dict(synth.make_utnet_state_dict(64, 123))
    net = net.eval().to(dev)
    fr = synth.make_frame(6000, 4000, seed=24)
    os.makedirs("/tmp/cli", exist_ok=True)
    imgcodec.write_tiff("/tmp/cli/in.tif", (fr.transpose(1, 2, 0) * 65535).round().astype(np.uint16))
    for rep in range(3):
        t = [time.time()]
        raw = imgcodec.read_tiff("/tmp/cli/in.tif"); t.append(time.time())
        a = np_imgops.img_path_to_np_flt("/tmp/cli/in.tif"); t.append(time.time())
        x = torch.from_numpy(a).to(dev); torch.cuda.synchronize(); t.append(time.time())
        y = pipeline.denoise_frame(net, x, 264, 200, 64, batch=256); torch.cuda.synchronize(); t.append(time.time())
        h = y.cpu(); t.append(time.time())
        pt_helpers.tensor_to_imgfile(h, "/tmp/cli/o.tiff"); t.append(time.time())
        pt_helpers.tensor_to_imgfile(h, "/tmp/cli/o.tif"); t.append(time.time())
        names = ["read_tiff only (u16 HWC)", "img_path_to_np_flt (read + CHW float)", "upload", "device loop", "download", "write float tiff", "write u16 tif"]
        print(rep, {n: round(t[i + 1] - t[i], 3) for i, n in enumerate(names)}, flush=True)


if __name__ == "__main__":
    main()
