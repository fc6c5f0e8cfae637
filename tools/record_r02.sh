#!/bin/bash
# records the round-2 numbers DESIGN.md quotes (run on the GPU box via gpurun; outputs under gpurun_out/, copied into profiles/)
set -e
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 > gpurun_out/r02_g24_f32_bench.json 2> gpurun_out/r02_g24_f32_bench.err
echo "g24 f32 done"
python bench.py --steps 10 --warmup 3 --cs 504 --ucs 480 --ol 6 --batch 64 --no-cpu-baseline > gpurun_out/r02_g24d_f32_bench.json 2> gpurun_out/r02_g24d_f32_bench.err
echo "g24d done"
python bench.py --steps 10 --warmup 3 --dtype bf16 --batch 160 --no-cpu-baseline > gpurun_out/r02_g24_bf16_bench.json 2> gpurun_out/r02_g24_bf16_bench.err
echo "bf16 done"
python bench.py --steps 5 --warmup 2 --dtype f16 --width 9504 --height 6336 --cs 520 --ucs 456 --ol 64 --batch 40 --no-cpu-baseline > gpurun_out/r02_g61_f16_bench.json 2> gpurun_out/r02_g61_f16_bench.err
echo "g61 f16 done"
python bench.py --steps 2 --warmup 1 --frames 8 --dtype bf16 --batch 160 --no-cpu-baseline --no-host-leg --no-roofline > gpurun_out/r02_frames8_bf16_bench.json 2> gpurun_out/r02_frames8_bf16_bench.err
echo "frames done"
python tools/bench_stream.py --frames 10 --dtype f32 > gpurun_out/r02_stream_f32.json 2>/dev/null
python tools/bench_stream.py --frames 16 --dtype bf16 --batch 160 > gpurun_out/r02_stream_bf16.json 2>/dev/null
echo "stream done"
# CLI wall time: one 24 MP 16-bit TIFF through python -m nind_denoise_amd.denoise_image (process start, model load, read, denoise, write)
python - <<'PY'
import numpy as np, torch, os, time, subprocess, sys, json
sys.path.insert(0, os.getcwd())
from nind_denoise_amd import synth
from nind_denoise_amd.common.libs import imgcodec
os.makedirs("/tmp/cli", exist_ok=True)
torch.save(synth.make_utnet_state_dict(64, 123), "/tmp/cli/generator_650.pt")
fr = synth.make_frame(6000, 4000, seed=24)
imgcodec.write_tiff("/tmp/cli/in.tif", (fr.transpose(1, 2, 0) * 65535).round().astype(np.uint16))
t0 = time.time()
out = subprocess.run([sys.executable, "-m", "nind_denoise_amd.denoise_image", "--network", "UtNet", "--model_path", "/tmp/cli/generator_650.pt", "-i", "/tmp/cli/in.tif",
                      "-o", "/tmp/cli/out.tiff", "--cs", "264", "--ucs", "200", "-ol", "64", "-b", "256", "--exif_method", "noexif"], capture_output=True, text=True)
wall = time.time() - t0
inner = [l for l in out.stdout.splitlines() if l.startswith("Elapsed time")]
json.dump({"what": "python -m nind_denoise_amd.denoise_image on one 6000x4000 16-bit TIFF, cs 264 / ucs 200 / ol 64, float TIFF out", "returncode": out.returncode,
           "wall_s": round(wall, 3), "reference_timer_line": inner[-1] if inner else None}, open("gpurun_out/r02_cli_wall.json", "w"))
print("cli", wall, inner)
PY
echo "cli done"
