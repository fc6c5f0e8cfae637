set -e
cd $GRAFT_REPO_ROOT
PROFILE_CONFIG='{"cs": 264, "tiles_per_launch": 256, "funit": 64, "dtype": "f32", "frame": "6000x4000"}' timeout -k 10 500 bash tools/profile_run.sh r03_f32 --steps 3 --warmup 1
PROFILE_CONFIG='{"cs": 264, "tiles_per_launch": 320, "funit": 64, "dtype": "bf16", "frame": "6000x4000"}' timeout -k 10 500 bash tools/profile_run.sh r03_bf16 --steps 3 --warmup 1 --dtype bf16 --batch 320
PROFILE_CONFIG='{"cs": 520, "tiles_per_launch": 80, "funit": 64, "dtype": "f16", "frame": "9504x6336"}' timeout -k 10 500 bash tools/profile_run.sh r03_g61_f16 --steps 2 --warmup 1 --dtype f16 --width 9504 --height 6336 --cs 520 --ucs 456 --ol 64 --batch 80
PROFILE_CONFIG='{"cs": 504, "tiles_per_launch": 64, "funit": 64, "dtype": "f32", "frame": "6000x4000"}' timeout -k 10 500 bash tools/profile_run.sh r03_g24d_f32 --steps 3 --warmup 1 --cs 504 --ucs 480 --ol 6 --batch 64
ls gpurun_out | grep r03_
