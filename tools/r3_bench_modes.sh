set -e
cd $GRAFT_REPO_ROOT
S="--width 1500 --height 1100 --funit 16 --batch 32 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-host-leg --no-whole-leg"
python bench.py $S --frames 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=1 frames', d['value'], d['scaling'], d['config']['parallelism'])"
ND_BENCH_REHEARSAL=1 python bench.py --gpus 2 $S --frames 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2 frames', d['value'], d['scaling'], d['config']['parallelism'])"
ND_BENCH_REHEARSAL=1 python bench.py --gpus 2 $S --frame-per-rank 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2 replicas', d['value'], d['scaling'], d['config']['parallelism'])"
ND_BENCH_REHEARSAL=1 python bench.py --gpus 3 $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=3 stream', d['value'], d['scaling'], d['pipeline'])"
python bench.py $S --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=1 bf16', d['value'], d['scaling'], 'other_configs' in d)"
