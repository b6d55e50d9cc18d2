cd $GRAFT_REPO_ROOT
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-baseline-seconds 0 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['fwd_bwd_only']['ms_per_step'],3), {k:v for k,v in list(d['kernel_ms_per_step_serial'].items())[:8]})"; }
python bench.py --steps 30 --warmup 5 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', round(d['ms_per_step'],3), round(d['fwd_bwd_only']['ms_per_step'],3))"
SNF_FORCE_COLLECTIVES=1 SNF_TABLE_PARALLEL=1 run tp
SNF_FORCE_COLLECTIVES=1 SNF_TABLE_PARALLEL=0 run zero
SNF_FORCE_COLLECTIVES=1 SNF_TABLE_PARALLEL=1 run tp
python bench.py --steps 30 --warmup 5 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', round(d['ms_per_step'],3), round(d['fwd_bwd_only']['ms_per_step'],3))"
