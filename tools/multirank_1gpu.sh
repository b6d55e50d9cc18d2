cd $GRAFT_REPO_ROOT
for s in 1 0; do
  echo "== SNF_STATIC_STEP=$s"
  SNF_STATIC_STEP=$s SNF_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'host', d['host'], 'rccl', d['rccl'])
print('fwd_bwd_only', d['fwd_bwd_only']['ms_per_step'])
"
done
