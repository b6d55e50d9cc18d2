cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for p in 1 0; do
SNF_PAIR_GRID_BWD=$p python bench.py --steps 100 --warmup 10 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=d['kernel_ms_per_step_serial']
f8=sum(v for n,v in k.items() if 'presorted_adam' in n and 'F8' in n)
print('pair', $p, 'step', round(d['ms_per_step'],3), 'live frac', round(r['frac'],3), 'serial frac', round(r['serial']['frac'],3), 'F8 bwd serial ms/step', round(f8,4))"
done; done
