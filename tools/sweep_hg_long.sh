cd $GRAFT_REPO_ROOT
one() { env $2 python bench.py --steps 40 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1'.ljust(16), 'step', round(d['ms_per_step'],3), 'pair serial', round(r['serial']['avg_launch_ms'],4), 'live', round(r['avg_launch_ms'],4))"
}
for v in 16 32 64 128 16 32 48 96; do one "HG_LONG=$v" SNF_HG_LONG=$v; done
