cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in 20 100; do
  python $R/bench.py --steps $st --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('plain steps', $st, d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  rm -rf /tmp/rp_$st; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$st -o s -- python $R/bench.py --steps $st --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('rocprof steps', $st, d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/rp_$st/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_hg_reduce<8, true>' in r['Name']: print('  stats', r['Calls'], float(r['AverageNs'])/1e6)
PY
done
