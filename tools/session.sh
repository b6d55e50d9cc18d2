#!/bin/bash
# One gpurun session = one invocation of this script on the GPU box:  gpurun -- 'bash tools/session.sh <tag> <step> [<step> ...]'
# Every step appends to gpurun_out/<tag>/out.txt (merged back into the build container).  Steps:
#   bench[:ENV=V,ENV=V]   bench.py with the driver's arguments (--steps 20 --warmup 5), no CPU baseline / other workloads; one summary line
#   benchfull             the default bench.py line (all workloads, CPU baseline) -> bench.json
#   ab:ENV=V[,ENV=V]      off / on / off / on of an environment switch with the driver's arguments
#   test:<pytest -k expr> pytest -m gpu -k <expr>
#   tests                 the whole GPU suite + smoke
#   py:<file> [args]      a tools/ script
#   abvar:A,B[:keys]      tools/ab_bench.sh over prebuilt tools/ab/lib{A,B}.so (tools/build_variant.sh), ROUNDS=2; keys = serial kernel times shown
#   sh:<command>          any shell command (output appended)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
SUM='import sys,json
ls=[l for l in sys.stdin.read().splitlines() if l.startswith("{")]
d=json.loads(ls[-1])
r=d.get("roofline") or {}
print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "value", "%.4g"%d["value"], "step_frac", round(d["step_frac_of_hbm_peak"],4), "dom", r.get("kernel"), "live_frac", r.get("frac"), "serial_frac", (r.get("serial") or {}).get("frac"), "steady", (d.get("steady_state") or {}).get("ms_per_step"), "fb", (d.get("fwd_bwd_only") or {}).get("ms_per_step"))'
bench_once() {  # $1 label, rest: env assignments
  local label=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 2>$OUT/last.err | python -c "$SUM" "$label" >> $OUT/out.txt 2>&1 || { echo "$label FAILED" >> $OUT/out.txt; tail -5 $OUT/last.err >> $OUT/out.txt; }
}
for step in "$@"; do
  echo "== $step" >> $OUT/out.txt
  case "$step" in
    bench) bench_once base A=1 ;;
    bench:*) IFS=, read -ra E <<< "${step#bench:}"; bench_once "${step#bench:}" "${E[@]}" ;;
    benchfull) python bench.py > $OUT/bench.json 2>$OUT/bench.err; python -c "$SUM" full < $OUT/bench.json >> $OUT/out.txt 2>&1 ;;
    ab:*) IFS=, read -ra E <<< "${step#ab:}"; for i in 1 2; do bench_once off A=1; bench_once "on ${step#ab:}" "${E[@]}"; done ;;
    test:*) timeout 1500 python -m pytest tests -m gpu -x -q -k "${step#test:}" 2>&1 | tail -15 >> $OUT/out.txt ;;
    tests) timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $OUT/out.txt; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 >> $OUT/out.txt ;;
    py:*) python ${step#py:} >> $OUT/out.txt 2>&1 ;;
    abvar:*) spec=${step#abvar:}; vars=${spec%%:*}; keys=${spec#*:}; [ "$keys" = "$spec" ] && keys=adam_pair
             VARIANTS="${vars//,/ }" KEYS="$keys" ROUNDS=${ROUNDS:-2} bash tools/ab_bench.sh >> $OUT/out.txt 2>&1 ;;
    sh:*) bash -c "${step#sh:}" >> $OUT/out.txt 2>&1 ;;
    *) echo "unknown step $step" >> $OUT/out.txt ;;
  esac
done
cat $OUT/out.txt
