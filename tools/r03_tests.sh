#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r03t}; mkdir -p $out
SNF_PARITY_VERBOSE=1 python -m pytest tests -m gpu -q -s > $out/tests.log 2>&1; echo "tests rc=$?"; tail -8 $out/tests.log | cut -c1-300
grep -o "\[grad_parity\] L1 / max / outliers / slices with one / worst slice L1: {.*}" $out/tests.log > $out/parity_reports.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-400
