#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s24; mkdir -p $O; : > $O/out.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> $O/out.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -3 >> $O/out.txt
python bench.py 2>/dev/null > $O/bench_default.json; python tools/benchsum.py $O/bench_default.json 2>&1 | grep -E "ms/step|other workload|step frac" >> $O/out.txt
cat $O/out.txt
