#!/bin/bash
# round-4 GPU session 1: suite, today's baseline, first A/Bs, timeline, counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s1; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/tests.txt 2>&1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
python tools/benchsum.py $O/bench.json > $O/benchsum.txt 2>&1
timeout 600 bash tools/ab_env.sh SNF_HG_PAIR_XCD=1 snf_hashgrid_bwd_presorted_adam_pair/F8L12+12 > $O/ab_xcd.txt 2>&1
timeout 600 bash tools/ab_env.sh SNF_HG_PAIR_XCD=2 snf_hashgrid_bwd_presorted_adam_pair/F8L12+12 > $O/ab_xcd2.txt 2>&1
ROUNDS=2 STEPS=40 timeout 900 bash tools/ab_bench.sh > $O/ab_prefetch.txt 2>&1
timeout 300 python tools/eager_timeline.py 2>/dev/null | cut -c1-200 > $O/timeline.txt
( python tools/bench_render.py; SNF_RENDER_REUSE_PASS1=0 python tools/bench_render.py ) 2>/dev/null | grep "^render" | cut -c1-300 > $O/render.txt
timeout 900 bash tools/step_counters.sh r04s1/counters > /dev/null 2>&1
cat $O/tests.txt $O/benchsum.txt $O/ab_xcd.txt $O/ab_xcd2.txt $O/ab_prefetch.txt $O/render.txt
