#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s21; mkdir -p $O
VARIANTS="A B C" ROUNDS=3 STEPS=60 KEYS="fwd_mean data_rows bwd_data/2304x256" bash tools/ab_bench.sh > $O/ab.txt 2>&1
cat $O/ab.txt
