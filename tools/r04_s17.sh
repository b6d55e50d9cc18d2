#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s17; mkdir -p $O
VARIANTS="${VARIANTS:-A B C}" ROUNDS=${ROUNDS:-3} STEPS=60 KEYS="${KEYS:-2304x256 256r 192r}" bash tools/ab_bench.sh > $O/ab.txt 2>&1
cat $O/ab.txt
