#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s16; mkdir -p $O
timeout 900 python tools/trajectory_spread.py 4 > $O/spread.txt 2>&1
echo "--- SNF_FUSED_SH_INPUT=0" >> $O/spread.txt
SNF_FUSED_SH_INPUT=0 timeout 900 python tools/trajectory_spread.py 4 >> $O/spread.txt 2>&1
cat $O/spread.txt
