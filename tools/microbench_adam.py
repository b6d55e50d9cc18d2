#!/usr/bin/env python3
"""Launch-shape sweep of the real fused Adam kernel (snf_adam_step) on arena-sized slices.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import samnerf_amd  # noqa
from samnerf_amd import ops, _lib

lib = _lib.load()
for n in (201_326_592, 25_165_824):
    bufs = [torch.zeros(n, device="cuda") for _ in range(4)]
    bufs[1].normal_()
    print(f"n = {n}")
    res = []
    for blocks in (256, 512, 768, 1024, 2048, 4096):
        for threads in (64, 128, 256):
            for U in (1, 2, 4):
                assert lib.snf_set_adam_launch(blocks, threads, U) == 0
                best = 1e9
                for rep in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.adam_step_(*bufs, 1e-3, 0.9, 0.999, 1e-15, rep + 1)
                    e1.record(); e1.synchronize()
                    best = min(best, e0.elapsed_time(e1))
                res.append((32.0 * n / best * 1e-6, blocks, threads, U, best))
    res.sort(reverse=True)
    for r in res[:12]:
        print("  %.0f GB/s  blocks=%d threads=%d U=%d  %.4f ms" % r)
    print("  ...worst: %.0f GB/s  blocks=%d threads=%d U=%d" % res[-1][:4])
    cur = [r for r in res if r[1:4] == (2048, 256, 2)][0]
    print("  current default (2048,256,2): %.0f GB/s" % cur[0])
