"""cProfile of the host side of the train step (where do the ~4.5 ms of Python per step go?)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
tr = bench.build_trainer(bench.WORKLOADS["distill_4096x128"], 0, 1)
tr.pipeline_steps = True
for i in range(8):
    tr.train_iteration(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.train_iteration(8 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOPN", "28")))
