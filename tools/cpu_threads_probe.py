import sys, time, torch
sys.path.insert(0, '.')
import bench
cfg, B = bench.WORKLOADS["autopet128"]
for n in (16, 32, 64):
    torch.set_num_threads(n)
    t0 = time.time()
    r = bench.cpu_baseline(cfg, 2, budget_s=8.0)
    print(n, r["value"], r["sample"][:12], time.time() - t0, flush=True)
