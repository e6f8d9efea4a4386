import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, numpy as np
job = bench.HodlrJob(262144, 0)
for i in range(3): job.step()
import torch
torch.cuda.synchronize()
t0=time.perf_counter()
for i in range(5): ll=job.step()
print("ms per step", (time.perf_counter()-t0)/5*1e3, ll)
