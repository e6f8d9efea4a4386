"""Dense-solver time against the number of other streams alive in the process (HIP maps streams onto a few
hardware queues; which queue the handle's streams land on changes the step time by 20 %)."""
import sys
sys.path.insert(0, '.')
import torch, bench
def run(n, steps=4, warm=2):
    j = bench.DenseJob(n, 0, 0, profile=False)
    e, ll = bench.run_timed(j, steps, warm, lambda: None)
    j.close()
    return e / steps * 1e3
print("fresh:   N=8192 %.2f ms  N=16384 %.2f ms" % (run(8192), run(16384)), flush=True)
other = bench.DenseJob(4096, 0, 0, profile=False); other.step()
print("a second dense handle alive:   N=8192 %.2f ms  N=16384 %.2f ms" % (run(8192), run(16384)), flush=True)
hj = bench.HodlrJob(65536, 0); hj.step()
print("+ a HODLR handle alive:        N=8192 %.2f ms  N=16384 %.2f ms" % (run(8192), run(16384)), flush=True)
hj.close(); other.close()
keep = []
for k in range(0, 5):
    print("other streams alive: %d   N=8192: %.2f ms   N=16384: %.2f ms" % (len(keep), run(8192), run(16384)), flush=True)
    q = torch.cuda.Stream()
    if len(sys.argv) < 2 or sys.argv[1] != "unused":
        with torch.cuda.stream(q):
            torch.zeros(16, device="cuda").sum().item()
    keep.append(q)
