"""compute()+log_likelihood() vs N on one MI355X through the C ABI (device-resident inputs):
python scripts/size_sweep.py [N ...]  ->  markdown table on stdout."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536]
    print("| N | ms per compute()+log_likelihood() | effective TFLOP/s | of 78.6 |\n|---|---|---|---|")
    for n in sizes:
        job = bench.DenseJob(n, 0, 0, profile=False)
        steps = 10 if n <= 16384 else 3
        el, ll = bench.run_timed(job, steps, 2, lambda: None)
        job.close()
        sec = el / steps
        tf = bench.flops_alg(n) / sec * 1e-12
        print("| %d | %.3f | %.2f | %.1f %% |" % (n, sec * 1e3, tf, 100 * tf / bench.PEAK_FP64_MFMA_TFLOPS), flush=True)


if __name__ == "__main__":
    main()
