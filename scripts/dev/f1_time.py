"""bench.f1_report alone: the hyper.rst 13-parameter model at N = 2048, 8192, 16384 (+ the CPU reference at 2048 unless 'nocpu')"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
r = bench.f1_report(0, cpu_n=0 if "nocpu" in sys.argv else 2048)
for k, v in r["sizes"].items():
    print(k, {q: round(v[q], 4) for q in ("compute_loglike_ms", "build_ms", "build_frac_of_hbm", "grad_ms", "fused_nll_and_grad_ms")}, "ll", v["log_likelihood"])
print(json.dumps(r.get("cpu_reference")))
