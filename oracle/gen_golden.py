"""TEST INFRASTRUCTURE ONLY.  Generates the committed golden fixtures under tests/golden/ by
running the REAL reference (``/root/reference`` Python + its compiled C++ kernel evaluator,
oracle/ref_loader.py) in the build container.  Run:  python -m oracle.gen_golden
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_loader  # noqa: E402
import zoo  # noqa: E402


def main():
    george = ref_loader.load_reference()
    if george is None:
        raise SystemExit("reference not available (need /root/reference and oracle/_ref built: make -C oracle)")
    K = george.kernels
    out = {}
    rng = np.random.RandomState(123)
    for name, k in zoo.kernel_zoo(K):
        t1 = rng.randn(20, k.ndim)
        t2 = rng.randn(7, k.ndim)
        ki = k.kernel
        which = np.ones(k.full_size, dtype=np.uint32)
        out[name + "/t1"] = t1
        out[name + "/t2"] = t2
        out[name + "/vsym"] = ki.value_symmetric(t1)
        out[name + "/vgen"] = ki.value_general(t1, t2)
        out[name + "/vdiag"] = ki.value_diagonal(t1, t1[::-1].copy())
        out[name + "/ggen"] = ki.gradient_general(which, t1, t2)
        out[name + "/gsym"] = ki.gradient_symmetric(which, t1)
        out[name + "/x1"] = ki.x1_gradient_general(t1, t2)
        out[name + "/x2"] = ki.x2_gradient_general(t1, t2)
        out[name + "/names"] = np.array(k.get_parameter_names(include_frozen=True))
        out[name + "/vector"] = k.get_parameter_vector(include_frozen=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "kernels.npz"), **out)

    gp_out = {}
    rng = np.random.RandomState(42)
    for name, (kernel, x, yerr, y) in zoo.gp_configs(K).items():
        gp = george.GP(kernel)
        gp.compute(x, yerr)
        gp_out[name + "/loglike"] = np.array(gp.log_likelihood(y))
        gp_out[name + "/logdet"] = np.array(gp.solver.log_determinant)
        t = rng.uniform(x.min(), x.max(), (64,) + x.shape[1:])
        mu, var = gp.predict(y, t, return_var=True)
        gp_out[name + "/t"] = t
        gp_out[name + "/mu"] = mu
        gp_out[name + "/var"] = var
        _, cov = gp.predict(y, t[:16])
        gp_out[name + "/cov16"] = cov
        gp_out[name + "/grad"] = gp.grad_log_likelihood(y)
        gp_out[name + "/alpha"] = gp.apply_inverse(y)
        Y5 = rng.randn(len(x), 5)
        gp_out[name + "/Y5"] = Y5
        gp_out[name + "/alpha5"] = gp.apply_inverse(Y5)
        r = rng.randn(3, len(x))
        gp_out[name + "/r3"] = r
        gp_out[name + "/sqrt3"] = gp.solver.apply_sqrt(r)
    # white-noise + mean gradient case (tests/test_gp.py:16-56 shape): kernel + fitted white noise and mean
    kernel, x, yerr, y = zoo.gp_configs(K)["C5small"]
    gp = george.GP(kernel, mean=0.3, fit_mean=True, white_noise=np.log(0.05), fit_white_noise=True)
    gp.compute(x, yerr)
    gp_out["C5wn/loglike"] = np.array(gp.log_likelihood(y))
    gp_out["C5wn/grad"] = gp.grad_log_likelihood(y)
    gp_out["C5wn/names"] = np.array(gp.get_parameter_names())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gp.npz"), **gp_out)
    print("wrote", len(out), "kernel arrays and", len(gp_out), "gp arrays")
    print("scaling100 loglike", float(gp_out["scaling100/loglike"]), "(published: 133.946394912, scaling.rst:76)")


if __name__ == "__main__":
    main()
