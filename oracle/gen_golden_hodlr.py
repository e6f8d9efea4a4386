"""TEST INFRASTRUCTURE ONLY.  Golden fixtures for the HODLR path, produced by the reference's own
``hodlr.h`` (oracle/_ref/_hodlr: the unmodified header compiled against oracle/mini_eigen behind
oracle/hodlr_ref_driver.cpp) with the reference's own kernel tree.  Writes
``tests/golden/hodlr.npz``: per configuration the (level, start, size, rank) list of the tree in
construction order, the log-determinant, K^-1 y and the inputs' seeds.

    python -m oracle.gen_golden_hodlr
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
    H = ref_loader.load_hodlr()
    if george is None or H is None:
        raise SystemExit("need /root/reference and oracle/_ref built: make -C oracle")
    out = {}
    for name, (kernel, x, yerr, y, kw) in zoo.hodlr_configs(george.kernels).items():
        x2 = np.ascontiguousarray(x.reshape(len(x), -1))
        h = H()
        h.compute(kernel, x2, yerr, **kw)
        out[name + "/nodes"] = np.array(h.nodes(), dtype=np.int64).reshape(-1, 4)
        out[name + "/logdet"] = np.array(h.log_determinant)
        out[name + "/alpha"] = h.apply_inverse(y)[:, 0]
        out[name + "/dot"] = np.array(h.dot_solve(y))
        print(name, len(x), kw, "logdet %.12g" % h.log_determinant, "max rank", out[name + "/nodes"][:, 3].max(initial=0))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hodlr.npz"), **out)


if __name__ == "__main__":
    main()
