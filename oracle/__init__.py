"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the dfm/george hot path -- kernel-matrix build,
dense Cholesky solver, HODLR solver and the GP-level formulas on top -- used
exclusively by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` as the *checker*.  Nothing under ``george_amd/`` imports it.

Parity status:
* dense path (kernels_np.py, solver_np.py): PINNED against the reference's own
  compiled C++ evaluator + reference Python (oracle/_ref, oracle/gen_golden.py)
  and the reference's published golden scalars (docs/tutorials/scaling.rst:76,91;
  first.rst:91,119,129).
* HODLR path (hodlr_np.py): PINNED against the reference's own `include/george/hodlr.h`, compiled unmodified against
  oracle/mini_eigen (the Eigen submodule is empty upstream) behind oracle/hodlr_ref_driver.cpp (oracle/_ref/_hodlr;
  goldens: oracle/gen_golden_hodlr.py, oracle/gen_golden_large.py C4 at N = 262144); the stand-in's LDLT / FullPivLU are
  checked against SciPy / NumPy (tests/test_oracle_hodlr.py).
* the reference's Python package itself is imported only where /root/reference exists (ref_loader.load_reference());
  nothing of it travels to the GPU box.
"""
