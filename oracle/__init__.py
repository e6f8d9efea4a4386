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
* HODLR path (hodlr_np.py): the reference extension is not buildable here
  (Eigen submodule absent) -> pinned only through the reference's own HODLR
  tests' criterion (agreement with the dense answer within allclose) and the
  N=100 golden log-likelihood; the RNG-dependent pivot sequence is unpinned.
"""
