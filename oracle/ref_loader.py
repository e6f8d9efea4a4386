"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the *real* reference (dfm/george) for oracle pinning:

* ``load_kernel_interface()`` -- the reference's own C++ kernel evaluator
  (``/root/reference/src/george/kernel_interface.cpp``), compiled by
  ``oracle/Makefile`` into ``oracle/_ref/``.  The shared object travels to the
  GPU box, so this works there too (it only needs a *spec object* exposing the
  attributes ``parser.h:14-35,344-403`` reads -- our own host classes do).
* ``load_reference()`` -- the full reference Python package as ``george``, imported
  from the sources where they lie under ``/root/reference`` (the build container
  only), plus the compiled ``kernel_interface`` / ``_hodlr``.  ``None`` where
  ``/root/reference`` does not exist (the GPU box): a Python reference does not
  travel in any form -- rounds 4-5 staged its byte-code under ``oracle/_ref/george``,
  round 6 removed that.

* ``load_hodlr()`` -- the reference's HODLR solver: its unmodified
  ``include/george/hodlr.h`` compiled against ``oracle/mini_eigen`` behind
  ``oracle/hodlr_ref_driver.cpp`` (the reference's ``_hodlr.cpp`` needs the real
  Eigen, an empty un-vendored submodule).  Installed as ``george.solvers._hodlr``
  by ``load_reference()``, so the reference's own ``HODLRSolver`` Python class
  (and its tests) run on it.
"""
import glob
import importlib.util
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("GEORGE_REFERENCE", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src", "george")


def _so_path():
    hits = glob.glob(os.path.join(HERE, "_ref", "kernel_interface*.so"))
    return hits[0] if hits else None


def load_kernel_interface():
    """Return the reference ``KernelInterface`` class, or None if not built."""
    if "george.kernel_interface" in sys.modules:
        return sys.modules["george.kernel_interface"].KernelInterface
    if "_george_ref_kernel_interface" in sys.modules:
        return sys.modules["_george_ref_kernel_interface"].KernelInterface
    so = _so_path()
    if so is None:
        return None
    # PyInit_kernel_interface is looked up from the last dotted component.
    spec = importlib.util.spec_from_file_location("kernel_interface", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["_george_ref_kernel_interface"] = mod
    return mod.KernelInterface


def load_hodlr():
    """Return the ``HODLRSolver`` interface class of oracle/_ref/_hodlr (same methods as the
    reference's ``george.solvers._hodlr.HODLRSolver`` plus ``nodes()``), or None if not built."""
    if "_george_ref_hodlr" in sys.modules:
        return sys.modules["_george_ref_hodlr"].HODLRSolver
    hits = glob.glob(os.path.join(HERE, "_ref", "_hodlr*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("_hodlr", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["_george_ref_hodlr"] = mod
    return mod.HODLRSolver


def load_reference():
    """Import the reference package as ``george`` (build container only)."""
    if "george" in sys.modules and getattr(sys.modules["george"], "_is_oracle_ref", False):
        return sys.modules["george"]
    if os.path.isdir(REF_SRC):
        src, init = REF_SRC, os.path.join(REF_SRC, "__init__.py")
    else:
        return None
    KI = load_kernel_interface()
    if KI is None:
        return None

    spec = importlib.util.spec_from_file_location(
        "george", init, submodule_search_locations=[src])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["george"] = pkg

    # normally written by setuptools_scm (pyproject.toml:27-28)
    ver = types.ModuleType("george.george_version")
    ver.version = "0.4.0+oracle"
    sys.modules["george.george_version"] = ver

    ki = types.ModuleType("george.kernel_interface")
    ki.KernelInterface = KI
    sys.modules["george.kernel_interface"] = ki

    class _NoHODLR(object):
        def __init__(self, *a, **k):
            raise ImportError("oracle/_ref/_hodlr is not built (make -C oracle)")

    hod = types.ModuleType("george.solvers._hodlr")
    hod.HODLRSolver = load_hodlr() or _NoHODLR
    sys.modules["george.solvers._hodlr"] = hod

    spec.loader.exec_module(pkg)
    pkg._is_oracle_ref = True
    return pkg
