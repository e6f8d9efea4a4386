"""Flatten a kernel *spec* (the Python object tree users build with ``+``/``*``)
into the postfix POD array consumed by ``gh_kernel_create``.

This is the MI355X-side replacement of ``george::parse_kernel_spec``
(reference ``include/george/parser.h:14-35`` for operators, ``:344-403`` for a
stationary leaf, ``:325-343`` for a non-stationary one): it reads exactly the
same attributes -- ``is_kernel, operator_type, k1, k2, kernel_type,
metric.metric_type / ndim / axes / get_parameter_vector(True), blocked,
min_block, max_block, ndim, axes`` and the named parameters/constants -- so it
accepts the reference's own ``george.kernels`` objects as well as ours.
Errors mirror the parser's: ``ValueError("invalid kernel" / "unrecognized
kernel" / "unrecognized operator" / "unrecognized metric")`` and
``RuntimeError("dimension mismatch")``.
"""
import numpy as np

from . import _native as N

# own-parameter names per kernel_type, in parameter_names order (kernels/*.yml `params`)
PARAMS = {
    0: ("log_gamma2",), 1: ("log_alpha",), 2: (), 3: ("location", "log_width"), 4: (),
    5: ("log_period",), 6: (), 7: ("gamma", "log_period"), 8: ("log_constant",), 9: (),
    10: (), 11: ("log_sigma2",), 12: (),
}
CONSTANTS = {0: "order", 11: "order"}
STATIONARY = (1, 2, 6, 9, 10)


def _leaf(spec):
    node = N.gh_knode()
    node.op = N.GH_OP_LEAF
    kt = int(spec.kernel_type)
    if kt not in PARAMS:
        raise ValueError("unrecognized kernel")
    node.kernel_type = kt
    names = PARAMS[kt]
    node.n_params = len(names)
    for i, name in enumerate(names):
        node.params[i] = float(getattr(spec, name))
    if kt in CONSTANTS:
        node.constant = float(getattr(spec, CONSTANTS[kt]))
    if kt in STATIONARY:
        metric = spec.metric
        mtype = int(metric.metric_type)
        if mtype not in (0, 1, 2):
            raise ValueError("unrecognized metric")
        node.metric_type = mtype
        node.ndim = int(metric.ndim)
        axes = [int(a) for a in np.atleast_1d(metric.axes)]
        vec = np.asarray(metric.get_parameter_vector(True), dtype=np.float64)
        if len(vec) > N.GH_MAX_METRIC:
            raise ValueError("metric has too many parameters for the HIP evaluator")
        node.n_metric = len(vec)
        for i, v in enumerate(vec):
            node.metric[i] = float(v)
        node.blocked = 1 if bool(spec.blocked) else 0
        if node.blocked:
            lo = np.atleast_1d(np.asarray(spec.min_block, dtype=np.float64))
            hi = np.atleast_1d(np.asarray(spec.max_block, dtype=np.float64))
            for i in range(len(axes)):
                node.min_block[i] = float(lo[i])
                node.max_block[i] = float(hi[i])
    else:
        node.metric_type = -1
        node.ndim = int(spec.ndim)
        axes = [int(a) for a in np.atleast_1d(spec.axes)]
    if len(axes) > N.GH_MAX_AXES:
        raise ValueError("too many active axes for the HIP evaluator (max %d)" % N.GH_MAX_AXES)
    node.naxes = len(axes)
    for i, a in enumerate(axes):
        node.axes[i] = a
    return node


def _walk(spec, out):
    if not hasattr(spec, "is_kernel"):
        raise ValueError("invalid kernel")                      # parser.h:16
    if not bool(spec.is_kernel):
        _walk(spec.k1, out)
        _walk(spec.k2, out)
        node = N.gh_knode()
        op = int(spec.operator_type)
        if op == 0:
            node.op = N.GH_OP_SUM
        elif op == 1:
            node.op = N.GH_OP_PRODUCT
        else:
            raise ValueError("unrecognized operator")           # parser.h:33
        out.append(node)
        return
    out.append(_leaf(spec))


def flatten(spec):
    """Return a ctypes array of ``gh_knode`` in postfix order."""
    nodes = []
    _walk(spec, nodes)
    if len(nodes) > N.GH_MAX_NODES:
        raise ValueError("kernel expression too large for the HIP evaluator")
    arr = (N.gh_knode * len(nodes))(*nodes)
    return arr


class DeviceKernel(object):
    """Owning wrapper of a ``gh_kernel*`` built from a spec."""

    def __init__(self, spec):
        arr = flatten(spec)
        handle = N._vp()
        N.check(N.lib.gh_kernel_create(arr, len(arr), N.C.byref(handle)))
        self.handle = handle
        self.ndim = N.lib.gh_kernel_ndim(handle)
        self.size = N.lib.gh_kernel_size(handle)

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                N.lib.gh_kernel_destroy(h)
            except Exception:
                pass
            self.handle = None
