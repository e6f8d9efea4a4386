"""Kernel *spec* classes (host side): what users combine with ``+`` and ``*``.

They carry names, parameters, metric, axes and ``block`` -- no arithmetic.  All
values and gradients are evaluated on the MI355X through
:class:`george_amd.kernel_interface.KernelInterface`.  The surface mirrors the
reference's generated ``src/george/kernels.py`` (base ``Kernel`` :29-200,
``Sum``/``Product`` :203-247, the 13 leaf classes :250-966; ``kernel_type`` ids
:273...:944) -- here the leaves are produced from one table instead of a Jinja
template (``templates/kernels.py`` + ``kernels/*.yml``).
"""
import numpy as np

from .modeling import Model, ModelSet
from .metrics import Metric, Subspace
from .kernel_interface import KernelInterface
from . import program

__all__ = ["Kernel", "Sum", "Product"]


class Kernel(ModelSet):
    """Abstract kernel spec."""

    is_kernel = True
    kernel_type = -1
    __array_priority__ = np.inf

    # ``np.float64(2.0) * kernel`` must build a kernel, not an object array
    def __array_wrap__(self, array, context=None, *unused):
        if context is None:
            raise TypeError("Invalid operation")
        ufunc, args, _ = context
        if ufunc.__name__ == "multiply":
            return float(args[0]) * args[1]
        if ufunc.__name__ == "add":
            return float(args[0]) + args[1]
        raise TypeError("Invalid operation")

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_ki_cache", None)                 # device handles are not picklable (kernels.py:52-55)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __getattr__(self, name):
        # named parameters of a leaf live on its base model (key None)
        if name.startswith("__") or name == "_ki_cache":
            raise AttributeError(name)
        models = self.__dict__.get("models")
        if models is not None:
            if name in models:
                return models[name]
            if None in models:
                return getattr(models[None], name)
        raise AttributeError(name)

    @property
    def kernel(self):
        """The device evaluator for the CURRENT parameters (rebuilt only when they change;
        the reference rebuilds its C++ tree on every access, kernels.py:67-69)."""
        key = bytes(program.flatten(self))
        cached = self.__dict__.get("_ki_cache")
        if cached is None or cached[0] != key:
            cached = (key, KernelInterface(self))
            self.__dict__["_ki_cache"] = cached
        return cached[1]

    # ---- arithmetic (kernels.py:83-100): scalars become ConstantKernel(log(b / ndim))
    def _constant(self, b):
        return ConstantKernel(log_constant=np.log(float(b) / self.ndim), ndim=self.ndim)

    def __add__(self, b):
        if not hasattr(b, "is_kernel"):
            return Sum(self._constant(b), self)
        return Sum(self, b)

    def __radd__(self, b):
        return self.__add__(b)

    def __mul__(self, b):
        if not hasattr(b, "is_kernel"):
            return Product(self._constant(b), self)
        return Product(self, b)

    def __rmul__(self, b):
        return self.__mul__(b)

    # ---- evaluation (kernels.py:102-143)
    def get_value(self, x1, x2=None, diag=False):
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        if x2 is None:
            return self.kernel.value_diagonal(x1, x1) if diag else self.kernel.value_symmetric(x1)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        return self.kernel.value_diagonal(x1, x2) if diag else self.kernel.value_general(x1, x2)

    def get_gradient(self, x1, x2=None, include_frozen=False):
        mask = np.ones(self.full_size, dtype=bool) if include_frozen else self.unfrozen_mask
        which = mask.astype(np.uint32)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        if x2 is None:
            g = self.kernel.gradient_symmetric(which, x1)
        else:
            g = self.kernel.gradient_general(which, x1, np.ascontiguousarray(x2, dtype=np.float64))
        return g[:, :, mask]

    def get_x1_gradient(self, x1, x2=None):
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = x1 if x2 is None else np.ascontiguousarray(x2, dtype=np.float64)
        return self.kernel.x1_gradient_general(x1, x2)

    def get_x2_gradient(self, x1, x2=None):
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = x1 if x2 is None else np.ascontiguousarray(x2, dtype=np.float64)
        return self.kernel.x2_gradient_general(x1, x2)

    # ---- finite-difference self checks (kernels.py:145-200), used by the test-suite
    def test_gradient(self, x1, x2=None, eps=1.32e-6, **kwargs):
        theta = self.get_parameter_vector()
        g0 = self.get_gradient(x1, x2=x2)
        for i, t in enumerate(theta):
            theta[i] = t + eps
            self.set_parameter_vector(theta)
            kp = self.get_value(x1, x2=x2)
            theta[i] = t - eps
            self.set_parameter_vector(theta)
            km = self.get_value(x1, x2=x2)
            theta[i] = t
            self.set_parameter_vector(theta)
            assert np.allclose(g0[:, :, i], 0.5 * (kp - km) / eps, **kwargs), \
                "incorrect gradient for parameter '{0}' ({1})".format(self.get_parameter_names()[i], i)

    def _fd_x(self, which, x1, x2, eps, kwargs):
        kwargs["atol"] = kwargs.get("atol", 0.5 * eps)
        if which == 1:
            g0 = self.get_x1_gradient(x1, x2=x2)
        else:
            g0 = self.get_x2_gradient(x1, x2=x2)
        if x2 is None:
            x2 = np.array(x1)
        moving = x1 if which == 1 else x2
        for i in range(len(moving)):
            for k in range(self.ndim):
                moving[i, k] += eps
                kp = self.get_value(x1, x2=x2)
                moving[i, k] -= 2 * eps
                km = self.get_value(x1, x2=x2)
                moving[i, k] += eps
                fd = 0.5 * (kp - km) / eps
                if which == 1:
                    assert np.allclose(g0[i, :, k], fd[i], **kwargs)
                else:
                    assert np.allclose(g0[:, i, k], fd[:, i], **kwargs)

    def test_x1_gradient(self, x1, x2=None, eps=1.32e-6, **kwargs):
        self._fd_x(1, x1, x2, eps, kwargs)

    def test_x2_gradient(self, x1, x2=None, eps=1.32e-6, **kwargs):
        self._fd_x(2, x1, x2, eps, kwargs)


class _Operator(Kernel):
    is_kernel = False
    kernel_type = -1
    operator_type = -1
    symbol = "?"

    def __init__(self, k1, k2):
        if k1.ndim != k2.ndim:
            raise ValueError("Dimension mismatch")
        self.ndim = k1.ndim
        self._dirty = True
        super(_Operator, self).__init__([("k1", k1), ("k2", k2)])

    @property
    def k1(self):
        return self.models["k1"]

    @property
    def k2(self):
        return self.models["k2"]

    @property
    def dirty(self):
        return self._dirty or self.k1.dirty or self.k2.dirty

    @dirty.setter
    def dirty(self, v):
        self._dirty = v
        self.k1.dirty = False
        self.k2.dirty = False

    def __repr__(self):
        return "{0} {1} {2}".format(self.k1, self.symbol, self.k2)


class Sum(_Operator):
    operator_type = 0
    symbol = "+"


class Product(_Operator):
    operator_type = 1
    symbol = "*"


# ---------------------------------------------------------------------------
# leaves: name -> (kernel_type, stationary, parameter names, constant names)
# ids / parameter order: reference src/george/kernels.py (generated; glob order)
# ---------------------------------------------------------------------------
_LEAVES = [
    ("LinearKernel",            0, False, ("log_gamma2",),            ("order",)),
    ("RationalQuadraticKernel", 1, True,  ("log_alpha",),             ()),
    ("ExpKernel",               2, True,  (),                         ()),
    ("LocalGaussianKernel",     3, False, ("location", "log_width"),  ()),
    ("EmptyKernel",             4, False, (),                         ()),
    ("CosineKernel",            5, False, ("log_period",),            ()),
    ("Matern52Kernel",          6, True,  (),                         ()),
    ("ExpSine2Kernel",          7, False, ("gamma", "log_period"),    ()),
    ("ConstantKernel",          8, False, ("log_constant",),          ()),
    ("ExpSquaredKernel",        9, True,  (),                         ()),
    ("Matern32Kernel",         10, True,  (),                         ()),
    ("PolynomialKernel",       11, False, ("log_sigma2",),            ("order",)),
    ("DotProductKernel",       12, False, (),                         ()),
]


def _get_block(self):
    if not self.blocked:
        return None
    return list(zip(self.min_block, self.max_block))


def _set_block(self, block):
    naxes = len(self.axes)
    if block is None:
        self.blocked = False
        self.min_block = np.full(naxes, -np.inf)
        self.max_block = np.full(naxes, np.inf)
        return
    block = np.atleast_2d(block)
    if block.shape != (naxes, 2):
        raise ValueError("dimension mismatch in block specification")
    self.blocked = True
    self.min_block, self.max_block = np.array(block[:, 0]), np.array(block[:, 1])


def _make_leaf(name, ktype, stationary, params, constants):
    base = type("Base" + name, (Model,), {"parameter_names": tuple(params)})

    def __init__(self, *args, **kwargs):
        # positional order follows the reference signature: params, constants, [metric, ...]
        order = list(params) + list(constants)
        if stationary:
            order += ["metric", "metric_bounds", "lower", "block"]
        order += ["bounds", "ndim", "axes"]
        if len(args) > len(order):
            raise TypeError("{0}() takes at most {1} positional arguments".format(name, len(order)))
        for key, val in zip(order, args):
            if key in kwargs:
                raise TypeError("{0}() got multiple values for argument '{1}'".format(name, key))
            kwargs[key] = val
        unknown = set(kwargs) - set(order)
        if unknown:
            raise TypeError("{0}() got an unexpected keyword argument '{1}'".format(name, sorted(unknown)[0]))
        ndim, axes = kwargs.get("ndim", 1), kwargs.get("axes")
        for c in constants:
            if kwargs.get(c) is None:
                raise ValueError("missing required parameter '{0}'".format(c))
            setattr(self, c, kwargs[c])
        models = []
        if stationary:
            if kwargs.get("metric") is None:
                raise ValueError("missing required parameter 'metric'")
            metric = Metric(kwargs["metric"], bounds=kwargs.get("metric_bounds"), ndim=ndim,
                            axes=axes, lower=kwargs.get("lower", True))
            self.ndim, self.axes = metric.ndim, metric.axes
            self.block = kwargs.get("block")
        else:
            self.subspace = Subspace(ndim, axes=axes)
            self.ndim, self.axes = self.subspace.ndim, self.subspace.axes
        pk = dict((p, kwargs.get(p)) for p in params)
        if kwargs.get("bounds") is not None:
            pk["bounds"] = kwargs["bounds"]
        models.append((None, base(**pk)))
        if stationary:
            models.append(("metric", metric))
        ModelSet.__init__(self, models)
        self.dirty = True

    def __repr__(self):
        inner = self.models[None]
        parts = ["{0}={1}".format(k, getattr(inner, k)) for k in inner.parameter_names]
        if stationary:
            parts += ["metric={0}".format(repr(self.metric)), "block={0}".format(repr(self.block))]
        else:
            parts += ["ndim={0}".format(self.ndim), "axes={0}".format(repr(self.axes))]
        return "{0}({1})".format(name, ", ".join(parts))

    namespace = {
        "kernel_type": ktype, "stationary": stationary, "__init__": __init__, "__repr__": __repr__,
        "__doc__": "{0} spec (kernel_type {1}); formula: reference kernels/{2}.yml".format(
            name, ktype, name.replace("Kernel", "")),
    }
    if stationary:
        namespace["block"] = property(_get_block, _set_block)
    cls = type(name, (Kernel,), namespace)
    return base, cls


for _name, _kt, _st, _pa, _co in _LEAVES:
    _base, _cls = _make_leaf(_name, _kt, _st, _pa, _co)
    globals()["Base" + _name] = _base
    globals()[_name] = _cls
    __all__.append(_name)
del _name, _kt, _st, _pa, _co, _base, _cls
