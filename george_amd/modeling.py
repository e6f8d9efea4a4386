"""Host-side parameter protocol (names, vectors, freezing, bounds, dirty flag).

Pure bookkeeping -- no arithmetic of the hot path lives here.  The public
surface mirrors the reference's modeling protocol
(``src/george/modeling.py``: ``Model`` :11-343, ``ModelSet`` :346-473,
``ConstantModel`` :476-491, ``CallableModel`` :494-507) so that kernels, mean
and white-noise models written for george keep working unchanged.
"""
from collections import OrderedDict

import numpy as np

__all__ = ["Model", "ModelSet", "ConstantModel", "CallableModel"]

_FD_STEP = 1.254e-5


def _normalise_bounds(names, bounds):
    if hasattr(bounds, "get"):
        out = [bounds.get(n, (None, None)) for n in names]
    else:
        out = list(bounds)
    if len(out) != len(names):
        raise ValueError("the number of bounds must equal the number of parameters")
    for b in out:
        if len(b) != 2:
            raise ValueError("the bounds for each parameter must have the format: '(min, max)'")
    return out


class Model(object):
    """A named, freezable parameter vector; parameters live as float attributes."""

    parameter_names = tuple()

    def __init__(self, *args, **kwargs):
        nfull = self.full_size
        self.unfrozen_mask = np.ones(nfull, dtype=bool)
        self.dirty = True
        self.parameter_bounds = _normalise_bounds(self.parameter_names, kwargs.pop("bounds", dict()))

        if args:
            if len(args) != nfull:
                raise ValueError("expected {0} arguments but got {1}".format(nfull, len(args)))
            if kwargs:
                raise ValueError("parameters must be fully specified by arguments or keyword arguments, not both")
            values = list(args)
        else:
            values = []
            for name in self.parameter_names:
                if kwargs.get(name) is None:
                    raise ValueError("missing parameter '{0}'".format(name))
                values.append(kwargs.pop(name))
            if kwargs:
                raise ValueError("unrecognized parameter(s) '{0}'".format(list(kwargs.keys())))
        self.parameter_vector = values

        if not np.isfinite(self.log_prior()):
            raise ValueError("non-finite log prior value")

    # -- evaluation hooks ---------------------------------------------------
    def get_value(self, *args, **kwargs):
        raise NotImplementedError("overloaded by subclasses")

    def compute_gradient(self, *args, **kwargs):
        """Forward finite differences; subclasses should override with analytic forms."""
        theta = self.get_parameter_vector()
        f0 = self.get_value(*args, **kwargs)
        out = np.empty((len(theta),) + np.shape(f0), dtype=np.float64)
        for i, t in enumerate(theta):
            theta[i] = t + _FD_STEP
            self.set_parameter_vector(theta)
            out[i] = (self.get_value(*args, **kwargs) - f0) / _FD_STEP
            theta[i] = t
            self.set_parameter_vector(theta)
        return out

    def get_gradient(self, *args, **kwargs):
        include_frozen = kwargs.pop("include_frozen", False)
        g = self.compute_gradient(*args, **kwargs)
        return g if include_frozen else g[self.unfrozen_mask]

    # -- sizes / indexing ---------------------------------------------------
    def __len__(self):
        return self.vector_size

    @property
    def full_size(self):
        return len(self.parameter_names)

    @property
    def vector_size(self):
        return self.unfrozen_mask.sum()

    def _resolve(self, key):
        try:
            return self.get_parameter_names()[int(key)]
        except (TypeError, ValueError):
            return key

    def __getitem__(self, key):
        return self.get_parameter(self._resolve(key))

    def __setitem__(self, key, value):
        return self.set_parameter(self._resolve(key), value)

    # -- the vector ---------------------------------------------------------
    @property
    def parameter_vector(self):
        return np.array([getattr(self, n) for n in self.parameter_names])

    @parameter_vector.setter
    def parameter_vector(self, values):
        if len(values) != self.full_size:
            raise ValueError("dimension mismatch")
        for n, v in zip(self.parameter_names, values):
            setattr(self, n, float(v))
        self.dirty = True

    def _select(self, seq, include_frozen):
        if include_frozen:
            return seq
        return [s for s, keep in zip(seq, self.unfrozen_mask) if keep]

    def get_parameter_names(self, include_frozen=False):
        names = self.parameter_names
        return names if include_frozen else tuple(self._select(names, False))

    def get_parameter_bounds(self, include_frozen=False):
        bounds = self.parameter_bounds
        return bounds if include_frozen else list(self._select(bounds, False))

    def get_parameter_vector(self, include_frozen=False):
        v = self.parameter_vector
        return v if include_frozen else v[self.unfrozen_mask]

    def get_parameter_dict(self, include_frozen=False):
        return OrderedDict(zip(self.get_parameter_names(include_frozen=include_frozen),
                               self.get_parameter_vector(include_frozen=include_frozen)))

    def set_parameter_vector(self, vector, include_frozen=False):
        v = self.parameter_vector
        if include_frozen:
            v[:] = vector
        else:
            v[self.unfrozen_mask] = vector
        self.parameter_vector = v
        self.dirty = True

    def check_parameter_vector(self, vector):
        saved, was_dirty = np.array(self.get_parameter_vector()), self.dirty
        self.set_parameter_vector(vector)
        ok = np.isfinite(self.log_prior())
        self.set_parameter_vector(saved)
        self.dirty = was_dirty
        return ok

    # -- by-name access -----------------------------------------------------
    def _index(self, name):
        return self.get_parameter_names(include_frozen=True).index(name)

    def freeze_parameter(self, name):
        self.unfrozen_mask[self._index(name)] = False

    def thaw_parameter(self, name):
        self.unfrozen_mask[self._index(name)] = True

    def freeze_all_parameters(self):
        self.unfrozen_mask[:] = False

    def thaw_all_parameters(self):
        self.unfrozen_mask[:] = True

    def get_parameter(self, name):
        return self.get_parameter_vector(include_frozen=True)[self._index(name)]

    def set_parameter(self, name, value):
        v = self.get_parameter_vector(include_frozen=True)
        v[self._index(name)] = value
        self.set_parameter_vector(v, include_frozen=True)

    def log_prior(self):
        """0 inside the box of ``parameter_bounds``, -inf outside."""
        for value, (lo, hi) in zip(self.parameter_vector, self.parameter_bounds):
            if (lo is not None and value < lo) or (hi is not None and value > hi):
                return -np.inf
        return 0.0

    @staticmethod
    def parameter_sort(f):
        def wrapped(self, *args, **kwargs):
            values = f(self, *args, **kwargs)
            ordered = [values[k] for k in self.get_parameter_names(include_frozen=True)]
            if len(ordered) and type(ordered[0]).__module__ == np.__name__:
                return np.vstack(ordered)
            return ordered
        return wrapped


class ModelSet(Model):
    """An ordered set of named sub-models presenting one concatenated vector;
    parameter names are prefixed ``"<model>:"`` (no prefix for the ``None`` key)."""

    def __init__(self, models):
        self.models = OrderedDict((name, model) for name, model in models)

    def __getattr__(self, name):
        if "models" in self.__dict__ and name in self.models:
            return self.models[name]
        raise AttributeError(name)

    def _each(self):
        return self.models.values()

    @property
    def dirty(self):
        return any(m.dirty for m in self._each())

    @dirty.setter
    def dirty(self, value):
        for m in self._each():
            m.dirty = value

    @property
    def full_size(self):
        return sum(m.full_size for m in self._each())

    @property
    def vector_size(self):
        return sum(m.vector_size for m in self._each())

    @property
    def unfrozen_mask(self):
        return np.concatenate([m.unfrozen_mask for m in self._each()])

    @property
    def parameter_vector(self):
        return np.concatenate([m.parameter_vector for m in self._each()])

    @parameter_vector.setter
    def parameter_vector(self, values):
        at = 0
        for m in self._each():
            m.parameter_vector = values[at:at + m.full_size]
            at += m.full_size

    @property
    def parameter_names(self):
        names = []
        for key, m in self.models.items():
            prefix = "" if key is None else "{0}:".format(key)
            names.extend(prefix + "{0}".format(n) for n in m.parameter_names)
        return tuple(names)

    @property
    def parameter_bounds(self):
        return [b for m in self._each() for b in m.parameter_bounds]

    def _dispatch(self, method, name, *args):
        head, _, tail = name.partition(":")
        if head in self.models:
            return getattr(self.models[head], method)(tail, *args)
        if None in self.models:
            return getattr(self.models[None], method)(name, *args)
        raise ValueError("unrecognized parameter '{0}'".format(name))

    def freeze_parameter(self, name):
        self._dispatch("freeze_parameter", name)

    def thaw_parameter(self, name):
        self._dispatch("thaw_parameter", name)

    def freeze_all_parameters(self):
        for m in self._each():
            m.freeze_all_parameters()

    def thaw_all_parameters(self):
        for m in self._each():
            m.thaw_all_parameters()

    def get_parameter(self, name):
        return self._dispatch("get_parameter", name)

    def set_parameter(self, name, value):
        self.dirty = True
        return self._dispatch("set_parameter", name, value)

    def log_prior(self):
        total = 0.0
        for m in self._each():
            total += m.log_prior()
            if not np.isfinite(total):
                return -np.inf
        return total


class ConstantModel(Model):
    """``value`` everywhere (used for the GP's default mean and log-white-noise)."""

    parameter_names = ("value",)

    def get_value(self, x):
        return self.value + np.zeros(len(x))

    def compute_gradient(self, x):
        return np.ones((1, len(x)))


class CallableModel(Model):
    """Wrap a parameter-free callable (and optionally its gradient)."""

    def __init__(self, function, gradient=None):
        self.function = function
        self.gradient = gradient
        super(CallableModel, self).__init__()

    def get_value(self, x):
        return self.function(x)

    def compute_gradient(self, x):
        if self.gradient is not None:
            return self.gradient(x)
        return super(CallableModel, self).compute_gradient(x)
