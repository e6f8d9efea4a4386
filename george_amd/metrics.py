"""Distance-metric *specs* for stationary kernels (host side only).

API-compatible with the reference's ``src/george/metrics.py`` (``Subspace``
:13-22, ``Metric`` :25-140): a scalar gives an isotropic metric
(``metric_type`` 0, parameter ``log_M_0_0``), a vector an axis-aligned one
(type 1, ``log_M_i_i``), a matrix a general one (type 2, packed Cholesky factor
with logged diagonal).  The arithmetic on these parameters happens on the
device (george_amd/csrc/gh_eval.h ``gh_metric``).
"""
import numpy as np

from .modeling import Model

__all__ = ["Metric", "Subspace"]


class Subspace(object):
    """The subset of input axes a kernel acts on."""

    def __init__(self, ndim, axes=None):
        self.ndim = int(ndim)
        self.axes = np.atleast_1d(np.arange(self.ndim) if axes is None else axes).astype(int)
        if np.any(self.axes >= self.ndim):
            raise ValueError("invalid axis for {0} dimensional metric".format(self.ndim))


def _packed_cholesky(matrix):
    """Row-major lower triangle of chol(matrix) with log() on the diagonal."""
    L = np.linalg.cholesky(matrix)
    L[np.diag_indices_from(L)] = np.log(np.diag(L))
    return L[np.tril_indices_from(L)]


class Metric(Model):

    def __init__(self, metric, bounds=None, ndim=None, axes=None, lower=True):
        if isinstance(metric, Metric):            # copy constructor
            self.metric_type = metric.metric_type
            self.parameter_names = metric.parameter_names
            self.unfrozen_mask = np.array(metric.unfrozen_mask)
            self.parameter_vector = metric.get_parameter_vector(include_frozen=True)
            self.parameter_bounds = list(metric.parameter_bounds)
            self.ndim, self.axes = metric.ndim, metric.axes
            return
        if ndim is None:
            raise ValueError("missing required parameter 'ndim'")
        sub = Subspace(ndim, axes=axes)
        self.ndim, self.axes = sub.ndim, sub.axes
        naxes = len(self.axes)

        try:
            scalar = float(metric)
        except TypeError:
            scalar = None

        if scalar is not None:
            self.metric_type = 0
            names, values = ["log_M_0_0"], [np.log(scalar)]
        else:
            arr = np.atleast_1d(metric)
            if arr.ndim == 1:
                self.metric_type = 1
                if len(arr) != naxes:
                    raise ValueError("dimension mismatch")
                if np.any(arr <= 0.0):
                    raise ValueError("invalid (negative) metric")
                names = ["log_M_{0}_{0}".format(i) for i in range(naxes)]
                values = list(np.log(arr))
            elif arr.ndim == 2:
                self.metric_type = 2
                if arr.shape[0] != arr.shape[1]:
                    raise ValueError("metric must be square")
                if len(arr) != naxes:
                    raise ValueError("dimension mismatch")
                values = list(_packed_cholesky(np.array(arr, dtype=np.float64)))
                # naming follows the reference (metrics.py:88-96): per row i, "log_L_i_i" then "L_i_j", j > i
                names = []
                for i in range(naxes):
                    names.append("log_L_{0}_{0}".format(i))
                    names.extend("L_{0}_{1}".format(i, j) for j in range(i + 1, naxes))
            else:
                raise ValueError("invalid metric dimensions")

        self.parameter_names = tuple(names)
        kwargs = dict(zip(names, values))
        if bounds is not None:
            kwargs["bounds"] = bounds
        super(Metric, self).__init__(**kwargs)

    def to_matrix(self):
        v = self.get_parameter_vector(include_frozen=True)
        n = len(self.axes)
        if self.metric_type == 0:
            return np.exp(v) * np.eye(n)
        if self.metric_type == 1:
            return np.diag(np.exp(v))
        L = np.zeros((n, n))
        L[np.tril_indices_from(L)] = v
        L[np.diag_indices_from(L)] = np.exp(np.diag(L))
        return np.dot(L, L.T)

    def __repr__(self):
        v = self.get_parameter_vector(include_frozen=True)
        if self.metric_type == 0:
            head = "{0}".format(float(np.exp(v[0])))
        elif self.metric_type == 1:
            head = repr(np.exp(v))
        else:
            head = repr(self.to_matrix().tolist())
        bounds = [(None if a is None else np.exp(a), None if b is None else np.exp(b))
                  for a, b in self.get_parameter_bounds(include_frozen=True)]
        return "Metric({0}, ndim={1}, axes={2}, bounds={3})".format(head, self.ndim, repr(self.axes), bounds)
