"""The ``GP`` facade: owns kernel / mean / white-noise models and a solver class.

Same public surface and semantics as the reference's ``src/george/gp.py``
(``GP`` :22-635) -- ``compute, recompute, log_likelihood, grad_log_likelihood,
nll, grad_nll, predict, apply_inverse, sample, sample_conditional, get_matrix``
plus the deprecated ``lnlikelihood`` aliases -- so user code ports by changing
the import.  Differences are behind the API:

* the default solver is the HIP :class:`george_amd.BasicSolver`;
* ``predict`` and ``grad_log_likelihood`` use the solver's fused device-resident
  entry points when it has them (the N x M cross-covariance, K^-1 and the
  (N, N, P) gradient tensor of gp.py:532-541 / :436-466 never reach the host);
  any other duck-typed solver takes the generic NumPy path below, formula for
  formula as the reference.
"""
import warnings

import numpy as np

from . import kernels
from .modeling import ModelSet, ConstantModel
from .solvers import TrivialSolver, BasicSolver
from .utils import multivariate_gaussian_samples

__all__ = ["GP"]

TINY = 1.25e-12          # default white-noise variance (gp.py:19)


def _as_model(obj):
    try:
        return ConstantModel(float(obj))
    except TypeError:
        return obj


def _is_number(obj):
    try:
        float(obj)
    except TypeError:
        return False
    return True


class GP(ModelSet):

    def __init__(self, kernel=None, fit_kernel=True, mean=None, fit_mean=None,
                 white_noise=None, fit_white_noise=None, solver=None, **kwargs):
        self._computed = False
        self._alpha = None
        self._y = None
        super(GP, self).__init__([
            ("mean", ConstantModel(0.0) if mean is None else _as_model(mean)),
            ("white_noise", ConstantModel(np.log(TINY)) if white_noise is None else _as_model(white_noise)),
            ("kernel", kernels.EmptyKernel() if kernel is None else kernel),
        ])
        # a plain number for mean / white_noise is held fixed unless asked otherwise (gp.py:96-122)
        if _is_number(mean) and fit_mean is None:
            fit_mean = False
        if _is_number(white_noise) and fit_white_noise is None:
            fit_white_noise = False
        if not fit_kernel:
            self.models["kernel"].freeze_all_parameters()
        if mean is None or (fit_mean is not None and not fit_mean):
            self.models["mean"].freeze_all_parameters()
        if white_noise is None or (fit_white_noise is not None and not fit_white_noise):
            self.models["white_noise"].freeze_all_parameters()

        if solver is None:
            empty = kernel is None or kernel.kernel_type == kernels.EmptyKernel.kernel_type
            solver = TrivialSolver if empty else BasicSolver
        self.solver_type = solver
        self.solver_kwargs = kwargs
        self.solver = None

    # -- sub-model helpers ----------------------------------------------------
    @property
    def mean(self):
        return self.models["mean"]

    @property
    def white_noise(self):
        return self.models["white_noise"]

    @staticmethod
    def _model_arg(x):
        return x[:, 0] if (x.ndim == 2 and x.shape[1] == 1) else x

    def _call_mean(self, x):
        mu = self.mean.get_value(self._model_arg(x)).flatten()
        if not np.all(np.isfinite(mu)):
            raise ValueError("mean function returned NaN or Inf for parameters:\n{0}".format(
                self.mean.get_parameter_dict(include_frozen=True)))
        return mu

    def _call_mean_gradient(self, x):
        g = self.mean.get_gradient(self._model_arg(x))
        if np.any(np.isnan(g)) or np.any(np.isinf(g)):
            raise ValueError("mean gradient function returned NaN or Inf for parameters:\n{0}".format(
                self.mean.get_parameter_dict(include_frozen=True)))
        return g

    def _call_white_noise(self, x):
        return self.white_noise.get_value(self._model_arg(x)).flatten()

    def _call_white_noise_gradient(self, x):
        return self.white_noise.get_gradient(self._model_arg(x))

    # -- state ------------------------------------------------------------------
    @property
    def computed(self):
        return (self._computed and self.solver.computed
                and (self.kernel is None or not self.kernel.dirty))

    @computed.setter
    def computed(self, v):
        self._computed = v
        if v and self.kernel is not None:
            self.kernel.dirty = False

    def parse_samples(self, t):
        t = np.atleast_1d(t)
        if t.ndim == 1:
            t = np.atleast_2d(t).T
        if t.ndim != 2 or (self.kernel is not None and t.shape[1] != self.kernel.ndim):
            raise ValueError("Dimension mismatch")
        return t

    def _check_dimensions(self, y, check_dim=True):
        y = np.atleast_1d(y)
        if check_dim and y.ndim > 1:
            raise ValueError("The predicted dimension must be 1-D")
        if len(y) != self._x.shape[0]:
            raise ValueError("Dimension mismatch")
        return y

    def _residual(self, y):
        return np.ascontiguousarray(self._check_dimensions(y) - self._call_mean(self._x), dtype=np.float64)

    def _residual_quiet(self, y, quiet):
        """Residual for the likelihood entry points: a wrongly shaped ``y`` ALWAYS raises; only a failing
        mean function is silenced by ``quiet`` (gp.py:383-392).  Returns None when silenced."""
        yv = self._check_dimensions(y)
        try:
            mu = self._call_mean(self._x)
        except ValueError:
            if quiet:
                return None
            raise
        return np.ascontiguousarray(yv - mu, dtype=np.float64)

    def _compute_alpha(self, y, cache):
        if not cache:
            return self.solver.apply_inverse(self._residual(y), in_place=True).flatten()
        if self._alpha is None or not np.array_equiv(y, self._y):
            self._y = y
            self._alpha = self.solver.apply_inverse(self._residual(y), in_place=True).flatten()
        return self._alpha

    # -- the hot path -------------------------------------------------------------
    def compute(self, x, yerr=0.0, **kwargs):
        """Build and factorise K(x, x) + diag(yerr^2 + exp(white_noise))  (gp.py:303-337)."""
        self._obj_cache = None           # a gradient cached by the fused objective belongs to the OLD (x, yerr)
        self._x = np.ascontiguousarray(self.parse_samples(x), dtype=np.float64)
        # scalar yerr broadcasts over the points, anything else must have one entry per point
        if np.ndim(yerr) == 0 or (np.size(yerr) == 1 and len(self._x) != 1):
            sig = np.full(len(self._x), float(np.reshape(yerr, ())))
        else:
            sig = self._check_dimensions(yerr)
        self._yerr2 = np.ascontiguousarray(sig ** 2, dtype=np.float64)

        self.solver = self.solver_type(self.kernel, **(self.solver_kwargs))
        sigma = np.sqrt(self._yerr2 + np.exp(self._call_white_noise(self._x)))
        self.solver.compute(self._x, sigma, **kwargs)

        self._const = -0.5 * (len(self._x) * np.log(2 * np.pi) + self.solver.log_determinant)
        self.computed = True
        self._alpha = None

    def recompute(self, quiet=False, **kwargs):
        if not self.computed:
            if not (hasattr(self, "_x") and hasattr(self, "_yerr2")):
                raise RuntimeError("You need to compute the model first")
            try:
                self.compute(self._x, np.sqrt(self._yerr2), **kwargs)
            except (ValueError, np.linalg.LinAlgError):
                if quiet:
                    return False
                raise
        return True

    def log_likelihood(self, y, quiet=False):
        """-1/2 r^T K^-1 r - 1/2 log|K| - N/2 log 2 pi   (gp.py:369-397)."""
        if not self.recompute(quiet=quiet):
            return -np.inf
        r = self._residual_quiet(y, quiet)
        if r is None:
            return -np.inf
        ll = self._const - 0.5 * self.solver.dot_solve(r)
        return ll if np.isfinite(ll) else -np.inf

    def grad_log_likelihood(self, y, quiet=False):
        """Gradient wrt the unfrozen parameters, ordered mean | white_noise | kernel (gp.py:406-468)."""
        if not self.recompute(quiet=quiet):
            return np.zeros(len(self), dtype=np.float64)
        r = self._residual_quiet(y, quiet)
        if r is None:
            return np.zeros(len(self), dtype=np.float64)

        n_wn, n_k = len(self.white_noise), len(self.kernel)
        fused = callable(getattr(self.solver, "grad", None))
        kgrad = diagA = None
        if fused and (n_wn or n_k):
            # alpha, diag(A) and 1/2 sum A_ij dK_ij/dtheta in one device-resident call
            which = self.kernel.unfrozen_mask.astype(np.uint32)
            kg_full, alpha, diagA = self.solver.grad(r, which)
            kgrad = kg_full[self.kernel.unfrozen_mask]
        else:
            alpha = self.solver.apply_inverse(r, in_place=True).flatten()
            if n_wn or n_k:
                A = np.einsum("i,j", alpha, alpha) - self.solver.get_inverse()
                diagA = np.diag(A)
                if n_k:
                    kgrad = 0.5 * np.einsum("ijk,ij", self.kernel.get_gradient(self._x), A)
        return self._assemble_grad(alpha, diagA, kgrad, quiet)

    def _assemble_grad(self, alpha, diagA, kgrad, quiet):
        """mean | white_noise | kernel blocks of the gradient from alpha, diag(alpha alpha^T - K^-1) and
        the kernel block (gp.py:443-466)."""
        n_wn, n_k = len(self.white_noise), len(self.kernel)
        grad = np.empty(len(self))
        at = 0
        n_m = len(self.mean)
        if n_m:
            try:
                mg = self._call_mean_gradient(self._x)
            except ValueError:
                if quiet:
                    return np.zeros(len(self), dtype=np.float64)
                raise
            grad[at:at + n_m] = np.dot(mg, alpha)
            at += n_m
        if n_wn:
            wn = self._call_white_noise(self._x)
            wng = self._call_white_noise_gradient(self._x)
            grad[at:at + n_wn] = 0.5 * np.sum((np.exp(wn) * diagA)[None, :] * wng, axis=1)
            at += n_wn
        if n_k:
            grad[at:at + n_k] = kgrad
        return grad

    # -- the optimiser objective (gp.py:470-480; docs/tutorials/hyper.rst:131-152) ----------------
    # ``minimize(gp.nll, p0, jac=gp.grad_nll, args=(y,))`` asks for the value and then the gradient at
    # every iterate.  With a solver that offers ``objective`` (the HIP BasicSolver) each iterate is ONE
    # fused device call -- build, factor, one forward solve shared by r^T K^-1 r and alpha, K^-1,
    # gradient reduction, one synchronisation -- instead of compute + dot_solve + grad with the matrix
    # factored twice (the reference marks the model dirty again when grad_nll re-sets the same vector):
    # once a gradient has been asked for, ``nll`` computes it eagerly and ``grad_nll`` at the same
    # (vector, y) returns it.  ``nll_and_grad`` is the explicit one-call form (``jac=True``).
    def set_parameter_vector(self, vector, include_frozen=False):
        if self._computed and self.solver is not None and not self.kernel.dirty:
            v = np.asarray(vector, dtype=np.float64)
            cur = self.get_parameter_vector(include_frozen=include_frozen)
            if v.shape == cur.shape and np.array_equal(v, cur):
                return                                  # same point: keep the factorisation
        super(GP, self).set_parameter_vector(vector, include_frozen=include_frozen)

    def _fused_capable(self):
        return (callable(getattr(self.solver_type, "objective", None))
                and hasattr(self, "_x") and hasattr(self, "_yerr2"))

    def _objective(self, y, want_grad, quiet):
        """(log-likelihood, gradient | None) at the current parameters through the solver's fused entry
        point; (-inf, zeros) where the reference's quiet mode returns them."""
        bad = (-np.inf, np.zeros(len(self)) if want_grad else None)
        r = self._residual_quiet(y, quiet)
        if r is None:
            return bad
        n_wn, n_k = len(self.white_noise), len(self.kernel)
        need_A = want_grad and (n_wn or n_k)
        self.solver = self.solver_type(self.kernel, **(self.solver_kwargs))
        sigma = np.sqrt(self._yerr2 + np.exp(self._call_white_noise(self._x)))
        which = self.kernel.unfrozen_mask.astype(np.uint32)
        try:
            if need_A:
                logdet, quad, kg_full, alpha, diagA = self.solver.objective(self._x, sigma, r, which, want_grad=True)
            else:
                logdet, quad, kg_full, alpha, diagA = self.solver.objective(self._x, sigma, r, which, want_grad=False)
        except (ValueError, np.linalg.LinAlgError):
            if quiet:
                return bad
            raise
        self._const = -0.5 * (len(self._x) * np.log(2 * np.pi) + logdet)
        self.computed = True
        self._alpha = None
        ll = self._const - 0.5 * quad
        ll = ll if np.isfinite(ll) else -np.inf
        if not want_grad:
            return ll, None
        if not need_A:                                   # only mean parameters vary: alpha is all that is needed
            alpha = self.solver.apply_inverse(r).flatten()
            return ll, self._assemble_grad(alpha, None, None, quiet)
        kgrad = kg_full[self.kernel.unfrozen_mask] if n_k else None
        return ll, self._assemble_grad(alpha, diagA, kgrad, quiet)

    def nll_and_grad(self, vector, y, quiet=True):
        """``(nll(vector, y), grad_nll(vector, y))`` from one fused device call."""
        self.set_parameter_vector(vector)
        if not np.isfinite(self.log_prior()):
            return np.inf, np.zeros(len(vector))
        if self.computed or not self._fused_capable():
            return -self.log_likelihood(y, quiet=quiet), -self.grad_log_likelihood(y, quiet=quiet)
        ll, g = self._objective(y, True, quiet)
        self._obj_cache = (np.array(vector, dtype=np.float64), np.array(y, dtype=np.float64), g)
        return -ll, -g

    def _cached_grad(self, vector, y):
        c = getattr(self, "_obj_cache", None)
        if c is None or not self.computed:
            return None
        v, yy = np.asarray(vector, dtype=np.float64), np.asarray(y, dtype=np.float64)
        if v.shape == c[0].shape and yy.shape == c[1].shape and np.array_equal(v, c[0]) and np.array_equal(yy, c[1]):
            return c[2]
        return None

    def nll(self, vector, y, quiet=True):
        self.set_parameter_vector(vector)
        if not np.isfinite(self.log_prior()):
            return np.inf
        if self.computed or not self._fused_capable():
            return -self.log_likelihood(y, quiet=quiet)
        # eager gradient only while the caller keeps asking for it: two value-only evaluations in a
        # row (a gradient-free optimiser, MCMC) switch it off again
        self._nll_since_grad = getattr(self, "_nll_since_grad", 0) + 1
        if self._nll_since_grad > 2:
            self._grad_seen = False
        want_grad = getattr(self, "_grad_seen", False)
        ll, g = self._objective(y, want_grad, quiet)
        self._obj_cache = (np.array(vector, dtype=np.float64), np.array(y, dtype=np.float64), g) if want_grad else None
        return -ll

    def grad_nll(self, vector, y, quiet=True):
        self._grad_seen = True
        self._nll_since_grad = 0
        self.set_parameter_vector(vector)
        if not np.isfinite(self.log_prior()):
            return np.zeros(len(vector))
        g = self._cached_grad(vector, y)
        if g is not None:
            return -g
        if self.computed or not self._fused_capable():
            return -self.grad_log_likelihood(y, quiet=quiet)
        ll, g = self._objective(y, True, quiet)
        self._obj_cache = (np.array(vector, dtype=np.float64), np.array(y, dtype=np.float64), g)
        return -g

    def predict(self, y, t, return_cov=True, return_var=False, cache=True, kernel=None):
        """Conditional mean and (co)variance at ``t``  (gp.py:482-545)."""
        self.recompute()
        xs = np.ascontiguousarray(self.parse_samples(t), dtype=np.float64)
        if kernel is None:
            kernel = self.kernel

        if callable(getattr(self.solver, "predict", None)):
            if cache:
                self._compute_alpha(y, True)       # keep the reference's alpha-cache semantics (gp.py:260-275)
            want_var = bool(return_var)
            want_cov = bool(return_cov) and not want_var
            mu, var, cov = self.solver.predict(kernel, self._residual(y), xs,
                                               return_var=want_var, return_cov=want_cov)
            mu = mu + self._call_mean(xs)
            if want_var:
                return mu, var
            if want_cov:
                return mu, cov
            return mu

        alpha = self._compute_alpha(y, cache)
        Kxs = kernel.get_value(xs, self._x)
        mu = np.dot(Kxs, alpha) + self._call_mean(xs)
        if not (return_var or return_cov):
            return mu
        KinvKxs = self.solver.apply_inverse(Kxs.T)
        if return_var:
            var = kernel.get_value(xs, diag=True)
            var -= np.sum(Kxs.T * KinvKxs, axis=0)
            return mu, var
        cov = kernel.get_value(xs)
        cov -= np.dot(Kxs, KinvKxs)
        return mu, cov

    def apply_inverse(self, y):
        """K^-1 (y - mean) for a vector or an (n, K) matrix  (gp.py:277-301)."""
        self.recompute(quiet=False)
        r = np.array(y, dtype=np.float64, order="F")
        r = self._check_dimensions(r, check_dim=False)
        r -= self._call_mean(self._x)[(slice(None),) + (np.newaxis,) * (r.ndim - 1)]
        b = self.solver.apply_inverse(r, in_place=True)
        return b.flatten() if r.ndim == 1 else b

    # -- sampling -------------------------------------------------------------------
    def sample_conditional(self, y, t, size=1):
        mu, cov = self.predict(y, t)
        return multivariate_gaussian_samples(cov, size, mean=mu)

    def sample(self, t=None, size=1):
        if t is None:
            self.recompute()
            n = self._x.shape[0]
            out = self.solver.apply_sqrt(np.random.randn(size, n))
            out += self._call_mean(self._x)
            return out[0] if size == 1 else out
        x = self.parse_samples(t)
        cov = self.get_matrix(x)
        cov[np.diag_indices_from(cov)] += TINY
        return multivariate_gaussian_samples(cov, size, mean=self._call_mean(x))

    def get_matrix(self, x1, x2=None):
        x1 = self.parse_samples(x1)
        if x2 is None:
            return self.kernel.get_value(x1)
        return self.kernel.get_value(x1, self.parse_samples(x2))

    # -- aliases ----------------------------------------------------------------------
    def lnlikelihood(self, y, quiet=False):
        warnings.warn("'lnlikelihood' is deprecated. Use 'log_likelihood'", DeprecationWarning)
        return self.log_likelihood(y, quiet=quiet)

    def grad_lnlikelihood(self, y, quiet=False):
        warnings.warn("'grad_lnlikelihood' is deprecated. Use 'grad_log_likelihood'", DeprecationWarning)
        return self.grad_log_likelihood(y, quiet=quiet)

    def get_value(self, *args, **kwargs):
        return self.log_likelihood(*args, **kwargs)

    def get_gradient(self, *args, **kwargs):
        return self.grad_log_likelihood(*args, **kwargs)
