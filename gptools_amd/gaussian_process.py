"""GaussianProcess: the reference's GP API with the covariance build + Cholesky + LML + predict
hot path running on an MI355X through the C ABI of ``libgpt_hip.so``.

ref: gptools/gaussian_process.py:56-503 (class, add_data), :623-783 (optimize_hyperparameters),
:785-1034 (predict), :1332-1416 (update_hyperparameters), :1418-1522 (compute_K_L_alpha_ll),
:1535-1605 (compute_Kij), :2443-2486 (_OptimizeHyperparametersEval).

What runs where
  * GPU (include/gpt_hip.h): ``K(X, X', n, n')`` (fused, never tiled), ``K_tot`` assembly, blocked
    Cholesky, ``z = L^-1 y`` (as an extra row of the factorisation), log-determinant, ``alpha``,
    and predict's ``K*^T alpha``, ``K*^T L^-T`` and ``K** - v^T v``.
  * Host Python: argument checking, hyperparameter bookkeeping, the O(num_params) log-prior, the
    mean function, ``T`` / ``output_transform`` products and scipy.optimize driving the MAP loop.
There is no CPU fallback for the hot path: without the HIP library every computation raises.

State attributes follow the reference: ``X n y err_y T K noise_K L alpha ll ll_deriv
K_up_to_date``.  ``K``, ``noise_K``, ``L`` and ``alpha`` are fetched lazily from the device the
first time they are read after a fit, so the MAP loop moves only ``params`` and a scalar.
"""
import sys
import traceback
import warnings

import numpy as np
import scipy.linalg
import scipy.optimize

from . import _lib, _ingest
from . import replicas
from .error_handling import GPArgumentError, GPImpossibleParamsError
from .kernel import Kernel, ZeroKernel, DiagonalNoiseKernel, SumKernel, ProductKernel
from .utils import CombinedBounds

__all__ = ["GaussianProcess"]

_NATIVE_FIT = (_lib.KERNEL_SE, _lib.KERNEL_M52, _lib.KERNEL_RQ, _lib.KERNEL_MATERN)


def _uniform_to_normal(u):
    from scipy.stats import norm
    return norm.ppf(u)


# draw_sample: how the given variates are distributed -> what turns them into standard normals
_VARIATE_KINDS = {"standard normal": lambda u: u, "uniform": _uniform_to_normal}
# draw_sample(modify_sign=...): e = the three entries of every eigenvector at the chosen end, in index order
_SIGN_FEATURES = {
    "value": lambda e, where: e[0] if where == "left" else e[-1],
    "slope": lambda e, where: (e[1] - e[0]) if where == "left" else (e[-1] - e[-2]),
    "concavity": lambda e, where: e[2] - 2.0 * e[1] + e[0],
}


def _combine(a, b, c=None):
    out = CombinedBounds(a, b)
    return CombinedBounds(out, c) if c is not None else out


class GaussianProcess(object):
    """Gaussian process regression with derivative observations and predictions.

    Parameters are those of ref: gptools/gaussian_process.py:196-238: ``k``, ``noise_k``, ``X``,
    ``y``, ``err_y``, ``n``, ``T``, ``diag_factor``, ``mu``, ``use_hyper_deriv``, ``verbose``;
    ``device`` (extra) selects the GPU.

    ``partitioned`` (extra attribute, default False): in a ``torch.distributed`` job with one rank per GPU,
    ``update_hyperparameters`` evaluates the log-posterior with the factorisation partitioned over the ranks
    (``gptools_amd.dist.DistributedLML``: block-cyclic block columns, panels over RCCL) instead of on this rank's GPU
    alone -- for one native kernel without ``T`` and without hyperparameter derivatives.  Every rank must then make
    the same calls with the same hyperparameters (an optimiser run identically on every rank does: all ranks see the
    same ``ll``).  ``L``, ``alpha``, ``predict`` ... need the factor on this GPU and refit locally on first use.
    """

    def __init__(self, k, noise_k=None, X=None, y=None, err_y=0, n=0, T=None, diag_factor=1e2, mu=None,
                 use_hyper_deriv=False, verbose=False, device=0, eager_alpha=False):
        if not isinstance(k, Kernel):
            raise TypeError("Argument k must be an instance of Kernel when constructing GaussianProcess!")
        if noise_k is None:
            noise_k = ZeroKernel(k.num_dim)
        elif not isinstance(noise_k, Kernel):
            raise TypeError("Keyword noise_k must be an instance of Kernel when constructing GaussianProcess!")
        self.mu = mu
        self.diag_factor = diag_factor
        self.k = k
        self.noise_k = noise_k
        self.use_hyper_deriv = use_hyper_deriv
        self.verbose = verbose
        self.device = device
        if eager_alpha:
            self.eager_alpha = True
        self.y = np.array([], dtype=float)
        self.X = None
        self.err_y = np.array([], dtype=float)
        self.n = None
        self.T = None
        self.ll = None
        self.ll_deriv = None
        self._reset_device_state()
        if X is not None:
            if y is None:
                raise GPArgumentError("Must pass both X and y when constructing GaussianProcess!")
            self.add_data(X, y, err_y=err_y, n=n, T=T)
        elif y is not None:
            raise GPArgumentError("Must pass both X and y when constructing GaussianProcess!")
        else:
            self.K_up_to_date = False

    # ---- device plumbing -------------------------------------------------------------------
    def _reset_device_state(self):
        self._ctx_obj = None
        self._scratch_ctx = None
        self._ctx_pool = None
        self._data_on_device = False
        self._cache = {}
        self._fit_mode = None
        self._dist_plan = None

    partitioned = False
    partition_block = 512
    #: ``eager_alpha = True``: every fit also computes ``alpha`` and brings it to the host, like the reference's
    #: ``compute_K_L_alpha_ll`` (gaussian_process.py:1462) -- for drop-in users who read ``gp.alpha`` after every
    #: ``update_hyperparameters``.  Default False: ``alpha`` is computed on first use (``gp.alpha``, ``predict``, the analytic
    #: gradient); the MAP loop never reads it.  Cost at N = 8192: ~0.3 ms per evaluation (bench.py ``with_alpha``).
    eager_alpha = False

    def _partitioned_possible(self):
        if not self.partitioned or self.use_hyper_deriv or self.T is not None or not self._fast_fit_possible():
            return False
        terms = self._native_terms()
        if terms is None or len(terms) != 1:
            return False
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    @property
    def _ctx(self):
        """Per-instance context (owns the resident factorisation); created on first use."""
        if self._ctx_obj is None:
            self._ctx_obj = _lib.Context(self.device)
            # The contexts of ll_batch are created NOW, before any of them has run: measured on MI355X / ROCm 7,
            # streams created after another context's streams have been busy end up sharing hardware queues with
            # them (two evaluations in flight: 173 evaluations/s instead of 210 at N=8192).  An idle context costs
            # two streams and a few small buffers; its matrix is allocated on first use.
            self._ctx_pool = [[_lib.Context(self.device), -1] for _ in range(max(0, int(self.batch_concurrency) - 1))]
        return self._ctx_obj

    def __getstate__(self):
        # The reference keeps the GP picklable for its process pools (gaussian_process.py:1701-1703);
        # device handles never cross a pickle: they are re-created lazily in the new process.
        st = dict(self.__dict__)
        st["_ctx_obj"] = None
        st["_scratch_ctx"] = None
        st["_ctx_pool"] = None
        st["_data_on_device"] = False
        st["_cache"] = {}
        st["_fit_mode"] = None
        st["_dist_plan"] = None
        st["K_up_to_date"] = False
        return st

    # ---- hyperparameter views (ref: gptools/gaussian_process.py:246-374) ---------------------
    @property
    def hyperprior(self):
        hp = self.k.hyperprior * self.noise_k.hyperprior
        if self.mu is not None:
            hp = hp * self.mu.hyperprior
        return hp

    @property
    def num_dim(self):
        return self.k.num_dim

    def _parts(self):
        return [self.k, self.noise_k] + ([self.mu] if self.mu is not None else [])

    def _view(self, attr):
        return _combine(*[getattr(p, attr) for p in self._parts()])

    def _scatter(self, attr, value, count_attr):
        pos = 0
        for p in self._parts():
            cnt = getattr(p, count_attr)
            setattr(p, attr, value[pos:pos + cnt])
            pos += cnt

    @property
    def fixed_params(self):
        return self._view("fixed_params")

    @fixed_params.setter
    def fixed_params(self, value):
        self._scatter("fixed_params", np.asarray(value, dtype=bool), "num_params")

    @property
    def params(self):
        return self._view("params")

    @params.setter
    def params(self, value):
        self.K_up_to_date = False
        self._scatter("params", np.asarray(value, dtype=float), "num_params")

    @property
    def param_bounds(self):
        return self.hyperprior.bounds

    @param_bounds.setter
    def param_bounds(self, value):
        self.hyperprior.bounds = value

    @property
    def param_names(self):
        return self._view("param_names")

    @param_names.setter
    def param_names(self, value):
        self._scatter("param_names", value, "num_params")

    @property
    def free_params(self):
        return self._view("free_params")

    @free_params.setter
    def free_params(self, value):
        self.K_up_to_date = False
        self._scatter("free_params", np.asarray(value, dtype=float), "num_free_params")

    @property
    def free_param_bounds(self):
        return self._view("free_param_bounds")

    @free_param_bounds.setter
    def free_param_bounds(self, value):
        self._scatter("free_param_bounds", np.asarray(value, dtype=float), "num_free_params")

    @property
    def free_param_names(self):
        return self._view("free_param_names")

    @free_param_names.setter
    def free_param_names(self, value):
        self.K_up_to_date = False
        self._scatter("free_param_names", np.asarray(value, dtype=str), "num_free_params")

    # ---- data ingest (ref: gptools/gaussian_process.py:376-503) ------------------------------
    def add_data(self, X, y, err_y=0, n=0, T=None):
        """Append observations: ``y`` (M,) observed at ``X`` (M, D) with standard deviations ``err_y`` (scalar or (M,))
        and derivative orders ``n`` (scalar or (M, D)).  With ``T`` (M, N') the observations are linear combinations
        ``T f(X)`` of the process at N' points ``X`` (line integrals); once any block has a transform, blocks without one
        get the identity and the blocks sit on the diagonal of ``self.T``.  Forms accepted: ``gptools_amd._ingest``."""
        y = _ingest.targets(y)
        err_y = _ingest.noise_levels(err_y, y)
        X = _ingest.points(X, self.num_dim)
        if T is None and X.shape[0] != y.size:
            raise ValueError("%d points for %d observations (without a transform T there is one point per observation)"
                             % (X.shape[0], y.size))
        n = _ingest.derivative_orders(n, X, self.num_dim, column_rule="not-column")
        if T is not None or self.T is not None:
            block = np.eye(y.size) if T is None else _ingest.linear_map(T, y.size, X.shape[0])
            if self.T is None and self.X is not None:
                self.T = np.eye(self.y.size)               # what was there so far was observed directly
            self.T = block if self.T is None else scipy.linalg.block_diag(self.T, block)
        self.X = X if self.X is None else np.concatenate((self.X, X), axis=0)
        self.n = n if self.n is None else np.concatenate((self.n, n), axis=0)
        self.y = np.concatenate((self.y, y))
        self.err_y = np.concatenate((self.err_y, err_y))
        self.K_up_to_date = False
        self._data_on_device = False
        self._dist_plan = None
        self._data_version = getattr(self, "_data_version", 0) + 1

    # ---- covariance matrices (ref: gptools/gaussian_process.py:1535-1605) --------------------
    def compute_Kij(self, Xi, Xj, ni, nj, noise=False, hyper_deriv=None, k=None):
        """Covariance matrix between ``Xi`` (M, D) and ``Xj`` (P, D) -> (M, P); ``Xj=None`` gives the
        symmetric ``K(Xi, Xi)``.  Native kernels use the fused GPU builder (``gpt_kbuild``); other
        ``Kernel`` subclasses receive the row-major pair list like the reference."""
        if k is None:
            k = self.noise_k if noise else self.k
        Xi = np.atleast_2d(np.asarray(Xi, dtype=float))
        ni = np.atleast_2d(np.asarray(ni, dtype=int))
        kid = getattr(k, "_gpt_kernel_id", None)
        if kid is not None and type(k).__call__ in (Kernel.__call__, _M52_CALL, _ZERO_CALL):
            if kid == _lib.KERNEL_M52 and hyper_deriv is not None:
                raise NotImplementedError("Hyperparameter derivatives have not been implemented!")
            Xj_ = None if Xj is None else np.atleast_2d(np.asarray(Xj, dtype=float))
            nj_ = None if Xj is None else np.atleast_2d(np.asarray(nj, dtype=int))
            return self._ctx.kbuild(kid, k.params, Xi, ni, Xj_, nj_, hyper_deriv=hyper_deriv,
                                    noise_n=getattr(k, "n", None))
        if type(k) is ProductKernel and hyper_deriv is None and k._native_factors() is not None:
            nat = k._native_factors()
            Xj_ = None if Xj is None else np.atleast_2d(np.asarray(Xj, dtype=float))
            nj_ = None if Xj is None else np.atleast_2d(np.asarray(nj, dtype=int))
            return self._ctx.kbuild2(nat[0], nat[1], nat[2], nat[3], Xi, ni, Xj_, nj_)
        symmetric = Xj is None
        if symmetric:
            Xj, nj = Xi, ni
        Xj = np.atleast_2d(np.asarray(Xj, dtype=float))
        nj = np.atleast_2d(np.asarray(nj, dtype=int))
        M, P = Xi.shape[0], Xj.shape[0]
        Kij = k(np.repeat(Xi, P, axis=0), np.tile(Xj, (M, 1)), np.repeat(ni, P, axis=0), np.tile(nj, (M, 1)),
                hyper_deriv=hyper_deriv, symmetric=symmetric)
        return np.reshape(Kij, (M, -1))

    # ---- lazily materialised state ------------------------------------------------------------
    def _cached(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    @property
    def K(self):
        """Noise-free training covariance (ref attribute ``K``, gaussian_process.py:1431)."""
        self.compute_K_L_alpha_ll()
        return self._cached("K", lambda: self.compute_Kij(self.X, None, self.n, None))

    @property
    def noise_K(self):
        """Noise part of the training covariance (ref: gaussian_process.py:1434-1439)."""
        self.compute_K_L_alpha_ll()
        return self._cached("noise_K", self._noise_K)

    def _noise_K(self):
        N = self.X.shape[0]
        if isinstance(self.noise_k, ZeroKernel):
            return np.zeros((N, N))
        if isinstance(self.noise_k, DiagonalNoiseKernel):
            return self.noise_k.params[0] ** 2.0 * np.eye(N)
        return self.compute_Kij(self.X, None, self.n, None, noise=True)

    @property
    def L(self):
        """Lower Cholesky factor of ``K_tot`` (ref: gaussian_process.py:1452)."""
        self.compute_K_L_alpha_ll()
        return self._cached("L", lambda: self._ctx.get_L(len(self.y)))

    @property
    def alpha(self):
        """``K_tot^-1 (y - T mu)`` as an (N_y, 1) column (ref: gaussian_process.py:1462)."""
        self.compute_K_L_alpha_ll()
        return self._cached("alpha", lambda: self._ctx.get_alpha(len(self.y)).reshape(-1, 1))

    # ---- the fit (ref: gptools/gaussian_process.py:1418-1522) --------------------------------
    def _y_alph(self):
        if self.mu is None:
            return self.y
        mu_alph = self.mu(self.X, self.n)
        if self.T is not None:
            mu_alph = self.T.dot(mu_alph)
        return self.y - mu_alph

    def _upload_data(self, ctx):
        """X, n and -- when the observations are linear transforms of the latent values -- T (ref :376-503)."""
        ctx.set_data(self.X, self.n)
        if self.T is not None:
            ctx.set_T(self.T)

    def _native_terms(self, k=None):
        """The covariance kernel as a list of ``(kernel_id, params)`` terms the HIP library evaluates itself -- a native
        kernel, or a SumKernel tree of them (ref: gptools/kernel/core.py:549-584) -- else ``None``."""
        k = self.k if k is None else k
        if type(k) is SumKernel:
            a, b = self._native_terms(k.k1), self._native_terms(k.k2)
            return None if a is None or b is None or len(a) + len(b) > 8 else a + b
        if (getattr(k, "_gpt_kernel_id", None) in _NATIVE_FIT and type(k).__call__ in (Kernel.__call__, _M52_CALL)):
            return [(k._gpt_kernel_id, np.array(k.params, dtype=float))]
        if type(k) is ProductKernel:
            # k1 * k2 of two native kernels: one PRODUCT term of the fused builder (ref: gptools/kernel/core.py:587-671)
            nat = k._native_factors()
            return None if nat is None else [nat]
        return None

    def _fast_fit_possible(self):
        return (self._native_terms() is not None and
                isinstance(self.noise_k, (ZeroKernel, DiagonalNoiseKernel)))

    def _device_fit(self, ctx, terms, noise_var, y_alph, diag_add):
        if any(len(t) == 4 for t in terms):
            return ctx.fit_terms(terms, noise_var, y_alph, self.err_y, diag_add)
        if len(terms) == 1:
            return ctx.fit(terms[0][0], terms[0][1], noise_var, y_alph, self.err_y, diag_add)
        return ctx.fit_sum([t[0] for t in terms], [t[1] for t in terms], noise_var, y_alph, self.err_y, diag_add)

    def compute_K_L_alpha_ll(self, need_factor=True):
        """Build ``K_tot``, factor it and evaluate the log-posterior ``ll`` on the GPU (no-op while
        ``K_up_to_date``).  Raises ``numpy.linalg.LinAlgError`` if ``K_tot`` is not positive
        definite, like ``scipy.linalg.cholesky`` in the reference.

        ``need_factor=False`` (``update_hyperparameters``): only ``ll`` is wanted, which lets a ``partitioned`` GP
        evaluate it with the factorisation spread over the ranks of the job; the factor then is not on this GPU and
        the next call that needs it refits locally."""
        if self.K_up_to_date and (self._fit_mode != "partitioned" or not need_factor):
            return
        if self.X is None:
            raise GPArgumentError("No data have been added to the GaussianProcess!")
        self._cache = {}
        y_alph = self._y_alph()
        diag_add = self.diag_factor * sys.float_info.epsilon
        if not need_factor and self._partitioned_possible():
            from .dist import DistributedLML
            if self._dist_plan is None:
                self._dist_plan = DistributedLML(self.X, self.n, nb=self.partition_block, device=self.device)
            kid, kparams = self._native_terms()[0]
            noise_var = 0.0 if isinstance(self.noise_k, ZeroKernel) else self.noise_k.params[0] ** 2.0
            ll_data, _ = self._dist_plan.fit(kid, kparams, y_alph, self.err_y, noise_var=noise_var,
                                             diag_factor=self.diag_factor)
            self._fit_mode = "partitioned"
            self.ll = ll_data + self.hyperprior(self.params)
            self.K_up_to_date = True
            return
        ctx = self._ctx
        # (eager: the device enqueues alpha behind the factorisation of the same call; get_alpha below is then a copy)
        ctx.set_option("eager_alpha", 1 if self.eager_alpha else 0)
        if self._fast_fit_possible():
            if not self._data_on_device:
                self._upload_data(ctx)
                self._data_on_device = True
            if isinstance(self.noise_k, ZeroKernel):
                noise_var = 0.0
            else:
                noise_var = self.noise_k.params[0] ** 2.0
            ll_data, _ = self._device_fit(ctx, self._native_terms(), noise_var, y_alph, diag_add)
            self._fit_mode = "kernel"
        else:
            # T (linear transform) or a Python-defined kernel: K is still built by the GPU builder
            # where the kernel is native, the small T products are host GEMMs, the factorisation and
            # solves run on the GPU (gpt_fit_matrix).
            K = self.compute_Kij(self.X, None, self.n, None)
            self._cache["K"] = K
            noise_K = self._noise_K()
            self._cache["noise_K"] = noise_K
            KnK = K + noise_K
            if self.T is not None:
                KnK = self.T.dot(KnK).dot(self.T.T)
            K_tot = KnK + np.diag(self.err_y ** 2.0) + diag_add * np.eye(len(self.y))
            ll_data, _ = ctx.fit_matrix(K_tot, y_alph)
            self._data_on_device = False
            self._fit_mode = "matrix"
        self.ll = ll_data + self.hyperprior(self.params)          # log-posterior (ref :1469)
        if self.eager_alpha:
            self._cache["alpha"] = ctx.get_alpha(len(self.y)).reshape(-1, 1)      # (ref :1462)
        if self.use_hyper_deriv:
            self._compute_ll_deriv()
        self.K_up_to_date = True

    def _compute_ll_deriv(self):
        """Gradient of the log-posterior with respect to the free hyperparameters
        (ref: gptools/gaussian_process.py:1471-1520): 1/2 (alpha^T dK alpha - tr(K_tot^-1 dK))."""
        warnings.warn("Use of hyperparameter derivatives is experimental!")
        ctx = self._ctx
        Ny = len(self.y)
        alpha = ctx.get_alpha(Ny)
        self._cache["alpha"] = alpha.reshape(-1, 1)

        def term(dK, transform=True):
            if self.T is not None and transform:
                dK = self.T.dot(dK).dot(self.T.T)
            W = ctx.solve_L(dK)                     # L^-1 dK
            W2 = ctx.solve_L(np.ascontiguousarray(W.T))    # L^-1 (L^-1 dK)^T -> trace equals tr(K^-1 dK)
            return 0.5 * (alpha.dot(dK.dot(alpha)) - np.trace(W2))

        ll_deriv = np.zeros(len(self.free_params))
        terms = self._native_terms() if self._fit_mode == "kernel" else None
        knk = self.k
        free_idx = np.arange(0, len(knk.params), dtype=int)[~np.asarray(knk.fixed_params, dtype=bool)]
        tix = lix = None
        if terms is not None:
            # which kernel term each free parameter belongs to; the device path needs every FREE parameter in a
            # squared-exponential term (terms of other kernels may take part in the sum with all their parameters fixed:
            # they have no hyperparameter derivatives in the reference either, kernel/core.py:723-726)
            bounds = np.cumsum([0] + [len(t[1]) + (len(t[3]) if len(t) == 4 else 0) for t in terms])
            tix = [int(np.searchsorted(bounds, pi, side="right") - 1) for pi in free_idx]
            lix = [int(pi - bounds[t]) for pi, t in zip(free_idx, tix)]
        if terms is not None and all(len(terms[t]) == 2 and terms[t][0] == _lib.KERNEL_SE for t in tix):
            # device path (gpt_ll_grad): K_tot^-1 once (2 N^3 / 3 flop on the MFMA GEMM, whatever the number of
            # parameters), then one fused pass over the pairs per group of parameters; dK never exists.  With a linear
            # transform T the pass runs over the latent points against T^T K_tot^-1 T (two more GEMMs on the device).
            g_dev = ctx.ll_grad(tix, lix)
            ll_deriv[:len(free_idx)] = g_dev[:-1]
            if isinstance(self.noise_k, DiagonalNoiseKernel) and not isinstance(self.noise_k, ZeroKernel) \
                    and not self.noise_k.fixed_params[0]:
                ll_deriv[len(self.k.free_params)] = 2.0 * self.noise_k.params[0] * g_dev[-1]
        else:
            if isinstance(self.noise_k, ZeroKernel):
                knk = self.k
            elif isinstance(self.noise_k, DiagonalNoiseKernel):
                knk = self.k
                if not self.noise_k.fixed_params[0]:
                    # (the reference does not transform this term: 2 sigma_n eye(len(y)), ref :1482-1488)
                    ll_deriv[len(self.k.free_params)] = term(2.0 * self.noise_k.params[0] * np.eye(Ny), transform=False)
            else:
                knk = self.k + self.noise_k
            free_idx = np.arange(0, len(knk.params), dtype=int)[~np.asarray(knk.fixed_params, dtype=bool)]
            for i, pi in enumerate(free_idx):
                ll_deriv[i] = term(self.compute_Kij(self.X, None, self.n, None, k=knk, hyper_deriv=int(pi)))
        if self.mu is not None:
            free_idx = np.arange(0, len(self.mu.params), dtype=int)[~self.mu.fixed_params]
            for i, pi in enumerate(free_idx):
                dmu = self.mu(self.X, self.n, hyper_deriv=int(pi))
                if self.T is not None:
                    dmu = self.T.dot(dmu)
                ll_deriv[i + len(knk.free_params)] = dmu.dot(alpha)
        free_idx = np.arange(0, len(self.params), dtype=int)[~np.asarray(self.fixed_params[:], dtype=bool)]
        for i, pi in enumerate(free_idx):
            ll_deriv[i] += self.hyperprior(self.params[:], hyper_deriv=int(pi))
        self.ll_deriv = ll_deriv

    # ---- hyperparameter update (ref: gptools/gaussian_process.py:1332-1416) -------------------
    def _assign_free(self, values):
        """Distribute a vector of FREE hyperparameters over the kernel, the noise kernel and the mean function, in the
        order ``free_params`` lists them (ref: gaussian_process.py:1377-1384)."""
        values = np.asarray(values, dtype=float)
        at = 0
        for part in self._parts():
            cnt = len(part.free_params)
            part.set_hyperparams(values[at:at + cnt])
            at += cnt
        self.K_up_to_date = False

    def update_hyperparameters(self, new_params, hyper_deriv_handling="default", exit_on_bounds=True,
                               inf_on_error=True):
        """Set the free hyperparameters and evaluate the negative log-posterior there -- the objective an optimiser
        minimises (ref: gptools/gaussian_process.py:1332-1416).

        Returns ``-ll``; ``(-ll, -ll_deriv)`` when the GP uses hyperparameter derivatives; ``hyper_deriv_handling`` =
        ``'value'`` / ``'deriv'`` asks for one of the two alone whatever the GP's setting.  Hyperparameters the prior
        rules out (``exit_on_bounds``) and any failure of the evaluation -- a covariance matrix that is not positive
        definite above all -- count as ``+inf`` (gradient: zeros) while ``inf_on_error`` is set; otherwise they raise."""
        want_value = hyper_deriv_handling != "deriv"
        want_deriv = hyper_deriv_handling == "deriv" or (hyper_deriv_handling == "default" and self.use_hyper_deriv)
        configured = self.use_hyper_deriv
        self.use_hyper_deriv = want_deriv
        try:
            self._assign_free(new_params)
            if exit_on_bounds and np.isinf(self.hyperprior(self.params)):
                raise GPImpossibleParamsError("the hyperprior excludes these hyperparameters")
            self.compute_K_L_alpha_ll(need_factor=False)
            value, deriv = -1.0 * self.ll, (-1.0 * self.ll_deriv if want_deriv else None)
        except Exception as exc:
            if not inf_on_error:
                raise
            if self.verbose and not isinstance(exc, GPImpossibleParamsError):
                warnings.warn("evaluation failed at free hyperparameters %s and counts as +inf:\n%s"
                              % (self.free_params[:], traceback.format_exc()))
            value, deriv = np.inf, np.zeros(len(self.free_params))
        finally:
            self.use_hyper_deriv = configured
        if want_value and want_deriv:
            return (value, deriv)
        return value if want_value else deriv

    # ---- independent evaluations (ref: gptools/gaussian_process.py:1607-1692) ----
    #: independent LML evaluations kept in flight on one GPU.  While one factorisation is in its latency-bound tail
    #: (panel chain, most CUs idle) another is in its update-bound head: measured on MI355X, 2 in flight give
    #: 211 evaluations/s against 164 at N=8192 and 811 against 476 at N=4096; more than 2 lose again (the
    #: diagonal-block kernels compete for the CUs reserved for them).
    batch_concurrency = 2
    #: at most this many points: ll_batch (hence compute_ll_matrix, the finite-difference gradients of
    #: optimize_hyperparameters) evaluates ``batch_grid`` hyperparameter vectors per launch sequence (gpt_fit_batch) instead
    #: of one per context and host thread; measured on MI355X at N = 1024: see DESIGN.md section 7.2
    batch_grid_max_n = 4096
    #: device memory the batch's matrices may take (bytes): the chunk size is batch_grid or what fits, whichever is smaller
    batch_grid_bytes = 16 << 30
    batch_grid = 256                                          # (measured: N = 256 262 k / 569 k / 699 k evaluations/s at 64 / 256 / 1024
                                                              #  per launch sequence, N = 1024 45 k / 62 k / 64 k at 64 / 256 / 512)

    def _batch_contexts(self, count):
        self._ctx                                            # creates the pool together with the main context
        pool = self._ctx_pool
        while len(pool) < count - 1:                         # batch_concurrency raised later: late contexts (slower)
            pool.append([_lib.Context(self.device), -1])
        return pool

    def ll_batch(self, param_list, exit_on_bounds=True):
        """Log-posterior at each free-parameter vector of ``param_list`` -- what ``-update_hyperparameters(p)``
        returns for every ``p`` (``-inf`` where that gives ``+inf``) -- with ``batch_concurrency`` evaluations in
        flight on this GPU and, under a ``torch.distributed`` job, the list spread over the ranks
        (``gptools_amd.replicas``).  The GP's own hyperparameters are left unchanged."""
        param_list = [np.asarray(p, dtype=float) for p in param_list]
        keep = np.array(self.free_params[:], dtype=float)
        world = replicas.world_size()
        # Independent evaluations are never partitioned: the ranks evaluate DIFFERENT hyperparameters here (their slices
        # of the list), and a partitioned evaluation is a collective that every rank must enter with the same ones.
        keep_partitioned = self.partitioned
        self.partitioned = False
        try:
            if world > 1 and len(param_list) > 1:
                import torch.distributed as dist
                rank = dist.get_rank()
                mine = self._ll_batch_local(param_list[rank::world], exit_on_bounds)
                parts = replicas.distributed_map(lambda r: mine if r == rank else None, range(world))
                out = np.empty(len(param_list))
                for r in range(world):
                    out[r::world] = parts[r]
                return out
            return self._ll_batch_local(param_list, exit_on_bounds)
        finally:
            self._assign_free(keep)
            self.partitioned = keep_partitioned

    def _ll_batch_local(self, param_list, exit_on_bounds):
        out = np.full(len(param_list), -np.inf)
        if not param_list:
            return out
        if self.X is None:
            raise GPArgumentError("No data have been added to the GaussianProcess!")
        B = max(1, int(self.batch_concurrency))
        if B == 1 or len(param_list) == 1 or not self._fast_fit_possible():
            for i, p in enumerate(param_list):
                out[i] = -1.0 * self.update_hyperparameters(p, hyper_deriv_handling="value",
                                                            exit_on_bounds=exit_on_bounds)
            return out
        # host part, in order: the kernel / mean objects are shared, so the numeric inputs of every evaluation are
        # extracted one after another; only the GPU work overlaps
        diag_add = self.diag_factor * sys.float_info.epsilon
        jobs = []
        for i, p in enumerate(param_list):
            self._assign_free(p)
            prior = self.hyperprior(self.params)
            if exit_on_bounds and np.isinf(prior):
                continue                                            # impossible parameters: stays -inf
            noise_var = 0.0 if isinstance(self.noise_k, ZeroKernel) else self.noise_k.params[0] ** 2.0
            jobs.append((i, self._native_terms(), None, noise_var,
                         np.array(self._y_alph(), dtype=float), prior))
        if not self._data_on_device:
            self._upload_data(self._ctx)
            self._data_on_device = True
        version = getattr(self, "_data_version", 0)
        if (max(len(self.y), self.X.shape[0]) <= self.batch_grid_max_n and int(self.batch_grid) > 1 and jobs
                and all([(t[0], t[2] if len(t) == 4 else -1) for t in j[1]] == [(t[0], t[2] if len(t) == 4 else -1) for t in jobs[0][1]]
                        for j in jobs)):
            # small N: the whole batch in ONE launch sequence (gpt_fit_batch_terms: every kernel of the factorisation carries the
            # batch in a grid dimension), batch_grid evaluations at a time; bit-identical to one gpt_fit_terms per vector.  Any
            # model of native kernels -- sums, products, with or without the linear transform T (the reference's
            # grids take any model: ref gaussian_process.py:1607-1692)
            err_y = np.asarray(self.err_y, dtype=float)
            self._cache = {}
            NP = -(-(len(self.y) + 1) // 128) * 128
            # chunk size: batch_grid, or what fits batch_grid_bytes, or what fits HALF of the device memory that is free right now
            # (other ranks / processes may share the GPU) -- and should the allocation still fail, half as many, down to the
            # one-context-per-thread path below (a MemoryError used to abort ll_batch / compute_ll_matrix)
            budget = min(int(self.batch_grid_bytes), self._ctx.mem_info()[0] // 2)
            per = 8 * NP * NP + 8 * 9216 * (NP // 128)
            if self.T is not None:                               # + every element's K over the latent points and T K
                NxP = -(-self.X.shape[0] // 16) * 16
                per += 8 * NxP * NxP + 8 * (-(-len(self.y) // 64) * 64) * NxP
            G = max(1, min(int(self.batch_grid), budget // per))
            s0 = 0
            while s0 < len(jobs) and G >= 1:
                chunk = jobs[s0:s0 + G]
                try:
                    ll, _, info = self._ctx.fit_batch_terms([j[1] for j in chunk], np.array([j[3] for j in chunk]),
                                                            np.array([j[4] for j in chunk]), err_y, diag_add)
                except MemoryError:
                    self._ctx.release_batch_scratch()
                    G //= 2
                    continue
                except (ValueError, ArithmeticError):
                    # an argument the library rejects for one element (e.g. a kernel parameter out of its domain) fails the
                    # whole call: that chunk one vector at a time, with the +inf policy of update_hyperparameters
                    ll, info = np.zeros(len(chunk)), np.ones(len(chunk), dtype=int)
                    for q, j in enumerate(chunk):
                        try:
                            ll[q], info[q] = self._device_fit(self._ctx, j[1], j[3], j[4], diag_add)[0], 0
                        except (np.linalg.LinAlgError, ValueError, ArithmeticError):
                            pass
                for j, l, bad in zip(chunk, ll, info):
                    if bad == 0:
                        out[j[0]] = l + j[5]
                s0 += len(chunk)
            if 8 * NP * NP * min(G, len(jobs)) > (2 << 30):
                self._ctx.release_batch_scratch()            # a large scratch goes back to the device once the list is done
            if s0 >= len(jobs):
                return out
            jobs = jobs[s0:]                                  # (no chunk size fits: the rest one context per thread)
        ctxs = [[self._ctx, version]] + self._batch_contexts(B)
        for c in ctxs[1:]:
            if c[1] != version:                                      # data added since this context last saw it
                self._upload_data(c[0])
                c[1] = version
        self._cache = {}
        err_y = np.asarray(self.err_y, dtype=float)
        import threading
        lock = threading.Lock()
        it = iter(jobs)

        def worker(ctx):
            while True:
                with lock:
                    job = next(it, None)
                if job is None:
                    return
                i, terms, _, noise_var, y_alph, prior = job
                try:
                    ll_data, _ = self._device_fit(ctx, terms, noise_var, y_alph, diag_add)
                    out[i] = ll_data + prior
                except (np.linalg.LinAlgError, ValueError, ArithmeticError):
                    pass                                            # the +inf policy of update_hyperparameters
        threads = [threading.Thread(target=worker, args=(c[0],)) for c in ctxs[:B]]
        with _lib.concurrent_evaluations():          # several chains at once: event edges (include/gpt_hip.h)
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        return out

    def _batched_fd(self, eps, bounds):
        """``(fun, jac)`` for scipy.optimize.minimize: ``fun`` is ``update_hyperparameters``; ``jac`` is the
        forward-difference gradient scipy would form itself (``approx_derivative(..., '2-point', abs_step=eps,
        bounds=...)``), with the perturbed points evaluated through the batched evaluator.  scipy's routine is run
        twice -- once to record the points it asks for, once on the cached values -- so step signs near bounds and
        the rounding of ``(x + h) - x`` are exactly its own."""
        from scipy.optimize._numdiff import approx_derivative
        last = {}

        def fun(x):
            x = np.array(x, dtype=float)
            v = self.update_hyperparameters(x)
            last["x"], last["f"] = x, v
            return v

        def jac(x):
            x = np.array(x, dtype=float)
            f0 = last["f"] if ("x" in last and np.array_equal(last["x"], x)) else fun(x)
            kw = dict(method="2-point", abs_step=eps, f0=f0, bounds=(bounds[:, 0], bounds[:, 1]))
            pts = []
            approx_derivative(lambda z: (pts.append(np.array(z, dtype=float)), 0.0)[1], x, **kw)
            vals = -1.0 * self._ll_batch_local(pts, True)
            self._assign_free(x)                                 # back to the expansion point
            table = {p.tobytes(): v for p, v in zip(pts, vals)}
            return approx_derivative(lambda z: table[np.array(z, dtype=float).tobytes()], x, **kw)
        return fun, jac

    def compute_ll_matrix(self, bounds, num_pts):
        """Log-posterior on a regular grid over the free hyperparameters (ref: gptools/gaussian_process.py:1607-1692).

        ``bounds``: one ``(low, high)`` pair for all free hyperparameters or one pair each; ``num_pts``: one count or
        one per hyperparameter.  Returns ``(ll_vals, param_vals)``: the grid axes ``param_vals[i] = linspace(low_i,
        high_i, num_pts[i])`` and ``ll_vals`` of shape ``num_pts`` (axis i = free hyperparameter i; ``-inf`` where the
        prior excludes the point or the evaluation fails).  The reference walks the grid recursively, one evaluation
        at a time; here all points go through :meth:`ll_batch`.  The GP is left at its present hyperparameters."""
        here = np.array(self.free_params[:], dtype=float)
        nfree = here.size
        box = np.asarray(bounds, dtype=float)
        if box.ndim == 1:
            box = box[None, :]
        if box.ndim != 2 or box.shape[1] != 2:
            raise ValueError("bounds must be a (low, high) pair or one pair per free hyperparameter")
        if box.shape[0] == 1:
            box = np.repeat(box, nfree, axis=0)
        counts = np.asarray(num_pts, dtype=int)
        if counts.ndim == 0:
            counts = np.full(box.shape[0], int(counts))
        elif counts.size != nfree:
            raise ValueError("num_pts lists %d counts for %d free hyperparameters" % (counts.size, nfree))
        axes = [np.linspace(lo, hi, c) for (lo, hi), c in zip(box[:nfree], counts)]
        mesh = np.meshgrid(*axes, indexing="ij")
        points = np.column_stack([m.ravel() for m in mesh])
        ll_vals = self.ll_batch(list(points)).reshape([int(c) for c in counts[:nfree]])
        self.update_hyperparameters(here)
        return (ll_vals, axes)

    # ---- MAP estimate (ref: gptools/gaussian_process.py:623-783, :2443-2486) ------------------
    def optimize_hyperparameters(self, method="SLSQP", opt_kwargs={}, verbose=False, random_starts=None,
                                 num_proc=None, max_tries=1, batch_fd=None):
        """Maximise the log-posterior with ``scipy.optimize.minimize`` from ``random_starts`` draws
        of the hyperprior (0: start from the current values).  Every objective evaluation is one GPU
        fit.  ``num_proc`` is accepted for compatibility: a HIP context must not be shared across forked
        workers, so inside one process the starts run one after another; under a ``torch.distributed`` job
        (one process per GPU) they are spread over the ranks instead (``gptools_amd.replicas``) and every
        rank returns the same best result."""
        opt_kwargs = dict(opt_kwargs or {})
        if "method" in opt_kwargs:
            method = opt_kwargs["method"]
        else:
            opt_kwargs["method"] = method
        if num_proc is None:
            num_proc = 1
        param_ranges = np.array(self.free_param_bounds[:], dtype=float)
        lo, hi = param_ranges[:, 0], param_ranges[:, 1]
        lo[np.isnan(lo) | np.isinf(lo)] = -1e16
        hi[np.isnan(hi) | np.isinf(hi)] = 1e16
        free_mask = ~np.asarray(self.fixed_params[:], dtype=bool)

        def draw():
            return np.atleast_2d(self.hyperprior.random_draw(size=random_starts).T)[:, free_mask]

        if random_starts == 0:
            param_samples = [np.array(self.free_params[:], dtype=float)]
        else:
            if random_starts is None:
                random_starts = max(num_proc, 1)
            param_samples = draw()
        opt_kwargs.setdefault("bounds", param_ranges)
        if self.use_hyper_deriv:
            opt_kwargs["jac"] = True

        objective = self.update_hyperparameters
        # Finite-difference gradients (what scipy does itself when no `jac` is given) through ll_batch: the p
        # perturbed points of one gradient are independent evaluations, kept two in flight on the GPU.  The points
        # and the difference formula are scipy's own (approx_derivative is replayed), so the iterates are the same.
        fd_eps = {"L-BFGS-B": 1e-8, "TNC": 1e-8, "SLSQP": 1.4901161193847656e-08}
        partitioned = self._partitioned_possible()       # every evaluation is a collective: no local batching,
        if (batch_fd is not False and int(self.batch_concurrency) > 1 and not self.use_hyper_deriv and not partitioned
                and "jac" not in opt_kwargs and method in fd_eps and self._fast_fit_possible()):
            eps = (opt_kwargs.get("options") or {}).get("eps", fd_eps[method])
            objective, opt_kwargs["jac"] = self._batched_fd(eps, np.asarray(opt_kwargs["bounds"], dtype=float))

        def run(samp):
            try:
                return scipy.optimize.minimize(objective, samp, **opt_kwargs)
            except Exception:
                if self.verbose:
                    warnings.warn("start %s dropped, the optimiser raised (free hyperparameters now %s):\n%s"
                                  % (samp, self.free_params[:], traceback.format_exc()), RuntimeWarning)
                return None

        best, finished = None, []
        for attempt in range(max(int(max_tries), 0)):
            if attempt > 0 and random_starts != 0:
                param_samples = draw()                        # nothing usable last time: fresh draws
            if partitioned and replicas.world_size() > 1:
                # ... and every rank walks all the starts (rank 0's draws), in the same order
                param_samples = list(replicas.shared(np.asarray(param_samples)))
                outcomes = [run(s) for s in param_samples]
            elif replicas.world_size() > 1 and len(param_samples) > 1:
                # one start per GPU: the ranks of the torch.distributed job replace the reference's process pool
                param_samples = replicas.shared(np.asarray(param_samples))
                outcomes = replicas.distributed_map(run, list(param_samples))
            else:
                outcomes = [run(s) for s in param_samples]
            finished = [r for r in outcomes if r is not None]
            usable = [r for r in finished if np.isfinite(r.fun)]
            if usable:
                best = min(usable, key=lambda r: r.fun)
                break
        if best is None:
            raise ValueError("no start of the optimiser ended at a finite log-posterior; widen or shift the parameter "
                             "bounds, start elsewhere or use more random starts")
        self.update_hyperparameters(best.x)
        if verbose:
            print("%d starts completed; the best one:" % len(finished))
            print(best)
            print("log-posterior %.6g" % (-best.fun))
            for name, v in zip(self.free_param_names[:], best.x):
                print("  %s = %.6g" % (str(name).replace("\\", ""), v))
        if not best.success:
            warnings.warn("%s stopped without converging (status %s: %s); the hyperparameters it ended at are kept but "
                          "are probably not the optimum -- other bounds, starting points or more random starts may help"
                          % (method, best.status, best.message), RuntimeWarning)
        box = np.asarray(self.free_param_bounds[:], dtype=float)
        # (within 0.1 % of an edge counts as on it, as in the reference)
        if np.any(best.x <= 1.001 * box[:, 0]) or np.any(best.x >= 0.999 * box[:, 1]):
            warnings.warn("the optimum sits on the edge of the parameter bounds\n%s\nat %s: the bounds, not the data, "
                          "decided it" % (box, best.x))
        return (best, len(finished))

    # ---- prediction (ref: gptools/gaussian_process.py:785-1034) ------------------------------
    def _check_predict_args(self, Xstar, n, output_transform=None):
        """Test points, their derivative orders and the optional output transform in the layout ``gpt_predict`` takes
        (forms accepted: ``gptools_amd._ingest``; ref: gaussian_process.py:913-963)."""
        Xstar = _ingest.points(Xstar, self.num_dim, "Xstar")
        n = _ingest.derivative_orders(n, Xstar, self.num_dim, column_rule="row")
        if output_transform is not None:
            output_transform = _ingest.linear_map(output_transform, None, Xstar.shape[0], "output_transform")
        return Xstar, n, output_transform

    def predict(self, Xstar, n=0, noise=False, return_std=True, return_cov=False, full_output=False,
                return_samples=False, num_samples=1, samp_kwargs={}, return_mean_func=False, use_MCMC=False,
                full_MC=False, rejection_func=None, ddof=1, output_transform=None, **kwargs):
        """Predictive mean (and std / covariance) at ``Xstar`` (M, D) for derivative orders ``n``.

        Returns ``mean``, ``(mean, std)``, ``(mean, cov)`` or the ``full_output`` dict exactly like the
        reference's non-MCMC branch (``return_samples`` / ``full_MC`` draw through :meth:`draw_sample`); marginalising
        the hyperparameters by MCMC is outside the accelerated path."""
        if use_MCMC:
            raise NotImplementedError("MCMC marginalisation of the hyperparameters is outside the accelerated hot "
                                      "path (SURVEY.md section 8).")
        Xstar, n, output_transform = self._check_predict_args(Xstar, n, output_transform)
        self.compute_K_L_alpha_ll()
        need_cov = (return_cov or full_output or return_samples or full_MC or
                    (output_transform is not None and (return_std or return_cov)))
        need_std = return_std or need_cov
        if self._fit_mode == "kernel":
            want = 2 if need_cov else (1 if need_std else 0)
            noise_params = noise_n = None
            if noise and isinstance(self.noise_k, DiagonalNoiseKernel) and not isinstance(self.noise_k, ZeroKernel):
                noise_params, noise_n = self.noise_k.params, self.noise_k.n
            mean, std, covariance = self._ctx.predict(Xstar, n, want, noise_params, noise_n)
        else:
            mean, std, covariance = self._predict_general(Xstar, n, noise, need_std)
        mean_func = None
        if self.mu is not None:
            mean_func = self.mu(Xstar, n)
            mean = mean + mean_func
        if output_transform is not None:
            mean = output_transform.dot(mean)
            if mean_func is not None:
                mean_func = output_transform.dot(mean_func)
            if covariance is not None:
                covariance = output_transform.dot(covariance.dot(output_transform.T))
                std = np.sqrt(np.diagonal(covariance))
        if not need_std:
            return mean
        samps = None
        if return_samples or full_MC:                                    # ref :990-1005
            samps = self.draw_sample(Xstar, n=n, num_samp=num_samples, mean=mean, cov=covariance, **samp_kwargs)
            if rejection_func:
                good = [samp for samp in samps.T if rejection_func(samp)]
                if len(good) == 0:
                    raise ValueError("Did not get any good samples!")
                samps = np.asarray(good, dtype=float).T
            if full_MC:
                mean = np.mean(samps, axis=1)
                covariance = np.cov(samps, rowvar=1, ddof=ddof)
                std = np.sqrt(np.diagonal(covariance))
        if full_output:
            out = {"mean": mean, "std": std, "cov": covariance}
            if samps is not None:
                out["samp"] = samps
            if return_mean_func and mean_func is not None:
                out["mean_func"] = mean_func
                out["cov_func"] = np.zeros((len(mean_func), len(mean_func)), dtype=float)
                out["std_func"] = np.zeros_like(mean_func)
                out["mean_without_func"] = mean - mean_func
                out["cov_without_func"] = covariance
                out["std_without_func"] = std
            return out
        if return_cov:
            return (mean, covariance)
        return (mean, std)

    # ---- posterior samples (ref: gptools/gaussian_process.py:1155-1330) --------------------------------------
    def draw_sample(self, Xstar, n=0, num_samp=1, rand_vars=None, rand_type="standard normal", diag_factor=1e3,
                    method="cholesky", num_eig=None, mean=None, cov=None, modify_sign=None, **kwargs):
        """Samples ``y* = mean + L u`` of the GP at ``Xstar`` -> (M, num_samp), like the reference: without
        ``rand_vars`` and with ``method='cholesky'`` the draw goes through ``numpy.random.multivariate_normal``
        (falling back on ``'eig'`` if that fails); with ``rand_vars`` (standard normal, or uniform mapped through
        the normal quantile function) the square root ``L`` of ``cov + diag_factor * eps * I`` is the lower Cholesky
        factor -- computed on the GPU -- or ``Q sqrt(Lambda)`` from ``scipy.linalg.eigh``.  The predictive mean and
        covariance come from :meth:`predict` (device path) unless given."""
        if (mean is None and cov is None and rand_vars is not None and method == "cholesky"
                and set(kwargs) <= {"noise"} and rand_type in ("standard normal", "uniform")):
            # Device route (gpt_cov_sample): the predictive covariance stays in HBM, its Cholesky factor is formed there and only
            # L u (M x num_samp) comes back -- nothing of size M^2 crosses PCIe.  Same arithmetic as below (ref :1295-1300, :1330).
            self.compute_K_L_alpha_ll()
            if self._fit_mode == "kernel":
                Xs, ns, _ = self._check_predict_args(Xstar, n)
                noise = bool(kwargs.get("noise", False))
                noise_params = noise_n = None
                if noise and isinstance(self.noise_k, DiagonalNoiseKernel) and not isinstance(self.noise_k, ZeroKernel):
                    noise_params, noise_n = self.noise_k.params, self.noise_k.n
                M = Xs.shape[0]
                ne = M if (num_eig is None or num_eig > M) else max(int(num_eig), 1)
                rv = np.atleast_2d(np.asarray(rand_vars, dtype=float))[:ne, :]
                if rv.shape[0] == M:              # (anything else is the reference's shape error: raised by the host route)
                    mean_d, _, _ = self._ctx.predict(Xs, ns, 2, noise_params, noise_n, device_cov=True)
                    if self.mu is not None:
                        mean_d = mean_d + self.mu(Xs, ns)
                    if rand_type == "uniform":
                        from scipy.stats import norm as _norm
                        rv = _norm.ppf(rv)
                    return np.atleast_2d(mean_d).T + self._ctx.cov_sample(diag_factor * sys.float_info.epsilon, rv)
        if mean is None or cov is None:
            both = self.predict(Xstar, n=n, full_output=True, **kwargs)
            mean, cov = both["mean"], both["cov"]
        mean, cov = np.asarray(mean, dtype=float), np.asarray(cov, dtype=float)
        M = mean.size
        if rand_vars is None and method != "eig":
            # no variates given: numpy's own multivariate normal (global random state), the eigen route as its fallback
            try:
                return np.random.multivariate_normal(mean, cov, num_samp).T
            except np.linalg.LinAlgError as exc:
                if self.verbose:
                    warnings.warn("multivariate_normal failed (%s); using the eigendecomposition instead" % exc, RuntimeWarning)
                method = "eig"
        if rand_type not in _VARIATE_KINDS:
            raise ValueError("rand_type %r is not one of %s" % (rand_type, sorted(_VARIATE_KINDS)))
        modes = M if (num_eig is None or num_eig > M) else max(int(num_eig), 1)
        u = np.random.standard_normal((modes, num_samp)) if rand_vars is None else np.asarray(rand_vars, dtype=float)
        u = _VARIATE_KINDS[rand_type](u)
        root = self._covariance_root(cov + diag_factor * sys.float_info.epsilon * np.eye(M), method, modes, modify_sign)
        return mean[:, None] + root.dot(u[:modes, :])

    def _covariance_root(self, loaded, method, modes, modify_sign):
        """A matrix ``R`` with ``R R^T`` = the (jittered) predictive covariance: its lower Cholesky factor (computed on the
        GPU), or ``Q sqrt(Lambda)`` over the ``modes`` largest eigenpairs, eigenvalues ascending like ``eigh`` returns them
        (ref: gaussian_process.py:1295-1329)."""
        if method == "cholesky":
            # a context of its own: factoring here must not disturb the resident factor of the fit
            if getattr(self, "_scratch_ctx", None) is None:
                self._scratch_ctx = _lib.Context(self.device)
            return np.tril(self._scratch_ctx.potrf_host(loaded))
        if method != "eig":
            raise ValueError("method %r is neither 'cholesky' nor 'eig'" % (method,))
        M = loaded.shape[0]
        lam, Q = scipy.linalg.eigh(loaded, subset_by_index=(M - modes, M - 1))
        if modify_sign is not None:
            # the sign of an eigenvector is arbitrary; these rules pin it by a feature of the vector's first / last entries
            where, feature = (str(modify_sign).split(" ") + [""])[:2]
            if where not in ("left", "right") or feature not in _SIGN_FEATURES:
                raise ValueError("modify_sign %r is not '<left|right> <value|slope|concavity>'" % (modify_sign,))
            ends = Q[:3, :] if where == "left" else Q[-3:, :]
            Q[:, _SIGN_FEATURES[feature](ends, where) < 0.0] *= -1.0
        return Q * np.sqrt(lam)[None, :]

    def _predict_general(self, Xstar, n, noise, need_std):
        """predict for fits that went through ``gpt_fit_matrix`` (``T`` present or a Python kernel):
        covariance blocks from ``compute_Kij``, triangular solves on the GPU."""
        Kstar = self.compute_Kij(self.X, Xstar, self.n, n)
        if noise:
            Kstar = Kstar + self.compute_Kij(self.X, Xstar, self.n, n, noise=True)
        if self.T is not None:
            Kstar = self.T.dot(Kstar)
        mean = Kstar.T.dot(self.alpha).ravel()
        if not need_std:
            return mean, None, None
        v = self._ctx.solve_L(Kstar)
        Kss = self.compute_Kij(Xstar, None, n, None)
        if noise:
            Kss = Kss + self.compute_Kij(Xstar, None, n, None, noise=True)
        covariance = Kss - v.T.dot(v)
        return mean, np.sqrt(np.diagonal(covariance)), covariance


from .kernel.matern import Matern52Kernel as _M52       # noqa: E402
_M52_CALL = _M52.__call__
_ZERO_CALL = ZeroKernel.__call__
