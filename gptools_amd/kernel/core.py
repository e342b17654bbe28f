"""Kernel plugin base class: hyperparameter storage + the ``__call__`` pair-list contract.

ref: gptools/kernel/core.py:44-421 (Kernel), :424-584 (BinaryKernel / SumKernel).

Covariance kernels are evaluated element-wise on a pair list: row m of the four (M, D) inputs
``Xi, Xj, ni, nj`` gives covariance m (ref: gptools/kernel/core.py:220-257).  Kernels that carry a
``_gpt_kernel_id`` are evaluated by the HIP library (``gpt_kpairs`` for the pair list,
``gpt_kbuild`` / ``gpt_fit`` for whole matrices, see include/gpt_hip.h); user subclasses written
in Python keep working through the same interface (``GaussianProcess.compute_Kij`` then feeds them
the tiled pair list exactly like ref: gptools/gaussian_process.py:1591-1603).
"""
import numpy as np

from ..error_handling import GPArgumentError
from .._hyper import HyperparameterSet
from .. import _lib

__all__ = ["ProductKernel", "Kernel", "BinaryKernel", "SumKernel"]


class Kernel(HyperparameterSet):
    """Covariance kernel base class (not meant to be instantiated directly).

    Constructor arguments as in ref: gptools/kernel/core.py:136-210 -- ``num_dim``, ``num_params``,
    ``initial_params``, ``fixed_params``, ``param_bounds``, ``param_names``, ``enforce_bounds``,
    ``hyperprior``; the hyperparameter storage and the free-parameter views (``free_params``,
    ``free_param_bounds``, ``set_hyperparams`` ...) come from :class:`gptools_amd._hyper.HyperparameterSet`.
    Length scales of stationary kernels are the last ``num_dim`` parameters.
    """

    _gpt_kernel_id = None     # set by the kernels the HIP library implements natively

    def __init__(self, num_dim=1, num_params=0, initial_params=None, fixed_params=None, param_bounds=None,
                 param_names=None, enforce_bounds=False, hyperprior=None):
        if isinstance(num_dim, bool) or not isinstance(num_dim, (int, np.integer)) or num_dim < 1:
            raise ValueError("num_dim must be an integer > 0!")
        self.num_dim = int(num_dim)
        self._init_hyperparameters(num_params, values=initial_params, fixed=fixed_params, bounds=param_bounds,
                                   names=param_names, prior=hyperprior, clamp=enforce_bounds,
                                   fixing_error=GPArgumentError, warn_if_unbounded=True)

    # ---- evaluation ----
    def __call__(self, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False):
        """Covariances of the M pairs ``(Xi[m], Xj[m])`` with derivative orders ``(ni[m], nj[m])``.

        Native kernels run ``gpt_kpairs`` on the GPU; the base class itself is abstract
        (ref: gptools/kernel/core.py:254-257).
        """
        if self._gpt_kernel_id is None:
            raise NotImplementedError("This is an abstract method -- please use one of the implementing subclasses!")
        return _lib.default_context().kpairs(
            self._gpt_kernel_id, self.params, np.atleast_2d(Xi), np.atleast_2d(Xj), np.atleast_2d(ni),
            np.atleast_2d(nj), hyper_deriv=hyper_deriv, symmetric=symmetric, noise_n=getattr(self, "n", None))

    def _compute_r2l2(self, tau, return_l=False):
        """Anisotropic ``sum_d tau_d^2 / l_d^2`` with the 0/0 -> 0 rule; a host helper for Python
        plugin kernels (ref: gptools/kernel/core.py:384-421)."""
        tau = np.asarray(tau, dtype=float)
        l_mat = np.tile(self.params[-self.num_dim:], (tau.shape[0], 1))
        with np.errstate(divide="ignore", invalid="ignore"):
            tau_over_l = tau / l_mat
        tau_over_l[(tau == 0) & (l_mat == 0)] = 0.0
        r2l2 = np.sum(tau_over_l ** 2, axis=1)
        return (r2l2, l_mat) if return_l else r2l2

    def __add__(self, other):
        return SumKernel(self, other)

    def __mul__(self, other):
        return ProductKernel(self, other)


class BinaryKernel(Kernel):
    """Two kernels combined; parameters are the concatenation ``k1.params, k2.params``
    (ref: gptools/kernel/core.py:424-546)."""

    def __init__(self, k1, k2):
        if not isinstance(k1, Kernel) or not isinstance(k2, Kernel):
            raise TypeError("Both arguments to BinaryKernel must be instances of type Kernel!")
        if k1.num_dim != k2.num_dim:
            raise ValueError("Only kernels having the same number of dimensions can be summed!")
        self.k1 = k1
        self.k2 = k2
        self._enforce_bounds = k1.enforce_bounds or k2.enforce_bounds
        self.num_dim = k1.num_dim
        self.num_params = k1.num_params + k2.num_params
        self.param_names = np.concatenate((np.asarray(k1.param_names, dtype=str), np.asarray(k2.param_names, dtype=str)))
        self.hyperprior = k1.hyperprior * k2.hyperprior

    @property
    def enforce_bounds(self):
        return self._enforce_bounds

    @enforce_bounds.setter
    def enforce_bounds(self, v):
        self._enforce_bounds = v
        self.k1.enforce_bounds = v
        self.k2.enforce_bounds = v

    @property
    def fixed_params(self):
        return np.concatenate((self.k1.fixed_params, self.k2.fixed_params))

    @fixed_params.setter
    def fixed_params(self, v):
        self.k1.fixed_params = np.asarray(v[:self.k1.num_params], dtype=bool)
        self.k2.fixed_params = np.asarray(v[self.k1.num_params:], dtype=bool)

    @property
    def params(self):
        return np.concatenate((self.k1.params, self.k2.params))

    @params.setter
    def params(self, v):
        self.k1.params = np.asarray(v[:self.k1.num_params], dtype=float)
        self.k2.params = np.asarray(v[self.k1.num_params:], dtype=float)

    @property
    def free_params(self):
        return np.concatenate((self.k1.free_params[:], self.k2.free_params[:]))

    @property
    def free_param_bounds(self):
        return list(self.k1.free_param_bounds[:]) + list(self.k2.free_param_bounds[:])

    @property
    def free_param_names(self):
        return np.concatenate((self.k1.free_param_names[:], self.k2.free_param_names[:]))

    def set_hyperparams(self, new_params):
        new_params = np.asarray(new_params, dtype=float)
        if len(new_params) != len(self.free_params):
            raise ValueError("Length of new_params must be {:d}!".format(len(self.free_params)))
        n1 = self.k1.num_free_params
        self.k1.set_hyperparams(new_params[:n1])
        self.k2.set_hyperparams(new_params[n1:])


class SumKernel(BinaryKernel):
    """``k1 + k2``: each term is evaluated by its own (GPU) kernel and the results are added
    (ref: gptools/kernel/core.py:549-584)."""

    def __call__(self, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False):
        if hyper_deriv is None:
            return (self.k1(Xi, Xj, ni, nj, symmetric=symmetric) +
                    self.k2(Xi, Xj, ni, nj, symmetric=symmetric))
        if hyper_deriv < self.k1.num_params:
            return self.k1(Xi, Xj, ni, nj, hyper_deriv=hyper_deriv, symmetric=symmetric)
        return self.k2(Xi, Xj, ni, nj, hyper_deriv=hyper_deriv - self.k1.num_params, symmetric=symmetric)


class ProductKernel(BinaryKernel):
    """``k1 * k2`` with the product rule for derivative observations (ref: gptools/kernel/core.py:587-671).

    The reference walks the power set of the derivative multiset of every pair (positions distinct, so equal subsets
    recur); grouped by how many of the ``n`` derivatives of each of the 2 D slots (``ni`` then ``nj``) go to ``k1`` that
    is the general Leibniz rule, ``sum_a prod_slots C(n, a) * k1^(a) * k2^(n - a)`` -- the same sum with each distinct
    term evaluated once.  With two native factors (SE, Matern52, RationalQuadratic, Matern) that sum runs per pair on
    the device (``GPT_KERNEL_PRODUCT``: ``gpt_kpairs2`` here, the fused builder in ``GaussianProcess``); otherwise both
    factors are evaluated through their own ``__call__`` and combined on the host."""

    def _native_factors(self):
        """``(kernel_id1, params1, kernel_id2, params2)`` when both factors are kernels the HIP library evaluates itself
        (then the product rule runs per pair on the device, GPT_KERNEL_PRODUCT), else ``None``."""
        from .. import _lib
        from .matern import Matern52Kernel
        ok = (_lib.KERNEL_SE, _lib.KERNEL_M52, _lib.KERNEL_RQ, _lib.KERNEL_MATERN)
        f = []
        for k in (self.k1, self.k2):
            kid = getattr(k, "_gpt_kernel_id", None)
            # (a subclass that overrides __call__ is a Python-defined kernel, whatever id it inherited)
            if kid not in ok or type(k).__call__ not in (Kernel.__call__, Matern52Kernel.__call__):
                return None
            f += [kid, np.array(k.params, dtype=float)]
        return tuple(f)

    def __call__(self, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False):
        if hyper_deriv is not None:
            raise NotImplementedError("hyper_deriv keyword not yet supported!")
        nat = self._native_factors()
        if nat is not None:
            from .. import _lib
            return _lib.default_context().kpairs2(nat[0], nat[1], nat[2], nat[3], np.atleast_2d(np.asarray(Xi, dtype=float)),
                                                  np.atleast_2d(np.asarray(Xj, dtype=float)),
                                                  np.atleast_2d(np.asarray(ni, dtype=int)),
                                                  np.atleast_2d(np.asarray(nj, dtype=int)))
        import itertools
        from math import comb
        Xi, Xj = np.atleast_2d(np.asarray(Xi, dtype=float)), np.atleast_2d(np.asarray(Xj, dtype=float))
        ni, nj = np.atleast_2d(np.asarray(ni, dtype=int)), np.atleast_2d(np.asarray(nj, dtype=int))
        D = self.num_dim
        nij = np.hstack((ni, nj))
        result = np.zeros(Xi.shape[0])
        for row in np.unique(nij, axis=0):
            idxs = (nij == row).all(axis=1)
            cnt = int(idxs.sum())
            xi, xj = Xi[idxs], Xj[idxs]
            for a in itertools.product(*[range(int(r) + 1) for r in row]):
                a = np.asarray(a, dtype=int)
                weight = 1
                for r, ai in zip(row, a):
                    weight *= comb(int(r), int(ai))
                n1 = np.tile(a, (cnt, 1))
                n2 = np.tile(row - a, (cnt, 1))
                result[idxs] += weight * (self.k1(xi, xj, n1[:, :D], n1[:, D:], symmetric=symmetric) *
                                          self.k2(xi, xj, n2[:, :D], n2[:, D:], symmetric=symmetric))
        return result

