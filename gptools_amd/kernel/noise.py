"""Noise kernels: homoscedastic diagonal noise and the all-zero default.

ref: gptools/kernel/noise.py:27-152.  ``DiagonalNoiseKernel`` evaluates to
``sigma_n^2 * [Xi == Xj and ni == n and nj == n]`` for ``symmetric`` calls and to zero otherwise;
``GaussianProcess`` special-cases both classes when it assembles ``K_tot`` (ref:
gptools/gaussian_process.py:1434-1437), which the fused K-builder's diagonal epilogue reproduces.
"""
import numpy as np

from .core import Kernel
from .. import _lib

__all__ = ["DiagonalNoiseKernel", "ZeroKernel"]


class DiagonalNoiseKernel(Kernel):
    _gpt_kernel_id = _lib.KERNEL_DIAGNOISE

    def __init__(self, num_dim=1, initial_noise=None, fixed_noise=False, noise_bound=None, n=0, hyperprior=None):
        try:
            iter(n)
        except TypeError:
            self.n = n * np.ones(num_dim, dtype=int)
        else:
            if len(n) != num_dim:
                raise ValueError("Length of n must be equal to num_dim!")
            self.n = np.asarray(n, dtype=int)
        super(DiagonalNoiseKernel, self).__init__(
            num_dim=num_dim, num_params=1,
            initial_params=None if initial_noise is None else [initial_noise],
            fixed_params=[True] if fixed_noise else None,
            param_bounds=None if noise_bound is None else [tuple(noise_bound)],
            hyperprior=hyperprior, param_names=[r"\sigma_n"])


class ZeroKernel(DiagonalNoiseKernel):
    """Always zero; the default ``noise_k`` (ref: gptools/kernel/noise.py:112-152)."""

    _gpt_kernel_id = _lib.KERNEL_ZERO

    def __init__(self, num_dim=1):
        super(ZeroKernel, self).__init__(num_dim=num_dim, initial_noise=0.0, fixed_noise=True)

    def __call__(self, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False):
        return np.zeros(np.atleast_2d(Xi).shape[0], dtype=float)
