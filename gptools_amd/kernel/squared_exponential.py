"""Anisotropic squared-exponential kernel with arbitrary-order derivatives, evaluated on the GPU.

ref: gptools/kernel/squared_exponential.py:31-174.  Hyperparameters ``[sigma_f, l_1 .. l_D]``:
``k = sigma_f^2 exp(-1/2 sum_d tau_d^2 / l_d^2)``; derivative observations bring in Hermite
polynomials (device code: gptools_amd/csrc/kpair.hpp, ``se_pair``).  ``hyper_deriv`` (derivative
with respect to one hyperparameter) is supported like the reference (ref :133-174).
"""
from .core import Kernel
from .. import _lib

__all__ = ["SquaredExponentialKernel"]


class SquaredExponentialKernel(Kernel):
    _gpt_kernel_id = _lib.KERNEL_SE

    def __init__(self, num_dim=1, **kwargs):
        names = [r"\sigma_f"] + ["l_{:d}".format(i + 1) for i in range(num_dim)]
        super(SquaredExponentialKernel, self).__init__(num_dim=num_dim, num_params=num_dim + 1,
                                                       param_names=names, **kwargs)
