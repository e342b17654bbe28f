"""Matern-5/2 kernel with value / first-derivative observations, evaluated on the GPU.

ref: gptools/kernel/matern.py:468-555 (Matern52Kernel), gptools/kernel/_matern.pyx:14-32,
gptools/kernel/src/matern.c:61-186.  Hyperparameters ``[sigma_f, l_1 .. l_D]``.  Each of ``ni[m]``,
``nj[m]`` may contain at most a single 1 (``ValueError`` otherwise, ref matern.py:545-546);
hyperparameter derivatives raise ``NotImplementedError`` (ref matern.py:543-544).
Device code: gptools_amd/csrc/kpair.hpp, ``m52_pair``.
"""
from .core import Kernel
from .. import _lib

__all__ = ["Matern52Kernel"]


class Matern52Kernel(Kernel):
    _gpt_kernel_id = _lib.KERNEL_M52

    def __init__(self, num_dim=1, **kwargs):
        names = [r"\sigma_f"] + ["l_{:d}".format(i + 1) for i in range(num_dim)]
        super(Matern52Kernel, self).__init__(num_dim=num_dim, num_params=num_dim + 1, param_names=names, **kwargs)

    def __call__(self, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False):
        if hyper_deriv is not None:
            raise NotImplementedError("Hyperparameter derivatives have not been implemented!")
        return super(Matern52Kernel, self).__call__(Xi, Xj, ni, nj, hyper_deriv=None, symmetric=symmetric)
