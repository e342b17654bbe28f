"""Rational-quadratic kernel with derivative observations, evaluated on the GPU.

ref: gptools/kernel/rational_quadratic.py:30-164 (RationalQuadraticKernel) through ChainRuleKernel.__call__,
gptools/kernel/core.py:691-816.  Hyperparameters ``[sigma_f, alpha, l_1 .. l_D]``:
``k = sigma_f^2 (1 + 1/(2 alpha) sum_d tau_d^2 / l_d^2)^-alpha``.  Derivative observations follow the reference's
Faa di Bruno sum over set partitions, regrouped on the device (gptools_amd/csrc/kpair.hpp, ``rq_pair``): the
derivative orders of a pair (``ni[m] + nj[m]`` summed over the dimensions) may reach ``GPT_RQ_MAXORD`` = 16, a
``ValueError`` beyond (the reference has no limit, but its cost grows with the Bell numbers: order 12 takes it minutes per pair).  Hyperparameter
derivatives raise ``NotImplementedError`` like the reference (core.py:723-726).
"""
from .core import Kernel
from .. import _lib

__all__ = ["RationalQuadraticKernel"]


class RationalQuadraticKernel(Kernel):
    _gpt_kernel_id = _lib.KERNEL_RQ

    def __init__(self, num_dim=1, **kwargs):
        names = [r"\sigma_f", r"\alpha"] + ["l_{:d}".format(i + 1) for i in range(num_dim)]
        super(RationalQuadraticKernel, self).__init__(num_dim=num_dim, num_params=num_dim + 2, param_names=names,
                                                      **kwargs)
