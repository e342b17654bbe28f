"""Matern kernels evaluated on the GPU: the fixed-order Matern-5/2 kernel and the general-order MaternKernel.

ref: gptools/kernel/matern.py:468-555 (Matern52Kernel), gptools/kernel/_matern.pyx:14-32,
gptools/kernel/src/matern.c:61-186.  Hyperparameters ``[sigma_f, l_1 .. l_D]``.  Each of ``ni[m]``,
``nj[m]`` may contain at most a single 1 (``ValueError`` otherwise, ref matern.py:545-546);
hyperparameter derivatives raise ``NotImplementedError`` (ref matern.py:543-544).
Device code: gptools_amd/csrc/kpair.hpp, ``m52_pair``.
"""
from .core import Kernel
from .. import _lib

__all__ = ["Matern52Kernel", "MaternKernel"]


class Matern52Kernel(Kernel):
    _gpt_kernel_id = _lib.KERNEL_M52

    def __init__(self, num_dim=1, **kwargs):
        names = [r"\sigma_f"] + ["l_{:d}".format(i + 1) for i in range(num_dim)]
        super(Matern52Kernel, self).__init__(num_dim=num_dim, num_params=num_dim + 1, param_names=names, **kwargs)

    def __call__(self, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False):
        if hyper_deriv is not None:
            raise NotImplementedError("Hyperparameter derivatives have not been implemented!")
        return super(Matern52Kernel, self).__call__(Xi, Xj, ni, nj, hyper_deriv=None, symmetric=symmetric)


class MaternKernel(Kernel):
    r"""Matern covariance kernel of general order :math:`\nu`, with derivative observations.

    ref: gptools/kernel/matern.py:251-465 (``MaternKernel``, a ``ChainRuleKernel``: core.py:672-816) and the helpers it
    reaches, gptools/utils.py:1369-1527 (``fixed_poch``, ``Kn2Der``, ``yn2Kn2Der``).  Hyperparameters
    ``[sigma_f, nu, l_1 .. l_D]``:

    .. math::  k = \sigma_f^2 \frac{2^{1-\nu}}{\Gamma(\nu)} y^{\nu/2} K_\nu(\sqrt y),\qquad y = 2\nu\sum_d \tau_d^2/l_d^2 .

    Device code: gptools_amd/csrc/kpair.hpp, ``matern_pair`` -- the modified Bessel function of real order by Temme's
    method, the derivatives from the closed form of ``d^m/dy^m [y^(nu/2) K_nu(sqrt y)]``, the reference's behaviour at and
    near the origin reproduced (one-term series below ``y = 5e-4``, the mean of ``nu -+ 0.001`` for integer ``nu`` there,
    ``0`` / ``NaN`` / ``inf`` at ``y = 0`` term by term).  The derivative orders of a pair may sum to 8 (``ValueError``
    beyond); hyperparameter derivatives raise ``NotImplementedError`` like the reference (core.py:723-726).
    """
    _gpt_kernel_id = _lib.KERNEL_MATERN

    def __init__(self, num_dim=1, **kwargs):
        names = [r"\sigma_f", r"\nu"] + ["l_{:d}".format(i + 1) for i in range(num_dim)]
        super(MaternKernel, self).__init__(num_dim=num_dim, num_params=num_dim + 2, param_names=names, **kwargs)

    @property
    def nu(self):
        r"""The order :math:`\nu` (ref: matern.py:460-464)."""
        return self.params[1]
