"""Parametric mean functions (host side: an O(N) vector per evaluation).

ref: gptools/mean.py:35-318.  The mean enters the hot path only as ``y - T mu(X, n)`` before the
solve (ref: gptools/gaussian_process.py:1455-1461) and as ``+ mu(Xstar, n)`` after it (ref :972-974).
"""
import inspect

import numpy as np

from ._hyper import HyperparameterSet

__all__ = ["MeanFunction", "constant", "ConstantMeanFunction", "linear", "LinearMeanFunction"]


def _count_parameters(fun, hyperprior, param_names, param_bounds):
    """How many hyperparameters ``fun(X, n, p1, ..., hyper_deriv=None)`` takes: its positional arguments behind
    ``X, n`` that have no default; a ``*args`` signature says nothing, so one of the per-parameter keywords must."""
    spec = inspect.getfullargspec(fun)
    if spec.varargs is None:
        return len(spec.args) - 2 - len(spec.defaults or ())
    for sized in (getattr(hyperprior, "bounds", None), param_names, param_bounds):
        if sized is not None:
            return len(sized)
    raise ValueError("fun takes *args: give num_params, or a hyperprior / param_names / param_bounds to count them by.")


class MeanFunction(HyperparameterSet):
    """Wrap ``fun(X, n, p1, p2, ..., hyper_deriv=None)`` for use by :class:`GaussianProcess`.

    ``fun`` gets the points sharing one derivative-order vector ``n`` (length D) and returns their
    mean values (ref: gptools/mean.py:35-200).  Hyperparameter storage and views:
    :class:`gptools_amd._hyper.HyperparameterSet`.
    """

    def __init__(self, fun, num_params=None, initial_params=None, fixed_params=None, param_bounds=None,
                 param_names=None, enforce_bounds=False, hyperprior=None):
        self.fun = fun
        if num_params is None:
            num_params = _count_parameters(fun, hyperprior, param_names, param_bounds)
        self._init_hyperparameters(num_params, values=initial_params, fixed=fixed_params, bounds=param_bounds,
                                   names=param_names, prior=hyperprior, clamp=enforce_bounds)

    def __call__(self, X, n, hyper_deriv=None):
        X = np.atleast_2d(np.asarray(X, dtype=float))
        n = np.atleast_2d(np.asarray(n, dtype=int))
        out = np.zeros(X.shape[0])
        # one call of fun per distinct derivative-order vector among the rows
        orders, group = np.unique(n, axis=0, return_inverse=True)
        group = np.asarray(group).ravel()
        for g, order in enumerate(orders):
            rows = group == g
            out[rows] = self.fun(X[rows], order, *self.params, hyper_deriv=hyper_deriv)
        return out


def constant(X, n, mu, hyper_deriv=None):
    """Constant mean (ref: gptools/mean.py:293-302)."""
    if (np.asarray(n) == 0).all():
        return np.ones(X.shape[0]) if hyper_deriv is not None else mu * np.ones(X.shape[0])
    return np.zeros(X.shape[0])


class ConstantMeanFunction(MeanFunction):
    """Constant mean with a uniform [-1e3, 1e3] default prior (ref: gptools/mean.py:304-318)."""

    def __init__(self, **kwargs):
        if "hyperprior" not in kwargs and "param_bounds" not in kwargs:
            kwargs["param_bounds"] = [(-1e3, 1e3)]
        super(ConstantMeanFunction, self).__init__(constant, param_names=["\\mu"], **kwargs)


def linear(X, n, *args, **kwargs):
    """Linear mean ``b + sum_d m_d x_d`` with args ``(m_1..m_D, b)`` (ref: gptools/mean.py:446-484)."""
    hyper_deriv = kwargs.pop("hyper_deriv", None)
    m = np.asarray(args[:-1], dtype=float)
    b = args[-1]
    n = np.asarray(n, dtype=int)
    if n.sum() > 1:
        return np.zeros(X.shape[0])
    if n.sum() == 0:
        if hyper_deriv is not None:
            return np.ones(X.shape[0]) if hyper_deriv == len(m) else X[:, hyper_deriv]
        return X.dot(m) + b
    d = int(np.argmax(n))
    if hyper_deriv is not None:
        return np.ones(X.shape[0]) if hyper_deriv == d else np.zeros(X.shape[0])
    return m[d] * np.ones(X.shape[0])


class LinearMeanFunction(MeanFunction):
    """Linear mean function (ref: gptools/mean.py:486-505)."""

    def __init__(self, num_dim=1, **kwargs):
        names = ["m{:d}".format(i) for i in range(num_dim)] + ["b"]
        if "hyperprior" not in kwargs and "param_bounds" not in kwargs:
            kwargs["param_bounds"] = [(-1e3, 1e3)] * (num_dim + 1)
        super(LinearMeanFunction, self).__init__(linear, num_params=num_dim + 1, param_names=names, **kwargs)
