"""Parametric mean functions (host side: an O(N) vector per evaluation).

ref: gptools/mean.py:35-318.  The mean enters the hot path only as ``y - T mu(X, n)`` before the
solve (ref: gptools/gaussian_process.py:1455-1461) and as ``+ mu(Xstar, n)`` after it (ref :972-974).
"""
import inspect

import numpy as np

from .utils import UniformJointPrior, IndependentJointPrior, MaskedBounds, unique_rows

__all__ = ["MeanFunction", "constant", "ConstantMeanFunction", "linear", "LinearMeanFunction"]


class MeanFunction(object):
    """Wrap ``fun(X, n, p1, p2, ..., hyper_deriv=None)`` for use by :class:`GaussianProcess`.

    ``fun`` gets the points sharing one derivative-order vector ``n`` (length D) and returns their
    mean values (ref: gptools/mean.py:180-200).
    """

    def __init__(self, fun, num_params=None, initial_params=None, fixed_params=None, param_bounds=None,
                 param_names=None, enforce_bounds=False, hyperprior=None):
        self.fun = fun
        if num_params is None:
            spec = inspect.getfullargspec(fun)
            nkw = len(spec.defaults) if spec.defaults else 0
            if spec.varargs is None:
                num_params = len(spec.args) - 2 - nkw
            elif hyperprior is not None:
                num_params = len(hyperprior.bounds)
            elif param_names is not None:
                num_params = len(param_names)
            elif param_bounds is not None:
                num_params = len(param_bounds)
            else:
                raise ValueError("If fun uses a variable number of arguments, you must also specify an explicit "
                                 "hyperprior, list of param_names and/or list of param_bounds.")
        elif not isinstance(num_params, (int, np.integer)) or num_params < 0:
            raise ValueError("num_params must be an integer >= 0!")
        self.num_params = int(num_params)
        if param_names is None:
            param_names = [""] * self.num_params
        elif len(param_names) != self.num_params:
            raise ValueError("param_names must be a list of length num_params!")
        self.param_names = np.asarray(param_names, dtype=str)
        self.enforce_bounds = enforce_bounds
        if initial_params is None:
            if fixed_params is not None:
                raise ValueError("Must pass explicit parameter values if fixing parameters!")
            initial_params = np.ones(self.num_params, dtype=float)
            fixed_params = np.zeros(self.num_params, dtype=bool)
        else:
            if len(initial_params) != self.num_params:
                raise ValueError("Length of initial_params must be equal to num_params!")
            if fixed_params is None:
                fixed_params = np.zeros(self.num_params, dtype=bool)
            elif len(fixed_params) != self.num_params:
                raise ValueError("Length of fixed_params must be equal to num_params!")
        if param_bounds is None:
            param_bounds = self.num_params * [(0.0, 1e16)]
        elif len(param_bounds) != self.num_params:
            raise ValueError("Length of param_bounds must be equal to num_params!")
        if hyperprior is None:
            hyperprior = UniformJointPrior(param_bounds)
        elif isinstance(hyperprior, (list, tuple)):
            if len(hyperprior) != self.num_params:
                raise ValueError("If hyperprior is a list its length must be equal to num_params!")
            hyperprior = IndependentJointPrior(hyperprior)
        self.params = np.array(initial_params, dtype=float)
        self.fixed_params = np.asarray(fixed_params, dtype=bool)
        self.hyperprior = hyperprior

    def __call__(self, X, n, hyper_deriv=None):
        n = np.atleast_2d(np.asarray(n, dtype=int))
        X = np.atleast_2d(np.asarray(X, dtype=float))
        mu = np.zeros(X.shape[0])
        for nn in unique_rows(n):
            idxs = (n == nn).all(axis=1)
            mu[idxs] = self.fun(X[idxs, :], nn, *self.params, hyper_deriv=hyper_deriv)
        return mu

    @property
    def param_bounds(self):
        return self.hyperprior.bounds

    @param_bounds.setter
    def param_bounds(self, value):
        self.hyperprior.bounds = value

    def set_hyperparams(self, new_params):
        new_params = np.array(new_params, dtype=float)
        if len(new_params) != len(self.free_params):
            raise ValueError("Length of new_params must be {:d}!".format(len(self.free_params)))
        if self.enforce_bounds:
            for idx, (p, b) in enumerate(zip(new_params, self.free_param_bounds)):
                if b[0] is not None and p < b[0]:
                    new_params[idx] = b[0]
                elif b[1] is not None and p > b[1]:
                    new_params[idx] = b[1]
        self.params[~self.fixed_params] = new_params

    @property
    def num_free_params(self):
        return int(np.sum(~self.fixed_params))

    @property
    def free_param_idxs(self):
        return np.arange(0, self.num_params)[~self.fixed_params]

    @property
    def free_params(self):
        return MaskedBounds(self.params, self.free_param_idxs)

    @free_params.setter
    def free_params(self, value):
        self.params[self.free_param_idxs] = np.asarray(value, dtype=float)

    @property
    def free_param_bounds(self):
        return MaskedBounds(self.hyperprior.bounds, self.free_param_idxs)

    @free_param_bounds.setter
    def free_param_bounds(self, value):
        for i, v in zip(self.free_param_idxs, value):
            self.hyperprior.bounds[i] = v

    @property
    def free_param_names(self):
        return MaskedBounds(self.param_names, self.free_param_idxs)

    @free_param_names.setter
    def free_param_names(self, value):
        self.param_names = np.asarray(self.param_names, dtype=str)
        self.param_names[~self.fixed_params] = value


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
