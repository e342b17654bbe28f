"""Hyperparameter bookkeeping shared by covariance kernels and mean functions (host side).

Both kinds of object carry a vector of hyperparameters, a mask of the ones held fixed, names, and a
joint prior whose ``bounds`` double as the box constraints of the optimiser; the GP concatenates the
vectors of its parts (``GaussianProcess.params`` etc.).  The behaviour follows the reference's
``Kernel`` / ``MeanFunction`` constructors and views (ref: gptools/kernel/core.py:136-352,
gptools/mean.py:95-291) -- which argument combinations are accepted, the defaults (ones, nothing
fixed, uniform over (0, 1e16)), which exception type a violation raises, clamping in
``set_hyperparams`` -- written once here instead of once per class.
"""
import warnings

import numpy as np

from .utils import IndependentJointPrior, MaskedBounds, UniformJointPrior

WIDE_OPEN = (0.0, 1e16)          # default box of a hyperparameter nobody gave bounds for


def _expect_len(seq, count, what):
    if len(seq) != count:
        raise ValueError("%s has %d entries, but there are %d hyperparameters" % (what, len(seq), count))
    return seq


class HyperparameterSet(object):
    """Mixin: ``params``, ``fixed_params``, ``param_names``, ``hyperprior``, ``enforce_bounds`` and the
    free-parameter views over them."""

    def _init_hyperparameters(self, count, values=None, fixed=None, bounds=None, names=None, prior=None,
                              clamp=False, fixing_error=ValueError, warn_if_unbounded=False):
        if isinstance(count, bool) or not isinstance(count, (int, np.integer)) or count < 0:
            raise ValueError("num_params must be an integer >= 0!")
        count = int(count)
        self.num_params = count
        self.enforce_bounds = clamp
        self.param_names = np.asarray([""] * count if names is None
                                      else _expect_len(names, count, "param_names"), dtype=str)
        if values is None:
            # nothing to fix a parameter AT: the reference refuses the combination too
            if fixed is not None:
                raise fixing_error("fixed_params needs initial_params: a fixed hyperparameter keeps its initial value")
            values, fixed = np.ones(count), np.zeros(count, dtype=bool)
        else:
            _expect_len(values, count, "initial_params")
            fixed = np.zeros(count, dtype=bool) if fixed is None else _expect_len(fixed, count, "fixed_params")
        self.fixed_params = np.asarray(fixed, dtype=bool)
        self.params = np.array(values, dtype=float)
        if prior is None:
            if bounds is None:
                if warn_if_unbounded and not self.fixed_params.all():
                    warnings.warn("Neither param_bounds nor hyperprior given: free hyperparameters default to a uniform "
                                  "prior over (0, 1e16), which is unlikely to suit the data.")
                bounds = [WIDE_OPEN] * count
            prior = UniformJointPrior(_expect_len(bounds, count, "param_bounds"))
        else:
            if bounds is not None:
                _expect_len(bounds, count, "param_bounds")
            if isinstance(prior, (list, tuple)) or not callable(prior):
                prior = IndependentJointPrior(_expect_len(prior, count, "hyperprior (as a list of univariate priors)"))
        self.hyperprior = prior

    # ---- box constraints live in the prior -------------------------------------------------------------------
    @property
    def param_bounds(self):
        return self.hyperprior.bounds

    @param_bounds.setter
    def param_bounds(self, value):
        self.hyperprior.bounds = value

    # ---- the free subset -------------------------------------------------------------------------------------
    @property
    def free_param_idxs(self):
        return np.flatnonzero(~np.asarray(self.fixed_params, dtype=bool))

    @property
    def num_free_params(self):
        return int(self.free_param_idxs.size)

    def _free(self, target):
        return MaskedBounds(target, self.free_param_idxs)

    @property
    def free_params(self):
        return self._free(self.params)

    @free_params.setter
    def free_params(self, value):
        self.params[self.free_param_idxs] = np.asarray(value, dtype=float)

    @property
    def free_param_bounds(self):
        return self._free(self.hyperprior.bounds)

    @free_param_bounds.setter
    def free_param_bounds(self, value):
        box = self.hyperprior.bounds
        for where, pair in zip(self.free_param_idxs, value):
            box[where] = pair

    @property
    def free_param_names(self):
        return self._free(self.param_names)

    @free_param_names.setter
    def free_param_names(self, value):
        names = np.asarray(self.param_names, dtype=str)
        names[self.free_param_idxs] = value
        self.param_names = names

    def set_hyperparams(self, new_params):
        """Assign the FREE hyperparameters (what an optimiser varies).  With ``enforce_bounds`` a value outside its box
        is moved onto the nearest edge first (ref: gptools/kernel/core.py:259-287); an edge given as ``None`` is open."""
        new = np.array(new_params, dtype=float).ravel()
        idx = self.free_param_idxs
        if new.size != idx.size:
            raise ValueError("%d values given for %d free hyperparameters" % (new.size, idx.size))
        if self.enforce_bounds and idx.size:
            box = list(self.free_param_bounds)
            lo = np.array([-np.inf if b[0] is None else b[0] for b in box], dtype=float)
            hi = np.array([np.inf if b[1] is None else b[1] for b in box], dtype=float)
            new = np.where(new < lo, lo, np.where(new > hi, hi, new))      # (comparisons with NaN edges leave the value alone)
        self.params[idx] = new
