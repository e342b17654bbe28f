"""Host-side scalar helpers that sit next to the hot path: hyperpriors and the list views used
for free/fixed hyperparameter bookkeeping.

These stay in Python on purpose: one evaluation costs O(num_params) flops (SURVEY.md section 8,
row a11).  The log-prior is *added to* the log-likelihood (ref: gptools/gaussian_process.py:1469)
and gates impossible parameters (ref: :1387-1389), so the classes keep the reference's call
contract ``prior(theta, hyper_deriv=None) -> float`` plus ``bounds`` / ``random_draw`` / ``*``.
ref: gptools/utils.py:53-1111 (JointPrior family), :137-230 (CombinedBounds, MaskedBounds).
"""
import numpy as np
import scipy.stats

__all__ = [
    "JointPrior", "ProductJointPrior", "UniformJointPrior", "IndependentJointPrior",
    "NormalJointPrior", "LogNormalJointPrior", "GammaJointPrior", "GammaJointPriorAlt",
    "CombinedBounds", "MaskedBounds", "unique_rows",
]


class CombinedBounds(object):
    """Concatenated view over two sequences that writes through to them (ref: gptools/utils.py:137-191)."""

    def __init__(self, l1, l2):
        self.l1 = l1
        self.l2 = l2

    def __getitem__(self, pos):
        return (list(self.l1) + list(self.l2))[pos]

    def __setitem__(self, pos, value):
        n1 = len(self.l1)
        if pos < n1:
            self.l1[pos] = value
        else:
            self.l2[pos - n1] = value

    def __len__(self):
        return len(self.l1) + len(self.l2)

    def __iter__(self):
        return iter(self[:])

    def __invert__(self):
        return ~np.asarray(self[:])

    def __str__(self):
        return str(self[:])

    __repr__ = __str__


class MaskedBounds(object):
    """View of ``a`` restricted to the indices ``m`` (ref: gptools/utils.py:193-230)."""

    def __init__(self, a, m):
        self.a = a
        self.m = m

    def __getitem__(self, pos):
        idx = self.m[pos]
        if isinstance(self.a, np.ndarray):
            return self.a[idx]
        if np.ndim(idx) == 0:
            return self.a[int(idx)]
        return [self.a[int(i)] for i in idx]

    def __setitem__(self, pos, value):
        idx = self.m[pos]
        if isinstance(self.a, np.ndarray) or np.ndim(idx) == 0:
            self.a[idx] = value
        else:
            for i, v in zip(idx, value):
                self.a[int(i)] = v

    def __len__(self):
        return len(self.m)

    def __iter__(self):
        return iter(self[:])

    def __str__(self):
        return str(self[:])

    __repr__ = __str__


def unique_rows(arr):
    """Distinct rows of a 2-D array (ref: gptools/utils.py:1666-1721)."""
    return np.unique(np.atleast_2d(arr), axis=0)


class JointPrior(object):
    """Abstract joint prior over hyperparameters (ref: gptools/utils.py:53-135)."""

    def __init__(self, i=1.0):
        self.i = i

    def __call__(self, theta, hyper_deriv=None):
        raise NotImplementedError("__call__ must be implemented in your own class.")

    def random_draw(self, size=None):
        raise NotImplementedError("random_draw must be implemented in your own class.")

    def __mul__(self, other):
        return ProductJointPrior(self, other)


class ProductJointPrior(JointPrior):
    """Two independent priors side by side; log-densities add (ref: gptools/utils.py:232-349)."""

    def __init__(self, p1, p2):
        if not isinstance(p1, JointPrior) or not isinstance(p2, JointPrior):
            raise TypeError("Both arguments to ProductPrior must be instances of JointPrior!")
        self.p1 = p1
        self.p2 = p2

    @property
    def i(self):
        return min(self.p1.i, self.p2.i)

    @i.setter
    def i(self, v):
        self.p1.i = v
        self.p2.i = v

    @property
    def bounds(self):
        return CombinedBounds(self.p1.bounds, self.p2.bounds)

    @bounds.setter
    def bounds(self, v):
        n1 = len(self.p1.bounds)
        self.p1.bounds = v[:n1]
        self.p2.bounds = v[n1:]

    def __call__(self, theta, hyper_deriv=None):
        n1 = len(self.p1.bounds)
        if hyper_deriv is not None:
            if hyper_deriv < n1:
                return self.p1(theta[:n1], hyper_deriv=hyper_deriv)
            return self.p2(theta[n1:], hyper_deriv=hyper_deriv - n1)
        return self.p1(theta[:n1]) + self.p2(theta[n1:])

    def random_draw(self, size=None):
        d1 = self.p1.random_draw(size=size)
        d2 = self.p2.random_draw(size=size)
        return np.hstack((d1, d2)) if d1.ndim == 1 else np.vstack((d1, d2))


class UniformJointPrior(JointPrior):
    """Independent uniform priors over ``bounds`` (ref: gptools/utils.py:351-456).

    ``UniformJointPrior([(lo, hi), ...])``, ``UniformJointPrior(lb_list, ub=ub_list)`` or the scalar
    form ``UniformJointPrior(lo, hi)`` used by demo/demo.py:111.
    """

    def __init__(self, bounds, ub=None, **kwargs):
        super(UniformJointPrior, self).__init__(**kwargs)
        if ub is not None:
            try:
                bounds = list(zip(bounds, ub))
            except TypeError:
                bounds = [(bounds, ub)]
        self.bounds = [tuple(b) for b in bounds]

    def __call__(self, theta, hyper_deriv=None):
        if hyper_deriv is not None:
            return 0.0
        ll = 0.0
        for v, b in zip(theta, self.bounds):
            if b[0] <= v <= b[1]:
                ll += -np.log(b[1] - b[0])
            else:
                return -np.inf
        return ll

    def random_draw(self, size=None):
        return np.asarray([np.random.uniform(low=b[0], high=b[1], size=size) for b in self.bounds])


class IndependentJointPrior(JointPrior):
    """One univariate prior per hyperparameter: callables of theta or frozen scipy.stats
    distributions (ref: gptools/utils.py:663-765)."""

    def __init__(self, univariate_priors, **kwargs):
        super(IndependentJointPrior, self).__init__(**kwargs)
        self.univariate_priors = list(univariate_priors)

    def __call__(self, theta, hyper_deriv=None):
        if hyper_deriv is not None:
            raise NotImplementedError("Hyperparameter derivatives not supported for IndependentJointPrior!")
        ll = 0.0
        for v, p in zip(theta, self.univariate_priors):
            ll += p.logpdf(v) if hasattr(p, "logpdf") else p(theta)
        return ll

    @property
    def bounds(self):
        return [p.interval(self.i) for p in self.univariate_priors]

    def random_draw(self, size=None):
        return np.asarray([p.rvs(size=size) for p in self.univariate_priors])


class _ScipyFamilyPrior(JointPrior):
    """Shared plumbing for priors that are products of one scipy.stats family."""

    def _dists(self):
        raise NotImplementedError

    def __call__(self, theta, hyper_deriv=None):
        if hyper_deriv is not None:
            return self._dlogpdf(theta, hyper_deriv)
        return float(sum(d.logpdf(v) for v, d in zip(theta, self._dists())))

    @property
    def bounds(self):
        return [d.interval(self.i) for d in self._dists()]

    def random_draw(self, size=None):
        return np.asarray([d.rvs(size=size) for d in self._dists()])


class NormalJointPrior(_ScipyFamilyPrior):
    """Independent normal priors (ref: gptools/utils.py:767-866)."""

    def __init__(self, mu, sigma, **kwargs):
        super(NormalJointPrior, self).__init__(**kwargs)
        self.mu = np.atleast_1d(np.asarray(mu, dtype=float))
        self.sigma = np.atleast_1d(np.asarray(sigma, dtype=float))
        if self.mu.shape != self.sigma.shape or self.mu.ndim != 1:
            raise ValueError("sigma and mu must be one dimensional and have the same shape!")

    def _dists(self):
        return [scipy.stats.norm(loc=m, scale=s) for m, s in zip(self.mu, self.sigma)]

    def _dlogpdf(self, theta, j):
        return (self.mu[j] - theta[j]) / self.sigma[j] ** 2.0


class LogNormalJointPrior(_ScipyFamilyPrior):
    """Independent log-normal priors (ref: gptools/utils.py:868-973)."""

    def __init__(self, mu, sigma, **kwargs):
        super(LogNormalJointPrior, self).__init__(**kwargs)
        self.sigma = np.atleast_1d(np.asarray(sigma, dtype=float))
        self.emu = np.exp(np.atleast_1d(np.asarray(mu, dtype=float)))
        if self.emu.shape != self.sigma.shape or self.emu.ndim != 1:
            raise ValueError("sigma and mu must be one dimensional and have the same shape!")

    def _dists(self):
        return [scipy.stats.lognorm(s, loc=0, scale=em) for s, em in zip(self.sigma, self.emu)]

    def _dlogpdf(self, theta, j):
        return -1.0 / theta[j] * (1.0 + (np.log(theta[j]) - np.log(self.emu[j])) / self.sigma[j] ** 2.0)


class GammaJointPrior(_ScipyFamilyPrior):
    """Independent gamma priors, shape ``a`` and rate ``b`` (ref: gptools/utils.py:975-1079)."""

    def __init__(self, a, b, **kwargs):
        super(GammaJointPrior, self).__init__(**kwargs)
        a = np.atleast_1d(np.asarray(a, dtype=float))
        b = np.atleast_1d(np.asarray(b, dtype=float))
        if a.shape != b.shape or a.ndim != 1:
            raise ValueError("a and b must be one dimensional and have the same shape!")
        self._a = a
        self._b = b

    @property
    def a(self):
        return self._a

    @property
    def b(self):
        return self._b

    def _dists(self):
        return [scipy.stats.gamma(a, loc=0, scale=1.0 / b) for a, b in zip(self.a, self.b)]

    def _dlogpdf(self, theta, j):
        if self.a[j] == 1.0 and theta[j] == 0.0:
            return -self.b[j]
        return (self.a[j] - 1.0) / theta[j] - self.b[j]


class GammaJointPriorAlt(GammaJointPrior):
    """Gamma priors parametrised by mode ``m`` and standard deviation ``s`` (ref: gptools/utils.py:1081-1111)."""

    def __init__(self, m, s, i=1.0):
        JointPrior.__init__(self, i=i)
        self.m = np.atleast_1d(np.asarray(m, dtype=float))
        self.s = np.atleast_1d(np.asarray(s, dtype=float))
        if self.m.shape != self.s.shape or self.m.ndim != 1:
            raise ValueError("s and mu must be one dimensional and have the same shape!")

    @property
    def a(self):
        return 1.0 + self.b * self.m

    @property
    def b(self):
        return (self.m + np.sqrt(self.m ** 2 + 4.0 * self.s ** 2)) / (2.0 * self.s ** 2)
