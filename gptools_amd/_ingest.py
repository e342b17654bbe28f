"""Normalisation of what users hand to ``GaussianProcess.add_data`` / ``predict``: the host data layout the C ABI
consumes (``gpt_set_data``: X (N, D) float64 row-major, n (N, D) integer orders; ``gpt_fit``: y, err_y (N,)).

Accepted forms follow the reference (ref: gptools/gaussian_process.py:376-503 for training data, :913-963 for test
points): scalars broadcast, a single row of 1-D points may come as a row vector, every violation is a ``ValueError``.
"""
import numpy as np


def _is_scalar(v):
    return np.ndim(v) == 0


def targets(y):
    """Observed values -> (M,) float64."""
    y = np.atleast_1d(np.asarray(y, dtype=float))
    if y.ndim != 1:
        raise ValueError("y must be a vector of observations; got an array of shape %s" % (y.shape,))
    return y


def noise_levels(err_y, like):
    """Standard deviations of the observations -> array shaped like ``like``; a scalar applies to all of them."""
    if _is_scalar(err_y):
        err = np.full(like.shape, float(err_y))
    else:
        err = np.asarray(err_y, dtype=float)
        if err.shape != like.shape:
            raise ValueError("err_y has shape %s, one standard deviation per observation would be %s"
                             % (err.shape, like.shape))
    if np.any(err < 0):
        raise ValueError("err_y holds a negative standard deviation")
    return err


def points(X, num_dim, what="X"):
    """Input locations -> (M, num_dim) float64.  For 1-D inputs a single row of M values means M points."""
    X = np.atleast_2d(np.asarray(X, dtype=float))
    if num_dim == 1 and X.shape[0] == 1:
        X = X.T
    if X.ndim != 2 or X.shape[1] != num_dim:
        raise ValueError("%s must have one row per point and num_dim = %d columns; got shape %s" % (what, num_dim, X.shape))
    return X


def derivative_orders(n, like, num_dim, what="n", column_rule="row"):
    """Derivative orders -> integer array shaped like the points ``like``; a scalar applies to every point and dimension.
    1-D inputs: ``column_rule`` = "row" turns a single row into a column (predict), "not-column" transposes anything that
    is not already a column (add_data) -- the two call sites of the reference differ in exactly this."""
    if _is_scalar(n):
        out = np.full(like.shape, int(n), dtype=int)
    else:
        out = np.atleast_2d(np.asarray(n, dtype=int))
        if num_dim == 1 and ((column_rule == "row" and out.shape[0] == 1) or
                             (column_rule == "not-column" and out.shape[1] != 1)):
            out = out.T
        if out.shape != like.shape:
            raise ValueError("%s has shape %s, the points it describes have shape %s" % (what, out.shape, like.shape))
    if np.any(out < 0):
        raise ValueError("%s holds a negative derivative order" % what)
    return out


def linear_map(T, rows, cols, what="T"):
    """A dense (rows, cols) float64 matrix (the transform of line-integrated observations; predict's output_transform)."""
    T = np.atleast_2d(np.asarray(T, dtype=float))
    if T.ndim != 2:
        raise ValueError("%s must be a matrix; got %d dimensions" % (what, T.ndim))
    if rows is not None and T.shape[0] != rows:
        raise ValueError("%s has %d rows, expected %d" % (what, T.shape[0], rows))
    if cols is not None and T.shape[1] != cols:
        raise ValueError("%s has %d columns, expected %d" % (what, T.shape[1], cols))
    return T
