"""Exceptions of the package (same names as the reference, ref: gptools/error_handling.py:23-31)."""


class GPArgumentError(Exception):
    """An incorrect combination of keyword arguments was given."""


class GPImpossibleParamsError(Exception):
    """The hyperparameters are impossible under the hyperprior."""
