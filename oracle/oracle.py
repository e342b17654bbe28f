"""ctypes front-end of the CPU parity oracle (``oracle/gpt_oracle.c``) and of ``oracle/_ref``.

TEST INFRASTRUCTURE ONLY.  Imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by ``gptools_amd`` (the product path).

The functions mirror the reference call sites they restate:
  kpairs   <-> Kernel.__call__(Xi, Xj, ni, nj, hyper_deriv, symmetric)   kernel/core.py:220-257
  kbuild   <-> GaussianProcess.compute_Kij                                gaussian_process.py:1535-1605
  fit      <-> GaussianProcess.compute_K_L_alpha_ll                       gaussian_process.py:1418-1469
  predict  <-> GaussianProcess.predict (non-MCMC branch)                  gaussian_process.py:965-1006
``ref_matern52`` calls the reference's own ``matern52()`` (kernel/src/matern.c:165-186) compiled
into ``oracle/_ref/libmatern52_ref.so``.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SE, M52, DIAGNOISE, ZERO, RQ = 0, 1, 2, 3, 4
MATERN = 5
KERNEL_IDS = {"se": SE, "m52": M52, "diagnoise": DIAGNOISE, "zero": ZERO, "rq": RQ, "matern": MATERN}

_lib = None
_ref = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(HERE, "libgpt_oracle.so")
    src = os.path.join(HERE, "gpt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL)
    elif not os.path.exists(os.path.join(HERE, "_ref", "libmatern52_ref.so")):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        # GPT_ORACLE_LIB: an alternative build of the same source (scripts/asan_cpu.sh points it at the ASan / UBSan one)
        alt = os.environ.get("GPT_ORACLE_LIB")
        if not alt:
            build()
        _lib = C.CDLL(alt or os.path.join(HERE, "libgpt_oracle.so"))
        _lib.orc_eval_hermite.restype = C.c_double
        _lib.orc_eval_hermite.argtypes = [C.c_long, C.c_double]
        _lib.orc_eval_hermitenorm.restype = C.c_double
        _lib.orc_eval_hermitenorm.argtypes = [C.c_long, C.c_double]
        _lib.orc_kpairs.restype = C.c_int
        _lib.orc_kpairs.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp, _ip, _ip, C.c_int64, C.c_int,
                                    C.c_int, C.c_int, _ip, _dp]
        _lib.orc_kbuild.restype = C.c_int
        _lib.orc_kbuild.argtypes = [C.c_int, _dp, C.c_int, _dp, _ip, C.c_int64, _dp, _ip, C.c_int64,
                                    C.c_int, C.c_int, _ip, _dp]
        _lib.orc_potrf_lower.restype = C.c_int64
        _lib.orc_potrf_lower.argtypes = [C.c_int64, _dp, C.c_int64]
        _lib.orc_solve_lower.restype = None
        _lib.orc_solve_lower.argtypes = [C.c_int64, _dp, C.c_int64, C.c_int64, _dp, C.c_int64]
        _lib.orc_solve_lower_t.restype = None
        _lib.orc_solve_lower_t.argtypes = [C.c_int64, _dp, C.c_int64, C.c_int64, _dp, C.c_int64]
        _lib.orc_fit.restype = C.c_int64
        _lib.orc_fit.argtypes = [C.c_int, _dp, C.c_int, C.c_double, _dp, _ip, _dp, _dp, C.c_int64,
                                 C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp]
        _lib.orc_predict.restype = C.c_int
        _lib.orc_predict.argtypes = [C.c_int, _dp, C.c_int, _dp, _ip, _dp, _ip, C.c_int64, C.c_int,
                                     _dp, _dp, _dp, _ip, C.c_int64, _dp, _dp, _dp, _dp]
    return _lib


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "libmatern52_ref.so"))


def ref_lib():
    global _ref
    if _ref is None:
        build()
        _ref = C.CDLL(os.path.join(HERE, "_ref", "libmatern52_ref.so"))
        _ref.matern52.restype = C.c_double
        _ref.matern52.argtypes = [_dp, _dp, _ip, _ip, C.c_int32, _dp]
    return _ref


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t=_dp):
    return None if a is None else a.ctypes.data_as(t)


def _raise(rc):
    if rc == 1:
        raise ValueError("Matern52Kernel only supports 0th and 1st order derivatives")
    if rc == 2:
        raise NotImplementedError("Hyperparameter derivatives have not been implemented!")
    if rc:
        raise RuntimeError("oracle: bad argument (code %d)" % rc)


def eval_hermite(n, x):
    return lib().orc_eval_hermite(int(n), float(x))


def kpairs(kernel, params, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False, noise_n=None):
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else kernel
    params, Xi, Xj, ni, nj = _f(params), _f(Xi), _f(Xj), _i(ni), _i(nj)
    M, D = Xi.shape
    out = np.empty(M)
    nn = None if noise_n is None else _i(noise_n)
    rc = lib().orc_kpairs(kid, _p(params), len(params), _p(Xi), _p(Xj), _p(ni, _ip), _p(nj, _ip), M, D,
                          -1 if hyper_deriv is None else int(hyper_deriv), int(bool(symmetric)),
                          _p(nn, _ip), _p(out))
    _raise(rc)
    return out


def kbuild(kernel, params, Xi, ni, Xj=None, nj=None, hyper_deriv=None, noise_n=None):
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else kernel
    params, Xi, ni = _f(params), _f(Xi), _i(ni)
    M, D = Xi.shape
    if Xj is None:
        P = M
        Xj_, nj_ = None, None
    else:
        Xj_, nj_ = _f(Xj), _i(nj)
        P = Xj_.shape[0]
    out = np.empty((M, P))
    nn = None if noise_n is None else _i(noise_n)
    rc = lib().orc_kbuild(kid, _p(params), len(params), _p(Xi), _p(ni, _ip), M, _p(Xj_), _p(nj_, _ip), P, D,
                          -1 if hyper_deriv is None else int(hyper_deriv), _p(nn, _ip), _p(out))
    _raise(rc)
    return out


def potrf_lower(A):
    L = np.array(A, dtype=np.float64, order="C")
    n = L.shape[0]
    info = lib().orc_potrf_lower(n, _p(L), n)
    if info:
        raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % info)
    return L


def solve_lower(L, B, trans=False):
    L = _f(L)
    B2 = np.array(B, dtype=np.float64, order="C")
    shp = B2.shape
    B2 = B2.reshape(L.shape[0], -1)
    fn = lib().orc_solve_lower_t if trans else lib().orc_solve_lower
    fn(L.shape[0], _p(L), L.shape[0], B2.shape[1], _p(B2), B2.shape[1])
    return B2.reshape(shp)


def fit(kernel, params, X, n, y, err_y, noise_var=0.0, diag_factor=1e2, want_K=False, chol="c", timings=None):
    """compute_K_L_alpha_ll without T; ``y`` already mean-subtracted.

    chol="c": this file's Crout Cholesky; chol="scipy": scipy.linalg (LAPACK, the library the
    reference itself calls at gaussian_process.py:1452,1462) -- used for large N.
    Returns dict(K, L, alpha, ll_data, logdet_half).
    """
    import sys
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else kernel
    params, X, n, y = _f(params), _f(X), _i(n), _f(y)
    N, D = X.shape
    err_y = _f(np.broadcast_to(err_y, (N,)))
    diag_add = diag_factor * sys.float_info.epsilon
    if chol == "c":
        K = np.empty((N, N)) if want_K else None
        L = np.empty((N, N))
        alpha = np.empty(N)
        ll = C.c_double()
        ld = C.c_double()
        info = lib().orc_fit(kid, _p(params), len(params), float(noise_var), _p(X), _p(n, _ip), _p(y),
                             _p(err_y), N, D, diag_add, _p(K), _p(L), _p(alpha), C.byref(ll), C.byref(ld))
        if info < 0:
            _raise(-info)
        if info > 0:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % info)
        return dict(K=K, L=L, alpha=alpha, ll_data=ll.value, logdet_half=ld.value)
    import scipy.linalg
    import time
    t0 = time.perf_counter()
    K = kbuild(kid, params, X, n)
    Kt = K.copy() if want_K else K
    idx = np.arange(N)
    Kt[idx, idx] = ((Kt[idx, idx] + noise_var) + err_y ** 2.0) + diag_add
    t1 = time.perf_counter()
    L = scipy.linalg.cholesky(Kt, lower=True, overwrite_a=not want_K, check_finite=False)
    t2 = time.perf_counter()
    alpha = scipy.linalg.cho_solve((L, True), y)
    ld = np.log(np.diag(L)).sum()
    ll = -0.5 * y.dot(alpha) - ld - 0.5 * N * math.log(2.0 * math.pi)
    if timings is not None:         # (bench.py's cpu_baseline: SURVEY 8d asks for the K-build and potrf times separately)
        timings["kbuild_s"] = timings.get("kbuild_s", 0.0) + (t1 - t0)
        timings["potrf_s"] = timings.get("potrf_s", 0.0) + (t2 - t1)
        timings["solve_ll_s"] = timings.get("solve_ll_s", 0.0) + (time.perf_counter() - t2)
    return dict(K=K if want_K else None, L=L, alpha=alpha, ll_data=float(ll), logdet_half=float(ld))


def predict(kernel, params, X, n, L, alpha, Xs, ns, noise_params=None, noise_n=None, want_cov=True):
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else kernel
    params, X, n, Xs, ns = _f(params), _f(X), _i(n), _f(Xs), _i(ns)
    L, alpha = _f(L), _f(alpha)
    N, D = X.shape
    M = Xs.shape[0]
    mean = np.empty(M)
    std = np.empty(M)
    cov = np.empty((M, M)) if want_cov else None
    work = np.empty(N * M)
    npar = None if noise_params is None else _f(noise_params)
    nn = None if noise_n is None else _i(noise_n)
    rc = lib().orc_predict(kid, _p(params), len(params), _p(npar), _p(nn, _ip), _p(X), _p(n, _ip), N, D,
                           _p(L), _p(alpha), _p(Xs), _p(ns, _ip), M, _p(mean), _p(std), _p(cov), _p(work))
    _raise(rc)
    return mean, std, cov


def ref_matern52(Xi, Xj, ni, nj, var):
    """The reference's own C function, pair by pair (mirrors kernel/_matern.pyx:14-32)."""
    Xi, Xj, ni, nj, var = _f(Xi), _f(Xj), _i(ni), _i(nj), _f(var)
    if (ni.sum(axis=1) > 1).any() or (nj.sum(axis=1) > 1).any():
        raise ValueError("Matern52Kernel only supports 0th and 1st order derivatives")
    M, D = Xi.shape
    out = np.empty(M)
    f = ref_lib().matern52
    for m in range(M):
        out[m] = f(_p(Xi[m]), _p(Xj[m]), _p(ni[m], _ip), _p(nj[m], _ip), D, _p(var))
    return out
