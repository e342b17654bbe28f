"""ctypes binding of ``libgpt_hip.so`` (C ABI declared in ``include/gpt_hip.h``).

This is the only route from the Python host code to the GPU.  There is no CPU fallback: if the
shared library is missing or no HIP device is usable, every entry point raises.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpt_hip.so")

GPT_OK = 0
GPT_E_ARG, GPT_E_VALUE, GPT_E_NOTIMPL, GPT_E_HIP, GPT_E_NOMEM, GPT_E_STATE = -1, -2, -3, -4, -5, -6
KERNEL_SE, KERNEL_M52, KERNEL_DIAGNOISE, KERNEL_ZERO, KERNEL_RQ, KERNEL_MATERN, KERNEL_PRODUCT = 0, 1, 2, 3, 4, 5, 6
MAX_DIM = 16

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); mirrors include/gpt_hip.h one to one
SIGNATURES = {
    "gpt_version": (C.c_int, []),
    "gpt_last_error": (C.c_char_p, []),
    "gpt_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "gpt_ctx_destroy": (C.c_int, [_vp]),
    "gpt_ctx_set_option": (C.c_int, [_vp, C.c_char_p, _i64]),
    "gpt_ctx_synchronize": (C.c_int, [_vp]),
    "gpt_ctx_stream": (_vp, [_vp]),
    "gpt_ctx_edge_count": (C.c_int64, [_vp]),
    "gpt_concurrency_hint": (C.c_int, [C.c_int]),
    "gpt_kpairs": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _dp, _ip, _ip, _i64, C.c_int, C.c_int, C.c_int, _ip, _dp]),
    "gpt_kbuild": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _ip, _i64, _dp, _ip, _i64, C.c_int, C.c_int, _ip, _dp]),
    "gpt_kpairs2": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, _ip, _ip, _i64, C.c_int, _dp]),
    "gpt_kbuild2": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int, _dp, _ip, _i64, _dp, _ip, _i64, C.c_int, _dp]),
    "gpt_fit_terms": (C.c_int, [_vp, C.c_int, _ip, _ip, _dp, _ip, _ip, C.c_double, _dp, _dp, C.c_double, _dp, _dp]),
    "gpt_set_data": (C.c_int, [_vp, _dp, _ip, _i64, C.c_int]),
    "gpt_set_T": (C.c_int, [_vp, _dp, _i64]),
    "gpt_fit": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, C.c_double, _dp, _dp]),
    "gpt_fit_sum": (C.c_int, [_vp, C.c_int, _ip, _dp, _ip, C.c_double, _dp, _dp, C.c_double, _dp, _dp]),
    "gpt_fit_batch": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, _dp, C.c_double, _dp, _dp, _ip]),
    "gpt_fit_batch_sum": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _dp, _ip, _dp, _dp, _dp, C.c_double, _dp, _dp, _ip]),
    "gpt_mem_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "gpt_release_batch_scratch": (C.c_int, [_vp]),
    "gpt_fit_batch_terms": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _ip, _dp, _ip, _ip, _dp, _dp, _dp, C.c_double, _dp, _dp, _ip]),
    "gpt_fit_matrix": (C.c_int, [_vp, _dp, _i64, _dp, _dp, _dp]),
    "gpt_get_L": (C.c_int, [_vp, _dp]),
    "gpt_get_alpha": (C.c_int, [_vp, _dp]),
    "gpt_ll_grad": (C.c_int, [_vp, C.c_int, _ip, _ip, _dp]),
    "gpt_predict": (C.c_int, [_vp, _dp, _ip, _i64, C.c_int, _dp, _ip, _dp, _dp, _dp]),
    "gpt_host_alloc": (C.c_int, [_i64, C.POINTER(_vp)]),
    "gpt_host_free": (C.c_int, [_vp]),
    "gpt_cov_sample": (C.c_int, [_vp, _i64, C.c_double, _dp, _i64, _dp]),
    "gpt_solve_L": (C.c_int, [_vp, _dp, _i64]),
    "gpt_cho_solve": (C.c_int, [_vp, _dp, _i64]),
    "gpt_last_timings": (C.c_int, [_vp, _dp, C.c_int]),
    "gpt_gemm_profile_read": (C.c_int, [_vp, _dp]),
    "gpt_gemm_profile_read4": (C.c_int, [_vp, _dp]),
    "gpt_plan_unique_id": (C.c_int, [_vp]),
    "gpt_plan_create": (C.c_int, [C.c_int, C.POINTER(_vp), C.POINTER(_i64), _i64, C.c_int, _vp, _vp, C.c_int, _vp, C.POINTER(_vp)]),
    "gpt_plan_set_comm": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "gpt_plan_set_channel": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "gpt_plan_run": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, C.c_double]),
    "gpt_plan_last_enqueue_ms": (C.c_double, [_vp]),
    "gpt_plan_destroy": (C.c_int, [_vp]),
    "gpt_potrf_host": (C.c_int, [_vp, _dp, _i64]),
    "gpt_gemm_nt_host": (C.c_int, [_vp, _i64, _i64, _i64, C.c_double, _dp, _dp, C.c_double, _dp]),
    "gpt_dev_kbuild": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _vp, _i64, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int,
                                 _ip, C.c_int, _i64, _i64, _vp, C.c_double, C.c_double, _vp, _i64]),
    "gpt_dev_gemm_nt": (C.c_int, [_vp, _i64, _i64, _i64, C.c_double, _vp, _i64, _vp, _i64, C.c_double, _vp, _i64, C.c_int]),
    "gpt_dev_gemm_nt_stair": (C.c_int, [_vp, _i64, _i64, _i64, _i64, C.c_double, _vp, _i64, _vp, _i64, _i64, _i64, C.c_double, _vp, _i64]),
    "gpt_dev_gemm_nt_gridstair": (C.c_int, [_vp, _i64, _i64, _i64, _i64, C.c_double, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64,
                                           C.c_double, _vp, _i64]),
    "gpt_dev_row_sumsq": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gpt_dev_potrf_panel": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64]),
    "gpt_dev_potrf": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "gpt_dev_trsm_rlt": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64]),
    "gpt_dev_trinv": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64]),
    "gpt_dev_copy2d": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, _i64]),
    "gpt_dev_copy2d_on": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64]),
    "gpt_dev_pad_block": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, C.c_double]),
    "gpt_dev_panel_scalars": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
}

_lib = None
_lock = threading.Lock()


class GPTBackendError(RuntimeError):
    """The HIP backend is missing or failed; there is deliberately no CPU fallback."""


def build(verbose=False):
    """Compile ``libgpt_hip.so`` for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise GPTBackendError("building libgpt_hip.so failed")
    return LIB_PATH


def load():
    """Load the shared library and declare every prototype; raises if it is absent."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise GPTBackendError(
                    "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(gptools_amd has no CPU fallback)" % LIB_PATH)
            try:
                # torch wheels bundle their own libamdhip64; it has to be the first HIP runtime in the process,
                # otherwise torch.cuda (streams, RCCL in gptools_amd.dist) stays unavailable afterwards
                import torch  # noqa: F401
            except ImportError:
                pass
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)          # AttributeError if the export is missing
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def last_error():
    return load().gpt_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a C status to the exception the reference would raise at the same place."""
    if rc == GPT_OK:
        return
    msg = last_error()
    if rc > 0:
        raise np.linalg.LinAlgError(msg or "%d-th leading minor of the array is not positive definite" % rc)
    if rc in (GPT_E_ARG, GPT_E_VALUE):
        raise ValueError(msg)
    if rc == GPT_E_NOTIMPL:
        raise NotImplementedError(msg)
    if rc == GPT_E_NOMEM:
        raise MemoryError(msg)
    raise GPTBackendError("libgpt_hip: %s (code %d)" % (msg, rc))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


class _PinnedOwner(object):
    """Returns a gpt_host_alloc block to the pool (or frees it) when the numpy array built over it is collected."""

    def __init__(self, lib, ptr, nbytes):
        self._lib, self._ptr, self._nbytes = lib, ptr, nbytes

    def __del__(self):
        try:
            if self._ptr:
                _pinned_release(self._lib, self._ptr, self._nbytes)
                self._ptr = None
        except Exception:
            pass


# Page-locking memory costs about as much as moving it once through the pageable path (~0.1 ms per MB), so blocks are kept
# for the next result of the same size -- the usual pattern is predict after predict at the same number of points -- up
# to PINNED_POOL_BYTES in total.
PINNED_POOL_BYTES = 2 << 30
_pinned_free = {}
_pinned_held = [0]


def _pinned_release(lib, ptr, nbytes):
    with _lock:
        if _pinned_held[0] + nbytes <= PINNED_POOL_BYTES:
            _pinned_free.setdefault(nbytes, []).append(ptr)
            _pinned_held[0] += nbytes
            return
    lib.gpt_host_free(ptr)


def pinned_empty(shape, min_bytes=1 << 20):
    """``numpy.empty(shape)`` of float64 in page-locked host memory (gpt_host_alloc) when it is at least ``min_bytes`` large
    -- the device writes such an array by asynchronous DMA (the (M, M) covariance of predict) -- else a plain array."""
    n = int(np.prod(shape))
    if n * 8 < min_bytes:
        return np.empty(shape, dtype=np.float64)
    lib = load()
    p = None
    with _lock:
        blocks = _pinned_free.get(n * 8)
        if blocks:
            p = blocks.pop()
            _pinned_held[0] -= n * 8
    if p is None:
        p = _vp()
        check(lib.gpt_host_alloc(n * 8, C.byref(p)))
    buf = (C.c_double * n).from_address(p.value)
    buf._owner = _PinnedOwner(lib, p, n * 8)     # lives as long as the array (numpy keeps `buf` as the array's base)
    return np.ctypeslib.as_array(buf).reshape(shape)


class concurrent_evaluations(object):
    """``with concurrent_evaluations():`` brackets a section in which several contexts of this process evaluate at the same
    time (one host thread each): the library then keeps every look-ahead on event edges (gpt_concurrency_hint)."""

    def __enter__(self):
        load().gpt_concurrency_hint(1)
        return self

    def __exit__(self, *exc):
        load().gpt_concurrency_hint(-1)
        return False


class Context(object):
    """One GPU + stream; owns the device-resident X, n, K_tot/L, invd, alpha."""

    def __init__(self, device=0, stream=None):
        lib = load()
        h = _vp()
        check(lib.gpt_ctx_create(int(device), _vp(stream) if stream else None, C.byref(h)))
        self._h = h
        self._lib = lib
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise GPTBackendError("context already destroyed")
        return self._h

    def set_option(self, key, value):
        check(self._lib.gpt_ctx_set_option(self.handle, key.encode(), int(value)))

    def synchronize(self):
        check(self._lib.gpt_ctx_synchronize(self.handle))

    @property
    def stream(self):
        return self._lib.gpt_ctx_stream(self.handle)

    @property
    def edge_count(self):
        """Flag edges raised so far (does not grow while the look-ahead runs on events: another evaluation in flight in the
        process, a profiler's counter collection, GPT_EDGE_FLAGS=0, n > 12288, after a flag timeout)."""
        return int(self._lib.gpt_ctx_edge_count(self.handle))

    # ---- Kernel.__call__ / compute_Kij -------------------------------------------------------
    def kpairs(self, kernel_id, params, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False, noise_n=None):
        params, Xi, Xj, ni, nj = f64(params), f64(Xi), f64(Xj), i32(ni), i32(nj)
        if Xi.ndim != 2 or Xi.shape != Xj.shape or ni.shape != Xi.shape or nj.shape != Xi.shape:
            raise ValueError("Lengths/widths of Xi, Xj, ni, nj don't match")
        M, D = Xi.shape
        out = np.empty(M, dtype=np.float64)
        nn = None if noise_n is None else i32(noise_n)
        check(self._lib.gpt_kpairs(self.handle, kernel_id, dptr(params), len(params), dptr(Xi), dptr(Xj), iptr(ni),
                                   iptr(nj), M, D, -1 if hyper_deriv is None else int(hyper_deriv),
                                   int(bool(symmetric)), iptr(nn), dptr(out)))
        return out

    def kbuild(self, kernel_id, params, Xi, ni, Xj=None, nj=None, hyper_deriv=None, noise_n=None):
        params, Xi, ni = f64(params), f64(Xi), i32(ni)
        M, D = Xi.shape
        if Xj is None:
            Xj_, nj_, P = None, None, M
        else:
            Xj_, nj_ = f64(Xj), i32(nj)
            P = Xj_.shape[0]
        out = np.empty((M, P), dtype=np.float64)
        nn = None if noise_n is None else i32(noise_n)
        check(self._lib.gpt_kbuild(self.handle, kernel_id, dptr(params), len(params), dptr(Xi), iptr(ni), M,
                                   dptr(Xj_), iptr(nj_), P, D, -1 if hyper_deriv is None else int(hyper_deriv),
                                   iptr(nn), dptr(out)))
        return out

    def kpairs2(self, kid1, params1, kid2, params2, Xi, Xj, ni, nj):
        """Pair list of the product of two native kernels (gpt_kpairs2)."""
        p1, p2, Xi, Xj, ni, nj = f64(params1), f64(params2), f64(Xi), f64(Xj), i32(ni), i32(nj)
        if Xi.ndim != 2 or Xi.shape != Xj.shape or ni.shape != Xi.shape or nj.shape != Xi.shape:
            raise ValueError("Lengths/widths of Xi, Xj, ni, nj don't match")
        M, D = Xi.shape
        out = np.empty(M, dtype=np.float64)
        check(self._lib.gpt_kpairs2(self.handle, kid1, dptr(p1), len(p1), kid2, dptr(p2), len(p2), dptr(Xi), dptr(Xj),
                                    iptr(ni), iptr(nj), M, D, dptr(out)))
        return out

    def kbuild2(self, kid1, params1, kid2, params2, Xi, ni, Xj=None, nj=None):
        """Covariance matrix of the product of two native kernels (gpt_kbuild2)."""
        p1, p2, Xi, ni = f64(params1), f64(params2), f64(Xi), i32(ni)
        M, D = Xi.shape
        if Xj is None:
            Xj_, nj_, P = None, None, M
        else:
            Xj_, nj_ = f64(Xj), i32(nj)
            P = Xj_.shape[0]
        out = np.empty((M, P), dtype=np.float64)
        check(self._lib.gpt_kbuild2(self.handle, kid1, dptr(p1), len(p1), kid2, dptr(p2), len(p2), dptr(Xi), iptr(ni), M,
                                    dptr(Xj_), iptr(nj_), P, D, dptr(out)))
        return out

    def fit_terms(self, terms, noise_var, y, err_y, diag_add):
        """gpt_fit_terms: ``terms`` is a list of ``(kernel_id, params)`` or ``(kernel_id1, params1, kernel_id2, params2)``
        (a product term)."""
        ids = i32(np.asarray([t[0] for t in terms]))
        ids2 = i32(np.asarray([t[2] if len(t) == 4 else -1 for t in terms]))
        npar1 = i32(np.asarray([len(t[1]) for t in terms]))
        npar = i32(np.asarray([len(t[1]) + (len(t[3]) if len(t) == 4 else 0) for t in terms]))
        flat = f64(np.concatenate([np.concatenate([np.asarray(t[1], dtype=float)] +
                                                  ([np.asarray(t[3], dtype=float)] if len(t) == 4 else [])) for t in terms]))
        y, err_y = f64(y), f64(err_y)
        ll = C.c_double()
        ld = C.c_double()
        check(self._lib.gpt_fit_terms(self.handle, len(ids), iptr(ids), iptr(ids2), dptr(flat), iptr(npar), iptr(npar1),
                                      float(noise_var), dptr(y), dptr(err_y), float(diag_add), C.byref(ll), C.byref(ld)))
        return ll.value, ld.value

    # ---- fit / state ---------------------------------------------------------------------------
    def set_data(self, X, n):
        X, n = f64(X), i32(n)
        check(self._lib.gpt_set_data(self.handle, dptr(X), iptr(n), X.shape[0], X.shape[1]))

    def set_T(self, T):
        """Resident linear transform (Ny, N) for the data set given to set_data; ``None`` removes it."""
        if T is None:
            check(self._lib.gpt_set_T(self.handle, None, 0))
            return
        T = f64(T)
        if T.ndim != 2:
            raise ValueError("T must be 2-dimensional")
        check(self._lib.gpt_set_T(self.handle, dptr(T), T.shape[0]))

    def fit(self, kernel_id, params, noise_var, y, err_y, diag_add):
        params, y, err_y = f64(params), f64(y), f64(err_y)
        ll = C.c_double()
        ld = C.c_double()
        check(self._lib.gpt_fit(self.handle, kernel_id, dptr(params), len(params), float(noise_var), dptr(y),
                                dptr(err_y), float(diag_add), C.byref(ll), C.byref(ld)))
        return ll.value, ld.value

    def fit_sum(self, kernel_ids, params_list, noise_var, y, err_y, diag_add):
        """gpt_fit for a sum of native kernels: ``kernel_ids[t]`` with parameters ``params_list[t]``."""
        ids = i32(np.asarray(kernel_ids))
        npar = i32(np.asarray([len(p) for p in params_list]))
        flat = f64(np.concatenate([np.asarray(p, dtype=float) for p in params_list]))
        y, err_y = f64(y), f64(err_y)
        ll = C.c_double()
        ld = C.c_double()
        check(self._lib.gpt_fit_sum(self.handle, len(ids), iptr(ids), dptr(flat), iptr(npar), float(noise_var), dptr(y),
                                    dptr(err_y), float(diag_add), C.byref(ll), C.byref(ld)))
        return ll.value, ld.value

    def fit_batch(self, kernel_id, params, noise_var, y, err_y, diag_add):
        """gpt_fit_batch: ``params`` (B, nparams), ``noise_var`` (B,), ``y`` (B, N), shared ``err_y`` (N,) ->
        ``(ll_data (B,), logdet_half (B,), info (B,) int32)``; ``info[b] > 0``: element b is not positive definite."""
        params, noise_var, y, err_y = f64(np.atleast_2d(params)), f64(noise_var), f64(np.atleast_2d(y)), f64(err_y)
        B = params.shape[0]
        if noise_var.shape != (B,) or y.shape[0] != B or y.shape[1] != err_y.shape[0]:
            raise ValueError("fit_batch: params (B, p), noise_var (B,), y (B, N), err_y (N,) expected")
        ll, ld, info = np.empty(B), np.empty(B), np.zeros(B, dtype=np.int32)
        check(self._lib.gpt_fit_batch(self.handle, B, kernel_id, dptr(params), params.shape[1], dptr(noise_var), dptr(y),
                                      dptr(err_y), float(diag_add), dptr(ll), dptr(ld), iptr(info)))
        return ll, ld, info

    def fit_batch_sum(self, kernel_ids, params, nparams, noise_var, y, err_y, diag_add):
        """gpt_fit_batch_sum: as :meth:`fit_batch` for a sum of native kernels; ``params`` (B, sum(nparams)) holds each
        element's term parameters concatenated."""
        ids, npar = i32(kernel_ids), i32(nparams)
        params, noise_var, y, err_y = f64(np.atleast_2d(params)), f64(noise_var), f64(np.atleast_2d(y)), f64(err_y)
        B = params.shape[0]
        if (noise_var.shape != (B,) or y.shape[0] != B or y.shape[1] != err_y.shape[0] or len(ids) != len(npar)
                or params.shape[1] != int(npar.sum())):
            raise ValueError("fit_batch_sum: params (B, sum(nparams)), noise_var (B,), y (B, N), err_y (N,) expected")
        ll, ld, info = np.empty(B), np.empty(B), np.zeros(B, dtype=np.int32)
        check(self._lib.gpt_fit_batch_sum(self.handle, B, len(ids), iptr(ids), dptr(params), iptr(npar), dptr(noise_var),
                                          dptr(y), dptr(err_y), float(diag_add), dptr(ll), dptr(ld), iptr(info)))
        return ll, ld, info

    def fit_batch_terms(self, terms_list, noise_var, y, err_y, diag_add):
        """gpt_fit_batch_terms: ``terms_list[b]`` is element b's model in the form :meth:`fit_terms` takes (the same kernels in
        every element, only the parameters differ); ``y`` (B, Ny); with a transform set (:meth:`set_T`) K_tot is T (K + noise) T^T."""
        first = terms_list[0]
        ids = i32(np.asarray([t[0] for t in first]))
        ids2 = i32(np.asarray([t[2] if len(t) == 4 else -1 for t in first]))
        npar1 = i32(np.asarray([len(t[1]) for t in first]))
        npar = i32(np.asarray([len(t[1]) + (len(t[3]) if len(t) == 4 else 0) for t in first]))
        params = f64(np.array([np.concatenate([np.concatenate([np.asarray(t[1], dtype=float)] +
                                                              ([np.asarray(t[3], dtype=float)] if len(t) == 4 else []))
                                               for t in terms]) for terms in terms_list]))
        noise_var, y, err_y = f64(noise_var), f64(np.atleast_2d(y)), f64(err_y)
        B = params.shape[0]
        if noise_var.shape != (B,) or y.shape[0] != B or y.shape[1] != err_y.shape[0] or params.shape[1] != int(npar.sum()):
            raise ValueError("fit_batch_terms: one parameter set per element, noise_var (B,), y (B, N), err_y (N,) expected")
        ll, ld, info = np.empty(B), np.empty(B), np.zeros(B, dtype=np.int32)
        check(self._lib.gpt_fit_batch_terms(self.handle, B, len(ids), iptr(ids), iptr(ids2), dptr(params), iptr(npar), iptr(npar1),
                                            dptr(noise_var), dptr(y), dptr(err_y), float(diag_add), dptr(ll), dptr(ld), iptr(info)))
        return ll, ld, info

    def mem_info(self):
        """(free, total) bytes of this context's GPU."""
        f, t = _i64(), _i64()
        check(self._lib.gpt_mem_info(self.handle, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def release_batch_scratch(self):
        check(self._lib.gpt_release_batch_scratch(self.handle))

    def ll_grad(self, term_idx, local_idx):
        """Data-term gradient for the listed (term, parameter) pairs plus the noise trace term (last entry)."""
        ti, li = i32(np.asarray(term_idx)), i32(np.asarray(local_idx))
        out = np.empty(len(ti) + 1)
        check(self._lib.gpt_ll_grad(self.handle, len(ti), iptr(ti), iptr(li), dptr(out)))
        return out

    def fit_matrix(self, K_tot, y):
        K_tot, y = f64(K_tot), f64(y)
        ll = C.c_double()
        ld = C.c_double()
        check(self._lib.gpt_fit_matrix(self.handle, dptr(K_tot), K_tot.shape[0], dptr(y), C.byref(ll), C.byref(ld)))
        return ll.value, ld.value

    def get_L(self, N):
        L = np.empty((N, N), dtype=np.float64)
        check(self._lib.gpt_get_L(self.handle, dptr(L)))
        return L

    def get_alpha(self, N):
        a = np.empty(N, dtype=np.float64)
        check(self._lib.gpt_get_alpha(self.handle, dptr(a)))
        return a

    def predict(self, Xstar, nstar, want, noise_params=None, noise_n=None, device_cov=False):
        """``device_cov`` (with ``want == 2``): the covariance stays on the device (for :meth:`cov_sample`); ``cov`` is None."""
        Xstar, nstar = f64(Xstar), i32(nstar)
        M = Xstar.shape[0]
        mean = np.empty(M)
        std = np.empty(M) if want >= 1 else None
        cov = pinned_empty((M, M)) if (want == 2 and not device_cov) else None
        npar = None if noise_params is None else f64(noise_params)
        nn = None if noise_n is None else i32(noise_n)
        check(self._lib.gpt_predict(self.handle, dptr(Xstar), iptr(nstar), M, int(want), dptr(npar), iptr(nn),
                                    dptr(mean), dptr(std), dptr(cov)))
        return mean, std, cov

    def cov_sample(self, diag_add, rand_vars):
        """``cholesky(cov + diag_add I) @ rand_vars`` for the covariance the last ``predict(..., want=2, device_cov=True)`` left
        on the device; ``rand_vars`` (M, S)."""
        R = f64(np.atleast_2d(rand_vars))
        out = np.empty(R.shape)
        check(self._lib.gpt_cov_sample(self.handle, R.shape[0], float(diag_add), dptr(R), R.shape[1], dptr(out)))
        return out

    def solve_L(self, B):
        B2 = np.array(B, dtype=np.float64, order="C")
        shp = B2.shape
        B2 = np.ascontiguousarray(B2.reshape(shp[0], -1))
        check(self._lib.gpt_solve_L(self.handle, dptr(B2), B2.shape[1]))
        return B2.reshape(shp)

    def cho_solve(self, B):
        B2 = np.array(B, dtype=np.float64, order="C")
        shp = B2.shape
        B2 = np.ascontiguousarray(B2.reshape(shp[0], -1))
        check(self._lib.gpt_cho_solve(self.handle, dptr(B2), B2.shape[1]))
        return B2.reshape(shp)

    def last_timings(self):
        out = np.zeros(5)
        self._lib.gpt_last_timings(self.handle, dptr(out), 5)
        return dict(upload=out[0], kbuild=out[1], potrf=out[2], tail=out[3], total=out[4])

    def gemm_profile_read(self, with_bytes=False):
        """(algorithmic flops, summed launch ms, launches[, algorithmic bytes]) of the profiled GEMM launches since the last read."""
        out = np.zeros(4)
        check(self._lib.gpt_gemm_profile_read4(self.handle, dptr(out)))
        if with_bytes:
            return float(out[0]), float(out[1]), int(out[2]), float(out[3])
        return float(out[0]), float(out[1]), int(out[2])

    def potrf_host(self, A):
        L = np.array(A, dtype=np.float64, order="C")
        check(self._lib.gpt_potrf_host(self.handle, dptr(L), L.shape[0]))
        return L

    def gemm_nt_host(self, alpha, A, B, beta, Cm):
        A, B = f64(A), f64(B)
        Cm = np.array(Cm, dtype=np.float64, order="C")
        check(self._lib.gpt_gemm_nt_host(self.handle, A.shape[0], B.shape[0], A.shape[1], float(alpha), dptr(A),
                                         dptr(B), float(beta), dptr(Cm)))
        return Cm


_default_ctx = {}


def default_context(device=0):
    """Process-wide lazily created context (one per device, re-created after fork)."""
    key = (os.getpid(), int(device))
    ctx = _default_ctx.get(key)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[key] = ctx
    return ctx
