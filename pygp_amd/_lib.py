"""
ctypes binding of libgpx.so (include/gpx.h). This is the only place the Python
side touches native code; there is no CPU fallback: if the library or an
MI355X is missing, the product path raises.
"""

import ctypes as C
import os
import threading

import numpy as np

__all__ = ['lib', 'Handle', 'default_handle', 'kspec_of', 'GpxError', 'device_count',
           'loglik_batch_multi', 'batch_partition',
           'KIND_SE', 'KIND_MATERN', 'KIND_PERIODIC', 'KIND_SUM', 'KIND_RQ', 'KIND_PRODUCT',
           'F64',
           'F32']

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, 'libgpx.so')

KIND_SE = 1
KIND_MATERN = {1: 2, 3: 3, 5: 4}
KIND_PERIODIC = 5
KIND_SUM = 6
KIND_RQ = 7
KIND_PRODUCT = 8
F64, F32 = 0, 1
NTIMERS = 10


class GpxError(RuntimeError):
    pass


class _KSpec(C.Structure):
    pass


_KSpec._fields_ = [
    ('kind', C.c_int32), ('iso', C.c_int32), ('ndim', C.c_int32),
    ('nhyper', C.c_int32), ('hyper', C.POINTER(C.c_double)),
    ('nparts', C.c_int32), ('parts', C.POINTER(_KSpec)),
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); mirrors include/gpx.h one to one
SIGNATURES = {
    'gpx_version': (C.c_int, []),
    'gpx_last_error': (C.c_char_p, []),
    'gpx_device_count': (C.c_int, [_ip]),
    'gpx_create': (C.c_int, [C.c_int, C.POINTER(_vp)]),
    'gpx_destroy': (C.c_int, [_vp]),
    'gpx_synchronize': (C.c_int, [_vp]),
    'gpx_kernel_get': (C.c_int, [_vp, C.POINTER(_KSpec), _vp, _i64, _vp, _i64,
                                 _i64, C.c_int, _vp]),
    'gpx_kernel_grad': (C.c_int, [_vp, C.POINTER(_KSpec), _vp, _i64, _vp, _i64,
                                  _i64, _vp]),
    'gpx_kernel_build_resident': (C.c_int, [_vp, C.POINTER(_KSpec), C.c_int,
                                            C.c_int, _dp]),
    'gpx_set_data': (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    'gpx_exact_update': (C.c_int, [_vp, C.POINTER(_KSpec), C.c_double,
                                   C.c_double, _ip]),
    'gpx_exact_append': (C.c_int, [_vp, _vp, _vp, _i64, _ip]),
    'gpx_exact_loglik': (C.c_int, [_vp, _dp, _vp]),
    'gpx_exact_eval': (C.c_int, [_vp, C.POINTER(_KSpec), C.c_double, C.c_double,
                                 C.c_int, _dp, _vp, _ip]),
    'gpx_exact_posterior': (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    'gpx_exact_posterior_grad': (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    'gpx_exact_posterior_full': (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    'gpx_kernel_gradx': (C.c_int, [_vp, C.POINTER(_KSpec), _vp, _i64, _vp, _i64, _i64,
                                   C.c_int, _vp]),
    'gpx_exact_get_factor': (C.c_int, [_vp, _i64, _vp, _vp]),
    'gpx_loglik_batch': (C.c_int, [_vp, C.POINTER(_KSpec), _vp, _i64, C.c_int,
                                   _vp, _vp, _vp]),
    'gpx_batch_plan': (C.c_int, [_vp, _i64, C.c_int, _ip]),
    'gpx_loglik_batch_multi': (C.c_int, [C.POINTER(_KSpec), _vp, _i64, _vp, _vp, _i64, _i64,
                                         C.c_int, C.c_int, _vp, _vp, _vp]),
    'gpx_posterior_batch_multi': (C.c_int, [C.POINTER(_KSpec), _vp, _i64, _vp, _vp, _i64, _i64,
                                            _vp, _i64, C.c_int, C.c_int, _vp, _vp, _vp, _vp,
                                            _vp]),
    'gpx_batch_partition': (None, [_i64, C.c_int, C.c_int, C.POINTER(_i64),
                                   C.POINTER(_i64)]),
    'gpx_multi_slot': (_i64, [_i64, C.c_int]),
    'gpx_multi_comm_size': (C.c_int, []),
    'gpx_multi_pack': (C.c_int, [_vp, _i64, C.c_int, _i64, _vp]),
    'gpx_multi_scatter': (C.c_int, [_vp, _i64, C.c_int, C.c_int, _vp]),
    'gpx_posterior_batch': (C.c_int, [_vp, C.POINTER(_KSpec), _vp, _i64, _vp, _i64, _vp,
                                      _vp, _vp, _vp, _vp]),
    'gpx_enable_timing': (C.c_int, [_vp, C.c_int]),
    'gpx_get_timings': (C.c_int, [_vp, _vp, C.c_int]),
    'gpx_batch_timings': (C.c_int, [_vp, _dp, C.POINTER(_i64)]),
    'gpx_timing_name': (C.c_char_p, [C.c_int]),
    'gpx_la_gemm': (C.c_int, [_vp, C.c_int, C.c_int, _i64, _i64, _i64,
                              C.c_double, _vp, _i64, _vp, _i64, C.c_double, _vp,
                              _i64]),
    'gpx_set_safe_mode': (C.c_int, [_vp, C.c_int]),
    'gpx_get_safe_mode': (C.c_int, [_vp, _ip]),
    'gpx_multi_enable_timing': (C.c_int, [C.c_int, C.c_int]),
    'gpx_multi_batch_info': (C.c_int, [C.c_int, _i64, C.c_int, _dp, C.POINTER(_i64), _ip]),
    'gpx_panel_grid_check': (C.c_int, [C.c_int, C.c_int, _ip, _ip]),
    'gpx_la_potrf': (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _ip]),
    'gpx_la_gemm_bench': (C.c_int, [_vp, C.c_int, C.c_int, _i64, C.c_int, _dp]),
    'gpx_la_gemm_bench_ex': (C.c_int, [_vp, C.c_int, C.c_int, _i64, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       _dp]),
    'gpx_la_gemm_bench_mnk': (C.c_int, [_vp, C.c_int, C.c_int, _i64, _i64, _i64, C.c_int,
                                        C.c_double, C.c_int, C.c_int, _dp]),
    'gpx_la_potrf_bench': (C.c_int, [_vp, _i64, C.c_int, C.c_int, _dp]),
    'gpx_panel_graph_check': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'gpx_panel_graph_check_wide': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'gpx_panel_graph_check_rhs': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'gpx_panel_graph_check_full': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'gpx_panel_solo_check': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'gpx_sweep_check': (C.c_int, [C.c_int, C.c_int]),
    'gpx_sweep_check_lite': (C.c_int, [C.c_int, C.c_int, C.c_int]),
}

_lib = None
_lock = threading.RLock()


def lib():
    """Load libgpx.so (once). Raises GpxError if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(_LIBPATH):
                    raise GpxError(
                        'pygp_amd/libgpx.so is missing: build it with '
                        '`python -m pygp_amd.build` (there is no CPU fallback)')
                L = C.CDLL(_LIBPATH)
                for name, (res, args) in SIGNATURES.items():
                    f = getattr(L, name)
                    f.restype, f.argtypes = res, args
                _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _f64(a, ndim=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise ValueError('expected a %d-d array' % ndim)
    return a


def check(code):
    """Map a C return code to the reference's exceptions: >0 is the LinAlgError
    scipy.linalg.cholesky raises at pygp/inference/exact.py:54."""
    if code == 0:
        return
    msg = lib().gpx_last_error().decode('utf-8', 'replace')
    if code > 0:
        raise np.linalg.LinAlgError(
            '%d-th leading minor of the array is not positive definite' % code)
    raise GpxError(msg or 'libgpx error %d' % code)


class KSpecHolder(object):
    """Owns the ctypes structs + hyper arrays of one kernel description."""

    def __init__(self, kind, iso, ndim, hyper=None, parts=()):
        self.parts = list(parts)
        self.hyper = None if hyper is None else _f64(np.atleast_1d(hyper))
        self.c = _KSpec()
        self.c.kind, self.c.iso, self.c.ndim = kind, int(bool(iso)), int(ndim)
        if self.parts:
            self.arr = (_KSpec * len(self.parts))(*[p.c for p in self.parts])
            self.c.nparts = len(self.parts)
            self.c.parts = C.cast(self.arr, C.POINTER(_KSpec))
            self.c.nhyper = sum(p.c.nhyper for p in self.parts)
            self.c.hyper = None
        else:
            self.c.nparts = 0
            self.c.parts = None
            self.c.nhyper = self.hyper.size
            self.c.hyper = self.hyper.ctypes.data_as(_dp)

    def ref(self):
        return C.byref(self.c)


def kspec_of(kernel):
    """Ask a pygp_amd kernel object for its C description."""
    return kernel._kspec()


def _serialised(method):
    """Calls on one handle must not interleave (include/gpx.h): ctypes drops the GIL
    during a call, so two Python threads sharing a handle -- the process-wide
    default handle behind Kernel.get() in particular -- take turns.

    A task-queue launch that runs into its wait bound ("... timed out waiting ...") is an
    ERROR by default: within one process the launches are ordered on the device, so the
    bound is only met when another process uses the same GPU -- or by a scheduling bug, which
    must not hide behind a retry. Only a handle made with auto_safe_mode=True (or
    GPX_AUTO_SAFE_MODE=1) switches to safe mode -- no kernel that waits for another
    workgroup; another order of arithmetic, same tolerances -- and repeats the call, once
    per handle, with a RuntimeWarning that carries the original error; `safe_mode` and
    `safe_mode_switches` on the handle say that it happened."""
    def call(self, *args, **kwargs):
        with self._lock:
            try:
                return method(self, *args, **kwargs)
            except GpxError as e:
                if 'timed out waiting' not in str(e) or not getattr(self, '_auto_safe', False) \
                        or getattr(self, 'safe_mode_switches', 0) or not getattr(self, '_h', None):
                    raise
                import warnings
                warnings.warn('pygp_amd: %s (is another process using this GPU?); this handle '
                              'continues in safe mode: same tolerances, not the same bits' % e,
                              RuntimeWarning)
                self.safe_mode_switches = 1
                check(self._L.gpx_set_safe_mode(self._h, 1))
                return method(self, *args, **kwargs)
    call.__name__ = method.__name__
    call.__doc__ = method.__doc__
    return call


class Handle(object):
    """One device context (gpx_t*): a GPU, a stream and its HBM buffers."""

    def __init__(self, device=None, auto_safe_mode=None):
        """auto_safe_mode: switch this handle to safe mode and repeat the call when a
        diagonal-block launch runs into its wait bound (see _serialised); default: the
        environment's GPX_AUTO_SAFE_MODE, else off -- the error is raised."""
        L = lib()
        if auto_safe_mode is None:
            auto_safe_mode = os.environ.get('GPX_AUTO_SAFE_MODE', '0') not in ('', '0')
        self._auto_safe = bool(auto_safe_mode)
        self.safe_mode_switches = 0          # automatic switches so far (0 or 1)
        if device is None:
            device = int(os.environ.get('GPX_DEVICE',
                                        os.environ.get('LOCAL_RANK', '0')))
            n = C.c_int(0)
            check(L.gpx_device_count(C.byref(n)))
            if n.value > 0:
                device %= n.value
        self.device = device
        h = _vp()
        check(L.gpx_create(device, C.byref(h)))
        self._h = h
        self._L = L
        self._lock = threading.RLock()

    @property
    def safe_mode(self):
        """True if the handle factors diagonal blocks by recursion (gpx_set_safe_mode: set by
        the caller, or by the automatic switch) -- another order of arithmetic."""
        on = C.c_int(0)
        check(self._L.gpx_get_safe_mode(self._h, C.byref(on)))
        return bool(on.value)

    def set_safe_mode(self, on=True):
        check(self._L.gpx_set_safe_mode(self._h, int(bool(on))))

    def close(self):
        if getattr(self, '_h', None):
            self._L.gpx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- kernels --
    def kernel_get(self, spec, X1, X2=None, dtype=np.float64):
        dt = F32 if np.dtype(dtype) == np.float32 else F64
        X1 = np.ascontiguousarray(X1, dtype=dtype)
        n1, d = X1.shape
        if X2 is not None:
            X2 = np.ascontiguousarray(X2, dtype=dtype)
            if X2.shape[1] != d:
                raise ValueError('X1 and X2 have different dimensions')
            n2 = X2.shape[0]
        else:
            n2 = n1
        out = np.empty((n1, n2), dtype=dtype)
        if out.size == 0:
            return out
        check(self._L.gpx_kernel_get(self._h, spec.ref(), _ptr(X1), n1, _ptr(X2),
                                     n2, d, dt, _ptr(out)))
        return out

    def kernel_grad(self, spec, X1, X2=None):
        X1 = _f64(X1, 2)
        n1, d = X1.shape
        if X2 is not None:
            X2 = _f64(X2, 2)
            n2 = X2.shape[0]
        else:
            n2 = n1
        out = np.empty((spec.c.nhyper, n1, n2))
        if out.size == 0:
            return out
        check(self._L.gpx_kernel_grad(self._h, spec.ref(), _ptr(X1), n1, _ptr(X2),
                                      n2, d, _ptr(out)))
        return out

    def kernel_build_resident(self, spec, dtype=np.float64, reps=5):
        ms = C.c_double(0)
        dt = F32 if np.dtype(dtype) == np.float32 else F64
        check(self._L.gpx_kernel_build_resident(self._h, spec.ref(), dt, reps,
                                                C.byref(ms)))
        return ms.value

    # -- exact GP --
    def set_data(self, X, y):
        X, y = _f64(X, 2), _f64(y, 1)
        if X.shape[0] != y.shape[0]:
            raise ValueError('X and y disagree')
        check(self._L.gpx_set_data(self._h, _ptr(X), X.shape[0], X.shape[1],
                                   _ptr(y)))

    def exact_update(self, spec, log_sn, mean):
        info = C.c_int(0)
        check(self._L.gpx_exact_update(self._h, spec.ref(), float(log_sn),
                                       float(mean), C.byref(info)))

    def exact_append(self, Xnew, ynew):
        """True if the factor was extended in place, False if the caller has to
        refactorise (the points do not fit / nothing to extend)."""
        Xnew, ynew = _f64(Xnew, 2), _f64(ynew, 1)
        info = C.c_int(0)
        code = self._L.gpx_exact_append(self._h, _ptr(Xnew), _ptr(ynew), Xnew.shape[0],
                                        C.byref(info))
        if code == -3:
            return False
        check(code)
        return True

    def exact_loglik(self, nhyper_kernel, grad=False):
        lZ = C.c_double(0)
        dlZ = np.empty(nhyper_kernel + 2) if grad else None
        check(self._L.gpx_exact_loglik(self._h, C.byref(lZ), _ptr(dlZ)))
        return (lZ.value, dlZ) if grad else lZ.value

    def exact_eval(self, spec, log_sn, mean, grad=True):
        lZ, info = C.c_double(0), C.c_int(0)
        dlZ = np.empty(spec.c.nhyper + 2) if grad else None
        check(self._L.gpx_exact_eval(self._h, spec.ref(), float(log_sn),
                                     float(mean), int(grad), C.byref(lZ),
                                     _ptr(dlZ), C.byref(info)))
        return (lZ.value, dlZ) if grad else lZ.value

    def exact_posterior(self, Xs):
        Xs = _f64(Xs, 2)
        m = Xs.shape[0]
        mu, s2 = np.empty(m), np.empty(m)
        check(self._L.gpx_exact_posterior(self._h, _ptr(Xs), m, _ptr(mu),
                                          _ptr(s2)))
        return mu, s2

    def exact_posterior_full(self, Xs):
        Xs = _f64(Xs, 2)
        m = Xs.shape[0]
        mu, Sigma = np.empty(m), np.empty((m, m))
        check(self._L.gpx_exact_posterior_full(self._h, _ptr(Xs), m, _ptr(mu), _ptr(Sigma)))
        return mu, Sigma

    def exact_posterior_grad(self, Xs):
        Xs = _f64(Xs, 2)
        m, d = Xs.shape
        mu, s2 = np.empty(m), np.empty(m)
        dmu, ds2 = np.empty((m, d)), np.empty((m, d))
        if m:
            check(self._L.gpx_exact_posterior_grad(self._h, _ptr(Xs), m, _ptr(mu),
                                                   _ptr(s2), _ptr(dmu), _ptr(ds2)))
        return mu, s2, dmu, ds2

    def kernel_gradx(self, spec, X1, X2=None, wrt=1):
        X1 = _f64(X1, 2)
        n1, d = X1.shape
        if X2 is not None:
            X2 = _f64(X2, 2)
            n2 = X2.shape[0]
        else:
            n2 = n1
        out = np.empty((n1, n2, d))
        if out.size:
            check(self._L.gpx_kernel_gradx(self._h, spec.ref(), _ptr(X1), n1, _ptr(X2),
                                           n2, d, wrt, _ptr(out)))
        return out

    def exact_get_factor(self, n, want_R=True):
        R = np.empty((n, n)) if want_R else None
        a = np.empty(n)
        check(self._L.gpx_exact_get_factor(self._h, n, _ptr(R), _ptr(a)))
        return R, a

    def loglik_batch(self, spec, thetas, grad=False):
        thetas = _f64(thetas, 2)
        B, nth = thetas.shape
        if nth != spec.c.nhyper + 2:
            raise ValueError('thetas must have %d columns' % (spec.c.nhyper + 2))
        lZ = np.empty(B)
        dlZ = np.empty((B, nth)) if grad else None
        info = np.zeros(B, dtype=np.int32)
        check(self._L.gpx_loglik_batch(self._h, spec.ref(), _ptr(thetas), B,
                                       int(grad), _ptr(lZ), _ptr(dlZ),
                                       _ptr(info)))
        return (lZ, dlZ) if grad else lZ

    def batch_plan(self, B, grad=False):
        """How a batch of B thetas would run (gpx_batch_plan): dict with the arrangement
        ('contexts', 'groups/panel', 'groups/lockstep' or 'groups/solo'), members per group and groups in
        flight."""
        plan = (C.c_int * 4)()
        check(self._L.gpx_batch_plan(self._h, int(B), int(grad), plan))
        return {'arrangement': ('contexts', 'groups/panel', 'groups/lockstep', 'groups/solo')[plan[0]],
                'members_per_group': int(plan[1]), 'groups_in_flight': int(plan[2]),
                'safe_mode': bool(plan[3])}

    def posterior_batch(self, spec, thetas, Xs, grad=False):
        """Posterior at Xs of every model theta on the resident data: arrays
        mu, s2 of shape (B, m) [and dmu, ds2 of shape (B, m, d)]."""
        thetas = _f64(thetas, 2)
        B, nth = thetas.shape
        if nth != spec.c.nhyper + 2:
            raise ValueError('thetas must have %d columns' % (spec.c.nhyper + 2))
        Xs = _f64(Xs, 2)
        m, d = Xs.shape
        mu, s2 = np.empty((B, m)), np.empty((B, m))
        dmu = np.empty((B, m, d)) if grad else None
        ds2 = np.empty((B, m, d)) if grad else None
        info = np.zeros(B, dtype=np.int32)
        check(self._L.gpx_posterior_batch(self._h, spec.ref(), _ptr(thetas), B, _ptr(Xs), m,
                                          _ptr(mu), _ptr(s2), _ptr(dmu), _ptr(ds2),
                                          _ptr(info)))
        return (mu, s2, dmu, ds2) if grad else (mu, s2)

    # -- instrumentation --
    def enable_timing(self, on=True):
        check(self._L.gpx_enable_timing(self._h, int(on)))

    def timings(self):
        ms = np.zeros(NTIMERS)
        check(self._L.gpx_get_timings(self._h, _ptr(ms), NTIMERS))
        names = [self._L.gpx_timing_name(i).decode() for i in range(NTIMERS)]
        return dict(zip(names, ms))

    def batch_timings(self):
        """(ms, members): event time of the factor (+ inverse) stages of the groups that
        gpx_loglik_batch ran since enable_timing(True), and the members they held."""
        ms, mem = C.c_double(0), _i64(0)
        check(self._L.gpx_batch_timings(self._h, C.byref(ms), C.byref(mem)))
        return ms.value, mem.value

    def synchronize(self):
        check(self._L.gpx_synchronize(self._h))

    # -- dense building blocks --
    def la_gemm(self, A, B, ta=False, tb=False, alpha=1.0, beta=0.0, Cin=None):
        A, B = _f64(A, 2), _f64(B, 2)
        M = A.shape[1] if ta else A.shape[0]
        K = A.shape[0] if ta else A.shape[1]
        N = B.shape[0] if tb else B.shape[1]
        out = np.zeros((M, N)) if Cin is None else _f64(Cin, 2).copy()
        check(self._L.gpx_la_gemm(self._h, int(ta), int(tb), M, N, K, alpha,
                                  _ptr(A), A.shape[1], _ptr(B), B.shape[1], beta,
                                  _ptr(out), N))
        return out

    def la_potrf(self, A, inverse=False):
        A = _f64(A, 2)
        n = A.shape[0]
        R = np.empty((n, n))
        Rinv = np.empty((n, n)) if inverse else None
        Ainv = np.empty((n, n)) if inverse else None
        info = C.c_int(0)
        check(self._L.gpx_la_potrf(self._h, _ptr(A), n, _ptr(R), _ptr(Rinv),
                                   _ptr(Ainv), C.byref(info)))
        return (R, Rinv, Ainv) if inverse else R

    def la_gemm_bench(self, n, ta=False, tb=False, reps=5):
        ms = C.c_double(0)
        check(self._L.gpx_la_gemm_bench(self._h, int(ta), int(tb), n, reps,
                                        C.byref(ms)))
        return ms.value

    def la_gemm_bench_ex(self, n, ta=0, tb=0, flags=0, order=0, swizzle=0, tile=0,
                         waves=0, same_ab=0, reps=1):
        ms = C.c_double(0)
        check(self._L.gpx_la_gemm_bench_ex(self._h, int(ta), int(tb), n, flags, order,
                                           swizzle, tile, waves, same_ab, reps,
                                           C.byref(ms)))
        return ms.value

    def la_gemm_bench_mnk(self, M, N, K, ta=0, tb=0, flags=0, beta=0.0, tile=0, reps=3):
        ms = C.c_double(0)
        check(self._L.gpx_la_gemm_bench_mnk(self._h, int(ta), int(tb), M, N, K, flags,
                                            float(beta), tile, reps, C.byref(ms)))
        return ms.value

    def la_potrf_bench(self, n, inverse=False, reps=3):
        ms = C.c_double(0)
        check(self._L.gpx_la_potrf_bench(self._h, n, int(inverse), reps,
                                         C.byref(ms)))
        return ms.value


for _name in [n for n, f in list(vars(Handle).items())
              if callable(f) and not n.startswith('_') and n != 'close']:
    setattr(Handle, _name, _serialised(getattr(Handle, _name)))


def device_count():
    n = C.c_int(0)
    check(lib().gpx_device_count(C.byref(n)))
    return n.value


# Whose data the library's per-device handles hold (they are process-wide): a caller
# that passes a `token` with its data uploads X, y only when another data set, another
# caller or fewer devices were there last, and NULL (resident data) otherwise.
_multi_resident = (None, 0)          # (token, ndev)


def _multi_call(call, X, y, token, ndev):
    """One multi-device call under the library lock: decide whether X, y have to go down
    (they do unless the devices hold the data of `token` already), call, and record what
    is resident afterwards -- all three inside the lock, so that two threads cannot
    interleave the decision and the update; a call that fails leaves the record empty
    (the C side drops its resident data on an error path too)."""
    global _multi_resident
    with _lock:
        tok, have = _multi_resident
        if X is not None and token is not None and tok == token and have >= ndev:
            X = y = None
        uploaded = X is not None
        try:
            out = call(X, y)
        except Exception:
            _multi_resident = (None, 0)
            raise
        if uploaded:
            _multi_resident = (token, ndev)
    return out


def loglik_batch_multi(spec, thetas, X=None, y=None, grad=False, ndev=1, token=None):
    """gpx_loglik_batch_multi: B thetas over the first `ndev` GPUs of the node from
    this one process (one host thread and handle per device inside the library, ONE
    RCCL all-gather; no torch). X = y = None evaluates on the data a previous call
    left resident. token (hashable, with X and y): upload X, y only if the devices do
    not already hold the data of this token (one upload per data set instead of one
    per call). Returns lZ (B,) [and dlZ (B, nth)]; members that are not positive
    definite come back as -inf / NaN."""
    return _multi_call(lambda X_, y_: _loglik_batch_multi(spec, thetas, X_, y_, grad, ndev),
                       X, y, token, ndev)


def _loglik_batch_multi(spec, thetas, X, y, grad, ndev):
    thetas = _f64(thetas, 2)
    B, nth = thetas.shape
    if nth != spec.c.nhyper + 2:
        raise ValueError('thetas must have %d columns' % (spec.c.nhyper + 2))
    if (X is None) != (y is None):
        raise ValueError('pass both X and y, or neither')
    n = d = 0
    if X is not None:
        X, y = _f64(X, 2), _f64(y, 1)
        if X.shape[0] != y.shape[0]:
            raise ValueError('X and y disagree')
        n, d = X.shape
    lZ = np.empty(B)
    dlZ = np.empty((B, nth)) if grad else None
    info = np.zeros(B, dtype=np.int32)
    check(lib().gpx_loglik_batch_multi(spec.ref(), _ptr(thetas), B, _ptr(X), _ptr(y), n, d,
                                       int(grad), int(ndev), _ptr(lZ), _ptr(dlZ),
                                       _ptr(info)))
    return (lZ, dlZ) if grad else lZ


def posterior_batch_multi(spec, thetas, Xs, X=None, y=None, grad=False, ndev=1, token=None):
    """gpx_posterior_batch_multi: the posteriors at Xs of the B models theta over the first
    `ndev` GPUs of the node from this one process (see loglik_batch_multi, also for
    `token`). Returns mu, s2 of shape (B, m) [and dmu, ds2 of shape (B, m, d)]; rows of
    members that are not positive definite are NaN."""
    return _multi_call(lambda X_, y_: _posterior_batch_multi(spec, thetas, Xs, X_, y_, grad, ndev),
                       X, y, token, ndev)


def _posterior_batch_multi(spec, thetas, Xs, X, y, grad, ndev):
    thetas = _f64(thetas, 2)
    B, nth = thetas.shape
    if nth != spec.c.nhyper + 2:
        raise ValueError('thetas must have %d columns' % (spec.c.nhyper + 2))
    if (X is None) != (y is None):
        raise ValueError('pass both X and y, or neither')
    Xs = _f64(Xs, 2)
    m, d = Xs.shape
    n = 0
    if X is not None:
        X, y = _f64(X, 2), _f64(y, 1)
        if X.shape[0] != y.shape[0] or X.shape[1] != d:
            raise ValueError('X, y and Xs disagree')
        n = X.shape[0]
    mu, s2 = np.empty((B, m)), np.empty((B, m))
    dmu = np.empty((B, m, d)) if grad else None
    ds2 = np.empty((B, m, d)) if grad else None
    info = np.zeros(B, dtype=np.int32)
    check(lib().gpx_posterior_batch_multi(spec.ref(), _ptr(thetas), B, _ptr(X), _ptr(y), n,
                                          d if X is not None else 0, _ptr(Xs), m, int(grad),
                                          int(ndev), _ptr(mu), _ptr(s2), _ptr(dmu), _ptr(ds2),
                                          _ptr(info)))
    return (mu, s2, dmu, ds2) if grad else (mu, s2)


def batch_partition(B, world, rank):
    lo, hi = _i64(0), _i64(0)
    lib().gpx_batch_partition(B, world, rank, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def multi_comm_size():
    """Ranks of the RCCL communicator of the last multi-device call (0: none yet)."""
    return int(lib().gpx_multi_comm_size())


def multi_enable_timing(ndev, on=True):
    """HIP-event timing of the groups on the library's handles of the first ndev devices
    (they exist after the first multi-device call with that many devices)."""
    check(lib().gpx_multi_enable_timing(int(ndev), int(bool(on))))


def multi_batch_info(ndev, B_per_dev, grad=False):
    """Per device of the in-library multi-device path: the event time (ms) of its groups'
    dense stages and the members they held since multi_enable_timing(ndev, True), and how its
    handle cuts a block of B_per_dev thetas (batch_plan's dict, safe_mode included)."""
    ms = (C.c_double * ndev)()
    mem = (_i64 * ndev)()
    plans = (C.c_int * (4 * ndev))()
    check(lib().gpx_multi_batch_info(int(ndev), int(B_per_dev), int(grad), ms, mem, plans))
    return [{'device': i, 'dense_ms': float(ms[i]), 'members': int(mem[i]),
             'arrangement': ('contexts', 'groups/panel', 'groups/lockstep', 'groups/solo')[plans[4 * i]],
             'members_per_group': int(plans[4 * i + 1]),
             'groups_in_flight': int(plans[4 * i + 2]),
             'safe_mode': bool(plans[4 * i + 3])} for i in range(ndev)]


def panel_grid_check(nmem, ncu=256):
    """(spine workgroups per member, workers) of a panel launch over nmem members on a device
    of ncu CUs, or RuntimeError if even one spine workgroup per member does not fit
    (gpx_panel_grid_check; host only)."""
    sp, wk = C.c_int(0), C.c_int(0)
    check(lib().gpx_panel_grid_check(int(nmem), int(ncu), C.byref(sp), C.byref(wk)))
    return sp.value, wk.value


def multi_pack(rows, slot):
    """gpx_multi_pack: one device's member rows [cnt][width] as its NaN-padded gather
    slot [slot][width] (host only)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    cnt, width = rows.shape
    pack = np.empty((int(slot), width))
    check(lib().gpx_multi_pack(_ptr(rows) if cnt else None, cnt, width, int(slot), _ptr(pack)))
    return pack


def multi_scatter(gathered, B, ndev, width):
    """gpx_multi_scatter: the gathered [ndev][slot][width] image -> out[B][width]."""
    gathered = np.ascontiguousarray(gathered, dtype=np.float64)
    out = np.full((int(B), int(width)), np.nan)
    check(lib().gpx_multi_scatter(_ptr(gathered) if B else None, int(B), int(ndev), int(width),
                                  _ptr(out) if B else None))
    return out


def sweep_check(T, aug=False):
    """Host-side replay of the lock-step sweep of a block of T tiles (gpx_sweep_check);
    raises RuntimeError naming the first violation."""
    check(lib().gpx_sweep_check(int(T), int(bool(aug))))


def panel_solo_check(T, aug=False, full=False, value_only=False):
    """Host-side check that the task graph of T tiles in generation order is a sequential
    order (the list of a solo launch, gpx_panel_solo_check): number of tasks, or RuntimeError."""
    n = C.c_int(0)
    check(lib().gpx_panel_solo_check(int(T), int(bool(aug)), int(bool(full)), int(bool(value_only)),
                                     C.byref(n)))
    return n.value


def sweep_check_lite(T, aug=False, depth=-1):
    """... with the dense row panels of round 5 (gpx_sweep_check_lite): each tile right of
    (s, s+1) applies its last `depth` trailing updates itself (-1: the library's rule)."""
    check(lib().gpx_sweep_check_lite(int(T), int(bool(aug)), int(depth)))


def panel_graph_check_rhs(T, workers=64):
    """The task graph of a launch over a whole matrix of T tiles with a right-hand-side
    tile column (gpx_panel_graph_check_rhs): number of tasks, or RuntimeError."""
    n = C.c_int(0)
    check(lib().gpx_panel_graph_check_rhs(int(T), int(workers), C.byref(n)))
    return n.value


def panel_graph_check_full(T, workers=64):
    """... with the whole of R^-1 assembled inside the launch (gpx_panel_graph_check_full)."""
    n = C.c_int(0)
    check(lib().gpx_panel_graph_check_full(int(T), int(workers), C.byref(n)))
    return n.value


def panel_graph_check(T, workers=64, stream=True, extra=0):
    """Host-side self-check of the diagonal-panel kernel's task graph for a block of T
    128-tiles (no GPU), optionally as a wide panel with `extra` more tile columns:
    returns the number of tasks, raises RuntimeError naming the first violation (see
    gpx_panel_graph_check in include/gpx.h)."""
    n = C.c_int(0)
    if extra:
        check(lib().gpx_panel_graph_check_wide(int(T), int(extra), int(workers), C.byref(n)))
    else:
        check(lib().gpx_panel_graph_check(int(T), int(workers), int(bool(stream)), C.byref(n)))
    return n.value


_default = None


def default_handle():
    """Process-wide handle used by Kernel.get()/grad() (device from GPX_DEVICE
    or LOCAL_RANK, default 0)."""
    global _default
    if _default is None:
        with _lock:
            if _default is None:
                _default = Handle()
    return _default
