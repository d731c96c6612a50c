"""ctypes mirror of plug point #2 (xrslam::Solver / MarginalizationFactor) over include/xrslam_hip.h."""
import ctypes as C

import numpy as np

from . import _lib, abi
from ._lib import check

_L = None

HOST_WORK = C.CFUNCTYPE(None, C.c_void_p)   # void (*host_work)(void *)


def configure(lib):
    """Declares the argument types of the plug point's entry points on `lib` (the product library, or -- in tests -- the CPU
    oracle's shim, which exports the same symbols)."""
    vp = C.c_void_p
    lib.xrhip_ba_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.xrhip_ba_destroy.argtypes = [vp]
    lib.xrhip_ba_destroy.restype = None
    lib.xrhip_ba_solve.argtypes = [vp, C.POINTER(abi.BaProblem), C.POINTER(abi.BaSummary)]
    if hasattr(lib, "xrhip_ba_solve_overlapped"):
        lib.xrhip_ba_solve_overlapped.argtypes = [vp, C.POINTER(abi.BaProblem), C.POINTER(abi.BaSummary), HOST_WORK, vp]
    if hasattr(lib, "xrhip_ba_marginalize"):
        lib.xrhip_ba_marginalize.argtypes = [vp, C.POINTER(abi.MargProblem), vp, vp, vp]
    if hasattr(lib, "xrhip_ba_marginalize_begin"):
        lib.xrhip_ba_marginalize_begin.argtypes = [vp, C.POINTER(abi.MargProblem)]
        lib.xrhip_ba_marginalize_end.argtypes = [vp, vp, vp, vp]
    if hasattr(lib, "xrhip_ba_preintegrate"):
        lib.xrhip_ba_preintegrate.argtypes = [vp, vp, C.c_int, C.c_double, vp, vp, vp, C.c_int, C.c_int, vp]
    if hasattr(lib, "xrhip_ba_preintegrate_begin"):
        lib.xrhip_ba_preintegrate_begin.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int]
    if hasattr(lib, "xrhip_ba_preintegrate_end"):
        lib.xrhip_ba_preintegrate_end.argtypes = [vp, vp]
    if hasattr(lib, "xrhip_ba_preintegrate_early"):
        lib.xrhip_ba_preintegrate_early.argtypes = [vp, C.c_int, vp]
    if hasattr(lib, "xrhip_ba_solve_chained"):
        lib.xrhip_ba_solve_chained.argtypes = [vp, C.POINTER(abi.BaProblem), C.POINTER(abi.BaSummary), C.c_int, vp,
                                               C.POINTER(abi.BaProblem), C.POINTER(abi.BaSummary), C.c_int, HOST_WORK, vp]
    if hasattr(lib, "xrhip_ba_solve_begin"):
        lib.xrhip_ba_solve_begin.argtypes = [vp, C.POINTER(abi.BaProblem)]
        lib.xrhip_ba_solve_end.argtypes = [vp, C.POINTER(abi.BaSummary)]
        lib.xrhip_ba_solve_abort.argtypes = [vp]
    if hasattr(lib, "xrhip_ba_preintegrate_cancel"):
        lib.xrhip_ba_preintegrate_cancel.argtypes = [vp]
    if hasattr(lib, "xrhip_ba_preintegrate_after_solve"):
        lib.xrhip_ba_preintegrate_after_solve.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int]
    if hasattr(lib, "xrhip_ba_debug_linearize"):
        lib.xrhip_ba_debug_linearize.argtypes = [vp, C.POINTER(abi.BaProblem), vp, vp, vp, vp, vp, vp]
    if hasattr(lib, "xrhip_ba_debug_schur"):
        lib.xrhip_ba_debug_schur.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    if hasattr(lib, "xrhip_ba_debug_set_schur_precision"):
        lib.xrhip_ba_debug_set_schur_precision.argtypes = [vp, C.c_int]
    return lib


def L():
    global _L
    if _L is None:
        _L = configure(_lib.lib())
    return _L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def study_schur_precision(W, w, reps=20):
    """BASELINE config 5's study (xrhip_study_schur_precision): T = W^T diag(w) W with f64 / f32 / bf16 matrix-core operands.
    Returns ({"f64": T, "f32": T, "bf16": T}, {"f64": ms, ...}) -- products [P][P] and the average kernel time per launch."""
    lib = L()
    W = np.ascontiguousarray(W, np.float64)
    w = np.ascontiguousarray(w, np.float64)
    Ln, P = W.shape
    outs = [np.zeros((P, P)) for _ in range(3)]
    ms = (C.c_float * 3)()
    lib.xrhip_study_schur_precision.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.POINTER(C.c_float)]
    check(lib.xrhip_study_schur_precision(_p(W), _p(w), Ln, P, int(reps), _p(outs[0]), _p(outs[1]), _p(outs[2]), ms))
    names = ("f64", "f32", "bf16")
    return dict(zip(names, outs)), dict(zip(names, [float(v) for v in ms]))


class BaContext:
    """One per sequence: owns the BA stream and device arenas."""

    def __init__(self, max_frames=24, max_landmarks=1024, max_obs=8192, lib=None):
        self._lib = lib if lib is not None else L()
        h = C.c_void_p()
        check(self._lib.xrhip_ba_create(int(max_frames), int(max_landmarks), int(max_obs), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.xrhip_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, pd, host_work=None):
        """Solver::solve(): optimises pd.frame_state / pd.inv_depth in place; returns abi.BaSummary.
        host_work: a callable run once beside the device's work (xrhip_ba_solve_overlapped)."""
        s = pd.struct()
        sm = abi.BaSummary()
        if host_work is None:
            check(self._lib.xrhip_ba_solve(self._h, C.byref(s), C.byref(sm)))
        else:
            raised = []   # ctypes only PRINTS what a callback raises: keep it and re-raise once the solve has returned

            def run(_arg):
                try:
                    host_work()
                except BaseException as e:   # noqa: BLE001  (re-raised below)
                    raised.append(e)
            cb = HOST_WORK(run)
            rc = self._lib.xrhip_ba_solve_overlapped(self._h, C.byref(s), C.byref(sm), cb, None)
            if raised:
                raise raised[0]
            check(rc)
        return sm

    def solve_chained(self, pd_first, link_first, second_ctx, pd_second, link_second, host_work=None):
        """xrhip_ba_solve_chained: pd_first on this context, pd_second on second_ctx starting from pd_first's result for the linked
        frame; both solved in place.  -> (summary_first, summary_second)"""
        s1, s2 = pd_first.struct(), pd_second.struct()
        m1, m2 = abi.BaSummary(), abi.BaSummary()
        raised = []

        def run(_arg):
            try:
                if host_work is not None:
                    host_work()
            except BaseException as e:   # noqa: BLE001  (re-raised below)
                raised.append(e)
        cb = HOST_WORK(run)
        rc = self._lib.xrhip_ba_solve_chained(self._h, C.byref(s1), C.byref(m1), int(link_first), second_ctx._h, C.byref(s2),
                                              C.byref(m2), int(link_second), cb, None)
        if raised:
            raise raised[0]
        check(rc)
        return m1, m2

    def solve_begin(self, pd):
        """xrhip_ba_solve_begin: True = queued (solve_end() collects it into pd), False = not a single-launch problem, nothing done."""
        self._begun = (pd, pd.struct())      # the problem must stay alive until solve_end
        rc = self._lib.xrhip_ba_solve_begin(self._h, C.byref(self._begun[1]))
        if rc < 0:
            check(rc)
        return rc == 1

    def solve_end(self):
        sm = abi.BaSummary()
        check(self._lib.xrhip_ba_solve_end(self._h, C.byref(sm)))
        self._begun = None
        return sm

    def solve_abort(self):
        check(self._lib.xrhip_ba_solve_abort(self._h))
        self._begun = None

    def preintegrate_begin(self, samples, t_end, bg, ba, noise36, jac=True, cov=True):
        """One interval queued (xrhip_ba_preintegrate_begin); preintegrate_early() / preintegrate_end() collect it."""
        samples = np.ascontiguousarray(samples, np.float64).reshape(-1, 7)
        bg, ba, noise36 = [np.ascontiguousarray(v, np.float64) for v in (bg, ba, noise36)]
        begin, count = (np.array([v], np.int32) for v in (0, len(samples)))
        t = np.array([t_end], np.float64)
        check(self._lib.xrhip_ba_preintegrate_begin(self._h, _p(samples), _p(begin), _p(count), _p(t), _p(bg), _p(ba), 1, _p(noise36),
                                                    int(jac), int(cov)))

    def preintegrate_early(self, job=0):
        """The delta (dt, dq, dp, dv) of a queued job as soon as the kernel has it (xrhip_ba_preintegrate_early)."""
        out = np.zeros(11)
        check(self._lib.xrhip_ba_preintegrate_early(self._h, int(job), _p(out)))
        return out

    def marginalize(self, md):
        s = md.struct()
        k = len(md.frame_state) - 1
        n = 15 * k
        si = np.zeros((n, n))
        iv = np.zeros(n)
        lin = np.zeros((k, 16))
        check(self._lib.xrhip_ba_marginalize(self._h, C.byref(s), _p(si), _p(iv), _p(lin)))
        return si, iv, lin

    def marg_guard(self):
        """-> (eigenvalue bound of the last marginalisation's Cholesky fast path, its eight status words)"""
        lam = C.c_double()
        st = (C.c_int * 8)()
        self._lib.xrhip_ba_debug_marg_guard.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        check(self._lib.xrhip_ba_debug_marg_guard(self._h, C.byref(lam), st))
        return lam.value, list(st)

    def marginalize_begin(self, md):
        """Queues the marginalisation (xrhip_ba_marginalize_begin); marginalize_end() returns its result."""
        self._marg_k = len(md.frame_state) - 1
        check(self._lib.xrhip_ba_marginalize_begin(self._h, C.byref(md.struct())))

    def marginalize_end(self):
        k = self._marg_k
        n = 15 * k
        si = np.zeros((n, n))
        iv = np.zeros(n)
        lin = np.zeros((k, 16))
        check(self._lib.xrhip_ba_marginalize_end(self._h, _p(si), _p(iv), _p(lin)))
        return si, iv, lin

    def preintegrate(self, samples, t_end, bg, ba, noise36, jac=True, cov=True):
        samples = np.ascontiguousarray(samples, np.float64).reshape(-1, 7)
        bg, ba, noise36 = [np.ascontiguousarray(v, np.float64) for v in (bg, ba, noise36)]
        out = np.zeros(abi.IMU_DIM)
        check(self._lib.xrhip_ba_preintegrate(self._h, _p(samples), len(samples), float(t_end), _p(bg), _p(ba), _p(noise36),
                                        int(jac), int(cov), _p(out)))
        return out

    def preintegrate_after_solve(self, samples, t_end, bias_frame, noise36, jac=True, cov=True):
        """Stages one interval whose integration starts from the biases the NEXT solve() gives frame `bias_frame` of its
        problem (xrhip_ba_preintegrate_after_solve); preintegrate_end() returns the record once that solve has run."""
        samples = np.ascontiguousarray(samples, np.float64).reshape(-1, 7)
        noise36 = np.ascontiguousarray(noise36, np.float64)
        begin, count, frame = (np.array([v], np.int32) for v in (0, len(samples), bias_frame))
        t = np.array([t_end], np.float64)
        check(self._lib.xrhip_ba_preintegrate_after_solve(self._h, _p(samples), _p(begin), _p(count), _p(t), _p(frame), 1, _p(noise36),
                                                    int(jac), int(cov)))

    def preintegrate_end(self, n_jobs=1):
        out = np.zeros((n_jobs, abi.IMU_DIM))
        check(self._lib.xrhip_ba_preintegrate_end(self._h, _p(out)))
        return out[0] if n_jobs == 1 else out

    def debug_linearize(self, pd):
        s = pd.struct()
        F, Ln = len(pd.frame_state), len(pd.inv_depth)
        H = np.zeros((15 * F, 15 * F))
        g = np.zeros(15 * F)
        hll = np.zeros(max(Ln, 1))
        gl = np.zeros(max(Ln, 1))
        W = np.zeros((max(Ln, 1), 6 * F))
        cost = C.c_double()
        check(self._lib.xrhip_ba_debug_linearize(self._h, C.byref(s), _p(H), _p(g), _p(hll), _p(gl), _p(W), C.byref(cost)))
        return dict(H=H, g=g, hll=hll[:Ln], gl=gl[:Ln], W=W[:Ln], cost=cost.value)

    def set_schur_precision(self, mode):
        """Study aid (BASELINE config 5): 0 = f64 (product), 1 = f32, 2 = bf16 operands in the Schur contraction of the solves that follow."""
        check(self._lib.xrhip_ba_debug_set_schur_precision(self._h, int(mode)))

    def debug_schur(self, W, w):
        W = np.ascontiguousarray(W, np.float64)
        w = np.ascontiguousarray(w, np.float64)
        Ln, P = W.shape
        out = np.zeros((P, P))
        check(self._lib.xrhip_ba_debug_schur(self._h, _p(W), _p(w), Ln, P, _p(out)))
        return out
