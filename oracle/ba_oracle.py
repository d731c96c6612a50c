"""ctypes front-end for oracle/ba_oracle.cpp (CPU restatement of the VI bundle adjustment).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from xrslam_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle_ba.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle_ba.so"])
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a):
    return np.ascontiguousarray(a, np.float64)


def solve(pd):
    """In-place solve of an abi.BaProblemData; returns abi.BaSummary."""
    s = pd.struct()
    sm = abi.BaSummary()
    rc = lib().orc_ba_solve(C.byref(s), C.byref(sm))
    assert rc == 0
    return sm


def solve_trace(pd, max_rows=256):
    """Like solve(); also returns the per-iteration record [rows][9] = {iteration, x_cost, candidate cost, model cost
    change, relative decrease, radius, |step|, mu, accepted} of the trials that reached the accept / reject decision."""
    s = pd.struct()
    sm = abi.BaSummary()
    tr = np.zeros((max_rows, 9))
    rows = C.c_int()
    rc = lib().orc_ba_solve_trace(C.byref(s), C.byref(sm), _p(tr), max_rows, C.byref(rows))
    assert rc == 0
    return sm, tr[:rows.value].copy()


def linearize(pd, want_H=True):
    s = pd.struct()
    nf, nl = len(pd.frame_state), len(pd.inv_depth)
    nmax = 15 * nf + nl
    cost = C.c_double()
    po = np.zeros(nf, np.int32)
    mo = np.zeros(nf, np.int32)
    lo = np.zeros(max(nl, 1), np.int32)
    if want_H:
        H = np.zeros(nmax * nmax)
        g = np.zeros(nmax)
        n = lib().orc_ba_linearize(C.byref(s), C.byref(cost), _p(H), _p(g), _p(po), _p(mo), _p(lo))
        return cost.value, H[:n * n].reshape(n, n).copy(), g[:n].copy(), po, mo, lo[:nl]
    lib().orc_ba_linearize(C.byref(s), C.byref(cost), None, None, _p(po), _p(mo), _p(lo))
    return cost.value


def marginalize(md):
    s = md.struct()
    k = len(md.frame_state) - 1
    n = 15 * k
    si = np.zeros((n, n))
    iv = np.zeros(n)
    lin = np.zeros((k, 16))
    rc = lib().orc_ba_marginalize(C.byref(s), _p(si), _p(iv), _p(lin))
    assert rc == 0, rc
    return si, iv, lin


def preintegrate(samples, t_end, bg, ba, noise36, jac=True, cov=True):
    samples = _c(samples).reshape(-1, 7)
    out = np.zeros(abi.IMU_DIM)
    bg, ba, noise36 = _c(bg), _c(ba), _c(noise36)
    rc = lib().orc_preintegrate(_p(samples), len(samples), C.c_double(t_end), _p(bg), _p(ba), _p(noise36), int(jac),
                                int(cov), _p(out))
    assert rc == 0, rc
    return out


def preintegrate_cov(samples, t_end, bg, ba, noise36):
    samples = _c(samples).reshape(-1, 7)
    out = np.zeros(225)
    bg, ba, noise36 = _c(bg), _c(ba), _c(noise36)
    lib().orc_preintegrate_cov(_p(samples), len(samples), C.c_double(t_end), _p(bg), _p(ba), _p(noise36), _p(out))
    return out.reshape(15, 15)


def predict(state_i, imu_data):
    out = np.zeros(16)
    a, b = _c(state_i), _c(imu_data)
    lib().orc_predict(_p(a), _p(b), _p(out))
    return out


def state_plus(state, delta15):
    out = np.zeros(16)
    a, b = _c(state), _c(delta15)
    lib().orc_state_plus(_p(a), _p(b), _p(out))
    return out


def eval_reprojection(st_t, st_r, inv_depth, z_t, z_r, cam7, sic2, jac=True):
    r = np.zeros(2)
    Jt = np.zeros(12)
    Jr = np.zeros(12)
    Jl = np.zeros(2)
    a = [_c(v) for v in (st_t, st_r, z_t, z_r, cam7, sic2)]
    lib().orc_eval_reprojection(_p(a[0]), _p(a[1]), C.c_double(inv_depth), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]),
                                _p(r), _p(Jt) if jac else None, _p(Jr) if jac else None, _p(Jl) if jac else None)
    return r, Jt.reshape(2, 6), Jr.reshape(2, 6), Jl


def eval_rotation(st_t, st_r, z_t, z_r, cam7, sic2, jac=True):
    r = np.zeros(2)
    Jq = np.zeros(6)
    a = [_c(v) for v in (st_t, st_r, z_t, z_r, cam7, sic2)]
    lib().orc_eval_rotation(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(r), _p(Jq) if jac else None)
    return r, Jq.reshape(2, 3)


def eval_imu(st_i, st_j, imu_data, bias_ref6, imu7, jac=True):
    r = np.zeros(15)
    Ji = np.zeros(225)
    Jj = np.zeros(225)
    a = [_c(v) for v in (st_i, st_j, imu_data, bias_ref6, imu7)]
    lib().orc_eval_imu(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(r), _p(Ji) if jac else None,
                       _p(Jj) if jac else None)
    return r, Ji.reshape(15, 15), Jj.reshape(15, 15)
