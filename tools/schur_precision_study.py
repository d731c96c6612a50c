"""BASELINE config 5 asks for an "fp32 vs bf16 BA solve (tolerance ...)" study.  This is the CPU half of it: take the
normal equations of a config-5 sized window (20 keyframes + the new frame, 600 landmarks) from the oracle, form
the Schur complement  S = H_pp - W^T H_ll^-1 W  and the Gauss-Newton step with the contraction's inputs rounded to
f32 / bf16 (products exact, accumulation in f32: what v_mfma_f32_32x32x8_f16-class instructions do) and compare
with the f64 contraction the product uses (v_mfma_f64_16x16x4_f64).  north_star's bound is 1e-4 relative on the
states; one Gauss-Newton step is the quantity that must stay inside it.

    python tools/schur_precision_study.py          # needs oracle/_build (make -C oracle); no GPU
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo  # noqa: E402  (development aid; never imported by the product)
from tests import ba_synth as bs  # noqa: E402


def round_bf16(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16          # round to nearest even on the upper 16 bits
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def contraction(W, w, mode):
    """T = W^T diag(w) W with the operand precision of `mode`."""
    if mode == "f64":
        return W.T @ (w[:, None] * W)
    A = (np.sqrt(w)[:, None] * W)                               # symmetric split so both operands round alike
    if mode == "f32":
        A32 = A.astype(np.float32)
        return (A32.T @ A32).astype(np.float64)                 # f32 products and accumulation
    A16 = round_bf16(A)
    return (A16.astype(np.float32).T @ A16.astype(np.float32)).astype(np.float64)


def main():
    pd, _ = bs.make_window(K=21, L=600, seed=4)
    _, H, g, po, mo, lo = bo.linearize(pd)
    nl = int((lo >= 0).sum())
    npose = H.shape[0] - nl
    Hpp, Hpl, hll = H[:npose, :npose], H[:npose, npose:], np.diag(H)[npose:]
    gp, gl = g[:npose], g[npose:]
    # Jacobi scaling like the solver (columns scaled by 1/(1+sqrt(diag)))
    sp = 1.0 / (1.0 + np.sqrt(np.diag(Hpp)))
    sl = 1.0 / (1.0 + np.sqrt(hll))
    Hpp_s = Hpp * np.outer(sp, sp)
    W = (Hpl * sp[:, None] * sl[None, :]).T                     # [landmark, pose dof]
    hll_s = hll * sl * sl
    rhs_p, rhs_l = gp * sp, gl * sl
    w = 1.0 / hll_s
    out = {}
    for mode in ("f64", "f32", "bf16"):
        T = contraction(W, w, mode)
        S = Hpp_s - T
        b = rhs_p - W.T @ (w * rhs_l)
        S = S + 1e-12 * np.eye(npose) * np.abs(np.diag(S)).max() if mode != "f64" else S
        try:
            x = np.linalg.solve(S, b)
        except np.linalg.LinAlgError:
            x = np.full(npose, np.nan)
        out[mode] = (T, x * sp)
    T0, x0 = out["f64"]
    print("config-5 window: %d frame unknowns, %d landmarks, %d observations" % (npose, nl, len(pd.obs_lm)))
    for mode in ("f32", "bf16"):
        T, x = out[mode]
        print("%-5s contraction: max |T - T64| / max|T64| = %.2e ; Gauss-Newton step: max relative deviation %.2e "
              "(|dx| / |x64|_inf)" % (mode, np.abs(T - T0).max() / np.abs(T0).max(), np.abs(x - x0).max() / np.abs(x0).max()))
    print("north_star tolerance on the states: 1e-4 relative")


if __name__ == "__main__":
    main()
