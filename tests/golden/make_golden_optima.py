"""Optima found by an INDEPENDENT optimiser (scipy TRF, exact dense trust-region solve, central-difference Jacobians) on the
oracle's residual functions, robustified per block by hand -- SURVEY 8(c)-3.  Written to tests/golden/scipy_optima.json:

  cfg1   BASELINE config 1 (poly3, 50 frames, no IMU), from the engine's start values to |g|_inf ~ 1e-5 (cost ~ 77):
         the intrinsics scipy converges to.
  vi60   mono kb4 + IMU, 60 frames: the oracle's stage machine run with function_tolerance 1e-12, then polished by scipy on
         the final stage's objective (4 copies of every reprojection block, 3 of every weighted Cauchy IMU block); scipy
         moves nothing by more than 2e-9 relative -- the stored state is that polished optimum.

tests/test_oracle.py re-derives cfg1 on every CPU run; the GPU suite compares the HIP solver (same tolerances) with this
file at 1e-6, the tolerance north_star states, without running scipy on the GPU box.
Usage: python tests/golden/make_golden_optima.py"""
import json
import os
import sys

import numpy as np
import scipy.optimize as scipy_opt

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402
from vicalib_amd import synth  # noqa: E402


def cfg1_scipy_optimum():
    """(K of scipy's optimum, scipy's first-order optimality) for BASELINE cfg1."""
    p = synth.generate(synth.BASELINE_CONFIGS["cfg1"])
    o2 = ol.Oracle().load(p); o2.set_options(calibrate_imu=False); o2.prepare()
    L = ol.lib()
    T0 = o2.frames()[0]; K0, Tck = o2.camera(0); n = o2.n_frames
    _, fr, _ = o2.residuals()
    rows_of = [np.nonzero(np.repeat(fr, 2) == f)[0] for f in range(n)]

    def apply(x):
        for f in range(n):
            Tn = np.zeros(7); L.vco_plus_se3(ol._d(T0[f]), ol._d(x[6 * f:6 * f + 6]), ol._d(Tn)); o2.set_frame(f, Tn)
        o2.set_camera(0, K0 + x[6 * n:], Tck)

    def fun(x):
        apply(x)
        r = o2.residuals()[0]
        s = (r * r).sum(axis=1)
        rho = 2 * 0.25 * (np.sqrt(1 + s / 0.25) - 1)                    # SoftLOneLoss(0.5), per block
        return (r * np.sqrt(rho / np.maximum(s, 1e-300))[:, None]).ravel()

    def jac(x):
        # central differences; the j-th pose coordinate of every frame is perturbed at once (their residual rows are disjoint)
        J = np.zeros((2 * len(fr), 6 * n + 7)); h = 1e-6
        for j in range(6):
            e = np.zeros_like(x); e[j:6 * n:6] = h
            d = (fun(x + e) - fun(x - e)) / (2 * h)
            for f in range(n):
                J[rows_of[f], 6 * f + j] = d[rows_of[f]]
        for j in range(7):
            hj = h * max(1.0, abs(K0[j]))
            e = np.zeros_like(x); e[6 * n + j] = hj
            J[:, 6 * n + j] = (fun(x + e) - fun(x - e)) / (2 * hj)
        return J

    sol = scipy_opt.least_squares(fun, np.zeros(6 * n + 7), jac=jac, method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-11, max_nfev=200)
    return K0 + sol.x[6 * n:], float(sol.optimality), float(sol.cost)


def main():
    out = {}
    K, opt, cost = cfg1_scipy_optimum()
    out["cfg1"] = {"K": K.tolist(), "optimality": opt, "cost": cost}
    sys.path.insert(0, os.path.dirname(HERE))
    import test_oracle_imu as t
    st = t.polished_vi60_optimum()
    out["vi60"] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in st.items()}
    with open(os.path.join(HERE, "scipy_optima.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
