"""Generate known-answer vectors for the third-party arithmetic the reference calls
(Calibu Project, Sophus exp/log, gravity) with mpmath at 50 digits.

Independent of oracle/ and of the HIP code: formulas are written out from SURVEY.md
section 9 (the published Calibu / Sophus definitions).  Run in the authoring container:
    python tests/golden/make_golden_math.py
writes tests/golden/math_kat.json (committed).  The reference has no golden vectors of
its own, so these pin the restatement against exact arithmetic,
not against the reference binary ("parity unpinned").
"""
import json
import os
import mpmath as mp

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


def F(x):
    return float(x)


def rnd(state):
    # small LCG so the file is reproducible without numpy
    state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) % (1 << 64)
    return mp.mpf(state[0] >> 11) / mp.mpf(1 << 53)


def project(model, ray, k):
    X, Y, Z = ray
    if model == "kb4":
        th = mp.atan2(mp.sqrt(X * X + Y * Y), Z)
        psi = mp.atan2(Y, X)
        r = th + k[4] * th**3 + k[5] * th**5 + k[6] * th**7 + k[7] * th**9
        return [k[0] * r * mp.cos(psi) + k[2], k[1] * r * mp.sin(psi) + k[3]]
    x, y = X / Z, Y / Z
    r = mp.sqrt(x * x + y * y)
    if model == "fov":
        w = k[4]
        if w * w > mp.mpf("1e-5"):
            m = 2 * mp.tan(w / 2)
            fac = m / w if r * r < mp.mpf("1e-5") else mp.atan(r * m) / (r * w)
        else:
            fac = mp.mpf(1)
    elif model == "poly2":
        fac = 1 + k[4] * r**2 + k[5] * r**4
    elif model == "poly3":
        fac = 1 + k[4] * r**2 + k[5] * r**4 + k[6] * r**6
    elif model == "rational6":
        fac = (1 + k[4] * r**2 + k[5] * r**4 + k[6] * r**6) / (1 + k[7] * r**2 + k[8] * r**4 + k[9] * r**6)
    else:
        fac = mp.mpf(1)
    return [fac * k[0] * x + k[2], fac * k[1] * y + k[3]]


def so3_exp(w):
    th = mp.sqrt(sum(c * c for c in w))
    if th == 0:
        return [mp.mpf(0)] * 3 + [mp.mpf(1)]
    s = mp.sin(th / 2) / th
    return [s * w[0], s * w[1], s * w[2], mp.cos(th / 2)]


def so3_log(q):
    n = mp.sqrt(q[0] ** 2 + q[1] ** 2 + q[2] ** 2)
    if n == 0:
        return [mp.mpf(0)] * 3
    c = 2 * mp.atan(n / q[3]) / n
    return [c * q[0], c * q[1], c * q[2]]


def hat(w):
    return mp.matrix([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def se3_exp(d):
    u, w = d[:3], d[3:]
    q = so3_exp(w)
    th = mp.sqrt(sum(c * c for c in w))
    O = hat(w)
    if th == 0:
        V = mp.eye(3)
    else:
        V = mp.eye(3) + (1 - mp.cos(th)) / th**2 * O + (th - mp.sin(th)) / th**3 * (O * O)
    t = V * mp.matrix(u)
    return q + [t[0], t[1], t[2]]


def se3_log(T):
    w = so3_log(T[:4])
    th = mp.sqrt(sum(c * c for c in w))
    O = hat(w)
    if th == 0:
        Vi = mp.eye(3)
    else:
        Vi = mp.eye(3) - O / 2 + (1 - th / (2 * mp.tan(th / 2))) / th**2 * (O * O)
    u = Vi * mp.matrix(T[4:])
    return [u[0], u[1], u[2]] + w


def main():
    st = [12345]
    out = {"project": [], "so3_exp": [], "so3_log": [], "se3_exp": [], "se3_log": [], "gravity": []}
    K = {
        "fov": [330, 331, 320.5, 240.25, mp.mpf("0.92")],
        "poly2": [400, 401, 320.5, 240.25, mp.mpf("-0.28"), mp.mpf("0.09")],
        "poly3": [400, 401, 320.5, 240.25, mp.mpf("-0.28"), mp.mpf("0.09"), mp.mpf("-0.012")],
        "kb4": [260, 261, 320.5, 240.25, mp.mpf("-0.012"), mp.mpf("0.004"), mp.mpf("-0.0015"), mp.mpf("0.0002")],
        "linear": [400, 401, 320.5, 240.25],
    }
    ids = {"fov": 0, "poly2": 1, "poly3": 2, "kb4": 3, "linear": 4}
    for name, k in K.items():
        k = [mp.mpf(v) for v in k]
        rays = []
        for _ in range(40):
            rays.append([(rnd(st) - 0.5) * 1.2, (rnd(st) - 0.5) * 0.9, mp.mpf("0.2") + rnd(st)])
        # branch boundaries: near-axis rays (fov r^2 ~ 1e-5 on either side, kb4 near axis)
        for eps in ["3.0e-3", "3.3e-3", "1e-4", "1e-6"]:
            rays.append([mp.mpf(eps), mp.mpf(eps) / 3, mp.mpf(1)])
        for ray in rays:
            pix = project(name, ray, k)
            out["project"].append({"model": ids[name], "ray": [F(v) for v in ray], "k": [F(v) for v in k], "pix": [F(v) for v in pix]})
    # a small-w fov case (w^2 < 1e-5 -> fac = 1)
    k = [mp.mpf(v) for v in [330, 331, 320.5, 240.25, "0.003"]]
    ray = [mp.mpf("0.1"), mp.mpf("-0.2"), mp.mpf("1.1")]
    out["project"].append({"model": 0, "ray": [F(v) for v in ray], "k": [F(v) for v in k], "pix": [F(v) for v in project("fov", ray, k)]})
    for scale in ["1e-12", "1e-9", "1e-5", "0.01", "0.5", "2.0", "3.0"]:
        for _ in range(6):
            w = [(rnd(st) - 0.5) * 2 * mp.mpf(scale) for _ in range(3)]
            q = so3_exp(w)
            out["so3_exp"].append({"w": [F(v) for v in w], "q": [F(v) for v in q]})
            out["so3_log"].append({"q": [F(v) for v in q], "w": [F(v) for v in so3_log(q)]})
            d = [(rnd(st) - 0.5) * 2 for _ in range(3)] + w
            T = se3_exp(d)
            out["se3_exp"].append({"d": [F(v) for v in d], "T": [F(v) for v in T]})
            out["se3_log"].append({"T": [F(v) for v in T], "d": [F(v) for v in se3_log(T)]})
    for _ in range(10):
        p, q = (rnd(st) - 0.5), (rnd(st) - 0.5)
        g = mp.mpf("9.8007")
        v = [-g * mp.cos(p) * mp.sin(q), g * mp.sin(p), -g * mp.cos(p) * mp.cos(q)]
        out["gravity"].append({"dir": [F(p), F(q)], "g": [F(x) for x in v]})
    # rational6 (added later: its own random stream, so that every vector above keeps its value)
    st2 = [987654321]
    k = [mp.mpf(v) for v in [400, 401, 320.5, 240.25, "0.12", "0.05", "0.004", "0.40", "-0.04", "0.002"]]
    rays = [[(rnd(st2) - 0.5) * 1.2, (rnd(st2) - 0.5) * 0.9, mp.mpf("0.2") + rnd(st2)] for _ in range(40)]
    rays += [[mp.mpf(eps), mp.mpf(eps) / 3, mp.mpf(1)] for eps in ["3.0e-3", "1e-6"]]
    for ray in rays:
        out["project"].append({"model": 5, "ray": [F(v) for v in ray], "k": [F(v) for v in k], "pix": [F(v) for v in project("rational6", ray, k)]})
    with open(os.path.join(HERE, "math_kat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
