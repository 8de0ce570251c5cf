"""Writes raytracing_amd/csrc/rt_fresnel_tables.h: Chebyshev coefficients for the Fresnel integrals S(x), C(x) and their
auxiliary functions f(x), g(x) (Abramowitz & Stegun 7.3.5-6: C = 1/2 + f sin(pi x^2/2) - g cos(pi x^2/2),
S = 1/2 - f cos(pi x^2/2) - g sin(pi x^2/2)), computed here with mpmath at 50 digits.

  x <  X0         C(x) = x   Pc(u),  S(x) = x^3 Ps(u),          u = x^4 on [0, X0^4]
  x >= X0         f(x) = Phi(u) / (pi x),  g(x) = Gam(u) / (pi^2 x^3),  u = x^-4, on the intervals of X_EDGES

Each series is cut where its coefficients fall below what a double can hold of the function; the script then evaluates
the tables in double precision exactly as the header does (Clenshaw) and prints the largest error against mpmath.

Usage: python scripts/make_fresnel_tables.py            (rewrites the header; takes a minute)
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
PI = mp.pi
X0 = mp.mpf("1.8")
X_EDGES = [mp.mpf("1.8"), mp.mpf("2.8"), mp.mpf("5.0")]  # the last interval runs to infinity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fg(x):
    z = PI * x * x / 2
    s, c = mp.fresnels(x), mp.fresnelc(x)
    half = mp.mpf(1) / 2
    return (half - s) * mp.cos(z) - (half - c) * mp.sin(z), (half - c) * mp.cos(z) + (half - s) * mp.sin(z)


def phi(u):
    if u == 0:
        return mp.mpf(1)
    x = u ** (-mp.mpf(1) / 4)
    return PI * x * fg(x)[0]


def gam(u):
    if u == 0:
        return mp.mpf(1)
    x = u ** (-mp.mpf(1) / 4)
    return PI * PI * x ** 3 * fg(x)[1]


def pc(u):  # C(x) / x
    if u == 0:
        return mp.mpf(1)
    x = u ** (mp.mpf(1) / 4)
    return mp.fresnelc(x) / x


def ps(u):  # S(x) / x^3
    if u == 0:
        return PI / 6
    x = u ** (mp.mpf(1) / 4)
    return mp.fresnels(x) / x ** 3


def chebyshev(fun, a, b, n):
    nodes = [mp.cos(PI * (k + mp.mpf(1) / 2) / n) for k in range(n)]
    vals = [fun((a + b) / 2 + (b - a) / 2 * t) for t in nodes]
    coef = [2 * sum(vals[k] * mp.cos(PI * j * (k + mp.mpf(1) / 2) / n) for k in range(n)) / n for j in range(n)]
    coef[0] /= 2
    return coef


def fitted(fun, a, b, weight, target=mp.mpf("1e-17")):
    """Coefficients on [a, b], cut at the first index from which all that follow are below target / weight."""
    coef = chebyshev(fun, a, b, 48)
    keep = len(coef)
    while keep > 1 and abs(coef[keep - 1]) * weight < target:
        keep -= 1
    return coef[:keep]


def clenshaw(coef, a, b, u):  # what the header does, in double precision
    t = (2.0 * u - (a + b)) / (b - a)
    b1 = np.zeros_like(u)
    b2 = np.zeros_like(u)
    for c in coef[:0:-1]:
        b1, b2 = c + 2.0 * t * b1 - b2, b1
    return coef[0] + t * b1 - b2


def main():
    tables = []
    u0 = X0 ** 4
    tables.append(("PC", mp.mpf(0), u0, fitted(pc, mp.mpf(0), u0, X0)))
    tables.append(("PS", mp.mpf(0), u0, fitted(ps, mp.mpf(0), u0, X0 ** 3)))
    for k, lo in enumerate(X_EDGES):
        hi = X_EDGES[k + 1] if k + 1 < len(X_EDGES) else None
        a = mp.mpf(0) if hi is None else hi ** -4
        b = lo ** -4
        tables.append((f"PHI{k}", a, b, fitted(phi, a, b, 1 / (PI * lo))))
        tables.append((f"GAM{k}", a, b, fitted(gam, a, b, 1 / (PI ** 2 * lo ** 3))))
    lines = ["// Chebyshev coefficients of the Fresnel integrals and their auxiliary functions.  Written by",
             "// scripts/make_fresnel_tables.py (mpmath, 50 digits); do not edit by hand.",
             "#pragma once",
             f"#define RT_FRESNEL_X0 {mp.nstr(X0, 20)}",
             f"#define RT_FRESNEL_X1 {mp.nstr(X_EDGES[1], 20)}",
             f"#define RT_FRESNEL_X2 {mp.nstr(X_EDGES[2], 20)}"]
    for name, a, b, coef in tables:
        lines.append(f"#define RT_FRESNEL_{name}_N {len(coef)}")
        lines.append(f"#define RT_FRESNEL_{name}_A {mp.nstr(a, 20)}")
        lines.append(f"#define RT_FRESNEL_{name}_B {mp.nstr(b, 20)}")
        lines.append(f"#define RT_FRESNEL_{name}_COEF {{ \\")
        lines.extend(f"    {float(c)!r}, \\" for c in coef)
        lines.append("}")
    path = os.path.join(ROOT, "raytracing_amd", "csrc", "rt_fresnel_tables.h")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("wrote", path, {name: len(coef) for name, _, _, coef in tables})

    # the tables in double precision against mpmath
    named = {name: (float(a), float(b), [float(c) for c in coef]) for name, a, b, coef in tables}
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(0, float(X0), 400), rng.uniform(float(X0), 8, 600), 10 ** rng.uniform(0.9, 3, 200),
                         [0.0, 1e-8, float(X0) - 1e-12, float(X0), 2.8, 5.0, 1e3]])
    worst = {"S": 0.0, "C": 0.0, "f": 0.0, "g": 0.0}
    for x in xs:
        xm = mp.mpf(float(x))
        s_true, c_true = mp.fresnels(xm), mp.fresnelc(xm)
        xa = np.array([float(x)])
        if x < float(X0):
            u = xa ** 4
            c = xa * clenshaw(named["PC"][2], named["PC"][0], named["PC"][1], u)
            s = xa ** 3 * clenshaw(named["PS"][2], named["PS"][0], named["PS"][1], u)
        else:
            k = 0 if x < 2.8 else (1 if x < 5.0 else 2)
            u = 1.0 / (xa * xa) ** 2
            f = clenshaw(named[f"PHI{k}"][2], named[f"PHI{k}"][0], named[f"PHI{k}"][1], u) / (np.pi * xa)
            g = clenshaw(named[f"GAM{k}"][2], named[f"GAM{k}"][0], named[f"GAM{k}"][1], u) / (np.pi ** 2 * xa ** 3)
            f_true, g_true = fg(xm)
            worst["f"] = max(worst["f"], abs(float(f[0] - f_true)))
            worst["g"] = max(worst["g"], abs(float(g[0] - g_true)))
            z = 0.5 * np.pi * xa * xa
            c = 0.5 + f * np.sin(z) - g * np.cos(z)
            s = 0.5 - f * np.cos(z) - g * np.sin(z)
        worst["S"] = max(worst["S"], abs(float(mp.mpf(float(s[0])) - s_true)))
        worst["C"] = max(worst["C"], abs(float(mp.mpf(float(c[0])) - c_true)))
    print("largest absolute errors in double precision:", worst)


if __name__ == "__main__":
    main()
