#!/usr/bin/env python3
"""Coefficients of csrc/common.hpp gelu_erf(): Q(a) ~ -log2(erfc(a)) / a on [0, 5] as a degree-8 polynomial, weighted
minimax (Lawson iteration on a Chebyshev least-squares fit).  The weight is what an error in Q does to gelu(u) = u Phi(u),
u = -sqrt(2) a, against a budget of 3e-8 max(|gelu|, 1e-2): the fit spends its accuracy where the result is not negligible.
    python3 tools/fit_gelu_erfc.py        (needs mpmath; prints the fp32 bit patterns, highest degree last)
"""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as Pn
import mpmath as mp
mp.mp.dps = 50
def P_exact(a):   # -log2(erfc(a)) / a  (Q, so that P = a * Q and P(0) = 0)
    out=[]
    for x in a:
        x=mp.mpf(float(x))
        if x==0: out.append(float(2/mp.sqrt(mp.pi)/mp.log(2)))
        else: out.append(float(-mp.log(mp.erfc(x))/mp.log(2)/x))
    return np.array(out)
def fit(amax, deg, tol, floor, n=4000, iters=80):
    a = np.linspace(0, amax, n)
    q = P_exact(a)
    P = q*a
    E = 2.0**(-P)
    u = np.sqrt(2)*a
    gneg = 0.5*u*E
    sens = 0.5*u*E*np.log(2)*a          # d gelu / d q
    allowed = tol*np.maximum(gneg, floor)
    w = np.maximum(sens/allowed, 1e-9)
    xs = a/amax*2-1
    ww = w.copy()
    for it in range(iters):
        c = C.chebfit(xs, q, deg, w=ww)
        r = np.abs(C.chebval(xs, c)-q)*w
        ww = np.maximum(ww*(0.3+0.7*r/r.max()), 1e-14)
    c = C.chebfit(xs, q, deg, w=ww)
    r = np.abs(C.chebval(xs, c)-q)*w
    # convert to monomial in a
    pc = C.cheb2poly(c)                  # in xs
    # xs = a*(2/amax) - 1
    from numpy.polynomial import Polynomial as Poly
    pa = Poly(pc)(Poly([-1, 2/amax]))
    return pa.coef, r.max(), a[r.argmax()]
if __name__ == "__main__":
    coef, m, at = fit(5.0, 8, 3e-8, 1e-2)
    print("weighted max error / budget:", round(m, 3), "at a =", round(at, 3))
    for i, c in enumerate(coef.astype(np.float32)):
        print(f"a^{i}: {float(c):.9e}  0x{np.float32(c).view(np.uint32):08x}")
