#!/usr/bin/env python3
"""Coefficients of the exp() used by the trace-gradient kernel (kmat.hip, gpx_exp):
exp(r) ~ 1 + r + r^2 q(r) on |r| <= ln2/2, q of degree 9 interpolated at Chebyshev nodes
in 50-digit arithmetic (near-minimax), rounded to double; prints the coefficients of
1, r, ..., r^11 and the worst relative error over a fine grid."""
import mpmath as mp
mp.mp.dps = 50
a = mp.log(2) / 2 * mp.mpf('1.0001')
n = 10                                            # degree-9 q
nodes = [a * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
f = lambda r: (mp.exp(r) - 1 - r) / r ** 2
A = mp.matrix(n, n)
b = mp.matrix(n, 1)
for i, x in enumerate(nodes):
    for j in range(n):
        A[i, j] = x ** j
    b[i] = f(x)
q = mp.lu_solve(A, b)
coef = [mp.mpf(1), mp.mpf(1)] + [q[j] for j in range(n)]
dbl = [float(c) for c in coef]
worst = 0
for k in range(-2000, 2001):
    r = mp.log(2) / 2 * k / 2000
    p = mp.mpf(0)
    for c in reversed(dbl):
        p = p * r + mp.mpf(c)
    worst = max(worst, abs(p / mp.exp(r) - 1))
for i, c in enumerate(dbl):
    print("c%-2d = %s  /* %.17g */" % (i, float.hex(c), c))
print("max relative error of the polynomial (exact arithmetic): %.3g" % float(worst))
