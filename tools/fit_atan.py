# degree-N fit of Q(z) = (r - atan r) / r^3, z = r^2 on [0, zmax], by interpolation at Chebyshev nodes in 60-digit arithmetic
from decimal import Decimal as D, getcontext
import math, sys
getcontext().prec = 70
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
zmax = D(sys.argv[2]) if len(sys.argv) > 2 else D("0.1716")   # tan(pi/8)^2 = 0.17157...
def Q(z):
    s = D(0); k = 0; term = D(1)
    while True:
        t = term / (2 * k + 3)
        s += t if k % 2 == 0 else -t
        if abs(t) < D(10) ** -65: break
        term *= z; k += 1
    return s
nodes = [zmax / 2 * (1 + D(math.cos(math.pi * (2 * i + 1) / (2 * (N + 1))))) for i in range(N + 1)]
vals = [Q(z) for z in nodes]
# Newton divided differences
c = vals[:]
for j in range(1, N + 1):
    for i in range(N, j - 1, -1):
        c[i] = (c[i] - c[i - 1]) / (nodes[i] - nodes[i - j])
# expand to monomials
poly = [D(0)] * (N + 1)
poly[0] = c[N]
deg = 0
for i in range(N - 1, -1, -1):
    # poly = poly * (z - nodes[i]) + c[i]
    new = [D(0)] * (N + 1)
    for k in range(deg + 1):
        new[k + 1] += poly[k]
        new[k] -= poly[k] * nodes[i]
    new[0] += c[i]
    poly = new; deg += 1
co = [float(p) for p in poly]
# error of the double-rounded polynomial (exact evaluation) over a fine grid
worst = D(0)
for i in range(2001):
    z = zmax * i / 2000
    pv = D(0)
    for k in range(N, -1, -1): pv = pv * z + D(co[k])
    r = z.sqrt()
    err = abs(pv - Q(z)) * r * z          # absolute error in atan(r)
    rel = err / (r if r > 0 else D(1))
    if rel > worst: worst = rel
print("degree", N, "zmax", zmax, "max relative error of atan(r)/r: %.3e" % float(worst))
print(", ".join("%.20e" % x for x in co))
