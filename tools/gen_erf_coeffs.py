"""Fit the fp32 erf() specification used by both the oracle and the HIP kernels
(bf_erf): three ranges, Chebyshev least-squares fits converted to monomials.
Prints C initialisers and the max error of an emulated-fp32 evaluation."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf, erfc

f32 = np.float32


def cheb_fit_mono(fn, lo, hi, deg, n=4000):
    # fit in the mapped variable u in [-1,1], convert to monomial in (x - c)/h
    k = np.arange(n)
    u = np.cos(np.pi * (k + 0.5) / n)
    x = 0.5 * (hi - lo) * u + 0.5 * (hi + lo)
    c = C.chebfit(u, fn(x), deg)
    mono_u = C.cheb2poly(c)
    # u = (x - mid)/half  -> polynomial in t = x - mid : coefficients mono_u[k] / half^k
    half = 0.5 * (hi - lo)
    return np.array([mono_u[k] / half ** k for k in range(deg + 1)]), 0.5 * (hi + lo)


def horner32(coef, t):
    acc = np.full_like(t, f32(coef[-1]), dtype=f32)
    for c in coef[-2::-1]:
        acc = (acc.astype(np.float64) * t.astype(np.float64) + np.float64(f32(c))).astype(f32)   # fma emulation
    return acc


# range A: |x| < 0.8 : erf(x) = x * PA(x^2)
def fa(z):
    x = np.sqrt(z)
    return erf(x) / x


ca, mid_a = cheb_fit_mono(fa, 1e-12, 0.64, 7)
# express as polynomial in z directly (shift back): poly in (z - mid) -> expand
pa = P.Polynomial(ca)(P.Polynomial([-mid_a, 1.0])).coef
# range B: 0.8 <= a < 1.6 : erf(a) = PB(a - 1.2)
cb, mid_b = cheb_fit_mono(erf, 0.8, 1.6, 9)
# range C: 1.6 <= a < 4 : erf(a) = 1 - exp(PC(a - 2.8)),  PC ~ log(erfc(a))
cc, mid_c = cheb_fit_mono(lambda a: np.log(erfc(a)), 1.6, 4.0, 10)

for name, c in (("kErfA", pa), ("kErfB", cb), ("kErfC", cc)):
    print("static const float %s[%d] = {%s};" % (name, len(c), ", ".join("%.9ef" % v for v in c)))
print("mids", mid_b, mid_c)

x = np.linspace(0, 0.8, 200001, dtype=f32)[1:-1]
z = (x.astype(np.float64) ** 2).astype(f32)
ya = (horner32(pa, z).astype(np.float64) * x).astype(f32)
ea = np.abs(ya.astype(np.float64) - erf(x.astype(np.float64)))
print("A max abs err", ea.max(), "ulp@", (ea / np.spacing(np.abs(ya))).max())
x = np.linspace(0.8, 1.6, 200001, dtype=f32)[:-1]
yb = horner32(cb, (x - f32(mid_b)).astype(f32))
eb = np.abs(yb.astype(np.float64) - erf(x.astype(np.float64)))
print("B max abs err", eb.max(), "ulp", (eb / np.spacing(yb)).max())
x = np.linspace(1.6, 4.0, 200001, dtype=f32)[:-1]
q = horner32(cc, (x - f32(mid_c)).astype(f32))
yc = (1.0 - np.exp(q.astype(np.float64))).astype(f32)
ec = np.abs(yc.astype(np.float64) - erf(x.astype(np.float64)))
print("C max abs err", ec.max(), "ulp", (ec / np.spacing(yc)).max())
