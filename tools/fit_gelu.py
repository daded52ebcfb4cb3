import numpy as np
from scipy.special import erfc
from numpy.polynomial import chebyshev as C
# gelu(x) = max(x,0) - t*Q(t), t=|x|, Q(t)=0.5*erfc(t/sqrt2). fit f(t)=t*Q(t) on [0,c] directly (f(0)=0): gelu = max(x,0) - f(min(|x|,c))
def remez_like(f, a, b, deg, iters=40):
    # weighted least squares on chebyshev nodes then iterate reweighting (Lawson) for minimax approx
    n = 4000
    xs = 0.5*(a+b) + 0.5*(b-a)*np.cos(np.pi*(np.arange(n)+0.5)/n)
    y = f(xs); w = np.ones(n)
    for _ in range(iters):
        V = np.vander(xs, deg+1, increasing=True)
        coef, *_ = np.linalg.lstsq(V*w[:,None], y*w, rcond=None)
        err = np.abs(V@coef - y)
        w = w*(err/err.max()+1e-3); w/=w.max()
    return coef, err.max()
f = lambda t: t*0.5*erfc(t/np.sqrt(2))
for c in (3.6, 3.8, 4.0, 4.2, 4.5):
    for deg in (7,8,9,10,11,12):
        coef, e = remez_like(f, 0, c, deg)
        # evaluate total error in float32 Horner incl. region beyond c
        t = np.linspace(0, 8, 200001).astype(np.float32)
        tc = np.minimum(t, np.float32(c))
        acc = np.float32(coef[-1])*np.ones_like(tc)
        for k in range(deg-1, -1, -1):
            acc = acc*tc + np.float32(coef[k])
        err = np.abs(acc.astype(np.float64) - f(t.astype(np.float64)))
        print(c, deg, f"fit {e:.2e} total(f32, incl tail) {err.max():.2e}")
print("----")
coef, e = remez_like(f, 0, 4.5, 10, iters=200)
print("c=4.5 deg=10 fit", e)
print(", ".join(f"{c:.9e}f" for c in coef))
x = np.linspace(-10, 10, 2000001).astype(np.float32)
t = np.minimum(np.abs(x), np.float32(4.5))
acc = np.float32(coef[-1])*np.ones_like(t)
for k in range(9, -1, -1):
    acc = acc*t + np.float32(coef[k])
g = np.maximum(x, 0) - acc
from scipy.special import erf
ref = 0.5*x.astype(np.float64)*(1+erf(x.astype(np.float64)/np.sqrt(2)))
print("max abs err of gelu", np.abs(g-ref).max(), "at", x[np.abs(g-ref).argmax()])

# ---- (r5) the same fit with f(0) = 0 PINNED: p(t) = t q(t), q of degree 9 fitted to f(t) / t = Q(t) with weight t (so
# that the residual minimised is still that of f).  gelu(0) = 0 exactly: the epilogue no longer writes -2.4e-5 into zero / padded
# columns, and the relative error stays bounded for small |x| (|p(t) - f(t)| <= t |q(t) - Q(t)|).
def remez_pinned(a, b, deg, iters=200):
    n = 4000
    xs = 0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * (np.arange(n) + 0.5) / n)
    y = f(xs); w = np.ones(n)
    for _ in range(iters):
        V = np.vander(xs, deg + 1, increasing=True)[:, 1:]          # no constant term
        coef, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
        err = np.abs(V @ coef - y)
        w = w * (err / err.max() + 1e-3); w /= w.max()
    return np.concatenate([[0.0], coef]), err.max()
coef0, e0 = remez_pinned(0, 4.5, 10)
print("---- pinned f(0) = 0: c=4.5 deg=10 fit", e0)
print(", ".join(f"{c:.9e}f" for c in coef0))
acc = np.float32(coef0[-1]) * np.ones_like(t)
for k in range(9, -1, -1):
    acc = acc * t + np.float32(coef0[k])
g0 = np.maximum(x, 0) - acc
print("max abs err of gelu (pinned)", np.abs(g0 - ref).max(), "at", x[np.abs(g0 - ref).argmax()], " gelu(0) =", g0[np.abs(x).argmin()])
small = np.abs(x) < 0.05
print("max rel err for |x| < 0.05:", np.abs((g0 - ref)[small & (x != 0)] / ref[small & (x != 0)]).max())
