"""Fit x * sigmoid(x (c0 + c1 x^2 + c2 x^4)) to the erf GELU on [-8, 8] (minimax via Nelder-Mead on the max error).
The coefficients used by csrc/common.h:gelu_logistic_fit are these times -log2(e)."""
import numpy as np
from scipy.special import erf, expit
from scipy.optimize import minimize

x = np.linspace(-8, 8, 200001)
gel = x * 0.5 * (1 + erf(x / np.sqrt(2)))


def model(p, x):
    x2 = x * x
    return x * expit(x * (p[0] + x2 * (p[1] + x2 * p[2])))


cost = lambda p: np.max(np.abs(model(p, x) - gel))
p = np.array([1.5957691216, 0.07135481283, 0.0])
for _ in range(8):
    p = minimize(cost, p, method="Nelder-Mead", options=dict(xatol=1e-13, fatol=1e-15, maxiter=40000, maxfev=40000)).x
print("c =", list(p), "max abs err", cost(p))
print("c * -log2(e) =", list(-p * np.log2(np.e)))
# float32 evaluation as the kernel does it
xf = x.astype(np.float32)
c = (-p * np.log2(np.e)).astype(np.float32)
xc = np.clip(xf, -8, 8); x2 = xc * xc
e = np.exp2((c[0] + x2 * (c[1] + x2 * c[2])) * xc).astype(np.float32)
print("f32 evaluation max abs err", np.max(np.abs(xf / (1 + e) - gel)))
