"""fp32 emulation of common.h::erf_as / gelu_erf against scipy's erf (documents the bound quoted in common.h)."""
import numpy as np
from scipy.special import erf
f = np.float32
x = np.linspace(-12, 12, 4_000_001).astype(f)


def erf_as(x):
    ax = np.abs(x).astype(f)
    t = (f(1) / (f(1) + f(0.3275911) * ax)).astype(f)
    p = f(1.061405429)
    for c in (-1.453152027, 1.421413741, -0.284496736, 0.254829592):
        p = (p * t + f(c)).astype(f)
    p = (p * t).astype(f)
    return np.copysign((f(1) - p * np.exp((-ax * ax).astype(f)).astype(f)).astype(f), x).astype(f)


print("erf  max abs err", np.abs(erf_as(x) - erf(x.astype(np.float64))).max())
g_ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
g = (f(0.5) * x * (f(1) + erf_as((x * f(0.70710678)).astype(f)))).astype(f)
print("gelu max abs err", np.abs(g - g_ref).max())
