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


def gelu_erfc5(x):
    """common.h::gelu_erfc5 (the ring GEMM's bf16 epilogue): Phi(x) = 0.5 erfc(-x / sqrt2), erfc(z) = 2^(z P5(z)) on 0 <= z <= 4"""
    z = np.minimum(np.abs(x) * f(0.70710678118654752440), f(4)).astype(f)
    p = (f(-0.00303853428) * z + f(0.0299264971)).astype(f)
    for c in (-0.149057642, -0.918339764, -1.62791028):
        p = (p * z + f(c)).astype(f)
    q = np.exp2((p * z - f(1)).astype(f)).astype(f)
    return (x * np.where(x > 0, f(1) - q, q)).astype(f)


g5 = gelu_erfc5(x)
print("gelu_erfc5 max abs err", np.abs(g5 - g_ref).max(), " max rel err where |gelu| > 1e-6:",
      (np.abs(g5 - g_ref) / np.maximum(np.abs(g_ref), 1e-30))[np.abs(g_ref) > 1e-6].max())
