"""Pivoted L D L^T (the semi-definite fallback) against the un-pivoted LL^T of the fast path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
ctx = ab.Context(0)
for n in (512, 2048, 4096):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 3)); A = G @ G.T / n + np.eye(n)
    ab.PivotedLDLT(A, ctx); ab.DenseFactor(A, ctx)
    t = time.perf_counter(); f = ab.PivotedLDLT(A, ctx); t1 = time.perf_counter() - t
    t = time.perf_counter(); g = ab.DenseFactor(A, ctx); t2 = time.perf_counter() - t
    b = rng.standard_normal((n, 8))
    t = time.perf_counter(); x = f.solve(b); t3 = time.perf_counter() - t
    print(f"n={n}: pivoted LDLT {t1*1e3:.1f} ms, LL^T {t2*1e3:.1f} ms (both incl. upload), pivoted solve of 8 rhs {t3*1e3:.1f} ms, "
          f"|x - x_llt| {np.abs(x - g.solve(b)).max():.1e}")
