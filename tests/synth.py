"""Seeded synthetic matrices over Z_p for the parity sweeps (SURVEY.md 8d "synthetic inputs")."""
import random


def small_valued(seed, p, mmax=7, nmax=6, density=0.6):
    """Small matrices with coefficients from a tiny set: provokes repeated
    coefficients per column/row and <ab|b;a|.> triangles (ProgramGen paths)."""
    rng = random.Random(seed)
    m, n = rng.randint(2, mmax), rng.randint(2, nmax)
    vals = [1, -1, 2, -2, 3, 4, 6, -6, 12, pow(2, -1, p), pow(3, -1, p)]
    rows = []
    for _ in range(m):
        r = {}
        for j in range(n):
            if rng.random() < density:
                r[j] = rng.choice(vals) % p
        rows.append(r)
    return m, n, rows


def sweep(seed, p, m, n, density=0.25, unit_frac=0.8):
    """row density 25 %, values from {1,-1} (80 %) and {2,-2,1/2,-1/2} (20 %)."""
    rng = random.Random(seed)
    h = pow(2, -1, p)
    rows = []
    for _ in range(m):
        r = {}
        for j in range(n):
            if rng.random() < density:
                r[j] = (rng.choice([1, p - 1]) if rng.random() < unit_frac else rng.choice([2, p - 2, h, p - h]))
        rows.append(r)
    return m, n, rows


def to_csr(rows, p):
    rp, c, v = [0], [], []
    for r in rows:
        for j, x in sorted(r.items()):
            if x % p:
                c.append(j)
                v.append(x % p)
        rp.append(len(c))
    return rp, c, v
