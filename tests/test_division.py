"""The sweep's short division (dvo_slam_amd/csrc/pixel_math.h::divide2_correctly_rounded): one reciprocal refined by a Newton step, then
two fused remainder corrections per quotient instead of the IEEE division sequence.  The projection u = q.x / q.z must be the CORRECTLY
ROUNDED quotient (the reference divides, dvo_core/src/dense_tracking_impl.cpp:192 in its exact form; residuals are compared bit for bit),
whatever the hardware reciprocal returns within its 1-ulp tolerance.  Here the instruction sequence is replayed in exact rational
arithmetic with every operation rounded to float32 (round to nearest even, fused multiply-adds rounded once), for reciprocals one ulp
below, at and one ulp above the correctly rounded one, over the operand ranges the sweep sees -- and beyond.  On the device the same
sequence is compared with the division instruction over 2^32 operand pairs (scripts/ubench/div_check.hip)."""
from fractions import Fraction

import numpy as np


def rnd32(fr):
    """Fraction -> the nearest float32 (ties to even), as a Fraction; normal range only."""
    if fr == 0:
        return Fraction(0)
    sign = -1 if fr < 0 else 1
    m = abs(fr)
    e = m.numerator.bit_length() - m.denominator.bit_length()
    if Fraction(2) ** e > m:
        e -= 1
    assert Fraction(2) ** e <= m < Fraction(2) ** (e + 1) and -126 <= e <= 127
    ulp = Fraction(2) ** (e - 23)
    q, r = divmod(m, ulp)                       # q in [2^23, 2^24)
    q = int(q)
    if r * 2 > ulp or (r * 2 == ulp and q % 2 == 1):
        q += 1
    return sign * q * ulp


def fma(a, b, c):
    return rnd32(a * b + c)


def ulp_of(x):
    m = abs(x)
    e = m.numerator.bit_length() - m.denominator.bit_length()
    if Fraction(2) ** e > m:
        e -= 1
    return Fraction(2) ** (e - 23)


def short_division(n, d, r):
    r = fma(fma(-d, r, Fraction(1)), r, r)      # one Newton step
    q = rnd32(n * r)
    q = fma(fma(-d, q, n), r, q)
    return fma(fma(-d, q, n), r, q)


def as_fraction(x):
    return Fraction(float(np.float32(x)))


def test_short_division_is_correctly_rounded_for_any_reciprocal_within_one_ulp():
    rng = np.random.default_rng(12)
    cases = []
    # the sweep's range: depths 0.05 .. 100 m, numerators up to +-1e5 (quotients in and around [0, 32768))
    cases += [(rng.uniform(-1e5, 1e5), rng.uniform(0.05, 100.0)) for _ in range(6000)]
    # quotients close to the integers the bounds test and the floor look at
    for _ in range(3000):
        d = np.float32(rng.uniform(0.3, 10.0))
        k = rng.integers(0, 640)
        cases.append((np.float32(d) * np.float32(k + rng.choice([0.0, 1e-7, -1e-7, 0.5, 0.9999999])), d))
    # far outside: tiny and huge operands of either sign
    cases += [(rng.choice([-1, 1]) * 10.0 ** rng.uniform(-20, 20), rng.choice([-1, 1]) * 10.0 ** rng.uniform(-15, 15)) for _ in range(3000)]
    checked = 0
    for n, d in cases:
        n, d = as_fraction(n), as_fraction(d)
        if n == 0 or d == 0:
            continue
        want_exact = n / d
        if not (Fraction(2) ** -100 < abs(want_exact) < Fraction(2) ** 100):
            continue
        want = rnd32(want_exact)
        r0 = rnd32(1 / d)
        for step in (-1, 0, 1):
            r = r0 + step * ulp_of(r0)
            assert short_division(n, d, r) == want, (float(n), float(d), step)
            checked += 1
    assert checked > 30000
