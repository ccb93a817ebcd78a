"""CPU: the pre-filter of the shared sweep on the matrix cores (bbq_mfma_kernels.hip: z_threshold, row_constants, acc_init - the
threshold on the integer dot product that the accumulator is initialised with), restated in numpy with f32 arithmetic, must never
reject a (row, query) pair whose exact f32 score beats the threshold - for all similarities, both query scalings (S = 8 for query
values <= 15, S = 1 up to 127), ordinary and hostile magnitudes, a v_rcp_f32 that is off by an ulp either way, and a caller-supplied
quantizedComponentSum that is NOT the sum of the query values (the slack must not depend on it being consistent)."""
import numpy as np
import pytest

import orclib as O

FBS = 1.0 / 15.0
F32 = np.float32


def key_of(s32):
    b = np.asarray(s32, np.float32).view(np.uint32).astype(np.int64)
    return np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)


def bf16_trunc(x):
    b = np.asarray(x, np.float64).astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
    return b.view(np.float32).astype(np.float64)


def fma32(a, b, c):
    """fmaf on f32 operands: the product is exact in f64, one rounding of the sum to f64 and one to f32 (within half an ulp of fmaf)"""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def z_threshold(th, qc, cdp, sim, one_bit):
    """bbq_mfma_kernels.hip z_threshold: exact f32 score > th  =>  z > zmin; -inf accepts everything"""
    big = np.finfo(np.float64).max
    qadd = qc[2]
    if np.isnan(th):
        return -big
    if sim == 1:
        if th < 0:
            return -big
        z = (2.0 * th - 1.0) - (qadd - cdp)
    elif sim == 2:
        if one_bit:
            t = th - 1.0 if th >= 1.0 else (1.0 - 1.0 / th if th > 0 else -big)
        else:
            t = (th - 1.0) * FBS if th >= 1.0 else ((1.0 - 1.0 / th) * FBS if th > 0 else -big)
        if t == -big:
            return -big
        z = t - (qadd - cdp)
    else:
        if not th > 0:
            return -big
        z = qadd + 1.0 - 1.0 / th
    if not abs(z) <= big:
        return -big
    return z - 1e-9 * (abs(z) + abs(qadd) + abs(cdp) + 1.0)


BIAS = 0x4B400000
BIAS_F = F32(12582912.0)
MAG_LIMIT = F32(4000000.0)


def prefilter_pass(qcdist, lower, upper, add, x1, qc, cdp, dim, sim, one_bit, theta_score, qsum, S, rcp_ulps=0):
    """bbq_mfma_kernels.hip: the prologue's per-query constants, row_constants(), acc_init() and the final compare, one query against
    all rows.  rcp_ulps moves the reciprocal by that many ulps (v_rcp_f32 is good to one)."""
    with np.errstate(all="ignore"):
        ay = qc[0]
        ly = (qc[1] - qc[0]) if one_bit else (qc[1] - qc[0]) * FBS
        y1 = qc[3]
        cs = 2.0 if sim == 0 else 1.0
        beta = cs * ly
        zt = z_threshold(float(theta_score), qc, cdp, sim, one_bit)
        A = -float(S) * (zt / beta)
        A = min(max(A, -3.0e38), 3.0e38) if A == A else 3.0e38
        qk = [F32(A), F32(-float(S) * (ay / ly)), F32(-float(S) * y1), F32(-float(S) / beta)]
        g0 = F32(np.abs(qk[0]) * F32(1.0000002)) if np.abs(qk[0]) < F32(1.0e38) else F32(0.0)
        g1 = F32(S) * F32(abs(ay / ly) * 1.000001)
        g2 = F32(S) * F32(abs(y1) * 1.000001)
        g3 = F32(S) * F32(1.0 / beta * 1.000001)
        g4 = F32(S) * F32(qsum * 1.000001)
        D = F32(dim)
        lxf = (upper - lower).astype(F32)
        alf, addf, x1f = lower.astype(F32), add.astype(F32), x1.astype(F32)
        r0 = (F32(1.0) / lxf).astype(F32)
        if rcp_ulps:
            r0 = (r0.view(np.int32) + np.int32(rcp_ulps)).view(F32)
        rho = (alf * r0).astype(F32)
        r1 = -fma32(rho, D, x1f)
        r2 = -rho
        r3 = ((addf if sim == 0 else -addf) * r0).astype(F32)
        mag = fma32(g0, np.abs(r0), fma32(g1, fma32(np.abs(rho), D, np.abs(x1f)), fma32(g2, np.abs(rho), fma32(g3, np.abs(r3), g4))))
        ok = (lxf > 0) & (r0 >= F32(1.0e-6)) & (mag < MAG_LIMIT)
        K = np.where(ok, BIAS_F + np.ceil(fma32(mag, F32(9.5367431640625e-07), F32(4.0))), F32(np.inf)).astype(F32)
        r0, r1, r2, r3 = [np.where(ok, v, F32(0.0)).astype(F32) for v in (r0, r1, r2, r3)]
        init = fma32(qk[0], r0, fma32(qk[1], r1, fma32(qk[2], r2, fma32(qk[3], r3, K))))
        final = init.view(np.int32).astype(np.int64) + np.int64(S) * qcdist.astype(np.int64)
        # the kernel's i32 accumulator wraps; it never does for a finite start value (S * qcDist < 2^21)
        final = ((final + 2**31) % 2**32) - 2**31
        return final > BIAS, ok


@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("qb", [1, 4, 7])
def test_prefilter_never_rejects_a_candidate(sim, qb):
    rng = np.random.default_rng(31 * sim + qb)
    n, dim = 150000, 128
    codes = rng.integers(0, 256, size=(n, dim // 8), dtype=np.uint8)
    pop = np.unpackbits(codes, axis=1).sum(axis=1).astype(np.float64)
    qq = rng.integers(0, 1 << qb, dim).astype(np.uint8)
    qsum = float(qq.sum())
    S = 8 if qq.max() <= 15 else 1
    for flavour in range(3):
        corr = np.zeros((n, 4))
        if flavour == 0:      # what a real index looks like
            corr[:, 0] = -0.04 * (0.9 + 0.2 * rng.random(n))
            corr[:, 1] = 0.04 * (0.9 + 0.2 * rng.random(n))
            corr[:, 2] = 1e-2 * rng.standard_normal(n) if sim else np.abs(rng.standard_normal(n))
        else:                 # magnitudes all over the place, zeros, sign flips, upper below lower
            scale = 10.0 ** rng.uniform(-8, 4, n)
            corr[:, 0] = rng.standard_normal(n) * scale
            corr[:, 1] = rng.standard_normal(n) * scale * 10.0 ** rng.uniform(-2, 2, n)
            corr[:, 2] = rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 4, n)
            corr[::97, 0] = 0
            corr[::89, 2] = 0
            corr[::83, 1] = corr[::83, 0]                                # zero width
            corr[1::83, 1] = corr[1::83, 0] * (1 + 2.0 ** -30)           # a width lost in the f32 image of the interval ends
        corr[:, 3] = pop
        # flavour 2: the caller's quantizedComponentSum of the QUERY is not the sum of its values
        y1 = qsum if flavour < 2 else qsum * 0.25
        qc = np.array([-0.15, 0.148, -0.0028 if sim else 0.7, y1])
        cdp = 0.0009
        one_bit = qb == 1
        d, s64, s32 = O.score_all(codes, corr, dim, qq, qc, qb, sim, cdp)
        ok = ~np.isnan(s32)
        for quantile in (0.5, 0.99, 0.9999):
            theta_score = np.float32(np.quantile(s32[ok], quantile))
            wins = ok & (key_of(s32) > key_of(np.array([theta_score]))[0])
            for ulps in (0, 1, -1):
                passed, ordinary = prefilter_pass(d.astype(np.float64), corr[:, 0], corr[:, 1], corr[:, 2], pop, qc, cdp, dim, sim, one_bit,
                                                  theta_score, qsum, S, ulps)
                assert passed[wins].all(), "the pre-filter rejected %d winning pairs (flavour %d, quantile %g, rcp %+d ulp)" % (
                    (~passed[wins]).sum(), flavour, quantile, ulps)
            if flavour == 0:
                assert ordinary.all()                   # no real row takes the pass-everything exit
                if quantile == 0.9999:                  # and it is a filter (EUCLIDEAN rows with a negative denominator are beyond z-space: a few %)
                    assert passed[~wins & ok].mean() < (0.05 if sim == 0 else 2e-4)
        # threshold key 0 (nothing known yet): every pair passes, weird rows included
        passed, _ = prefilter_pass(d.astype(np.float64), corr[:, 0], corr[:, 1], corr[:, 2], pop, qc, cdp, dim, sim, one_bit, np.float32(np.nan), qsum, S)
        assert passed.all()


def test_prefilter_slack_is_small_on_a_real_shape():
    """768-d, 4-bit query, corrections of a really quantized COSINE index: the start value's slack is a few accumulator units
    (eighths of a qcDist unit), i.e. the test on the integer is as sharp as the f64 score itself"""
    rng = np.random.default_rng(5)
    n, dim, qb, sim = 20000, 768, 4, 1
    codes = rng.integers(0, 256, size=(n, dim // 8), dtype=np.uint8)
    pop = np.unpackbits(codes, axis=1).sum(axis=1).astype(np.float64)
    qq = rng.integers(0, 16, dim).astype(np.uint8)
    corr = np.zeros((n, 4))
    corr[:, 0] = -0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 1] = 0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 2] = 1e-4 * rng.standard_normal(n)
    corr[:, 3] = pop
    qc = np.array([-0.15, 0.148, -0.0028, float(qq.sum())])
    d, s64, s32 = O.score_all(codes, corr, dim, qq, qc, qb, sim, 0.0009)
    theta_score = np.float32(np.quantile(s32, 0.999))
    wins = key_of(s32) > key_of(np.array([theta_score]))[0]
    passed, ordinary = prefilter_pass(d.astype(np.float64), corr[:, 0], corr[:, 1], corr[:, 2], pop, qc, 0.0009, dim, sim, False, theta_score,
                                      float(qq.sum()), 8)
    assert ordinary.all() and passed[wins].all()
    assert (passed & ~wins).sum() <= max(3, int(0.05 * wins.sum()))
