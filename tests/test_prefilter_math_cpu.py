"""CPU: the pre-filter of the shared sweep on the matrix cores (bbq_mfma_kernels.hip: z_threshold, row_constants, start_values -
the threshold on the integer dot product that the accumulator is initialised with), restated in numpy with f32 arithmetic, must
never reject a (row, query) pair whose exact f32 score beats the threshold - for all similarities, both forms (FP6 x FP4 with
products of q, q / 2 or q / 4 for query values <= 15; int8 up to 127), ordinary and hostile magnitudes, a v_rcp_f32 that is off by
an ulp either way, either order of summing the start value's terms, and a caller-supplied quantizedComponentSum that is NOT the sum
of the query values (the slack must not depend on it being consistent)."""
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


# bbq_mfma_kernels.hip MfmaNum<FP>: int8 form (query values up to 127: the i32 accumulator holds the start value's bits) and FP form
# (query values <= 15 on FP6 x FP4: the f32 accumulator holds the value, in quarters of a qcDist unit)
FORMS = {"int8": dict(S=1.0, ulp=1.0, bias=12582912.0, mag_limit=4000000.0, pass_all=12582912.0 + 2097152.0),
         "fp": dict(S=0.25, ulp=0.0625, bias=786432.0, mag_limit=110000.0, pass_all=786432.0 + 131072.0)}


def fp_scale(max_q):
    """bbq_core.cpp enqueue_subbatch: products of q, q / 2 or q / 4 - the largest the query values leave room for in e2m3"""
    return 1.0 if max_q <= 3 else 0.5 if max_q <= 7 else 0.25



def prefilter_pass(qcdist, lower, upper, add, x1, qc, cdp, dim, sim, one_bit, theta_score, qsum, form, rcp_ulps=0, max_q=15, start="mfma"):
    """bbq_mfma_kernels.hip: the prologue's per-query constants, row_constants(), start_values() and the final compare, one query
    against all rows.  rcp_ulps moves the reciprocal by that many ulps (v_rcp_f32 is good to one).  start: how the start value's terms
    are summed - "mfma" as the kernel does (three v_mfma_f32_32x32x2_f32 on C = 0, an FMA per k: K x 1 first, the term with the row's
    popcount last), "k_first" a plain FMA chain onto K, "k_last" the products first - the slack must cover all of them.  Returns (passes, ordinary rows,
    flagged): flagged = the prologue hands the query to a sweep of its own (no usable threshold)."""
    N = FORMS[form]
    S, ulp, bias = (fp_scale(max_q) if form == "fp" else 1.0), F32(N["ulp"]), F32(N["bias"])
    with np.errstate(all="ignore"):
        ay = qc[0]
        ly = (qc[1] - qc[0]) if one_bit else (qc[1] - qc[0]) * FBS
        y1 = qc[3]
        cs = 2.0 if sim == 0 else 1.0
        beta = cs * ly
        zt = z_threshold(float(theta_score), qc, cdp, sim, one_bit)
        A = -S * (zt / beta)
        if not A < 1.0e30:
            return np.ones(len(qcdist), bool), np.ones(len(qcdist), bool), True
        qk = [F32(A) if A > -3.0e38 else F32(-3.0e38), F32(-S * (ay / ly)), F32(-S * y1), F32(-S / beta)]
        g0 = F32(np.abs(qk[0]) * F32(1.0000002)) if np.abs(qk[0]) < F32(1.0e30) else F32(0.0)
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
        # (|x1| enters the budget through a bound known before the row's popcount: the dimension for 1-bit rows)
        x1_bound = np.maximum(np.abs(x1f), D)
        mag = fma32(g0, np.abs(r0), fma32(g1, fma32(np.abs(rho), D, x1_bound), fma32(g2, np.abs(rho), fma32(g3, np.abs(r3), g4))))
        ok = (lxf > 0) & (mag < F32(N["mag_limit"]))
        K = np.where(ok, bias + ulp * np.ceil(fma32(mag, F32(9.5367431640625e-07) / ulp, F32(2.25))), F32(N["pass_all"])).astype(F32)
        r0, r1, r2, r3 = [np.where(ok, v, F32(0.0)).astype(F32) for v in (r0, r1, r2, r3)]
        if start == "mfma":       # the kernel: K x 1, q0 r0 | q2 r2, q3 r3 | q1 r1 (the term with the popcount last)
            init = fma32(qk[1], r1, fma32(qk[3], r3, fma32(qk[2], r2, fma32(qk[0], r0, fma32(F32(1.0), K, F32(0.0))))))
        elif start == "k_last":   # the products at their own grain first, one rounding at the binade's
            init = fma32(F32(1.0), K, fma32(qk[3], r3, fma32(qk[2], r2, fma32(qk[1], r1, fma32(qk[0], r0, F32(0.0))))))
        else:
            init = fma32(qk[0], r0, fma32(qk[1], r1, fma32(qk[2], r2, fma32(qk[3], r3, K))))
        if form == "int8":     # the i32 accumulator: the start value's bits + qcDist
            final = init.view(np.int32).astype(np.int64) + qcdist.astype(np.int64)
            passed = final > np.int64(F32(bias).view(np.int32))
            back = final - init.view(np.int32).astype(np.int64)
        else:                  # the f32 accumulator: every partial sum is a multiple of 1/16 inside one binade, so the sum is exact
            final = (init.astype(np.float64) + S * qcdist).astype(F32)
            passed = final > bias
            back = (final.astype(np.float64) - init.astype(np.float64)) / S
        # whatever passes must give qcDist back exactly (the survivors' exact scores are computed from it)
        assert (back[passed] == qcdist[passed]).all()
        return passed, ok, False


@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("qb,form", [(1, "fp"), (2, "fp"), (3, "fp"), (4, "fp"), (4, "int8"), (7, "int8")])
def test_prefilter_never_rejects_a_candidate(sim, qb, form):
    rng = np.random.default_rng(31 * sim + qb)
    n, dim = 150000, 128
    codes = rng.integers(0, 256, size=(n, dim // 8), dtype=np.uint8)
    pop = np.unpackbits(codes, axis=1).sum(axis=1).astype(np.float64)
    qq = rng.integers(0, 1 << qb, dim).astype(np.uint8)
    qsum = float(qq.sum())
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
            for ulps, start in ((0, "mfma"), (1, "mfma"), (-1, "mfma"), (0, "k_first"), (0, "k_last")):
                passed, ordinary, flagged = prefilter_pass(d.astype(np.float64), corr[:, 0], corr[:, 1], corr[:, 2], pop, qc, cdp, dim, sim,
                                                           one_bit, theta_score, qsum, form, ulps, int(qq.max()), start)
                assert passed[wins].all(), "the pre-filter rejected %d winning pairs (flavour %d, quantile %g, rcp %+d ulp, %s)" % (
                    (~passed[wins]).sum(), flavour, quantile, ulps, start)
            if flavour == 0:
                assert ordinary.all() and not flagged   # no real row takes the pass-everything exit
                if quantile == 0.9999:                  # and it is a filter (EUCLIDEAN rows with a negative denominator are beyond z-space: a few %)
                    # (1-bit queries over 128 dimensions: qcDist has a standard deviation of 4 and the slack is 1.25)
                    assert passed[~wins & ok].mean() < (0.05 if sim == 0 else 5e-3 if qb == 1 else 2e-4)
        # threshold key 0 (nothing known yet): the query is handed to a sweep of its own
        assert prefilter_pass(d.astype(np.float64), corr[:, 0], corr[:, 1], corr[:, 2], pop, qc, cdp, dim, sim, one_bit, np.float32(np.nan), qsum, form,
                              0, int(qq.max()))[2]


@pytest.mark.parametrize("form,qb", [("fp", 4), ("int8", 7)])
def test_prefilter_slack_is_small_on_a_real_shape(form, qb):
    """768-d, corrections of a really quantized COSINE index, each form with the query values it is used for: the start value's slack
    is three grains (3/4 of a qcDist unit in the FP form, 3 units in the int8 form, whose qcDist is eight times as wide), i.e. the
    test on the integer is as sharp as the f64 score itself"""
    rng = np.random.default_rng(5)
    n, dim, sim = 20000, 768, 1
    codes = rng.integers(0, 256, size=(n, dim // 8), dtype=np.uint8)
    pop = np.unpackbits(codes, axis=1).sum(axis=1).astype(np.float64)
    qq = rng.integers(0, 1 << qb, dim).astype(np.uint8)
    corr = np.zeros((n, 4))
    corr[:, 0] = -0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 1] = 0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 2] = 1e-4 * rng.standard_normal(n)
    corr[:, 3] = pop
    qc = np.array([-0.15, 0.148, -0.0028, float(qq.sum())])
    d, s64, s32 = O.score_all(codes, corr, dim, qq, qc, qb, sim, 0.0009)
    theta_score = np.float32(np.quantile(s32, 0.999))
    wins = key_of(s32) > key_of(np.array([theta_score]))[0]
    passed, ordinary, flagged = prefilter_pass(d.astype(np.float64), corr[:, 0], corr[:, 1], corr[:, 2], pop, qc, 0.0009, dim, sim, False,
                                               theta_score, float(qq.sum()), form, 0, int(qq.max()))
    assert ordinary.all() and not flagged and passed[wins].all()
    assert (passed & ~wins).sum() <= max(3, int(0.1 * wins.sum()))


def test_fp6_codes_of_the_staged_query_values():
    """fill_query_mfma_fp (bbq_core.cpp): e2m3 holds q / 2, q / 4, q / 8 exactly for q = 0..15; restated here and decoded again"""
    for q in range(16):
        for mult in (4, 2, 1, 8, 16):                # the value x 8 (x 2 and x 4 on top for query values up to 7 and up to 3)
            if (mult == 8 and q > 7) or (mult == 16 and q > 3):
                continue
            e8 = q * mult
            if e8 < 8:
                code = e8
            else:
                e = 1
                while e8 >= (8 << e):
                    e += 1
                assert (e8 & ((1 << (e - 1)) - 1)) == 0
                code = (e << 3) | ((e8 >> (e - 1)) - 8)
            assert 0 <= code < 32                    # no sign bit
            ex, m = code >> 3, code & 7
            val = m / 8.0 if ex == 0 else (1 + m / 8.0) * 2.0 ** (ex - 1)
            assert val == e8 / 8.0
