"""CPU: the f32 pre-filter of the shared sweep on the matrix cores (bbq_mfma_kernels.hip: z_threshold + the per-pair test), restated
in numpy with f32 arithmetic, must never reject a (row, query) pair whose exact f32 score beats the threshold - for all
similarities, compact and exact corrections, ordinary and hostile magnitudes, and a caller-supplied quantizedComponentSum that is
NOT the sum of the query values (the slack must not depend on it being consistent)."""
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


def prefilter_pass(qcdist, lower, upper, add, x1, qc, cdp, dim, sim, one_bit, compact, theta_score, qsum):
    with np.errstate(all="ignore"):
        if compact:
            al, au = bf16_trunc(lower), bf16_trunc(upper)
            aadd = np.asarray(add, np.float64).astype(np.float32).astype(np.float64)
            rel = 0.0078125 * (1.0 + 1.0 / 65536.0)
            ea, eu, eadd = np.abs(al) * rel + 1e-37, np.abs(au) * rel + 1e-37, np.abs(aadd) * 1.1920928955078125e-07 + 1e-37
        else:
            al, au, aadd = lower, upper, add
            ea, eu, eadd = np.abs(al) * 6e-8 + 1e-37, np.abs(au) * 6e-8 + 1e-37, np.abs(aadd) * 6e-8 + 1e-37
        ay = qc[0]
        ly = (qc[1] - qc[0]) if one_bit else (qc[1] - qc[0]) * FBS
        y1 = qc[3]
        AYmax, LYmax = F32(abs(ay) * 1.000001), F32(abs(ly) * 1.000001)
        Y1max = F32((abs(y1) + qsum) * 1.000001)
        D = float(dim)
        lx = au - al
        R1 = al * D + lx * x1
        cs_d, ca_d = (2.0, -1.0) if sim == 0 else (1.0, 1.0)
        Fm = np.float64(AYmax) * np.abs(R1) + np.float64(LYmax) * np.float64(Y1max) * (np.abs(al) + np.abs(lx)) + np.abs(aadd) + 1.0
        slack = cs_d * (2e-6 * Fm + 1e-3 * (ea + eu) * (np.float64(AYmax) * D + 2.0 * np.float64(LYmax) * np.float64(Y1max))) + eadd * 1.001
        weird = ~(np.abs(R1) + np.abs(al) + np.abs(lx) + np.abs(aadd) < 1e30)
        slack32 = np.where(weird, np.float32(np.inf), slack.astype(np.float32) * F32(1.001) + F32(1e-30)).astype(np.float32)
        k0x, k0y, k0z, k0w = R1.astype(F32), (D - x1).astype(F32), x1.astype(F32), al.astype(F32)
        k1x, k1y = lx.astype(F32), (ca_d * aadd).astype(F32) + slack32
        k1z, k1w = (ea * 1.001).astype(F32), (eu * 1.001).astype(F32)   # unscaled: A and B below carry cs (1 or 2: commutes with every rounding)
        cs = F32(2.0 if sim == 0 else 1.0)
        ayq, lyq, y1q = F32(ay), F32(ly), F32(y1)
        ayz, lyz = cs * ayq, cs * lyq
        zth = F32(z_threshold(float(theta_score), qc, cdp, sim, one_bit))
        margin = F32(1e-6) * (np.abs(zth) + F32(1.0))
        qcf = qcdist.astype(F32)
        u = fma32(k1x, qcf, k0w * y1q)
        z = fma32(lyz, u, fma32(ayz, k0x, k1y))
        Ae = fma32(lyz, y1q - qcf, ayz * k0y)
        Be = fma32(lyz, qcf, ayz * k0z)
        zu = fma32(np.abs(Ae), k1z, fma32(np.abs(Be), k1w, z))
        return ~(zu <= (zth - margin)) | ~(np.abs(zu) <= F32(3.0e38))


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("qb", [1, 4, 7])
def test_prefilter_never_rejects_a_candidate(sim, qb, compact):
    rng = np.random.default_rng(31 * sim + qb + (100 if compact else 0))
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
        else:                 # magnitudes all over the place, zeros, sign flips
            scale = 10.0 ** rng.uniform(-8, 4, n)
            corr[:, 0] = rng.standard_normal(n) * scale
            corr[:, 1] = rng.standard_normal(n) * scale * 10.0 ** rng.uniform(-2, 2, n)
            corr[:, 2] = rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 4, n)
            corr[::97, 0] = 0
            corr[::89, 2] = 0
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
            passed = prefilter_pass(d.astype(np.float64), corr[:, 0], corr[:, 1], corr[:, 2], pop, qc, cdp, dim, sim, one_bit, compact,
                                    theta_score, qsum)
            assert passed[wins].all(), "the pre-filter rejected %d winning pairs (flavour %d, quantile %g)" % ((~passed[wins]).sum(), flavour, quantile)
            if flavour == 0 and quantile == 0.9999:   # and it is a filter: almost everything below the threshold is rejected
                assert passed[~wins & ok].mean() < 0.05
