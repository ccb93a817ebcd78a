"""CPU: the score upper bound used with compact corrections (bbq_kernels.hip: score_upper_bound), restated in numpy
float64 (same IEEE operations), must dominate the oracle's exact score for every row - over hostile magnitudes,
all similarities, 1-bit and multi-bit queries."""
import numpy as np
import pytest

import orclib as O

BF16_REL = 0.0078125 * (1.0 + 1.0 / 65536.0)
F32_REL = 2.0 ** -23
ABS_SLACK = 1e-37
ROUND_REL = 1e-9
FBS = 1.0 / 15.0


def bf16_trunc(x):
    b = np.asarray(x, np.float64).astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
    return b.view(np.float32).astype(np.float64)


def upper_bound(qc, al, au, aadd, x1, ay, ly, y1, qadd, cdp, dim, sim, one_bit):
    with np.errstate(all="ignore"):
        lx = au - al
        t1 = (al * ay) * dim
        t2 = (ay * lx) * x1
        t3 = (al * ly) * y1
        t4 = (lx * ly) * qc
        s = ((t1 + t2) + t3) + t4
        A = ay * (dim - x1) + ly * (y1 - qc)
        Bc = ay * x1 + ly * qc
        mag = np.abs(t1) + np.abs(t2) + np.abs(t3) + np.abs(t4) + abs(qadd) + np.abs(aadd) + abs(cdp) + 1.0
        es = np.abs(A) * (np.abs(al) * BF16_REL + ABS_SLACK) + np.abs(Bc) * (np.abs(au) * BF16_REL + ABS_SLACK)
        eadd = np.abs(aadd) * F32_REL + ABS_SLACK
        slop = ROUND_REL * (mag + np.abs(A) + np.abs(Bc))
        if sim == 0:
            e_low = ((qadd + aadd) - (2.0 * s)) - (2.0 * es + eadd + slop)
            den = 1.0 + e_low
            u = np.where(den > 0.0, 1.0 / den, np.inf)
            u = u + ROUND_REL * (u + 1.0)
        else:
            t_up = (((s + qadd) + aadd) - cdp) + (es + eadd + slop)
            if sim == 1:
                u = np.maximum((1.0 + t_up) / 2.0, 0.0)
            elif one_bit:
                u = np.where(t_up < 0.0, 1.0 / (1.0 - t_up), t_up + 1.0)
            else:
                u = np.where(t_up < 0.0, 1.0 / (1.0 - t_up / FBS), t_up / FBS + 1.0)
            u = u + ROUND_REL * (np.abs(u) + 1.0)
        u = np.where(mag < 1e290, u, np.nan)
    return u


@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("qb", [1, 4, 8])
def test_upper_bound_dominates_exact_score(sim, qb):
    rng = np.random.default_rng(7 * sim + qb)
    n, dim = 200000, 128
    codes = rng.integers(0, 256, size=(n, dim // 8), dtype=np.uint8)
    pop = np.unpackbits(codes, axis=1).sum(axis=1).astype(np.float64)
    corr = np.zeros((n, 4))
    scale = 10.0 ** rng.uniform(-12, 6, n)
    corr[:, 0] = rng.standard_normal(n) * scale
    corr[:, 1] = rng.standard_normal(n) * scale * 10.0 ** rng.uniform(-2, 2, n)
    corr[:, 2] = rng.standard_normal(n) * 10.0 ** rng.uniform(-10, 6, n)
    corr[::101, 0] = 0
    corr[::103, 1] = 0
    corr[::107, 2] = 0
    corr[:, 3] = pop
    qq = rng.integers(0, 1 << qb, dim).astype(np.uint8)
    for qc in (np.array([-0.15, 0.148, -0.0028, float(qq.sum())]), np.array([-30.0, 55.0, 4.0, float(qq.sum())]),
               np.array([1e-5, 2e-5, 0.0, float(qq.sum())])):
        cdp = 0.0009
        d, s64, s32 = O.score_all(codes, corr, dim, qq, qc, qb, sim, cdp)
        one_bit = qb == 1
        ly = (qc[1] - qc[0]) if one_bit else (qc[1] - qc[0]) * FBS
        al, au = bf16_trunc(corr[:, 0]), bf16_trunc(corr[:, 1])
        aadd = corr[:, 2].astype(np.float32).astype(np.float64)
        u = upper_bound(d.astype(np.float64), al, au, aadd, pop, qc[0], ly, qc[3], qc[2], cdp, float(dim), sim, one_bit)
        finite = ~np.isnan(u) & ~np.isnan(s64)
        assert finite.sum() > n * 0.9
        assert (u[finite] >= s64[finite]).all(), "upper bound below the exact score"
        # and it is tight enough to be useful on ordinary magnitudes
        u32 = u.astype(np.float32)
        ordinary = finite & (np.abs(corr[:, 0]) < 1.0) & (np.abs(corr[:, 1]) < 1.0) & (np.abs(s64) < 10)
        if ordinary.sum() > 100 and sim == 1:
            assert np.median(u32[ordinary] - s32[ordinary]) < 0.05

