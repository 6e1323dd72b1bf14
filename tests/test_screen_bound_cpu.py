"""The error model of the two-stage MATCH (csrc/match_screen.hip), checked in host arithmetic.

The screen ranks rows by w~ = sum_k f16(q_k) f16(d_k) - dd/2; the exact stage by the canonical distance
a = max(0, fmaf(-2, p, qq + dd)), p = f32 fmaf chain.  What makes the two-stage search exact is
    every row among a query's two nearest (by a) has w~ > T - mh_screen_margin(qq, Dmax)
for any T <= the second largest w~ over distinct rows.  Here: f16 rounding emulated with numpy.float16
(round to nearest even, as v_cvt_f16_f32), accumulation in f32 in k order (an upper bound on what the
matrix pipe's f32 accumulator loses), on SIFT-like rows, quantised rows whose rounding errors all point
the same way, unnormalised rows and rows inside f16's subnormal range."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth


def _margin(qq, dmax):
    return float(capi.load().mh_screen_margin(np.float32(qq), np.float32(dmax)))


def _chain_f32(q, d):
    """p[i, j] = f32 fmaf chain over k of q[i, k] * d[j, k] (product exact in f64, one rounding per step)."""
    s = np.zeros((q.shape[0], d.shape[0]), np.float32)
    for k in range(q.shape[1]):
        s = (q[:, k:k + 1].astype(np.float64) * d[None, :, k].astype(np.float64) + s.astype(np.float64)).astype(np.float32)
    return s


def _screen_f32(q, d, dd):
    """w~ with f16 operands and an f32 accumulator started at -dd/2."""
    q16, d16 = q.astype(np.float16).astype(np.float32), d.astype(np.float16).astype(np.float32)
    s = np.broadcast_to((-0.5 * dd)[None, :], (q.shape[0], d.shape[0])).astype(np.float32).copy()
    for k in range(q.shape[1]):
        s = (q16[:, k:k + 1] * d16[None, :, k] + s).astype(np.float32)       # f16 x f16 is exact in f32
    return s


def _cases():
    base, _, _ = synth.load_sift_fixture()
    rng = np.random.default_rng(5)
    unit = orclib.normalize(base[rng.choice(len(base), 600)])
    yield "sift-like unit rows", unit[:150], unit[150:]
    # every coordinate just below a rounding boundary of f16: all errors point the same way
    e = rng.integers(-9, -2, size=(500, 128))
    worst = ((1 + 2.0 ** -11 - 2.0 ** -20) * 2.0 ** e).astype(np.float32)
    yield "errors aligned", worst[:100], worst[100:]
    yield "unnormalised", (unit[:100] * rng.uniform(0.2, 4, (100, 1))).astype(np.float32), \
        (unit[100:] * rng.uniform(0.2, 4, (500, 1))).astype(np.float32)
    tiny = (unit * np.float32(3e-5)).astype(np.float32)          # coordinates in f16's subnormal range
    yield "subnormal range", tiny[:100], unit[100:]
    yield "zero rows", unit[:50], np.concatenate([np.zeros((10, 128), np.float32), unit[50:300]])


@pytest.mark.parametrize("name,q,d", list(_cases()), ids=[c[0] for c in _cases()])
def test_screen_error_is_inside_the_margin(name, q, d):
    qq = (q.astype(np.float64) ** 2).sum(1)
    dd64 = (d.astype(np.float64) ** 2).sum(1)
    dd = orclib.row_norms(d)
    dmax = float(np.sqrt(dd.max()))
    w = _chain_f32(q, d).astype(np.float64) - 0.5 * dd.astype(np.float64)[None, :]
    wt = _screen_f32(q, d, dd).astype(np.float64)
    err = np.abs(wt - w).max(1)
    # the margin is 2 E + slack: the error itself must stay below half of it
    half = np.array([0.5 * _margin(np.float32(x), dmax) for x in qq])
    assert np.all(err <= half), (name, float((err / half).max()))
    # and the model is not wildly loose where it matters (unit rows: within 4x of the worst case seen)
    if name == "errors aligned":
        assert (err / half).max() > 0.25


def test_two_nearest_are_always_emitted():
    """The property itself, on a DB with near-duplicates: the rows of the exact 2-NN (by canonical distance,
    ties included) all pass tau for a threshold taken from a sample of the rows."""
    base, _, _ = synth.load_sift_fixture()
    rng = np.random.default_rng(11)
    b = orclib.normalize(base[rng.choice(len(base), 60)])
    rows = [b[k] + 10.0 ** rng.uniform(-5, -3, (40, 1)) * rng.normal(size=(40, 128)) for k in range(60)]
    d = np.ascontiguousarray(np.maximum(np.concatenate(rows), 0), np.float32)
    d = d[rng.permutation(len(d))]
    q = np.ascontiguousarray(np.concatenate([b, b + rng.normal(0, 2e-4, b.shape)]), np.float32)
    dd, qq = orclib.row_norms(d), orclib.row_norms(q)
    dmax = float(np.sqrt(dd.max()))
    p = _chain_f32(q, d)
    a = np.maximum((np.float32(-2) * p.astype(np.float64) + (qq[:, None] + dd[None, :]).astype(np.float64)).astype(np.float32), 0)
    wt = _screen_f32(q, d, dd)
    sample = np.arange(3, len(d), 8)                         # pass A sees every 8th row here
    emitted_total = 0
    for i in range(len(q)):
        T = np.sort(wt[i, sample])[-2]
        tau = T - _margin(qq[i], dmax)
        emitted = wt[i] > tau
        emitted_total += int(emitted.sum())
        second = np.sort(a[i])[1]
        must = a[i] <= second                                # the two nearest and everything tied with them
        assert np.all(emitted[must]), i
    assert emitted_total < 0.2 * wt.size                     # and the screen still screens


# ---- pass C's record pruning (csrc/screen.h: screen_record_value / screen_record_bounds) in host arithmetic ----------
def _dots_f32(q, d):
    """Pass B's accumulators: f16 operands, f32 accumulation from ZERO in k order."""
    q16, d16 = q.astype(np.float16).astype(np.float32), d.astype(np.float16).astype(np.float32)
    s = np.zeros((q.shape[0], d.shape[0]), np.float32)
    for k in range(q.shape[1]):
        s = (q16[:, k:k + 1] * d16[None, :, k] + s).astype(np.float32)
    return s


# rows of a 32-row block that one lane holds (match_screen.hip: register r of half h = row (r & 3) + 8 (r >> 2) + 4 h)
_LANE_ROWS = [np.array([(r & 3) + 8 * (r >> 2) + 4 * h for r in range(16)]) for h in (0, 1)]


def _record(dots16, neg_hi, tau):
    """What pass B emits for a lane's 16 dots of a block: (row mask, value bits) or None."""
    L = capi.load()
    thr = np.float32(tau) - np.float32(neg_hi)
    top = np.float32(dots16.max())
    if not top > thr:
        return None
    mask = dots16 > thr
    return mask, int(L.mh_screen_record_value(float(top), float(thr)))


@pytest.mark.parametrize("kind", ["unit", "unnormalised", "tiny"])
def test_record_bounds_bracket_the_largest_screen_value_of_the_records_rows(kind):
    """hi >= max over the record's rows of (dot + -dd/2) >= lo: the two facts pass C's pruning rests on."""
    base, _, _ = synth.load_sift_fixture()
    rng = np.random.default_rng(21)
    unit = orclib.normalize(base[rng.choice(len(base), 32 * 12 + 40)])
    q, d = unit[:40], unit[40:]
    if kind == "unnormalised":
        d = (d * rng.uniform(0.5, 3.0, (len(d), 1))).astype(np.float32)
    if kind == "tiny":
        q = (q * np.float32(2e-3)).astype(np.float32)
    dd = orclib.row_norms(d)
    neg = (np.float32(-0.5) * dd).astype(np.float32)
    dmax = float(np.sqrt(dd.max()))
    dots = _dots_f32(q, d)
    n_blocks = len(d) // 32
    spread = max(float(neg[b * 32:(b + 1) * 32].max() - neg[b * 32:(b + 1) * 32].min()) for b in range(n_blocks))
    N = len(d)
    n_rec = 0
    for i in range(len(q)):
        w = dots[i].astype(np.float64) + neg.astype(np.float64)        # the rows' screen values
        tau = float(np.sort(w)[-12])                                   # lets about a dozen rows through
        for b in range(n_blocks):
            hi_b = float(neg[b * 32:(b + 1) * 32].max())
            for h in (0, 1):
                rows = b * 32 + _LANE_ROWS[h]
                rec = _record(dots[i, rows], hi_b, tau)
                if rec is None:
                    assert not np.any(w[rows] > tau + 1e-6)            # nothing above the threshold was left behind
                    continue
                mask, bits = rec
                assert np.all(mask[w[rows] > tau])                     # a superset of the rows above tau
                lo, hi = capi.screen_record_bounds(bits, b * 32 + 4 * h, tau, spread, N, dmax)
                w_max = float(w[rows].max())
                assert hi >= w_max, (kind, i, b, h, hi - w_max)
                assert lo <= w_max, (kind, i, b, h, w_max - lo)
                n_rec += 1
    assert n_rec > 100


def test_record_bounds_give_no_lower_bound_for_blocks_with_padding_or_without_information():
    lo, hi = capi.screen_record_bounds(0x3C00, 96, 0.25, 1e-7, 100, 1.0)    # rows 96..127 of a 100-row DB: padding inside
    assert lo == -np.inf and hi > 1.25 - 1e-3
    lo, hi = capi.screen_record_bounds(0x7C00, 0, 0.25, 1e-7, 4096, 1.0)    # value +inf: no information either way
    assert lo == -np.inf and hi == np.inf
    lo, hi = capi.screen_record_bounds(0x3C00, 0, 0.25, np.inf, 4096, 1.0)  # a DB without a whole block: spread unknown
    assert lo == -np.inf and np.isfinite(hi)
    lo, hi = capi.screen_record_bounds(0x3C00, 0, 0.25, 1e-7, 4096, 1.0)
    assert lo < 1.25 < hi and hi - lo < 5e-3


def test_pruning_never_drops_a_row_of_the_exact_top2():
    """Pass B + pass C's pruning replayed in host arithmetic on a DB with near-duplicates: whatever records the
    second-largest-lower-bound rule drops, the rows of the exact 2-NN (canonical distance, ties included) stay."""
    base, _, _ = synth.load_sift_fixture()
    rng = np.random.default_rng(31)
    b = orclib.normalize(base[rng.choice(len(base), 24)])
    rows = [b[k] + 10.0 ** rng.uniform(-5, -3, (32, 1)) * rng.normal(size=(32, 128)) for k in range(24)]
    d = orclib.normalize(np.ascontiguousarray(np.maximum(np.concatenate(rows), 0), np.float32))
    d = d[rng.permutation(len(d))]
    q = orclib.normalize(np.ascontiguousarray(np.concatenate([b, b + rng.normal(0, 2e-4, b.shape)]), np.float32))
    dd, qq = orclib.row_norms(d), orclib.row_norms(q)
    neg = (np.float32(-0.5) * dd).astype(np.float32)
    dmax = float(np.sqrt(dd.max()))
    N, n_blocks = len(d), len(d) // 32
    spread = max(float(neg[k * 32:(k + 1) * 32].max() - neg[k * 32:(k + 1) * 32].min()) for k in range(n_blocks))
    p = _chain_f32(q, d)
    a = np.maximum((np.float32(-2) * p.astype(np.float64) + (qq[:, None] + dd[None, :]).astype(np.float64)).astype(np.float32), 0)
    dots = _dots_f32(q, d)
    wt = _screen_f32(q, d, dd)
    sample = np.arange(5, N, 8)
    dropped = kept = 0
    for i in range(len(q)):
        tau = float(np.float32(np.sort(wt[i, sample])[-2]) - np.float32(_margin(qq[i], dmax)))
        recs = []
        for k in range(n_blocks):
            hi_b = float(neg[k * 32:(k + 1) * 32].max())
            for h in (0, 1):
                r = k * 32 + _LANE_ROWS[h]
                rec = _record(dots[i, r], hi_b, tau)
                if rec is not None:
                    recs.append((r[rec[0]], capi.screen_record_bounds(rec[1], k * 32 + 4 * h, tau, spread, N, dmax)))
        los = sorted((lo for _, (lo, _) in recs), reverse=True)
        keep_from = (los[1] if len(los) > 1 else -np.inf) - _margin(qq[i], dmax)
        survivors = set()
        for rws, (lo, hi) in recs:
            if hi < keep_from:
                dropped += 1
            else:
                kept += 1
                survivors.update(rws.tolist())
        second = np.sort(a[i])[1]
        must = np.nonzero(a[i] <= second)[0]
        assert set(must.tolist()) <= survivors, i
    assert dropped > 0 and kept > 0          # and the rule does prune (on a DB of near-duplicates most records must stay)
