"""Host-side pieces of moped3d's pipeline that produce inputs for the device (product code; the
C++ plugins hold the same logic for libmoped hosts)."""
import numpy as np

_f = np.float32


def adaptive_control_points(bbox_min, bbox_max, K, n_features, min_ratio=(0.6, 0.75), max_ratio=(0.65, 0.8),
                            dimension_peak=150.0, dimension_fade=50.0):
    """What MATCH_ADAPTIVE_BRUTE_HIP::Update computes (moped_amd/host/MATCH_ADAPTIVE_BRUTE_HIP.hpp), in numpy, for
    Python hosts: MATCH_ADAPTIVE_FLANN_CPU::Update's per-model control points (:100-177 with
    solveProjectionDepth :318-357, getAverageProjectedLength :262-312, getProjectedArea :238-256):
    -> (maxRatioDepth, minRatioDepth, ratioLow, ratioHigh).  Defaults = moped3d config.hpp:43."""
    K = [float(_f(k)) for k in K]
    rng = [float(_f(bbox_max[i]) - _f(bbox_min[i])) for i in range(3)]

    def projected_area(pts):
        us = [_f(_f(_f(K[0]) * p[0] + _f(K[2]) * p[2]) / p[2]) for p in pts]
        vs = [_f(_f(_f(K[1]) * p[1] + _f(K[3]) * p[2]) / p[2]) for p in pts]
        return _f(_f(max(us) - min(us)) * _f(max(vs) - min(vs)))

    def avg_len(depth):
        xr, yr, zr = (_f(r) for r in rng)
        mnx, mxx, mny, mxy, mnz, mxz = _f(-xr / 2), _f(xr / 2), _f(-yr / 2), _f(yr / 2), _f(-zr / 2), _f(zr / 2)
        zc, yc, xc = _f((mxx - mnx) * (mxy - mny)), _f((mxx - mnx) * (mxz - mnz)), _f((mxy - mny) * (mxz - mnz))
        d = _f(depth)
        if zc >= xc and zc >= yc:
            s = [(mnx, mny, d), (mnx, mxy, d), (mxx, mxy, d), (mxx, mny, d)]
        elif yc >= xc and yc >= zc:
            s = [(mnx, mnz, d), (mnx, mxz, d), (mxx, mxz, d), (mxx, mnz, d)]
        else:
            s = [(mny, mnz, d), (mny, mxz, d), (mxy, mxz, d), (mxy, mnz, d)]
        return np.sqrt(projected_area(s), dtype=_f)

    def solve(target, iters=100, tol=0.01):
        left, right, it = _f(0), _f(2), 0
        target = _f(target)
        while it < iters:
            it += 1
            if avg_len(right) > target:
                right = _f(right * 2)
            else:
                break
        max_err = _f(target * _f(tol))
        while it < iters:                      # the reference keeps counting with the same `iter`
            it += 1
            mid = _f(_f(left + right) / 2)
            length = avg_len(mid)
            if abs(_f(length - target)) < max_err:
                return mid
            if length > target:
                left = mid
            else:
                right = mid
        return _f(_f(left + right) / 2)

    d_peak, d_fade = solve(dimension_peak), solve(dimension_fade)
    adj = _f(1.0 / (1.0 + np.exp(-1.0 * float(_f((_f(1750) - _f(n_features)) / _f(250))))))   # canonicalSigmoid
    lo = _f(_f(min_ratio[0]) + adj * _f(_f(min_ratio[1]) - _f(min_ratio[0])))
    hi = _f(_f(max_ratio[0]) + adj * _f(_f(max_ratio[1]) - _f(max_ratio[0])))
    return np.array([d_peak, d_fade, lo, hi], _f)



def ratio_table(db_xyz, model_of, n_models, K, **kw):
    """mh_depth_rules.ratio_table for a model database: one row of control points per model, from its
    bounding box and feature count."""
    rows = []
    for m in range(n_models):
        sel = model_of == m
        rows.append(adaptive_control_points(db_xyz[sel].min(0), db_xyz[sel].max(0), K, int(sel.sum()), **kw))
    return np.stack(rows)
