"""Second witnesses for the rows of SURVEY.md 8(a) that no reference build can pin (the STEP headers need OpenCV):
restatements of the reference's TEXT written independently of oracle/oracle.cpp and by another route -- numpy arrays
and index arithmetic where the oracle (like the reference) walks std::list / std::map -- so that a misreading would have
to be made twice, in two different shapes, to go unnoticed.  tests/test_witness_cpu.py runs the oracle against these on
tens of thousands of adversarial cases.  Test infrastructure: nothing outside tests/ imports this.

Arithmetic is float32 in the reference's operation order (Float = float, include/moped.hpp:74-78); `+=` of a double
into a Float goes through double and back like the C++ does."""
import numpy as np

f32 = np.float32


# ---------------------------------------------------------------------------------------------------------------------
# MeanShift  (src/cluster/CLUSTER_MEAN_SHIFT_CPU.hpp:80-158)
# ---------------------------------------------------------------------------------------------------------------------
def meanshift(points, radius, merge, min_pts, max_iter):
    """-> (clusters: list of lists of point indices in the reference's emission and member order, iterations).

    Route: `canopiesRemaining` (:89, a std::list<int> that starts as 0..n-1 and only ever loses elements, :145) is a
    boolean mask walked in ascending id; a canopy's `merges` pointer (:66) is an index; boundPoints (:62) a Python list
    whose len() plays boundPoints.size() (:139) next to the separately kept boundPointsSize (:63)."""
    P = np.asarray(points, f32)
    n = P.shape[0]
    dim = P.shape[1] if P.ndim == 2 else 2
    sq_radius, sq_merge = f32(radius) * f32(radius), f32(merge) * f32(merge)
    center = P.copy().reshape(n, dim)
    agg = np.zeros((n, dim), f32)
    bound = [[i] for i in range(n)]
    size = np.ones(n, np.int64)                # boundPointsSize
    merges = np.arange(n)
    alive = np.ones(n, bool)
    done, it = False, 0
    while not done and it < max_iter:
        done = True
        ids = np.nonzero(alive)[0]
        C, S = center[ids], size[ids]
        # (1) :98-117 -- touchPtsAggregate = center * size, then += other.center * other.size for the others within
        # Radius in list order, then /= touchPtsN.  sqEuclDist (moped.hpp:121): d = other - this per coordinate, r += d*d.
        for a, c in enumerate(ids):
            d = C - C[a]
            dist = np.zeros(len(ids), f32)
            for x in range(dim):
                dist = dist + d[:, x] * d[:, x]
            near = dist < sq_radius
            near[a] = False
            terms = C[near] * S[near].astype(f32)[:, None]          # Pt * int -> every coordinate times (float)size
            acc = C[a] * f32(S[a])
            for t in terms:                                          # list order, one += per neighbour
                acc = acc + t
            agg[c] = acc / f32(S[a] + S[near].sum())                 # Pt /= int
        # (2) :119-129 -- for c in list: for o in list before c: if close, o's current target is redirected to c, then o
        G = agg[ids]
        for a, c in enumerate(ids):
            if a == 0:
                continue
            d = G[:a] - G[a]
            dist = np.zeros(a, f32)
            for x in range(dim):
                dist = dist + d[:, x] * d[:, x]
            for b in np.nonzero(dist < sq_merge)[0]:
                o = ids[b]
                merges[merges[o]] = c
                merges[o] = c
        # (3) :131-147 -- in list order, a canopy that points elsewhere is folded into its target and leaves the list;
        # the target may itself have left already (its point list spliced away: size() == 0, boundPointsSize kept)
        for c in ids:
            t = merges[c]
            if t == c:
                continue
            center[t] = center[t] * f32(len(bound[t])) + center[c] * f32(size[c])
            bound[t].extend(bound[c])
            bound[c] = []
            size[t] += size[c]
            center[t] = center[t] / f32(size[t])
            alive[c] = False
            done = False
        it += 1
    clusters = [list(bound[c]) for c in np.nonzero(alive)[0] if size[c] >= min_pts]     # :150-156
    return clusters, it


# ---------------------------------------------------------------------------------------------------------------------
# randSample + RANSAC  (src/pose/POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:76-98, 182-211)
# ---------------------------------------------------------------------------------------------------------------------
def rand_sample(uv, addr, n_samples, rand):
    """:76-98 -> (ok, sample positions in pick order).  `rand` = the stream ((Float)rand() keys, :83).  The deque of
    pair<Float, LmData*> is sorted (:84): key first, then POINTER = address order `addr` (ascending match index, :287-288).
    Points at an image coordinate already used are skipped but still count an entry of `used` only once (:91-92)."""
    k = len(uv)
    keys = np.array([f32(rand()) for _ in range(k)], f32)            # one rand() per point, in cluster order
    order = np.lexsort((np.asarray(addr), keys))                     # by key, ties by address
    used, picked = [], []
    for i in order:
        if len(used) >= n_samples:
            break
        c = (float(uv[i][0]), float(uv[i][1]))
        if c not in used:
            used.append(c)
            picked.append(int(i))
    return len(used) == n_samples, picked


def ransac(uv, xyz, addr, prm, rand, optimize_camera, test_all_points):
    """:188-211 -> (found, pose7).  optimize_camera(pose7, uv, xyz, itmax) -> (ret, pose7, info) and
    test_all_points(pose7, uv, xyz, thr) -> (count, mask) are the pinned arithmetic (oracle functions checked against the
    reference's levmar / project()); what this restates is the skeleton: how many rand() calls in which order, the
    sample's ties and duplicates, the early return on too few distinct points (:194), `continue` on LM failure (:199),
    the STRICT `>` MinNPtsObject (:204), first success wins and is refined on its inliers (:206-207)."""
    uv, xyz = np.asarray(uv, f32), np.asarray(xyz, f32)
    pose = np.zeros(7, f32)
    for _ in range(prm["max_ransac_tests"]):
        ok, pick = rand_sample(uv, addr, prm["n_pts_align"], rand)
        if not ok:
            return False, pose
        # initPose :182-186 (x, y, z, w drawn in that order here as in the oracle: the reference leaves it to the compiler)
        pose = np.array([f32((rand() & 255) / 256.0) for _ in range(4)] + [0.0, 0.0, 0.5], f32)
        ret, p, info = optimize_camera(pose, uv[pick], xyz[pick], prm["max_lm_tests"])
        if ret >= 0:
            pose = p
        lm_iterations = int(info[1]) if ret >= 0 else ret              # Float -> int (:198)
        if lm_iterations == -1:
            continue
        cnt, mask = test_all_points(pose, uv, xyz, prm["error_threshold"])
        if cnt > prm["min_n_pts_object"]:
            ret, p, _ = optimize_camera(pose, uv[mask], xyz[mask], prm["max_lm_tests"])
            if ret >= 0:
                pose = p
            return True, pose
    return False, pose


# ---------------------------------------------------------------------------------------------------------------------
# FILTER_PROJECTION  (src/filter/FILTER_PROJECTION_CPU.hpp:80-162)
# ---------------------------------------------------------------------------------------------------------------------
def filter_projection(uv, model_off, obj_model, proj_err2, min_points, feature_distance, min_score):
    """-> (score [n_obj], keep [n_obj], order of kept objects, their clusters as indices into matches[model]).

    proj_err2(o) = the squared reprojection errors of object o over ITS model's matches (project() of moped.hpp:330-354:
    pinned arithmetic, handed in).  Route: the reference keys a std::map by (coord2D, image) (:89); here every distinct
    coordinate gets an integer id once (np.unique over the bit patterns) and the map is two arrays.  One image."""
    uv = np.asarray(uv, f32).reshape(-1, 2)
    n_models, n_obj = len(model_off) - 1, len(obj_model)
    _, point_id = np.unique(uv.view(np.uint32).astype(np.uint64) @ np.array([1 << 32, 1], np.uint64), return_inverse=True)
    best_score = np.zeros(point_id.max() + 1 if len(uv) else 0, f32)       # pair<Float, Object*> default: (0, NULL)
    best_obj = np.full(len(best_score), -1)
    score = np.zeros(n_obj, f32)
    for m in range(n_models):                                              # :94-96: models outside, the list inside
        lo, hi = model_off[m], model_off[m + 1]
        for o in range(n_obj):
            if obj_model[o] != m:
                continue
            e = np.asarray(proj_err2(o), f32)
            inside = np.nonzero(e < f32(feature_distance))[0]              # :104-106, match order
            s = f32(0)
            for i in inside:                                               # Float += double (:107)
                s = f32(np.float64(s) + 1.0 / (np.float64(e[i]) + 1.0))
            score[o] = s
            for i in inside:                                               # :114-125: strictly better takes the point
                pid = point_id[lo + i]
                if best_score[pid] < s:
                    best_score[pid] = s
                    best_obj[pid] = o
    owned = [[] for _ in range(n_obj)]                                     # :129-138
    for m in range(n_models):
        for i in range(model_off[m], model_off[m + 1]):
            o = best_obj[point_id[i]]
            if o >= 0 and obj_model[o] == m:
                owned[o].append(i - model_off[m])
    keep = np.zeros(n_obj, bool)
    order, clusters = [], []
    for m in range(n_models):                                              # :143-158
        for o in range(n_obj):
            if obj_model[o] != m:
                continue
            if len(owned[o]) < min_points or score[o] < f32(min_score):
                continue
            keep[o] = True
            order.append(o)
            clusters.append(owned[o])
    return score, keep, order, clusters
