// oracle/linkage_oracle.cpp -- TEST INFRASTRUCTURE ONLY (part of liboracle.so).
//
// CPU restatement of moped3d's default clusterer CLUSTER_LINKAGE_CPU
// (moped3d/libmoped/src/cluster/CLUSTER_LINKAGE_CPU.hpp; line numbers below are that file's)
// as moped3d's config.hpp:45 constructs it: per model, a similarity matrix over the model's
// matches -- Gaussian kernels on image and camera-frame distances with sigmas = average
// nearest-neighbour distances (:97-123, 133-149), a depth-discontinuity kernel sampled along
// the image line between two matches (:176-285), the model/world distance consistency kernel
// (:151-173), the fill-distance weighted sum (:325-366) -- then agglomerative clustering with
// the reference's update rule and list handling (:416-540), quirks included:
//   * the merged-away cluster is erased from the index list only during the NEXT scan, when
//     the scan reaches it as first index; pairs (earlier index, it) are still candidates in
//     that scan with their stale similarities, and the element after it is skipped as first
//     index by the erase-then-increment (:446-463);
//   * `valid` is never cleared (:500), so emptied clusters simply fail `size() > MinPts`;
//   * merging appends the second cluster's members in reverse order (:493-496).
// Float = float, unsuffixed literals = double, exp/atan2/sqrt/fabs/modf on Float arguments are the
// float overloads.  Coordinates outside the depth / distance maps are clamped to them (the
// reference clamps for the depth map only, :222-229).
//
// PARITY UNPINNED: moped3d's step headers need OpenCV -> no reference build; restated from the
// source text, hand-worked cases in tests/test_linkage_cpu.py.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <list>
#include <utility>
#include <vector>

#include "oracle.h"

namespace {

struct DepthMaps {
  const float* img;   // [h][w][4]
  const float* fill;  // [h][w] or null
  int w, h;
  float depth(int x, int y) const { return img[((size_t)y * w + x) * 4 + 2]; }   // Image::getDepth
};

inline float eucl2(const float* a, const float* b, int n) {   // Pt<N>::sqEuclDist(pt): d = pt - this
  float r = 0.f;
  for (int x = 0; x < n; ++x) {
    const float d = b[x] - a[x];
    r += d * d;
  }
  return r;
}
inline float eucl(const float* a, const float* b, int n) { return std::sqrt(eucl2(a, b, n)); }

inline void saturate(int& x, int& y, const DepthMaps& D) {   // saturatePair (:222-229)
  x = (x < 0) ? 0 : x;
  x = (x >= D.w) ? D.w - 1 : x;
  y = (y < 0) ? 0 : y;
  y = (y >= D.h) ? D.h - 1 : y;
}

// bresenhamIterate (:176-220)
void bresenham(std::vector<std::pair<int, int> >& coords, int px, int py, int qx, int qy, int numberSamples) {
  int x0 = px, y0 = py, x1 = qx, y1 = qy, t;
  const bool steep = std::abs(y1 - y0) > std::abs(x1 - x0);
  if (steep) {
    t = x0; x0 = y0; y0 = t;
    t = x1; x1 = y1; y1 = t;
  }
  if (x0 > x1) {
    t = x0; x0 = x1; x1 = t;
    t = y0; y0 = y1; y1 = t;
  }
  const float deltaX = (float)x1 - x0, deltaY = std::fabs((float)y1 - y0);
  const int yStep = (y0 < y1) ? 1 : -1;
  int perStep = (int)(x1 - x0) / numberSamples;
  if (perStep < 1) perStep = 1;
  float error = 0.0, deltaError = ((float)deltaY) / deltaX;
  float intPart;
  int y = y0;
  for (int x = x0; x <= x1;) {
    if (steep) coords.push_back(std::make_pair(y, x));
    else coords.push_back(std::make_pair(x, y));
    x += perStep;
    error += deltaError * perStep * yStep;
    error = std::modf(error, &intPart);
    y += intPart;
  }
}

}  // namespace

extern "C" int orc_cluster_linkage(const float* uv, const float* model_xyz, const float* world_xyz, int n,
                                   const float* depth_img, int w, int h, const float* fill_img, float cutoff,
                                   int min_pts, int use3d_filter, int linkage_type, float sigma2d, float sigma3d,
                                   int32_t* members, int32_t* cluster_off, float* K_out) {
  cluster_off[0] = 0;
  if (n <= 0) return 0;
  const DepthMaps D = {depth_img, fill_img, w, h};
  const int N = n;
  const size_t NN = (size_t)N * N;
  // ---- sigmas: getAverageNNDistances (:97-123) ----
  float k2DSigma = sigma2d, k3DSigma = sigma3d;
  if (sigma2d == -1 || sigma3d == -1) {
    float nn2D = 0, nn3D = 0;
    for (int i = 0; i < N; ++i) {
      float nn2Di = DBL_MAX, nn3Di = DBL_MAX;
      for (int j = 0; j < N; ++j) {
        if (i == j) continue;
        const float dist2D = eucl(uv + 2 * i, uv + 2 * j, 2);
        const float dist3D = eucl(model_xyz + 3 * i, model_xyz + 3 * j, 3);   // Match::coord3D = the model point
        if (nn2Di > dist2D) nn2Di = dist2D;
        if (nn3Di > dist3D) nn3Di = dist3D;
      }
      nn2D += nn2Di;
      nn3D += nn3Di;
    }
    nn2D /= N;
    nn3D /= N;
    if (sigma2d == -1) k2DSigma = nn2D;
    if (sigma3d == -1) k3DSigma = nn3D;
  }
  std::vector<float> K2D(NN), K3D(NN), K(NN);
  // ---- getGaussK (:133-149) ----
  {
    const float two2 = 2 * k2DSigma * k2DSigma, two3 = 2 * k3DSigma * k3DSigma;
    for (int i = 0; i < N; ++i)
      for (int j = i; j < N; ++j) {
        const float v2 = std::exp(-1 * eucl2(uv + 2 * i, uv + 2 * j, 2) / two2);
        const float v3 = std::exp(-1 * eucl2(world_xyz + 3 * i, world_xyz + 3 * j, 3) / two3);
        K2D[(size_t)i * N + j] = K2D[(size_t)j * N + i] = v2;
        K3D[(size_t)i * N + j] = K3D[(size_t)j * N + i] = v3;
      }
  }
  // ---- getDiscontinuityMatrix (:231-285), added to K3D (getSum, :681) ----
  {
    std::vector<std::pair<int, int> > coords;
    const float discontinuityDiv = -2 * (M_PI / 128) * (M_PI / 128);
    for (int i = 0; i < N; ++i) {
      int lix = (int)uv[2 * i], liy = (int)uv[2 * i + 1];
      saturate(lix, liy, D);
      for (int j = i; j < N; ++j) {
        int ljx = (int)uv[2 * j], ljy = (int)uv[2 * j + 1];
        saturate(ljx, ljy, D);
        coords.clear();
        bresenham(coords, lix, liy, ljx, ljy, 20);
        const float depthStart = D.depth(lix, liy), depthEnd = D.depth(ljx, ljy);
        const int xDiff = lix - ljx, yDiff = liy - ljy;
        const float imagePlaneDist = std::sqrt((float)(xDiff * xDiff + yDiff * yDiff));
        const float directAngle = std::atan2(depthEnd - depthStart, imagePlaneDist);
        float maxAngleDiff = -1;
        for (int pix = 0; pix < (int)coords.size() - 1; ++pix) {
          int ax = coords[pix].first, ay = coords[pix].second, bx = coords[pix + 1].first, by = coords[pix + 1].second;
          // the walk can step outside the map by rounding: read the border pixel there
          int cax = ax, cay = ay, cbx = bx, cby = by;
          saturate(cax, cay, D);
          saturate(cbx, cby, D);
          const float depth1 = D.depth(cax, cay), depth2 = D.depth(cbx, cby);
          const float dx = ax - bx, dy = ay - by;
          const float pixDistance = std::sqrt(dx * dx + dy * dy);
          const float pixAngle = std::atan2(depth2 - depth1, pixDistance);
          const float angleDiff = std::fabs(directAngle - pixAngle);
          if (angleDiff > maxAngleDiff) maxAngleDiff = angleDiff;
        }
        const float val = std::exp(maxAngleDiff * maxAngleDiff / discontinuityDiv);
        K3D[(size_t)i * N + j] = K3D[(size_t)i * N + j] + val;
        if (j != i) K3D[(size_t)j * N + i] = K3D[(size_t)j * N + i] + val;
      }
    }
  }
  // normalizeSimilarityMatrix (:306-322)
  auto normalize = [&](std::vector<float>& M) {
    float maxValue = -1;
    for (size_t e = 0; e < NN; ++e)
      if (M[e] > maxValue) maxValue = M[e];
    for (size_t e = 0; e < NN; ++e) M[e] = M[e] / maxValue;
  };
  normalize(K3D);
  // ---- get3DFilterK (:151-173), getSum / getProduct (:683-692) ----
  if (use3d_filter) {
    const float sigma = 0.1;
    const float twoSigmaSq = 2 * sigma * sigma;
    for (int i = 0; i < N; ++i)
      for (int j = i; j < N; ++j) {
        float val = 1.0;
        if (j != i) {
          const float distanceModel = eucl(model_xyz + 3 * i, model_xyz + 3 * j, 3);
          const float distanceRealWorld = eucl(world_xyz + 3 * i, world_xyz + 3 * j, 3);
          const float distanceError = std::fabs(distanceModel - distanceRealWorld) / distanceModel;
          val = std::exp((-1 * distanceError * distanceError) / twoSigmaSq);
        }
        for (int rep = 0; rep < (j != i ? 2 : 1); ++rep) {
          float& e = rep ? K3D[(size_t)j * N + i] : K3D[(size_t)i * N + j];
          e = (use3d_filter == 1) ? e + val : e * val;
        }
      }
    normalize(K3D);
  }
  // ---- adaptiveWeightSum(matches, distanceMap, K2D, K3D, 0.5, 25) (:325-366) ----
  {
    const float alpha = 0.5, gamma = 25;
    const float gammaSq = gamma * gamma;
    std::vector<float> weights(N);
    for (int i = 0; i < N; ++i) {
      int x = (int)uv[2 * i], y = (int)uv[2 * i + 1];
      saturate(x, y, D);
      const float d = D.fill ? D.fill[(size_t)y * D.w + x] : 0.f;
      weights[i] = 1.0 / (1 + (d * d / gammaSq));
    }
    const float alphaBar = 1.0 - alpha;
    for (int i = 0; i < N; ++i)
      for (int j = i; j < N; ++j) {
        const float K2De = K2D[(size_t)j * N + i], K3De = K3D[(size_t)j * N + i];
        const float jointWeight = weights[i] * weights[j];
        const float w2D = (alpha + alphaBar * (1.0 - jointWeight)), w3D = alphaBar * jointWeight;
        const float val = w2D * K2De + w3D * K3De;
        K[(size_t)i * N + j] = K[(size_t)j * N + i] = val;
      }
  }
  if (K_out)
    for (size_t e = 0; e < NN; ++e) K_out[e] = K[e];
  // ---- hierarchicalCluster (:416-540) ----
  std::vector<std::list<int> > clusters(N);
  std::vector<float> distances(NN);
  std::list<int> validIndices;
  for (int i = 0; i < N; ++i) {
    clusters[i].push_back(i);
    validIndices.push_back(i);
    for (int j = i; j < N; ++j) distances[(size_t)j * N + i] = distances[(size_t)i * N + j] = K[(size_t)j * N + i];
  }
  int removeValue = -1;
  while (true) {
    float maxSimilarity = -1;
    std::pair<int, int> maxPair(0, 0);
    std::list<int>::iterator index1_it, index2_it;
    for (index1_it = validIndices.begin(); index1_it != validIndices.end(); index1_it++) {
      const int index1 = *index1_it;
      if (index1 == removeValue) {
        index1_it = validIndices.erase(index1_it);
        if (index1_it == validIndices.end()) break;   // (the reference would increment end(): the list is done)
        continue;
      }
      index2_it = index1_it;
      index2_it++;
      for (; index2_it != validIndices.end(); index2_it++) {
        const int index2 = *index2_it;
        if (distances[(size_t)index1 * N + index2] > maxSimilarity) {
          maxSimilarity = distances[(size_t)index1 * N + index2];
          maxPair = std::make_pair(index1, index2);
        }
      }
    }
    if (maxSimilarity < cutoff) break;
    const int SToUpdate = (int)clusters[maxPair.first].size(), SRemoveValue = (int)clusters[maxPair.second].size();
    while (clusters[maxPair.second].size() != 0) {
      clusters[maxPair.first].push_back(clusters[maxPair.second].back());
      clusters[maxPair.second].pop_back();
    }
    const int toUpdate = maxPair.first;
    removeValue = maxPair.second;
    for (int i = 0; i < N; ++i) {
      if (linkage_type == 1) {   // average linkage update rule (:515-523)
        distances[(size_t)toUpdate * N + i] =
            (1.0 / (SToUpdate + SRemoveValue)) *
            (SToUpdate * distances[(size_t)toUpdate * N + i] + SRemoveValue * distances[(size_t)removeValue * N + i]);
        distances[(size_t)i * N + toUpdate] = distances[(size_t)toUpdate * N + i];
      } else {                   // minimum (0) / maximum (2) linkage over the original similarities (:380-413)
        float link = linkage_type == 0 ? 1e20f : -1.f;
        for (int a : clusters[i])
          for (int b : clusters[toUpdate]) {
            const float v = K[(size_t)b * N + a];
            if (linkage_type == 0 ? v < link : v > link) link = v;
          }
        distances[(size_t)toUpdate * N + i] = distances[(size_t)i * N + toUpdate] = link;
      }
    }
  }
  int ncl = 0, wpos = 0;
  for (int i = 0; i < N; ++i) {
    if ((int)clusters[i].size() > min_pts) {   // strictly more than MinPts (:535)
      for (int m : clusters[i]) members[wpos++] = m;
      cluster_off[++ncl] = wpos;
    }
  }
  return ncl;
}
