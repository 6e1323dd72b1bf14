// TEST INFRASTRUCTURE ONLY -- CPU restatement of moped3d's DEPTHFILL step, the checker of mh_depth_fill.
// Follows moped3d/libmoped/src/depthfill/DEPTH_FILL_EXACT_CPU.hpp (config.hpp:39:
// `new DEPTH_FILL_EXACT_CPU(8, false)`); Float = float (include/moped.hpp:73-77).  Parity unpinned: the step's header
// needs OpenCV through util.hpp and the reference holds no fixture for it; written from the source text,
// quirks kept:
//   - the nearest-neighbour upsampling advances its source row one output row late (:80-82: `ly` is bumped at the
//     END of the row whose index is a multiple of the factor), the bilinear one does not (:106-110);
//   - with a scale factor of 1 the filled map is assigned to the by-value parameter only (:336-338): the frame's depth
//     map stays as it was, only the distance map is the fill's;
//   - the fill is a FIFO wavefront, not an exact distance transform: a source is propagated from a pixel whether or
//     not the pixel still belongs to it, improvements are strict (`<`), so ties go to the first arrival (:215-236).
// Where the reference would read outside the downscaled map (width not a multiple of the factor) the index is clamped.
#include <cmath>
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

#include "oracle.h"

namespace {

// fillIn (:176-243) on a [dh][dw] map of depths; dist = bestDistance
void fill_in(std::vector<float>& z, int dw, int dh, float dilate, std::vector<float>& dist) {
  const int n = dw * dh;
  std::vector<char> valid(n);
  for (int i = 0; i < n; ++i) valid[i] = z[i] >= 0 ? 1 : 0;   // :190
  auto data_valid = [&](int x, int y) { return y < 0 || y >= dh || x < 0 || x >= dw || valid[y * dw + x]; };   // :55-61
  auto all8 = [&](int x, int y) {   // :33-45
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy) {
        if (dx == 0 && dy == 0) continue;
        if (!data_valid(x + dx, y + dy)) return false;
      }
    return true;
  };
  std::queue<std::pair<std::pair<int, int>, std::pair<int, int>>> grow;
  dist.assign(n, 0.f);
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      if (data_valid(x, y) && !all8(x, y)) grow.push({{x, y}, {x, y}});   // :202-204
      dist[y * dw + x] = valid[y * dw + x] ? 0.f : (float)1e30;           // :205
    }
  while (!grow.empty()) {
    const auto elt = grow.front();
    grow.pop();
    const int x0 = elt.first.first, y0 = elt.first.second, x1 = elt.second.first, y1 = elt.second.second;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int xp = x1 + dx, yp = y1 + dy;
        if (data_valid(xp, yp)) continue;
        const float cur = std::sqrt((float)((xp - x0) * (xp - x0) + (yp - y0) * (yp - y0))) * dilate;   // :227
        if (cur < dist[yp * dw + xp]) {
          dist[yp * dw + xp] = cur;
          z[yp * dw + xp] = z[y0 * dw + x0];
          grow.push({{x0, y0}, {xp, yp}});
        }
      }
  }
}

}  // namespace

extern "C" int orc_depth_fill(float* depth, int w, int h, int scale, int bilinear, const float K[4], float* dist_out) {
  // fillInScaled (:268-349)
  std::vector<char> valid((size_t)w * h);
  int valid_count = 0;
  for (int i = 0; i < w * h; ++i) {
    valid[i] = depth[4 * (size_t)i + 2] >= 0;
    valid_count += valid[i];
  }
  if (scale == -1) {   // :283-296
    const float invalid_ratio = ((float)w * h - valid_count) / (w * h);
    scale = invalid_ratio < 0.1 ? 1 : invalid_ratio < 0.2 ? 2 : invalid_ratio < 0.4 ? 4 : invalid_ratio < 0.6 ? 8 : 16;
  }
  if (scale < 1 || w / scale < 1 || h / scale < 1) return -1;
  for (int i = 0; i < w * h; ++i) dist_out[i] = 0.f;   // :306-308
  const int dw = w / scale, dh = h / scale;
  std::vector<float> z((size_t)dw * dh), fd;
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) z[y * dw + x] = depth[4 * ((size_t)(y * scale) * w + x * scale) + 2];   // :317-323
  fill_in(z, dw, dh, (float)scale, fd);
  if (scale == 1) {   // :336-338: only the distance map reaches the frame
    for (int i = 0; i < w * h; ++i) dist_out[i] = fd[i];
    return scale;
  }
  auto at = [&](const std::vector<float>& m, int x, int y) {   // (clamped where the reference reads out of bounds)
    return m[(size_t)std::min(y, dh - 1) * dw + std::min(x, dw - 1)];
  };
  if (!bilinear) {
    // NNInterp (:69-85), once for the depths and once for the distances
    int ly = 0;
    for (int uy = 0; uy < h; ++uy) {
      int lx = 0;
      for (int ux = 0; ux < w; ++ux) {
        if (ux != 0 && ux % scale == 0) ++lx;
        if (valid[(size_t)uy * w + ux]) continue;
        depth[4 * ((size_t)uy * w + ux) + 2] = at(z, lx, ly);
        dist_out[(size_t)uy * w + ux] = at(fd, lx, ly);
      }
      if (uy != 0 && uy % scale == 0) ++ly;
    }
  } else {
    // bilinearInterp (:93-168)
    const double delta = 1.0 / scale;
    double up = 0, left = 0;
    int ly = -1;
    for (int uy = 0; uy < h; ++uy) {
      up -= delta;
      if (uy % scale == 0) {
        ++ly;
        up = 1;
      }
      int lx = -1;
      for (int ux = 0; ux < w; ++ux) {
        left -= delta;
        if (ux % scale == 0) {
          ++lx;
          left = 1;
        }
        if (valid[(size_t)uy * w + ux]) continue;
        int x0 = lx, y0 = ly, x1 = lx + 1, y1 = ly + 1;
        float w00 = left * up, w01 = left * (1 - up), w10 = (1 - left) * up, w11 = (1 - left) * (1 - up);
        if (x1 == dw) {
          w00 += w01; w01 = 0;
          w01 += w11; w11 = 0;
          x1 = x0;
        }
        if (y1 == dh) {
          w00 += w10; w10 = 0;
          w10 += w11; w11 = 0;
          y1 = y0;
        }
        depth[4 * ((size_t)uy * w + ux) + 2] = w00 * at(z, x0, y0) + w01 * at(z, x0, y1) + w10 * at(z, x1, y0) + w11 * at(z, x1, y1);
        dist_out[(size_t)uy * w + ux] = w00 * at(fd, x0, y0) + w01 * at(fd, x0, y1) + w10 * at(fd, x1, y0) + w11 * at(fd, x1, y1);
      }
    }
  }
  // normalizeDepthmap (:249-273)
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      if (valid[(size_t)v * w + u]) continue;
      float* b = depth + 4 * ((size_t)v * w + u);
      float x = (u - K[2]) / K[0], y = (v - K[3]) / K[1];
      b[0] = x * b[2];
      b[1] = y * b[2];
      x = b[0];
      y = b[1];
      const float zz = b[2];
      b[3] = std::sqrt(x * x + y * y + zz * zz);
    }
  return scale;
}
