// tests/test_sift_rows_cpu.py: desc_row_interval (moped_amd/csrc/sift_rows.h) must contain every column of a window row
// that passes KeySample's tests as describe_kernel evaluates them (the same fp32 expressions, no contraction).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>

#include "../../moped_amd/csrc/sift_rows.h"

static uint64_t rng = 88172645463325252ull;
static double uni() {
  rng ^= rng << 13;
  rng ^= rng >> 7;
  rng ^= rng << 17;
  return (double)(rng >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv) {
  const int n_keys = argc > 1 ? atoi(argv[1]) : 20000;
  long rows_checked = 0, kept = 0, inside = 0, passing = 0;
  for (int t = 0; t < n_keys; ++t) {
    const int rows = 20 + (int)(uni() * 900), cols = 20 + (int)(uni() * 1200);
    const float fSize = 1.6f * powf(2.0f, (float)(uni() * 4.5) / 3.0f);   // index + x0 in [0, 4.5]
    const float frow = (float)(uni() * rows), fcol = (float)(uni() * cols);
    float ang;
    const int kind = t % 8;
    const float kPi = 3.141592654f;
    if (kind == 0) ang = (float)((int)(uni() * 8) - 4) * (kPi / 2);                            // the axes: a slope ~ 1e-8
    else if (kind == 1) ang = (float)((int)(uni() * 8) - 4) * (kPi / 2) + (float)((uni() - 0.5) * 2e-3);   // slopes around 1e-4
    else ang = (float)((uni() * 2 - 1) * kPi);
    const int rowstart = (int)(frow + 0.5f), colstart = (int)(fcol + 0.5f);
    const float sinang = sinf(ang), cosang = cosf(ang);
    const float fdrow = frow - (float)rowstart, fdcol = fcol - (float)colstart;
    const float frealsize = 3.0f * fSize, firealsize = 1.0f / (3.0f * fSize);
    const int win = (int)(frealsize * 1.4142136f * 5.0f * 0.5f + 0.5f);
    const float fsr = sinang * firealsize, fcr = cosang * firealsize;
    const float fdrr = -fdrow * firealsize, fdcr = -fdcol * firealsize;
    for (int row = -win; row <= win; ++row) {
      int lo, hi;
      mh::desc_row_interval(fsr, fcr, fdrr, fdcr, row, win, rowstart, colstart, rows, cols, lo, hi);
      ++rows_checked;
      if (lo < -win || hi > win || hi < lo - 1) {
        printf("FAIL interval out of range: key %d row %d [%d, %d] win %d\n", t, row, lo, hi, win);
        return 1;
      }
      kept += hi - lo + 1;
      for (int col = -win; col <= win; ++col) {
        const float fr = (float)row, fc = (float)col;
        const float rpos = (fsr * fc + fcr * fr) + fdrr;
        const float cpos = (fcr * fc - fsr * fr) + fdcr;
        const float rx = rpos + (2.0f - 0.5f), cx = cpos + (2.0f - 0.5f);
        const int r = rowstart + row, c = colstart + col;
        const bool ok = rx > -0.9999f && rx < 3.9999f && cx > -0.9999f && cx < 3.9999f && r >= 0 && r < rows && c >= 0 && c < cols;
        if (!ok) continue;
        ++passing;
        if (col < lo || col > hi) {
          printf("FAIL key %d (fSize %g ang %g) row %d col %d passes the tests outside [%d, %d]\n", t, fSize, ang, row, col, lo, hi);
          return 1;
        }
        ++inside;
      }
    }
  }
  printf("OK keys %d rows %ld passing %ld kept %ld overhead %.4f\n", n_keys, rows_checked, passing, kept, (double)kept / (double)(passing ? passing : 1));
  return 0;
}
