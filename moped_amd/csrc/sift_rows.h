// describe_kernel's window rows (sift.hip): the columns of a window row outside of which KeySample's tests
// (libsiftfast.cpp:1560-1567: rx, cx in (-0.9999, 3.9999), the pixel inside the image) cannot pass.
// Host + device so that tests/test_sift_rows_cpu.py can check the interval against the tests themselves.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define MH_HD __host__ __device__
#else
#define MH_HD
#endif

namespace mh {

// rx(col) = fsr col + (fcr row + fdrr + 1.5), cx(col) = fcr col + (-fsr row + fdcr + 1.5) up to fp32 rounding (a few
// 1e-6 for |values| < 16).  Each condition is solved for col with the bounds widened to (-1.01, 4.01) and one more
// column on either side; a slope below 1e-4 moves the value by less than 0.005 over the widest window (48 columns
// either way), so the row is kept whole or dropped on the constant term with a margin of 0.01.  [lo, hi] in window
// coordinates (-win .. win), hi = lo - 1 for an empty row.
MH_HD inline void desc_row_interval(float fsr, float fcr, float fdrr, float fdcr, int row, int win, int rowstart,
                                    int colstart, int rows, int cols, int& lo, int& hi) {
  lo = -win;
  hi = win;
  const int r = rowstart + row;
  if (r < 0 || r >= rows) {
    hi = lo - 1;
    return;
  }
  if (-colstart > win || cols - 1 - colstart < -win) {   // the window lies beside the image
    hi = lo - 1;
    return;
  }
  if (lo < -colstart) lo = -colstart;
  if (hi > cols - 1 - colstart) hi = cols - 1 - colstart;
  const float fr = (float)row;
  const float a[2] = {fsr, fcr};
  const float b[2] = {fcr * fr + fdrr + 1.5f, -fsr * fr + fdcr + 1.5f};
  for (int i = 0; i < 2; ++i) {
    const float L = -1.01f, H = 4.01f;
    if (fabsf(a[i]) < 1e-4f) {
      if (!(b[i] > L - 0.01f && b[i] < H + 0.01f)) hi = lo - 1;
      continue;
    }
    float x0 = (L - b[i]) / a[i], x1 = (H - b[i]) / a[i];
    if (x0 > x1) {
      const float t = x0;
      x0 = x1;
      x1 = t;
    }
    // (|x| <= 5.1 / 1e-4: inside int)
    const int l = (int)floorf(x0) - 1, h = (int)ceilf(x1) + 1;
    if (lo < l) lo = l;
    if (hi > h) hi = h;
  }
  if (hi < lo) {
    lo = -win;
    hi = lo - 1;
  }
}

}  // namespace mh
