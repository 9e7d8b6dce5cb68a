// host_sincosf.h — float sine and cosine as the host's libm returns them.
//
// applyState (lamp_utils/include/lamp_utils/gicp.hpp:619-634) builds its rotation from
// std::sin / std::cos of float half angles.  The BFGS back end is compared with a host run of the
// same solver evaluation by evaluation, and a sine that differs in its last bit moves every
// transformed point of that evaluation; the device's own sinf / cosf differ from a host libm in the
// last bit for a few arguments in a thousand.  glibc (2.28 and later, all architectures) computes
// both for |x| < pi/4 as one double-precision polynomial in x rounded to float — the minimax
// polynomials of Arm's optimized routines — which is restated here; tests/test_host_sincosf.py
// checks it bit for bit against the host's sinf / cosf.  Arguments beyond pi/4 (half angles of
// more than 45 degrees never occur between two GICP iterations) take the double-precision
// function rounded once.  Plain multiplications and additions: whether the host's build fuses them
// changes a float result once in about 1e9 arguments.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define LGICP_HOST_DEVICE __host__ __device__
#else
#define LGICP_HOST_DEVICE
#endif

namespace lgicp {

LGICP_HOST_DEVICE inline float host_sinf(float y) {
  const float ay = y < 0.f ? -y : y;
  if (!(ay < 0x1.921fb6p-1f)) return (float)sin((double)y);
  if (ay < 0x1p-12f) return y;
  const double x = y, x2 = x * x;
  const double x3 = x * x2;
  const double s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
  const double x7 = x3 * x2;
  const double s = x + x3 * -0x1.555545995a603p-3;
  return (float)(s + x7 * s1);
}

LGICP_HOST_DEVICE inline float host_cosf(float y) {
  const float ay = y < 0.f ? -y : y;
  if (!(ay < 0x1.921fb6p-1f)) return (float)cos((double)y);
  if (ay < 0x1p-12f) return 1.0f;
  const double x = y, x2 = x * x;
  const double x4 = x2 * x2;
  const double c2 = -0x1.6c087e89a359dp-10 + x2 * 0x1.99343027bf8c3p-16;
  const double c1 = 1.0 + x2 * -0x1.ffffffd0c621cp-2;
  const double x6 = x4 * x2;
  const double c = c1 + x4 * 0x1.55553e1068f19p-5;
  return (float)(c + x6 * c2);
}

}  // namespace lgicp
