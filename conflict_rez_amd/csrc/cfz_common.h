// cfz_common.h -- shared by the two translation units of libconfrez_hip.so (cfz_engine.hip: the MPC step and its closed
// loop; cfz_planning.hip: state_ws and the collocation plans): error reporting of the C ABI and small host helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>

extern thread_local std::string cfz_g_err;  // defined in cfz_engine.hip; what cfz_last_error() returns

namespace {

inline int fail(const char *what, hipError_t e = hipSuccess) {
  cfz_g_err = what;
  if (e != hipSuccess) { cfz_g_err += ": "; cfz_g_err += hipGetErrorString(e); }
  return -1;
}
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(#call, e_); } while (0)

// vertices of {A p <= b} for a bounded quadrilateral
inline bool quad_vertices(const double A[4][2], const double b[4], double V[4][2]) {
  int n = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = i + 1; j < 4; ++j) {
      const double det = A[i][0] * A[j][1] - A[i][1] * A[j][0];
      if (std::fabs(det) < 1e-9) continue;
      const double px = (b[i] * A[j][1] - A[i][1] * b[j]) / det, py = (A[i][0] * b[j] - b[i] * A[j][0]) / det;
      bool in = true;
      for (int q = 0; q < 4; ++q) in = in && (A[q][0] * px + A[q][1] * py <= b[q] + 1e-9);
      if (in) { if (n == 4) return false; V[n][0] = px; V[n][1] = py; ++n; }
    }
  if (n != 4) return false;
  // counter-clockwise around the polygon, starting from the first vertex found: v-1 and v+1 (mod 4) are then the
  // neighbours of v, which cfz::select_rows relies on
  const double cx = 0.25 * (V[0][0] + V[1][0] + V[2][0] + V[3][0]), cy = 0.25 * (V[0][1] + V[1][1] + V[2][1] + V[3][1]);
  const double two_pi = 6.283185307179586, a0 = std::atan2(V[0][1] - cy, V[0][0] - cx);
  double key[4], W[4][2];
  int ord[4] = {0, 1, 2, 3};
  for (int i = 0; i < 4; ++i) { double a = std::atan2(V[i][1] - cy, V[i][0] - cx) - a0; while (a < 0.0) a += two_pi; while (a >= two_pi) a -= two_pi; key[i] = a; }
  for (int i = 1; i < 4; ++i) for (int q = i; q > 0 && key[ord[q]] < key[ord[q - 1]]; --q) { const int t = ord[q]; ord[q] = ord[q - 1]; ord[q - 1] = t; }
  for (int i = 0; i < 4; ++i) { W[i][0] = V[ord[i]][0]; W[i][1] = V[ord[i]][1]; }
  memcpy(V, W, sizeof W);
  return true;
}

}  // namespace
