// cfz_common.h -- shared by the two translation units of libconfrez_hip.so (cfz_engine.hip: the MPC step and its closed
// loop; cfz_planning.hip: state_ws and the collocation plans): error reporting of the C ABI and small host helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

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


// Device memory of one caller, kept between calls: a bump allocator over blocks that are released by arena_destroy, merged
// into one larger block at the next reset, or given back when the calls have become much smaller than the block (a 256-plan
// four-vehicle joint launch takes 25 GB; the next reset after a call that used less than a quarter of a block above 256 MB
// frees it).  Replaces hipMalloc/hipFree per call in the planning entry points (each of those is a device-wide synchronisation).
struct CfzArena {
  struct Block { char *base; size_t cap, off; };
  std::vector<Block> blocks;
  size_t used_last = 0;
};

inline int arena_reset(CfzArena &a) {  // start of a call: everything handed out before is free again
  size_t used = 0, cap = 0;
  for (auto &b : a.blocks) { used += b.off; cap += b.cap; b.off = 0; }
  a.used_last = used;
  if (a.blocks.size() > 1) {  // the last call spilled into extra blocks: one block of the total size from now on
    for (auto &b : a.blocks) (void)hipFree(b.base);
    a.blocks.clear();
    char *p = nullptr;
    HIP_OK(hipMalloc(&p, cap));
    a.blocks.push_back({p, cap, 0});
  } else if (a.blocks.size() == 1 && cap > ((size_t)1 << 28) && used < cap / 4) {
    (void)hipFree(a.blocks[0].base);  // the previous call needed a fraction of what an earlier one left behind
    a.blocks.clear();
  }
  return 0;
}

inline int arena_alloc(CfzArena &a, void **out, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  for (auto &b : a.blocks)
    if (b.cap - b.off >= bytes) { *out = b.base + b.off; b.off += bytes; return 0; }
  const size_t cap = bytes > ((size_t)1 << 20) ? bytes : ((size_t)1 << 20);
  char *p = nullptr;
  HIP_OK(hipMalloc(&p, cap));
  a.blocks.push_back({p, cap, bytes});
  *out = p;
  return 0;
}

inline void arena_destroy(CfzArena &a) {
  for (auto &b : a.blocks) (void)hipFree(b.base);
  a.blocks.clear();
}
#define ARENA_ALLOC(arena, ptr, bytes) do { void *p_ = nullptr; if (arena_alloc((arena), &p_, (bytes))) return -1; (ptr) = (decltype(ptr))p_; } while (0)

}  // namespace
