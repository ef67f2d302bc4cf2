// Test-only CPU build of the planning solver source (conflict_rez_amd/csrc/cfz_plan.inl), see cfz_emu.cpp.
#include <stdlib.h>

#include "../../conflict_rez_amd/csrc/cfz_plan.inl"

extern "C" {
int cfzp_emu_sizeof_spec(void) { return (int)sizeof(cfzp::PSpec); }
int cfzp_emu_n(const cfzp::PSpec *sp) { return cfzp::dims(*sp).n; }
// returns the half-bandwidth the stage-interleaved ordering needs for this problem (must be <= kKB)
int cfzp_emu_bandwidth(const cfzp::PSpec *sp) {
  const cfzp::PDims d = cfzp::dims(*sp);
  int *px = (int *)malloc(sizeof(int) * (d.n + d.m)), *pc = px + d.n;
  cfzp::build_order(*sp, px, pc);
  int bw = 0;
  auto upd = [&](int a, int b) { int v = a > b ? a - b : b - a; if (v > bw) bw = v; };
  for (int i = 0; i < 7; ++i) upd(pc[i], px[i]);
  for (int k = 0; k < sp->T; ++k)
    for (int i = 0; i < 5; ++i) { for (int j = 0; j < 7; ++j) upd(pc[7 + 5 * k + i], px[7 * k + j]); upd(pc[7 + 5 * k + i], px[7 * (k + 1) + i]); }
  for (int k = 0; k < sp->T; ++k) for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) upd(px[7 * k + i], px[7 * k + j]);
  for (int i = 0; i < sp->n_chk; ++i)
    for (int q = 0; q < 8; ++q) { for (int j = 0; j < 3; ++j) upd(pc[d.r0 + 8 * i + q], px[7 * sp->N * (i + 1) + j]); upd(pc[d.r0 + 8 * i + q], px[d.s0 + 8 * i + q]); }
  free(px);
  return bw;
}
int cfzp_emu_state_ws(const cfzp::PSpec *sp, const double *tube, double *X, int *out_i, double *out_d) {
  double *slab = (double *)calloc(cfzp::work_doubles(*sp), sizeof(double));
  if (!slab) return -1;
  // a plain buffer stands in for the LDS window so that the windowed elimination is what the CPU tests run
  double *win = (double *)calloc((size_t)cfzp::kWinCols * cfzp::kLd, sizeof(double));
  cfzp::solve_state_ws<true>(*sp, tube, X, slab, out_i, out_d, win);
  free(win);
  free(slab);
  return 0;
}
}
