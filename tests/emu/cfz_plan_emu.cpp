// Test-only CPU build of the planning solver source (conflict_rez_amd/csrc/cfz_plan.inl), see cfz_emu.cpp.
#include <stdlib.h>

#include "../../conflict_rez_amd/csrc/cfz_plan.inl"

extern "C" {
int cfzp_emu_sizeof_spec(void) { return (int)sizeof(cfzp::PSpec); }
int cfzp_emu_n(const cfzp::PSpec *sp) { return cfzp::dims(*sp).n; }
int cfzp_emu_state_ws(const cfzp::PSpec *sp, const double *tube, double *X, int *out_i, double *out_d) {
  double *slab = (double *)calloc(cfzp::work_doubles(*sp), sizeof(double));
  if (!slab) return -1;
  cfzp::solve_state_ws<false>(*sp, tube, X, slab, out_i, out_d);
  free(slab);
  return 0;
}
}
