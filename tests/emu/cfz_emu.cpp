// Test-only CPU build of the kernel source: conflict_rez_amd/csrc/cfz_solver.inl compiled with
// g++, the 64 lanes of the wavefront executed as a loop.  Lets the `not gpu` suite check the
// kernel's logic against the oracle (and run it under sanitizers).  Never shipped, never loaded
// by the product package.
#include <stdlib.h>
#include <string.h>

#include "../../conflict_rez_amd/csrc/cfz_solver.inl"

extern "C" {

int cfz_emu_sizeof_kspec(void) { return (int)sizeof(cfz::KSpec); }

// zu[7][N] in/out; out_i[2] = iters,status ; out_d[3] = cost,err,min_sep ; duals may be NULL
int cfz_emu_carry_doubles(int N, int nb) { return cfz::carry_layout(N, nb).stride; }

// wst: carry record of the instance (cfz_emu_carry_doubles doubles, may be NULL), carry_in: start from it
int cfz_emu_solve(const cfz::KSpec *sp, const double *x0, const double *ref, const double *nbr, double *zu,
                  int *out_i, double *out_d, double *l, double *mm, double *lam_ij, double *lam_ji, double *s,
                  double *wst, int carry_in) {
  const int nb = sp->n_obs + sp->n_nbr;
  cfz::Lay L = cfz::make_layout(sp->N, nb, sp->n_nbr);
  double *m = (double *)calloc((size_t)L.total, sizeof(double));
  if (!m) return -1;
  cfz::DualOut duo = {l, mm, lam_ij, lam_ji, s, nullptr};
  cfz::solve_instance(*sp, cfz::derive(*sp), x0, ref, nbr, zu, m, L, out_i, out_d, duo, wst, carry_in);
  free(m);
  return L.total;
}
}
