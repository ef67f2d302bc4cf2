// Test-only CPU build of the collocation planning solver source (conflict_rez_amd/csrc/cfz_colloc.inl), see cfz_emu.cpp.
#include <stdlib.h>
#include <string.h>

#include "../../conflict_rez_amd/csrc/cfz_colloc.inl"

static int order_half_bandwidth(const cfzc::CSpec *sp) {
  const cfzc::CDims d = cfzc::cdims(*sp);
  int *px = (int *)malloc(sizeof(int) * (d.n + d.m)), *pc = px + d.n;
  const int cnt = cfzc::build_order(*sp, px, pc);
  const int kb = cnt == d.nk ? cfzc::half_bandwidth(*sp, px, pc) : -1;
  free(px);
  return kb;
}

extern "C" {
int cfzc_emu_sizeof_spec(void) { return (int)sizeof(cfzc::CSpec); }
void cfzc_emu_dims(const cfzc::CSpec *sp, int *out) {
  const cfzc::CDims d = cfzc::cdims(*sp);
  const int v[] = {d.np, d.nr, d.n, d.m, d.nk, d.iDt, d.sO, d.sT, d.sP, d.rO, d.rC, d.rR, d.rT, d.rF, d.rP, d.npp};
  memcpy(out, v, sizeof(v));
}
int cfzc_emu_half_bandwidth(const cfzc::CSpec *sp) { return order_half_bandwidth(sp); }
// working set at X's poses (sel: np * n_obs + npp bytes; first = 1: from scratch, else with hysteresis from the codes in sel)
void cfzc_emu_select(const cfzc::CSpec *sp, const double *X, unsigned char *sel, int first) {
  const cfzc::CDims d = cfzc::cdims(*sp);
  double *slab = (double *)calloc(cfzc::work_doubles(*sp, 1), sizeof(double));
  cfzc::CWork w = cfzc::carve(*sp, 1, slab);
  memcpy(w.x, X, sizeof(double) * d.n);
  memcpy(w.sel, sel, d.np * sp->n_obs + d.npp);
  cfzc::refresh_working_set(*sp, w, w.x, 1e-3, first != 0);
  memcpy(sel, w.sel, d.np * sp->n_obs + d.npp);
  free(slab);
}
void cfzc_emu_eval(const cfzc::CSpec *sp, const unsigned char *sel, const double *X, const double *nu, double *f, double *c,
                   double *g, double *jtnu) {
  *f = cfzc::objective(*sp, X);
  cfzc::constraints(*sp, sel, X, c);
  cfzc::gradient(*sp, X, g);
  cfzc::jt_nu(*sp, sel, X, nu, jtnu);
}
// dense (n+m)^2 matrix [[W + diag(sig) + delta I, J'], [J, -reg_dual I]] in the natural ordering, from the band and the
// border, with the collision slacks and rows condensed into the pose blocks (their own rows and columns stay zero);
// returns the half-bandwidth the ordering needs
int cfzc_emu_kkt(const cfzc::CSpec *sp, const unsigned char *sel, const double *X, const double *nu, const double *sig,
                 double delta, double *K) {
  const cfzc::CDims d = cfzc::cdims(*sp);
  int kb = order_half_bandwidth(sp) + 64;
  if (kb > d.nk - 1) kb = d.nk - 1;
  double *slab = (double *)calloc(cfzc::work_doubles(*sp, kb), sizeof(double));
  cfzc::CWork w = cfzc::carve(*sp, kb, slab);
  cfzc::build_order(*sp, w.posx, w.posc);
  memcpy(w.x, X, sizeof(double) * d.n); memcpy(w.nu, nu, sizeof(double) * d.m); memcpy(w.sig, sig, sizeof(double) * d.n);
  memcpy(w.sel, sel, d.np * sp->n_obs + d.npp);
  const cfzc::Band Bd = {w.ab, kb, 3 * kb + 1, 2 * kb};
  const double hdd = cfzc::assemble(*sp, w, Bd, delta);
  const int nt = d.n + d.m;
  int *nat = (int *)malloc(sizeof(int) * d.nk);  // band position -> natural index
  for (int i = 0; i < d.n; ++i) if (w.posx[i] >= 0) nat[w.posx[i]] = i;
  for (int i = 0; i < d.m; ++i) if (w.posc[i] >= 0) nat[w.posc[i]] = d.n + i;
  int bw = 0;
  memset(K, 0, sizeof(double) * nt * nt);
  for (int a = 0; a < d.nk; ++a)
    for (int b = (a - kb > 0 ? a - kb : 0); b <= (a + kb < d.nk - 1 ? a + kb : d.nk - 1); ++b) {
      const double v = cfzc::bnd(Bd, a, b);
      if (v != 0.0) { K[(size_t)nat[a] * nt + nat[b]] = v; const int df = a > b ? a - b : b - a; if (df > bw) bw = df; }
    }
  for (int a = 0; a < d.nk; ++a) { K[(size_t)nat[a] * nt + d.iDt] = w.bord[a]; K[(size_t)d.iDt * nt + nat[a]] = w.bord[a]; }
  K[(size_t)d.iDt * nt + d.iDt] = hdd + sig[d.iDt];
  free(nat); free(slab);
  return bw;
}
int cfzc_emu_solve(const cfzc::CSpec *sp, double *X, int *out_i, double *out_d) {
  const int kb = order_half_bandwidth(sp);
  if (kb < 0) return -2;
  double *slab = (double *)calloc(cfzc::work_doubles(*sp, kb), sizeof(double));
  if (!slab) return -1;
  cfzc::solve_colloc<0>(*sp, X, slab, kb, out_i, out_d, 0);
  free(slab);
  return 0;
}
}
