// Banded LU with partial pivoting in LAPACK gb layout (kl = ku = kb, ld = 3 kb + 1 rows per column), shared by the planning
// solvers (cfz_plan.inl: state_ws, cfz_colloc.inl: collocation plan).  GPU part: one wavefront, window in LDS.
#pragma once

namespace cfzb {

struct Band { double *ab; int kb, ld; };

#if defined(__HIP_DEVICE_COMPILE__)
// The two routines below are functions of their own (registers of their own: inlined into a solver their loops reloaded
// spilled values from scratch at every pivot).  On this toolchain such a function must not NAME any LDS (see
// cfz_colloc.inl), so the window comes in as an address-space-3 pointer from an inlined wrapper that names it, and the band
// as an address-space-1 pointer so that it is read with GLOBAL rather than FLAT instructions.
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(1))) double glb_f64;
typedef __attribute__((address_space(1))) int glb_i32;
#define CFZB_LDS_FN __device__ __attribute__((noinline))
// Elimination by one wavefront with the kv + 1 columns it is working on in LDS (the kernel's dynamic LDS: (kv + 1) x ld
// doubles plus one spare slot per lane; 124 KiB for the collocation plan, 78 KiB for state_ws): column q lives in slot q mod 103 while j <= q <= j + kv, enters from `ab` when pivot step j = q - kv - 1
// ends (fetched into registers at its start) and is written back after its own pivot step.  Lane i owns row j + i of the
// pivot column and of every column it updates, so all LDS traffic of the rank-1 update is unit stride.
// orders the LDS traffic of the one wavefront that runs the solver: DS instructions of a wavefront execute in issue
// order, so only the compiler has to be kept from moving accesses across (no s_waitcnt on outstanding global stores)
__device__ inline void wave_sync() {
#if defined(CFZC_FULL_SYNC)
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
}

CFZB_LDS_FN int band_factor_lds_core(glb_f64 *ab, int kb, int ld, int n, glb_i32 *ipiv, lds_f64 *cfzb_lds, lds_f64 *ptk) {
  const int kl = kb, kv = 2 * kb, wc = kv + 1, lane = threadIdx.x;
  {
    const int cnt = ((kv < n - 1 ? kv : n - 1) + 1) * ld;
    for (int t = lane; t < cnt; t += 64) cfzb_lds[t] = ab[t];
  }
  __syncthreads();
  int ju = 0, sj = 0;  // sj = j mod wc: slot of column j; column q sits in slot sj + (q - j), wrapped
  for (int j = 0; j < n; ++j, sj = sj + 1 == wc ? 0 : sj + 1) {
    const int km = (kl < n - 1 - j) ? kl : n - 1 - j, qn = j + kv + 1;
    long long tp0 = (long long)wall_clock64();
    double pre[3] = {0.0, 0.0, 0.0};
    if (qn < n) for (int t = 0; t < 3; ++t) { const int r = lane + 64 * t; if (r < ld) pre[t] = ab[(size_t)qn * ld + r]; }
    const int cj = sj * ld;  // offset of column j in the window
    double best = lane <= km ? fabs(cfzb_lds[cj + kv + lane]) : -1.0;
    int jp = lane;
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off); const int oj = __shfl_xor(jp, off);
      if (ob > best || (ob == best && oj < jp)) { best = ob; jp = oj; }
    }
    if (lane == 0) ipiv[j] = j + jp;
    if (!(best > 0.0)) return 1;
    const int reach = j + kl + jp < n - 1 ? j + kl + jp : n - 1;
    ju = ju > reach ? ju : reach;
    if (jp != 0) {
      for (int dq = lane; j + dq <= ju; dq += 64) {
        const int sl = sj + dq < wc ? sj + dq : sj + dq - wc;
        const int cq = sl * ld + (kv - dq);  // row j of column j + dq
        const double t = cfzb_lds[cq]; cfzb_lds[cq] = cfzb_lds[cq + jp]; cfzb_lds[cq + jp] = t;
      }
      wave_sync();
    }
    { const long long t1 = (long long)wall_clock64(); if (lane == 0) ptk[0] += (double)(t1 - tp0); tp0 = t1; }
    const bool mine = lane >= 1 && lane <= km;
    const double inv = 1.0 / cfzb_lds[cj + kv];
    const double l = mine ? cfzb_lds[cj + kv + lane] * inv : 0.0;
    if (mine) cfzb_lds[cj + kv + lane] = l;
    // rank-1 update of the columns j+1..ju whose entry in the pivot row is not zero (typically a third of them).  Lane t
    // fetches the multipliers u of columns j+1+t and j+65+t; the columns with u != 0 are then taken sixteen at a time,
    // branch-free, so that the sixteen reads and then the sixteen writes of a batch are in flight together (a branch
    // around a write costs an s_waitcnt lgkmcnt(0), i.e. one LDS round trip per column): a short batch repeats its
    // last column (the same value is stored twice), lanes without a row read and write a spare slot behind the window
    const int nq = ju - j;
    for (int half = 0; half < 2; ++half) {
      const int dl = 1 + 64 * half + lane;
      double um = 0.0;
      if (dl <= nq) { const int sl = sj + dl < wc ? sj + dl : sj + dl - wc; um = cfzb_lds[sl * ld + (kv - dl)]; }
      unsigned long long todo = __ballot(um != 0.0);
      const int uh = __double2hiint(um), ul = __double2loint(um);
      while (todo) {
        int tq[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          tq[c] = todo ? (int)__builtin_ctzll(todo) : tq[c ? c - 1 : 0];
          todo &= todo - 1;  // 0 stays 0
        }
        double xv[16];
        int at[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const int dq = 1 + 64 * half + tq[c];
          const int sl = sj + dq < wc ? sj + dq : sj + dq - wc;
          at[c] = mine ? sl * ld + (kv - dq) + lane : wc * ld + lane;
          xv[c] = cfzb_lds[at[c]];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const double u = __hiloint2double(__builtin_amdgcn_readlane(uh, tq[c]), __builtin_amdgcn_readlane(ul, tq[c]));
          cfzb_lds[at[c]] = xv[c] - l * u;
        }
      }
    }
    wave_sync();
    { const long long t1 = (long long)wall_clock64(); if (lane == 0) ptk[1] += (double)(t1 - tp0); tp0 = t1; }
    for (int t = 0; t < 3; ++t) {
      const int r = lane + 64 * t;
      if (r < ld) { ab[(size_t)j * ld + r] = cfzb_lds[cj + r]; if (qn < n) cfzb_lds[cj + r] = pre[t]; }
    }
    wave_sync();
    { const long long t1 = (long long)wall_clock64(); if (lane == 0) ptk[2] += (double)(t1 - tp0); tp0 = t1; }
  }
  __syncthreads();
  return 0;
}

// the right-hand side(s) in LDS (b at offset 0, b2 at offset n when TWO), the factor's columns fetched eight pivot steps ahead
template <bool TWO>
CFZB_LDS_FN void band_substitute_lds_core(const glb_f64 *ab, int kb, int ld, int n, const glb_i32 *ipiv, glb_f64 *b, glb_f64 *b2, lds_f64 *cfzb_lds) {
  const int kl = kb, kv = 2 * kb, lane = threadIdx.x;
  for (int t = lane; t < n; t += 64) { cfzb_lds[t] = b[t]; if (TWO) cfzb_lds[n + t] = b2[t]; }
  __syncthreads();
  constexpr int CH = 8;
  for (int j0 = 0; j0 < n; j0 += CH) {
    double Lr[CH]; int pv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int j = j0 + c, km = j < n ? ((kl < n - 1 - j) ? kl : n - 1 - j) : 0;
      Lr[c] = (lane >= 1 && lane <= km) ? ab[(size_t)j * ld + kv + lane] : 0.0;
      pv[c] = j < n ? ipiv[j] : j;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int j = j0 + c;
      if (j < n) {
        const int km = (kl < n - 1 - j) ? kl : n - 1 - j, p = pv[c];
        if (p != j) {
          if (lane == 0) {
            const double t = cfzb_lds[j]; cfzb_lds[j] = cfzb_lds[p]; cfzb_lds[p] = t;
            if (TWO) { const double t2 = cfzb_lds[n + j]; cfzb_lds[n + j] = cfzb_lds[n + p]; cfzb_lds[n + p] = t2; }
          }
          wave_sync();
        }
        const double bj = cfzb_lds[j], cj = TWO ? cfzb_lds[n + j] : 0.0;
        if (lane >= 1 && lane <= km) { cfzb_lds[j + lane] -= Lr[c] * bj; if (TWO) cfzb_lds[n + j + lane] -= Lr[c] * cj; }
        wave_sync();
      }
    }
  }
  for (int j1 = n - 1; j1 >= 0; j1 -= CH) {
    double Ur[CH][2], dg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int j = j1 - c;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int off = lane + 64 * t, i = j - kv + off;
        Ur[c][t] = (j >= 0 && off < kv && i >= 0) ? ab[(size_t)j * ld + off] : 0.0;
      }
      dg[c] = j >= 0 ? ab[(size_t)j * ld + kv] : 1.0;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int j = j1 - c;
      if (j >= 0) {
        const double bj = cfzb_lds[j] / dg[c], cj = TWO ? cfzb_lds[n + j] / dg[c] : 0.0;
        wave_sync();
        if (lane == 0) { cfzb_lds[j] = bj; if (TWO) cfzb_lds[n + j] = cj; }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int off = lane + 64 * t, i = j - kv + off;
          if (off < kv && i >= 0) { cfzb_lds[i] -= Ur[c][t] * bj; if (TWO) cfzb_lds[n + i] -= Ur[c][t] * cj; }
        }
        wave_sync();
      }
    }
  }
  for (int t = lane; t < n; t += 64) { b[t] = cfzb_lds[t]; if (TWO) b2[t] = cfzb_lds[n + t]; }
  __syncthreads();
}
// the wrappers are inlined into the kernel and name its dynamic LDS
__device__ inline int band_factor_lds(const Band &B, int n, int *ipiv, long long *ptk) {
  extern __shared__ double cfzb_dyn[];
  __shared__ double tks[3];
  if (threadIdx.x == 0) { tks[0] = 0.0; tks[1] = 0.0; tks[2] = 0.0; }
  __syncthreads();
  const int fail = band_factor_lds_core((glb_f64 *)B.ab, B.kb, B.ld, n, (glb_i32 *)ipiv, (lds_f64 *)cfzb_dyn, (lds_f64 *)tks);
  __syncthreads();
  for (int i = 0; i < 3; ++i) ptk[i] += (long long)tks[i];
  return fail;
}
template <bool TWO>
__device__ inline void band_substitute_lds(const Band &B, int n, const int *ipiv, double *b, double *b2) {
  extern __shared__ double cfzb_dyn[];
  band_substitute_lds_core<TWO>((const glb_f64 *)B.ab, B.kb, B.ld, n, (const glb_i32 *)ipiv, (glb_f64 *)b, (glb_f64 *)b2, (lds_f64 *)cfzb_dyn);
}
#endif

}  // namespace cfzb
