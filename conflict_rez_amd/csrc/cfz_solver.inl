// cfz_solver.inl -- one MPC-step NLP solved by one 128-lane workgroup (two wavefronts).
//
// Included by cfz_engine.hip (gfx950 device code: iterate and stage data in LDS) and by tests/emu/cfz_emu.cpp (same
// source, lanes run as a loop, so the kernel logic can be checked and sanitised on a CPU).  Not a CPU fallback: the
// product library only ever contains the device build.
//
// NLP: reference confrez/control/vehicle_follower.py:146-368 with the OBCA duals eliminated into closed-form separation
// certificates; dynamics confrez/control/dynamic_model.py:5-58; algorithm DESIGN.md "CFZ-IPM" (interior point, filter
// line search, slack elimination + Riccati recursion).
//
// Lane map: lane = 4 k + sub.  The four lanes of a DPP quad own stage k (N <= 32 stages):
//   rows      lane (k, sub) owns the separation blocks j = sub, sub + 4, .. of its stage (working set, row values, slack
//             and multiplier steps, their share of the condensed stage Hessian); the shares meet in a quad sum (two DPP
//             quad_perm exchanges, no LDS, no barrier)
//   dynamics  all four lanes integrate the stage (RK4, identical instruction stream, so the redundancy costs no time) and
//             each propagates its own columns of the sensitivities: lane sub column sub, every lane column 4
//   stage     condensed H_k, g_k, bound multipliers, costs: computed by the whole quad, written by sub 0
//   Riccati   backward sweep: 30 dependent stages, out of line (registers of its own) -- since round 5 on the matrix cores, the first
//             wavefront's 64 lanes holding the stage's 16 x 16 tiles (riccati_backward_mfma; the one-lane sweep remains as
//             -DCFZ_RICCATI_SCALAR); forward step and costates: linear recurrences once the gains are known -> Kogge-Stone scans,
//             one stage per lane of wavefront 0
// (CFZ_LPS = 8: eight lanes per stage, four wavefronts per instance -- round 6's go / no-go, a diagnostic build; the product is 4.)
// Reductions over the workgroup run on registers: DPP butterflies inside a row of 16 lanes, v_readlane across the four
// rows of a wavefront, one 16-double LDS exchange between the two wavefronts.
//
// Conventions: code inside CFZ_LANES(tid){...}CFZ_END runs once per lane and may only write lane-private locals or
// workspace cells it owns; everything outside runs uniformly (every lane computes the same value).  A lane-private value
// that another lane of the quad (CFZ_QSUM) or a reduction (CFZ_REDUCE) reads goes through a CFZ_PART array: registers on
// the device, [lane] arrays in the CPU build; CFZ_MID separates the producing from the consuming half of a lane block
// (nothing on the device; the CPU build closes the lane loop and opens it again, so that all lanes have produced).

#ifndef CFZ_SOLVER_INL
#define CFZ_SOLVER_INL

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CFZ_FN __host__ __device__ __forceinline__
#define CFZ_CALL __host__ __device__ __forceinline__  // measured: out-of-line helpers with array arguments cost 30 % (the arrays go to scratch)
#else
#define CFZ_FN static inline
#define CFZ_CALL static inline
#endif

#ifndef CFZ_LPS
#define CFZ_LPS 4
#endif
namespace cfz {
constexpr int kLPS = CFZ_LPS;  // lanes per stage: 4 (a DPP quad, two wavefronts per instance) or 8 (half a DPP row, four wavefronts)
static_assert(kLPS == 4 || kLPS == 8, "lanes per stage");
constexpr int kLPSBits = kLPS == 4 ? 2 : 3;
constexpr int kMaxN = 32;           // stages
constexpr int kNL = kMaxN * kLPS;   // lanes per instance
constexpr int kNW = kNL / 64;       // wavefronts per instance
constexpr int kXS = 7;              // values one workgroup reduction can carry (slots per wavefront in the exchange area)
}  // namespace cfz

#if defined(__HIP_DEVICE_COMPILE__)
#define CFZ_LANES(tid) { const int tid = (int)threadIdx.x;
#define CFZ_MID } { const int tid = (int)threadIdx.x;
#define CFZ_END } __syncthreads();
// uniform code (every wavefront runs it on its own) that has READ workspace words which a lane is about to overwrite: the
// wavefronts meet first, or the slower one reads the new words, decides differently and the two fall out of step at the barriers
#define CFZ_SYNC() __syncthreads()
#define CFZ_PART(name, n) double name[n]
#define CFZ_P(name, i) name[i]
#define CFZ_QSUM(name, i) cfz::quad_sum(name[i])
#define CFZ_REDUCE(NS, NX, NI, part, out) cfz::reduce_all<NS, NX, NI>(m, L, xpar, part, out)
// the sweeps run on lane 0 of the workgroup, out of line; every lane calls (uniform control flow), lane 0 works
#define CFZ_SERIAL(call) do { call; __syncthreads(); } while (0)
#define CFZ_WAVE0(call) do { if (threadIdx.x < cfz::kMaxN) { call; } __syncthreads(); } while (0)  // the lanes that own a stage
#define CFZ_WSP(p) cfz::opaque_wsp((cfz::wsp_f64 *)(p))
#define CFZ_UNIFORM(v) cfz::uniform_value(v)
#else
#define CFZ_LANES(tid) for (int tid = 0; tid < cfz::kNL; ++tid) {
#define CFZ_MID } for (int tid = 0; tid < cfz::kNL; ++tid) {
#define CFZ_END }
#define CFZ_SYNC() do { } while (0)
#define CFZ_PART(name, n) double name[n][cfz::kNL]
#define CFZ_P(name, i) name[i][tid]
#define CFZ_QSUM(name, i) cfz::quad_sum_emu(name[i], tid)
#define CFZ_REDUCE(NS, NX, NI, part, out) cfz::reduce_all_emu<NS, NX, NI>(part, out)
#define CFZ_SERIAL(call) do { call; } while (0)
#define CFZ_WAVE0(call) do { call; } while (0)
#define CFZ_WSP(p) (p)
#define CFZ_UNIFORM(v) (v)
#endif

// Diagnostic build only (-DCFZ_STAMPS): shader-clock cycles per phase of the solver, summed over the
// iterations of one instance and written to a debug buffer (never to an output).
#if defined(CFZ_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define CFZ_STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_last; stamp_last = __builtin_amdgcn_s_memtime(); } while (0)
#define CFZ_STAMP_DECL unsigned long long stamp_acc[24] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; const unsigned long long stamp_wall0 = wall_clock64(); unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#else
#define CFZ_STAMP(i) do {} while (0)
#define CFZ_STAMP_DECL
#endif

namespace cfz {

constexpr int kNP = 7;      // x y psi v delta a w
constexpr int kMaxObs = 8;  // = CFZ_MAX_OBS

// Everything the kernel needs besides per-instance data (plain old data, passed by value).
struct KSpec {
  int N, n_obs, n_nbr, rk_substeps;
  int max_iter, max_backtrack, filter_cap, stall_iters;
  int row_curvature, vv_rows;  // vv_rows 1: vertex-vertex rows (kind 3) in the working set
  int stag_win, err_stall;       // err_stall: iterations without a halving of the error after which the solve ends with status 5 (0: never)
                                 // stag_win: iterations without a halving of the error, at a feasible iterate, after which the late shift may start
                                 // at iteration kShiftStagMin already (0: never; oracle/ipm.py shift_stagnation)
  int carry_shift, pad_ks;       // carry_shift: a converged solve that shifted some stage tells its successor to shift from the start (oracle/ipm.py)
  int shift_after, resto;        // shift_after: iteration from which a stage whose row curvature would be scaled is shifted instead (0: never)
                                 // resto: restoration phases a solve may go through (0: none; oracle/ipm.py restoration)
  double dt, wb, dmin;
  double g[4], bounds[12], weights[6];
  double A_obs[kMaxObs][4][2], b_obs[kMaxObs][4], V_obs[kMaxObs][4][2];
  double tol, constr_viol_tol, dual_inf_tol, compl_inf_tol, mu_init, kappa_eps, kappa_mu, theta_mu, tau_min,
      bound_push, bound_frac, s_max, kappa_sigma, eta_phi, gamma_theta, gamma_phi, delta_sw, s_theta, s_phi,
      reg_primal, stall_kappa, warm_push,
      reg_dual_rows,  // IPOPT's dual regularisation delta_c on the separation rows (oracle/ipm.py reg_dual_rows)
      resto_first;    // a start whose rows are violated by more than this goes through the restoration phase first (0: never)
  // static obstacles as the kernel reads them, n_obs x 20 doubles in global memory: A[4][2], b[4], V[4][2]
  // (L1/L2-resident; indexing the arrays above with a lane-varying j would copy this struct to scratch)
  const double *obs_tab;
};

// Constants derived from the spec, formed on the HOST and passed to the kernel beside it (scalar registers).  Formed in the
// kernel they are loop-invariant vector-pipe results: the compiler hoists them out of the iteration loop into VGPR pairs,
// two dozen of them, and spills them to scratch.
struct KDer {
  double w2[6];        // 2 w_i
  double hb[5];        // constant part of the stage Hessian diagonal: 2 w0 + reg, 2 w1 + reg, 2 w2 + reg, 2 w5 + reg, 2 w3 + reg
  double mg, q0, q1, q2, iq0, iq1;  // convexity safeguard of the row curvature (assembly)
  double iks;          // 1 / kappa_sigma
  double rk_h, rk_hh, rk_h6, iwb;   // dt / M, half of it, a sixth of it, 1 / wheelbase
  double mu_floor;
};
CFZ_FN KDer derive(const KSpec &sp) {
  KDer d;
  const double *w = sp.weights;
  for (int i = 0; i < 6; ++i) d.w2[i] = 2 * w[i];
  d.hb[0] = 2 * w[0] + sp.reg_primal; d.hb[1] = 2 * w[1] + sp.reg_primal; d.hb[2] = 2 * w[2] + sp.reg_primal;
  d.hb[3] = 2 * w[5] + sp.reg_primal; d.hb[4] = 2 * w[3] + sp.reg_primal;
  d.mg = 0.2 * fmin(w[0], fmin(w[1], w[2]));
  d.q0 = 2 * w[0] - d.mg; d.q1 = 2 * w[1] - d.mg; d.q2 = 2 * w[2] - d.mg;
  d.iq0 = 1.0 / d.q0; d.iq1 = 1.0 / d.q1;
  d.iks = 1.0 / sp.kappa_sigma;
  d.rk_h = sp.dt / sp.rk_substeps; d.rk_hh = 0.5 * d.rk_h; d.rk_h6 = d.rk_h * (1.0 / 6.0); d.iwb = 1.0 / sp.wb;
  d.mu_floor = fmin(sp.tol, sp.compl_inf_tol) / (sp.kappa_eps + 1.0);
  return d;
}

// Workspace layout (offsets in doubles) for one instance.
struct Lay {
  int N, nb, nr;                            // stages, blocks per stage, rows per stage (2 per block)
  int p, sg, nuc, zs, zl, zu, pi0, pi;      // iterate
  int dp, dsg, dpi0, dpi;                   // step
  int cj, ab, d, hc, gk, kk;                // stage data
  int sel, ref, nb4, x0, cs, rP, filt, xw, total;
};

CFZ_FN Lay make_layout(int N, int nb, int n_nbr) {
  Lay L; int o = 0;
  const int nr = 2 * nb;
  L.N = N; L.nb = nb; L.nr = nr;
  L.p = o; o += N * kNP;
  L.sg = o; o += N * nr; L.nuc = o; o += N * nr; L.zs = o; o += N * nr;
  L.zl = o; o += N * 6; L.zu = o; o += N * 6;
  L.pi0 = o; o += 5; L.pi = o; o += N * 5;
  L.dp = o; o += N * kNP; L.dpi0 = o; o += 5;
  L.cj = o; o += N * nr; L.dsg = L.cj;  // the slack step overwrites the row residual it is computed from
  L.ab = o; o += N * 15; L.d = o; o += N * 5;
  L.dpi = L.d;  // the costate sweep (last reader of the defects d_k was the forward sweep) overwrites them with d(pi)
  L.hc = o; o += N * 11; L.gk = o; o += N * kNP; L.kk = o; o += N * 12;
  L.sel = o; o += (N * nb + 7) / 8;  // working set codes (<= 255), one byte each
  L.ref = 0;  // the reference stays in global memory (read-only, L2-resident)
  L.nb4 = o; o += N * n_nbr * 4; L.x0 = o; o += 5;
  L.cs = o; o += (2 * N > 32) ? 2 * N : 32;  // cos, sin of the pose heading of every stage at the current iterate
  // value function of stage 0 (30 numbers) between the backward sweep and the forward scan: in the step's slots of stages 1.. (dead until the
  // scan writes them, and the scan reads the value function first); for horizons too short for that it shares the cos / sin slots, whose
  // readers after the sweeps then form the heading's cosine and sine again (rounds 1-5: always)
  L.rP = (N - 1) * kNP >= 30 ? L.dp + kNP : L.cs;
  L.filt = o; o += 32;
  L.xw = o; o += 2 * kXS * kNW;  // exchange between the wavefronts of a reduction: [parity 2][wavefront kNW][kXS values]
  L.total = o;  // N = 30, 9 blocks: 5,119 doubles = 40,952 B = 20 LDS granules of 2 KiB (8 bytes to spare): four instances per CU (cfz_create)
  return L;
}

CFZ_FN unsigned char *sel_ptr(double *m, const Lay &L) { return reinterpret_cast<unsigned char *>(m + L.sel); }
CFZ_FN const unsigned char *sel_ptr(const double *m, const Lay &L) { return reinterpret_cast<const unsigned char *>(m + L.sel); }

// bounded columns of p: x y v delta a w  (psi is free)
CFZ_FN int bcol(int q) { return q < 2 ? q : q + 1; }

// ------------------------------------------------------------------------------ reductions
// One tree for both builds, so that the CPU build of this source sums in the order the wavefronts do:
//   inside a row of 16 lanes: partner lane^1, lane^2 (quad_perm), mirror inside 8 (row_half_mirror), mirror inside 16
//   (row_mirror); then (row0 . row1) . (row2 . row3) of a wavefront; then wavefront 0 . wavefront 1.
// OP 0 sum, 1 max, 2 min.  Every lane ends with the same bits (each level combines a pair commutatively).
template <int OP> CFZ_FN double op2(double a, double b) { return OP == 0 ? a + b : (OP == 1 ? fmax(a, b) : fmin(a, b)); }

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) double wsp_f64;  // the workspace as the out-of-line sweeps see it: DS instructions, no FLAT
// ... handed over as a value the compiler cannot see through: otherwise interprocedural constant propagation moves the name of the
// kernel's dynamic LDS into the sweeps, which then look it up in llvm.amdgcn.dynlds.offset.table at every call (cfz_band.inl: opaque)
__device__ __forceinline__ wsp_f64 *opaque_wsp(wsp_f64 *p) { asm volatile("" : "+v"(p)); return p; }
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {  // sum over the kLPS lanes of a stage
  v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
  if (kLPS == 8) v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad's sum
  return v;
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// a value every lane holds identically -> scalar registers (the compiler cannot know it is uniform after an LDS read; as a
// vector value each of the solver's two dozen scalars would occupy a VGPR pair for the whole iteration and spill)
__device__ __forceinline__ double uniform_value(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int OP> __device__ __forceinline__ double wave_reduce(double v) {
  v = op2<OP>(v, dpp_mov<0xB1>(v));
  v = op2<OP>(v, dpp_mov<0x4E>(v));
  v = op2<OP>(v, dpp_mov<0x141>(v));  // row_half_mirror
  v = op2<OP>(v, dpp_mov<0x140>(v));  // row_mirror
  return op2<OP>(op2<OP>(lane_value(v, 0), lane_value(v, 16)), op2<OP>(lane_value(v, 32), lane_value(v, 48)));
}
// part[0..NS) are summed, part[NS..NS+NX) maximised, the next NI minimised, over all lanes of the workgroup.  One
// barrier; the exchange buffer alternates (xpar) so that a wavefront may already write the next reduction while the
// other still reads this one.
template <int NS, int NX, int NI>
__device__ __forceinline__ void reduce_all(double *m, const Lay &L, int &xpar, const double *part, double *out) {
  constexpr int n = NS + NX + NI;
  static_assert(n <= kXS, "exchange slots");
  double w[n];
#pragma unroll
  for (int i = 0; i < n; ++i) w[i] = i < NS ? wave_reduce<0>(part[i]) : (i < NS + NX ? wave_reduce<1>(part[i]) : wave_reduce<2>(part[i]));
  double *x = m + L.xw + xpar * kXS * kNW;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < n; ++i) x[(threadIdx.x >> 6) * kXS + i] = w[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < n; ++i) {
    if (kNW == 2) out[i] = uniform_value(i < NS ? op2<0>(x[i], x[kXS + i]) : (i < NS + NX ? op2<1>(x[i], x[kXS + i]) : op2<2>(x[i], x[kXS + i])));
    else out[i] = uniform_value(i < NS ? op2<0>(op2<0>(x[i], x[kXS + i]), op2<0>(x[2 * kXS + i], x[3 * kXS + i]))
                                : (i < NS + NX ? op2<1>(op2<1>(x[i], x[kXS + i]), op2<1>(x[2 * kXS + i], x[3 * kXS + i]))
                                               : op2<2>(op2<2>(x[i], x[kXS + i]), op2<2>(x[2 * kXS + i], x[3 * kXS + i]))));
  }
  xpar ^= 1;
}
#else
typedef double wsp_f64;
static inline double quad_sum_emu(const double *v, int tid) {
  const int q = tid & ~3;
  const double mine = (tid & 2) ? (v[q + 2] + v[q + 3]) + (v[q] + v[q + 1]) : (v[q] + v[q + 1]) + (v[q + 2] + v[q + 3]);
  if (kLPS == 4) return mine;
  const int o = q ^ 4;  // the other quad of the stage's eight lanes (row_half_mirror: lane i meets lane 7 - i)
  const double other = (tid & 2) ? (v[o] + v[o + 1]) + (v[o + 2] + v[o + 3]) : (v[o + 2] + v[o + 3]) + (v[o] + v[o + 1]);
  return mine + other;
}
template <int OP> static inline double wave_reduce_emu(const double *v) {
  double a[64], b[64];
  for (int i = 0; i < 64; ++i) b[i] = op2<OP>(v[i], v[i ^ 1]);
  for (int i = 0; i < 64; ++i) a[i] = op2<OP>(b[i], b[i ^ 2]);
  for (int i = 0; i < 64; ++i) b[i] = op2<OP>(a[i], a[(i & ~7) | (7 - (i & 7))]);
  for (int i = 0; i < 64; ++i) a[i] = op2<OP>(b[i], b[(i & ~15) | (15 - (i & 15))]);
  return op2<OP>(op2<OP>(a[0], a[16]), op2<OP>(a[32], a[48]));
}
template <int NS, int NX, int NI> static inline void reduce_all_emu(const double (*part)[kNL], double *out) {
  for (int i = 0; i < NS + NX + NI; ++i) {
    if (kNW == 2) {
      if (i < NS) out[i] = op2<0>(wave_reduce_emu<0>(part[i]), wave_reduce_emu<0>(part[i] + 64));
      else if (i < NS + NX) out[i] = op2<1>(wave_reduce_emu<1>(part[i]), wave_reduce_emu<1>(part[i] + 64));
      else out[i] = op2<2>(wave_reduce_emu<2>(part[i]), wave_reduce_emu<2>(part[i] + 64));
    } else {
      if (i < NS) out[i] = op2<0>(op2<0>(wave_reduce_emu<0>(part[i]), wave_reduce_emu<0>(part[i] + 64)), op2<0>(wave_reduce_emu<0>(part[i] + 128), wave_reduce_emu<0>(part[i] + 192)));
      else if (i < NS + NX) out[i] = op2<1>(op2<1>(wave_reduce_emu<1>(part[i]), wave_reduce_emu<1>(part[i] + 64)), op2<1>(wave_reduce_emu<1>(part[i] + 128), wave_reduce_emu<1>(part[i] + 192)));
      else out[i] = op2<2>(op2<2>(wave_reduce_emu<2>(part[i]), wave_reduce_emu<2>(part[i] + 64)), op2<2>(wave_reduce_emu<2>(part[i] + 128), wave_reduce_emu<2>(part[i] + 192)));
    }
  }
}
#endif

// ------------------------------------------------------------------------------ dynamics
// RK4 (M sub-steps) of the kinematic bicycle.  The state rows v, delta integrate exactly
// (v+ = v + a t, delta+ = delta + w t) and x, y never feed back, so only the sensitivities of
// (x, y, psi) with respect to (psi0, v0, delta0, a, w) are propagated: S[3][5].
// sin, cos of a small angle by Taylor series (|e| <= 0.06: truncation < 1e-21), sincos otherwise
CFZ_FN void small_sincos(double e, double *s, double *c) {
  if (fabs(e) > 0.06) { sincos(e, s, c); return; }
  // Horner with the reciprocal factorial ratios as constants: written as x / 6.0 these were nine real double-precision
  // divisions (ten dependent instructions each) per call, 192 per RK4 interval
  const double e2 = e * e;
  *s = e * (1.0 - e2 * (1.0 / 6.0) * (1.0 - e2 * (1.0 / 20.0) * (1.0 - e2 * (1.0 / 42.0) * (1.0 - e2 * (1.0 / 72.0)))));
  *c = 1.0 - e2 * 0.5 * (1.0 - e2 * (1.0 / 12.0) * (1.0 - e2 * (1.0 / 30.0) * (1.0 - e2 * (1.0 / 56.0) * (1.0 - e2 * (1.0 / 90.0)))));
}

// Three library sincos calls per interval instead of 32: the steering angle advances by w h/2 between RK
// stage points (one fixed rotation), the heading by small increments (|h psi'| <= 0.03 rad inside the
// actuator limits), so the stage-point sines/cosines come from rotating the previous ones.
template <bool SENS>
CFZ_CALL void rk4_step_h(const double z[5], double a, double w, double h, double hh, double h6, double iwb, int M, double out[5],
                       double S[3][5]);
template <bool SENS>
CFZ_CALL void rk4_step(const double z[5], double a, double w, double dt, double wb, int M, double out[5],
                     double S[3][5]) {
  const double h = dt / M;
  rk4_step_h<SENS>(z, a, w, h, 0.5 * h, h * (1.0 / 6.0), 1.0 / wb, M, out, S);
}
// the same with the step constants formed by the caller (the solver passes host-derived ones, KDer)
template <bool SENS>
CFZ_CALL void rk4_step_h(const double z[5], double a, double w, double h, double hh, double h6, double iwb, int M, double out[5],
                       double S[3][5]) {
  double x = z[0], y = z[1], psi = z[2], v = z[3], de = z[4];
  if (SENS) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 5; ++c) S[r][c] = 0.0;
    S[2][0] = 1.0;
  }
  double sp_, cp_, sd0, cd0, sh, ch;  // heading and steering angle at the sub-step start, half-step steering rotation
  sincos(psi, &sp_, &cp_);
  sincos(de, &sd0, &cd0);
  small_sincos(hh * w, &sh, &ch);
  double tsub = 0.0;  // time since the start of the interval: dv/da = ddelta/dw = tsub
  for (int m = 0; m < M; ++m) {
    double ax = 0, ay = 0, ap = 0;  // weighted stage sums
    double AS[3][5];
    if (SENS)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 5; ++c) AS[r][c] = 0.0;
    double kp = 0;  // previous stage derivative of psi
    double KS[3][5];
    if (SENS)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 5; ++c) KS[r][c] = 0.0;
    // steering angle at the three distinct stage times of this sub-step: +0, +h/2, +h
    const double sd1 = sd0 * ch + cd0 * sh, cd1 = cd0 * ch - sd0 * sh;
    const double sd2 = sd1 * ch + cd1 * sh, cd2 = cd1 * ch - sd1 * sh;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const double wprev = (st == 0) ? 0.0 : ((st == 3) ? h : hh);
      const double wsum = (st == 0 || st == 3) ? 1.0 : 2.0;
      const double vs = v + wprev * a;
      double s = sp_, c = cp_;
      if (st > 0) {
        double se, ce;
        small_sincos(wprev * kp, &se, &ce);
        s = sp_ * ce + cp_ * se; c = cp_ * ce - sp_ * se;
      }
      const double sd = (st == 0) ? sd0 : ((st == 3) ? sd2 : sd1), cd = (st == 0) ? cd0 : ((st == 3) ? cd2 : cd1);
      const double t = sd / cd;
      const double fx = vs * c, fy = vs * s, fp = vs * iwb * t;
      if (SENS) {
        // stage point sensitivities: psi row from S/KS, v and delta rows analytic
        const double tau = tsub + wprev;
        const double j24 = vs * iwb * (1.0 + t * t), j23 = t * iwb;
        double NS[3][5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          const double dps = S[2][q] + wprev * KS[2][q];
          const double dvs = (q == 1) ? 1.0 : ((q == 3) ? tau : 0.0);
          const double dds = (q == 2) ? 1.0 : ((q == 4) ? tau : 0.0);
          NS[0][q] = -vs * s * dps + c * dvs;
          NS[1][q] = vs * c * dps + s * dvs;
          NS[2][q] = j23 * dvs + j24 * dds;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int q = 0; q < 5; ++q) { KS[r][q] = NS[r][q]; AS[r][q] += wsum * NS[r][q]; }
      }
      kp = fp;
      ax += wsum * fx; ay += wsum * fy; ap += wsum * fp;
    }
    x += h6 * ax; y += h6 * ay; psi += h6 * ap;
    v += h * a; de += h * w;
    {
      double se, ce;
      small_sincos(h6 * ap, &se, &ce);
      const double sn = sp_ * ce + cp_ * se, cn = cp_ * ce - sp_ * se;
      sp_ = sn; cp_ = cn; sd0 = sd2; cd0 = cd2;
    }
    if (SENS)
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 5; ++q) S[r][q] += h6 * AS[r][q];
    tsub += h;
  }
  out[0] = x; out[1] = y; out[2] = psi; out[3] = v; out[4] = de;
}

// ------------------------------------------------------------------------------ separation certificates
// A block (stage k, obstacle or neighbour j) is certified along a face normal: "every vertex of
// one polygon lies at least dmin outside face f of the other".  kind 1 = polygon face / body
// vertices, kind 2 = body face / polygon vertices.  The working set of a block is the face and
// the two vertices whose rows are imposed: sel = kind*64 + face*16 + vA*4 + vB, vA < vB.
// kind 3 (spec.vv_rows): the closest features of the two polygons are two vertices, polygon vertex u and body vertex v, each
// in the other's normal cone; no face normal certifies their distance, so the row is the Euclidean distance |W_v - V_u| itself,
// imposed twice (the block keeps its two slots: twice the barrier weight, same optimum): sel = 192 + u*16 + v*4 + v.
constexpr double kHyst = 1e-3;  // m: a block keeps its face until another is better by this much
constexpr double kVvInert = 1.0;  // m: margin of the second slot of a vertex-vertex block in the planning kernels (rows_for)
constexpr int kWsStallDiv = 4;     // iterates that change the working set count 1 / kWsStallDiv towards the stall test
constexpr int kShiftStagMin = 40;  // earliest iteration of a stagnation-triggered curvature shift (KSpec::stag_win)

template <bool GRAD>
CFZ_FN void vertex_dist(const double A[4][2], const double b[4], const double V[4][2], double x, double y,
                        double c, double s, const double g[4], int kind, int f, double d[4], double gr[4][3]) {
  if (kind == 1) {
    const double BV[4][2] = {{g[0], g[1]}, {-g[2], g[1]}, {-g[2], -g[3]}, {g[0], -g[3]}};
    double ax = 0.0, ay = 0.0, bf = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i == f) { ax = A[i][0]; ay = A[i][1]; bf = b[i]; }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double dwx = -s * BV[v][0] - c * BV[v][1], dwy = c * BV[v][0] - s * BV[v][1];
      d[v] = (x + dwy) * ax + (y - dwx) * ay - bf;
      if (GRAD) { gr[v][0] = ax; gr[v][1] = ay; gr[v][2] = ax * dwx + ay * dwy; }
    }
  } else {
    const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
    double gf = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i == f) gf = g[i];
    const double nx = c * gx - s * gy, ny = s * gx + c * gy, dnx = -s * gx - c * gy, dny = c * gx - s * gy;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      d[v] = (V[v][0] - x) * nx + (V[v][1] - y) * ny - gf;
      if (GRAD) { gr[v][0] = -nx; gr[v][1] = -ny; gr[v][2] = dnx * (V[v][0] - x) + dny * (V[v][1] - y); }
    }
  }
}

// max(0, -lambda_min) of the symmetric 3 x 3 [[a00 a01 a02] [a01 a11 a12] [a02 a12 a22]] (trigonometric closed form;
// oracle/mpc_nlp.py pose_shift).  Only reached from iteration spec.shift_after on.
#if defined(__HIPCC__)
static __host__ __device__ __attribute__((noinline))  // cold: kept out of the assembly stage's register allocation
#else
static
#endif
double pose_shift(double a00, double a11, double a22, double a01, double a02, double a12) {
  const double d2 = a00 * a11 - a01 * a01;
  const double d3 = a22 * d2 - (a02 * a02 * a11 - 2.0 * a02 * a12 * a01 + a12 * a12 * a00);
  if (a00 > 0.0 && d2 > 0.0 && d3 >= 0.0) return 0.0;
  const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
  const double qm = (a00 + a11 + a22) * (1.0 / 3.0);
  const double b00 = a00 - qm, b11 = a11 - qm, b22 = a22 - qm;
  const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1) * (1.0 / 6.0));
  const double ip = 1.0 / p;
  const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
  double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
  r = fmin(1.0, fmax(-1.0, r));
  const double lam = qm + 2.0 * p * cos(acos(r) * (1.0 / 3.0) + 2.0943951023931953);
  return fmax(0.0, -lam);
}

CFZ_FN double pick4(const double d[4], int v) {
  // two levels of selects on the bits of v, between four values loaded unconditionally first: written `c ? d[1] : d[0]` the
  // compiler folds the choice into the address (one load at d + (v & 1)), the array has a runtime index and lives in scratch
  const double a = d[0], b = d[1], c = d[2], e = d[3];
  const double lo = (v & 1) ? b : a, hi = (v & 1) ? e : c;
  return (v & 2) ? hi : lo;
}

// All 32 face-vertex distances of a block in one pass: D[f] (f = 0..3) kind 1, polygon face f against the four body
// vertices; D[4 + f] kind 2, body face f against the four polygon vertices.  Same expressions as vertex_dist, with what
// the faces of a kind share (the rotated body vertices; the vertices relative to the pose) formed once.
CFZ_FN void block_dists(const double A[4][2], const double b[4], const double V[4][2], double x, double y, double c,
                        double s, const double g[4], double D[8][4], double px[4], double py[4]) {
  const double BV[4][2] = {{g[0], g[1]}, {-g[2], g[1]}, {-g[2], -g[3]}, {g[0], -g[3]}};
  double rx[4], ry[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const double dwx = -s * BV[v][0] - c * BV[v][1], dwy = c * BV[v][0] - s * BV[v][1];
    px[v] = x + dwy; py[v] = y - dwx;
    rx[v] = V[v][0] - x; ry[v] = V[v][1] - y;
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
    const double nx = c * gx - s * gy, ny = s * gx + c * gy;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      D[f][v] = px[v] * A[f][0] + py[v] * A[f][1] - b[f];
      D[4 + f][v] = rx[v] * nx + ry[v] * ny - g[f];
    }
  }
}

// Working set of a block from its 32 distances (the rule of oracle/mpc_nlp.py select_rows): the face with the largest
// minimum over its four vertices (first such face; the previous face is kept while it is within kHyst of the best), on it
// the nearest vertex and its nearer neighbour (the previous pair is kept while it still holds the nearest vertex and is
// within kHyst).  dsel: the four distances of the chosen face.
// vv: also look for a vertex-vertex pair (kind 3): polygon vertex u lies outside exactly the two body faces that meet at
// body vertex v (signs of the kind-2 distances), and W_v lies beyond both polygon edges that leave V_u; such a pair is THE
// closest pair of the two convex polygons, and it replaces the face rows when its distance exceeds theirs.
CFZ_FN int select_from(const double D[8][4], int prev, double dsel[4], int vv, const double V[4][2], const double px[4],
                       const double py[4], double vv_enter = 0.0) {
  const int pidx = ((prev >> 6) - 1) * 4 + ((prev >> 4) & 3);  // face index of the previous working set
  const bool hp = prev != 0 && (prev >> 6) != 3;               // there is a previous face to prefer
  double best = 0.0, prev_val = 0.0;
  int bi = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double val = fmin(fmin(D[i][0], D[i][1]), fmin(D[i][2], D[i][3]));
    if (hp && i == pidx) prev_val = val;
    if (i == 0 || val > best) { best = val; bi = i; }
  }
  if (hp && prev_val >= best - kHyst) bi = pidx;
  // the chosen face's row of D: binary select over the three bits of bi
  double d[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const double e0 = D[0][v], e1 = D[1][v], e2 = D[2][v], e3 = D[3][v], e4 = D[4][v], e5 = D[5][v], e6 = D[6][v], e7 = D[7][v];
    const double q0 = (bi & 1) ? e1 : e0, q1 = (bi & 1) ? e3 : e2, q2 = (bi & 1) ? e5 : e4, q3 = (bi & 1) ? e7 : e6;
    const double h0 = (bi & 2) ? q1 : q0, h1 = (bi & 2) ? q3 : q2;
    d[v] = (bi & 4) ? h1 : h0;
    dsel[v] = d[v];
  }
  int v0 = 0;
#pragma unroll
  for (int v = 1; v < 4; ++v) if (d[v] < pick4(d, v0)) v0 = v;
  const int n1 = (v0 + 1) & 3, n2 = (v0 + 3) & 3;
  const double d0 = pick4(d, v0), dn1 = pick4(d, n1), dn2 = pick4(d, n2);
  int v1 = (dn1 < dn2) ? n1 : ((dn2 < dn1) ? n2 : (n1 < n2 ? n1 : n2));
  if (hp && bi == pidx) {
    const int oa = (prev >> 2) & 3, ob = prev & 3;
    const double da = pick4(d, oa), db = pick4(d, ob);
    if (fmin(da, db) <= d0 + 1e-12 && fmax(da, db) <= pick4(d, v1) + kHyst) { v0 = oa; v1 = ob; }
  }
  const int va = v0 < v1 ? v0 : v1, vb = v0 < v1 ? v1 : v0;
  int code = ((bi >> 2) + 1) * 64 + (bi & 3) * 16 + va * 4 + vb;
  if (vv) {
    const double dn = fmin(pick4(d, v0), pick4(d, v1));  // the separation this face certifies (a kept pair is not ordered by distance)
    int pc = 0;
    double r2 = 0.0;
#pragma unroll
    for (int u = 3; u >= 0; --u) {  // the lowest qualifying u wins (they all have the same distance)
      const bool ox = D[4][u] >= 0.0, oy = D[5][u] >= 0.0;
      const bool cand = (ox || D[6][u] >= 0.0) && (oy || D[7][u] >= 0.0);
      const int v = ox ? (oy ? 0 : 3) : (oy ? 1 : 2);
      const double wx = pick4(px, v) - V[u][0], wy = pick4(py, v) - V[u][1];
      const int u1 = (u + 1) & 3, u3 = (u + 3) & 3;
      const bool in = wx * (V[u1][0] - V[u][0]) + wy * (V[u1][1] - V[u][1]) <= 0.0 &&
                      wx * (V[u3][0] - V[u][0]) + wy * (V[u3][1] - V[u][1]) <= 0.0;
      if (cand && in) { pc = 192 + u * 16 + v * 5; r2 = wx * wx + wy * wy; }
    }
    if (pc != 0 && dn > 0.0) {
      const double r = sqrt(r2);
      // vv_enter (the joint plan: 1e-4 while mu >= 1e-4): a FACE block turns into a vertex-vertex block only when the pair's
      // distance exceeds the face's separation by that much -- near a transition of the closest features the two certificates
      // differ by micrometres while their curvatures differ by multiplier / distance, and a block that flipped at every iterate
      // took a joint plan from 55 to 157 iterations; the margin is dropped for the last barrier problems, so the limit is exact
      const double enter = ((prev >> 6) != 3 && prev != 0 && vv_enter > 1e-9) ? vv_enter : 1e-9;
      if (r > dn + enter) { code = pc; dsel[0] = r; dsel[1] = r; dsel[2] = r; dsel[3] = r; }
    }
  }
  return code;
}

CFZ_CALL int select_rows(const double A[4][2], const double b[4], const double V[4][2], double x, double y, double c,
                       double s, const double g[4], int prev, int vv = 0, double vv_enter = 0.0) {  // the planning kernels (values come from rows_for)
  double D[8][4], dsel[4], px[4], py[4];
  block_dists(A, b, V, x, y, c, s, g, D, px, py);
  return select_from(D, prev, dsel, vv, V, px, py, vv_enter);
}

// working set AND the values of its two rows in one pass (the rows are two of the distances the selection looked at)
CFZ_CALL int select_rows_sep(const double A[4][2], const double b[4], const double V[4][2], double x, double y, double c,
                           double s, const double g[4], int prev, double sep[2], int vv = 0) {
  double D[8][4], dsel[4], px[4], py[4];
  block_dists(A, b, V, x, y, c, s, g, D, px, py);
  const int sel = select_from(D, prev, dsel, vv, V, px, py);
  sep[0] = pick4(dsel, (sel >> 2) & 3); sep[1] = pick4(dsel, sel & 3);
  return sel;
}

// values (and gradients wrt x,y,psi) of the two rows of working set `sel`
// (the MPC solver itself imposes a vertex-vertex row twice, block_sep / block_grad below; the planning kernels, through rows_for,
// keep the second slot inert: value r + kVvInert, same gradient)
template <bool GRAD>
CFZ_CALL void rows_for(const double A[4][2], const double b[4], const double V[4][2], double x, double y, double c,
                     double s, const double g[4], int sel, double sep[2], double grad[2][3]) {
  if ((sel >> 6) == 3) {  // vertex-vertex: polygon vertex u, body vertex v, the row is their distance (twice)
    const int u = (sel >> 4) & 3, v = sel & 3;
    double ux = V[0][0], uy = V[0][1];
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i == u) { ux = V[i][0]; uy = V[i][1]; }
    const double bx = (v == 0 || v == 3) ? g[0] : -g[2], by = (v < 2) ? g[1] : -g[3];
    const double dwx = -s * bx - c * by, dwy = c * bx - s * by;  // d(R b_v)/dpsi; R b_v = (dwy, -dwx)
    const double wx = x + dwy - ux, wy = y - dwx - uy;
    const double r = sqrt(wx * wx + wy * wy), ir = 1.0 / r;
    sep[0] = r; sep[1] = r + kVvInert;  // the block's second slot restates the row with a margin: always inactive
    if (GRAD) {
      const double n0 = wx * ir, n1 = wy * ir, n2 = n0 * dwx + n1 * dwy;
      grad[0][0] = n0; grad[0][1] = n1; grad[0][2] = n2; grad[1][0] = n0; grad[1][1] = n1; grad[1][2] = n2;
    }
    return;
  }
  double d[4], gr[4][3];
  vertex_dist<GRAD>(A, b, V, x, y, c, s, g, sel >> 6, (sel >> 4) & 3, d, gr);
  const int va = (sel >> 2) & 3, vb = sel & 3;
  sep[0] = pick4(d, va); sep[1] = pick4(d, vb);
  if (GRAD) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (v == va) { grad[0][0] = gr[v][0]; grad[0][1] = gr[v][1]; grad[0][2] = gr[v][2]; }
      if (v == vb) { grad[1][0] = gr[v][0]; grad[1][1] = gr[v][1]; grad[1][2] = gr[v][2]; }
    }
  }
}

// polygon of block j at stage k: static obstacle from the spec, neighbour from its pose
CFZ_FN void block_polygon(const KSpec &sp, const double *m, const Lay &L, int k, int j, double A[4][2], double b[4],
                          double V[4][2]) {
  if (j < sp.n_obs) {
    const double *o = sp.obs_tab + j * 20;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      A[i][0] = o[2 * i]; A[i][1] = o[2 * i + 1]; b[i] = o[8 + i];
      V[i][0] = o[12 + 2 * i]; V[i][1] = o[12 + 2 * i + 1];
    }
  } else {
    const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
    const double xo = q[0], yo = q[1], co = q[2], so = q[3];
    const double g0 = sp.g[0], g1 = sp.g[1], g2 = sp.g[2], g3 = sp.g[3];
    A[0][0] = co; A[0][1] = so; A[1][0] = -so; A[1][1] = co;
    A[2][0] = -co; A[2][1] = -so; A[3][0] = so; A[3][1] = -co;
    b[0] = A[0][0] * xo + A[0][1] * yo + g0; b[1] = A[1][0] * xo + A[1][1] * yo + g1;
    b[2] = A[2][0] * xo + A[2][1] * yo + g2; b[3] = A[3][0] * xo + A[3][1] * yo + g3;
    const double BV[4][2] = {{g0, g1}, {-g2, g1}, {-g2, -g3}, {g0, -g3}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      V[i][0] = xo + co * BV[i][0] - so * BV[i][1];
      V[i][1] = yo + so * BV[i][0] + co * BV[i][1];
    }
  }
}

// vertex v of the polygon of block j at stage k
CFZ_FN void block_vertex(const KSpec &sp, const double *m, const Lay &L, int k, int j, int v, double &vx, double &vy) {
  if (j < sp.n_obs) {
    const double *o = sp.obs_tab + j * 20;
    vx = o[12 + 2 * v]; vy = o[13 + 2 * v];
  } else {
    const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
    const double xo = q[0], yo = q[1], co = q[2], so = q[3];
    const double bx = (v == 0 || v == 3) ? sp.g[0] : -sp.g[2], by = (v < 2) ? sp.g[1] : -sp.g[3];
    vx = xo + co * bx - so * by; vy = yo + so * bx + co * by;
  }
}

// A vertex-vertex row (kind 3, sl = 192 + u*16 + v*5): w = t + R b_v - V_u, r = |w|, n = w / r, dw = d(R b_v)/dpsi.
CFZ_FN void vv_row(const KSpec &sp, const double *m, const Lay &L, int k, int j, int sl, double x, double y, double c, double s,
                   double &r, double &n0, double &n1, double &dwx, double &dwy) {
  const int u = (sl >> 4) & 3, v = sl & 3;
  double vx, vy;
  block_vertex(sp, m, L, k, j, u, vx, vy);
  const double bx = (v == 0 || v == 3) ? sp.g[0] : -sp.g[2], by = (v < 2) ? sp.g[1] : -sp.g[3];
  dwx = -s * bx - c * by; dwy = c * bx - s * by;
  const double wx = x + dwy - vx, wy = y - dwx - vy;
  r = sqrt(wx * wx + wy * wy);
  const double ir = 1.0 / r;
  n0 = wx * ir; n1 = wy * ir;
}

// Gradients of the two rows of block j at stage k, rebuilt from the working-set code instead of being kept in LDS
// (same expressions as vertex_dist<true>): both rows share d/dx = a0, d/dy = a1 (same face); ap[r] = d/dpsi of row r.
CFZ_FN void block_grad(const KSpec &sp, const double *m, const Lay &L, int k, int j, int sl, double x, double y, double c,
                       double s, double &a0, double &a1, double ap[2]) {
  const int f = (sl >> 4) & 3;
  const double g0 = sp.g[0], g1 = sp.g[1], g2 = sp.g[2], g3 = sp.g[3];
  if ((sl >> 6) == 3) {
    double r, dwx, dwy;
    vv_row(sp, m, L, k, j, sl, x, y, c, s, r, a0, a1, dwx, dwy);
    ap[0] = a0 * dwx + a1 * dwy; ap[1] = ap[0];
  } else if ((sl >> 6) == 1) {
    double ax, ay;
    if (j < sp.n_obs) {
      const double *o = sp.obs_tab + j * 20;
      ax = o[2 * f]; ay = o[2 * f + 1];
    } else {
      const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
      const double co = q[2], so = q[3];
      ax = f == 0 ? co : (f == 1 ? -so : (f == 2 ? -co : so));
      ay = f == 0 ? so : (f == 1 ? co : (f == 2 ? -so : -co));
    }
    a0 = ax; a1 = ay;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int v = r == 0 ? ((sl >> 2) & 3) : (sl & 3);
      const double bx = (v == 0 || v == 3) ? g0 : -g2, by = (v < 2) ? g1 : -g3;
      const double dwx = -s * bx - c * by, dwy = c * bx - s * by;
      ap[r] = ax * dwx + ay * dwy;
    }
  } else {
    const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
    const double nx = c * gx - s * gy, ny = s * gx + c * gy, dnx = -s * gx - c * gy, dny = c * gx - s * gy;
    a0 = -nx; a1 = -ny;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int v = r == 0 ? ((sl >> 2) & 3) : (sl & 3);
      double vx, vy;
      if (j < sp.n_obs) {
        const double *o = sp.obs_tab + j * 20;
        vx = o[12 + 2 * v]; vy = o[13 + 2 * v];
      } else {
        const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
        const double xo = q[0], yo = q[1], co = q[2], so = q[3];
        const double bx = (v == 0 || v == 3) ? g0 : -g2, by = (v < 2) ? g1 : -g3;
        vx = xo + co * bx - so * by; vy = yo + so * bx + co * by;
      }
      ap[r] = dnx * (vx - x) + dny * (vy - y);
    }
  }
}

// Values of the two rows of block j at stage k for a GIVEN working-set code (the line search holds the working set
// fixed): the two distances only, from the code, like block_grad -- no polygon is built, no other distance formed.
CFZ_FN void block_sep(const KSpec &sp, const double *m, const Lay &L, int k, int j, int sl, double x, double y, double c,
                      double s, double sep[2]) {
  const int f = (sl >> 4) & 3;
  const double g0 = sp.g[0], g1 = sp.g[1], g2 = sp.g[2], g3 = sp.g[3];
  const double gf = f == 0 ? g0 : (f == 1 ? g1 : (f == 2 ? g2 : g3));
  if ((sl >> 6) == 3) {
    double n0, n1, dwx, dwy;
    vv_row(sp, m, L, k, j, sl, x, y, c, s, sep[0], n0, n1, dwx, dwy);
    sep[1] = sep[0];
  } else if ((sl >> 6) == 1) {
    double ax, ay, bf;
    if (j < sp.n_obs) {
      const double *o = sp.obs_tab + j * 20;
      ax = o[2 * f]; ay = o[2 * f + 1]; bf = o[8 + f];
    } else {
      const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
      const double xo = q[0], yo = q[1], co = q[2], so = q[3];
      ax = f == 0 ? co : (f == 1 ? -so : (f == 2 ? -co : so));
      ay = f == 0 ? so : (f == 1 ? co : (f == 2 ? -so : -co));
      bf = ax * xo + ay * yo + gf;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int v = r == 0 ? ((sl >> 2) & 3) : (sl & 3);
      const double bx = (v == 0 || v == 3) ? g0 : -g2, by = (v < 2) ? g1 : -g3;
      const double dwx = -s * bx - c * by, dwy = c * bx - s * by;
      sep[r] = (x + dwy) * ax + (y - dwx) * ay - bf;
    }
  } else {
    const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
    const double nx = c * gx - s * gy, ny = s * gx + c * gy;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int v = r == 0 ? ((sl >> 2) & 3) : (sl & 3);
      double vx, vy;
      if (j < sp.n_obs) {
        const double *o = sp.obs_tab + j * 20;
        vx = o[12 + 2 * v]; vy = o[13 + 2 * v];
      } else {
        const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
        const double xo = q[0], yo = q[1], co = q[2], so = q[3];
        const double bx = (v == 0 || v == 3) ? g0 : -g2, by = (v < 2) ? g1 : -g3;
        vx = xo + co * bx - so * by; vy = yo + so * bx + co * by;
      }
      sep[r] = (vx - x) * nx + (vy - y) * ny - gf;
    }
  }
}

// ------------------------------------------------------------------------------ objective pieces
CFZ_FN double stage_cost(const KSpec &sp, const double *ref, int k, const double p[kNP]) {
  const double *w = sp.weights; const int N = sp.N;
  const double ex = p[0] - ref[k], ey = p[1] - ref[N + k], ep = p[2] - ref[2 * N + k];
  return w[0] * ex * ex + w[1] * ey * ey + w[2] * ep * ep + w[3] * p[5] * p[5] + w[4] * p[3] * p[3] * p[6] * p[6] +
         w[5] * p[4] * p[4];
}
CFZ_FN void stage_grad(const KSpec &sp, const KDer &dv, const double *ref, int k, const double p[kNP], double gr[kNP]) {
  const double *w2 = dv.w2; const int N = sp.N;
  gr[0] = w2[0] * (p[0] - ref[k]);
  gr[1] = w2[1] * (p[1] - ref[N + k]);
  gr[2] = w2[2] * (p[2] - ref[2 * N + k]);
  gr[3] = w2[4] * p[3] * p[6] * p[6];
  gr[4] = w2[5] * p[4];
  gr[5] = w2[3] * p[5];
  gr[6] = w2[4] * p[3] * p[3] * p[6];
}

// dense views of the compact stage storage
CFZ_FN void load_AB(const double *m, const Lay &L, int k, double dt, double A[5][5], double B[5][2]) {
  const double *s = m + L.ab + k * 15;
  for (int i = 0; i < 5; ++i) {
    for (int q = 0; q < 5; ++q) A[i][q] = (i == q) ? 1.0 : 0.0;
    B[i][0] = 0.0; B[i][1] = 0.0;
  }
  for (int r = 0; r < 3; ++r) {
    A[r][2] = s[r * 5 + 0]; A[r][3] = s[r * 5 + 1]; A[r][4] = s[r * 5 + 2];
    B[r][0] = s[r * 5 + 3]; B[r][1] = s[r * 5 + 4];
  }
  B[3][0] = dt; B[4][1] = dt;
}
CFZ_FN void load_H(const double *m, const Lay &L, int k, double H[kNP][kNP]) {
  const double *h = m + L.hc + k * 11;
  for (int i = 0; i < kNP; ++i)
    for (int q = 0; q < kNP; ++q) H[i][q] = 0.0;
  for (int i = 0; i < kNP; ++i) H[i][i] = h[i];
  H[0][1] = H[1][0] = h[7]; H[0][2] = H[2][0] = h[8]; H[1][2] = H[2][1] = h[9]; H[3][6] = H[6][3] = h[10];
}
CFZ_FN void sym2_solve6(const double M[2][2], const double rhs[2][6], double out[2][6]) {
  const double l00 = sqrt(M[0][0]), l10 = M[1][0] / l00, l11 = sqrt(M[1][1] - l10 * l10);
  for (int q = 0; q < 6; ++q) {
    const double y0 = rhs[0][q] / l00, y1 = (rhs[1][q] - l10 * y0) / l11;
    const double x1 = y1 / l11;
    out[0][q] = (y0 - l10 * x1) / l00; out[1][q] = x1;
  }
}

// RK4 with the sensitivities of (x, y, psi) for TWO columns of (psi0, v0, delta0, a, w): column qa (0..3, chosen by the
// lane) and column 4.  The four lanes of a stage's quad run this in lockstep with qa = 0, 1, 2, 3 and so cover all five
// columns; the nominal trajectory is the same in all of them.  Same recurrences as rk4_step<true>, column by column.
CFZ_CALL void rk4_sens2(const double z[5], double a, double w, double h, double hh, double h6, double iwb, int M, int qa,
                        double out[5], double Sa[3], double Sb[3]) {
  double x = z[0], y = z[1], psi = z[2], v = z[3], de = z[4];
  Sa[0] = 0.0; Sa[1] = 0.0; Sa[2] = (qa == 0) ? 1.0 : 0.0;
  Sb[0] = 0.0; Sb[1] = 0.0; Sb[2] = 0.0;
  // d(v at a stage point)/d(column), d(delta at a stage point)/d(column): 1 for the state's own column, the elapsed
  // time for its input's column
  const double va1 = (qa == 1) ? 1.0 : 0.0, vat = (qa == 3) ? 1.0 : 0.0;  // dvs = va1 + vat * tau
  const double da1 = (qa == 2) ? 1.0 : 0.0;                                 // dds = da1        (column 4: dds = tau)
  double sp_, cp_, sd0, cd0, sh, ch;
  sincos(psi, &sp_, &cp_);
  sincos(de, &sd0, &cd0);
  small_sincos(hh * w, &sh, &ch);
  double tsub = 0.0;
  for (int m_ = 0; m_ < M; ++m_) {
    double ax = 0, ay = 0, ap = 0, kp = 0;
    double ASa[3] = {0.0, 0.0, 0.0}, ASb[3] = {0.0, 0.0, 0.0}, KSa2 = 0.0, KSb2 = 0.0;
    const double sd1 = sd0 * ch + cd0 * sh, cd1 = cd0 * ch - sd0 * sh;
    const double sd2 = sd1 * ch + cd1 * sh, cd2 = cd1 * ch - sd1 * sh;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const double wprev = (st == 0) ? 0.0 : ((st == 3) ? h : hh);
      const double wsum = (st == 0 || st == 3) ? 1.0 : 2.0;
      const double vs = v + wprev * a;
      double s = sp_, c = cp_;
      if (st > 0) {
        double se, ce;
        small_sincos(wprev * kp, &se, &ce);
        s = sp_ * ce + cp_ * se; c = cp_ * ce - sp_ * se;
      }
      const double sd = (st == 0) ? sd0 : ((st == 3) ? sd2 : sd1), cd = (st == 0) ? cd0 : ((st == 3) ? cd2 : cd1);
      const double t = sd / cd;
      const double fx = vs * c, fy = vs * s, fp = vs * iwb * t;
      const double tau = tsub + wprev;
      const double j24 = vs * iwb * (1.0 + t * t), j23 = t * iwb;
      {  // column qa
        const double dps = Sa[2] + wprev * KSa2, dvs = va1 + vat * tau, dds = da1;
        const double n0 = -vs * s * dps + c * dvs, n1 = vs * c * dps + s * dvs, n2 = j23 * dvs + j24 * dds;
        KSa2 = n2; ASa[0] += wsum * n0; ASa[1] += wsum * n1; ASa[2] += wsum * n2;
      }
      {  // column 4 (w): dvs = 0, dds = tau
        const double dps = Sb[2] + wprev * KSb2;
        const double n0 = -vs * s * dps, n1 = vs * c * dps, n2 = j24 * tau;
        KSb2 = n2; ASb[0] += wsum * n0; ASb[1] += wsum * n1; ASb[2] += wsum * n2;
      }
      kp = fp;
      ax += wsum * fx; ay += wsum * fy; ap += wsum * fp;
    }
    x += h6 * ax; y += h6 * ay; psi += h6 * ap;
    v += h * a; de += h * w;
    {
      double se, ce;
      small_sincos(h6 * ap, &se, &ce);
      const double sn = sp_ * ce + cp_ * se, cn = cp_ * ce - sp_ * se;
      sp_ = sn; cp_ = cn; sd0 = sd2; cd0 = cd2;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { Sa[r] += h6 * ASa[r]; Sb[r] += h6 * ASb[r]; }
    tsub += h;
  }
  out[0] = x; out[1] = y; out[2] = psi; out[3] = v; out[4] = de;
}

// ------------------------------------------------------------------------------ trial-point evaluation
// theta = |c|_1 and barrier objective at (p + alpha dp, sg + alpha dsg).  Lane partials: part 0 theta, 1 phi without the
// log terms, 2 sum of logs, 3 = 1 if a bound or a slack is not strictly inside.
// sens (uniform): the trial is expected to be accepted (the first trial of an iteration whose predecessor's first trial was), so the
// stage's quad integrates the dynamics WITH the sensitivities, as the rows phase of the next iteration would at this very point, and
// leaves what that phase needs -- the sensitivities in L.ab, the defects in the (dead) Hessian slots L.hc + 5 k, the heading's cosine and
// sine in L.cs -- so that it can skip its own integration if the trial is accepted (round 6: 14 k of an iteration's 180 k cycles).  The
// nominal trajectory of rk4_sens2 is rk4_step_h<false>'s, statement for statement.
CFZ_CALL void merit_partials(const KSpec &sp, const KDer &dv, const double *refg, double *m, const Lay &L, double alpha, int tid, int sens,
                             double &th_o, double &ph_o, double &ll_o, double &bad_o) {
  const int N = sp.N, nb = L.nb;
  const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
  // sum of logs taken as the log of a per-lane product (<= 18 factors in [1e-10, 1e2]: no over/underflow)
  double th = 0.0, ph = 0.0, lprod = 1.0, bad = 0.0;
  if (k < N) {
    double pt[kNP];
    for (int i = 0; i < kNP; ++i) pt[i] = m[L.p + k * kNP + i] + alpha * m[L.dp + k * kNP + i];
    double sn, cn;
    sincos(pt[2], &sn, &cn);
    for (int j = sub; j < nb; j += kLPS) {
      const int t = k * nb + j;
      double sep[2];
      block_sep(sp, m, L, k, j, sel_ptr(m, L)[t], pt[0], pt[1], cn, sn, sep);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double sg = m[L.sg + 2 * t + r] + alpha * m[L.dsg + 2 * t + r];
        if (k > 0) th += fabs(sep[r] - sp.dmin - sg);  // (the rows of stage 0 are constants: no part of the violation)
        if (!(sg > 0.0)) bad = 1.0; else lprod *= sg;
      }
    }
    double F[5];
    if (sens) {
      if (sub == 0) { m[L.cs + 2 * k] = cn; m[L.cs + 2 * k + 1] = sn; }
      if (k + 1 < N) {
        double Sa[3], Sb[3];
        rk4_sens2(pt, pt[5], pt[6], dv.rk_h, dv.rk_hh, dv.rk_h6, dv.iwb, sp.rk_substeps, sub, F, Sa, Sb);
        if (sub < 4) for (int r = 0; r < 3; ++r) m[L.ab + k * 15 + r * 5 + sub] = Sa[r];
        if (sub == 0) for (int r = 0; r < 3; ++r) m[L.ab + k * 15 + r * 5 + 4] = Sb[r];
      }
    }
    if (sub == 0) {
      for (int q = 0; q < 6; ++q) {
        const double dl = pt[bcol(q)] - sp.bounds[2 * q], du = sp.bounds[2 * q + 1] - pt[bcol(q)];
        if (!(dl > 0.0) || !(du > 0.0)) bad = 1.0; else lprod *= dl * du;
      }
      ph += stage_cost(sp, refg, k, pt);
      if (k == 0) for (int i = 0; i < 5; ++i) th += fabs(pt[i] - m[L.x0 + i]);
      if (k + 1 < N) {
        if (!sens) rk4_step_h<false>(pt, pt[5], pt[6], dv.rk_h, dv.rk_hh, dv.rk_h6, dv.iwb, sp.rk_substeps, F, nullptr);
        for (int i = 0; i < 5; ++i) {
          const double d = F[i] - (m[L.p + (k + 1) * kNP + i] + alpha * m[L.dp + (k + 1) * kNP + i]);
          th += fabs(d);
          if (sens) m[L.hc + k * 5 + i] = d;  // (the defect at the trial point: the next iteration's, if the trial is accepted)
        }
      }
    }
  }
  th_o = th; ph_o = ph; ll_o = (bad == 0.0) ? log(lprod) : 0.0; bad_o = bad;
}

// ------------------------------------------------------------------------------ Riccati sweeps (lane 0, out of line)
// Functions of their own so that their registers are their own (inlined into the solver the backward sweep kept ~100
// doubles live on top of the solver's state and spilled into AGPRs and scratch).  Device: they must not NAME any LDS
// (toolchain, see cfz_band.inl); the workspace comes in as an address-space-3 pointer, so every access is a DS
// instruction.  Every lane calls them (uniform control flow), lane 0 does the work.
// Structure used: A_k = I + [0 0 s00 s01 s02; 0 0 s10 s11 s12; 0 0 0 s21 s22; 0; 0] (s20 = 1),
// B_k = [s03 s04; s13 s14; s23 s24; dt 0; 0 dt]; H_k = diag(h0..h6) + pose off-diagonals h7 (0,1), h8 (0,2), h9 (1,2) +
// the v-w cross term h10 (3,6).  A lane-parallel variant (matrix entries spread over lanes, exchange through LDS) measured
// 2.2x slower: every exchange is a dependent LDS round trip.  Round 4 built the register-only variant (five lanes, lane j
// owning column j of P; W, e, three rows of M and the gains' columns broadcast by v_readlane, 40 doubles per stage, no LDS, no
// barrier): bit-compatible results, 70 k cycles per sweep against 60 k here -- the 80 v_readlane of a stage cost more than the 150
// multiply-adds they save a lane (docs/notebook.md); taken out again.
#if defined(__HIP_DEVICE_COMPILE__) && defined(CFZ_SWEEP_INLINE)
#define CFZ_SWEEP __device__ __forceinline__ void
#define CFZ_SWEEP_GUARD if (threadIdx.x != 0) return;
#elif defined(__HIP_DEVICE_COMPILE__)
#define CFZ_SWEEP __device__ __attribute__((noinline)) void
#define CFZ_SWEEP_GUARD if (threadIdx.x != 0) return;
#else
#define CFZ_SWEEP static void
#define CFZ_SWEEP_GUARD
#endif

// The sweep is software-pipelined by hand: the loads of the NEXT stage are issued before the stores of this one.
// The compiler cannot do that itself (loads and stores go through the same workspace pointer, so it must assume that a
// store may feed a later load) and without it every stage waits for an LDS round trip per operand group.

// backward sweep: gains K_k (12 per stage) into kk, value function of stage 0 into rP (25) and rP + 25 (5).
// P is kept as its upper triangle.  With W = B'P:  Hux = W A,  Huu = R + W B,  hu = g_u + B'(p + P d);  M = P A,
// Hxx = Q + A'M (upper triangle only);  K = -Huu^-1 [Hux hu];  P <- Hxx - Hux' Huu^-1 Hux,  p <- hx - Hux' Huu^-1 hu.
CFZ_SWEEP riccati_backward(wsp_f64 *m, int N, double dt, int o_ab, int o_hc, int o_gk, int o_d, int o_kk, int o_rP) {
  CFZ_SWEEP_GUARD
  // upper triangle of P: P00 P01 P02 P03 P04 | P11 P12 P13 P14 | P22 P23 P24 | P33 P34 | P44
  double P00, P01, P02, P03 = 0.0, P04 = 0.0, P11, P12, P13 = 0.0, P14 = 0.0, P22, P23 = 0.0, P24 = 0.0, P33, P34 = 0.0, P44;
  double p0, p1, p2, p3, p4;
  {
    const int k = N - 1;  // terminal stage: its inputs a,w are costed but drive no dynamics
    const wsp_f64 *h = m + o_hc + k * 11, *gk = m + o_gk + k * kNP;
    wsp_f64 *K = m + o_kk + k * 12;
    const double h10 = h[10], h6 = h[6];
    const double k53 = -h10 / h6, k10 = -gk[5] / h[5], k11 = -gk[6] / h6;
    P00 = h[0]; P11 = h[1]; P22 = h[2]; P33 = h[3] + h10 * k53; P44 = h[4];
    P01 = h[7]; P02 = h[8]; P12 = h[9];
    p0 = gk[0]; p1 = gk[1]; p2 = gk[2]; p3 = gk[3] + h10 * k11; p4 = gk[4];
    for (int q = 0; q < 10; ++q) K[q] = 0.0;
    K[5 + 3] = k53; K[10] = k10; K[11] = k11;
  }
  // dynamics data (s, d: needed first) of the stage being processed are loaded one stage ahead; its cost data (h, g: needed
  // later in the stage) at the top of the stage, where their latency hides behind the first hundred multiply-adds
  double sc[15], dc[5];
  if (N >= 2) {
    const int k = N - 2;
    const wsp_f64 *s = m + o_ab + k * 15, *d = m + o_d + k * 5;
#pragma unroll
    for (int i = 0; i < 15; ++i) sc[i] = s[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) dc[i] = d[i];
  }
  for (int k = N - 2; k >= 0; --k) {
    double hc_[11], gc[7];
    {
      const wsp_f64 *h = m + o_hc + k * 11, *g = m + o_gk + k * kNP;
#pragma unroll
      for (int i = 0; i < 11; ++i) hc_[i] = h[i];
#pragma unroll
      for (int i = 0; i < 7; ++i) gc[i] = g[i];
    }
    const double s00 = sc[0], s01 = sc[1], s02 = sc[2], s03 = sc[3], s04 = sc[4];
    const double s10 = sc[5], s11 = sc[6], s12 = sc[7], s13 = sc[8], s14 = sc[9];
    const double s21 = sc[11], s22 = sc[12], s23 = sc[13], s24 = sc[14];
    // W = B'P (2 x 5); rows of B': (s03 s13 s23 dt 0), (s04 s14 s24 0 dt)
    const double W00 = s03 * P00 + s13 * P01 + s23 * P02 + dt * P03;
    const double W01 = s03 * P01 + s13 * P11 + s23 * P12 + dt * P13;
    const double W02 = s03 * P02 + s13 * P12 + s23 * P22 + dt * P23;
    const double W03 = s03 * P03 + s13 * P13 + s23 * P23 + dt * P33;
    const double W04 = s03 * P04 + s13 * P14 + s23 * P24 + dt * P34;
    const double W10 = s04 * P00 + s14 * P01 + s24 * P02 + dt * P04;
    const double W11 = s04 * P01 + s14 * P11 + s24 * P12 + dt * P14;
    const double W12 = s04 * P02 + s14 * P12 + s24 * P22 + dt * P24;
    const double W13 = s04 * P03 + s14 * P13 + s24 * P23 + dt * P34;
    const double W14 = s04 * P04 + s14 * P14 + s24 * P24 + dt * P44;
    // Hux = W A (+ the v-w cross term of the stage cost)
    const double X00 = W00, X01 = W01, X02 = W02 + s00 * W00 + s10 * W01;
    const double X03 = W03 + s01 * W00 + s11 * W01 + s21 * W02, X04 = W04 + s02 * W00 + s12 * W01 + s22 * W02;
    const double X10 = W10, X11 = W11, X12 = W12 + s00 * W10 + s10 * W11;
    const double X13 = W13 + s01 * W10 + s11 * W11 + s21 * W12 + hc_[10], X14 = W14 + s02 * W10 + s12 * W11 + s22 * W12;
    // Huu = R + W B
    const double a00 = hc_[5] + s03 * W00 + s13 * W01 + s23 * W02 + dt * W03;
    const double a01 = s04 * W00 + s14 * W01 + s24 * W02 + dt * W04;
    const double a11 = hc_[6] + s04 * W10 + s14 * W11 + s24 * W12 + dt * W14;
    // Pd = p + P d
    const double d0 = dc[0], d1 = dc[1], d2 = dc[2], d3 = dc[3], d4 = dc[4];
    const double e0 = p0 + P00 * d0 + P01 * d1 + P02 * d2 + P03 * d3 + P04 * d4;
    const double e1 = p1 + P01 * d0 + P11 * d1 + P12 * d2 + P13 * d3 + P14 * d4;
    const double e2 = p2 + P02 * d0 + P12 * d1 + P22 * d2 + P23 * d3 + P24 * d4;
    const double e3 = p3 + P03 * d0 + P13 * d1 + P23 * d2 + P33 * d3 + P34 * d4;
    const double e4 = p4 + P04 * d0 + P14 * d1 + P24 * d2 + P34 * d3 + P44 * d4;
    const double hu0 = gc[5] + s03 * e0 + s13 * e1 + s23 * e2 + dt * e3;
    const double hu1 = gc[6] + s04 * e0 + s14 * e1 + s24 * e2 + dt * e4;
    const double hx0 = gc[0] + e0, hx1 = gc[1] + e1;
    const double hx2 = gc[2] + e2 + s00 * e0 + s10 * e1;
    const double hx3 = gc[3] + e3 + s01 * e0 + s11 * e1 + s21 * e2;
    const double hx4 = gc[4] + e4 + s02 * e0 + s12 * e1 + s22 * e2;
    // M = P A: columns 0, 1 are P's; columns 2..4 of every row
    const double M02 = P02 + s00 * P00 + s10 * P01, M03 = P03 + s01 * P00 + s11 * P01 + s21 * P02, M04 = P04 + s02 * P00 + s12 * P01 + s22 * P02;
    const double M12 = P12 + s00 * P01 + s10 * P11, M13 = P13 + s01 * P01 + s11 * P11 + s21 * P12, M14 = P14 + s02 * P01 + s12 * P11 + s22 * P12;
    const double M22 = P22 + s00 * P02 + s10 * P12, M23 = P23 + s01 * P02 + s11 * P12 + s21 * P22, M24 = P24 + s02 * P02 + s12 * P12 + s22 * P22;
    const double M33 = P33 + s01 * P03 + s11 * P13 + s21 * P23, M34 = P34 + s02 * P03 + s12 * P13 + s22 * P23;
    const double M43 = P34 + s01 * P04 + s11 * P14 + s21 * P24, M44 = P44 + s02 * P04 + s12 * P14 + s22 * P24;
    // Hxx = Q + A'M, upper triangle
    const double H00 = P00 + hc_[0], H01 = P01 + hc_[7], H02 = M02 + hc_[8], H03 = M03, H04 = M04;
    const double H11 = P11 + hc_[1], H12 = M12 + hc_[9], H13 = M13, H14 = M14;
    const double H22 = M22 + s00 * M02 + s10 * M12 + hc_[2], H23 = M23 + s00 * M03 + s10 * M13, H24 = M24 + s00 * M04 + s10 * M14;
    const double H33 = M33 + s01 * M03 + s11 * M13 + s21 * M23 + hc_[3], H34 = M34 + s01 * M04 + s11 * M14 + s21 * M24;
    const double H44 = M44 + s02 * M04 + s12 * M14 + s22 * M24 + hc_[4];
    (void)M43;
    const double idet = 1.0 / (a00 * a11 - a01 * a01);
    const double i00 = a11 * idet, i01 = -a01 * idet, i11 = a00 * idet;
    // t = Huu^-1 [Hux hu]
    const double t00 = i00 * X00 + i01 * X10, t01 = i00 * X01 + i01 * X11, t02 = i00 * X02 + i01 * X12, t03 = i00 * X03 + i01 * X13,
                 t04 = i00 * X04 + i01 * X14, t05 = i00 * hu0 + i01 * hu1;
    const double t10 = i01 * X00 + i11 * X10, t11 = i01 * X01 + i11 * X11, t12 = i01 * X02 + i11 * X12, t13 = i01 * X03 + i11 * X13,
                 t14 = i01 * X04 + i11 * X14, t15 = i01 * hu0 + i11 * hu1;
    // new value function (registers), then the loads of the next stage, then this stage's gains: in that order, see above
    const double N00 = H00 - (X00 * t00 + X10 * t10), N01 = H01 - (X00 * t01 + X10 * t11), N02 = H02 - (X00 * t02 + X10 * t12);
    const double N03 = H03 - (X00 * t03 + X10 * t13), N04 = H04 - (X00 * t04 + X10 * t14);
    const double N11 = H11 - (X01 * t01 + X11 * t11), N12 = H12 - (X01 * t02 + X11 * t12), N13 = H13 - (X01 * t03 + X11 * t13);
    const double N14 = H14 - (X01 * t04 + X11 * t14);
    const double N22 = H22 - (X02 * t02 + X12 * t12), N23 = H23 - (X02 * t03 + X12 * t13), N24 = H24 - (X02 * t04 + X12 * t14);
    const double N33 = H33 - (X03 * t03 + X13 * t13), N34 = H34 - (X03 * t04 + X13 * t14);
    const double N44 = H44 - (X04 * t04 + X14 * t14);
    p0 = hx0 - (X00 * t05 + X10 * t15); p1 = hx1 - (X01 * t05 + X11 * t15); p2 = hx2 - (X02 * t05 + X12 * t15);
    p3 = hx3 - (X03 * t05 + X13 * t15); p4 = hx4 - (X04 * t05 + X14 * t15);
    P00 = N00; P01 = N01; P02 = N02; P03 = N03; P04 = N04; P11 = N11; P12 = N12; P13 = N13; P14 = N14;
    P22 = N22; P23 = N23; P24 = N24; P33 = N33; P34 = N34; P44 = N44;
    {
      const int kn = k > 0 ? k - 1 : 0;  // the last pass re-reads stage 0 (harmless) so that the loop body has no branch here
      const wsp_f64 *s = m + o_ab + kn * 15, *d = m + o_d + kn * 5;
#pragma unroll
      for (int i = 0; i < 15; ++i) sc[i] = s[i];
#pragma unroll
      for (int i = 0; i < 5; ++i) dc[i] = d[i];
    }
    wsp_f64 *K = m + o_kk + k * 12;
    K[0] = -t00; K[1] = -t01; K[2] = -t02; K[3] = -t03; K[4] = -t04;
    K[5] = -t10; K[6] = -t11; K[7] = -t12; K[8] = -t13; K[9] = -t14;
    K[10] = -t05; K[11] = -t15;
  }
  wsp_f64 *rP = m + o_rP;
  rP[0] = P00; rP[1] = P01; rP[2] = P02; rP[3] = P03; rP[4] = P04;
  rP[5] = P01; rP[6] = P11; rP[7] = P12; rP[8] = P13; rP[9] = P14;
  rP[10] = P02; rP[11] = P12; rP[12] = P22; rP[13] = P23; rP[14] = P24;
  rP[15] = P03; rP[16] = P13; rP[17] = P23; rP[18] = P33; rP[19] = P34;
  rP[20] = P04; rP[21] = P14; rP[22] = P24; rP[23] = P34; rP[24] = P44;
  rP[25] = p0; rP[26] = p1; rP[27] = p2; rP[28] = p3; rP[29] = p4;
}

// Operands of the sweep in homogeneous coordinates (below), as workspace words or constants, and the sweep itself in plain loops with
// every sum in the order v_mfma_f64_16x16x4_f64 forms it (k ascending, one fused multiply-add per k onto the accumulator): what the CPU
// build runs, bit for bit what the matrix-core sweep returns (tools/src/riccati_mfma_bench.hip: 0 of 2,880 gains differ); oracle/cfz_port.c
// has the same loops.
struct RicEnt { int off, stride; double c; };  // value at stage k: stride ? m[off + k * stride] : c
CFZ_FN RicEnt ric_T(int r, int j, double dt, int o_ab, int o_d) {  // T[r][j], r < 6, j < 8
  RicEnt e = {0, 0, 0.0};
  if (r >= 6 || j >= 8) return e;
  if (r == 5) { e.c = j == 5 ? 1.0 : 0.0; return e; }
  if (j < 5) {
    e.c = r == j ? 1.0 : 0.0;
    int idx = -1;
    if (r == 0 && j >= 2) idx = j - 2; else if (r == 1 && j >= 2) idx = 5 + j - 2; else if (r == 2 && j >= 3) idx = 11 + j - 3;
    if (idx >= 0) { e.off = o_ab + idx; e.stride = 15; }
    return e;
  }
  if (j == 5) { e.off = o_d + r; e.stride = 5; return e; }
  const int u = j - 6;
  if (r < 3) { e.off = o_ab + 5 * r + 3 + u; e.stride = 15; } else e.c = (r - 3 == u) ? dt : 0.0;
  return e;
}
CFZ_FN RicEnt ric_H(int a, int b, int o_hc, int o_gk) {  // Ht[a][b], a, b < 8 (variables z0..z4, 1, u0, u1)
  RicEnt e = {0, 0, 0.0};
  if (a < 0 || a >= 8 || b >= 8) return e;
  if (a > b) { const int t = a; a = b; b = t; }
  const int za = a < 5 ? a : (a == 5 ? -1 : a - 1), zb = b < 5 ? b : (b == 5 ? -1 : b - 1);  // index in (z, u) = 0..6, -1 = the constant
  if (za < 0 && zb < 0) return e;
  if (za < 0 || zb < 0) { e.off = o_gk + (za < 0 ? zb : za); e.stride = kNP; return e; }
  int idx = -1;
  if (za == zb) idx = za; else if (za == 0 && zb == 1) idx = 7; else if (za == 0 && zb == 2) idx = 8; else if (za == 1 && zb == 2) idx = 9; else if (za == 3 && zb == 6) idx = 10;
  if (idx >= 0) { e.off = o_hc + idx; e.stride = 11; }
  return e;
}
CFZ_FN double ric_val(const wsp_f64 *m, const RicEnt &e, int k) { return e.stride ? m[e.off + k * e.stride] : e.c; }
#if !defined(__HIP_DEVICE_COMPILE__)
static void riccati_backward_dense(wsp_f64 *m, int N, double dt, int o_ab, int o_hc, int o_gk, int o_d, int o_kk, int o_rP) {
  double P[6][6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) P[i][j] = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    double T[6][8], Ht[8][8], Y[6][8], M[8][8], V[2][6];
    for (int r = 0; r < 6; ++r) for (int j = 0; j < 8; ++j) T[r][j] = k < N - 1 ? ric_val(m, ric_T(r, j, dt, o_ab, o_d), k) : 0.0;  // (the terminal stage has no dynamics)
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 8; ++b) Ht[a][b] = ric_val(m, ric_H(a, b, o_hc, o_gk), k);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) { double s_ = 0.0; for (int q = 0; q < 6; ++q) s_ = fma(P[q][i], T[q][j], s_); Y[i][j] = s_; }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { double s_ = Ht[i][j]; for (int q = 0; q < 6; ++q) s_ = fma(T[q][i], Y[q][j], s_); M[i][j] = s_; }
    const double idet = 1.0 / fma(M[6][6], M[7][7], -(M[6][7] * M[7][6]));
    const double i00 = M[7][7] * idet, i01 = -M[6][7] * idet, i10 = -M[7][6] * idet, i11 = M[6][6] * idet;
    for (int j = 0; j < 6; ++j) { V[0][j] = fma(i00, M[6][j], i01 * M[7][j]); V[1][j] = fma(i10, M[6][j], i11 * M[7][j]); }
    for (int g = 0; g < 2; ++g) for (int j = 0; j < 6; ++j) m[o_kk + k * 12 + (j < 5 ? 5 * g + j : 10 + g)] = -V[g][j];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) P[i][j] = fma(-M[7][i], V[1][j], fma(-M[6][i], V[0][j], M[i][j]));
  }
  for (int i = 0; i < 5; ++i) { for (int j = 0; j < 5; ++j) m[o_rP + 5 * i + j] = P[i][j]; m[o_rP + 25 + i] = P[5][i]; }
}
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CFZ_RICCATI_SCALAR)
// The same sweep on the matrix cores (round 5; tools/src/riccati_mfma_bench.hip is the go / no-go measurement: 1,150 cycles per stage
// against 1,850 for the one-lane sweep above, gains equal to 9e-16).  The first wavefront, all 64 lanes.  Homogeneous coordinates, variables
// ordered [z (5), 1, u (2)]:  [z+; 1] = T [z; 1; u],  T = [[A d B], [0 1 0]]  (6 x 8),  Pt = [[P p], [p' 0]]  (6 x 6),
//   M = T' Pt T + Ht  (8 x 8; Ht = the stage's Hessian with its gradient in row / column 5),   Pt <- M_kk - M_ke M_ee^-1 M_ek,  K = -M_ee^-1 M_ek,
// as five dependent v_mfma_f64_16x16x4_f64 per stage (result register r of lane l = row (l >> 4) + 4 r, column l & 15):
//   Y = Pt T     two k-steps; the A operand is Pt's accumulator as it stands (Pt is symmetric: register s of lane l is A[l & 15][(l >> 4) + 4 s]);
//   M = Tt' Y    two k-steps onto an accumulator that starts as Ht; the B operand is Y's accumulator as it stands; Tt' repeats the rows of
//                u at rows 8, 9 and (crossed) 12, 13, so that the 16-lane groups 0 and 1 hold M's rows of u in their own registers;
//   Pt <- M - U V   one k-step: A = -M_ke (groups 0, 1: register 2), B = V = M_ee^-1 M_ek (from registers 2 and 3; M_ee by v_readlane).
// The operands are built on the fly from the stage data the one-lane sweep reads (15 sensitivities, d, 11 Hessian entries, 7 gradient
// entries): every lane knows once and for all which workspace word (or constant) each of its eight operand entries is.  The terminal
// stage is the same step from Pt = 0.  (The CPU build and the oracle run the one-lane algebra: same recursion, sums in another order.)
typedef double ric_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double ric_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// operand entry without a branch: value at stage k = fma(m[off + k stride], mask, c) -- a data entry has mask 1, c 0 (the product with 1 and
// the sum with 0 are exact); a constant reads a harmless finite word (the stage's first Hessian entry) with mask 0.  The exec-masked loads the
// `stride ? .. : ..` form compiled to were a third of the sweep's instructions, and the sweep is bound by their issue, not by the five
// matrix instructions (tools/src/riccati_mfma_bench.hip: 140 instructions per stage)
struct RicFetch { int addr, stride; double mask, c; };
__device__ __forceinline__ RicFetch ric_fetch(const RicEnt &e, int k, int o_hc) {
  RicFetch f;  // (the harmless word: the stage's first Hessian entry, written for every stage and finite)
  f.stride = e.stride ? e.stride : 11; f.addr = (e.stride ? e.off : o_hc) + k * f.stride; f.mask = e.stride ? 1.0 : 0.0; f.c = e.stride ? 0.0 : e.c;
  return f;
}
__device__ __forceinline__ double ric_get(const wsp_f64 *m, const RicFetch &f) { return fma(m[f.addr], f.mask, f.c); }
__device__ __attribute__((noinline)) void riccati_backward_mfma(wsp_f64 *m, int N, double dt, int o_ab, int o_hc, int o_gk, int o_d, int o_kk, int o_rP) {
  const int lane = threadIdx.x & 63, lo = lane & 15, g = lane >> 4;
  // row i of Tt' / Ht stands for variable: 0..7 themselves, 8 -> u0, 9 -> u1, 12 -> u1, 13 -> u0, the others are zero rows
  const int vlo = lo < 8 ? lo : (lo == 8 ? 6 : lo == 9 ? 7 : lo == 12 ? 7 : lo == 13 ? 6 : -1);
  int k = N - 1;
  RicFetch fB[2], fA[2], fH[4];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    fB[s_] = ric_fetch(ric_T(g + 4 * s_, lo, dt, o_ab, o_d), k > 0 ? k - 1 : 0, o_hc);  // (T is first needed for stage N - 2)
    fA[s_] = ric_fetch(vlo >= 0 ? ric_T(g + 4 * s_, vlo, dt, o_ab, o_d) : RicEnt{0, 0, 0.0}, k > 0 ? k - 1 : 0, o_hc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = g + 4 * r, vr = row < 8 ? row : (row == 8 ? 6 : row == 9 ? 7 : row == 12 ? 7 : row == 13 ? 6 : -1);
    fH[r] = ric_fetch(ric_H(vr, lo, o_hc, o_gk), k, o_hc);
  }
  ric_v4d P = {0.0, 0.0, 0.0, 0.0};
  double b0 = 0.0, b1 = 0.0, a0 = 0.0, a1 = 0.0;  // (the terminal stage has no dynamics: T = 0 against Pt = 0)
  ric_v4d H = {ric_get(m, fH[0]), ric_get(m, fH[1]), ric_get(m, fH[2]), ric_get(m, fH[3])};
  const int kout = o_kk + (lo < 5 ? 5 * g + lo : 10 + g);  // where this lane's gain goes (groups 0, 1, columns 0..5)
  for (; k >= 0; --k) {
    ric_v4d Y = {0.0, 0.0, 0.0, 0.0};
    Y = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], b0, Y, 0, 0, 0);
    Y = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], b1, Y, 0, 0, 0);
    ric_v4d M = H;
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, Y[0], M, 0, 0, 0);
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Y[1], M, 0, 0, 0);
    // the next stage's operands: requested now, used after this stage's last matrix instruction (the last pass re-reads stage 0: harmless)
    if (k > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) fH[r].addr -= fH[r].stride;
    }
    const double nb0 = ric_get(m, fB[0]), nb1 = ric_get(m, fB[1]), na0 = ric_get(m, fA[0]), na1 = ric_get(m, fA[1]);
    const ric_v4d nH = {ric_get(m, fH[0]), ric_get(m, fH[1]), ric_get(m, fH[2]), ric_get(m, fH[3])};
    if (k > 1) {
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) { fB[s_].addr -= fB[s_].stride; fA[s_].addr -= fA[s_].stride; }
    }
    // M_ee: rows 6, 7 = groups 2, 3, register 1; columns 6, 7
    const double m66 = ric_lane(M[1], 38), m67 = ric_lane(M[1], 39), m76 = ric_lane(M[1], 54), m77 = ric_lane(M[1], 55);
    const double idet = 1.0 / fma(m66, m77, -(m67 * m76));  // (explicit fused multiply-adds: the CPU mirrors of this sweep use the same)
    const double i00 = m77 * idet, i01 = -m67 * idet, i10 = -m76 * idet, i11 = m66 * idet;
    const double m6 = g == 0 ? M[2] : M[3], m7 = g == 0 ? M[3] : M[2];  // rows of u0, u1 in this group's registers (8 / 12, 13 / 9)
    const double ia = g == 0 ? i00 : i10, ib = g == 0 ? i01 : i11;
    const double Vf = fma(ia, m6, ib * m7), V = g < 2 ? Vf : 0.0;
    const double U = g < 2 ? -M[2] : 0.0;
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(U, V, M, 0, 0, 0);
    if (g < 2 && lo < 6) m[kout + k * 12] = -V;
    P = M; b0 = nb0; b1 = nb1; a0 = na0; a1 = na1; H = nH;
  }
  // value function of stage 0: P (5 x 5, rows 0..3 in register 0 of group = row, row 4 in register 1 of group 0), p = row 5 (register 1 of group 1)
  if (lo < 5) {
    m[o_rP + 5 * g + lo] = P[0];
    if (g == 0) m[o_rP + 20 + lo] = P[1];
    if (g == 1) m[o_rP + 25 + lo] = P[1];
  }
}
#define CFZ_RICCATI(...) do { if (threadIdx.x < 64) { riccati_backward_mfma(__VA_ARGS__); } __syncthreads(); } while (0)
#elif defined(__HIP_DEVICE_COMPILE__)
#define CFZ_RICCATI(...) CFZ_SERIAL(riccati_backward(__VA_ARGS__))  // (-DCFZ_RICCATI_SCALAR: the one-lane sweep, for comparison)
#else
#define CFZ_RICCATI(...) riccati_backward_dense(__VA_ARGS__)
#endif

// ------------------------------------------------------------------------------ forward step and costates as scans
// Once the gains are known the forward sweep is a LINEAR recurrence, z_{k+1} = (A_k + B_k K_k) z_k + (B_k k_k + d_k), and
// so is the costate sweep, lam_{k-1} = q_k + A_k' lam_k.  Both are prefix compositions of affine maps, done here in
// log2(32) = 5 Kogge-Stone steps by the first wavefront, one stage per lane (cross-lane traffic: ds_bpermute), instead
// of 30 dependent stages on one lane.  The CPU build runs the same combine steps in the same order on arrays.
struct Aff5 { double M[25], v[5]; };  // z -> M z + v

// t <- t after s  (s is applied first): M = t.M s.M,  v = t.M s.v + t.v
CFZ_FN void aff5_after(Aff5 &t, const Aff5 &s) {
  // row i of the product needs row i of t.M only: rows are replaced one by one, no second copy of the matrix is live
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double a0 = t.M[i * 5 + 0], a1 = t.M[i * 5 + 1], a2 = t.M[i * 5 + 2], a3 = t.M[i * 5 + 3], a4 = t.M[i * 5 + 4];
    t.v[i] = t.v[i] + a0 * s.v[0] + a1 * s.v[1] + a2 * s.v[2] + a3 * s.v[3] + a4 * s.v[4];
#pragma unroll
    for (int j = 0; j < 5; ++j)
      t.M[i * 5 + j] = a0 * s.M[0 * 5 + j] + a1 * s.M[1 * 5 + j] + a2 * s.M[2 * 5 + j] + a3 * s.M[3 * 5 + j] + a4 * s.M[4 * 5 + j];
  }
}

// closed-loop transition of stage k (k < N - 1), identity beyond
CFZ_FN void aff5_stage(const wsp_f64 *m, int k, int N, double dt, int o_ab, int o_d, int o_kk, Aff5 &t) {
#pragma unroll
  for (int i = 0; i < 25; ++i) t.M[i] = (i % 6 == 0) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 5; ++i) t.v[i] = 0.0;
  if (k >= N - 1) return;
  const wsp_f64 *K = m + o_kk + k * 12, *s = m + o_ab + k * 15, *d = m + o_d + k * 5;
  double K0[5], K1[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) { K0[j] = K[j]; K1[j] = K[5 + j]; }
  const double k0 = K[10], k1 = K[11];
  // rows 0..2: A = I + [0 0 s_r0 s_r1 s_r2] (s20 = 1 is the diagonal), B = [s_r3 s_r4]
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double b0 = s[r * 5 + 3], b1 = s[r * 5 + 4];
#pragma unroll
    for (int j = 0; j < 5; ++j) t.M[r * 5 + j] += b0 * K0[j] + b1 * K1[j];
    t.v[r] = d[r] + b0 * k0 + b1 * k1;
  }
  t.M[0 * 5 + 2] += s[0]; t.M[0 * 5 + 3] += s[1]; t.M[0 * 5 + 4] += s[2];
  t.M[1 * 5 + 2] += s[5]; t.M[1 * 5 + 3] += s[6]; t.M[1 * 5 + 4] += s[7];
  t.M[2 * 5 + 3] += s[11]; t.M[2 * 5 + 4] += s[12];
#pragma unroll
  for (int j = 0; j < 5; ++j) { t.M[3 * 5 + j] += dt * K0[j]; t.M[4 * 5 + j] += dt * K1[j]; }
  t.v[3] = d[3] + dt * k0; t.v[4] = d[4] + dt * k1;
}

// costate map of stage k: lam -> A_k' lam + q_k with A_k' = I + X, X = [x20 x21 | x30 x31 x32 | x40 x41 x42] (the
// transposed sensitivities); products of such matrices keep the pattern
struct Cos5 { double x[8], v[5]; };
CFZ_FN void cos5_after(Cos5 &t, const Cos5 &s) {  // t <- t after s
  const double *x = t.x, *y = s.x;
  double vn[5];
  vn[0] = t.v[0] + s.v[0]; vn[1] = t.v[1] + s.v[1];
  vn[2] = t.v[2] + s.v[2] + x[0] * s.v[0] + x[1] * s.v[1];
  vn[3] = t.v[3] + s.v[3] + x[2] * s.v[0] + x[3] * s.v[1] + x[4] * s.v[2];
  vn[4] = t.v[4] + s.v[4] + x[5] * s.v[0] + x[6] * s.v[1] + x[7] * s.v[2];
  double xn[8];
  xn[0] = x[0] + y[0]; xn[1] = x[1] + y[1];
  xn[2] = x[2] + y[2] + x[4] * y[0]; xn[3] = x[3] + y[3] + x[4] * y[1]; xn[4] = x[4] + y[4];
  xn[5] = x[5] + y[5] + x[7] * y[0]; xn[6] = x[6] + y[6] + x[7] * y[1]; xn[7] = x[7] + y[7];
#pragma unroll
  for (int i = 0; i < 8; ++i) t.x[i] = xn[i];
#pragma unroll
  for (int i = 0; i < 5; ++i) t.v[i] = vn[i];
}

// lane k's part before the costate scan: u_k = K_k z_k + k_k (into dp), q_k = (H dp + g)_z of stage k, the map of stage k
CFZ_FN void cos5_stage(wsp_f64 *m, int k, int N, int o_ab, int o_hc, int o_gk, int o_kk, int o_dp, Cos5 &t) {
#pragma unroll
  for (int i = 0; i < 8; ++i) t.x[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 5; ++i) t.v[i] = 0.0;
  if (k >= N) return;
  const wsp_f64 *K = m + o_kk + k * 12, *h = m + o_hc + k * 11, *gk = m + o_gk + k * kNP;
  wsp_f64 *z = m + o_dp + k * kNP;
  const double z0 = z[0], z1 = z[1], z2 = z[2], z3 = z[3], z4 = z[4];
  const double u0 = (K[10] + K[0] * z0 + K[1] * z1) + (K[2] * z2 + K[3] * z3 + K[4] * z4);
  const double u1 = (K[11] + K[5] * z0 + K[6] * z1) + (K[7] * z2 + K[8] * z3 + K[9] * z4);
  z[5] = u0; z[6] = u1;
  if (k < 1) return;
  t.v[0] = gk[0] + h[0] * z0 + h[7] * z1 + h[8] * z2;
  t.v[1] = gk[1] + h[7] * z0 + h[1] * z1 + h[9] * z2;
  t.v[2] = gk[2] + h[8] * z0 + h[9] * z1 + h[2] * z2;
  t.v[3] = gk[3] + h[3] * z3 + h[10] * u1;
  t.v[4] = gk[4] + h[4] * z4;
  if (k + 1 < N) {  // stage N - 1 has no dynamics: identity
    const wsp_f64 *s = m + o_ab + k * 15;
    t.x[0] = s[0]; t.x[1] = s[5];
    t.x[2] = s[1]; t.x[3] = s[6]; t.x[4] = s[11];
    t.x[5] = s[2]; t.x[6] = s[7]; t.x[7] = s[12];
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
// The scans are out of line like the backward sweep (inlined they cost 11 % of the throughput: the kernel's register
// allocation suffers), but only the first 32 lanes call them (CFZ_WAVE0; one stage per lane, N <= 32): the prologue of
// forward_scan saves 37 callee-saved registers of every calling lane to scratch, and every finished solve's release writes
// that back to HBM.
#ifndef CFZ_SCAN
#define CFZ_SCAN __device__ __attribute__((noinline)) void
#endif
CFZ_SCAN forward_scan(wsp_f64 *m, int N, double dt, int o_ab, int o_d, int o_kk, int o_rP, int o_p, int o_dp, int o_x0,
                       int o_pi0, int o_dpi0) {
  if (threadIdx.x >= 64) return;  // the first wavefront, stage k on lane k
  const int k = threadIdx.x;
  Aff5 t;
  aff5_stage(m, k, N, dt, o_ab, o_d, o_kk, t);
  // One Kogge-Stone step in two halves, the partner's vector first and its matrix after: t.v += t.M s.v needs only the five values of
  // s.v beside t, so that the step's peak is t (30 doubles) + s.M (25) + a row's temporaries instead of t + all of s -- the function then
  // fits the 144 caller-saved VGPRs of the calling convention and its prologue no longer saves 37 registers per lane to scratch at every
  // call (= every interior-point iteration: 3.4 GB written per bench launch, profiles/r5c_pmc_WRITE_SIZE.csv).  Same sums in the same
  // order as aff5_after (which the CPU build calls).
#pragma unroll
  for (int dd = 1; dd < 32; dd <<= 1) {
    {
      double sv[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) sv[i] = __shfl_up(t.v[i], dd, 64);
      if (k >= dd) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
          t.v[i] = t.v[i] + t.M[i * 5 + 0] * sv[0] + t.M[i * 5 + 1] * sv[1] + t.M[i * 5 + 2] * sv[2] + t.M[i * 5 + 3] * sv[3] + t.M[i * 5 + 4] * sv[4];
      }
    }
    asm volatile("" ::: "memory");
    {
      double sM[25];
#pragma unroll
      for (int i = 0; i < 25; ++i) sM[i] = __shfl_up(t.M[i], dd, 64);
      if (k >= dd) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const double a0 = t.M[i * 5 + 0], a1 = t.M[i * 5 + 1], a2 = t.M[i * 5 + 2], a3 = t.M[i * 5 + 3], a4 = t.M[i * 5 + 4];
#pragma unroll
          for (int j = 0; j < 5; ++j)
            t.M[i * 5 + j] = a0 * sM[0 * 5 + j] + a1 * sM[1 * 5 + j] + a2 * sM[2 * 5 + j] + a3 * sM[3 * 5 + j] + a4 * sM[4 * 5 + j];
        }
      }
    }
    asm volatile("" ::: "memory");
  }
  const double z0 = m[o_x0 + 0] - m[o_p + 0], z1 = m[o_x0 + 1] - m[o_p + 1], z2 = m[o_x0 + 2] - m[o_p + 2],
               z3 = m[o_x0 + 3] - m[o_p + 3], z4 = m[o_x0 + 4] - m[o_p + 4];
  // (the value function is read BEFORE the step is stored: it stands in the step's slots of stages 1..5, make_layout; one wavefront,
  // program order, and a wavefront's DS instructions complete in the order they were issued)
  if (k < 5) {  // step of the initial-state multiplier from the value function at stage 0
    const wsp_f64 *rP = m + o_rP;
    const double s_ = rP[25 + k] + rP[k * 5 + 0] * z0 + rP[k * 5 + 1] * z1 + rP[k * 5 + 2] * z2 + rP[k * 5 + 3] * z3 + rP[k * 5 + 4] * z4;
    m[o_dpi0 + k] = -s_ - m[o_pi0 + k];
  }
  if (k + 1 < N) {
    wsp_f64 *zn = m + o_dp + (k + 1) * kNP;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      zn[i] = t.v[i] + t.M[i * 5 + 0] * z0 + t.M[i * 5 + 1] * z1 + t.M[i * 5 + 2] * z2 + t.M[i * 5 + 3] * z3 + t.M[i * 5 + 4] * z4;
  }
  if (k == 0) { wsp_f64 *dp = m + o_dp; dp[0] = z0; dp[1] = z1; dp[2] = z2; dp[3] = z3; dp[4] = z4; }
}

CFZ_SCAN costate_scan(wsp_f64 *m, int N, int o_ab, int o_hc, int o_gk, int o_kk, int o_dp, int o_dpi, int o_pi) {
  if (threadIdx.x >= 64) return;
  const int k = threadIdx.x;
  Cos5 t;
  cos5_stage(m, k, N, o_ab, o_hc, o_gk, o_kk, o_dp, t);
#pragma unroll
  for (int dd = 1; dd < 32; dd <<= 1) {  // suffix composition: T_k = C_k after-applied-to (C_{k+1} ... C_{N-1})
    Cos5 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) s.x[i] = __shfl_down(t.x[i], dd, 64);
#pragma unroll
    for (int i = 0; i < 5; ++i) s.v[i] = __shfl_down(t.v[i], dd, 64);
    if (k + dd < N) cos5_after(t, s);
  }
  if (k >= 1 && k < N) {
    wsp_f64 *q = m + o_dpi + (k - 1) * 5;
    const wsp_f64 *pi = m + o_pi + (k - 1) * 5;
#pragma unroll
    for (int i = 0; i < 5; ++i) q[i] = t.v[i] - pi[i];
  }
}
#else
static void forward_scan(double *m, int N, double dt, int o_ab, int o_d, int o_kk, int o_rP, int o_p, int o_dp, int o_x0,
                         int o_pi0, int o_dpi0) {
  static thread_local Aff5 t[64], s[64];
  for (int k = 0; k < 64; ++k) aff5_stage(m, k, N, dt, o_ab, o_d, o_kk, t[k]);
  for (int dd = 1; dd < 32; dd <<= 1) {
    for (int k = 0; k < 64; ++k) s[k] = t[k >= dd ? k - dd : k];
    for (int k = dd; k < 64; ++k) aff5_after(t[k], s[k]);
  }
  const double z0 = m[o_x0 + 0] - m[o_p + 0], z1 = m[o_x0 + 1] - m[o_p + 1], z2 = m[o_x0 + 2] - m[o_p + 2],
               z3 = m[o_x0 + 3] - m[o_p + 3], z4 = m[o_x0 + 4] - m[o_p + 4];
  const double *rP = m + o_rP;  // (read before the step is stored: it stands in the step's slots of stages 1..5, make_layout)
  for (int k = 0; k < 5; ++k) {
    const double s_ = rP[25 + k] + rP[k * 5 + 0] * z0 + rP[k * 5 + 1] * z1 + rP[k * 5 + 2] * z2 + rP[k * 5 + 3] * z3 + rP[k * 5 + 4] * z4;
    m[o_dpi0 + k] = -s_ - m[o_pi0 + k];
  }
  for (int k = 0; k + 1 < N; ++k) {
    double *zn = m + o_dp + (k + 1) * kNP;
    for (int i = 0; i < 5; ++i)
      zn[i] = t[k].v[i] + t[k].M[i * 5 + 0] * z0 + t[k].M[i * 5 + 1] * z1 + t[k].M[i * 5 + 2] * z2 + t[k].M[i * 5 + 3] * z3 + t[k].M[i * 5 + 4] * z4;
  }
  double *dp = m + o_dp; dp[0] = z0; dp[1] = z1; dp[2] = z2; dp[3] = z3; dp[4] = z4;
}

static void costate_scan(double *m, int N, int o_ab, int o_hc, int o_gk, int o_kk, int o_dp, int o_dpi, int o_pi) {
  static thread_local Cos5 t[64], s[64];
  for (int k = 0; k < 64; ++k) cos5_stage(m, k, N, o_ab, o_hc, o_gk, o_kk, o_dp, t[k]);
  for (int dd = 1; dd < 32; dd <<= 1) {
    for (int k = 0; k < 64; ++k) s[k] = t[k + dd < 64 ? k + dd : k];
    for (int k = 0; k + dd < N; ++k) cos5_after(t[k], s[k]);
  }
  for (int k = 1; k < N; ++k) {
    double *q = m + o_dpi + (k - 1) * 5;
    const double *pi = m + o_pi + (k - 1) * 5;
    for (int i = 0; i < 5; ++i) q[i] = t[k].v[i] - pi[i];
  }
}
#endif

#if defined(CFZ_NO_DELTAC)  // diagnostic builds: what the dual regularisation costs the hot loop
#define kDeltaC(sp) 0.0
#else
#define kDeltaC(sp) (sp).reg_dual_rows
#endif
// ------------------------------------------------------------------------------ feasibility restoration
// IPOPT answers a failed line search with its restoration phase (paper sec. 3.3).  Here (oracle/mpc_nlp.py MpcNlp.restore, oracle/cfz_port.c
// restore): Levenberg-Marquardt on the squared violations of the separation rows of stages >= 1 and of the boxes,
//     min  rho / 2 sum max(0, dmin + eps - sep_kr)^2  +  rho_b / 2 sum (excess over the boxes shrunk by m_b)^2     s.t. z_0 = x0, dynamics
// on the solver's own stage recursion (H_k = (zeta + lambda) I + rho sum a a' over the violated rows + rho_b on the violated boxes,
// zeta = sqrt(mu)), Armijo line search on the l1 merit (objective + eta |dynamics defects|_1), working set refreshed at every iterate
// and held inside a line search.  Cold code: a function of its own (registers and instructions of its own), called by every lane.
constexpr double kRestoRho = 1000.0, kRestoRhoBox = 1e5, kRestoBoxMargin = 2e-3, kRestoKappa = 0.1, kRestoArmijo = 1e-4;
constexpr int kRestoMaxIter = 40, kRestoStall = 8;

// lane partials at p + alpha dp with the working set held: th |dynamics defects|_1, ph objective
CFZ_CALL void resto_partials(const KSpec &sp, const KDer &dv, double *m, const Lay &L, double alpha, double eps, int tid, double &th_o,
                             double &ph_o) {
  const int N = sp.N, nb = L.nb;
  const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
  double th = 0.0, ph = 0.0;
  if (k < N) {
    double pt[kNP];
    for (int i = 0; i < kNP; ++i) pt[i] = m[L.p + k * kNP + i] + alpha * m[L.dp + k * kNP + i];
    if (k >= 1) {
      double sn, cn;
      sincos(pt[2], &sn, &cn);
      for (int j = sub; j < nb; j += kLPS) {
        double sep[2];
        block_sep(sp, m, L, k, j, sel_ptr(m, L)[k * nb + j], pt[0], pt[1], cn, sn, sep);
#pragma unroll
        for (int r = 0; r < 2; ++r) { const double v = sp.dmin + eps - sep[r]; if (v > 0.0) ph += 0.5 * kRestoRho * v * v; }
      }
    }
    if (sub == 0) {
      for (int q = (k == 0 ? 4 : 0); q < 6; ++q) {
        const double el = sp.bounds[2 * q] + kRestoBoxMargin - pt[bcol(q)], eu = pt[bcol(q)] - sp.bounds[2 * q + 1] + kRestoBoxMargin;
        if (el > 0.0) ph += 0.5 * kRestoRhoBox * el * el;
        if (eu > 0.0) ph += 0.5 * kRestoRhoBox * eu * eu;
      }
      if (k == 0) for (int i = 0; i < 5; ++i) th += fabs(pt[i] - m[L.x0 + i]);
      if (k + 1 < N) {
        double F[5];
        rk4_step_h<false>(pt, pt[5], pt[6], dv.rk_h, dv.rk_hh, dv.rk_h6, dv.iwb, sp.rk_substeps, F, nullptr);
        for (int i = 0; i < 5; ++i) th += fabs(F[i] - (m[L.p + (k + 1) * kNP + i] + alpha * m[L.dp + (k + 1) * kNP + i]));
      }
    }
  }
  th_o = th; ph_o = ph;
}

#if defined(__HIP_DEVICE_COMPILE__)
#if defined(CFZ_RESTO_INLINE)
#define CFZ_COLD __device__ __forceinline__
#else
#define CFZ_COLD __device__ __attribute__((noinline, cold))
#endif
#else
#define CFZ_COLD static
#endif
// Returns the solve's iteration counter after the restoration, +1, negated if the restoration failed: positive when the iterate in L.p has
// been restored (rows of stages >= 1 within the goal, boxes with margin, dynamics no worse than at entry; clipped half the margin inside
// the boxes), negative when the restoration stalls, fails a line search or reaches its iteration limit (the caller ends with status 5:
// locally infeasible).  Uses the row-residual array L.cj for the row values; slacks and multipliers are not touched (the caller
// re-initialises them).  Everything goes in and out by value: a counter of the solver's hot loop whose address is taken lives in scratch.
// The parity of the reduction exchange (xpar) is 0 at entry and at exit: the caller's is parked while this runs (an even number of
// reductions is not guaranteed here, so the function realigns with one barrier at its end).
// (the workspace comes in as an address-space-3 pointer, like the sweeps': cast back to a generic one here, the compiler then knows where
// every access of the inlined helpers goes and emits DS instructions instead of FLAT ones)
CFZ_COLD int restore_instance(const KSpec &sp, const KDer &dv, wsp_f64 *mw, const Lay &L, double mu, int iter) {
  double *m = (double *)mw;
  const int N = sp.N, nb = L.nb, nr = L.nr;
  int xpar = 0;
  (void)xpar;
  CFZ_PART(rd, 6);
  CFZ_PART(qx, 15);
  double ro[6];
  const double zeta = CFZ_UNIFORM(sqrt(mu));
  double eta = 0.0, eps = 0.0, lm = 0.0, vgoal = 0.0, vref = INFINITY, dgoal = 0.0;
  int ref_it = 0;
  for (int rit = 0;; ++rit) {
    // ---- working set (refreshed from the second iteration on), row values into L.cj, dynamics with sensitivities --------------
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double cmax = 0.0, csum = 0.0, v0 = 0.0;
      if (k < N) {
        const double *pk = m + L.p + k * kNP;
        const double x = pk[0], y = pk[1];
        double sn, cn;
        sincos(pk[2], &sn, &cn);
        if (sub == 0) { m[L.cs + 2 * k] = cn; m[L.cs + 2 * k + 1] = sn; }
        for (int j = sub; j < nb; j += kLPS) {
          const int t = k * nb + j;
          double sep[2];
          const int c0 = sel_ptr(m, L)[t];
          if (rit > 0) {
            double A[4][2], b[4], V[4][2];
            block_polygon(sp, m, L, k, j, A, b, V);
            const int c1 = select_rows_sep(A, b, V, x, y, cn, sn, sp.g, c0, sep, sp.vv_rows);
            if (c1 != c0) sel_ptr(m, L)[t] = c1;
          } else {
            block_sep(sp, m, L, k, j, c0, x, y, cn, sn, sep);
          }
          m[L.cj + 2 * t] = sep[0]; m[L.cj + 2 * t + 1] = sep[1];
          if (k >= 1) v0 = fmax(v0, sp.dmin - fmin(sep[0], sep[1]));
        }
        if (tid == 0) for (int i = 0; i < 5; ++i) { const double r = m[L.p + i] - m[L.x0 + i]; cmax = fmax(cmax, fabs(r)); csum += fabs(r); }
        if (k + 1 < N) {
          double F[5], Sa[3], Sb[3];
          rk4_sens2(pk, pk[5], pk[6], dv.rk_h, dv.rk_hh, dv.rk_h6, dv.iwb, sp.rk_substeps, sub, F, Sa, Sb);
          if (sub < 4) for (int r = 0; r < 3; ++r) m[L.ab + k * 15 + r * 5 + sub] = Sa[r];
          if (sub == 0) {
            for (int r = 0; r < 3; ++r) m[L.ab + k * 15 + r * 5 + 4] = Sb[r];
            for (int i = 0; i < 5; ++i) {
              const double d = F[i] - m[L.p + (k + 1) * kNP + i];
              m[L.d + k * 5 + i] = d; cmax = fmax(cmax, fabs(d)); csum += fabs(d);
            }
          }
        }
      }
      CFZ_P(rd, 0) = csum; CFZ_P(rd, 1) = cmax; CFZ_P(rd, 2) = v0;
    CFZ_END
    CFZ_REDUCE(1, 2, 0, rd, ro);
    const double th_dyn = ro[0], cv_dyn = ro[1];
    if (rit == 0) {  // the margin the rows are restored with: bound_push, but no more than the worst violation at entry
      eps = CFZ_UNIFORM(fmin(sp.bound_push, ro[2]));
      vgoal = CFZ_UNIFORM(fmax(0.5 * eps, kRestoKappa * (ro[2] + eps)));
      dgoal = CFZ_UNIFORM(fmax(sp.constr_viol_tol, cv_dyn));  // dynamics: no worse than at entry
    }
    // ---- stage problems: H_k = (zeta + lambda) I + rho sum a a' (violated rows) + rho_b (violated boxes), g_k the gradient ----
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double ac[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // g0 g1 g2 | h0 h1 h2 h7 h8 h9
      double vmax = 0.0, bmax = 0.0, phi = 0.0;
      if (k >= 1 && k < N) {
        const double *pk = m + L.p + k * kNP;
        const double cpsi = m[L.cs + 2 * k], spsi = m[L.cs + 2 * k + 1];
        for (int jb = sub; jb < nb; jb += kLPS) {
          double a0, a1, bap[2];
          block_grad(sp, m, L, k, jb, sel_ptr(m, L)[k * nb + jb], pk[0], pk[1], cpsi, spsi, a0, a1, bap);
          for (int r_ = 0; r_ < 2; ++r_) {
            const double v = sp.dmin + eps - m[L.cj + k * nr + 2 * jb + r_];
            if (v > 0.0) {
              const double a2 = bap[r_], w = kRestoRho * v;
              vmax = fmax(vmax, v); phi += 0.5 * w * v;
              ac[0] -= w * a0; ac[1] -= w * a1; ac[2] -= w * a2;
              ac[3] += kRestoRho * a0 * a0; ac[4] += kRestoRho * a1 * a1; ac[5] += kRestoRho * a2 * a2;
              ac[6] += kRestoRho * a0 * a1; ac[7] += kRestoRho * a0 * a2; ac[8] += kRestoRho * a1 * a2;
            }
          }
        }
      }
      for (int i = 0; i < 9; ++i) CFZ_P(qx, i) = ac[i];
      CFZ_P(rd, 0) = phi; CFZ_P(rd, 1) = vmax; CFZ_P(rd, 2) = bmax;
    CFZ_MID
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double phi = CFZ_P(rd, 0), bmax = 0.0;
      if (k < N) {
        const double *pk = m + L.p + k * kNP;
        double g[kNP], h[11];
        for (int i = 0; i < kNP; ++i) { g[i] = 0.0; h[i] = zeta + lm; }
        h[7] = 0.0; h[8] = 0.0; h[9] = 0.0; h[10] = 0.0;
        for (int q = (k == 0 ? 4 : 0); q < 6; ++q) {  // the states of stage 0 are the measurement: nothing to restore there
          const int c = bcol(q);
          const double el = sp.bounds[2 * q] + kRestoBoxMargin - pk[c], eu = pk[c] - sp.bounds[2 * q + 1] + kRestoBoxMargin;
          if (el > 0.0) { g[c] -= kRestoRhoBox * el; h[c] += kRestoRhoBox; bmax = fmax(bmax, el); if (sub == 0) phi += 0.5 * kRestoRhoBox * el * el; }
          if (eu > 0.0) { g[c] += kRestoRhoBox * eu; h[c] += kRestoRhoBox; bmax = fmax(bmax, eu); if (sub == 0) phi += 0.5 * kRestoRhoBox * eu * eu; }
        }
        g[0] += CFZ_QSUM(qx, 0); g[1] += CFZ_QSUM(qx, 1); g[2] += CFZ_QSUM(qx, 2);
        h[0] += CFZ_QSUM(qx, 3); h[1] += CFZ_QSUM(qx, 4); h[2] += CFZ_QSUM(qx, 5);
        h[7] += CFZ_QSUM(qx, 6); h[8] += CFZ_QSUM(qx, 7); h[9] += CFZ_QSUM(qx, 8);
        if (sub == 0) {
          for (int i = 0; i < 11; ++i) m[L.hc + k * 11 + i] = h[i];
          for (int i = 0; i < kNP; ++i) m[L.gk + k * kNP + i] = g[i];
        }
      }
      CFZ_P(rd, 0) = phi; CFZ_P(rd, 2) = bmax;
    CFZ_END
    CFZ_REDUCE(1, 2, 0, rd, ro);
    const double phi = ro[0], vmax = ro[1], bmax = ro[2];
    if (vmax <= vgoal && bmax <= 0.5 * kRestoBoxMargin && cv_dyn <= dgoal) {
      CFZ_LANES(tid)
        const int k = tid >> kLPSBits;
        if (k < N && (tid & (kLPS - 1)) == 0)
          for (int q = 0; q < 6; ++q) {
            const int c = bcol(q);
            m[L.p + k * kNP + c] = fmin(fmax(m[L.p + k * kNP + c], sp.bounds[2 * q] + 0.5 * kRestoBoxMargin), sp.bounds[2 * q + 1] - 0.5 * kRestoBoxMargin);
          }
      CFZ_END
      return iter + 1;
    }
    // stalled: the worst violation has not dropped by a tenth in kRestoStall iterations -> a stationary point of the violation
    if (vmax <= 0.9 * vref || vmax <= vgoal) { vref = vmax; ref_it = rit; }
    if (rit - ref_it >= kRestoStall) { CFZ_SYNC(); return -(iter + 1); }
    if (rit == kRestoMaxIter || iter >= sp.max_iter) { CFZ_SYNC(); return -(iter + 1); }
    // ---- step: the solver's sweeps.  The scans return dpi = pi_new - pi: the multipliers of the equality rows are taken as zero ----
    CFZ_LANES(tid)
      if (tid < 5) m[L.pi0 + tid] = 0.0;
      for (int i = tid; i < N * 5; i += kNL) m[L.pi + i] = 0.0;
    CFZ_END
    CFZ_RICCATI(CFZ_WSP(m), N, sp.dt, L.ab, L.hc, L.gk, L.d, L.kk, L.rP);
    CFZ_WAVE0(forward_scan(CFZ_WSP(m), N, sp.dt, L.ab, L.d, L.kk, L.rP, L.p, L.dp, L.x0, L.pi0, L.dpi0));
    CFZ_WAVE0(costate_scan(CFZ_WSP(m), N, L.ab, L.hc, L.gk, L.kk, L.dp, L.dpi, L.pi));
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double dphi = 0.0, pim = 0.0;
      if (k < N && sub == 0) {
        for (int i = 0; i < kNP; ++i) dphi += m[L.gk + k * kNP + i] * m[L.dp + k * kNP + i];
        if (k + 1 < N) for (int i = 0; i < 5; ++i) pim = fmax(pim, fabs(m[L.dpi + k * 5 + i]));
        if (k == 0) for (int i = 0; i < 5; ++i) pim = fmax(pim, fabs(m[L.dpi0 + i]));
      }
      CFZ_P(rd, 0) = dphi; CFZ_P(rd, 1) = pim;
    CFZ_END
    CFZ_REDUCE(1, 1, 0, rd, ro);
    const double dphi = ro[0], pim = ro[1];
    if (eta < 1.1 * pim) eta = CFZ_UNIFORM(2.0 * pim);
    const double M0 = CFZ_UNIFORM(phi + eta * th_dyn), dM = CFZ_UNIFORM(dphi - eta * th_dyn);
    if (!(dM < -1e-10 * (1.0 + fabs(M0)))) { CFZ_SYNC(); return -(iter + 1); }  // stationary with rows still violated
    double alpha = 1.0;
    int accepted = 0;
    for (int bt = 0; bt < sp.max_backtrack; ++bt) {
      CFZ_LANES(tid)
        double th_, ph_;
        resto_partials(sp, dv, m, L, alpha, eps, tid, th_, ph_);
        CFZ_P(rd, 0) = th_; CFZ_P(rd, 1) = ph_;
      CFZ_END
      CFZ_REDUCE(2, 0, 0, rd, ro);
      const double M_t = CFZ_UNIFORM(ro[1] + eta * ro[0]);
      if (isfinite(M_t) && M_t <= M0 + kRestoArmijo * alpha * dM) { accepted = 1; break; }
      alpha = CFZ_UNIFORM(alpha * 0.5);
    }
    if (!accepted) { CFZ_SYNC(); return -(iter + 1); }
    if (alpha < 0.2) lm = fmax(4.0 * lm, 1.0); else if (alpha == 1.0) lm *= 0.25;
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits;
      if (k < N && (tid & (kLPS - 1)) == 0) for (int i = 0; i < kNP; ++i) m[L.p + k * kNP + i] += alpha * m[L.dp + k * kNP + i];
    CFZ_END
    ++iter;
  }
}

// Cold multipliers at the point in L.p (after a restoration; IPOPT resets its bound multipliers there and recomputes the others): fresh
// working set, slacks from the rows (at least half of bound_push), z = mu / distance, the equality rows at zero.
CFZ_COLD void cold_multipliers(const KSpec &sp, wsp_f64 *mw, const Lay &L, double mu) {
  double *m = (double *)mw;
  const int N = sp.N, nb = L.nb;
  CFZ_LANES(tid)
    const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
    if (k < N) {
      const double *pk = m + L.p + k * kNP;
      double sn, cn;
      sincos(pk[2], &sn, &cn);
      for (int j = sub; j < nb; j += kLPS) {
        const int t = k * nb + j;
        double A[4][2], b[4], V[4][2], sep[2];
        block_polygon(sp, m, L, k, j, A, b, V);
        sel_ptr(m, L)[t] = select_rows_sep(A, b, V, pk[0], pk[1], cn, sn, sp.g, sel_ptr(m, L)[t], sep, sp.vv_rows);
        for (int r = 0; r < 2; ++r) {
          const double sg = fmax(sep[r] - sp.dmin, 0.5 * sp.bound_push), z = mu / sg;
          m[L.sg + 2 * t + r] = sg; m[L.zs + 2 * t + r] = z; m[L.nuc + 2 * t + r] = -z;
        }
      }
      if (sub == 0) {
        for (int q = 0; q < 6; ++q) {
          m[L.zl + k * 6 + q] = mu / (pk[bcol(q)] - sp.bounds[2 * q]);
          m[L.zu + k * 6 + q] = mu / (sp.bounds[2 * q + 1] - pk[bcol(q)]);
        }
        if (k + 1 < N) for (int i = 0; i < 5; ++i) m[L.pi + k * 5 + i] = 0.0;
      }
    }
    if (tid < 5) m[L.pi0 + tid] = 0.0;
  CFZ_END
}

// ------------------------------------------------------------------------------ the solver
// x0[5], ref[3][N], nbr[n_nbr][3][N], zu[7][N] (warm start in, solution out) in global memory;
// m = this instance's workspace (LDS on the device).  out: iters,status ; cost,err,min_sep.
// dual_out (optional): l,m [N][4 n_obs], lam_ij, lam_ji [n_nbr][N][4], s [n_nbr][N][2].
struct DualOut { double *l, *mm, *lam_ij, *lam_ji, *s; unsigned long long *stamps; };

// State a converged solve leaves in global memory for the next MPC iteration of the same vehicle
// (oracle/mpc_nlp.py carry_state), in doubles: z[N][nr] | zl[N][6] | zu[N][6] | pi0[5] | pi[N][5] | mu | valid | shifted |
// working-set codes, one byte per block.
struct CarryLay { int z, zl, zu, pi0, pi, mu, valid, shifted, sel, stride; };
CFZ_FN CarryLay carry_layout(int N, int nb) {
  CarryLay c; int o = 0;
  c.z = o; o += N * 2 * nb; c.zl = o; o += N * 6; c.zu = o; o += N * 6; c.pi0 = o; o += 5; c.pi = o; o += N * 5;
  c.mu = o; o += 1; c.valid = o; o += 1; c.shifted = o; o += 1; c.sel = o; o += (N * nb + 7) / 8;
  c.stride = (o + 7) & ~7;
  return c;
}

// wst: this instance's carry record (nullptr: none kept).  carry_in != 0: start from it if it is valid
// (oracle/mpc_nlp.py warm_from_carry).  A converged solve refreshes the record, any other outcome invalidates it.
// preloaded != 0 (the persistent loop): the caller has already put the measured state (L.x0), the neighbours' poses with
// cos / sin (L.nb4) and the warm start (L.p) into the workspace and reads the solution from L.p afterwards; x0g, nbrg and zu
// are then not touched (zu may be null).
CFZ_FN void solve_instance(const KSpec &sp, const KDer &dv, const double *x0g, const double *refg, const double *nbrg, double *zu,
                           double *m, const Lay &L, int *out_i, double *out_d, const DualOut &duo, double *wst = nullptr,
                           int carry_in = 0, int preloaded = 0) {
  const int N = sp.N, nb = L.nb, nr = L.nr, n_obs = sp.n_obs, n_nbr = sp.n_nbr;
  const int m_eq = 5 + 5 * (N - 1) + nr * N, n_bnd = N * (12 + nr);
  const double mu_floor = dv.mu_floor;
  double stall_ref = 0.0;
  int stall_cnt = 0, stall_ws = 0;
  int xpar = 0;  // which half of the wavefront exchange buffer the next reduction uses
  (void)xpar;
  CFZ_PART(rd, 7);   // lane partials of the workgroup reductions
  CFZ_PART(qx, 15);  // lane shares that meet in a quad sum
  CFZ_PART(shf, 1);  // this lane has shifted its stage's curvature in some iteration (-> carry record, carry_shift)
  double ro[7];      // results of a reduction (uniform)

  // ---- load parameters, initial point ---------------------------------------------------
  CFZ_LANES(tid)
    if (!preloaded) {
      for (int t = tid; t < N * n_nbr; t += kNL) {
        const int k = t / n_nbr, o = t - k * n_nbr;
        const double po = nbrg[(o * 3 + 2) * N + k];
        double *q = m + L.nb4 + t * 4;
        q[0] = nbrg[(o * 3 + 0) * N + k]; q[1] = nbrg[(o * 3 + 1) * N + k]; q[2] = cos(po); q[3] = sin(po);
      }
      if (tid < 5) m[L.x0 + tid] = x0g[tid];
      for (int i = tid; i < N * kNP; i += kNL) { const int k = i / kNP, c = i - k * kNP; m[L.p + i] = zu[c * N + k]; }
    }
    if (tid < 5) m[L.pi0 + tid] = 0.0;
    for (int i = tid; i < N * 5; i += kNL) m[L.pi + i] = 0.0;
    CFZ_P(shf, 0) = 0.0;
  CFZ_END
  // The pose of stage 0 is pinned to the measured state: a collision row violated there by more
  // than constr_viol_tol cannot be repaired (status 4; reference: IPOPT fails, step() falls back).
  // Its nb blocks are checked by the lanes of the stage slots BEYOND the horizon (kMaxN - N slots: eight lanes at N = 30), inside the
  // passes of the first working-set selection below, where they would idle -- until round 6 the check was a selection pass and a reduction
  // of its own in front of it (2 % of a 2.5-iteration solve).  Horizons that leave fewer spare lanes than a stage has check first, as before.
  const CarryLay CL = carry_layout(N, nb);
  const int nspare = kNL - N * kLPS;
  const bool merged = nspare >= kLPS;
  double worst0 = INFINITY;
  if (!merged) {
    CFZ_LANES(tid)
      double worst = INFINITY;
      if (tid < nb) {
        double A[4][2], b[4], V[4][2], sep[2];
        block_polygon(sp, m, L, 0, tid, A, b, V);
        double s0, c0_;
        sincos(m[L.x0 + 2], &s0, &c0_);
        select_rows_sep(A, b, V, m[L.x0], m[L.x0 + 1], c0_, s0, sp.g, 0, sep, sp.vv_rows);
        worst = fmin(sep[0], sep[1]);
      }
      CFZ_P(rd, 0) = worst;
    CFZ_END
    CFZ_REDUCE(0, 0, 1, rd, ro);
    worst0 = ro[0];
  }
  // ... and so is a measured state outside the boxes on x, y, v, delta by more than constr_viol_tol (stage 0 is bounded like every
  // other stage, vehicle_follower.py:205-240, and pinned to the measurement, :194-199)
  bool x0_out = false;
  for (int q = 0; q < 4; ++q) {
    const double v = CFZ_UNIFORM(m[L.x0 + bcol(q)]);
    x0_out = x0_out || v < sp.bounds[2 * q] - sp.constr_viol_tol || v > sp.bounds[2 * q + 1] + sp.constr_viol_tol;
  }
  if (!merged && (worst0 < sp.dmin - sp.constr_viol_tol || x0_out)) {
    out_i[0] = 0; out_i[1] = 4; out_d[0] = 0.0; out_d[1] = INFINITY; out_d[2] = worst0;
    if (wst) { CFZ_LANES(tid) if (tid == 0) wst[CL.valid] = 0.0; CFZ_END }
    return;
  }
  const bool warm = wst != nullptr && carry_in != 0 && CFZ_UNIFORM(wst[CL.valid]) != 0.0;
  const double mu0 = CFZ_UNIFORM(warm ? fmin(fmax(CFZ_UNIFORM(wst[CL.mu]), mu_floor), sp.mu_init) : sp.mu_init);
  const bool shift_hint = warm && sp.carry_shift != 0 && CFZ_UNIFORM(wst[CL.shifted]) != 0.0;  // oracle/ipm.py carry_shift
  CFZ_LANES(tid)
    const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
    const bool pre = merged && k >= N;  // a lane beyond the horizon: the check of stage 0 at the measured state
    double worst = INFINITY;
    if (k < N || pre) {
      // working set and slacks from the un-pushed warm start (IPOPT: s = g(x0)), then pushed inside
      const int ks = pre ? 0 : k;
      const double x = pre ? m[L.x0] : m[L.p + k * kNP], y = pre ? m[L.x0 + 1] : m[L.p + k * kNP + 1];
      double sn, cn;
      sincos(pre ? m[L.x0 + 2] : m[L.p + k * kNP + 2], &sn, &cn);
      for (int j = pre ? tid - N * kLPS : sub; j < nb; j += pre ? nspare : kLPS) {
        const int t = ks * nb + j;
        double A[4][2], b[4], V[4][2], sep[2];
        block_polygon(sp, m, L, ks, j, A, b, V);
        const int c0 = select_rows_sep(A, b, V, x, y, cn, sn, sp.g, 0, sep, sp.vv_rows);
        if (pre) { worst = fmin(worst, fmin(sep[0], sep[1])); continue; }
        sel_ptr(m, L)[t] = c0;
        if (!warm) {
          for (int r = 0; r < 2; ++r) {
            m[L.sg + 2 * t + r] = fmax(sep[r] - sp.dmin, sp.bound_push);
            m[L.zs + 2 * t + r] = 1.0; m[L.nuc + 2 * t + r] = 0.0;
          }
        } else {
          // a row whose (face, vertex) identity exists in the carried working set of stage k+1 keeps its multiplier
          const int ko = k + 1 < N ? k + 1 : N - 1;
          const int so = reinterpret_cast<const unsigned char *>(wst + CL.sel)[ko * nb + j];
          for (int r = 0; r < 2; ++r) {
            const int vn = r == 0 ? ((c0 >> 2) & 3) : (c0 & 3);
            double z = 0.0;
            if ((so >> 4) == (c0 >> 4)) {
              if (((so >> 2) & 3) == vn) z = wst[CL.z + (ko * nb + j) * 2];
              else if ((so & 3) == vn) z = wst[CL.z + (ko * nb + j) * 2 + 1];
            }
            const double gap = sep[r] - sp.dmin;
            double sg;
            if (z > 0.0) sg = fmax(fmax(gap, mu0 / z), sp.warm_push);
            else { sg = fmax(gap, sp.bound_push); z = mu0 / sg; }
            m[L.sg + 2 * t + r] = sg; m[L.zs + 2 * t + r] = z; m[L.nuc + 2 * t + r] = -z;
          }
        }
      }
    }
    CFZ_P(rd, 0) = worst;
  CFZ_END
  if (merged) {
    CFZ_REDUCE(0, 0, 1, rd, ro);
    worst0 = ro[0];
    if (worst0 < sp.dmin - sp.constr_viol_tol || x0_out) {
      out_i[0] = 0; out_i[1] = 4; out_d[0] = 0.0; out_d[1] = INFINITY; out_d[2] = worst0;
      if (wst) { CFZ_LANES(tid) if (tid == 0) wst[CL.valid] = 0.0; CFZ_END }
      return;
    }
  }
#if defined(CFZ_NO_RESTO)  // diagnostic builds: what the restoration phase's presence costs the hot loop
  const bool resto_on = false;
#else
  const bool resto_on = sp.resto > 0 && L.nr > 0;
#endif
  CFZ_LANES(tid)
    const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
    if (k < N && sub == 0) {
      const int ko = k + 1 < N ? k + 1 : N - 1;
      for (int q = 0; q < 6; ++q) {
        const double lo = sp.bounds[2 * q], hi = sp.bounds[2 * q + 1];
        double v = m[L.p + k * kNP + bcol(q)];
        if (!warm) {
          const double pl = fmin(sp.bound_push * fmax(1.0, fabs(lo)), sp.bound_frac * (hi - lo));
          const double pu = fmin(sp.bound_push * fmax(1.0, fabs(hi)), sp.bound_frac * (hi - lo));
          v = fmax(v, lo + pl); v = fmin(v, hi - pu);
          m[L.zl + k * 6 + q] = 1.0; m[L.zu + k * 6 + q] = 1.0;
        } else {
          v = fmin(fmax(v, lo + sp.warm_push), hi - sp.warm_push);
          m[L.zl + k * 6 + q] = fmax(wst[CL.zl + ko * 6 + q], mu0 / (hi - lo));
          m[L.zu + k * 6 + q] = fmax(wst[CL.zu + ko * 6 + q], mu0 / (hi - lo));
        }
        m[L.p + k * kNP + bcol(q)] = v;
      }
      if (warm && k + 1 < N) {
        const int kp = k + 1 < N - 1 ? k + 1 : N - 2;
        for (int i = 0; i < 5; ++i) m[L.pi + k * 5 + i] = wst[CL.pi + kp * 5 + i];
      }
    }
    if (warm && tid < 5) m[L.pi0 + tid] = wst[CL.pi + tid];
  CFZ_END

  double mu = mu0, filt_mu = -1.0, theta_min = -1.0, theta_max = -1.0, err0 = INFINITY, fval_last = 0.0;
  CFZ_STAMP_DECL
  CFZ_STAMP(0);  // setup
  int nfilt = 0, status = 1, iter = 0;  // (iter is set before the loop below)
  int stagnant = 0, best_it = 0;  // the error has not halved for stag_win iterations at a feasible iterate (sticky)
  double best_err = INFINITY;

  int iter0 = 0, resto_calls = 0;
  bool first_checked = false;
  int have_dyn = 0;    // the rows phase finds heading, sensitivities and defects of its iterate in place (left by the accepted trial's evaluation)
  int sens_first = 1;  // the next line search evaluates its first trial with sensitivities: the last one accepted its first trial (or there was none)
  // The restoration phase is called from OUTSIDE the iteration loop: the loop leaves with `want_resto` set, the phase runs, the loop is
  // entered again.  (Called from inside, the two call sites cost the hot loop three times its scratch traffic: every value the
  // register allocator keeps across a call site needs a callee-saved register or a spill slot -- measured 7 % of the throughput.)
  for (;;) {
  int want_resto = 0;  // 1: before the first iteration (resto_first), 2: after a failed line search
  for (; iter <= sp.max_iter; ++iter) {
    // ---- working set refresh (iter > iter0), rows and dynamics at the current point --------------
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double cmax = 0.0, csum = 0.0, chg = 0.0, v0 = 0.0;  // chg: a block of this lane changed its working set; v0: worst row violation of stages >= 1
      if (k < N) {
        const double *pk = m + L.p + k * kNP;
        const double x = pk[0], y = pk[1];
        double sn, cn;
        // (have_dyn: this iterate is the accepted trial point of a line search that evaluated it with sensitivities -- merit_partials --:
        // the heading's cosine and sine, the sensitivities and the defects are in place)
        if (have_dyn) { cn = m[L.cs + 2 * k]; sn = m[L.cs + 2 * k + 1]; }
        else {
          sincos(pk[2], &sn, &cn);
          if (sub == 0) { m[L.cs + 2 * k] = cn; m[L.cs + 2 * k + 1] = sn; }
        }
        CFZ_STAMP(18);
        for (int j = sub; j < nb; j += kLPS) {
          CFZ_STAMP(19 + (j >= kLPS) + (j >= 2 * kLPS));
          const int t = k * nb + j;
          double sep[2];
          int c1 = sel_ptr(m, L)[t];
          if (iter > iter0) {
            const int c0 = c1;
            double A[4][2], b[4], V[4][2];
            block_polygon(sp, m, L, k, j, A, b, V);
            c1 = select_rows_sep(A, b, V, x, y, cn, sn, sp.g, c0, sep, sp.vv_rows);
            if (c1 != c0) sel_ptr(m, L)[t] = c1;
            if (c1 != c0) {
              chg = 1.0;
              // a row that keeps its (face, vertex) identity keeps slack and multipliers; a new row
              // starts at sigma = max(sep - dmin, bound_push), z = mu / sigma, nu = -z
              const int same_face = (c0 >> 4) == (c1 >> 4);
              const int ov0 = (c0 >> 2) & 3, ov1 = c0 & 3;
              const double o_sg[2] = {m[L.sg + 2 * t], m[L.sg + 2 * t + 1]};
              const double o_zs[2] = {m[L.zs + 2 * t], m[L.zs + 2 * t + 1]};
              const double o_nu[2] = {m[L.nuc + 2 * t], m[L.nuc + 2 * t + 1]};
              for (int r = 0; r < 2; ++r) {
                const int nv = r == 0 ? ((c1 >> 2) & 3) : (c1 & 3);
                const int src = same_face ? (nv == ov0 ? 0 : (nv == ov1 ? 1 : -1)) : -1;
                if (src >= 0) {
                  m[L.sg + 2 * t + r] = src == 0 ? o_sg[0] : o_sg[1];
                  m[L.zs + 2 * t + r] = src == 0 ? o_zs[0] : o_zs[1];
                  m[L.nuc + 2 * t + r] = src == 0 ? o_nu[0] : o_nu[1];
                } else {
                  const double sg = fmax(sep[r] - sp.dmin, sp.bound_push);
                  m[L.sg + 2 * t + r] = sg; m[L.zs + 2 * t + r] = mu / sg; m[L.nuc + 2 * t + r] = -mu / sg;
                }
              }
            }
          } else {
            block_sep(sp, m, L, k, j, c1, x, y, cn, sn, sep);
          }
          // (the rows of stage 0 are constants: its pose is the measurement.  Satisfied, or violated by less than constr_viol_tol --
          // the pre-check above -- they take no part in the iteration: zero residual here, zero gradient below.  As rows of the variable
          // z_0 they fought the initial-state row whenever a parked vehicle sat a centimetre inside a clearance.)
          for (int r = 0; r < 2; ++r) {
            const double c = k == 0 ? 0.0 : sep[r] - sp.dmin - m[L.sg + 2 * t + r];
            m[L.cj + 2 * t + r] = c;
            cmax = fmax(cmax, fabs(c)); csum += fabs(c);
          }
          if (k >= 1) v0 = fmax(v0, sp.dmin - fmin(sep[0], sep[1]));
        }
        CFZ_STAMP(12);  // (diagnostic) working set and rows
        if (tid == 0) for (int i = 0; i < 5; ++i) { const double r = m[L.p + i] - m[L.x0 + i]; cmax = fmax(cmax, fabs(r)); csum += fabs(r); }
        if (k + 1 < N) {
          if (have_dyn) {
            if (sub == 0)
              for (int i = 0; i < 5; ++i) {
                const double d = m[L.hc + k * 5 + i];
                m[L.d + k * 5 + i] = d; cmax = fmax(cmax, fabs(d)); csum += fabs(d);
              }
          } else {
          // the quad integrates the stage together: lane sub carries sensitivity column sub, every lane column 4
          double F[5], Sa[3], Sb[3];
          rk4_sens2(pk, pk[5], pk[6], dv.rk_h, dv.rk_hh, dv.rk_h6, dv.iwb, sp.rk_substeps, sub, F, Sa, Sb);
          if (sub < 4) for (int r = 0; r < 3; ++r) m[L.ab + k * 15 + r * 5 + sub] = Sa[r];
          if (sub == 0) {
            for (int r = 0; r < 3; ++r) m[L.ab + k * 15 + r * 5 + 4] = Sb[r];
            for (int i = 0; i < 5; ++i) {
              const double d = F[i] - m[L.p + (k + 1) * kNP + i];
              m[L.d + k * 5 + i] = d; cmax = fmax(cmax, fabs(d)); csum += fabs(d);
            }
          }
          }
        }
        CFZ_STAMP(13);  // (diagnostic) dynamics
      }
      CFZ_P(rd, 0) = csum; CFZ_P(rd, 1) = cmax; CFZ_P(rd, 2) = chg; CFZ_P(rd, 3) = v0;
    CFZ_END
    CFZ_REDUCE(1, 3, 0, rd, ro);
    have_dyn = 0;
    const double theta = ro[0], cviol = ro[1];
    const bool ws_changed = ro[2] != 0.0;
    CFZ_STAMP(1);  // working set, rows, dynamics
    // A start whose rows are violated by more than resto_first (a neighbour's prediction has moved into the path) goes through the
    // restoration phase first and starts again from the restored point with cold multipliers; a restoration that fails ends the solve
    // with status 5 (IPOPT: "converged to a point of local infeasibility")
    if (!first_checked) {
      first_checked = true;
      if (resto_on && sp.resto_first > 0.0 && ro[3] > sp.resto_first) { want_resto = 1; break; }
    }
    if (theta_min < 0.0) { theta_min = CFZ_UNIFORM(1e-4 * fmax(1.0, theta)); theta_max = CFZ_UNIFORM(1e4 * fmax(1.0, theta)); }
    // ---- dual infeasibility, multiplier sums, complementarity, objective, log terms ----------
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double r0 = 0.0, r1 = 0.0, r2 = 0.0, dinf = 0.0, snu = 0.0, sz = 0.0, c0 = 0.0, lprod = 1.0;
      double cm = 0.0;  // complementarity against the current mu: what the barrier update's first test needs (rides along in this phase's reduction)
      if (k < N) {
        const double *pk = m + L.p + k * kNP;
        const double cpsi = m[L.cs + 2 * k], spsi = m[L.cs + 2 * k + 1];
        for (int jb = sub; jb < nb; jb += kLPS) {
          double a0, a1, ap[2];
          block_grad(sp, m, L, k, jb, sel_ptr(m, L)[k * nb + jb], pk[0], pk[1], cpsi, spsi, a0, a1, ap);
          if (k == 0) { a0 = 0.0; a1 = 0.0; ap[0] = 0.0; ap[1] = 0.0; }  // the rows of stage 0 are constants
          for (int r_ = 0; r_ < 2; ++r_) {
            const int t = k * nr + 2 * jb + r_;
            const double nu = m[L.nuc + t], zs = m[L.zs + t], sg = m[L.sg + t];
            r0 += a0 * nu; r1 += a1 * nu; r2 += ap[r_] * nu;
            dinf = fmax(dinf, fabs(-nu - zs));
            snu += fabs(nu); sz += zs; c0 = fmax(c0, fabs(sg * zs)); lprod *= sg;
            cm = fmax(cm, fabs(m[L.sg + t] * m[L.zs + t] - mu));
          }
        }
      }
      CFZ_STAMP(14);  // (diagnostic) residuals: the rows
      CFZ_P(qx, 0) = r0; CFZ_P(qx, 1) = r1; CFZ_P(qx, 2) = r2;
      CFZ_P(rd, 0) = snu; CFZ_P(rd, 1) = sz; CFZ_P(rd, 3) = lprod; CFZ_P(rd, 4) = dinf; CFZ_P(rd, 5) = c0; CFZ_P(rd, 6) = cm;
    CFZ_MID
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double snu = CFZ_P(rd, 0), sz = CFZ_P(rd, 1), lprod = CFZ_P(rd, 3), dinf = CFZ_P(rd, 4), c0 = CFZ_P(rd, 5), cm = CFZ_P(rd, 6), fv = 0.0;
      if (k < N) {
        const bool first = sub == 0;  // the stage's own terms enter the sums once, through the quad's first lane
        const double *pk = m + L.p + k * kNP;
        double r[kNP];
        stage_grad(sp, dv, refg, k, pk, r);
        r[0] += CFZ_QSUM(qx, 0); r[1] += CFZ_QSUM(qx, 1); r[2] += CFZ_QSUM(qx, 2);
        if (first) fv = stage_cost(sp, refg, k, pk);
        if (tid == 0) for (int i = 0; i < 5; ++i) snu += fabs(m[L.pi0 + i]);
        if (k + 1 < N) {
          double A[5][5], B[5][2];
          load_AB(m, L, k, sp.dt, A, B);
          for (int i = 0; i < 5; ++i) {
            const double pi = m[L.pi + k * 5 + i];
            if (first) snu += fabs(pi);
            for (int q = 0; q < 5; ++q) r[q] += A[i][q] * pi;
            r[5] += B[i][0] * pi; r[6] += B[i][1] * pi;
          }
        }
        if (k == 0) for (int i = 0; i < 5; ++i) r[i] += m[L.pi0 + i];
        else for (int i = 0; i < 5; ++i) r[i] -= m[L.pi + (k - 1) * 5 + i];
        for (int q = 0; q < 6; ++q) {
          const double zl = m[L.zl + k * 6 + q], zu_ = m[L.zu + k * 6 + q];
          r[bcol(q)] += -zl + zu_;
          const double dl = pk[bcol(q)] - sp.bounds[2 * q], du = sp.bounds[2 * q + 1] - pk[bcol(q)];
          c0 = fmax(c0, fmax(fabs(dl * zl), fabs(du * zu_)));
          cm = fmax(cm, fmax(fabs((pk[bcol(q)] - sp.bounds[2 * q]) * m[L.zl + k * 6 + q] - mu), fabs((sp.bounds[2 * q + 1] - pk[bcol(q)]) * m[L.zu + k * 6 + q] - mu)));
          if (first) { sz += zl + zu_; lprod *= dl * du; }
        }
        for (int i = 0; i < kNP; ++i) dinf = fmax(dinf, fabs(r[i]));
      }
      CFZ_P(rd, 0) = snu; CFZ_P(rd, 1) = sz; CFZ_P(rd, 2) = fv; CFZ_P(rd, 3) = log(lprod); CFZ_P(rd, 4) = dinf; CFZ_P(rd, 5) = c0; CFZ_P(rd, 6) = cm;
    CFZ_END
    CFZ_REDUCE(4, 3, 0, rd, ro);
    const double sum_nu = ro[0], sum_z = ro[1], fval = ro[2], logsum = ro[3], dual_inf = ro[4], cmp0 = ro[5];
    const double cmp_mu = CFZ_UNIFORM(ro[6]);  // max |slack x multiplier - mu| at the current mu (a maximum: exact whatever the order)
    // scalars every lane computes identically go back to scalar registers: a double computed on the vector pipe lives in a
    // VGPR pair, and two dozen of them live across the whole iteration were what spilled to scratch
    const double s_d = CFZ_UNIFORM(fmax(sp.s_max, (sum_nu + sum_z) / (double)(m_eq + n_bnd)) / sp.s_max);
    const double s_c = CFZ_UNIFORM(fmax(sp.s_max, sum_z / (double)n_bnd) / sp.s_max);
    err0 = CFZ_UNIFORM(fmax(dual_inf / s_d, fmax(cviol, cmp0 / s_c)));
    CFZ_STAMP(2);  // residuals
    fval_last = fval;
    if (!isfinite(err0)) { status = 3; break; }
    if (err0 <= sp.tol && dual_inf <= sp.dual_inf_tol && cviol <= sp.constr_viol_tol && cmp0 <= sp.compl_inf_tol) { status = 0; break; }
    if (iter == sp.max_iter) { status = 1; break; }
    // infeasibility stall (oracle/ipm.py): violation stuck above the tolerance -> locally infeasible, status 5
    // (an iterate that changed the working set counts a quarter: its new rows start with their own violation -- but a solve that
    // changes it at every iterate is cycling and has to end)
    if (iter == iter0 || err0 < 0.5 * best_err) { best_err = err0; best_it = iter; }
    if (sp.stag_win > 0 && !stagnant && cviol <= sp.constr_viol_tol && iter - best_it >= sp.stag_win) stagnant = 1;
    if (sp.err_stall > 0 && iter - best_it >= sp.err_stall) { status = 5; break; }  // a cycle below constr_viol_tol: the stall test above never fires
    if (iter == iter0 || cviol <= sp.stall_kappa * stall_ref) { stall_ref = cviol; stall_cnt = 0; stall_ws = 0; }
    else if (!ws_changed) ++stall_cnt;
    else if (++stall_ws >= kWsStallDiv) { stall_ws = 0; ++stall_cnt; }
    if (sp.stall_iters > 0 && stall_cnt >= sp.stall_iters && cviol > sp.constr_viol_tol) { status = 5; break; }
    // ---- barrier update (monotone, Fiacco-McCormick) ------------------------------------------
    // (the first pass of this loop used to be a phase of its own -- a sweep over slacks, multipliers and boxes, a reduction, a barrier: 2 % of
    // an iteration; its maximum now rides along in the residuals' reduction above, and only a mu that has just been lowered is tested by a
    // sweep of its own)
    bool mu_same = true;
    while (mu > mu_floor) {
      double cmv = cmp_mu;
      if (!mu_same) {
      CFZ_LANES(tid)
        const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
        double cm = 0.0;
        if (k < N) {
          for (int jb = sub; jb < nb; jb += kLPS)
            for (int r_ = 0; r_ < 2; ++r_) { const int t = k * nr + 2 * jb + r_; cm = fmax(cm, fabs(m[L.sg + t] * m[L.zs + t] - mu)); }
          for (int q = 0; q < 6; ++q) {
            const double v = m[L.p + k * kNP + bcol(q)];
            cm = fmax(cm, fmax(fabs((v - sp.bounds[2 * q]) * m[L.zl + k * 6 + q] - mu),
                               fabs((sp.bounds[2 * q + 1] - v) * m[L.zu + k * 6 + q] - mu)));
          }
        }
        CFZ_P(rd, 0) = cm;
      CFZ_END
      CFZ_REDUCE(0, 1, 0, rd, ro);
      cmv = ro[0];
      }
      mu_same = false;
      const double emu = fmax(dual_inf / s_d, fmax(cviol, cmv / s_c));
      if (emu <= sp.kappa_eps * mu) mu = CFZ_UNIFORM(fmax(mu_floor, fmin(sp.kappa_mu * mu, pow(mu, sp.theta_mu))));
      else break;
    }
    const double tau = CFZ_UNIFORM(fmax(sp.tau_min, 1.0 - mu));
    CFZ_STAMP(3);  // barrier update
    // ---- condensed stage QP: H_k (compact), g_k ---------------------------------------------------
    // rows: g += a (S c - (mu / sigma) / D + delta_c S nu), H += S a a' with S = S0 / D, D = 1 + delta_c S0, and the curvature of the separation rows weighted with their multipliers,
    // sum_r nu_r d2 sep_r / d(x,y,psi)^2 = [[0,0,ca],[0,0,cb],[ca,cb,cc]] (oracle/mpc_nlp.py row_curvature):
    //   kind 1 (polygon face A_f = (a0,a1), body vertex b_v): d2/dpsi2 = -A_f.(R b_v)
    //   kind 2 (body face normal n = -(a0,a1), polygon vertex): d2/dx dpsi = -a1, d2/dy dpsi = a0, d2/dpsi2 = -(sep + g_f)
    //   kind 3 (distance r of two vertices, n = (a0,a1)): tau tau' / r + kappa e_psi e_psi' with tau = (t, t.dw), t = (-a1, a0)
    //           the unit tangent, dw = d(R b_v)/dpsi, kappa = -n.(R b_v); the only rows that curve x and y (cxx, cyy, cxy)
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double ac[15] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // g0 g1 g2 | h0 h1 h2 h7 h8 h9 | ca cb cc | cxx cyy cxy
      if (k < N) {
        const double *pk = m + L.p + k * kNP;
        const double cpsi = m[L.cs + 2 * k], spsi = m[L.cs + 2 * k + 1];
        for (int jb = sub; jb < nb; jb += kLPS) {
          const int sl = sel_ptr(m, L)[k * nb + jb];
          double a0, a1, bap[2];
          block_grad(sp, m, L, k, jb, sl, pk[0], pk[1], cpsi, spsi, a0, a1, bap);
          if (k == 0) { a0 = 0.0; a1 = 0.0; bap[0] = 0.0; bap[1] = 0.0; }  // the rows of stage 0 are constants
          for (int r_ = 0; r_ < 2; ++r_) {
            const int t = k * nr + 2 * jb + r_;
            // delta_c on the row (IPOPT's dual regularisation, eliminated together with the slack): the row's stiffness is
            // S0 / (1 + delta_c S0) and its right-hand side sees the distance of nu from its centred value -mu / sigma
#if defined(CFZ_NO_DELTAC)
            const double isg = 1.0 / m[L.sg + t], S = m[L.zs + t] * isg + sp.reg_primal;
            const double coef = S * m[L.cj + t] - mu * isg;
#else
            const double isg = 1.0 / m[L.sg + t], S0 = m[L.zs + t] * isg + sp.reg_primal;
            const double iD = 1.0 / (1.0 + kDeltaC(sp) * S0), S = S0 * iD;
            const double coef = S * m[L.cj + t] - mu * isg * iD + kDeltaC(sp) * S * m[L.nuc + t];
#endif
            const double a2 = bap[r_];
            ac[0] += a0 * coef; ac[1] += a1 * coef; ac[2] += a2 * coef;
            ac[3] += S * a0 * a0; ac[4] += S * a1 * a1; ac[5] += S * a2 * a2;
            ac[6] += S * a0 * a1; ac[7] += S * a0 * a2; ac[8] += S * a1 * a2;
            if (sp.row_curvature && k > 0) {
              const int f = (sl >> 4) & 3, v = r_ ? (sl & 3) : ((sl >> 2) & 3);
              const double nu = m[L.nuc + t];
              if ((sl >> 6) == 3) {
                const double bx = (v == 0 || v == 3) ? sp.g[0] : -sp.g[2], by = (v < 2) ? sp.g[1] : -sp.g[3];
                const double rbx = cpsi * bx - spsi * by, rby = spsi * bx + cpsi * by;  // R b_v; dw = (-rby, rbx)
                const double nr_ = nu / (m[L.cj + t] + sp.dmin + m[L.sg + t]);         // nu / r
                const double t2 = a1 * rby + a0 * rbx;                                  // t.dw
                ac[9] -= nr_ * a1 * t2; ac[10] += nr_ * a0 * t2; ac[11] += nr_ * t2 * t2 - nu * (a0 * rbx + a1 * rby);
                ac[12] += nr_ * a1 * a1; ac[13] += nr_ * a0 * a0; ac[14] -= nr_ * a1 * a0;
              } else if ((sl >> 6) == 1) {
                const double bx = (v == 0 || v == 3) ? sp.g[0] : -sp.g[2], by = (v < 2) ? sp.g[1] : -sp.g[3];
                ac[11] -= nu * (a0 * (cpsi * bx - spsi * by) + a1 * (spsi * bx + cpsi * by));
              } else {
                const double gf = f == 0 ? sp.g[0] : (f == 1 ? sp.g[1] : (f == 2 ? sp.g[2] : sp.g[3]));
                ac[9] -= nu * a1; ac[10] += nu * a0;
                ac[11] -= nu * (m[L.cj + t] + sp.dmin + m[L.sg + t] + gf);
              }
            }
          }
        }
      }
      CFZ_STAMP(15);  // (diagnostic) assembly: the rows
      for (int i = 0; i < 15; ++i) CFZ_P(qx, i) = ac[i];
    CFZ_MID
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      if (k < N) {
        const double *pk = m + L.p + k * kNP;
        double g[kNP], h[11];
        stage_grad(sp, dv, refg, k, pk, g);
        const double w42 = dv.w2[4];
        h[0] = dv.hb[0]; h[1] = dv.hb[1]; h[2] = dv.hb[2]; h[3] = w42 * pk[6] * pk[6] + sp.reg_primal; h[4] = dv.hb[3];
        h[5] = dv.hb[4]; h[6] = w42 * pk[3] * pk[3] + sp.reg_primal; h[7] = 0.0; h[8] = 0.0; h[9] = 0.0;
        h[10] = w42 * pk[3] * pk[6];
        for (int q = 0; q < 6; ++q) {
          const double il = 1.0 / (pk[bcol(q)] - sp.bounds[2 * q]), iu = 1.0 / (sp.bounds[2 * q + 1] - pk[bcol(q)]);
          h[bcol(q)] += m[L.zl + k * 6 + q] * il + m[L.zu + k * 6 + q] * iu;
          g[bcol(q)] += mu * (iu - il);
        }
        g[0] += CFZ_QSUM(qx, 0); g[1] += CFZ_QSUM(qx, 1); g[2] += CFZ_QSUM(qx, 2);
        h[0] += CFZ_QSUM(qx, 3); h[1] += CFZ_QSUM(qx, 4); h[2] += CFZ_QSUM(qx, 5);
        h[7] += CFZ_QSUM(qx, 6); h[8] += CFZ_QSUM(qx, 7); h[9] += CFZ_QSUM(qx, 8);
        if (sp.row_curvature) {
          const double ca = CFZ_QSUM(qx, 9), cb = CFZ_QSUM(qx, 10), cc = CFZ_QSUM(qx, 11);
          // convexity safeguard: scale by th in {1, 1/2, .., 2^-9, 0} until diag(2w) + th C keeps the margin 0.2 min(w)
          const double q2 = dv.q2;
          const double quad = ca * ca * dv.iq0 + cb * cb * dv.iq1;
          double cxx = 0.0, cyy = 0.0, cxy = 0.0;
          if (sp.vv_rows) { cxx = CFZ_QSUM(qx, 12); cyy = CFZ_QSUM(qx, 13); cxy = CFZ_QSUM(qx, 14); }
          const bool full = cxx != 0.0 || cyy != 0.0 || cxy != 0.0;  // a vertex-vertex row in this stage
          double th = 1.0;
          for (int hh = 0; hh < 11; ++hh) {
            if (hh == 10) { th = 0.0; break; }
            if (!full) {
              if (q2 + th * cc - th * th * quad >= 0.0) break;
            } else {  // diag(q) + th C positive semidefinite: leading principal minors
              const double m00 = dv.q0 + th * cxx, m11 = dv.q1 + th * cyy, m22 = q2 + th * cc, m01 = th * cxy, m02 = th * ca, m12 = th * cb;
              const double d2 = m00 * m11 - m01 * m01;
              const double d3 = m22 * d2 - (m02 * m02 * m11 - 2.0 * m02 * m12 * m01 + m12 * m12 * m00);
              if (m00 > 0.0 && d2 > 0.0 && d3 >= 0.0) break;
            }
            th *= 0.5;
          }
          if (sp.shift_after > 0 && (iter >= sp.shift_after || (stagnant && iter >= kShiftStagMin) || shift_hint) && th < 1.0) {
            CFZ_P(shf, 0) = 1.0;
            // late in a long solve the scaled model cycles: whole curvature + the smallest identity shift that keeps the margin
            const double dl = pose_shift(dv.q0 + cxx, dv.q1 + cyy, q2 + cc, cxy, ca, cb);
            h[0] += dl; h[1] += dl; h[2] += dl;
            th = 1.0;
          }
          h[2] += th * cc; h[8] += th * ca; h[9] += th * cb;
          h[0] += th * cxx; h[1] += th * cyy; h[7] += th * cxy;
        }
        if (sub == 0) {
          for (int i = 0; i < 11; ++i) m[L.hc + k * 11 + i] = h[i];
          for (int i = 0; i < kNP; ++i) m[L.gk + k * kNP + i] = g[i];
        }
      }
    CFZ_END
    CFZ_STAMP(4);  // assembly
    // ---- Riccati backward sweep, forward step, costates (lane 0, out of line) -----------------------------
    CFZ_RICCATI(CFZ_WSP(m), N, sp.dt, L.ab, L.hc, L.gk, L.d, L.kk, L.rP);
    CFZ_STAMP(11);  // Riccati backward sweep
    // forward step and costates: linear recurrences once the gains are known -> two scans by the first wavefront
    CFZ_WAVE0(forward_scan(CFZ_WSP(m), N, sp.dt, L.ab, L.d, L.kk, L.rP, L.p, L.dp, L.x0, L.pi0, L.dpi0));
    CFZ_STAMP(9);  // forward step
    CFZ_WAVE0(costate_scan(CFZ_WSP(m), N, L.ab, L.hc, L.gk, L.kk, L.dp, L.dpi, L.pi));
    CFZ_STAMP(5);  // costates
    // ---- slack step, fraction to the boundary, directional derivative ------------------------------------
    // The ratio tests keep the largest -d(.)/(.) and divide once at the end; 1/distance is formed once
    // per bound and reused (a DP division is ~12 dependent instructions on this pipe).
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      double rpri = 0.0, rdual = 0.0, dphi = 0.0;  // max of -dx/dist and -dz/z
      if (k < N) {
        const double *pk = m + L.p + k * kNP, *dpk = m + L.dp + k * kNP;
        // heading of the current iterate: still in its slots (rounds 1-5: formed again, the slots carried the value function through the
        // sweeps -- as they still do for horizons of fewer than six stages)
        double sps, cps;
        if (L.rP != L.cs) { cps = m[L.cs + 2 * k]; sps = m[L.cs + 2 * k + 1]; } else sincos(pk[2], &sps, &cps);
        for (int jb = sub; jb < nb; jb += kLPS) {
          double ba0, ba1, bap[2];
          block_grad(sp, m, L, k, jb, sel_ptr(m, L)[k * nb + jb], pk[0], pk[1], cps, sps, ba0, ba1, bap);
          if (k == 0) { ba0 = 0.0; ba1 = 0.0; bap[0] = 0.0; bap[1] = 0.0; }  // the rows of stage 0 are constants
          for (int r_ = 0; r_ < 2; ++r_) {
            const int t = k * nr + 2 * jb + r_;
            const double sg = m[L.sg + t], zs = m[L.zs + t], isg = 1.0 / sg;
#if defined(CFZ_NO_DELTAC)
            const double ds = m[L.cj + t] + ba0 * dpk[0] + ba1 * dpk[1] + bap[r_] * dpk[2];
#else
            const double ds = (m[L.cj + t] + ba0 * dpk[0] + ba1 * dpk[1] + bap[r_] * dpk[2] + kDeltaC(sp) * (mu * isg + m[L.nuc + t])) /
                              (1.0 + kDeltaC(sp) * (zs * isg + sp.reg_primal));
#endif
            m[L.dsg + t] = ds;
            const double dzs = mu * isg - zs - zs * isg * ds;
            dphi -= mu * isg * ds;
            rpri = fmax(rpri, -ds * isg);
            rdual = fmax(rdual, -dzs / zs);
          }
        }
        CFZ_STAMP(16);  // (diagnostic) step: the rows
        if (sub == 0) {
          double g[kNP];
          stage_grad(sp, dv, refg, k, pk, g);
          for (int q = 0; q < 6; ++q) {
            const double dx = dpk[bcol(q)];
            const double il = 1.0 / (pk[bcol(q)] - sp.bounds[2 * q]), iu = 1.0 / (sp.bounds[2 * q + 1] - pk[bcol(q)]);
            const double zl = m[L.zl + k * 6 + q], zu_ = m[L.zu + k * 6 + q];
            g[bcol(q)] += mu * (iu - il);
            const double dzl = mu * il - zl - zl * il * dx, dzu = mu * iu - zu_ + zu_ * iu * dx;
            rpri = fmax(rpri, fmax(-dx * il, dx * iu));
            rdual = fmax(rdual, fmax(-dzl / zl, -dzu / zu_));
          }
          for (int i = 0; i < kNP; ++i) dphi += g[i] * dpk[i];
        }
      }
      CFZ_P(rd, 0) = dphi; CFZ_P(rd, 1) = rpri; CFZ_P(rd, 2) = rdual;
    CFZ_END
    CFZ_REDUCE(1, 2, 0, rd, ro);
    const double dphi = ro[0], rp_max = ro[1], rd_max = ro[2];
    const double a_pri = CFZ_UNIFORM((rp_max > tau) ? tau / rp_max : 1.0), a_dual = CFZ_UNIFORM((rd_max > tau) ? tau / rd_max : 1.0);
    CFZ_STAMP(6);  // step
    // ---- filter line search --------------------------------------------------------------------------------
    const double phi0 = CFZ_UNIFORM(fval - mu * logsum);
    if (filt_mu != mu) { nfilt = 0; filt_mu = mu; }
    if (ws_changed) nfilt = 0;  // the entries belong to the problem with the previous working set
    double alpha = a_pri; int accepted = 0, f_type = 0;
    int acc_bt = -1;
    for (int bt = 0; bt < sp.max_backtrack; ++bt) {
      const int sens_trial = (bt == 0 && sens_first) ? 1 : 0;
      CFZ_LANES(tid)
        double th_, ph_, ll_, bad_;
        merit_partials(sp, dv, refg, m, L, alpha, tid, sens_trial, th_, ph_, ll_, bad_);
        CFZ_P(rd, 0) = th_; CFZ_P(rd, 1) = ph_; CFZ_P(rd, 2) = ll_; CFZ_P(rd, 3) = bad_;
      CFZ_END
      CFZ_REDUCE(3, 1, 0, rd, ro);
      const double th_t = ro[0], ph_t = CFZ_UNIFORM(ro[1] - mu * ro[2]);
      int ok = (ro[3] == 0.0) && isfinite(th_t) && isfinite(ph_t) && th_t <= theta_max;
      if (ok) for (int q = 0; q < nfilt; ++q) if (th_t >= m[L.filt + 2 * q] && ph_t >= m[L.filt + 2 * q + 1]) { ok = 0; break; }
      f_type = 0;
      if (ok) {
        const int sw = theta <= theta_min && dphi < 0.0 && alpha * pow(-dphi, sp.s_phi) > sp.delta_sw * pow(theta, sp.s_theta);
        if (sw) { f_type = 1; ok = ph_t <= phi0 + sp.eta_phi * alpha * dphi; }
        else ok = th_t <= (1.0 - sp.gamma_theta) * theta || ph_t <= phi0 - sp.gamma_phi * theta;
      }
      if (ok) { accepted = 1; acc_bt = bt; break; }
      alpha = CFZ_UNIFORM(alpha * 0.5);
    }
    have_dyn = (acc_bt == 0 && sens_first) ? 1 : 0;
    sens_first = acc_bt == 0 ? 1 : 0;
    CFZ_STAMP(7);  // line search
    if (!accepted) {
      // IPOPT's answer to a failed line search at an infeasible iterate: the restoration phase, then on with cold multipliers and an
      // empty filter; a restoration that fails ends the solve with status 5
      if (resto_on && resto_calls < sp.resto && cviol > sp.constr_viol_tol) { want_resto = 2; break; }
      status = 2; break;
    }
    if (!f_type) {
      // the line search above reads the filter in uniform code: a full filter is shifted below, so the other wavefront must be
      // through with its comparisons first (found as a GPU memory fault of long closed loops: the wavefront that read a half-shifted
      // entry rejected the step the other had accepted, ran one more reduction and was a barrier behind from then on)
      if (nfilt == sp.filter_cap) CFZ_SYNC();
      CFZ_LANES(tid)
        if (tid == 0) {
          int n = nfilt;
          if (n == sp.filter_cap) { for (int q = 0; q + 1 < n; ++q) { m[L.filt + 2 * q] = m[L.filt + 2 * q + 2]; m[L.filt + 2 * q + 1] = m[L.filt + 2 * q + 3]; } n--; }
          m[L.filt + 2 * n] = (1.0 - sp.gamma_theta) * theta; m[L.filt + 2 * n + 1] = phi0 - sp.gamma_phi * theta;
        }
      CFZ_END
      nfilt = (nfilt == sp.filter_cap) ? nfilt : nfilt + 1;
    }
    // ---- update ------------------------------------------------------------------------------------------------
    CFZ_LANES(tid)
      const int k = tid >> kLPSBits, sub = tid & (kLPS - 1);
      if (tid < 5) m[L.pi0 + tid] += alpha * m[L.dpi0 + tid];
      if (k < N) {
        const double ks = sp.kappa_sigma, iks = dv.iks;
        for (int jb = sub; jb < nb; jb += kLPS)
          for (int r_ = 0; r_ < 2; ++r_) {
            const int t = k * nr + 2 * jb + r_;
            const double sg = m[L.sg + t], zs = m[L.zs + t], ds = m[L.dsg + t], isg = 1.0 / sg;
            const double S = zs * isg + sp.reg_primal;
            const double dnu = S * ds - mu * isg - m[L.nuc + t];
            const double dzs = mu * isg - zs - zs * isg * ds;
            const double sgn = sg + alpha * ds, msn = mu / sgn;
            m[L.sg + t] = sgn; m[L.nuc + t] += alpha * dnu;
            m[L.zs + t] = fmin(fmax(zs + a_dual * dzs, msn * iks), ks * msn);
          }
        CFZ_STAMP(17);  // (diagnostic) update: the rows
        if (sub == 0) {
          double *pk = m + L.p + k * kNP; const double *dpk = m + L.dp + k * kNP;
          for (int q = 0; q < 6; ++q) {
            const double dx = dpk[bcol(q)];
            const double il = 1.0 / (pk[bcol(q)] - sp.bounds[2 * q]), iu = 1.0 / (sp.bounds[2 * q + 1] - pk[bcol(q)]);
            const double zl = m[L.zl + k * 6 + q], zu_ = m[L.zu + k * 6 + q];
            const double dzl = mu * il - zl - zl * il * dx, dzu = mu * iu - zu_ + zu_ * iu * dx;
            const double xn = pk[bcol(q)] + alpha * dx;
            const double mln = mu / (xn - sp.bounds[2 * q]), mun = mu / (sp.bounds[2 * q + 1] - xn);  // mu / new distance
            m[L.zl + k * 6 + q] = fmin(fmax(zl + a_dual * dzl, mln * iks), ks * mln);
            m[L.zu + k * 6 + q] = fmin(fmax(zu_ + a_dual * dzu, mun * iks), ks * mun);
          }
          for (int i = 0; i < kNP; ++i) pk[i] += alpha * dpk[i];
          if (k + 1 < N) for (int i = 0; i < 5; ++i) m[L.pi + k * 5 + i] += alpha * m[L.dpi + k * 5 + i];
        }
      }
    CFZ_END
    CFZ_STAMP(8);  // update
  }
  if (__builtin_expect(want_resto == 0, 1)) break;
  {
    const int r = restore_instance(sp, dv, CFZ_WSP(m), L, mu, iter);
    if (r < 0) { iter = -r - 1; status = iter >= sp.max_iter ? 1 : 5; break; }  // (out of iterations inside the restoration: the limit, not local infeasibility)
    cold_multipliers(sp, CFZ_WSP(m), L, mu);
    have_dyn = 0; sens_first = 1;
    if (want_resto == 1) { iter0 = r - 1; iter = iter0; }  // the first iteration again, from the restored point
    else {  // the iteration of the failed line search is counted; the filter and the stall tests start afresh
      iter = r;
      ++resto_calls;
      nfilt = 0; stall_ref = INFINITY; stall_cnt = 0; stall_ws = 0; best_err = INFINITY; best_it = r - 1;
    }
  }
  }

  CFZ_STAMP(10);
  // ---- write back: trajectory, separations, dual certificates ---------------------------------------------------
  CFZ_LANES(tid)
    if (!preloaded) for (int i = tid; i < N * kNP; i += kNL) { const int k = i / kNP, c = i - k * kNP; zu[c * N + k] = m[L.p + i]; }
    double smin = INFINITY;
    // (preloaded == 2, the persistent closed loop: neither the certificates nor the minimum separation have a reader there -- cfz_loop_get
    // returns states, predictions, status and iterations --, and the fresh selection of all N x nb blocks this loop runs for them was 4 % of
    // a 2.5-iteration solve; out_d[2] is then +inf)
    for (int t = tid; t < (preloaded == 2 ? 0 : N * nb); t += kNL) {
      const int k = t / nb, j = t - k * nb;
      double A[4][2], b[4], V[4][2], sep2[2];
      block_polygon(sp, m, L, k, j, A, b, V);
      const double psi = m[L.p + k * kNP + 2];
      double s, c;
      sincos(psi, &s, &c);
      const int c1 = select_rows_sep(A, b, V, m[L.p + k * kNP], m[L.p + k * kNP + 1], c, s, sp.g, sel_ptr(m, L)[t], sep2, sp.vv_rows);
      const double sep = fmin(sep2[0], sep2[1]);
      const int cert = (c1 >> 6) * 16 + ((c1 >> 4) & 3) * 4 + (sep2[0] <= sep2[1] ? ((c1 >> 2) & 3) : (c1 & 3));
      smin = fmin(smin, sep);
      if (duo.l) {
        const int kind = cert >> 4, f = (cert >> 2) & 3;
        double lam[4] = {0, 0, 0, 0}, muv[4] = {0, 0, 0, 0};
        // kind 3: unit vector from polygon vertex u = f to body vertex v = cert & 3 (the separating direction)
        double vn0 = 0.0, vn1 = 0.0;
        if (kind == 3) {
          const int vb_ = cert & 3;
          const double bx = (vb_ == 0 || vb_ == 3) ? sp.g[0] : -sp.g[2], by = (vb_ < 2) ? sp.g[1] : -sp.g[3];
          double ux = V[0][0], uy = V[0][1];
#pragma unroll
          for (int i = 1; i < 4; ++i) if (i == f) { ux = V[i][0]; uy = V[i][1]; }
          const double wx = m[L.p + k * kNP] + (c * bx - s * by) - ux, wy = m[L.p + k * kNP + 1] + (s * bx + c * by) - uy;
          const double ir = 1.0 / sqrt(wx * wx + wy * wy);
          vn0 = wx * ir; vn1 = wy * ir;
        }
        if (j < n_obs) {
          // (every index into A, V, lam, mu below is resolved by selects: a runtime index would put the arrays in scratch)
          if (kind == 1) {  // n = A_f ; G' mu = -R' n
            double af0 = A[0][0], af1 = A[0][1];
#pragma unroll
            for (int i = 1; i < 4; ++i) if (i == f) { af0 = A[i][0]; af1 = A[i][1]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) lam[i] = (i == f) ? 1.0 : 0.0;
            const double mx = -(c * af0 + s * af1), my = -(-s * af0 + c * af1);
            muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
          } else {  // kind 2: n = -R G_f, mu = e_f; kind 3: n = the unit vector, G' mu = -R' n; A' lam = n from the two obstacle
                    // faces through the polygon vertex (v of kind 2, u of kind 3)
            const int v = kind == 2 ? (cert & 3) : f;
            const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
            double nx = -(c * gx - s * gy), ny = -(s * gx + c * gy);
            if (kind == 2) {
#pragma unroll
              for (int i = 0; i < 4; ++i) muv[i] = (i == f) ? 1.0 : 0.0;
            } else {
              nx = vn0; ny = vn1;
              const double mx = -(c * nx + s * ny), my = -(-s * nx + c * ny);
              muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
            }
            double vx = V[0][0], vy = V[0][1];
#pragma unroll
            for (int i = 1; i < 4; ++i) if (i == v) { vx = V[i][0]; vy = V[i][1]; }
            // the two faces active at vertex v: those with |A_i.V_v - b_i| smallest
            int i0 = 0, i1 = 1; double r0 = INFINITY, r1 = INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const double r = fabs(A[i][0] * vx + A[i][1] * vy - b[i]);
              if (r < r0) { r1 = r0; i1 = i0; r0 = r; i0 = i; } else if (r < r1) { r1 = r; i1 = i; }
            }
            const int ia = i0 < i1 ? i0 : i1, ib = i0 < i1 ? i1 : i0;
            double a0x = 0, a0y = 0, a1x = 0, a1y = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { if (i == ia) { a0x = A[i][0]; a0y = A[i][1]; } if (i == ib) { a1x = A[i][0]; a1y = A[i][1]; } }
            const double det = a0x * a1y - a1x * a0y;
            const double la = fmax((a1y * nx - a1x * ny) / det, 0.0), lb = fmax((-a0y * nx + a0x * ny) / det, 0.0);
#pragma unroll
            for (int i = 0; i < 4; ++i) lam[i] = (i == ia) ? la : ((i == ib) ? lb : 0.0);
          }
          for (int i = 0; i < 4; ++i) { duo.l[k * 4 * n_obs + 4 * j + i] = lam[i]; duo.mm[k * 4 * n_obs + 4 * j + i] = muv[i]; }
        } else {
          const int o = j - n_obs;
          const double *q = m + L.nb4 + (k * n_nbr + o) * 4;
          const double co = q[2], so = q[3];
          double wx, wy;  // separating direction from this vehicle to the other
          const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
          if (kind == 3) {  // vertex against vertex: w = -n, lam = posneg(R' w), mu = posneg(-Ro' w)
            wx = -vn0; wy = -vn1;
            const double lx = c * wx + s * wy, ly = -s * wx + c * wy;
            lam[0] = fmax(lx, 0.0); lam[1] = fmax(ly, 0.0); lam[2] = fmax(-lx, 0.0); lam[3] = fmax(-ly, 0.0);
            const double mx = -(co * wx + so * wy), my = -(-so * wx + co * wy);
            muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
          } else if (kind == 1) {  // a face of the OTHER vehicle: w = -Ro G_f, mu = e_f, lam = posneg(R' w)
            wx = -(co * gx - so * gy); wy = -(so * gx + co * gy);
#pragma unroll
            for (int i = 0; i < 4; ++i) muv[i] = (i == f) ? 1.0 : 0.0;
            const double lx = c * wx + s * wy, ly = -s * wx + c * wy;
            lam[0] = fmax(lx, 0.0); lam[1] = fmax(ly, 0.0); lam[2] = fmax(-lx, 0.0); lam[3] = fmax(-ly, 0.0);
          } else {  // a face of this vehicle: w = R G_f, lam = e_f, mu = posneg(-Ro' w)
            wx = c * gx - s * gy; wy = s * gx + c * gy;
#pragma unroll
            for (int i = 0; i < 4; ++i) lam[i] = (i == f) ? 1.0 : 0.0;
            const double mx = -(co * wx + so * wy), my = -(-so * wx + co * wy);
            muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
          }
          for (int i = 0; i < 4; ++i) { duo.lam_ij[(o * N + k) * 4 + i] = lam[i]; duo.lam_ji[(o * N + k) * 4 + i] = muv[i]; }
          // s = -A_this' lam = -R G' lam   (vehicle_follower.py:350)
          const double glx = lam[0] - lam[2], gly = lam[1] - lam[3];
          duo.s[(o * N + k) * 2 + 0] = -(c * glx - s * gly);
          duo.s[(o * N + k) * 2 + 1] = -(s * glx + c * gly);
        }
      }
    }
    CFZ_P(rd, 0) = CFZ_P(shf, 0); CFZ_P(rd, 1) = smin;
  CFZ_END
  CFZ_REDUCE(0, 1, 1, rd, ro);
  const bool shifted_any = ro[0] != 0.0;
  out_d[0] = fval_last; out_d[1] = err0; out_d[2] = ro[1];
  out_i[0] = iter; out_i[1] = status;
  if (wst) {  // leave the multipliers for the next MPC iteration of this vehicle, or say that there are none
    CFZ_LANES(tid)
      if (status == 0) {
        for (int i = tid; i < N * nr; i += kNL) wst[CL.z + i] = m[L.zs + i];
        for (int i = tid; i < N * 6; i += kNL) { wst[CL.zl + i] = m[L.zl + i]; wst[CL.zu + i] = m[L.zu + i]; }
        for (int i = tid; i < N * 5; i += kNL) wst[CL.pi + i] = i < (N - 1) * 5 ? m[L.pi + i] : 0.0;
        if (tid < 5) wst[CL.pi0 + tid] = m[L.pi0 + tid];
        unsigned char *ws = reinterpret_cast<unsigned char *>(wst + CL.sel);
        for (int i = tid; i < N * nb; i += kNL) ws[i] = sel_ptr(m, L)[i];
        if (tid == 0) { wst[CL.mu] = mu; wst[CL.valid] = 1.0; wst[CL.shifted] = shifted_any ? 1.0 : 0.0; }
      } else if (tid == 0) {
        wst[CL.valid] = 0.0;
      }
    CFZ_END
  }
  CFZ_STAMP(10);  // output
#if defined(CFZ_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  // slot 0 ("setup", a handful of ticks) carries the wall time of the solve in 10 ns units (constant 100 MHz counter)
  stamp_acc[0] = wall_clock64() - stamp_wall0;
  if (duo.stamps && threadIdx.x == 0) for (int i = 0; i < 24; ++i) duo.stamps[i] = stamp_acc[i];
#endif
}

}  // namespace cfz
#endif  // CFZ_SOLVER_INL
