// cfz_engine.hip -- libconfrez_hip.so: gfx950 kernels and the C ABI of include/confrez_hip.h.
//
// Kernels
//   solve_kernel   one 64-lane workgroup (one wavefront) per MPC-step NLP; iterate, stage data
//                  and reduction scratch in LDS (cfz_solver.inl); parameters, warm start and
//                  solution are the only global-memory traffic (8*(5 + 3N + 3N n_nbr + 2*7N) B
//                  per instance).
//   loop_prep      closed loop: parameters and shifted warm start of every vehicle from the
//                  previous predictions (reference vehicle_follower.py:432-476, 636-637)
//   loop_post      closed loop: read-back or shift fallback, plant integration, clock
//                  (reference :484-563)
// Host side: a handle owns all device buffers, one stream and two events.

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/confrez_hip.h"
#include "cfz_solver.inl"
#include "cfz_plan.inl"
#include "cfz_colloc.inl"

namespace {

thread_local std::string g_err;

int fail(const char *what, hipError_t e = hipSuccess) {
  g_err = what;
  if (e != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(e); }
  return -1;
}
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(#call, e_); } while (0)

// Plant (vehicle_follower.py:528-543, CasADi integrator "idas"): RK4 with this many sub-steps per dt.  10 sub-steps are
// within 6e-11 of the converged solution over the whole input range, tighter than IDAS's default tolerances.
constexpr int kPlantSubsteps = 10;

struct DualPtrs { double *l, *m, *lam_ij, *lam_ji, *s; };

#ifndef CFZ_WAVES_PER_SIMD
#define CFZ_WAVES_PER_SIMD 2
#endif
__global__ __launch_bounds__(cfz::kNL, CFZ_WAVES_PER_SIMD) void solve_kernel(const cfz::KSpec sp, const cfz::Lay L, int B, const double *x0,
                                                   const double *ref, const double *nbr, double *zu, int32_t *status,
                                                   int32_t *iters, double *stats, DualPtrs du, const int32_t *order,
                                                   double *wst, int wst_stride, const int32_t *carry, int carry_all,
                                                   const int32_t *slots) {
  extern __shared__ double smem[];
  if ((int)blockIdx.x >= B) return;
  // workgroups are dispatched in index order: `order` puts the instances expected to run longest first
  const int b = order ? order[blockIdx.x] : (int)blockIdx.x;
  const int N = sp.N, no = sp.n_obs, nn = sp.n_nbr;
  int oi[2]; double od[3];
  cfz::DualOut duo = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#ifdef CFZ_STAMPS
  duo.stamps = reinterpret_cast<unsigned long long *>(stats) + (size_t)B * 3 + (size_t)b * 12;  // diagnostic build: stats has room
#endif
  if (du.l) {
    duo.l = du.l + (size_t)b * N * 4 * no; duo.mm = du.m + (size_t)b * N * 4 * no;
    duo.lam_ij = du.lam_ij + (size_t)b * nn * N * 4; duo.lam_ji = du.lam_ji + (size_t)b * nn * N * 4;
    duo.s = du.s + (size_t)b * nn * N * 2;
  }
  // carry record of the instance's slot (default: slot b): used when the caller says that this solve is the successor of
  // the previous one in that slot
  const int slot = slots ? slots[b] : b;
  cfz::solve_instance(sp, x0 + (size_t)b * 5, ref + (size_t)b * 3 * N, nbr + (size_t)b * nn * 3 * N,
                      zu + (size_t)b * 7 * N, smem, L, oi, od, duo, wst ? wst + (size_t)slot * wst_stride : nullptr,
                      carry_all || (carry && carry[b]));
  if (threadIdx.x == 0) {
    iters[b] = oi[0]; status[b] = oi[1];
    stats[b * 3 + 0] = od[0]; stats[b * 3 + 1] = od[1]; stats[b * 3 + 2] = od[2];
  }
}

// ---- closed loop ------------------------------------------------------------------------------
// pred[S][V][7][N] last predictions, state[S][V][5], kidx[S] reference sample index.
// One thread per (instance, stage).
__global__ void loop_prep(int S, int V, int N, int T, const double *ref_table, const int32_t *kidx,
                          const double *pred, const double *state, double *x0, double *ref, double *nbr,
                          double *zu) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)S * V * N) return;
  const int k = (int)(tid % N);
  const int b = (int)(tid / N);
  const int s = b / V, v = b - s * V;
  const int ka = (k + 1 < N) ? k + 1 : N - 1;  // _adv_onestep (:413-426)
  if (k < 5) x0[b * 5 + k] = state[b * 5 + k];
  int kr = kidx[s] + k; if (kr > T - 1) kr = T - 1;
  for (int c = 0; c < 3; ++c) ref[((size_t)b * 3 + c) * N + k] = ref_table[((size_t)v * T + kr) * 7 + c];
  for (int c = 0; c < 7; ++c) zu[((size_t)b * 7 + c) * N + k] = pred[((size_t)b * 7 + c) * N + ka];
  int o = 0;
  for (int u = 0; u < V; ++u) {
    if (u == v) continue;
    const size_t bo = (size_t)s * V + u;
    for (int c = 0; c < 3; ++c) nbr[(((size_t)b * (V - 1) + o) * 3 + c) * N + k] = pred[(bo * 7 + c) * N + ka];
    ++o;
  }
}

__global__ void advance_clock(int S, int K, int32_t *kidx) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S) kidx[s] += K;
}

// One thread per instance: accept the solution or shift the old prediction, integrate the plant.
__global__ void loop_post(int S, int V, int N, double dt, double wb, int plant_substeps, const int32_t *status,
                          const double *zu, double *pred, double *state, int32_t *kidx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S * V) return;
  double *pb = pred + (size_t)b * 7 * N;
  if (status[b] == 0) {
    for (int i = 0; i < 7 * N; ++i) pb[i] = zu[(size_t)b * 7 * N + i];
  } else {
    for (int c = 0; c < 7; ++c)
      for (int k = 0; k + 1 < N; ++k) pb[c * N + k] = pb[c * N + k + 1];
  }
  double z[5], out[5];
  for (int i = 0; i < 5; ++i) z[i] = state[b * 5 + i];
  cfz::rk4_step<false>(z, pb[5 * N], pb[6 * N], dt, wb, plant_substeps, out, nullptr);
  for (int i = 0; i < 5; ++i) state[b * 5 + i] = out[i];
  if (b % V == 0) kidx[b / V] += 1;
}

// ---- vehicle-sharded closed loop (partitioning B: a rank owns n_own vehicles of S scenarios) ------------------------
// The glue of one MPC iteration around the solve, on the caller's stream: instances are ordered [s][o] (o = index into the
// owned vehicles).  allpred[S][V][3][N] holds x, y, psi of EVERY vehicle's last prediction (gathered over RCCL by the
// caller), pred[S][n_own][7][N] this rank's own predictions, table[n_own][T][7] the owned vehicles' plans.
// vs_prep: parameters and shifted warm start (vehicle_follower.py:432-476).  One thread per (instance, stage).
__global__ void vs_prep(int S, int V, int n_own, int N, int T, const int32_t *own, const double *table, const int32_t *k0, int t,
                        const double *allpred, const double *pred, const double *state, double *x0, double *ref, double *nbr,
                        double *zu) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)S * n_own * N) return;
  const int k = (int)(tid % N);
  const int b = (int)(tid / N);
  const int s = b / n_own, o = b - s * n_own, v = own[o];
  const int ka = (k + 1 < N) ? k + 1 : N - 1;  // _adv_onestep (:413-426)
  if (k < 5) x0[b * 5 + k] = state[b * 5 + k];
  int kr = k0[s] + t + k; if (kr > T - 1) kr = T - 1;
  for (int c = 0; c < 3; ++c) ref[((size_t)b * 3 + c) * N + k] = table[((size_t)o * T + kr) * 7 + c];
  for (int c = 0; c < 7; ++c) zu[((size_t)b * 7 + c) * N + k] = pred[((size_t)b * 7 + c) * N + ka];
  int q = 0;
  for (int u = 0; u < V; ++u) {
    if (u == v) continue;
    for (int c = 0; c < 3; ++c) nbr[(((size_t)b * (V - 1) + q) * 3 + c) * N + k] = allpred[(((size_t)s * V + u) * 3 + c) * N + ka];
    ++q;
  }
}

// vs_post: read-back or shift fallback (:484-524), plant (:528-543), and the carry flag of the next iteration (a vehicle
// whose solve converged starts its next one from these multipliers).  One thread per instance.
__global__ void vs_post(int B, int N, double dt, double wb, int plant_substeps, const int32_t *status, const double *zu,
                        double *pred, double *state, int32_t *carry) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double *pb = pred + (size_t)b * 7 * N;
  if (status[b] == 0) {
    for (int i = 0; i < 7 * N; ++i) pb[i] = zu[(size_t)b * 7 * N + i];
  } else {
    for (int c = 0; c < 7; ++c)
      for (int k = 0; k + 1 < N; ++k) pb[c * N + k] = pb[c * N + k + 1];
  }
  double z[5], out[5];
  for (int i = 0; i < 5; ++i) z[i] = state[b * 5 + i];
  cfz::rk4_step<false>(z, pb[5 * N], pb[6 * N], dt, wb, plant_substeps, out, nullptr);
  for (int i = 0; i < 5; ++i) state[b * 5 + i] = out[i];
  carry[b] = status[b] == 0;
}

// Longest-processing-time-first dispatch order for the next step: instances sorted by the iteration count of
// the step just finished, descending (counting sort, one workgroup).  The solve of an instance does not depend
// on where it runs, only the makespan of the launch does.
__global__ __launch_bounds__(1024) void order_by_iters(int B, const int32_t *iters, int32_t *order) {
  __shared__ int hist[1024];
  const int t = threadIdx.x;
  hist[t] = 0;
  __syncthreads();
  for (int b = t; b < B; b += 1024) atomicAdd(&hist[1023 - min(iters[b], 1023)], 1);
  __syncthreads();
  if (t == 0) { int acc = 0; for (int i = 0; i < 1024; ++i) { const int c = hist[i]; hist[i] = acc; acc += c; } }
  __syncthreads();
  for (int b = t; b < B; b += 1024) order[atomicAdd(&hist[1023 - min(iters[b], 1023)], 1)] = b;
}

// ---- persistent closed loop -----------------------------------------------------------------------
// K MPC iterations of every scenario in ONE launch.  The Jacobi exchange only couples the V vehicles of a
// scenario, so there is no reason to stop the whole GPU after every iteration: work items (scenario, vehicle,
// iteration t) sit in a queue; a wavefront pops one, builds its parameters from the scenario's predictions of
// iteration t-1, solves, writes prediction t and the new plant state, and the last of the V vehicles to finish
// iteration t publishes the V items of t+1.  Hard instances then delay only their own scenario.
//   qbuf          [head K][tail K][slots K x B]: one ticket queue per iteration t.  tail[t] = slots reserved by
//                 publishers, head[t] = tickets handed out, slot = instance id b or -1 while not yet written.
//                 Poppers serve the LOWEST iteration that has unclaimed slots first, so a scenario that is behind
//                 never waits behind scenarios that are ahead (critical path first).  A ticket can be taken a
//                 moment before its slot is written (or, in a race for the last slots, before it is reserved):
//                 the holder polls the slot; every iteration has exactly B slots, so tickets >= B are void.
//   ctrl[0] lowest iteration whose tickets are not exhausted (monotone hint; K = all work handed out)
//   ctrl[1] items completed   ctrl[2] error flag
//   done[S]       finished vehicles of the scenario (monotone: iteration t is complete at (t+1)*V)
//   pred[2][B][7][N] double-buffered by iteration parity (read t%2, write (t+1)%2)
// Hand-offs between workgroups follow the agent-scope release/acquire recipe: payload stores, __threadfence()
// (release), device-scope atomic; consumer: atomic load of the slot, __threadfence() (acquire), payload loads.
// The pop has no lane-divergent control flow (all lanes issue the same loads; atomics add 1 from lane 0 and 0 from
// the others): a value defined under `if (lane == 0)` and broadcast afterwards was miscompiled by hipcc 7.2.
#ifdef CFZ_LOOP_TRACE
#define CFZ_MARK(c) do { if (threadIdx.x == 0) __hip_atomic_store(&ctrl[4 + blockIdx.x], (c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define CFZ_MARK(c) do { } while (0)
#endif
__global__ __launch_bounds__(cfz::kNL, CFZ_WAVES_PER_SIMD) void loop_kernel(const cfz::KSpec sp, const cfz::Lay L, int S, int V, int K, int T,
                                                        const double *ref_table, const int32_t *kidx0, int t_base,
                                                        double *pred, double *state, double *scratch, int32_t *qbuf,
                                                        int32_t *ctrl, int32_t *done, int32_t *status, int32_t *iters,
                                                        double *stats, int32_t *iter_sum, double *wst, int wst_stride, int prio_lag) {
  extern __shared__ double smem[];
  const int N = sp.N, nn = sp.n_nbr, B = S * V, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *my = scratch + (size_t)blockIdx.x * (5 + 3 * N + nn * 3 * N + 7 * N);
  double *x0 = my, *ref = x0 + 5, *nbr = ref + 3 * N, *zu = nbr + nn * 3 * N;
  int32_t *head = qbuf, *tail = qbuf + K, *slots = qbuf + 2 * K;
  // what wavefront 0 popped, for wavefront 1: {iteration t (-1: leave), instance b}.  Lives in the reduction exchange
  // area of the workspace, which is idle between two solves.
  volatile int32_t *cmd = reinterpret_cast<volatile int32_t *>(smem + L.xw);
#define CFZ_LD(p) __builtin_amdgcn_readfirstlane(__hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
  int idle = 0;
  while (true) {
    if (wave == 0) {
      // ---- pop (wavefront 0 only): lowest iteration first -----------------------------------------------
      int t = -1, b = -1;
      while (true) {
        int t0 = CFZ_LD(&ctrl[0]);
        const int hint = t0;
        while (t0 < K && CFZ_LD(&head[t0]) >= B) ++t0;
        if (t0 > hint && lane == 0) atomicMax(&ctrl[0], t0);
        if (t0 >= K) break;  // every item of every iteration has been handed out
        int idx = 0;
        for (int tt = t0; tt < K; ++tt) {
          const int hd = CFZ_LD(&head[tt]), tl = CFZ_LD(&tail[tt]);
          if (tl == 0) break;  // no scenario has reached iteration tt yet, hence none is further either
          if (hd >= tl) continue;
          idx = __builtin_amdgcn_readfirstlane(atomicAdd(&head[tt], lane == 0 ? 1 : 0));
          if (idx < B) { t = tt; break; }
        }
        if (t < 0) {  // nothing to hand out right now
          __builtin_amdgcn_s_sleep(32);
          if (++idle > (1 << 22) || CFZ_LD(&ctrl[2])) { if (lane == 0) atomicExch(&ctrl[2], 1); break; }
          continue;
        }
        idle = 0;
        for (int spins = 0; (b = CFZ_LD(&slots[(size_t)t * B + idx])) < 0; ++spins) {
          __builtin_amdgcn_s_sleep(8);
          if (spins > (1 << 23) || CFZ_LD(&ctrl[2])) break;
        }
        if (b < 0) { if (lane == 0) atomicExch(&ctrl[2], 1); t = -1; }
        break;
      }
      CFZ_MARK(1);
      // acquire: predictions / states written by other workgroups.  One agent-scope acquire by the polling wavefront
      // (invalidates this CU's L1), completed before the barrier that releases the other wavefront's loads.
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) { cmd[0] = t; cmd[1] = b; }
    }
    __syncthreads();
    const int t = cmd[0], b = cmd[1];
    if (t < 0) break;
    CFZ_MARK(2);
    // The launch ends with its slowest scenario (a chain of K dependent iterations).  A workgroup serving the oldest open
    // iteration is on that critical path: its two wavefronts take issue priority over the wavefronts they share their
    // SIMDs with (VALU issue is arbitrated by priority, then age), the others give way.
    if (prio_lag >= 0) {
      if (t <= CFZ_LD(&ctrl[0]) + prio_lag) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    }
    const int s = b / V, v = b - s * V;
    const double *pin = pred + (size_t)(t & 1) * B * 7 * N;   // predictions after iteration t-1
    double *pout = pred + (size_t)((t + 1) & 1) * B * 7 * N;
    // ---- parameters and shifted warm start (vehicle_follower.py:432-476) -------------------------------
    if (tid < 5) x0[tid] = state[b * 5 + tid];
    for (int k = tid; k < N; k += cfz::kNL) {
      const int ka = (k + 1 < N) ? k + 1 : N - 1;
      int kr = kidx0[s] + t_base + t + k; if (kr > T - 1) kr = T - 1;
      for (int c = 0; c < 3; ++c) ref[c * N + k] = ref_table[((size_t)v * T + kr) * 7 + c];
      for (int c = 0; c < 7; ++c) zu[c * N + k] = pin[((size_t)b * 7 + c) * N + ka];
      int o = 0;
      for (int u = 0; u < V; ++u) {
        if (u == v) continue;
        const size_t bo = (size_t)s * V + u;
        for (int c = 0; c < 3; ++c) nbr[(o * 3 + c) * N + k] = pin[(bo * 7 + c) * N + ka];
        ++o;
      }
    }
    __syncthreads();
    CFZ_MARK(3);
    int oi[2]; double od[3];
    cfz::DualOut duo = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    cfz::solve_instance(sp, x0, ref, nbr, zu, smem, L, oi, od, duo, wst ? wst + (size_t)b * wst_stride : nullptr, 1);
    __syncthreads();
    CFZ_MARK(4);
    // ---- read-back or shift fallback (:484-524), plant (:528-543) ------------------------------------------
    for (int i = tid; i < 7 * N; i += cfz::kNL) {
      const int c = i / N, k = i - c * N;
      const int ka = (k + 1 < N) ? k + 1 : N - 1;
      pout[(size_t)b * 7 * N + i] = (oi[1] == 0) ? zu[i] : pin[((size_t)b * 7 + c) * N + ka];
    }
    CFZ_MARK(5);
    if (tid == 0) {
      const double a0 = (oi[1] == 0) ? zu[5 * N] : pin[((size_t)b * 7 + 5) * N + 1];
      const double w0 = (oi[1] == 0) ? zu[6 * N] : pin[((size_t)b * 7 + 6) * N + 1];
      double z[5], out[5];
      for (int i = 0; i < 5; ++i) z[i] = x0[i];
      cfz::rk4_step<false>(z, a0, w0, sp.dt, sp.wb, kPlantSubsteps, out, nullptr);
      for (int i = 0; i < 5; ++i) state[b * 5 + i] = out[i];
      status[b] = oi[1]; iters[b] = oi[0];
      stats[b * 3] = od[0]; stats[b * 3 + 1] = od[1]; stats[b * 3 + 2] = od[2];
      atomicAdd(iter_sum, oi[0]);
      if (oi[1] == 0) atomicAdd(iter_sum + 1, 1);  // converged solves of this launch
    }
    // release: prediction and state of (s, v, t).  Every storing wavefront drains its stores, the workgroup meets, one
    // lane writes the XCD's L2 back and only then signals (the asm wait keeps the compiler from dropping the drain).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CFZ_MARK(6);
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int c = atomicAdd(&done[s], 1);
      if ((c % V) == V - 1 && t + 1 < K) {  // last vehicle of the scenario: publish iteration t+1
        const int pos = atomicAdd(&tail[t + 1], V);
        for (int u = 0; u < V; ++u)
          __hip_atomic_store(&slots[(size_t)(t + 1) * B + pos + u], s * V + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      atomicAdd(&ctrl[1], 1);
    }
    CFZ_MARK(8);
  }
  CFZ_MARK(9);
}

// state_ws (reference vehicle.py:99-231): one planning NLP per workgroup, workspace in global memory; see cfz_plan.inl.
// bound 512 = at most 256 VGPRs, no AGPRs: see colloc_kernel
__global__ __launch_bounds__(512) void state_ws_kernel(int B, const cfzp::PSpec *specs, const double *tube, const long long *tube_off, double *X,
                                const long long *x_off, double *slab, const long long *slab_off, int32_t *oi, double *od) {
  const int b = blockIdx.x;
  extern __shared__ double plan_win[];  // the 81 band columns the elimination is working on (cfz_plan.inl)
  if (b >= B) return;
  // all 64 lanes run the solver redundantly and share the marked loops (cfz_plan.inl)
  cfzp::solve_state_ws<true>(specs[b], tube + tube_off[b], X + x_off[b], slab + slab_off[b], oi + 2 * b, od + 3 * b, plan_win);
}

// single-vehicle collocation plan (reference vehicle.py:360-661): one NLP per workgroup, workspace in global memory; see
// cfz_colloc.inl.
// One wavefront runs it, but the bound is 512: with 64 (or 256) the register allocator may use AGPRs beyond 256 VGPRs, and
// every such build of this kernel died with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on gfx950 / ROCm 7.2 while the
// 256-VGPR builds of the same source run (measured, tools/colloc_timing_one.sh).
#ifndef CFZC_BOUNDS
#define CFZC_BOUNDS 512
#endif
// MODE 1: single-vehicle plans, one wavefront each, elimination in the LDS window; MODE 2: joint plans, 512 threads each,
// elimination from global memory.  Two kernels so that each carries one elimination only (fewer spilled registers).
template <int MODE>
__global__ __launch_bounds__(CFZC_BOUNDS) void colloc_kernel(int B, const cfzc::CSpec *specs, double *X, const long long *x_off, double *slab,
                              const long long *slab_off, const int32_t *kbs, int32_t *oi, double *od, int lds_doubles) {
  const int b = blockIdx.x;
  // dynamic LDS: MODE 1 the 103 band columns the elimination is working on, then the right-hand sides; MODE 2 one
  // right-hand side of the substitution (lds_doubles of them, 0 = none)
  if (b >= B) return;
  cfzc::solve_colloc<MODE>(specs[b], X + x_off[b], slab + slab_off[b], kbs[b], oi + 2 * b, od + cfzc::kOutD * b, lds_doubles);
}

// dual_ws (reference vehicle.py:233-296): for fixed poses, the dual certificate of every (pose, obstacle)
// pair and the separation it certifies -- closed form over the face normals of both polygons (the
// reference maximises the same separation with IPOPT over lambda, mu).  One thread per pair.
__global__ void dual_ws_kernel(const cfz::KSpec sp, int n, const double *poses, double *l, double *mu_out, double *d) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int no = sp.n_obs;
  if (tid >= n * no) return;
  const int k = tid / no, j = tid - k * no;
  double A[4][2], b[4], V[4][2];
  for (int i = 0; i < 4; ++i) {
    A[i][0] = sp.A_obs[j][i][0]; A[i][1] = sp.A_obs[j][i][1]; b[i] = sp.b_obs[j][i];
    V[i][0] = sp.V_obs[j][i][0]; V[i][1] = sp.V_obs[j][i][1];
  }
  const double x = poses[k * 3], y = poses[k * 3 + 1], psi = poses[k * 3 + 2];
  double s, c, sep2[2];
  sincos(psi, &s, &c);
  const int sel = cfz::select_rows(A, b, V, x, y, c, s, sp.g, 0);
  cfz::rows_for<false>(A, b, V, x, y, c, s, sp.g, sel, sep2, nullptr);
  const int kind = sel >> 6, f = (sel >> 4) & 3, v = sep2[0] <= sep2[1] ? (sel >> 2) & 3 : sel & 3;
  double lam[4] = {0, 0, 0, 0}, muv[4] = {0, 0, 0, 0};
  if (kind == 1) {  // n = A_f ; G' mu = -R' n
    lam[f] = 1.0;
    const double mx = -(c * A[f][0] + s * A[f][1]), my = -(-s * A[f][0] + c * A[f][1]);
    muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
  } else {  // n = -R G_f ; A' lam = n from the two obstacle faces through vertex v
    muv[f] = 1.0;
    const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
    const double nx = -(c * gx - s * gy), ny = -(s * gx + c * gy);
    int i0 = 0, i1 = 1; double r0 = INFINITY, r1 = INFINITY;
    for (int i = 0; i < 4; ++i) {
      const double r = fabs(A[i][0] * V[v][0] + A[i][1] * V[v][1] - b[i]);
      if (r < r0) { r1 = r0; i1 = i0; r0 = r; i0 = i; } else if (r < r1) { r1 = r; i1 = i; }
    }
    const int ia = i0 < i1 ? i0 : i1, ib = i0 < i1 ? i1 : i0;
    const double det = A[ia][0] * A[ib][1] - A[ib][0] * A[ia][1];
    lam[ia] = fmax((A[ib][1] * nx - A[ib][0] * ny) / det, 0.0);
    lam[ib] = fmax((-A[ia][1] * nx + A[ia][0] * ny) / det, 0.0);
  }
  for (int i = 0; i < 4; ++i) { l[(size_t)k * 4 * no + 4 * j + i] = lam[i]; mu_out[(size_t)k * 4 * no + 4 * j + i] = muv[i]; }
  if (d) d[(size_t)k * no + j] = fmin(sep2[0], sep2[1]);
}

// joint_dual_ws (reference multi_vehicle_planner.py:208-341): for n pairs of fixed poses of two vehicles, the duals
// lam (faces of the first), mu (faces of the second), s and the certified separation d of the rows
//   -b_this'lam - b_other'mu = d,  A_this'lam + s = 0,  A_other'mu - s = 0,  |s| <= 1,  lam, mu >= 0   (:292-295)
// The reference maximises d with IPOPT; here it is the closed-form maximum over the face normals of both rectangles
// (exact whenever the closest features are a face and a vertex), built like the neighbour blocks of the MPC kernel.
__global__ void joint_dual_ws_kernel(const cfz::KSpec sp, int n, const double *pa, const double *pb, double *lam_o,
                                     double *mu_o, double *s_o, double *d_o) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double x = pa[3 * k], y = pa[3 * k + 1], xo = pb[3 * k], yo = pb[3 * k + 1];
  double s, c, so, co;
  sincos(pa[3 * k + 2], &s, &c); sincos(pb[3 * k + 2], &so, &co);
  const double g0 = sp.g[0], g1 = sp.g[1], g2 = sp.g[2], g3 = sp.g[3];
  double A[4][2] = {{co, so}, {-so, co}, {-co, -so}, {so, -co}}, b[4], V[4][2];
  for (int i = 0; i < 4; ++i) b[i] = A[i][0] * xo + A[i][1] * yo + sp.g[i];
  const double BV[4][2] = {{g0, g1}, {-g2, g1}, {-g2, -g3}, {g0, -g3}};
  for (int i = 0; i < 4; ++i) { V[i][0] = xo + co * BV[i][0] - so * BV[i][1]; V[i][1] = yo + so * BV[i][0] + co * BV[i][1]; }
  double sep2[2];
  const int sel = cfz::select_rows(A, b, V, x, y, c, s, sp.g, 0);
  cfz::rows_for<false>(A, b, V, x, y, c, s, sp.g, sel, sep2, nullptr);
  const int kind = sel >> 6, f = (sel >> 4) & 3;
  const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
  double lam[4] = {0, 0, 0, 0}, mu[4] = {0, 0, 0, 0}, wx, wy;  // w: unit direction from this vehicle towards the other
  if (kind == 1) {  // a face of the OTHER vehicle separates: its outward normal points at this vehicle, w = -Ro G_f
    mu[f] = 1.0;
    wx = -(co * gx - so * gy); wy = -(so * gx + co * gy);
    const double lx = c * wx + s * wy, ly = -s * wx + c * wy;  // R' w = G' lam
    lam[0] = fmax(lx, 0.0); lam[1] = fmax(ly, 0.0); lam[2] = fmax(-lx, 0.0); lam[3] = fmax(-ly, 0.0);
  } else {  // a face of THIS vehicle separates: w = R G_f
    lam[f] = 1.0;
    wx = c * gx - s * gy; wy = s * gx + c * gy;
    const double mx = -(co * wx + so * wy), my = -(-so * wx + co * wy);  // Ro' (-w) = G' mu
    mu[0] = fmax(mx, 0.0); mu[1] = fmax(my, 0.0); mu[2] = fmax(-mx, 0.0); mu[3] = fmax(-my, 0.0);
  }
  for (int i = 0; i < 4; ++i) { lam_o[4 * k + i] = lam[i]; mu_o[4 * k + i] = mu[i]; }
  s_o[2 * k] = -wx; s_o[2 * k + 1] = -wy;  // s = -A_this' lam = A_other' mu
  if (d_o) d_o[k] = fmin(sep2[0], sep2[1]);
}

// first prediction = the planned trajectory at the horizon times, as get_current_ref seeds it
// (:397-400); state = planned state at k0 + noise
__global__ void loop_seed(int S, int V, int N, int T, const double *ref_table, const int32_t *kidx,
                          const double *noise, double *pred, double *state) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)S * V * N) return;
  const int k = (int)(tid % N);
  const int b = (int)(tid / N);
  const int s = b / V, v = b - s * V;
  int kr = kidx[s] + k; if (kr > T - 1) kr = T - 1;
  for (int c = 0; c < 7; ++c) pred[((size_t)b * 7 + c) * N + k] = ref_table[((size_t)v * T + kr) * 7 + c];
  if (k == 0)
    for (int c = 0; c < 5; ++c)
      state[b * 5 + c] = ref_table[((size_t)v * T + kidx[s]) * 7 + c] + (noise ? noise[b * 5 + c] : 0.0);
}

}  // namespace

struct cfz_handle {
  int device = 0, max_batch = 0;
  cfz::KSpec ks;
  cfz::Lay lay;
  size_t lds_bytes = 0;
  int blocks_per_cu = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = 0.f;
  double *obs_tab = nullptr;  // n_obs x 20: A[4][2], b[4], V[4][2] (KSpec::obs_tab)
  // carry records (multipliers handed from one MPC iteration to the next), one per slot; per-solve flags
  double *wst = nullptr;
  int32_t *carry = nullptr, *slots = nullptr;   // device: per-solve flags and slot ids
  int32_t *stage_host = nullptr;                // pinned staging for both (2 x max_batch): no blocking copy per solve
  hipEvent_t ev_stage = nullptr;                // the staged copy, for solves launched on a caller's stream
  int wst_stride = 0, carry_duals = 1;
  bool carry_set = false, slots_set = false, ms_pending = false;
  const int32_t *carry_ext = nullptr;           // cfz_mpc_set_carry_device: the caller's device array, for one solve
  // per-instance buffers
  double *x0 = nullptr, *ref = nullptr, *nbr = nullptr, *zu = nullptr, *stats = nullptr;
  int32_t *status = nullptr, *iters = nullptr;
  double *l = nullptr, *m = nullptr, *lam_ij = nullptr, *lam_ji = nullptr, *s = nullptr;
  // closed loop
  int S = 0, T = 0;
  double *ref_table = nullptr, *pred = nullptr, *state = nullptr;
  int32_t *kidx = nullptr, *order = nullptr;
  bool have_order = false;
  // persistent loop
  double *pred2 = nullptr, *scratch = nullptr;
  int32_t *queue = nullptr, *ctrl = nullptr, *done = nullptr, *iter_sum = nullptr;
  int queue_cap = 0, grid_blocks = 0, steps_done = 0;
  long last_iter_sum = 0, last_converged = 0;
};

namespace {

// vertices of {A p <= b} for a bounded quadrilateral
bool quad_vertices(const double A[4][2], const double b[4], double V[4][2]) {
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

int launch_solve(cfz_handle *h, int B, const double *x0, const double *ref, const double *nbr, double *zu,
                 int32_t *status, int32_t *iters, double *stats, bool duals, hipStream_t st,
                 const int32_t *order = nullptr, int carry_all = 0) {
  DualPtrs du = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (duals) du = {h->l, h->m, h->lam_ij, h->lam_ji, h->s};
  if ((h->carry_set || h->slots_set) && st != h->stream) HIP_OK(hipStreamWaitEvent(st, h->ev_stage, 0));  // staged on the handle's stream
  HIP_OK(hipEventRecord(h->ev0, st));
  hipLaunchKernelGGL(solve_kernel, dim3(B), dim3(cfz::kNL), h->lds_bytes, st, h->ks, h->lay, B, x0, ref, nbr, zu, status,
                     iters, stats, du, order, h->carry_duals ? h->wst : nullptr, h->wst_stride,
                     h->carry_ext ? h->carry_ext : (h->carry_set ? h->carry : nullptr), carry_all, h->slots_set ? h->slots : nullptr);
  h->carry_set = false; h->slots_set = false; h->carry_ext = nullptr;  // the flags of cfz_mpc_set_carry / cfz_mpc_set_slots hold for one solve
  h->ms_pending = true;
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(h->ev1, st));
  return 0;
}

int create_fill(cfz_handle *h, const cfz_spec *spec, const cfz_options *opt);

int check(cfz_handle *h, int B) {
  if (!h) return fail("null handle");
  if (B < 1 || B > h->max_batch) return fail("batch size out of range");
  HIP_OK(hipSetDevice(h->device));
  return 0;
}

}  // namespace

extern "C" {

const char *cfz_last_error(void) { return g_err.c_str(); }

void cfz_default_spec(cfz_spec *s) {
  memset(s, 0, sizeof *s);
  s->N = 30; s->n_obs = 0; s->n_nbr = 0; s->rk_substeps = 4;
  s->dt = 0.1; s->wb = 2.5; s->dmin = 0.05;
  const double g[4] = {3.3, 0.9, 0.6, 0.9};
  const double bd[12] = {2.5, 32.5, 7.5, 27.5, -2.5, 2.5, -0.85, 0.85, -1.5, 1.5, -1.0, 1.0};
  const double w[6] = {100, 100, 100, 1, 1, 1};
  memcpy(s->g, g, sizeof g); memcpy(s->bounds, bd, sizeof bd); memcpy(s->weights, w, sizeof w);
}

void cfz_default_options(cfz_options *o) {
  memset(o, 0, sizeof *o);
  o->max_iter = 600; o->max_backtrack = 25; o->filter_cap = 16;
  o->tol = 1e-2; o->constr_viol_tol = 1e-2; o->dual_inf_tol = 1.0; o->compl_inf_tol = 1e-4;
  o->mu_init = 1e-3; o->kappa_eps = 10.0; o->kappa_mu = 0.2; o->theta_mu = 1.5; o->tau_min = 0.99;
  o->bound_push = 1e-2; o->bound_frac = 1e-2; o->s_max = 100.0; o->kappa_sigma = 1e10;
  o->eta_phi = 1e-8; o->gamma_theta = 1e-5; o->gamma_phi = 1e-8; o->delta_sw = 1.0; o->s_theta = 1.1; o->s_phi = 2.3;
  o->reg_primal = 1e-8;
  o->stall_iters = 10; o->stall_kappa = 0.9; o->row_curvature = 1; o->carry_duals = 1; o->warm_push = 1e-6;
}

int cfz_create(const cfz_spec *spec, const cfz_options *opt, int device, int max_batch, cfz_handle **out) {
  if (!spec || !out) return fail("null argument");
  cfz_options od;
  if (!opt) { cfz_default_options(&od); opt = &od; }
  if (spec->N < 2 || spec->N > CFZ_MAX_N) return fail("N out of range");
  if (spec->n_obs < 0 || spec->n_obs > CFZ_MAX_OBS || spec->n_nbr < 0 || spec->n_nbr > CFZ_MAX_NBR)
    return fail("n_obs / n_nbr out of range");
  if (spec->N > cfz::kMaxN) return fail("N exceeds the four-lanes-per-stage kernel");
  if (max_batch < 1) return fail("max_batch must be positive");
  if (opt->filter_cap < 1 || opt->filter_cap > 32) return fail("filter_cap must be in 1..32");
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (ndev == 0) return fail("no HIP device: libconfrez_hip has no CPU path");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  HIP_OK(hipSetDevice(device));

  cfz_handle *h = new cfz_handle();
  h->device = device; h->max_batch = max_batch;
  if (create_fill(h, spec, opt) != 0) { cfz_destroy(h); return -1; }  // g_err is set; everything allocated so far is released
  *out = h;
  return 0;
}

}  // extern "C"

namespace {
int create_fill(cfz_handle *h, const cfz_spec *spec, const cfz_options *opt) {
  const int max_batch = h->max_batch;
  cfz::KSpec &k = h->ks;
  memset(&k, 0, sizeof k);
  k.N = spec->N; k.n_obs = spec->n_obs; k.n_nbr = spec->n_nbr; k.rk_substeps = spec->rk_substeps;
  k.max_iter = opt->max_iter; k.max_backtrack = opt->max_backtrack; k.filter_cap = opt->filter_cap;
  k.dt = spec->dt; k.wb = spec->wb; k.dmin = spec->dmin;
  memcpy(k.g, spec->g, sizeof k.g); memcpy(k.bounds, spec->bounds, sizeof k.bounds);
  memcpy(k.weights, spec->weights, sizeof k.weights);
  for (int j = 0; j < spec->n_obs; ++j) {
    memcpy(k.A_obs[j], spec->A_obs[j], sizeof k.A_obs[j]); memcpy(k.b_obs[j], spec->b_obs[j], sizeof k.b_obs[j]);
    if (!quad_vertices(spec->A_obs[j], spec->b_obs[j], k.V_obs[j])) return fail("obstacle is not a bounded quadrilateral");
  }
  k.tol = opt->tol; k.constr_viol_tol = opt->constr_viol_tol; k.dual_inf_tol = opt->dual_inf_tol;
  k.compl_inf_tol = opt->compl_inf_tol; k.mu_init = opt->mu_init; k.kappa_eps = opt->kappa_eps;
  k.kappa_mu = opt->kappa_mu; k.theta_mu = opt->theta_mu; k.tau_min = opt->tau_min; k.bound_push = opt->bound_push;
  k.bound_frac = opt->bound_frac; k.s_max = opt->s_max; k.kappa_sigma = opt->kappa_sigma; k.eta_phi = opt->eta_phi;
  k.gamma_theta = opt->gamma_theta; k.gamma_phi = opt->gamma_phi; k.delta_sw = opt->delta_sw;
  k.s_theta = opt->s_theta; k.s_phi = opt->s_phi; k.reg_primal = opt->reg_primal;
  k.stall_iters = opt->stall_iters; k.stall_kappa = opt->stall_kappa; k.row_curvature = opt->row_curvature; k.warm_push = opt->warm_push;
  h->lay = cfz::make_layout(k.N, k.n_obs + k.n_nbr, k.n_nbr);
  h->lds_bytes = (size_t)h->lay.total * sizeof(double);
  if (const char *pad = std::getenv("CFZ_LDS_PAD")) h->lds_bytes += (size_t)std::atoi(pad);  // occupancy experiments only
  if (h->lds_bytes > 160 * 1024) return fail("problem does not fit the 160 KiB LDS of one CU");
  if (h->lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
    if (e != hipSuccess) return fail("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    e = hipFuncSetAttribute((const void *)loop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
    if (e != hipSuccess) return fail("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
  }
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&h->blocks_per_cu, (const void *)solve_kernel, cfz::kNL, h->lds_bytes);
  {
    // The runtime's answer ignores that gfx950 hands out LDS in 2 KiB granules (measured with tools/src/occupancy_test.hip:
    // 54,208 B per workgroup -> the query says 3 per CU, 2 run); report what the hardware does.
    const int granules = (int)((h->lds_bytes + 2047) / 2048);
    const int by_lds = granules ? (160 * 1024 / 2048) / granules : h->blocks_per_cu;
    if (by_lds < h->blocks_per_cu) h->blocks_per_cu = by_lds;
  }
  const size_t B = (size_t)max_batch, N = (size_t)k.N, no = (size_t)k.n_obs, nn = (size_t)k.n_nbr;
  HIP_OK(hipStreamCreate(&h->stream));
  {
    std::vector<double> tab((size_t)std::max(k.n_obs, 1) * 20, 0.0);
    for (int j = 0; j < k.n_obs; ++j) {
      double *o = tab.data() + (size_t)j * 20;
      for (int i = 0; i < 4; ++i) { o[2 * i] = k.A_obs[j][i][0]; o[2 * i + 1] = k.A_obs[j][i][1]; o[8 + i] = k.b_obs[j][i];
                                    o[12 + 2 * i] = k.V_obs[j][i][0]; o[13 + 2 * i] = k.V_obs[j][i][1]; }
    }
    HIP_OK(hipMalloc(&h->obs_tab, tab.size() * 8));
    HIP_OK(hipMemcpy(h->obs_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    h->ks.obs_tab = h->obs_tab;
  }
  h->carry_duals = opt->carry_duals;
  h->wst_stride = cfz::carry_layout(k.N, k.n_obs + k.n_nbr).stride;
  HIP_OK(hipMalloc(&h->wst, (size_t)max_batch * h->wst_stride * 8));
  HIP_OK(hipMemset(h->wst, 0, (size_t)max_batch * h->wst_stride * 8));
  HIP_OK(hipMalloc(&h->carry, (size_t)max_batch * 4)); HIP_OK(hipMalloc(&h->slots, (size_t)max_batch * 4));
  HIP_OK(hipHostMalloc(&h->stage_host, (size_t)max_batch * 2 * 4, hipHostMallocDefault));
  HIP_OK(hipEventCreate(&h->ev0)); HIP_OK(hipEventCreate(&h->ev1)); HIP_OK(hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
  HIP_OK(hipMalloc(&h->x0, B * 5 * 8)); HIP_OK(hipMalloc(&h->ref, B * 3 * N * 8));
  HIP_OK(hipMalloc(&h->nbr, (B * nn * 3 * N + 1) * 8)); HIP_OK(hipMalloc(&h->zu, B * 7 * N * 8));
  // stats: 3 doubles per instance (+ 12 phase counters per instance for the -DCFZ_STAMPS diagnostic build)
  HIP_OK(hipMalloc(&h->stats, B * (3 + 12) * 8)); HIP_OK(hipMalloc(&h->status, B * 4)); HIP_OK(hipMalloc(&h->iters, B * 4));
  HIP_OK(hipMalloc(&h->l, (B * N * 4 * no + 1) * 8)); HIP_OK(hipMalloc(&h->m, (B * N * 4 * no + 1) * 8));
  HIP_OK(hipMalloc(&h->lam_ij, (B * nn * N * 4 + 1) * 8)); HIP_OK(hipMalloc(&h->lam_ji, (B * nn * N * 4 + 1) * 8));
  HIP_OK(hipMalloc(&h->s, (B * nn * N * 2 + 1) * 8));
  HIP_OK(hipMemset(h->status, 0, B * 4)); HIP_OK(hipMemset(h->iters, 0, B * 4));
  return 0;
}
}  // namespace

extern "C" {

int cfz_destroy(cfz_handle *h) {
  if (!h) return 0;
  hipSetDevice(h->device);
  void *bufs[] = {h->x0, h->ref, h->nbr, h->zu, h->stats, h->status, h->iters, h->l, h->m, h->lam_ij, h->lam_ji, h->s,
                  h->ref_table, h->pred, h->state, h->kidx, h->order, h->pred2, h->scratch, h->queue, h->ctrl, h->done,
                  h->iter_sum, h->obs_tab, h->wst, h->carry, h->slots};
  for (void *p : bufs) if (p) hipFree(p);
  if (h->stage_host) hipHostFree(h->stage_host);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->ev_stage) hipEventDestroy(h->ev_stage);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int cfz_max_batch(const cfz_handle *h) { return h ? h->max_batch : 0; }

int cfz_kernel_info(const cfz_handle *h, int32_t *lds_bytes_per_instance, int32_t *instances_per_cu) {
  if (!h) return fail("null handle");
  if (lds_bytes_per_instance) *lds_bytes_per_instance = (int32_t)h->lds_bytes;
  if (instances_per_cu) *instances_per_cu = h->blocks_per_cu;
  return 0;
}

int cfz_mpc_set_params(cfz_handle *h, int B, const double *x0, const double *ref, const double *nbr) {
  if (check(h, B)) return -1;
  if (!x0 || !ref || (h->ks.n_nbr && !nbr)) return fail("null parameter array");
  const size_t N = h->ks.N, nn = h->ks.n_nbr;
  HIP_OK(hipMemcpyAsync(h->x0, x0, (size_t)B * 5 * 8, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->ref, ref, (size_t)B * 3 * N * 8, hipMemcpyHostToDevice, h->stream));
  if (nn) HIP_OK(hipMemcpyAsync(h->nbr, nbr, (size_t)B * nn * 3 * N * 8, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

int cfz_mpc_set_warm(cfz_handle *h, int B, const double *zu) {
  if (check(h, B)) return -1;
  if (!zu) return fail("null warm start");
  HIP_OK(hipMemcpyAsync(h->zu, zu, (size_t)B * 7 * h->ks.N * 8, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

int cfz_mpc_set_carry(cfz_handle *h, int B, const int32_t *carry) {
  if (check(h, B)) return -1;
  if (!carry) { h->carry_set = false; return 0; }
  // staged through pinned memory and copied on the handle's stream, so the caller's array is free at once and no
  // device-wide synchronisation happens; the stream is drained first because the staging buffer may still be feeding
  // the previous copy (free after cfz_mpc_solve, which ends synchronised).  Device-resident loops that cannot afford
  // the drain pass their flags with cfz_mpc_set_carry_device.
  HIP_OK(hipStreamSynchronize(h->stream));
  memcpy(h->stage_host, carry, (size_t)B * 4);
  HIP_OK(hipMemcpyAsync(h->carry, h->stage_host, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipEventRecord(h->ev_stage, h->stream));
  h->carry_set = true;
  return 0;
}

int cfz_mpc_set_carry_device(cfz_handle *h, int B, const int32_t *d_carry) {
  if (check(h, B)) return -1;
  h->carry_ext = d_carry;  // read by the next solve kernel on whatever stream it is launched on; nothing is copied
  return 0;
}

int cfz_mpc_set_slots(cfz_handle *h, int B, const int32_t *slots) {
  if (check(h, B)) return -1;
  if (!slots) { h->slots_set = false; return 0; }
  for (int b = 0; b < B; ++b) if (slots[b] < 0 || slots[b] >= h->max_batch) return fail("slot index out of range");
  HIP_OK(hipStreamSynchronize(h->stream));
  int32_t *stage = h->stage_host + h->max_batch;
  memcpy(stage, slots, (size_t)B * 4);
  HIP_OK(hipMemcpyAsync(h->slots, stage, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipEventRecord(h->ev_stage, h->stream));
  h->slots_set = true;
  return 0;
}

int cfz_mpc_solve(cfz_handle *h, int B) {
  if (check(h, B)) return -1;
  if (launch_solve(h, B, h->x0, h->ref, h->nbr, h->zu, h->status, h->iters, h->stats, true, h->stream)) return -1;
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  h->ms_pending = false;
  return 0;
}

int cfz_mpc_get(cfz_handle *h, int B, double *zu, double *l, double *m, double *lam_ij, double *lam_ji, double *s) {
  if (check(h, B)) return -1;
  const size_t N = h->ks.N, no = h->ks.n_obs, nn = h->ks.n_nbr, b = (size_t)B;
  if (zu) HIP_OK(hipMemcpy(zu, h->zu, b * 7 * N * 8, hipMemcpyDeviceToHost));
  if (l && no) HIP_OK(hipMemcpy(l, h->l, b * N * 4 * no * 8, hipMemcpyDeviceToHost));
  if (m && no) HIP_OK(hipMemcpy(m, h->m, b * N * 4 * no * 8, hipMemcpyDeviceToHost));
  if (lam_ij && nn) HIP_OK(hipMemcpy(lam_ij, h->lam_ij, b * nn * N * 4 * 8, hipMemcpyDeviceToHost));
  if (lam_ji && nn) HIP_OK(hipMemcpy(lam_ji, h->lam_ji, b * nn * N * 4 * 8, hipMemcpyDeviceToHost));
  if (s && nn) HIP_OK(hipMemcpy(s, h->s, b * nn * N * 2 * 8, hipMemcpyDeviceToHost));
  return 0;
}

int cfz_mpc_stats(cfz_handle *h, int B, int32_t *status, int32_t *iters, double *cost, double *kkt_err, double *min_sep) {
  if (check(h, B)) return -1;
  if (status) HIP_OK(hipMemcpy(status, h->status, (size_t)B * 4, hipMemcpyDeviceToHost));
  if (iters) HIP_OK(hipMemcpy(iters, h->iters, (size_t)B * 4, hipMemcpyDeviceToHost));
  if (cost || kkt_err || min_sep) {
    std::vector<double> st((size_t)B * 3);
    HIP_OK(hipMemcpy(st.data(), h->stats, (size_t)B * 3 * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
      if (cost) cost[b] = st[b * 3];
      if (kkt_err) kkt_err[b] = st[b * 3 + 1];
      if (min_sep) min_sep[b] = st[b * 3 + 2];
    }
  }
  return 0;
}

double cfz_last_solve_ms(const cfz_handle *h_) {
  cfz_handle *h = const_cast<cfz_handle *>(h_);
  if (!h) return -1.0;
  if (h->ms_pending) {  // launched through cfz_mpc_solve_device: the events have not been read yet
    if (hipSetDevice(h->device) == hipSuccess && hipEventSynchronize(h->ev1) == hipSuccess)
      (void)hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1);
    h->ms_pending = false;
  }
  return (double)h->last_ms;
}

#ifdef CFZ_STAMPS
// diagnostic build only: the 12 phase counters of every instance of the last solve_kernel launch
int cfz_debug_stamps(cfz_handle *h, int B, unsigned long long *out) {
  if (check(h, B)) return -1;
  HIP_OK(hipMemcpy(out, h->stats + (size_t)B * 3, (size_t)B * 12 * 8, hipMemcpyDeviceToHost));
  return 0;
}
#endif

int cfz_mpc_solve_device(cfz_handle *h, int B, const double *d_x0, const double *d_ref, const double *d_nbr,
                         double *d_zu, int32_t *d_status, int32_t *d_iters, double *d_stats, void *stream) {
  if (check(h, B)) return -1;
  if (!d_x0 || !d_ref || !d_zu || !d_status || !d_iters || !d_stats) return fail("null device pointer");
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  return launch_solve(h, B, d_x0, d_ref, d_nbr ? d_nbr : h->nbr, d_zu, d_status, d_iters, d_stats, false, st);
}

int cfz_vsl_step(cfz_handle *h, int S, int V, int n_own, const int32_t *d_own, int T, const double *d_table, const int32_t *d_k0,
                 int t, const double *d_allpred, double *d_pred, double *d_state, int32_t *d_status, int32_t *d_iters,
                 double *d_stats, int32_t *d_carry, void *stream) {
  if (S < 1 || n_own < 1 || check(h, S * n_own)) return fail("S * n_own outside the handle's batch");
  if (V != h->ks.n_nbr + 1) return fail("V must be n_nbr + 1 of the handle's spec");
  if (!d_own || !d_table || !d_k0 || !d_allpred || !d_pred || !d_state || !d_status || !d_iters || !d_stats || !d_carry)
    return fail("null device pointer");
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const int N = h->ks.N, B = S * n_own;
  const long nt = (long)B * N;
  hipLaunchKernelGGL(vs_prep, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, S, V, n_own, N, T, d_own, d_table, d_k0, t, d_allpred,
                     d_pred, d_state, h->x0, h->ref, h->nbr, h->zu);
  HIP_OK(hipGetLastError());
  h->carry_ext = t > 0 ? d_carry : nullptr;  // iteration 0 has nothing to carry
  if (launch_solve(h, B, h->x0, h->ref, h->nbr, h->zu, d_status, d_iters, d_stats, false, st)) return -1;
  hipLaunchKernelGGL(vs_post, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, B, N, h->ks.dt, h->ks.wb, kPlantSubsteps, d_status, h->zu,
                     d_pred, d_state, d_carry);
  HIP_OK(hipGetLastError());
  return 0;
}

int cfz_state_ws(int device, int B, const cfz_plan_options *po, const int32_t *n_sets, const double *init_pose,
                 const double *final_heading, const double *tube, const double *guess, double *traj, int32_t *status,
                 int32_t *iters, double *cost) {
  if (B < 1 || !po || !n_sets || !init_pose || !tube || !traj) return fail("bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device: libconfrez_hip has no CPU path");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  HIP_OK(hipSetDevice(device));
  std::vector<cfzp::PSpec> specs(B);
  std::vector<long long> toff(B), xoff(B), soff(B);
  long long nt = 0, nx = 0, ns = 0, npts = 0;
  for (int b = 0; b < B; ++b) {
    if (n_sets[b] < 2 || po->N < 1) return fail("a plan needs at least two strategy steps");
    cfzp::PSpec &p = specs[b];
    memset(&p, 0, sizeof p);
    p.N = po->N; p.n_chk = n_sets[b] - 1; p.T = po->N * p.n_chk;
    p.has_final = final_heading && final_heading[b] == final_heading[b]; p.final_heading = p.has_final ? final_heading[b] : 0.0;
    p.bounded_input = po->bounded_input;
    p.max_iter = po->max_iter; p.max_backtrack = 25; p.filter_cap = 16; p.stall_iters = 0;
    p.dt = po->dt; p.wb = po->wb; p.shrink = po->shrink_tube;
    for (int i = 0; i < 3; ++i) p.init_pose[i] = init_pose[b * 3 + i];
    memcpy(p.bounds, po->bounds, sizeof p.bounds);
    p.tol = po->tol; p.constr_viol_tol = po->constr_viol_tol; p.dual_inf_tol = 1.0; p.compl_inf_tol = 1e-4; p.mu_init = po->mu_init;
    p.kappa_eps = 10.0; p.kappa_mu = 0.2; p.theta_mu = 1.5; p.tau_min = 0.99; p.bound_push = 1e-2; p.bound_frac = 1e-2; p.s_max = 100.0;
    p.kappa_sigma = 1e10; p.eta_phi = 1e-8; p.gamma_theta = 1e-5; p.gamma_phi = 1e-8; p.delta_sw = 1.0; p.s_theta = 1.1; p.s_phi = 2.3;
    p.reg_primal = 1e-8; p.reg_dual = 1e-9; p.curv_kappa = po->curv_kappa; p.stall_kappa = 0.9;
    toff[b] = nt; xoff[b] = nx; soff[b] = ns;
    nt += (long long)p.n_chk * 24; nx += cfzp::dims(p).n; ns += (long long)cfzp::work_doubles(p); npts += p.T + 1;
  }
  // initial guess: x, y, psi of every stage (vehicle.py:199-205), everything else zero
  std::vector<double> X((size_t)nx, 0.0);
  long long g0 = 0;
  for (int b = 0; b < B; ++b) {
    const int T = specs[b].T;
    if (guess) {
      for (int k = 0; k <= T; ++k) for (int c = 0; c < 3; ++c) X[(size_t)xoff[b] + 7 * k + c] = guess[(size_t)(g0 + k) * 3 + c];
      // The reference seeds x, y, psi only.  With v = 0 everywhere the heading rows of the linearisation have no control
      // authority (rank deficient once a terminal heading is fixed); the signed speed along the guessed path costs
      // nothing and takes the solver from 14-150 iterations (one failure) to 8-35 on the four-vehicle strategy.
      for (int k = 1; k < T; ++k) {
        const double *p0 = guess + (size_t)(g0 + k) * 3, *p1 = p0 + 3;
        const double dx = p1[0] - p0[0], dy = p1[1] - p0[1], along = dx * cos(p0[2]) + dy * sin(p0[2]);
        X[(size_t)xoff[b] + 7 * k + 3] = (along > 0.0 ? 1.0 : (along < 0.0 ? -1.0 : 0.0)) * sqrt(dx * dx + dy * dy) / po->dt;
      }
    } else for (int k = 0; k <= T; ++k) for (int c = 0; c < 3; ++c) X[(size_t)xoff[b] + 7 * k + c] = init_pose[b * 3 + c];
    g0 += T + 1;
  }
  cfzp::PSpec *dspec = nullptr; double *dtube = nullptr, *dX = nullptr, *dslab = nullptr, *dod = nullptr;
  long long *doff = nullptr; int32_t *doi = nullptr;
  HIP_OK(hipMalloc(&dspec, sizeof(cfzp::PSpec) * B)); HIP_OK(hipMalloc(&dtube, (size_t)nt * 8)); HIP_OK(hipMalloc(&dX, (size_t)nx * 8));
  HIP_OK(hipMalloc(&dslab, (size_t)ns * 8)); HIP_OK(hipMalloc(&doff, (size_t)B * 3 * 8)); HIP_OK(hipMalloc(&doi, (size_t)B * 2 * 4));
  HIP_OK(hipMalloc(&dod, (size_t)B * 3 * 8));
  HIP_OK(hipMemcpy(dspec, specs.data(), sizeof(cfzp::PSpec) * B, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dtube, tube, (size_t)nt * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dX, X.data(), (size_t)nx * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(doff, toff.data(), (size_t)B * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(doff + B, xoff.data(), (size_t)B * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(doff + 2 * B, soff.data(), (size_t)B * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(dslab, 0, (size_t)ns * 8));
  const size_t win_bytes = ((size_t)cfzp::kWinCols * cfzp::kLd + 64) * sizeof(double);  // window + one spare slot per lane (cfz_band.inl)
  HIP_OK(hipFuncSetAttribute((const void *)state_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes));
  hipLaunchKernelGGL(state_ws_kernel, dim3(B), dim3(64), win_bytes, 0, B, dspec, dtube, doff, dX, doff + B, dslab, doff + 2 * B, doi, dod);
  HIP_OK(hipGetLastError());
  HIP_OK(hipDeviceSynchronize());
  std::vector<int32_t> oi((size_t)B * 2); std::vector<double> od((size_t)B * 3);
  HIP_OK(hipMemcpy(X.data(), dX, (size_t)nx * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(oi.data(), doi, (size_t)B * 2 * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(od.data(), dod, (size_t)B * 3 * 8, hipMemcpyDeviceToHost));
  for (void *p : {(void *)dspec, (void *)dtube, (void *)dX, (void *)dslab, (void *)doff, (void *)doi, (void *)dod}) (void)hipFree(p);
  long long o = 0;
  for (int b = 0; b < B; ++b) {
    const int T = specs[b].T;
    for (int k = 0; k <= T; ++k) {
      const int ku = k < T ? k : T - 1;  // the last input is repeated (vehicle.py:226-229)
      for (int c = 0; c < 5; ++c) traj[(size_t)(o + k) * 7 + c] = X[(size_t)xoff[b] + 7 * k + c];
      for (int c = 5; c < 7; ++c) traj[(size_t)(o + k) * 7 + c] = X[(size_t)xoff[b] + 7 * ku + c];
    }
    o += T + 1;
    if (status) status[b] = oi[2 * b + 1];
    if (iters) iters[b] = oi[2 * b];
    if (cost) cost[b] = od[3 * b];
  }
  return 0;
}

// Lagrange basis on tau = [0, Radau IIA points of degree 5]: A[j][k] = l_j'(tau_k), B[j] = int_0^1 l_j (vehicle.py:54-97)
static void radau5_tables(double A[6][6], double B[6]) {
  const double tau[6] = {0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0};
  for (int j = 0; j < 6; ++j) {
    double c[7] = {1.0, 0, 0, 0, 0, 0, 0};  // coefficients of l_j, ascending powers
    int deg = 0;
    for (int m = 0; m < 6; ++m) {
      if (m == j) continue;
      const double den = tau[j] - tau[m];
      for (int q = deg + 1; q >= 0; --q) c[q] = ((q > 0 ? c[q - 1] : 0.0) - tau[m] * (q <= deg ? c[q] : 0.0)) / den;
      ++deg;
    }
    B[j] = 0.0;
    for (int q = 0; q <= deg; ++q) B[j] += c[q] / (q + 1);
    for (int k = 0; k < 6; ++k) {
      double dv = 0.0, pw = 1.0;
      for (int q = 1; q <= deg; ++q) { dv += q * c[q] * pw; pw *= tau[k]; }
      A[j][k] = dv;
    }
  }
}

// B collocation problems in one launch; problem b plans nveh[b] vehicles with one shared dt (1: the single-vehicle plan).
// Vehicles are numbered through all problems: n_sets, init_pose, final_heading, tube, guess and traj are per vehicle,
// dt0, dt, status, iters, cost per problem; pairs[b]: vehicle pairs (local indices) with a separation row, per problem.
static int colloc_run(int device, int B, const int32_t *nveh, const std::vector<std::vector<std::pair<int, int>>> &pairs, const cfz_spec *spec,
                      const cfz_colloc_options *co, const int32_t *n_sets, const double *init_pose, const double *final_heading,
                      const double *tube, const double *guess, const double *dt0, double *traj, double *dt, int32_t *status,
                      int32_t *iters, double *cost) {
  if (spec->n_obs < 0 || spec->n_obs > cfzc::kMaxObs || co->N_per_set < 1) return fail("problem size outside compiled limits");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device: libconfrez_hip has no CPU path");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  HIP_OK(hipSetDevice(device));
  std::vector<double> tab((size_t)std::max(spec->n_obs, 1) * 20, 0.0);
  for (int j = 0; j < spec->n_obs; ++j) {
    double V[4][2];
    if (!quad_vertices(spec->A_obs[j], spec->b_obs[j], V)) return fail("obstacle is not a bounded quadrilateral");
    double *o = tab.data() + (size_t)j * 20;
    for (int i = 0; i < 4; ++i) { o[2 * i] = spec->A_obs[j][i][0]; o[2 * i + 1] = spec->A_obs[j][i][1]; o[8 + i] = spec->b_obs[j][i];
                                  o[12 + 2 * i] = V[i][0]; o[13 + 2 * i] = V[i][1]; }
  }
  int nv_total = 0;
  long long nt = 0;
  for (int b = 0; b < B; ++b) {
    if (nveh[b] < 1 || nveh[b] > cfzc::kMaxVeh || (int)pairs[b].size() > cfzc::kMaxPairs) return fail("problem size outside compiled limits");
    for (int a = 0; a < nveh[b]; ++a) { if (n_sets[nv_total + a] < 2) return fail("a plan needs at least two strategy steps"); nt += (long long)(n_sets[nv_total + a] - 1) * 24; }
    nv_total += nveh[b];
  }
  double *dtab = nullptr, *dtube = nullptr;
  HIP_OK(hipMalloc(&dtab, tab.size() * 8)); HIP_OK(hipMalloc(&dtube, (size_t)nt * 8));
  HIP_OK(hipMemcpy(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dtube, tube, (size_t)nt * 8, hipMemcpyHostToDevice));
  std::vector<cfzc::CSpec> specs(B);
  std::vector<long long> xoff(B), soff(B);
  std::vector<int32_t> kbs(B);
  long long nx = 0, ns = 0, to = 0;
  int v0 = 0;
  for (int b = 0; b < B; ++b) {
    cfzc::CSpec &p = specs[b];
    memset(&p, 0, sizeof p);
    p.V = nveh[b]; p.Nps = co->N_per_set; p.n_obs = spec->n_obs; p.n_pairs = (int)pairs[b].size();
    long long npts = 0;
    for (int a = 0; a < p.V; ++a) {
      const int v = v0 + a;
      p.n_chk[a] = n_sets[v] - 1; p.N[a] = p.Nps * p.n_chk[a];
      p.has_final[a] = final_heading && final_heading[v] == final_heading[v]; p.final_heading[a] = p.has_final[a] ? final_heading[v] : 0.0;
      for (int i = 0; i < 3; ++i) p.init_pose[a][i] = init_pose[v * 3 + i];
      p.tube[a] = dtube + to; to += (long long)p.n_chk[a] * 24;
      npts += (long long)p.N[a] * cfzc::kPts;
    }
    for (int e = 0; e < p.n_pairs; ++e) {
      p.pair_a[e] = pairs[b][e].first; p.pair_b[e] = pairs[b][e].second;
      if (p.pair_a[e] < 0 || p.pair_b[e] >= p.V || p.pair_a[e] >= p.pair_b[e]) return fail("bad vehicle pair");
    }
    p.max_iter = co->max_iter; p.max_backtrack = 25; p.filter_cap = 16;
    p.wb = spec->wb; p.dmin = spec->dmin; p.shrink = co->shrink_tube; p.dt0 = dt0[b];
    memcpy(p.bounds, spec->bounds, sizeof p.bounds); memcpy(p.g, spec->g, sizeof p.g);
    radau5_tables(p.A, p.B);
    p.tol = co->tol; p.constr_viol_tol = co->constr_viol_tol; p.dual_inf_tol = 1.0; p.compl_inf_tol = 1e-4; p.mu_init = co->mu_init;
    p.kappa_eps = 10.0; p.kappa_mu = 0.2; p.theta_mu = 1.5; p.tau_min = 0.99; p.bound_push = 1e-2; p.bound_frac = 1e-2; p.s_max = 100.0;
    p.kappa_sigma = 1e10; p.eta_phi = 1e-8; p.gamma_theta = 1e-5; p.gamma_phi = 1e-8; p.delta_sw = 1.0; p.s_theta = 1.1; p.s_phi = 2.3;
    // delta_c = 1e-7 of proximal type (cfz_colloc.inl): while a vehicle stands still with its heading along an axis, the
    // six ODE rows of x (or y) of an interval only see the rank-5 derivative matrix and their multipliers are not
    // determined.  Measured on the synthetic strategy: IPOPT's form of delta_c needs 288 iterations at 1e-9, 38 at 1e-7
    // and 30 at 3e-6 for the vehicle that waits, and leaves three of the four joint test problems unconverged at any
    // value; the proximal form solves all of them in 26-38 iterations at 1e-7, where the rows are met to ~2e-4 and the
    // cost is 0.65 % below the delta_c = 1e-9 value (constr_viol_tol is 1e-2, vehicle.py:651).
    p.reg_primal = 1e-8; p.reg_dual = 1e-7; p.curv_kappa = co->curv_kappa;
    p.obs_tab = dtab;
    {  // half-bandwidth of this problem's ordering (51 for one vehicle)
      const cfzc::CDims d = cfzc::cdims(p);
      std::vector<int> pos((size_t)d.n + d.m);
      if (cfzc::build_order(p, pos.data(), pos.data() + d.n) != d.nk) return fail("internal: ordering does not cover the band system");
      kbs[b] = cfzc::half_bandwidth(p, pos.data(), pos.data() + d.n);
    }
    xoff[b] = nx; nx += 7 * npts + 1;
    soff[b] = ns; ns += (long long)cfzc::work_doubles(p, kbs[b]);
    v0 += p.V;
  }
  std::vector<double> X((size_t)nx);
  long long g0 = 0;
  for (int b = 0; b < B; ++b) {  // guess: x, y, psi, v, delta, a, w at every point (:629-636), dt0 (:388-389)
    const long long np_ = (long long)cfzc::cdims(specs[b]).np;
    memcpy(X.data() + xoff[b], guess + g0 * 7, (size_t)np_ * 7 * 8);
    X[(size_t)(xoff[b] + 7 * np_)] = dt0[b];
    g0 += np_;
  }
  cfzc::CSpec *dspec = nullptr; double *dX = nullptr, *dslab = nullptr, *dod = nullptr; long long *doff = nullptr; int32_t *doi = nullptr, *dkb = nullptr;
  HIP_OK(hipMalloc(&dspec, sizeof(cfzc::CSpec) * B)); HIP_OK(hipMalloc(&dX, (size_t)nx * 8)); HIP_OK(hipMalloc(&dslab, (size_t)ns * 8));
  HIP_OK(hipMalloc(&doff, (size_t)B * 2 * 8)); HIP_OK(hipMalloc(&doi, (size_t)B * 2 * 4)); HIP_OK(hipMalloc(&dod, (size_t)B * cfzc::kOutD * 8));
  HIP_OK(hipMalloc(&dkb, (size_t)B * 4));
  HIP_OK(hipMemcpy(dspec, specs.data(), sizeof(cfzc::CSpec) * B, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dX, X.data(), (size_t)nx * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(doff, xoff.data(), (size_t)B * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(doff + B, soff.data(), (size_t)B * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dkb, kbs.data(), (size_t)B * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(dslab, 0, (size_t)ns * 8));
  // one wavefront per single-vehicle plan (LDS-window elimination); the joint plan's band is too wide for LDS: its
  // elimination runs from global memory and the whole solver is spread over eight wavefronts to hide the latency
  bool wide = false;
  for (int b = 0; b < B; ++b) if (kbs[b] != cfzc::kCB) wide = true;
  if (std::getenv("CFZ_COLLOC_WIDE")) wide = true;  // experiments: single plans through the wide path
  if (!wide) {
    const size_t win_bytes = (size_t)cfzc::kCLdsDoubles * sizeof(double);
    HIP_OK(hipFuncSetAttribute((const void *)colloc_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes));
    hipLaunchKernelGGL(colloc_kernel<1>, dim3(B), dim3(64), win_bytes, 0, B, dspec, dX, doff, dslab, doff + B, dkb, doi, dod, (int)cfzc::kCLdsDoubles);
  } else {
    int nk_max = 0;  // the right-hand side of the largest instance in LDS if it fits beside the static arrays of the elimination
    for (int b = 0; b < B; ++b) nk_max = std::max(nk_max, cfzc::cdims(specs[b]).nk);
    const int lds_doubles = (size_t)nk_max * 8 <= 120 * 1024 ? nk_max : 0;
    if (lds_doubles) HIP_OK(hipFuncSetAttribute((const void *)colloc_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_doubles * 8));
    hipLaunchKernelGGL(colloc_kernel<2>, dim3(B), dim3(512), (size_t)lds_doubles * 8, 0, B, dspec, dX, doff, dslab, doff + B, dkb, doi, dod, lds_doubles);
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipDeviceSynchronize());
  std::vector<int32_t> oi((size_t)B * 2); std::vector<double> od((size_t)B * cfzc::kOutD);
  HIP_OK(hipMemcpy(X.data(), dX, (size_t)nx * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(oi.data(), doi, (size_t)B * 2 * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(od.data(), dod, (size_t)B * cfzc::kOutD * 8, hipMemcpyDeviceToHost));
  for (void *p : {(void *)dspec, (void *)dtab, (void *)dtube, (void *)dX, (void *)dslab, (void *)doff, (void *)doi, (void *)dod, (void *)dkb}) (void)hipFree(p);
  g0 = 0;
  for (int b = 0; b < B; ++b) {
    const long long np_ = (long long)cfzc::cdims(specs[b]).np;
    memcpy(traj + g0 * 7, X.data() + xoff[b], (size_t)np_ * 7 * 8);
    dt[b] = X[(size_t)(xoff[b] + 7 * np_)];
    g0 += np_;
    if (status) status[b] = oi[2 * b + 1];
    if (iters) iters[b] = oi[2 * b];
    if (cost) cost[b] = od[(size_t)cfzc::kOutD * b];
    if (std::getenv("CFZ_COLLOC_PROFILE")) {  // milliseconds per phase (100 MHz device clock)
      const double *t = od.data() + (size_t)cfzc::kOutD * b + 3;
      fprintf(stderr, "cfz_colloc[%d]: %d vehicle(s), half-bandwidth %d, %d iterations, evaluate %.2f assemble %.2f factor %.2f substitute %.2f line search %.2f total %.2f ms (factor: pivot+swap %.2f update %.2f refill %.2f)\n",
              b, specs[b].V, kbs[b], oi[2 * b], t[0] * 1e-5, t[1] * 1e-5, t[2] * 1e-5, t[3] * 1e-5, t[4] * 1e-5, t[5] * 1e-5, t[6] * 1e-5, t[7] * 1e-5, t[8] * 1e-5);
    }
  }
  return 0;
}

int cfz_colloc(int device, int B, const cfz_spec *spec, const cfz_colloc_options *co, const int32_t *n_sets,
               const double *init_pose, const double *final_heading, const double *tube, const double *guess,
               const double *dt0, double *traj, double *dt, int32_t *status, int32_t *iters, double *cost) {
  if (B < 1 || !spec || !co || !n_sets || !init_pose || !tube || !guess || !dt0 || !traj || !dt) return fail("bad argument");
  std::vector<int32_t> one((size_t)B, 1);
  std::vector<std::vector<std::pair<int, int>>> none((size_t)B);
  return colloc_run(device, B, one.data(), none, spec, co, n_sets, init_pose, final_heading, tube, guess, dt0, traj, dt, status, iters, cost);
}

int cfz_joint_colloc(int device, int B, int V, const cfz_spec *spec, const cfz_colloc_options *co, const int32_t *n_sets,
                     const double *init_pose, const double *final_heading, const double *tube, const double *guess, const double *dt0,
                     int n_pairs, const int32_t *pairs, double *traj, double *dt, int32_t *status, int32_t *iters, double *cost) {
  if (B < 1 || V < 1 || !spec || !co || !n_sets || !init_pose || !tube || !guess || !dt0 || !traj || !dt || n_pairs < 0) return fail("bad argument");
  std::vector<std::pair<int, int>> pr;
  if (pairs) for (int e = 0; e < n_pairs; ++e) pr.push_back({pairs[2 * e], pairs[2 * e + 1]});
  else for (int a = 0; a < V; ++a) for (int b = a + 1; b < V; ++b) pr.push_back({a, b});  // :56-58 all pairs
  std::vector<std::vector<std::pair<int, int>>> all((size_t)B, pr);
  std::vector<int32_t> nv((size_t)B, V);
  return colloc_run(device, B, nv.data(), all, spec, co, n_sets, init_pose, final_heading, tube, guess, dt0, traj, dt, status, iters, cost);
}

void cfz_default_colloc_options(cfz_colloc_options *o) {
  memset(o, 0, sizeof *o);
  o->N_per_set = 5; o->max_iter = 3000; o->shrink_tube = 0.5;
  // mu_init: IPOPT's default; 1e-3 (the MPC step's value) leaves a tail of plans that jam against a bound for 100+ iterations
  o->tol = 1e-2; o->constr_viol_tol = 1e-2; o->mu_init = 0.1; o->curv_kappa = 1e-8;
}

void cfz_default_plan_options(cfz_plan_options *o) {
  memset(o, 0, sizeof *o);
  o->N = 30; o->max_iter = 500; o->bounded_input = 0;
  o->dt = 0.1; o->wb = 2.5; o->shrink_tube = 0.5;
  const double bd[12] = {2.5, 32.5, 7.5, 27.5, -2.5, 2.5, -0.85, 0.85, -1.5, 1.5, -1.0, 1.0};
  memcpy(o->bounds, bd, sizeof bd);
  o->tol = 1e-2; o->constr_viol_tol = 1e-2; o->mu_init = 1e-3; o->curv_kappa = 1e-8;
}

int cfz_dual_ws(cfz_handle *h, int n, const double *poses, double *l, double *m, double *d) {
  if (!h) return fail("null handle");
  if (n < 1 || !poses || !l || !m) return fail("bad argument");
  HIP_OK(hipSetDevice(h->device));
  const size_t no = h->ks.n_obs;
  if (no == 0) return 0;
  double *dp = nullptr, *dl = nullptr, *dm = nullptr, *dd = nullptr;
  HIP_OK(hipMalloc(&dp, (size_t)n * 3 * 8)); HIP_OK(hipMalloc(&dl, (size_t)n * 4 * no * 8));
  HIP_OK(hipMalloc(&dm, (size_t)n * 4 * no * 8)); HIP_OK(hipMalloc(&dd, (size_t)n * no * 8));
  HIP_OK(hipMemcpy(dp, poses, (size_t)n * 3 * 8, hipMemcpyHostToDevice));
  const int nt = n * (int)no;
  hipLaunchKernelGGL(dual_ws_kernel, dim3((nt + 127) / 128), dim3(128), 0, h->stream, h->ks, n, dp, dl, dm, dd);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipMemcpy(l, dl, (size_t)n * 4 * no * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(m, dm, (size_t)n * 4 * no * 8, hipMemcpyDeviceToHost));
  if (d) HIP_OK(hipMemcpy(d, dd, (size_t)n * no * 8, hipMemcpyDeviceToHost));
  (void)hipFree(dp); (void)hipFree(dl); (void)hipFree(dm); (void)hipFree(dd);
  return 0;
}

int cfz_joint_dual_ws(cfz_handle *h, int n, const double *poses_this, const double *poses_other, double *lam, double *mu,
                      double *s, double *d) {
  if (!h) return fail("null handle");
  if (n < 1 || !poses_this || !poses_other || !lam || !mu || !s) return fail("bad argument");
  HIP_OK(hipSetDevice(h->device));
  double *dpa = nullptr, *dpb = nullptr, *dout = nullptr;
  HIP_OK(hipMalloc(&dpa, (size_t)n * 3 * 8)); HIP_OK(hipMalloc(&dpb, (size_t)n * 3 * 8)); HIP_OK(hipMalloc(&dout, (size_t)n * 11 * 8));
  HIP_OK(hipMemcpy(dpa, poses_this, (size_t)n * 3 * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dpb, poses_other, (size_t)n * 3 * 8, hipMemcpyHostToDevice));
  double *dl = dout, *dm = dout + (size_t)n * 4, *ds = dout + (size_t)n * 8, *dd = dout + (size_t)n * 10;
  hipLaunchKernelGGL(joint_dual_ws_kernel, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->ks, n, dpa, dpb, dl, dm, ds, dd);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipMemcpy(lam, dl, (size_t)n * 4 * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(mu, dm, (size_t)n * 4 * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(s, ds, (size_t)n * 2 * 8, hipMemcpyDeviceToHost));
  if (d) HIP_OK(hipMemcpy(d, dd, (size_t)n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(dpa); (void)hipFree(dpb); (void)hipFree(dout);
  return 0;
}

int cfz_loop_init(cfz_handle *h, int S, int T, const double *ref_table, const int32_t *k0, const double *noise) {
  if (!h) return fail("null handle");
  const int V = h->ks.n_nbr + 1, N = h->ks.N;
  if (S < 1 || (long)S * V > h->max_batch) return fail("S * (n_nbr+1) exceeds max_batch");
  if (T < 1 || !ref_table || !k0) return fail("bad reference table");
  HIP_OK(hipSetDevice(h->device));
  for (void *p : {(void *)h->ref_table, (void *)h->pred, (void *)h->state, (void *)h->kidx, (void *)h->order}) if (p) hipFree(p);
  h->ref_table = h->pred = h->state = nullptr; h->kidx = nullptr; h->order = nullptr; h->have_order = false;
  h->S = S; h->T = T;
  const size_t B = (size_t)S * V;
  HIP_OK(hipMalloc(&h->ref_table, (size_t)V * T * 7 * 8)); HIP_OK(hipMalloc(&h->pred, B * 7 * N * 8));
  HIP_OK(hipMalloc(&h->state, B * 5 * 8)); HIP_OK(hipMalloc(&h->kidx, (size_t)S * 4));
  HIP_OK(hipMalloc(&h->order, B * 4));
  HIP_OK(hipMemset(h->wst, 0, (size_t)h->max_batch * h->wst_stride * 8));  // first iteration: cold multipliers
  for (void *p : {(void *)h->pred2, (void *)h->scratch, (void *)h->queue, (void *)h->ctrl, (void *)h->done, (void *)h->iter_sum}) if (p) (void)hipFree(p);
  h->pred2 = h->scratch = nullptr; h->queue = h->ctrl = h->done = h->iter_sum = nullptr; h->queue_cap = 0; h->grid_blocks = 0; h->steps_done = 0;
  HIP_OK(hipMemcpy(h->ref_table, ref_table, (size_t)V * T * 7 * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->kidx, k0, (size_t)S * 4, hipMemcpyHostToDevice));
  double *dn = nullptr;
  if (noise) { HIP_OK(hipMalloc(&dn, B * 5 * 8)); HIP_OK(hipMemcpy(dn, noise, B * 5 * 8, hipMemcpyHostToDevice)); }
  const long nt = (long)B * N;
  hipLaunchKernelGGL(loop_seed, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->stream, S, V, N, T, h->ref_table,
                     h->kidx, dn, h->pred, h->state);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
  if (dn) hipFree(dn);
  return 0;
}

int cfz_loop_step(cfz_handle *h) {
  if (!h || !h->pred) return fail("cfz_loop_init has not been called");
  HIP_OK(hipSetDevice(h->device));
  const int V = h->ks.n_nbr + 1, N = h->ks.N, S = h->S, B = S * V;
  const long nt = (long)B * N;
  hipLaunchKernelGGL(loop_prep, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->stream, S, V, N, h->T, h->ref_table,
                     h->kidx, h->pred, h->state, h->x0, h->ref, h->nbr, h->zu);
  HIP_OK(hipGetLastError());
  if (launch_solve(h, B, h->x0, h->ref, h->nbr, h->zu, h->status, h->iters, h->stats, false, h->stream,
                   h->have_order ? h->order : nullptr, 1)) return -1;
  hipLaunchKernelGGL(order_by_iters, dim3(1), dim3(1024), 0, h->stream, B, h->iters, h->order);
  HIP_OK(hipGetLastError());
  h->have_order = true;
  hipLaunchKernelGGL(loop_post, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream, S, V, N, h->ks.dt, h->ks.wb, kPlantSubsteps,
                     h->status, h->zu, h->pred, h->state, h->kidx);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  h->ms_pending = false;
  return 0;
}

int cfz_loop_run(cfz_handle *h, int K) {
  if (!h || !h->pred) return fail("cfz_loop_init has not been called");
  if (K < 1) return fail("K must be positive");
  HIP_OK(hipSetDevice(h->device));
  const int V = h->ks.n_nbr + 1, N = h->ks.N, S = h->S, B = S * V;
  const size_t total = (size_t)B * K;
  if (total > (size_t)1 << 30) return fail("too many work items");
  int ncu = 0;
  HIP_OK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device));
  int per_cu = 0;
  HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)loop_kernel, cfz::kNL, h->lds_bytes));
  per_cu = std::min(per_cu, h->blocks_per_cu);  // the 2 KiB LDS granules (cfz_create): what the hardware really keeps resident
  if (per_cu < 1) return fail("loop kernel does not fit on a CU");
  // one workgroup per resident slot: more would only queue behind them (any workgroup can serve any item, so a surplus
  // is harmless, just useless)
  if (const char *cap = std::getenv("CFZ_LOOP_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(per_cu, std::atoi(cap)));  // experiments
  const int grid = std::min(B, per_cu * ncu);
  const size_t per_block = 5 + 3 * (size_t)N + (size_t)h->ks.n_nbr * 3 * N + 7 * (size_t)N;
  if (!h->pred2) {
    HIP_OK(hipMalloc(&h->pred2, (size_t)2 * B * 7 * N * 8)); HIP_OK(hipMalloc(&h->ctrl, (4 + 1024) * 4));
    HIP_OK(hipMalloc(&h->done, (size_t)S * 4)); HIP_OK(hipMalloc(&h->iter_sum, 8));
  }
  if (h->grid_blocks < grid) {
    if (h->scratch) (void)hipFree(h->scratch);
    HIP_OK(hipMalloc(&h->scratch, (size_t)grid * per_block * 8)); h->grid_blocks = grid;
  }
  const size_t qwords = 2 * (size_t)K + total;  // [head K][tail K][slots K x B]
  if ((size_t)h->queue_cap < qwords) {
    if (h->queue) (void)hipFree(h->queue);
    HIP_OK(hipMalloc(&h->queue, qwords * 4)); h->queue_cap = (int)qwords;
  }
  // parity 0 of the double buffer <- current predictions; queue <- all items of iteration 0
  HIP_OK(hipMemcpyAsync(h->pred2, h->pred, (size_t)B * 7 * N * 8, hipMemcpyDeviceToDevice, h->stream));
  HIP_OK(hipMemsetAsync(h->queue, 0, 2 * (size_t)K * 4, h->stream));
  HIP_OK(hipMemsetAsync(h->queue + 2 * K, 0xff, total * 4, h->stream));
  {
    std::vector<int32_t> first(B);
    for (int b = 0; b < B; ++b) first[b] = b;
    HIP_OK(hipMemcpyAsync(h->queue + 2 * K, first.data(), (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipMemcpyAsync(h->queue + K, &B, 4, hipMemcpyHostToDevice, h->stream));  // tail[0] = B
    const int32_t ctrl0[4] = {0, 0, 0, 0};
    HIP_OK(hipMemcpyAsync(h->ctrl, ctrl0, sizeof ctrl0, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipMemsetAsync(h->done, 0, (size_t)S * 4, h->stream));
    HIP_OK(hipMemsetAsync(h->iter_sum, 0, 8, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));  // `first` and `ctrl0` are host temporaries
  }
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(loop_kernel, dim3(grid), dim3(cfz::kNL), h->lds_bytes, h->stream, h->ks, h->lay, S, V, K, h->T,
                     h->ref_table, h->kidx, 0, h->pred2, h->state, h->scratch, h->queue, h->ctrl, h->done, h->status,
                     h->iters, h->stats, h->iter_sum, h->carry_duals ? h->wst : nullptr, h->wst_stride,
                     std::getenv("CFZ_LOOP_PRIO_LAG") ? std::atoi(std::getenv("CFZ_LOOP_PRIO_LAG")) : 0);
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  // predictions after K iterations live in parity K%2; advance the scenario clocks by K
  HIP_OK(hipMemcpyAsync(h->pred, h->pred2 + (size_t)(K & 1) * B * 7 * N, (size_t)B * 7 * N * 8, hipMemcpyDeviceToDevice, h->stream));
  hipLaunchKernelGGL(advance_clock, dim3((S + 255) / 256), dim3(256), 0, h->stream, S, K, h->kidx);
  HIP_OK(hipGetLastError());
  if (const char *dbg = std::getenv("CFZ_LOOP_WATCHDOG")) {
    // diagnostic: watch the queue counters from a second stream while the kernel runs; stop it if it stalls
    const double limit_s = std::atof(dbg) > 0 ? std::atof(dbg) : 10.0;
    hipStream_t s2; HIP_OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int32_t last_head = -1; double stalled = 0.0;
    while (hipEventQuery(h->ev1) == hipErrorNotReady) {
      usleep(100000);
      int32_t c[4] = {0, 0, 0, 0};
      HIP_OK(hipMemcpyAsync(c, h->ctrl, sizeof c, hipMemcpyDeviceToHost, s2)); HIP_OK(hipStreamSynchronize(s2));
      std::fprintf(stderr, "[cfz watchdog] lowest open iteration %d, completed %d of %zu, err %d (grid %d)", c[0], c[1], total, c[2], grid);
#ifdef CFZ_LOOP_TRACE
      int32_t mk[8];
      HIP_OK(hipMemcpyAsync(mk, h->ctrl + 4, sizeof mk, hipMemcpyDeviceToHost, s2)); HIP_OK(hipStreamSynchronize(s2));
      for (int i = 0; i < 8 && i < grid; ++i) std::fprintf(stderr, " m%d=%d", i, mk[i]);
#endif
      std::fprintf(stderr, "\n");
      stalled = (c[1] == last_head) ? stalled + 0.1 : 0.0; last_head = c[1];
      if (stalled > limit_s) {
        const int32_t one = 1;
        HIP_OK(hipMemcpyAsync(h->ctrl + 2, &one, 4, hipMemcpyHostToDevice, s2)); HIP_OK(hipStreamSynchronize(s2));
        stalled = -1e9;
      }
    }
    (void)hipStreamDestroy(s2);
  }
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  h->ms_pending = false;
  int32_t ctrl[4] = {0, 0, 0, 0}, isum[2] = {0, 0};
  HIP_OK(hipMemcpy(ctrl, h->ctrl, sizeof ctrl, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(isum, h->iter_sum, 8, hipMemcpyDeviceToHost));
  h->last_iter_sum = isum[0]; h->last_converged = isum[1];
  h->have_order = false;
  if (ctrl[2]) return fail("persistent loop kernel timed out waiting for a work item");
  return 0;
}

long cfz_loop_last_iterations(const cfz_handle *h) { return h ? h->last_iter_sum : -1; }
long cfz_loop_last_converged(const cfz_handle *h) { return h ? h->last_converged : -1; }

int cfz_loop_get(cfz_handle *h, double *state, double *pred, int32_t *status, int32_t *iters) {
  if (!h || !h->pred) return fail("cfz_loop_init has not been called");
  HIP_OK(hipSetDevice(h->device));
  const size_t B = (size_t)h->S * (h->ks.n_nbr + 1), N = h->ks.N;
  if (state) HIP_OK(hipMemcpy(state, h->state, B * 5 * 8, hipMemcpyDeviceToHost));
  if (pred) HIP_OK(hipMemcpy(pred, h->pred, B * 7 * N * 8, hipMemcpyDeviceToHost));
  if (status) HIP_OK(hipMemcpy(status, h->status, B * 4, hipMemcpyDeviceToHost));
  if (iters) HIP_OK(hipMemcpy(iters, h->iters, B * 4, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
