// cfz_engine.hip -- libconfrez_hip.so: gfx950 kernels and the C ABI of include/confrez_hip.h.
//
// Kernels
//   solve_kernel   one 128-lane workgroup (two wavefronts; four lanes of a DPP quad per stage) per MPC-step NLP; iterate,
//                  stage data and reduction scratch in LDS (cfz_solver.inl: 40,952 B per instance, four instances per CU);
//                  parameters, warm start and solution are the only algorithmic global-memory traffic
//                  (8*(5 + 3N + 3N n_nbr + 2*7N) B per instance) beside the carry record of the slot (8.8 KB)
//   loop_kernel    the persistent closed loop (cfz_loop_run): the same solver body fed from per-iteration ticket queues of
//                  (scenario, vehicle, iteration) work items, hand-offs between workgroups at agent scope
//   loop_prep      closed loop: parameters and shifted warm start of every vehicle from the
//                  previous predictions (reference vehicle_follower.py:432-476, 636-637)
//   loop_post      closed loop: read-back or shift fallback, plant integration, clock
//                  (reference :484-563)
// The planning kernels (state_ws, collocation plans) and their entry points live in cfz_planning.hip, a translation unit
// of its own (two units compile in parallel; both at -O3 since round 3, see __graft_entry__.build).
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
#include "cfz_common.h"

thread_local std::string cfz_g_err;  // cfz_last_error(); shared with cfz_planning.hip (cfz_common.h)

namespace {

// Plant (vehicle_follower.py:528-543, CasADi integrator "idas"): RK4 with this many sub-steps per dt.  10 sub-steps are
// within 6e-11 of the converged solution over the whole input range, tighter than IDAS's default tolerances.
constexpr int kPlantSubsteps = 10;

struct DualPtrs { double *l, *m, *lam_ij, *lam_ji, *s; };
// spec, derived constants and workspace layout in device memory: the solver kernels read the fields where they use them
// (scalar loads, scalar cache) instead of holding all ~200 of them in scalar registers from the kernel's first instruction
struct KArgs { cfz::KSpec sp; cfz::KDer dv; cfz::Lay L; };

#ifndef CFZ_WAVES_PER_SIMD
#define CFZ_WAVES_PER_SIMD 2
#endif
__global__ __launch_bounds__(cfz::kNL, CFZ_WAVES_PER_SIMD) void solve_kernel(const KArgs *__restrict__ ka, int B, const double *x0,
                                                   const double *ref, const double *nbr, double *zu, int32_t *status,
                                                   int32_t *iters, double *stats, DualPtrs du, const int32_t *order,
                                                   double *wst, int wst_stride, const int32_t *carry, int carry_all,
                                                   const int32_t *slots) {
  extern __shared__ double smem[];
  if ((int)blockIdx.x >= B) return;
  const cfz::KSpec &sp = ka->sp; const cfz::KDer &dv = ka->dv; const cfz::Lay &L = ka->L;
  // workgroups are dispatched in index order: `order` puts the instances expected to run longest first
  const int b = order ? order[blockIdx.x] : (int)blockIdx.x;
  const int N = sp.N, no = sp.n_obs, nn = sp.n_nbr;
  int oi[2]; double od[3];
  cfz::DualOut duo = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#ifdef CFZ_STAMPS
  duo.stamps = reinterpret_cast<unsigned long long *>(stats) + (size_t)B * 3 + (size_t)b * 24;  // diagnostic build: stats has room
#endif
  if (du.l) {
    duo.l = du.l + (size_t)b * N * 4 * no; duo.mm = du.m + (size_t)b * N * 4 * no;
    duo.lam_ij = du.lam_ij + (size_t)b * nn * N * 4; duo.lam_ji = du.lam_ji + (size_t)b * nn * N * 4;
    duo.s = du.s + (size_t)b * nn * N * 2;
  }
  // carry record of the instance's slot (default: slot b): used when the caller says that this solve is the successor of
  // the previous one in that slot
  const int slot = slots ? slots[b] : b;
  cfz::solve_instance(sp, dv, x0 + (size_t)b * 5, ref + (size_t)b * 3 * N, nbr + (size_t)b * nn * 3 * N,
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
__global__ __launch_bounds__(cfz::kNL, CFZ_WAVES_PER_SIMD) void loop_kernel(const KArgs *__restrict__ ka, int S, int V, int K, int T,
                                                        const double *ref_table, const int32_t *kidx0, int t_base,
                                                        double *pred, double *state, double *scratch, int32_t *qbuf,
                                                        int32_t *ctrl, int32_t *done, int32_t *status, int32_t *iters,
                                                        double *stats, int32_t *iter_sum, double *wst, int wst_stride, int prio_lag) {
  extern __shared__ double smem[];
  const cfz::KSpec &sp = ka->sp; const cfz::KDer &dv = ka->dv; const cfz::Lay &L = ka->L;
  const int N = sp.N, nn = sp.n_nbr, B = S * V, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *my = scratch + (size_t)blockIdx.x * (5 + 3 * N + nn * 3 * N + 7 * N);
  double *ref = my + 5;  // (the record keeps the layout x0 | ref | nbr | zu of the stepwise path; only ref is used here)
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
    // ---- parameters and shifted warm start (vehicle_follower.py:432-476), straight into the solver's workspace: measured
    // state, neighbours' poses with cos / sin, warm start (solve_instance's `preloaded` form); only the reference goes through
    // a global record (the solver reads it from there in every iteration)
    if (tid < 5) smem[L.x0 + tid] = state[b * 5 + tid];
    for (int k = tid; k < N; k += cfz::kNL) {
      const int ka = (k + 1 < N) ? k + 1 : N - 1;
      int kr = kidx0[s] + t_base + t + k; if (kr > T - 1) kr = T - 1;
      for (int c = 0; c < 3; ++c) ref[c * N + k] = ref_table[((size_t)v * T + kr) * 7 + c];
      for (int c = 0; c < 7; ++c) smem[L.p + k * cfz::kNP + c] = pin[((size_t)b * 7 + c) * N + ka];
      int o = 0;
      for (int u = 0; u < V; ++u) {
        if (u == v) continue;
        const size_t bo = (size_t)s * V + u;
        double *q = smem + L.nb4 + (k * nn + o) * 4;
        const double po = pin[(bo * 7 + 2) * N + ka];
        q[0] = pin[(bo * 7 + 0) * N + ka]; q[1] = pin[(bo * 7 + 1) * N + ka]; q[2] = cos(po); q[3] = sin(po);
        ++o;
      }
    }
    __syncthreads();
    CFZ_MARK(3);
    int oi[2]; double od[3];
    cfz::DualOut duo = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    cfz::solve_instance(sp, dv, nullptr, ref, nullptr, nullptr, smem, L, oi, od, duo, wst ? wst + (size_t)b * wst_stride : nullptr, 1, 2);
    __syncthreads();
    CFZ_MARK(4);
    // ---- read-back (the solution is still in the workspace) or shift fallback (:484-524), plant (:528-543) ------------
    for (int i = tid; i < 7 * N; i += cfz::kNL) {
      const int c = i / N, k = i - c * N;
      const int ka = (k + 1 < N) ? k + 1 : N - 1;
      pout[(size_t)b * 7 * N + i] = (oi[1] == 0) ? smem[L.p + k * cfz::kNP + c] : pin[((size_t)b * 7 + c) * N + ka];
    }
    CFZ_MARK(5);
    if (tid == 0) {
      const double a0 = (oi[1] == 0) ? smem[L.p + 5] : pin[((size_t)b * 7 + 5) * N + 1];
      const double w0 = (oi[1] == 0) ? smem[L.p + 6] : pin[((size_t)b * 7 + 6) * N + 1];
      double z[5], out[5];
      for (int i = 0; i < 5; ++i) z[i] = smem[L.x0 + i];
      cfz::rk4_step<false>(z, a0, w0, sp.dt, sp.wb, kPlantSubsteps, out, nullptr);
      for (int i = 0; i < 5; ++i) state[b * 5 + i] = out[i];
      status[b] = oi[1]; iters[b] = oi[0];
      stats[b * 3] = od[0]; stats[b * 3 + 1] = od[1]; stats[b * 3 + 2] = od[2];
      atomicAdd(iter_sum, oi[0]);
      if (oi[1] == 0) atomicAdd(iter_sum + 1, 1);  // converged solves of this launch
      atomicAdd(iter_sum + 2 + (oi[1] < 0 ? 0 : (oi[1] > 5 ? 5 : oi[1])), 1);  // ... and how every solve of it ended (status 0..5)
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

// ---- dual warm starts: exact separation of two convex quadrilaterals and the duals that certify it ------------------
// The reference maximises d over the OBCA duals (vehicle.py:233-296, multi_vehicle_planner.py:208-341); the optimum is
// the Euclidean distance of the two polygons and the optimal duals encode the unit direction n* between their closest
// points: lam >= 0 with A'lam = n* (two faces of the polygon through its support vertex), mu >= 0 with G'mu = -R'n*
// (the body rectangle's normals are +-e_x, +-e_y, so mu is the positive/negative part).  Closest points of two disjoint
// convex polygons: a vertex of one and a point of an edge of the other (possibly its end point, the vertex-vertex case),
// so 2 x 4 x 4 point-segment distances decide.  If the polygons touch or overlap, the best face normal stands in (its
// value is then <= 0; the reference's optimum with |A'lam| <= 1 would be 0 at lam = mu = 0).
__device__ inline void point_segment(double qx, double qy, double ax, double ay, double bx, double by, double &dist2, double &nx, double &ny) {
  const double ex = bx - ax, ey = by - ay, wx = qx - ax, wy = qy - ay;
  double t = (wx * ex + wy * ey) / (ex * ex + ey * ey);
  t = fmin(fmax(t, 0.0), 1.0);
  nx = wx - t * ex; ny = wy - t * ey;  // from the segment's closest point to q
  dist2 = nx * nx + ny * ny;
}

// unit direction n from polygon P (vertices PV, counter-clockwise) towards polygon Q and their distance; false if they
// are not strictly apart
__device__ inline bool polygon_gap(const double PV[4][2], const double QV[4][2], double &nx, double &ny, double &dist) {
  double best = INFINITY, bx = 0.0, by = 0.0;
  for (int v = 0; v < 4; ++v)
    for (int e = 0; e < 4; ++e) {
      double d2, ux, uy;
      point_segment(QV[v][0], QV[v][1], PV[e][0], PV[e][1], PV[(e + 1) & 3][0], PV[(e + 1) & 3][1], d2, ux, uy);  // P's edge -> Q's vertex
      if (d2 < best) { best = d2; bx = ux; by = uy; }
      point_segment(PV[v][0], PV[v][1], QV[e][0], QV[e][1], QV[(e + 1) & 3][0], QV[(e + 1) & 3][1], d2, ux, uy);  // Q's edge -> P's vertex
      if (d2 < best) { best = d2; bx = -ux; by = -uy; }
    }
  dist = sqrt(best);
  if (!(dist > 1e-12)) return false;
  nx = bx / dist; ny = by / dist;
  return true;
}

// lam >= 0 on the faces of {A p <= b} (vertices V) with A'lam = n: the two faces through the support vertex in direction n
__device__ inline void cone_duals(const double A[4][2], const double b[4], const double V[4][2], double nx, double ny, double lam[4]) {
  int v = 0; double sup = -INFINITY;
  for (int i = 0; i < 4; ++i) { const double h = nx * V[i][0] + ny * V[i][1]; if (h > sup) { sup = h; v = i; } }
  int i0 = 0, i1 = 1; double r0 = INFINITY, r1 = INFINITY;
  for (int i = 0; i < 4; ++i) {
    const double r = fabs(A[i][0] * V[v][0] + A[i][1] * V[v][1] - b[i]);
    if (r < r0) { r1 = r0; i1 = i0; r0 = r; i0 = i; } else if (r < r1) { r1 = r; i1 = i; }
  }
  const int ia = i0 < i1 ? i0 : i1, ib = i0 < i1 ? i1 : i0;
  const double det = A[ia][0] * A[ib][1] - A[ib][0] * A[ia][1];
  for (int i = 0; i < 4; ++i) lam[i] = 0.0;
  lam[ia] = fmax((A[ib][1] * nx - A[ib][0] * ny) / det, 0.0);
  lam[ib] = fmax((-A[ia][1] * nx + A[ia][0] * ny) / det, 0.0);
}

// dual_ws (reference vehicle.py:233-296): for fixed poses, the optimal dual certificate of every (pose, obstacle) pair and
// the separation it certifies.  One thread per pair.
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
  double s, c;
  sincos(psi, &s, &c);
  const double g0 = sp.g[0], g1 = sp.g[1], g2 = sp.g[2], g3 = sp.g[3];
  const double BV[4][2] = {{g0, g1}, {-g2, g1}, {-g2, -g3}, {g0, -g3}};
  double W[4][2];  // body vertices in the world frame, counter-clockwise
  for (int i = 0; i < 4; ++i) { W[i][0] = x + c * BV[i][0] - s * BV[i][1]; W[i][1] = y + s * BV[i][0] + c * BV[i][1]; }
  double lam[4] = {0, 0, 0, 0}, muv[4] = {0, 0, 0, 0}, nx, ny, dist;
  if (polygon_gap(V, W, nx, ny, dist)) {  // n: from the obstacle towards the vehicle
    cone_duals(A, b, V, nx, ny, lam);
    const double mx = -(c * nx + s * ny), my = -(-s * nx + c * ny);  // G'mu = -R'n
    muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
  } else {  // touching or overlapping: the best face normal
    double sep2[2];
    const int sel = cfz::select_rows_sep(A, b, V, x, y, c, s, sp.g, 0, sep2);
    const int kind = sel >> 6, f = (sel >> 4) & 3;
    if (kind == 1) { nx = A[f][0]; ny = A[f][1]; }
    else { const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3); nx = -(c * gx - s * gy); ny = -(s * gx + c * gy); }
    cone_duals(A, b, V, nx, ny, lam);
    const double mx = -(c * nx + s * ny), my = -(-s * nx + c * ny);
    muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
  }
  for (int i = 0; i < 4; ++i) { l[(size_t)k * 4 * no + 4 * j + i] = lam[i]; mu_out[(size_t)k * 4 * no + 4 * j + i] = muv[i]; }
  if (d) {  // the value the rows certify: -g'mu + (A t - b)'lam  (:276)
    double v = 0.0;
    for (int i = 0; i < 4; ++i) v += -sp.g[i] * muv[i] + (A[i][0] * x + A[i][1] * y - b[i]) * lam[i];
    d[(size_t)k * no + j] = v;
  }
}

// joint_dual_ws (reference multi_vehicle_planner.py:208-341): for n pairs of fixed poses of two vehicles, the duals
// lam (faces of the first), mu (faces of the second), s and the separation d of the rows
//   -b_this'lam - b_other'mu = d,  A_this'lam + s = 0,  A_other'mu - s = 0,  |s| <= 1,  lam, mu >= 0   (:292-295)
// at the optimum: d = the distance of the two bodies, s = -w with w the unit direction from this vehicle to the other.
__global__ void joint_dual_ws_kernel(const cfz::KSpec sp, int n, const double *pa, const double *pb, double *lam_o,
                                     double *mu_o, double *s_o, double *d_o) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double x = pa[3 * k], y = pa[3 * k + 1], xo = pb[3 * k], yo = pb[3 * k + 1];
  double s, c, so, co;
  sincos(pa[3 * k + 2], &s, &c); sincos(pb[3 * k + 2], &so, &co);
  const double g0 = sp.g[0], g1 = sp.g[1], g2 = sp.g[2], g3 = sp.g[3];
  const double BV[4][2] = {{g0, g1}, {-g2, g1}, {-g2, -g3}, {g0, -g3}};
  double W[4][2], V[4][2];
  for (int i = 0; i < 4; ++i) {
    W[i][0] = x + c * BV[i][0] - s * BV[i][1]; W[i][1] = y + s * BV[i][0] + c * BV[i][1];
    V[i][0] = xo + co * BV[i][0] - so * BV[i][1]; V[i][1] = yo + so * BV[i][0] + co * BV[i][1];
  }
  double wx, wy, dist;  // w: unit direction from this vehicle towards the other
  if (!polygon_gap(W, V, wx, wy, dist)) {  // touching or overlapping: the best face normal of either body
    double A[4][2] = {{co, so}, {-so, co}, {-co, -so}, {so, -co}}, b[4], sep2[2];
    for (int i = 0; i < 4; ++i) b[i] = A[i][0] * xo + A[i][1] * yo + sp.g[i];
    const int sel = cfz::select_rows_sep(A, b, V, x, y, c, s, sp.g, 0, sep2);
    const int kind = sel >> 6, f = (sel >> 4) & 3;
    const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
    if (kind == 1) { wx = -(co * gx - so * gy); wy = -(so * gx + co * gy); }  // a face of the other body, normal towards this one
    else { wx = c * gx - s * gy; wy = s * gx + c * gy; }
  }
  double lam[4], mu[4];
  const double lx = c * wx + s * wy, ly = -s * wx + c * wy;          // R' w = G' lam
  lam[0] = fmax(lx, 0.0); lam[1] = fmax(ly, 0.0); lam[2] = fmax(-lx, 0.0); lam[3] = fmax(-ly, 0.0);
  const double mx = -(co * wx + so * wy), my = -(-so * wx + co * wy);  // Ro' (-w) = G' mu
  mu[0] = fmax(mx, 0.0); mu[1] = fmax(my, 0.0); mu[2] = fmax(-mx, 0.0); mu[3] = fmax(-my, 0.0);
  for (int i = 0; i < 4; ++i) { lam_o[4 * k + i] = lam[i]; mu_o[4 * k + i] = mu[i]; }
  s_o[2 * k] = -wx; s_o[2 * k + 1] = -wy;  // s = -A_this' lam = A_other' mu
  if (d_o) {  // -b_this'lam - b_other'mu with b = G R(-psi) t + g   (:292)
    const double tl = c * x + s * y, tm = -s * x + c * y, ol = co * xo + so * yo, om = -so * xo + co * yo;
    const double bt[4] = {tl + g0, tm + g1, -tl + g2, -tm + g3}, bo[4] = {ol + g0, om + g1, -ol + g2, -om + g3};
    double v = 0.0;
    for (int i = 0; i < 4; ++i) v -= bt[i] * lam[i] + bo[i] * mu[i];
    d_o[k] = v;
  }
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
  KArgs *kargs = nullptr;     // device copy of {ks, derive(ks), lay}
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
  long last_iter_sum = 0, last_converged = 0, last_status[6] = {0, 0, 0, 0, 0, 0};
  CfzArena arena;  // device buffers of cfz_dual_ws / cfz_joint_dual_ws, kept between calls
};

namespace {


int launch_solve(cfz_handle *h, int B, const double *x0, const double *ref, const double *nbr, double *zu,
                 int32_t *status, int32_t *iters, double *stats, bool duals, hipStream_t st,
                 const int32_t *order = nullptr, int carry_all = 0) {
  DualPtrs du = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (duals) du = {h->l, h->m, h->lam_ij, h->lam_ji, h->s};
  if ((h->carry_set || h->slots_set) && st != h->stream) HIP_OK(hipStreamWaitEvent(st, h->ev_stage, 0));  // staged on the handle's stream
  HIP_OK(hipEventRecord(h->ev0, st));
  hipLaunchKernelGGL(solve_kernel, dim3(B), dim3(cfz::kNL), h->lds_bytes, st, h->kargs, B, x0, ref, nbr, zu, status,
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

const char *cfz_last_error(void) { return cfz_g_err.c_str(); }

#ifndef CFZ_SRC_HASH
#define CFZ_SRC_HASH "unknown"
#endif
const char *cfz_source_hash(void) { return CFZ_SRC_HASH; }

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
  o->stall_iters = 10; o->stall_kappa = 0.9; o->row_curvature = 1; o->carry_duals = 1; o->vv_rows = 1; o->shift_after = 60; o->restoration = 2; o->shift_stagnation = 10; o->err_stall_iters = 150; o->carry_shift = 1; o->warm_push = 1e-6;
  o->reg_dual_rows = 1e-8; o->resto_first = 0.3;
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
  if (opt->restoration < 0 || !(opt->reg_dual_rows >= 0.0) || !(opt->resto_first >= 0.0)) return fail("restoration, reg_dual_rows, resto_first must not be negative");
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
  k.stall_iters = opt->stall_iters; k.stall_kappa = opt->stall_kappa; k.row_curvature = opt->row_curvature; k.vv_rows = opt->vv_rows; k.shift_after = opt->shift_after; k.resto = opt->restoration; k.stag_win = opt->shift_stagnation; k.err_stall = opt->err_stall_iters; k.carry_shift = opt->carry_shift ? 1 : 0; k.pad_ks = 0; k.warm_push = opt->warm_push; k.reg_dual_rows = opt->reg_dual_rows; k.resto_first = opt->resto_first;
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
    KArgs host_args = {h->ks, cfz::derive(h->ks), h->lay};
    HIP_OK(hipMalloc(&h->kargs, sizeof(KArgs)));
    HIP_OK(hipMemcpy(h->kargs, &host_args, sizeof(KArgs), hipMemcpyHostToDevice));
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
  // stats: 3 doubles per instance (+ 24 phase counters per instance for the -DCFZ_STAMPS diagnostic build)
  HIP_OK(hipMalloc(&h->stats, B * (3 + 24) * 8)); HIP_OK(hipMalloc(&h->status, B * 4)); HIP_OK(hipMalloc(&h->iters, B * 4));
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
                  h->iter_sum, h->obs_tab, h->wst, h->carry, h->slots, h->kargs};
  for (void *p : bufs) if (p) hipFree(p);
  arena_destroy(h->arena);
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
  {  // two instances of one launch on the same carry record would race on it (and mix two vehicles' multipliers)
    std::vector<char> seen((size_t)h->max_batch, 0);
    for (int b = 0; b < B; ++b) { if (seen[slots[b]]) return fail("duplicate carry slot within one solve"); seen[slots[b]] = 1; }
  }
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
// diagnostic build only: the 24 phase counters of every instance of the last solve_kernel launch
int cfz_debug_stamps(cfz_handle *h, int B, unsigned long long *out) {
  if (check(h, B)) return -1;
  HIP_OK(hipMemcpy(out, h->stats + (size_t)B * 3, (size_t)B * 24 * 8, hipMemcpyDeviceToHost));
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

int cfz_dual_ws(cfz_handle *h, int n, const double *poses, double *l, double *m, double *d) {
  if (!h) return fail("null handle");
  if (n < 1 || !poses || !l || !m) return fail("bad argument");
  HIP_OK(hipSetDevice(h->device));
  const size_t no = h->ks.n_obs;
  if (no == 0) return 0;
  double *dp = nullptr, *dl = nullptr, *dm = nullptr, *dd = nullptr;
  if (arena_reset(h->arena)) return -1;
  ARENA_ALLOC(h->arena, dp, (size_t)n * 3 * 8); ARENA_ALLOC(h->arena, dl, (size_t)n * 4 * no * 8);
  ARENA_ALLOC(h->arena, dm, (size_t)n * 4 * no * 8); ARENA_ALLOC(h->arena, dd, (size_t)n * no * 8);
  HIP_OK(hipMemcpyAsync(dp, poses, (size_t)n * 3 * 8, hipMemcpyHostToDevice, h->stream));
  const int nt = n * (int)no;
  hipLaunchKernelGGL(dual_ws_kernel, dim3((nt + 127) / 128), dim3(128), 0, h->stream, h->ks, n, dp, dl, dm, dd);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(l, dl, (size_t)n * 4 * no * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(m, dm, (size_t)n * 4 * no * 8, hipMemcpyDeviceToHost, h->stream));
  if (d) HIP_OK(hipMemcpyAsync(d, dd, (size_t)n * no * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

int cfz_joint_dual_ws(cfz_handle *h, int n, const double *poses_this, const double *poses_other, double *lam, double *mu,
                      double *s, double *d) {
  if (!h) return fail("null handle");
  if (n < 1 || !poses_this || !poses_other || !lam || !mu || !s) return fail("bad argument");
  HIP_OK(hipSetDevice(h->device));
  double *dpa = nullptr, *dpb = nullptr, *dout = nullptr;
  if (arena_reset(h->arena)) return -1;
  ARENA_ALLOC(h->arena, dpa, (size_t)n * 3 * 8); ARENA_ALLOC(h->arena, dpb, (size_t)n * 3 * 8); ARENA_ALLOC(h->arena, dout, (size_t)n * 11 * 8);
  HIP_OK(hipMemcpyAsync(dpa, poses_this, (size_t)n * 3 * 8, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(dpb, poses_other, (size_t)n * 3 * 8, hipMemcpyHostToDevice, h->stream));
  double *dl = dout, *dm = dout + (size_t)n * 4, *ds = dout + (size_t)n * 8, *dd = dout + (size_t)n * 10;
  hipLaunchKernelGGL(joint_dual_ws_kernel, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->ks, n, dpa, dpb, dl, dm, ds, dd);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(lam, dl, (size_t)n * 4 * 8, hipMemcpyDeviceToHost, h->stream)); HIP_OK(hipMemcpyAsync(mu, dm, (size_t)n * 4 * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(s, ds, (size_t)n * 2 * 8, hipMemcpyDeviceToHost, h->stream));
  if (d) HIP_OK(hipMemcpyAsync(d, dd, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
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
  if (noise) { if (arena_reset(h->arena)) return -1; ARENA_ALLOC(h->arena, dn, B * 5 * 8); HIP_OK(hipMemcpy(dn, noise, B * 5 * 8, hipMemcpyHostToDevice)); }
  const long nt = (long)B * N;
  hipLaunchKernelGGL(loop_seed, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->stream, S, V, N, T, h->ref_table,
                     h->kidx, dn, h->pred, h->state);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
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
    HIP_OK(hipMalloc(&h->done, (size_t)S * 4)); HIP_OK(hipMalloc(&h->iter_sum, 32));
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
    HIP_OK(hipMemsetAsync(h->iter_sum, 0, 32, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));  // `first` and `ctrl0` are host temporaries
  }
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(loop_kernel, dim3(grid), dim3(cfz::kNL), h->lds_bytes, h->stream, h->kargs, S, V, K, h->T,
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
  int32_t ctrl[4] = {0, 0, 0, 0}, isum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIP_OK(hipMemcpy(ctrl, h->ctrl, sizeof ctrl, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(isum, h->iter_sum, 32, hipMemcpyDeviceToHost));
  h->last_iter_sum = isum[0]; h->last_converged = isum[1];
  for (int i = 0; i < 6; ++i) h->last_status[i] = isum[2 + i];
  h->have_order = false;
  if (ctrl[2]) return fail("persistent loop kernel timed out waiting for a work item");
  return 0;
}

long cfz_loop_last_iterations(const cfz_handle *h) { return h ? h->last_iter_sum : -1; }
long cfz_loop_last_converged(const cfz_handle *h) { return h ? h->last_converged : -1; }
int cfz_loop_last_status_counts(const cfz_handle *h, long counts[6]) {
  if (!h || !counts) return fail("null argument");
  for (int i = 0; i < 6; ++i) counts[i] = h->last_status[i];
  return 0;
}

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
