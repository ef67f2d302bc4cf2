/* confrez_hip.h -- C ABI of libconfrez_hip.so, the MI355X (gfx950) engine behind the
 * collision-free planning hot path of XuShenLZ/conflict_rez.
 *
 * The reference has no FFI: its seam is the CasADi call `opti.solve()` inside
 * `VehicleFollower.step` (confrez/control/vehicle_follower.py:428-563).  Every entry point
 * below names the reference statements it stands in for.  All arrays are contiguous fp64
 * (int32 where stated), instance-major: the leading index is the problem instance
 * b in [0, B).  Host-pointer calls copy in/out during the call and retain nothing;
 * `_device` calls take HIP device pointers and enqueue on the handle's stream.
 *
 * Return value: 0 on success, <0 on an API error (cfz_last_error() explains).  Solver
 * outcomes are per instance, in `status`:
 *   0 converged | 1 iteration limit | 2 line search failed | 3 non-finite iterate |
 *   4 measured state already violates a collision row (NLP infeasible, nothing iterated) |
 *   5 constraint violation stalled above constr_viol_tol (locally infeasible; IPOPT: restoration failed)
 * The reference turns any non-zero outcome into a Python exception that `step()` catches
 * to apply its shift fallback (vehicle_follower.py:478-524); the Python shim does the same.
 *
 * A handle is not thread-safe (the reference is single-threaded).
 */
#ifndef CONFREZ_HIP_H
#define CONFREZ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFZ_MAX_OBS 8   /* static obstacles (reference: 6, compute_sets.py:259-330) */
#define CFZ_MAX_NBR 7   /* neighbouring vehicles (reference: 3) */
#define CFZ_MAX_N 32    /* horizon stages (reference: 30, vehicle_follower.py:146); four lanes of a 128-lane workgroup per stage */

/* Constants of the MPC-step NLP: what `setup_controller(dt, N, dmin)` bakes into the CasADi
 * graph (vehicle_follower.py:146-368). */
typedef struct cfz_spec {
  int32_t N;           /* horizon stages                       :146 */
  int32_t n_obs;       /* static obstacles, 4 half-planes each :152-156 */
  int32_t n_nbr;       /* other vehicles                       :303 */
  int32_t rk_substeps; /* RK4 sub-steps M of dynamic_model.py:30 (reference 4) */
  double dt;           /*                                      :146 */
  double wb;           /* wheelbase, vehicle_types.py:19 */
  double dmin;         /* clearance                            :146 */
  double g[4];         /* body rectangle lf, w/2, lr, w/2 (rows +x,+y,-x,-y), vehicle_types.py:65-71 */
  double bounds[12];   /* lo,hi of x, y, v, delta, a, w        :204-240 */
  double weights[6];   /* (x-xr)^2,(y-yr)^2,(psi-psir)^2,a^2,(v w)^2,delta^2   :263-271 */
  double A_obs[CFZ_MAX_OBS][4][2]; /* unit outward normals     :280-290 */
  double b_obs[CFZ_MAX_OBS][4];
} cfz_spec;

/* Stopping rule and interior-point constants: `opti.solver("ipopt", p_opts, s_opts)`
 * (vehicle_follower.py:356-368) plus the IPOPT defaults that call leaves untouched. */
typedef struct cfz_options {
  int32_t max_iter;       /* :364 600 */
  int32_t max_backtrack;  /* line-search halvings before status 2 */
  int32_t filter_cap;     /* filter entries kept per barrier problem */
  int32_t stall_iters;    /* 10: iterations without progress of the constraint violation (those that changed the working set
                           *     of the separation rows count a quarter) before status 5; 0 = off */
  int32_t row_curvature;  /* 1: Hessian = Gauss-Newton objective part + multiplier-weighted curvature of the separation rows */
  int32_t carry_duals;    /* 1: keep one carry record per slot (multipliers of the last converged solve); 0: never */
  int32_t vv_rows;        /* 1: a block whose closest features are two vertices is constrained by their Euclidean distance
                           *    (exactly the reference's OBCA rows there); 0: face-normal certificates only (a restriction) */
  int32_t shift_after;    /* 60: from this iteration on a stage whose row curvature the convexity safeguard would scale keeps it
                           *     whole and is shifted by the smallest multiple of the identity instead (a scaled model can
                           *     cycle for hundreds of iterations on a vehicle pressed into a corner); 0 = never */
  int32_t restoration;    /* 2: restoration phases a solve may go through (IPOPT's answer to a failed line search, paper sec. 3.3):
                           *    Levenberg-Marquardt on the squared violations of the separation rows and the boxes, on the solver's own
                           *    stage recursion, until the worst violation is a tenth of what it was; then cold multipliers at the
                           *    restored point.  Entered after a failed line search at an iterate whose violation exceeds
                           *    constr_viol_tol, and before the first iteration of a start that resto_first describes.  A restoration
                           *    that does not reach its goal ends the solve with status 5 (IPOPT: "converged to a point of local
                           *    infeasibility").  0 = none: a failed line search is status 2 */
  int32_t shift_stagnation; /* 10: once the scaled optimality error has not halved for this many iterations at a feasible iterate
                           *    (violation <= constr_viol_tol) the late curvature shift may start at iteration 40 instead of waiting for
                           *    shift_after; ends the sawtooth of the scaled model 15-20 iterations sooner; 0 = off */
  int32_t err_stall_iters; /* 150: a solve whose scaled optimality error has not halved for this many iterations ends with status 5
                           *    instead of running to max_iter (limit cycles below constr_viol_tol escape stall_iters' test); 0 = off */
  int32_t carry_shift;    /* 1: a converged solve in which some stage's row curvature had to be shifted tells the next solve of the same
                           *    carry slot (cfz_mpc_set_carry / the closed loop) to shift from its first iteration instead of waiting for
                           *    shift_after / shift_stagnation: a cornered vehicle otherwise repeats the 40+ iterations of the scaled
                           *    model at every MPC iteration.  Solves without carried multipliers are unaffected.  0 = off */
  double tol;             /* :362 1e-2 */
  double constr_viol_tol; /* :363 1e-2 */
  double dual_inf_tol;    /* IPOPT default 1 */
  double compl_inf_tol;   /* IPOPT default 1e-4 */
  double mu_init /* 1e-3 (IPOPT: 0.1) */, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac, s_max, kappa_sigma;
  double eta_phi, gamma_theta, gamma_phi, delta_sw, s_theta, s_phi, reg_primal;
  double stall_kappa;     /* 0.9: progress = violation below stall_kappa x its last checkpoint */
  double warm_push;       /* 1e-6: distance from a bound kept by a start that carries multipliers */
  double reg_dual_rows;   /* 1e-8: IPOPT's dual regularisation delta_c (paper sec. 3.1), permanently on the separation rows: two rows of
                           *    a block, or rows of different blocks of a stage, become dependent at a contact and their multipliers
                           *    run away along the null space; 0 = off */
  double resto_first;     /* 0.3 (metres): a start whose separation rows of stages >= 1 are violated by more than this -- the warm
                           *    start of a vehicle whose neighbour's prediction has moved into its path -- goes through the restoration
                           *    phase before the first iteration; 0 = never */
} cfz_options;

typedef struct cfz_handle cfz_handle;

void cfz_default_spec(cfz_spec *spec);       /* reference constants, no obstacles */
void cfz_default_options(cfz_options *opt);  /* reference s_opts + IPOPT defaults */

/* Builds the solver for one NLP structure on `device` for up to `max_batch` instances.
 * Replaces the graph construction half of setup_controller (:165-368). */
int cfz_create(const cfz_spec *spec, const cfz_options *opt, int device, int max_batch, cfz_handle **out);
int cfz_destroy(cfz_handle *h);
int cfz_max_batch(const cfz_handle *h);
/* LDS bytes one instance (= one wavefront) occupies and how many instances the runtime keeps resident per CU. */
int cfz_kernel_info(const cfz_handle *h, int32_t *lds_bytes_per_instance, int32_t *instances_per_cu);

/* opti.set_value(current_x..current_delta, current_ref_*, p_other_pred[*]) (:432-456).
 * x0[B][5]; ref[B][3][N] rows x,y,psi; nbr[B][n_nbr][3][N] (already advanced by the caller
 * as `_adv_onestep` does, :445-455). */
int cfz_mpc_set_params(cfz_handle *h, int B, const double *x0, const double *ref, const double *nbr);

/* opti.set_initial(x..w) (:466-473).  zu[B][7][N] rows x,y,psi,v,delta,a,w.  The dual
 * initial guesses of :458-464,:475-476 are not taken: the engine eliminates the OBCA duals
 * and rebuilds them from the poses (DESIGN.md "Certificate elimination"). */
int cfz_mpc_set_warm(cfz_handle *h, int B, const double *zu);

/* Multipliers from the previous MPC iteration (IPOPT: warm_start_init_point; the reference hands the previous
 * duals to opti.set_initial at :458-464, :475-476).  carry[b] != 0 declares that the next cfz_mpc_solve of slot b
 * is the MPC iteration following the one last solved in slot b (horizon moved on by one stage): if that solve
 * converged, the interior-point iteration starts from its slack and bound multipliers, costates and barrier
 * parameter, shifted by one stage, instead of z = 1, nu = 0, mu = mu_init.  The flags hold for one solve; NULL or
 * no call = cold.  The closed loop (cfz_loop_*) always carries.  Typically 1.8 instead of 4.2 iterations. */
int cfz_mpc_set_carry(cfz_handle *h, int B, const int32_t *carry);

/* The same flags as a HIP device array int32[B] (e.g. `status == 0` of the previous iteration, computed on the device):
 * nothing is copied or synchronised; the array is read by the next solve's kernel and must stay valid until it ends. */
int cfz_mpc_set_carry_device(cfz_handle *h, int B, const int32_t *d_carry);

/* Which carry record each instance of the next solve reads and refreshes: slots[b] in [0, max_batch).  Default (no
 * call, or NULL): instance b uses record b.  Lets several callers share one handle without mixing their multipliers:
 * the reference's `for v in vehicles: v.step()` loop (:642-647) and the ROS nodes (vehicle_node.py:150-152) step ONE
 * vehicle at a time, so each vehicle solves a batch of one in its own slot.  Holds for one solve, like the flags.
 * The slots of one call must be pairwise distinct (two instances on one record would race on it): duplicates are an API error. */
int cfz_mpc_set_slots(cfz_handle *h, int B, const int32_t *slots);

/* sol = opti.solve() (:479): runs the batched solver and blocks until done. */
int cfz_mpc_solve(cfz_handle *h, int B);

/* sol.value(...) (:484-500).  Any output pointer may be NULL.
 * zu[B][7][N]; l,m [B][N][4*n_obs]; lam_ij,lam_ji [B][n_nbr][N][4]; s [B][n_nbr][N][2]. */
int cfz_mpc_get(cfz_handle *h, int B, double *zu, double *l, double *m, double *lam_ij, double *lam_ji, double *s);

/* sol.stats() (:481): per-instance outcome.  Any pointer may be NULL.
 * status,iters int32[B]; cost, kkt_err (scaled optimality error E_0), min_sep fp64[B].
 * status: 0 converged (IPOPT Solve_Succeeded); 1 iteration limit; 2 line search failed; 3 non-finite iterate;
 *         4 measured state infeasible: it violates a collision row, or a box on x, y, v, delta, by more than constr_viol_tol
 *           (stage 0 is pinned to the measurement: no iterations; the rows of stage 0 take no part in the iteration otherwise);
 *         5 locally infeasible: the constraint violation stalled above constr_viol_tol, the scaled optimality error stalled
 *           (err_stall_iters), or a restoration phase did not reach its goal (IPOPT: "converged to a point of local infeasibility").
 *         Every status != 0 is what
 *         the reference sees as an exception from opti.solve() and answers with the shift fallback (:501-524). */
int cfz_mpc_stats(cfz_handle *h, int B, int32_t *status, int32_t *iters, double *cost, double *kkt_err,
                  double *min_sep);

/* Milliseconds the last cfz_mpc_solve / cfz_loop_step spent in its solver kernel (HIP events
 * on the handle's stream). */
double cfz_last_solve_ms(const cfz_handle *h);

/* ---- device-resident path ------------------------------------------------------------- */
/* Same as set_params + set_warm + solve + get(zu) on HIP device pointers, asynchronous on
 * `stream` (a hipStream_t; NULL = the handle's own stream).  d_status/d_iters int32[B],
 * d_stats fp64[B][3] = cost, kkt_err, min_sep.  d_zu is read as the warm start and overwritten
 * with the solution. */
int cfz_mpc_solve_device(cfz_handle *h, int B, const double *d_x0, const double *d_ref, const double *d_nbr,
                         double *d_zu, int32_t *d_status, int32_t *d_iters, double *d_stats, void *stream);

/* ---- vehicle-sharded closed loop (SURVEY.md 8e partitioning B; the reference's ROS deployment runs one process per vehicle
 * and exchanges predictions as messages, ros2_ws/src/confrez_ros/src/vehicle_node.py:111-189) --------------------------------
 * One MPC iteration of the n_own vehicles this rank owns in S scenarios, entirely on `stream` (NULL = the handle's), no host
 * synchronisation: parameters + shifted warm start from the gathered predictions (:432-476), solve started from the carried
 * multipliers, read-back or shift fallback (:484-524), plant (:528-543).  Instances are ordered [s][o].  All pointers are HIP
 * device pointers:
 *   d_own[n_own]            global indices of the owned vehicles, ascending
 *   d_table[n_own][T][7]    their plans sampled every dt; d_k0[S] start sample of every scenario; t = iteration number
 *   d_allpred[S][V][3][N]   x, y, psi of every vehicle's last prediction, vehicle order (the caller's all-gather)
 *   d_pred[S][n_own][7][N], d_state[S][n_own][5]   in/out;  d_status, d_iters int32[S n_own], d_stats fp64[S n_own][3] out
 *   d_carry int32[S n_own]  in/out: written with status == 0, read by the next call (t > 0) as cfz_mpc_set_carry_device */
int cfz_vsl_step(cfz_handle *h, int S, int V, int n_own, const int32_t *d_own, int T, const double *d_table, const int32_t *d_k0,
                 int t, const double *d_allpred, double *d_pred, double *d_state, int32_t *d_status, int32_t *d_iters,
                 double *d_stats, int32_t *d_carry, void *stream);

/* ---- dual warm start ----------------------------------------------------------------------
 * Vehicle.dual_ws (confrez/control/vehicle.py:233-296): for n fixed poses[n][3] = (x, y, psi) the duals
 * l, m [n][4*n_obs] that certify the separation d[n][n_obs] (may be NULL) of the vehicle body from every
 * static obstacle of the handle's spec.  The reference maximises that separation with IPOPT
 * (tol 1e-8); here it is the optimum in closed form: d = the Euclidean distance of the two polygons, the duals
 * encode the unit direction between their closest points (face-vertex and vertex-vertex cases alike) and satisfy
 * the rows (:276-280) exactly.  Polygons that touch or overlap get their best face normal (d <= 0). */
int cfz_dual_ws(cfz_handle *h, int n, const double *poses, double *l, double *m, double *d);

/* MultiVehiclePlanner.joint_dual_ws (confrez/control/multi_vehicle_planner.py:208-341): for n pairs of fixed poses
 * poses_this[n][3], poses_other[n][3] of two vehicles the duals lam[n][4] (faces of the first), mu[n][4] (faces of the
 * second), s[n][2] of the rows -b_this'lam - b_other'mu = d, A_this'lam + s = 0, A_other'mu - s = 0, |s| <= 1,
 * lam, mu >= 0 (:292-295) and the separation d[n] they certify (may be NULL).  The reference maximises d with IPOPT;
 * here: the optimum in closed form, d = the Euclidean distance of the two bodies, s = minus the unit direction between
 * their closest points (vertex-vertex closest features included). */
int cfz_joint_dual_ws(cfz_handle *h, int n, const double *poses_this, const double *poses_other, double *lam, double *mu,
                      double *s, double *d);

/* ---- Vehicle.state_ws (confrez/control/vehicle.py:99-231): warm-start plan through the strategy's tube -----------
 * One NLP per vehicle, B of them in one call (no handle: nothing is kept).  Instance b has n_sets[b] strategy steps,
 * T_b = N (n_sets[b] - 1) Euler steps and T_b + 1 trajectory points.
 *   init_pose[B][3]       x, y, psi of the first point (:131-138; v, delta, a0, w0 start at zero)
 *   final_heading[B]      psi of the last point (:194-195), NaN = free; the pointer may be NULL
 *   tube                  for every instance, for strategy steps 1..n_sets-1: back cell then front cell, each as
 *                         A[4][2] row-major followed by b[4] (24 doubles per step), instances back to back (:178-192)
 *   guess                 x, y, psi of every point, instances back to back (spline_ws, :199-205); NULL (spline_ws = False, what the
 *                         reference's scripts configure for vehicle_0: IPOPT then starts from zeros) = cfz_state_ws_default_guess
 *   traj (out)            x, y, psi, v, delta, a, w of every point, instances back to back; the last input is repeated
 *   status, iters, cost   per instance (may be NULL); status as in cfz_mpc_stats, the reference raises on status != 0
 * The solver is the interior point of the MPC path with the exact Hessian of the Lagrangian and IPOPT's delta_w ladder
 * driven by a curvature test; the Newton system is solved as a stage recursion (a Riccati sweep over the T stages with
 * the terminal-heading row, the only one that can lose rank, bordered and regularised: delta_c = 1e-9;
 * csrc/cfz_plan.inl).  With a terminal heading and a guess that stands still (guess = NULL) the first linearisation
 * is rank deficient -- the headings cannot move -- and the solve ends with status 2 after a few iterations: hence
 * the default guess through the tube when the caller has none (the speed along any guess is seeded here). */
#define CFZ_KERNEL_AUTO 0
#define CFZ_KERNEL_WIDE 1
#define CFZ_KERNEL_NARROW 2

typedef struct cfz_plan_options {
  int32_t N;              /* :100 steps per strategy step, 30 */
  int32_t max_iter;       /* :210 500 */
  int32_t bounded_input;  /* :104, :155-167 */
  int32_t stall_iters;    /* 0 (as the reference: an infeasible plan runs to max_iter); n > 0: status 5 after n iterations without
                           *   progress of the constraint violation, as in the MPC step -- a batch then does not wait for such a plan */
  int32_t kernel;         /* where the Riccati sweep keeps its per-stage data: 0 = by batch size (up to one plan per CU: CFZ_KERNEL_WIDE,
                           *   else CFZ_KERNEL_NARROW), CFZ_KERNEL_WIDE = 1 (in LDS, one plan per CU at a time: the fastest single plan),
                           *   CFZ_KERNEL_NARROW = 2 (in the workspace, several plans per CU: the highest throughput; also what a plan
                           *   too long for the LDS, T > 498, gets).  Same arithmetic in the same order: the two agree bit for bit. */
  int32_t reserved0;
  double dt;              /* :101 0.1 */
  double wb;              /* wheelbase */
  double shrink_tube;     /* :106 0.5 in the callers */
  double bounds[12];      /* lo,hi of x, y, v, delta, a, w */
  double tol;             /* :208 1e-2 */
  double constr_viol_tol; /* :209 1e-2 */
  double mu_init;         /* 0.1 (IPOPT's default, what the reference's state_ws runs with; 1e-3 until round 3) */
  double curv_kappa;      /* 1e-8 */
} cfz_plan_options;

void cfz_default_plan_options(cfz_plan_options *opt);

/* Workspace of the planning entry points: one HIP stream and the device buffers of the last call, kept and reused (a
 * planning call used to hipMalloc / hipFree seven to nine buffers and end in hipDeviceSynchronize, stalling every other
 * stream of the process).  The `_w` forms run on the workspace's stream and wait for that stream only; the plain forms
 * (cfz_state_ws, cfz_colloc, cfz_joint_colloc) use a workspace of their own per calling thread and device.  A workspace
 * is not thread-safe (one call at a time), like a handle. */
typedef struct cfz_plan_ws cfz_plan_ws;
int cfz_plan_ws_create(int device, cfz_plan_ws **out);
int cfz_plan_ws_destroy(cfz_plan_ws *ws);
/* Gives the device memory a workspace holds back to the runtime now (it is allocated again by the next call).  ws = NULL: the
 * calling thread's own workspaces behind the plain entry points, which otherwise live as long as the thread.  A workspace also
 * shrinks by itself: a call that used less than a quarter of a block above 256 MB releases that block at the start of the next. */
int cfz_plan_ws_trim(cfz_plan_ws *ws);
int cfz_state_ws_w(cfz_plan_ws *ws, int B, const cfz_plan_options *opt, const int32_t *n_sets, const double *init_pose,
                   const double *final_heading, const double *tube, const double *guess, double *traj, int32_t *status,
                   int32_t *iters, double *cost);
int cfz_state_ws(int device, int B, const cfz_plan_options *opt, const int32_t *n_sets, const double *init_pose,
                 const double *final_heading, const double *tube, const double *guess, double *traj, int32_t *status,
                 int32_t *iters, double *cost);
/* The guess cfz_state_ws takes for an instance when the caller passes none (host arithmetic, no GPU): x, y, psi of the
 * N (n_sets - 1) + 1 points of the piecewise-linear path from the initial pose through the centres of the back cells of strategy steps
 * 1 .. n_sets - 1 (centre = mean of the cell's vertices), heading at a step = direction from the back cell's centre to the front
 * cell's on the branch nearest the previous heading, the terminal heading (NaN = free) at the last step.
 * tube: (n_sets - 1) x 24 doubles as for cfz_state_ws; guess (out): (N (n_sets - 1) + 1) x 3.  0 = ok, -1 = bad argument. */
int cfz_state_ws_default_guess(int32_t n_sets, int32_t N, const double init_pose[3], double final_heading, const double *tube,
                               double *guess);

/* ---- Vehicle.setup_single_final_problem + solve_single_final_problem (confrez/control/vehicle.py:360-661) -------------
 * The single-vehicle collocation plan: N = N_per_set (n_sets - 1) intervals of free length dt, K = 5 Radau points each,
 * 6 N points per instance; B instances in one call (no handle: nothing is kept).
 *   spec                  wb, dmin (:369), g, bounds (:439-478) and the static obstacles (:523-541); N, dt, n_nbr, weights unused
 *   init_pose, final_heading, tube   as in cfz_state_ws (:426-436, :619-620, :570-617)
 *   guess                 x, y, psi, v, delta, a, w at every point, instances back to back (interp_ws_for_collocation's
 *                         output, :629-636); dt0[B] the initial interval length (:388-389)
 *   traj (out)            x, y, psi, v, delta, a, w at every point; dt (out) [B]
 *   status, iters, cost   per instance (may be NULL); the reference raises on status != 0
 * The OBCA duals l, m (:411-416) are eliminated as in the MPC step and rebuilt from the poses by cfz_dual_ws.  Same
 * interior point as cfz_state_ws; dt is bordered out of the banded system (csrc/cfz_colloc.inl). */
#define CFZ_COLLOC_K 5
typedef struct cfz_colloc_options {
  int32_t N_per_set;      /* :368 5 */
  int32_t max_iter;       /* IPOPT default 3000 (:652 leaves it unset) */
  int32_t exact_rows;     /* 0: dual regularisation of proximal type, delta_c = 1e-7 (rows met to delta_c x multiplier, robust where
                           *    the reference's rows lose rank); 1: IPOPT's form, delta_c = 1e-9 (exact optimum at tight tolerances) */
  int32_t one_pivot;      /* 0: the band is eliminated a panel of 16 pivots at a time (single plans: for batches up to two per CU);
                           *    1: one pivot at a time -- joint plan: same pivots, same arithmetic per entry, same factor; single
                           *    plans: the one-wavefront kernel with its LDS window.  Kept as the check of the panel version, ~2x slower */
  int32_t vv_rows;        /* 1 (default): a (point, obstacle) or (point, pair of vehicles) block whose closest features are two
                           *    vertices is constrained by their Euclidean distance -- the reference's OBCA rows admit any unit
                           *    direction (vehicle.py:523-541, multi_vehicle_planner.py:419-451); 0: face-normal certificates only
                           *    (a restriction at corner-to-corner contacts, kept to show the gap) */
  int32_t kernel;         /* must be 0: the field named the one-wavefront collocation kernel that round 4 retired (slower at every batch size,
                           *    and not deterministic: docs/notebook.md); any other value is an error since round 5 (the slot keeps the layout) */
  double shrink_tube;     /* :370; 0.5 in plan_single_path */
  double tol;             /* :650 1e-2 */
  double constr_viol_tol; /* :651 1e-2 */
  double mu_init;         /* 0.1 (IPOPT's default) */
  double curv_kappa;      /* 1e-8 */
  int32_t structured;     /* 0 or 1 (any other value is an error since round 6).  1 (default): the Newton system is eliminated interval by interval,
                           *    csrc/cfz_jstruct.inl -- vehicle-major ordering (a band of half-bandwidth 51 per vehicle, the condensed pair blocks of a
                           *    joint plan beside it), tube rows condensed, every vehicle's interiors (64 unknowns) by themselves on the matrix cores,
                           *    the pair-coupled poses of an interval index through a 64 x 64 capacitance system, a block recursion over separators of
                           *    at most 15 unknowns per vehicle that keeps every hand-off in registers; single plans (cfz_colloc) run the same scheme
                           *    with 16-row blocks.  0 (also what one_pivot = 1 takes, and what a plan of more than 255 intervals per vehicle is routed
                           *    to): pivot by pivot along the band, for a joint plan the band across the vehicles (half-bandwidth ~300, 88 MB per
                           *    four-vehicle plan) a panel at a time: same matrix, same solution to rounding, 3-6x slower.  (Round 4's scheme for single
                           *    plans, 2 until round 5, is gone: its separator recursion handed blocks from wavefront to wavefront through global
                           *    memory.) */
  int32_t reserved1;
} cfz_colloc_options;

void cfz_default_colloc_options(cfz_colloc_options *opt);
int cfz_colloc_w(cfz_plan_ws *ws, int B, const cfz_spec *spec, const cfz_colloc_options *opt, const int32_t *n_sets,
                 const double *init_pose, const double *final_heading, const double *tube, const double *guess,
                 const double *dt0, double *traj, double *dt, int32_t *status, int32_t *iters, double *cost);
int cfz_colloc(int device, int B, const cfz_spec *spec, const cfz_colloc_options *opt, const int32_t *n_sets,
               const double *init_pose, const double *final_heading, const double *tube, const double *guess,
               const double *dt0, double *traj, double *dt, int32_t *status, int32_t *iters, double *cost);

/* ---- MultiVehiclePlanner.solve_final_problem_obca (confrez/control/multi_vehicle_planner.py:343-480) -----------------------
 * The joint plan: the collocation problems of V vehicles (each as in cfz_colloc, from its own single-vehicle result) with ONE
 * shared interval length dt (:365-366), cost sum_a J_a (:387), and for every pair of vehicles and every collocation point
 * of the shorter plan the two bodies at least spec->dmin apart (:389-456).  B independent instances (scenarios) of V vehicles
 * each in one launch, one workgroup per instance (BASELINE.json configs[3]); the reference solves one.
 *   n_sets[B*V], init_pose[B*V][3], final_heading[B*V], tube, guess   per vehicle, instance-major, as in cfz_colloc
 *   dt0[B]                initial shared interval length (:361 the mean of the single results')
 *   pairs[n_pairs][2]     vehicle index pairs a < b with a separation row, the same for every instance; NULL = all pairs (:56-58)
 *   traj (out)            x, y, psi, v, delta, a, w at every point, vehicles back to back; dt (out) [B]
 *   status, iters, cost   per instance (may be NULL)
 * The vehicle-vehicle OBCA duals (:404-431) are eliminated like the obstacle duals (two smooth rows per pair and point over a
 * working set) and rebuilt from the poses by cfz_joint_dual_ws.  The vehicles' interval blocks are interleaved in time, which
 * widens the band to ~100 V; the elimination then runs from global memory (csrc/cfz_colloc.inl). */
int cfz_joint_colloc_w(cfz_plan_ws *ws, int B, int V, const cfz_spec *spec, const cfz_colloc_options *opt, const int32_t *n_sets,
                       const double *init_pose, const double *final_heading, const double *tube, const double *guess, const double *dt0,
                       int n_pairs, const int32_t *pairs, double *traj, double *dt, int32_t *status, int32_t *iters, double *cost);
int cfz_joint_colloc(int device, int B, int V, const cfz_spec *spec, const cfz_colloc_options *opt, const int32_t *n_sets,
                     const double *init_pose, const double *final_heading, const double *tube, const double *guess, const double *dt0,
                     int n_pairs, const int32_t *pairs, double *traj, double *dt, int32_t *status, int32_t *iters, double *cost);

/* Size of the banded primal-dual system one (joint) collocation plan is eliminated on (host arithmetic, no device): unknowns nk,
 * half-bandwidth kb of the ordering, bytes of the band as stored (LAPACK general-band layout, nk x (3 kb + 1) doubles).  V vehicles
 * with n_sets[V] strategy steps each, has_final[V] (NULL = all have a terminal heading), pairs as in cfz_joint_colloc (NULL = all
 * pairs; V = 1: none).  bench.py prices the planning kernels' HBM traffic with it.  Any output pointer may be NULL. */
int cfz_colloc_band_info(int V, const int32_t *n_sets, const int32_t *has_final, int N_per_set, int n_obs, int n_pairs,
                         const int32_t *pairs, int32_t *nk, int32_t *kb, int64_t *band_bytes);

/* The same for the elimination a plan actually goes through (`structured` as in cfz_colloc_options: 1 = interval by interval,
 * cfz_jstruct.inl -- vehicle-major ordering, half-bandwidth 51, tube rows condensed; 0 = along the band):
 * alg_bytes = the bytes one Newton system's elimination moves between its phases (every array written once and read where another phase
 * consumes it: csrc/cfz_jstruct.inl jstruct_alg_doubles; the band path: 3 x the band), what bench.py
 * prices the planning kernels' HBM traffic with; workspace_bytes = the plan's slab in HBM.  Any output pointer may be NULL. */
int cfz_colloc_elimination_info(int V, const int32_t *n_sets, const int32_t *has_final, int N_per_set, int n_obs, int n_pairs, const int32_t *pairs,
                                int structured, int32_t *nk, int32_t *kb, int64_t *band_bytes, int64_t *alg_bytes, int64_t *workspace_bytes);

/* Layout version of the structs of this header (cfz_spec, cfz_options, cfz_colloc_options, cfz_plan_options): a binding compares it with the
 * CFZ_ABI_VERSION it was written against before it passes a struct (conflict_rez_amd/engine.py does).  Round 5: 5; round 6: 6
 * (cfz_colloc_options.structured lost its value 2: a caller that still passes it is refused instead of silently taking another path). */
#define CFZ_ABI_VERSION 6
int cfz_abi_version(void);

/* ---- batched closed loop of MultiDistributedFollower.solve (:630-663) ---------------------
 * S scenarios x V vehicles (V = n_nbr + 1), B = S*V instances ordered [s][v].
 * ref_table[V][T][7]: each vehicle's planned trajectory (x,y,psi,v,delta,a,w) sampled every dt
 * (what get_current_ref :370-404 interpolates); scenario s starts at sample k0[s].
 * cfz_loop_init sets state = ref_table[v][k0][0:5] + noise[s][v][5] and the first prediction =
 * the plan at the horizon times (:397-400).
 * cfz_loop_step does one iteration for all scenarios, entirely on the device:
 *   neighbours' predictions copied first (get_others_pred :636-637, Jacobi), every vehicle's
 *   step(): parameters + shifted warm start (:432-476), solve (:479), read-back or shift
 *   fallback (:484-524), clock += dt (:526), plant integration over dt with (a0,w0) (:528-543). */
int cfz_loop_init(cfz_handle *h, int S, int T, const double *ref_table, const int32_t *k0, const double *noise);
int cfz_loop_step(cfz_handle *h);
/* K iterations for all scenarios in one persistent launch: the V solves of a scenario still exchange predictions
 * after every iteration (Jacobi), but scenarios no longer wait for each other between iterations.  Same results
 * as K calls of cfz_loop_step.  cfz_loop_last_iterations: IPM iterations summed over all solves of that call. */
int cfz_loop_run(cfz_handle *h, int K);
long cfz_loop_last_iterations(const cfz_handle *h);
/* solves of that call that converged (status 0); the others took the reference's shift fallback (:501-524) */
long cfz_loop_last_converged(const cfz_handle *h);
/* how the solves of that call ended: counts[s] = solves with status s (0 converged, 1 iteration limit, 2 line search, 3 non-finite,
 * 4 measured state infeasible (in collision or outside the boxes: no iteration), 5 stalled / locally infeasible) */
int cfz_loop_last_status_counts(const cfz_handle *h, long counts[6]);
/* state[S][V][5], pred[S][V][7][N], status int32[S][V] of the last step; NULL to skip. */
int cfz_loop_get(cfz_handle *h, double *state, double *pred, int32_t *status, int32_t *iters);

const char *cfz_last_error(void);
/* First 16 hex digits of the SHA-256 over the kernel sources this library was built from (__graft_entry__.source_hash; "unknown" for
 * a build made by hand).  bench.py quotes profiler counters from profiles/ only if they were taken on a library with the same hash. */
const char *cfz_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* CONFREZ_HIP_H */
