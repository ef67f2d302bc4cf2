"""Oracle (test infrastructure): primal-dual interior-point NLP solver, full-space sparse form.

The reference hands its NLPs to IPOPT through CasADi (`opti.solver("ipopt", ...)`,
`vehicle_follower.py:356-368`; un-pinned, absent here).  This file restates IPOPT's
published algorithm (Waechter & Biegler, "On the implementation of an interior-point
filter line-search algorithm for large-scale nonlinear programming", Math. Prog. 106,
2006) in the simplified, fully deterministic form that the HIP kernel implements
(DESIGN.md "CFZ-IPM"):

  kept from the paper   barrier problem and primal-dual equations (eq. 3-4), optimality
                        error E_mu with s_d/s_c scaling (eq. 5-6), filter line search (sec. 2.3),
                        monotone barrier update
                        (eq. 7) with kappa_eps/kappa_mu/theta_mu, fraction-to-the-boundary
                        (eq. 15), primal/dual step sizes (eq. 14), multiplier safeguard
                        (eq. 16), initial-point push (sec. 3.6), slack reformulation of
                        inequalities, termination on tol + constr_viol_tol + dual_inf_tol +
                        compl_inf_tol, mu floor = min(tol,compl_inf_tol)/(kappa_eps+1).
  replaced              exact Hessian + inertia correction -> Gauss-Newton Hessian (PSD, so with
                        the barrier terms the reduced system is positive definite by
                        construction; `hessian="exact"` keeps the paper's form for the planning
                        NLPs, whose objective has no state curvature);
                        restoration phase and second-order correction -> none (a failed line search is
                        reported as status 2; `VehicleFollower.step` then applies the reference's
                        own shift fallback, vehicle_follower.py:501-524); the filter keeps at most
                        `filter_cap` entries.

It works on the whole KKT matrix with a general sparse LU, i.e. it shares no linear
algebra with the structure-exploiting kernel; agreement of iterates between the two is
the check that the block elimination + Riccati recursion in the kernel is a correct
factorisation of the same Newton system.
"""
from dataclasses import dataclass
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


@dataclass
class IpmOptions:
    tol: float = 1e-2  # vehicle_follower.py:362
    constr_viol_tol: float = 1e-2  # :363
    max_iter: int = 600  # :364
    # Infeasibility stall test (stands in for the outcome of IPOPT's restoration phase, "converged to a point of
    # local infeasibility"): give up when the max-norm constraint violation has not dropped below stall_kappa x its
    # last checkpoint for stall_iters iterates (those that changed the working set of a working-set NLP count a quarter) while
    # still above constr_viol_tol.  0 disables.
    # Hessian of the Lagrangian = Gauss-Newton objective part + exact curvature of the separation rows weighted with
    # their multipliers (NLPs that provide hess_gn(x, nu)).  Without it the iteration is not contractive when a
    # vehicle is pushed hard against a separation row (period-2 oscillation, hundreds of iterations).
    row_curvature: bool = True
    # The convexity safeguard scales a stage's row curvature (hess_gn).  A scaled model can cycle for hundreds of iterations
    # on a vehicle pressed deep into a corner; from iteration shift_after on such a stage keeps the whole curvature and
    # is shifted by the smallest multiple of the identity instead (0 = never).  Solves that end earlier are untouched.
    shift_after: int = 60
    warm_push: float = 1e-6  # distance from a bound kept by a start that carries the previous solve's multipliers
    stall_kappa: float = 0.9
    stall_iters: int = 10
    dual_inf_tol: float = 1.0  # IPOPT default
    compl_inf_tol: float = 1e-4  # IPOPT default
    mu_init: float = 1e-3  # IPOPT's default is 0.1; the MPC warm start sits near the end of the central path and
    #                        1e-3 needs a third fewer iterations at unchanged failure rate and cost
    kappa_eps: float = 10.0
    kappa_mu: float = 0.2
    theta_mu: float = 1.5
    tau_min: float = 0.99
    bound_push: float = 1e-2
    bound_frac: float = 1e-2
    s_max: float = 100.0
    kappa_sigma: float = 1e10
    eta_phi: float = 1e-8
    gamma_theta: float = 1e-5
    gamma_phi: float = 1e-8
    delta_sw: float = 1.0
    s_theta: float = 1.1
    s_phi: float = 2.3
    filter_cap: int = 16
    max_backtrack: int = 25
    reg_primal: float = 1e-8
    reg_dual: float = 0.0
    # IPOPT's dual regularisation delta_c (paper sec. 3.1: -delta_c I in the (2,2) block of the Newton system whenever the Jacobian may
    # have lost rank), here permanently on the separation rows of a working-set NLP (`nlp.c_blk0`: first of those rows): two rows of a
    # block, or rows of different blocks of a stage, become dependent at a contact and their multipliers run away along the null space.
    reg_dual_rows: float = 1e-8
    # Restoration phase (paper sec. 3.3) of NLPs that provide `restore` (oracle/mpc_nlp.py): at most `restoration` calls per solve,
    # each after a failed line search at an iterate whose violation exceeds constr_viol_tol; the iteration resumes with cold multipliers
    # at the restored point.  A restoration that does not reach its goal ends the solve with status 5
    # (IPOPT: "converged to a point of local infeasibility").  0 = no restoration phase (a failed line search is status 2).
    restoration: int = 2
    # A start whose separation rows are violated by more than this (metres; the warm start of a vehicle whose neighbour's prediction
    # has moved into its path) goes through the restoration phase BEFORE the first iteration and starts with cold multipliers at the
    # restored point: the interior-point iteration from half a metre inside a clearance is where the multipliers run away.  0 = never.
    resto_first: float = 0.3
    hessian: str = "gn"  # "gn" (kernel's choice) or "exact" (needs nlp.hess_exact; planning NLPs)
    curv_kappa: float = 1e-8  # exact Hessian: inertia-free curvature test d'(W+Sigma)d >= kappa d'd
    # Stagnation shift: once the scaled optimality error has not halved for `shift_stagnation` iterations at a feasible iterate
    # (cviol <= constr_viol_tol), the late shift of the row curvature (see shift_after) starts as soon as iteration SHIFT_STAG_MIN
    # is reached instead of waiting for shift_after.  The sawtooth of the scaled model (planned-table closed loop: one vehicle at
    # 62-65 iterations per solve, every solve) ends 15-20 iterations sooner; solves that make progress are untouched, and the
    # independent-solver populations keep their outcomes (with a minimum below 40 instance 7 of the first one does not).  0 = off.
    shift_stagnation: int = 10
    # A solve whose scaled optimality error has not halved for this many iterations ends with status 5 (stalled) instead of running
    # to max_iter: limit cycles BELOW constr_viol_tol escape the violation-based stall test (planned-table closed loop: a period-3
    # cycle at violation 9.6e-3 ran 600 iterations, 50 ms on the critical path of a launch).  0 = off.
    err_stall_iters: int = 150
    # MPC closed loop: a solve in which some stage's curvature had to be shifted tells the next solve of the same vehicle (with the carried
    # multipliers, oracle/mpc_nlp.py carry_state) to shift from its first iteration instead of waiting for shift_after / stagnation: a
    # cornered vehicle otherwise repeats the 40+ iterations of the scaled model at every MPC iteration.  Cold solves are unaffected.
    # (planned-table closed loop, three sampler seeds: 99th percentile of a scenario's iteration chain 580 -> 370, mean 109 -> 98,
    # more solves converge; docs/notebook.md round 3.)  0 = off.
    carry_shift: int = 1
    lower_mu_on_failure: bool = False  # a failed line search lowers mu once instead of ending the solve (independent solvers only)


def next_delta_w(delta, last):
    """IPOPT's inertia correction, Algorithm IC of the paper (sec. 3.1), the primal perturbation: a Newton system that fails the
    curvature test at delta_w = 0 is tried again with delta_w^0 = 1e-4 if no iteration has needed a perturbation yet, else with a
    third of the last one that worked (kappa_w^-); every further failure multiplies by kappa_w^+ = 8.  (The paper multiplies by 100
    while no iteration has needed a perturbation yet; without a restoration phase to fall back on that overshoot -- delta_w = 100
    where 3 would do -- ends small plans in a failed line search, so the first round climbs by 8 as well.)
    `last`: the last nonzero perturbation of an earlier iteration (0: none yet)."""
    if delta == 0.0:
        return 1e-4 if last == 0.0 else max(1e-20, last / 3.0)
    return delta * 8.0


def kkt_inertia_ok(H, J, n, m):
    """True if [[H, J'], [J, 0]] has n positive and m negative eigenvalues (H's reduced Hessian on the null space of J is positive
    definite): dense LDL' (Bunch-Kaufman), eigenvalue signs of its 1 x 1 and 2 x 2 pivot blocks."""
    from scipy.linalg import ldl

    K = np.block([[H.toarray(), J.T.toarray()], [J.toarray(), np.zeros((m, m))]])
    _, D, _ = ldl(K)
    d0, d1 = np.diag(D).copy(), np.diag(D, -1)
    neg = 0
    i = 0
    while i < n + m:
        if i + 1 < n + m and d1[i] != 0.0:  # 2 x 2 block: one eigenvalue of each sign iff its determinant is negative
            det, tr = d0[i] * d0[i + 1] - d1[i] * d1[i], d0[i] + d0[i + 1]
            neg += 1 if det < 0 else (2 if tr < 0 else 0)
            if det == 0.0:
                return False
            i += 2
        else:
            if d0[i] == 0.0:
                return False
            neg += d0[i] < 0
            i += 1
    return neg == m


STATUS_OK, STATUS_MAXITER, STATUS_LINESEARCH, STATUS_NAN = 0, 1, 2, 3
STATUS_STALLED = 5  # constraint violation stopped decreasing above constr_viol_tol, or a restoration failed (4 is taken by mpc_nlp)


def push_to_interior(x, xl, xu, opt: IpmOptions):
    """IPOPT sec. 3.6 initial-point projection."""
    x = x.copy()
    hasl, hasu = np.isfinite(xl), np.isfinite(xu)
    both = hasl & hasu
    pl = np.where(hasl, opt.bound_push * np.maximum(1.0, np.abs(np.where(hasl, xl, 0.0))), 0.0)
    pu = np.where(hasu, opt.bound_push * np.maximum(1.0, np.abs(np.where(hasu, xu, 0.0))), 0.0)
    width = np.where(both, xu - xl, np.inf)
    pl = np.where(both, np.minimum(pl, opt.bound_frac * width), pl)
    pu = np.where(both, np.minimum(pu, opt.bound_frac * width), pu)
    x = np.where(hasl, np.maximum(x, xl + pl), x)
    x = np.where(hasu, np.minimum(x, xu - pu), x)
    return x


WS_STALL_DIV = 4
SHIFT_STAG_MIN = 40  # earliest iteration of a stagnation-triggered curvature shift (IpmOptions.shift_stagnation)


def solve(nlp, X0, opt: IpmOptions = IpmOptions(), trace=None, warm=None):
    """Returns dict(X, nu, zl, zu, status, iters, mu, err, f).
    warm: optional dict(zl, zu, nu, mu) -- start from these multipliers and barrier parameter instead of
    z = 1, nu = 0, mu = mu_init; X0 is then taken as it is (the caller has placed it inside the bounds)."""
    xl, xu = nlp.xl, nlp.xu
    hasl, hasu = np.isfinite(xl), np.isfinite(xu)
    n, m = nlp.n, nlp.m
    if warm is None:
        x = push_to_interior(np.asarray(X0, float), xl, xu, opt)
        zl = np.where(hasl, 1.0, 0.0)
        zu = np.where(hasu, 1.0, 0.0)
        nu = np.zeros(m)
        mu = opt.mu_init
    else:
        x = np.asarray(X0, float).copy()
        zl = np.where(hasl, np.asarray(warm["zl"], float), 0.0)
        zu = np.where(hasu, np.asarray(warm["zu"], float), 0.0)
        nu = np.asarray(warm["nu"], float).copy()
        mu = float(warm["mu"])
    mu_floor = min(opt.tol, opt.compl_inf_tol) / (opt.kappa_eps + 1.0)
    filt, filt_mu = None, None
    delta_w_last = 0.0
    theta_min = theta_max = None
    status = STATUS_MAXITER
    nb = int(hasl.sum() + hasu.sum())

    def dist(xx):
        dl = np.where(hasl, xx - np.where(hasl, xl, 0.0), 1.0)
        du = np.where(hasu, np.where(hasu, xu, 0.0) - xx, 1.0)
        return dl, du

    def barrier_obj(xx, mu_):
        dl, du = dist(xx)
        if (dl <= 0).any() or (du <= 0).any():
            return np.inf
        return nlp.f(xx) - mu_ * (np.log(dl[hasl]).sum() + np.log(du[hasu]).sum())

    dual_reg = np.full(m, float(opt.reg_dual))
    if hasattr(nlp, "dual_reg_rows"):  # an NLP that names the only rows able to lose rank takes delta_c on those alone (StateWsNlp)
        dual_reg = dual_reg * nlp.dual_reg_rows
    if opt.reg_dual_rows > 0.0 and hasattr(nlp, "c_blk0"):
        dual_reg[nlp.c_blk0:] += opt.reg_dual_rows
    it = 0
    err0 = np.inf
    stagnant, best_err, best_it = False, np.inf, 0
    shift_hint = bool(warm is not None and opt.carry_shift and warm.get("shift_hint"))  # IpmOptions.carry_shift
    shifted = False  # some stage's curvature was shifted in some iteration of this solve
    mu_forced = False
    it0 = 0
    resto_calls = 0
    can_restore = opt.restoration > 0 and hasattr(nlp, "restore")
    if can_restore and opt.resto_first > 0.0 and nlp.start_violation(x) > opt.resto_first:  # at the pushed start
        ok, x, it0 = nlp.restore(x, mu, opt, 0)
        if not ok:  # (a restoration that runs into the solve's iteration limit is the limit, status 1, not local infeasibility: ADVICE r4)
            return dict(X=x, nu=nu, zl=zl, zu=zu, status=1 if it0 >= opt.max_iter else STATUS_STALLED, iters=it0, mu=mu, err=np.inf, f=nlp.f(x), shifted=False)
        x, zl, zu, nu = nlp.cold_multipliers(x, mu, opt)
    it = it0 - 1
    while it < opt.max_iter:
        it += 1
        if it > it0 and hasattr(nlp, "new_iterate"):  # working-set NLPs refresh their rows here
            x, zl, nu = nlp.new_iterate(x, zl, nu, mu, opt.bound_push)
        g = nlp.grad(x)
        c, J = nlp._cons_jac(x, True)
        dl, du = dist(x)
        if theta_min is None:
            th0 = np.abs(c).sum()
            theta_min, theta_max = 1e-4 * max(1.0, th0), 1e4 * max(1.0, th0)
        # ---- optimality error (paper eq. 5) -----------------------------------------
        r_dual = g + J.T @ nu - zl + zu
        s_d = max(opt.s_max, (np.abs(nu).sum() + zl.sum() + zu.sum()) / max(m + nb, 1)) / opt.s_max
        s_c = max(opt.s_max, (zl.sum() + zu.sum()) / max(nb, 1)) / opt.s_max
        dual_inf = np.abs(r_dual).max()
        cviol = np.abs(c).max() if m else 0.0
        compl_l, compl_u = dl * zl * hasl, du * zu * hasu

        def compl_err(mu_):
            a = np.abs(compl_l - mu_)[hasl].max() if hasl.any() else 0.0
            b = np.abs(compl_u - mu_)[hasu].max() if hasu.any() else 0.0
            return max(a, b)

        def E(mu_):
            return max(dual_inf / s_d, cviol, compl_err(mu_) / s_c)

        err0 = E(0.0)
        if not np.isfinite(err0):
            status = STATUS_NAN
            break
        if trace is not None:
            trace.append(dict(it=it, mu=mu, err0=err0, dual_inf=dual_inf, cviol=cviol, compl=compl_err(0.0),
                              f=nlp.f(x), x=x.copy(), nu=nu.copy(), zl=zl.copy(), zu=zu.copy()))
        if (
            err0 <= opt.tol
            and dual_inf <= opt.dual_inf_tol
            and cviol <= opt.constr_viol_tol
            and compl_err(0.0) <= opt.compl_inf_tol
        ):
            status = STATUS_OK
            break
        if it == opt.max_iter:
            break
        if it == it0 or err0 < 0.5 * best_err:
            best_err, best_it = err0, it
        if opt.shift_stagnation > 0 and not stagnant and cviol <= opt.constr_viol_tol and it - best_it >= opt.shift_stagnation:
            stagnant = True
        if opt.err_stall_iters > 0 and it - best_it >= opt.err_stall_iters:
            status = STATUS_STALLED
            break
        if it == it0 or cviol <= opt.stall_kappa * stall_ref:
            stall_ref, stall_cnt, stall_ws = cviol, 0, 0
        elif not getattr(nlp, "ws_changed", False):
            stall_cnt += 1
        else:  # an iterate that changed the working set counts a quarter: its new rows start with their own violation, but a
            stall_ws += 1  # solve that changes it at EVERY iterate cycles and must end
            if stall_ws >= WS_STALL_DIV:
                stall_ws, stall_cnt = 0, stall_cnt + 1
        if opt.stall_iters > 0 and stall_cnt >= opt.stall_iters and cviol > opt.constr_viol_tol:
            status = STATUS_STALLED
            break
        # ---- barrier update (paper eq. 7) ---------------------------------------------
        while mu > mu_floor and E(mu) <= opt.kappa_eps * mu:
            mu = max(mu_floor, min(opt.kappa_mu * mu, mu**opt.theta_mu))
        tau = max(opt.tau_min, 1.0 - mu)
        # ---- Newton step on the primal-dual equations (paper eq. 11-13) ---------------
        sig = zl / dl * hasl + zu / du * hasu
        gphi = g - mu / dl * hasl + mu / du * hasu  # gradient of the barrier objective
        rhs = -np.concatenate([gphi + J.T @ nu, c])
        if opt.hessian == "gn":
            if opt.row_curvature and getattr(nlp, "has_row_curvature", False):
                late = opt.shift_after > 0 and (it >= opt.shift_after or (stagnant and it >= SHIFT_STAG_MIN) or shift_hint)
                H = nlp.hess_gn(x, nu, shift=late) + sp.diags(sig + opt.reg_primal)
                shifted = shifted or bool(getattr(nlp, "shift_applied", False))
            else:
                H = nlp.hess_gn(x) + sp.diags(sig + opt.reg_primal)
            K = sp.bmat([[H, J.T], [J, -sp.diags(dual_reg)]], format="csc")
            sol = spla.splu(K).solve(rhs)
            dx, dnu = sol[:n], sol[n:]
        else:
            # exact Hessian; IPOPT's inertia correction (paper sec. 3.1) restated with the
            # inertia-free curvature test of Chiang & Zavala (2016): raise delta_w until
            # the step sees positive curvature.
            W = nlp.hess_exact(x, nu) + sp.diags(sig)
            trial_delta = 0.0
            while True:
                H = W + (trial_delta + opt.reg_primal) * sp.eye(n)
                K = sp.bmat([[H, J.T], [J, -sp.diags(dual_reg)]], format="csc")
                sol = spla.splu(K).solve(rhs)
                dx, dnu = sol[:n], sol[n:]
                if float(dx @ (H @ dx)) >= opt.curv_kappa * float(dx @ dx) and np.isfinite(sol).all():
                    break
                trial_delta = next_delta_w(trial_delta, delta_w_last)
                if trial_delta > 1e20:
                    break
            delta_w_last = trial_delta if trial_delta > 0 else delta_w_last
        dzl = (mu / dl - zl - zl / dl * dx) * hasl
        dzu = (mu / du - zu + zu / du * dx) * hasu
        # ---- fraction to the boundary (paper eq. 15) -----------------------------------
        a_pri = 1.0
        neg = hasl & (dx < 0)
        if neg.any():
            a_pri = min(a_pri, (-tau * dl[neg] / dx[neg]).min())
        pos = hasu & (dx > 0)
        if pos.any():
            a_pri = min(a_pri, (tau * du[pos] / dx[pos]).min())
        a_dual = 1.0
        neg = hasl & (dzl < 0)
        if neg.any():
            a_dual = min(a_dual, (-tau * zl[neg] / dzl[neg]).min())
        neg = hasu & (dzu < 0)
        if neg.any():
            a_dual = min(a_dual, (-tau * zu[neg] / dzu[neg]).min())
        # ---- filter line search (paper sec. 2.3, Algorithm A steps A-5.*) ------------------
        theta = np.abs(c).sum()
        phi0 = barrier_obj(x, mu)
        dphi = float(gphi @ dx)
        if filt is None or filt_mu != mu:  # the filter is re-initialised for every barrier problem
            filt, filt_mu = [], mu
        if getattr(nlp, "ws_changed", False):  # ... and when the working set (hence the problem the entries belong to) changed
            filt = []
        alpha = a_pri
        accepted = False
        for _ in range(opt.max_backtrack):
            xt = x + alpha * dx
            th_t = np.abs(nlp.cons(xt)).sum()
            ph_t = barrier_obj(xt, mu)
            ok = np.isfinite(ph_t) and np.isfinite(th_t) and th_t <= theta_max
            if ok:
                for th_f, ph_f in filt:
                    if th_t >= th_f and ph_t >= ph_f:
                        ok = False
                        break
            f_type = False
            if ok:
                switching = (
                    theta <= theta_min
                    and dphi < 0.0
                    and alpha * (-dphi) ** opt.s_phi > opt.delta_sw * theta**opt.s_theta
                )
                if switching:
                    f_type = True
                    ok = ph_t <= phi0 + opt.eta_phi * alpha * dphi
                else:
                    ok = th_t <= (1.0 - opt.gamma_theta) * theta or ph_t <= phi0 - opt.gamma_phi * theta
            if ok:
                accepted = True
                break
            alpha *= 0.5
        if not accepted and can_restore and resto_calls < opt.restoration and cviol > opt.constr_viol_tol:
            # IPOPT's answer to a failed line search: the restoration phase, then on with an empty filter and cold multipliers
            ok, x, rit = nlp.restore(x, mu, opt, it)
            it += rit
            if not ok:
                status = 1 if it >= opt.max_iter else STATUS_STALLED
                break
            resto_calls += 1
            x, zl, zu, nu = nlp.cold_multipliers(x, mu, opt)
            filt, stall_ref, stall_cnt, stall_ws, best_err, best_it = [], np.inf, 0, 0, np.inf, it
            continue
        if not accepted:
            # No restoration phase for this NLP (or its calls are used up, or the iterate is feasible).  The independent solvers of oracle/independent_*.py (`lower_mu_on_failure`) give up the
            # barrier problem at hand instead: mu falls, the filter starts afresh, the iterate stays; a second failure in a row
            # (or mu at its floor) ends the solve.  The MPC kernel's oracle path (default) reports status 2 at once, like the MPC kernel;
            # the PLANNING kernels (cfz_plan.inl, cfz_colloc.inl) do lower mu once and then retry the iterate with a larger inertia
            # perturbation -- their parity tests compare against the CPU build of their own source and against the independent solvers,
            # not against this loop.
            if opt.lower_mu_on_failure and mu > mu_floor and not mu_forced:
                mu = max(mu_floor, min(opt.kappa_mu * mu, mu**opt.theta_mu))
                mu_forced, filt, filt_mu = True, [], mu
                continue
            status = STATUS_LINESEARCH
            break
        mu_forced = False
        if not f_type:
            filt.append(((1.0 - opt.gamma_theta) * theta, phi0 - opt.gamma_phi * theta))
            if len(filt) > opt.filter_cap:
                filt.pop(0)
        if trace is not None:
            trace[-1].update(alpha=alpha, a_pri=a_pri, a_dual=a_dual, f_type=f_type, theta=theta, dphi=dphi)
        x = xt
        nu = nu + alpha * dnu
        zl = zl + a_dual * dzl
        zu = zu + a_dual * dzu
        # ---- multiplier safeguard (paper eq. 16) ----------------------------------------
        dl, du = dist(x)
        zl = np.where(hasl, np.clip(zl, mu / (opt.kappa_sigma * dl), opt.kappa_sigma * mu / dl), 0.0)
        zu = np.where(hasu, np.clip(zu, mu / (opt.kappa_sigma * du), opt.kappa_sigma * mu / du), 0.0)
    return dict(X=x, nu=nu, zl=zl, zu=zu, status=status, iters=it, mu=mu, err=err0, f=nlp.f(x), shifted=shifted)
