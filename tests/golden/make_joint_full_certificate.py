"""Generates tests/golden/joint_kernel_0123_d20_full.npz: the FULL-LENGTH four-vehicle joint plan (BASELINE configs[3]'s problem: 50 / 30 /
30 / 40 Radau intervals, six pairs, one shared dt; multi_vehicle_planner.py:343-480) as the GPU kernel solves it at TIGHT tolerances, with
a SOLVER-FREE KKT certificate on the INDEPENDENT statement of the problem (oracle/independent_joint.py: polygon distances instead of OBCA
duals or working sets; make_independent_joint.joint_kkt_certificate: bounded least squares for multipliers of the right sign that combine
the active gradients into the cost gradient).  VERDICT r5 item 7a: the independent solver does not converge on this problem (vehicle 0 at
full length), so the plan is judged by the certificate at the kernel's own plan, as `02_d20_s66` is.

Two steps, because the kernel needs the GPU and the certificate five minutes of a CPU:
    (GPU box)  python tools/joint_full_tight.py gpurun_out/joint_full_d20.npz 0.2 3000
    (here)     python tests/golden/make_joint_full_certificate.py gpurun_out/joint_full_d20.npz
Stored: the guess (the kernel's single plans at the reference's tolerance, on their mean dt: what `solve_final_problem_obca` starts from), the
tight plan with its status / iteration count / cost, its rows on the independent statement, the certificate's residual, the active pair
rows [(pair, point, multiplier)] and the number of vertex-vertex contacts among them; the plan at the reference's tolerance with its rows."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def statement(dmin):
    from conflict_rez_amd import scenarios
    from make_independent_joint import plans_of_strategy
    from oracle.independent_colloc import GeometricColloc

    plans = plans_of_strategy()
    agents = sorted(plans)
    sp = scenarios.parking_lot_spec(dmin=dmin)
    gs = [GeometricColloc(plans[a][1][0], plans[a][0], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=float(plans[a][1][-1, 2]), dmin=dmin) for a in agents]
    return gs, [(a, b) for a in range(4) for b in range(a + 1, 4)]


def rows_of(nlp, gs, pairs, z):
    eq = max(np.abs(g.eq(nlp.z_of(z, a))).max() for a, g in enumerate(gs))
    ineq = min(min(g.ineq(nlp.z_of(z, a)).min() for a, g in enumerate(gs)), min(nlp.pair_dist(z, a, b).min() for a, b in pairs) - nlp.dmin)
    return float(eq), float(ineq)


def main(dump):
    from make_independent_joint import joint_kkt_certificate, vertex_pair_contacts
    from oracle.independent_joint import GeometricJointIpm

    d = np.load(dump)
    dmin = float(d["dmin"])
    gs, pairs = statement(dmin)
    out = dict(dmin=dmin, dt0=float(d["dt0"]))
    for a in range(4):
        out[f"guess{a}"] = d[f"guess{a}"]
    for tag, pre in (("k", ""), ("p", "p")):
        trajs = [np.asarray(d[f"{pre}traj{a}"], float) for a in range(4)]
        dt = float(d[f"{pre}dt"])
        z = np.concatenate([t.ravel() for t in trajs] + [[dt]])
        nlp = GeometricJointIpm(gs, pairs, z)
        eq, ineq = rows_of(nlp, gs, pairs, z)
        for a in range(4):
            out[f"{tag}traj{a}"] = trajs[a]
        out.update({f"{tag}dt": dt, f"{tag}cost": float(nlp.f(z)), f"{tag}status": int(d[f"{pre}status"]), f"{tag}iters": int(d[f"{pre}iters"]), f"{tag}eq": eq, f"{tag}ineq": ineq})
        print(f"{'tight' if tag == 'k' else 'reference tolerance'}: status {out[tag + 'status']} iterations {out[tag + 'iters']} cost {out[tag + 'cost']:.9f} rows eq {eq:.2e} ineq {ineq:.2e}", flush=True)
        if tag == "k":
            t0 = time.time()
            res, _, _, active = joint_kkt_certificate(nlp, z)
            vv = vertex_pair_contacts(nlp, z, active)
            out.update(kcertificate=float(res), kactive=np.array([[e, pt, lam] for e, pt, lam in active], float), kcontacts=len(vv))
            print(f"   certificate {res:.3e}; {len(active)} active pair rows in pairs {sorted({int(e) for e, _, _ in active})}, {len(vv)} vertex-vertex contacts ({time.time() - t0:.0f} s)", flush=True)
    fn = os.path.join(HERE, "joint_kernel_0123_d20_full.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn), "bytes")


if __name__ == "__main__":
    main(sys.argv[1])
