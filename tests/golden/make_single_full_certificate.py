"""Generates tests/golden/colloc_kernel_v0_full.npz: vehicle 0's single-vehicle collocation plan at FULL length (50 Radau intervals, six
obstacles; vehicle.py:360-661) as the GPU kernel solves it at tight tolerances, with the solver-free KKT certificate on the INDEPENDENT
statement (oracle/independent_colloc.py; make_independent_colloc_vv.kkt_certificate).  VERDICT r5 item 7a.  Neither the independent solver
nor the kernel reaches 1e-8 on this plan: the kernel stops after ~2,200 iterations with rows at 1.5e-6 and a
certificate of 1.3e-5 -- a weaker statement than the joint fixture's 1.8e-7, stored as what it is.
    (GPU box)  python tools/single_full_tight.py gpurun_out/single_full_v0.npz vehicle_0 3000
    (here)     python tests/golden/make_single_full_certificate.py gpurun_out/single_full_v0.npz"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def statement(agent="vehicle_0"):
    from conflict_rez_amd import scenarios
    from make_independent_joint import plans_of_strategy
    from oracle.independent_colloc import GeometricColloc

    plans, sp = plans_of_strategy(), scenarios.parking_lot_spec()
    return GeometricColloc(plans[agent][1][0], plans[agent][0], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=float(plans[agent][1][-1, 2]), dmin=sp.dmin), plans


def main(dump):
    from make_independent_colloc_vv import kkt_certificate

    d = np.load(dump)
    agent = str(d["agent"])
    g, _ = statement(agent)
    out = dict(agent=agent, guess=d["guess"], dt0=float(d["dt0"]))
    for tag, pre in (("k", ""), ("p", "p")):
        z = np.append(np.asarray(d[f"{pre}traj"], float).ravel(), float(d[f"{pre}dt"]))
        out.update({f"{tag}traj": np.asarray(d[f"{pre}traj"], float), f"{tag}dt": float(d[f"{pre}dt"]), f"{tag}cost": float(g.cost(z)), f"{tag}status": int(d[f"{pre}status"]),
                    f"{tag}iters": int(d[f"{pre}iters"]), f"{tag}eq": float(np.abs(g.eq(z)).max()), f"{tag}ineq": float(g.ineq(z).min())})
        print(tag, {k: out[k] for k in out if k.startswith(tag) and not k.endswith("traj")}, flush=True)
        if tag == "k":
            t0 = time.time()
            res, _, lam, act = kkt_certificate(g, z)
            out.update(kcertificate=float(res), kactive=np.stack([act.astype(float), lam], 1))
            print(f"   certificate {res:.3e}, {len(act)} active rows ({time.time() - t0:.0f} s)", flush=True)
    fn = os.path.join(HERE, "colloc_kernel_v0_full.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn), "bytes")


if __name__ == "__main__":
    main(sys.argv[1])
