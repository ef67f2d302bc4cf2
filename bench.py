#!/usr/bin/env python3
"""Headline benchmark: OBCA MPC-step solves/sec (4 vehicles, N=30) -- BASELINE.json config 3.

A "step" is one closed-loop iteration of the distributed MPC (`MultiDistributedFollower.solve`,
reference vehicle_follower.py:630-663) for S = 1024 scenarios x 4 vehicles = 4096 NLP solves
per GPU, executed entirely on the device: the K timed steps are ONE persistent launch (`cfz_loop_run`: parameters +
shifted warm start, solve, read-back / fallback, plant integration, Jacobi exchange inside each scenario; `--mode step`
takes one `cfz_loop_step` launch per iteration instead).  Inputs are resident in HBM before the timed
region.  N GPUs = N independent shards of scenarios (weak scaling, no data-path collective).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit, `roofline`
(algorithmic HBM bytes of SURVEY.md 8d / measured solver-kernel time, plus the FP64 and VALU-active fractions),
`value` = CONVERGED solves per second (status 0; `value_all` counts every solve of the timed region, also those that did no work:
`config.status_counts`), `cold_step` (the first MPC iteration, cold multipliers), `extra.seeds` (the same measurement on three sampler
seeds, outside the timed region: the launch lasts as long as its slowest scenario, a maximum of 1024 draws), `extra.all_moving` (no
parked vehicle in the sample; `config.parked_fraction` says how many solves of the headline are for one) and `cpu_baseline`
(oracle/cfz_port.c, the plain-C port, running the SAME closed loop on this host's usable cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SOLVE = 8 * (365 + 2 * 2550)  # parameters + warm start in, solution out
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


FLOP_PER_IPM_ITERATION = 0.75e6  # SURVEY.md 8(d)'s ESTIMATE of the algorithm's flops: ~25 kflop per stage and iteration x 30 stages (blocks,
                                  # RK4 + sensitivities, Riccati); quoted only when no counter pass of the loaded library is committed -- see profiled_fp64
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)


def _cpu_closed_loop_worker(args):
    """One process of the CPU baseline: the SAME closed loop the GPU runs (`warmup` + `steps` MPC iterations, multipliers
    carried from one iteration to the next) for `n_scen` scenarios of the same sampler, through the oracle's plain-C port
    (oracle/closed_loop.py).  Returns solves, IPM iterations and converged solves of the timed iterations, and their time."""
    seed, n_scen, warmup, steps, ref_kind, feasible = args
    from conflict_rez_amd import scenarios
    from oracle.closed_loop import replay
    from oracle.mpc_nlp import MpcSpec

    spec = scenarios.parking_lot_spec()
    # ref_kind: "planned" / "state_ws" (package data) or the path of an .npz with the table the GPU run uses (--reference replan)
    table = np.load(ref_kind)["table"] if ref_kind.endswith(".npz") else scenarios.load_reference_table(kind=ref_kind)[0]
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=spec.n_nbr)
    k0, noise = scenarios.sample_scenarios(n_scen, table, seed=seed, spec=spec if feasible else None)
    n = its = ok = 0
    cold = None
    t0 = time.perf_counter()
    for t, (_, _, status, iters) in enumerate(replay(ospec, table, k0, noise, warmup + steps, dt=spec.dt, wb=spec.wb)):
        if t == 0:
            cold = (status.size, int(iters.sum()), time.perf_counter() - t0)
        if t == warmup - 1:
            t0 = time.perf_counter()
        if t >= warmup:
            n += status.size; its += int(iters.sum()); ok += int((status == 0).sum())
    return n, its, ok, time.perf_counter() - t0, cold


def usable_cores():
    """(cores this job can really use, description): the affinity mask capped by the cgroup's CPU quota (a GPU box shows 256 logical
    cores to a job whose quota is a fraction of them: 64 processes were measured slower than 16 there), and the CPU model."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    n = aff if quota is None else max(1, min(aff, int(np.ceil(quota))))
    return n, {"logical": os.cpu_count(), "affinity": aff, "cgroup_quota_cores": quota, "model": model}


def casadi_probe():
    """the reference's own CPU path is CasADi + IPOPT.  Report what importing it on THIS host does."""
    try:
        import casadi  # noqa: F401

        return f"casadi {casadi.__version__} importable on this host (the reference NLP is not timed here: no recorded strategy file)"
    except Exception as e:  # noqa: BLE001
        return f"CasADi unavailable on this host: {type(e).__name__}: {e}"


def cpu_baseline(warmup, steps, n_scen_per_core=32, max_cores=64, ref_kind="planned", feasible=True):
    """`cpu_baseline` of the bench line: the oracle's C port ("kind": "port") on the host cores, like for like with the
    GPU's timed region -- the closed loop after `warmup` iterations, carried multipliers -- on a bounded sample of the same
    scenario sampler (about 512 scenarios x 4 vehicles in all, at most 32 per process).  One process per USABLE core (`usable_cores`:
    affinity mask capped by the cgroup quota; the GPU boxes show 256 logical cores to a job that may use a fraction of them -- 16
    processes measured 17.4 k solves/s there, 64 processes 15.6 k), at most `max_cores`; the CPU model is reported with it."""
    import concurrent.futures as cf
    import multiprocessing as mp

    usable, cpu_info = usable_cores()
    cores = max(1, min(max_cores, usable))
    n_scen_per_core = max(4, min(n_scen_per_core, 512 // cores))  # the sample stays bounded: about 512 scenarios in all
    jobs = [(2024 + 1000 * i, n_scen_per_core, warmup, steps, ref_kind, feasible) for i in range(cores)]
    t_wall = time.perf_counter()
    try:
        with cf.ProcessPoolExecutor(cores, mp_context=mp.get_context("spawn")) as pool:  # spawn: this process holds a GPU context
            res = list(pool.map(_cpu_closed_loop_worker, jobs, timeout=600))
    except Exception as e:  # noqa: BLE001 - reporting only
        res = [_cpu_closed_loop_worker(jobs[0])]
        cores = 1
        note = f"process pool failed ({type(e).__name__}); one process"
    else:
        note = None
    t_wall = time.perf_counter() - t_wall
    n = sum(r[0] for r in res); its = sum(r[1] for r in res); ok = sum(r[2] for r in res)
    slow = max(r[3] for r in res)  # the slowest worker sets the rate (start-up and imports are not counted)
    cn = sum(r[4][0] for r in res); ci = sum(r[4][1] for r in res); ct = max(r[4][2] for r in res)
    out = {"value": n / slow, "unit": "solves/s", "cores": cores, "kind": "port",
           "value_converged": ok / slow, "mean_ipm_iters": its / max(n, 1), "ipm_iterations_per_s": its / slow,
           "per_core": n / slow / cores,
           "sample": f"closed loop, {cores} processes x {n_scen_per_core} scenarios x 4 vehicles, {warmup} warm-up + {steps} timed MPC "
                     f"iterations with carried multipliers = {n} timed solves, slowest worker {slow:.1f} s (wall incl. start-up {t_wall:.0f} s); "
                     f"{os.cpu_count()} logical cores on the host, {usable} usable by this job (affinity mask and cgroup quota)",
           "cpu": cpu_info,
           "cold_step": {"value": cn / ct, "unit": "solves/s", "mean_ipm_iters": ci / max(cn, 1),
                         "sample": f"first MPC iteration of the same scenarios (cold multipliers), {cn} solves"},
           "casadi": casadi_probe(),
           "note": "the reference's CasADi/IPOPT/MA97 path cannot travel to this host (not installable: no network); its implied range "
                   "is 10-90 ms per solve = 11-100 solves/s per core (BASELINE.md, unpublished, read off plot limits)"}
    if note:
        out["note"] = note + "; " + out["note"]
    return out


def profile_is_current(tag_path):
    """True if profiles/<tag>_meta.json says the profile was taken on a library built from the same kernel sources as the one loaded
    now (`cfz_source_hash`); a profile without meta file (rounds 1-2) or of other sources is stale: its counters are not quoted."""
    try:
        from conflict_rez_amd import engine

        meta = json.load(open(tag_path + "_meta.json"))
        return meta.get("csrc_sha16") == engine.source_hash() != "unknown"
    except (OSError, ValueError):
        return False


def profiled_traffic(kernel, require_current=True):
    """HBM bytes per launch of `kernel` from the committed PMC summaries of this same command (profiles/, separate
    --pmc passes of rocprofv3): raw FETCH_SIZE + WRITE_SIZE in KiB of the timed (last) dispatch.  None if absent."""
    import glob

    here = os.path.dirname(os.path.abspath(__file__))
    tags = sorted(t for t in glob.glob(os.path.join(here, "profiles", "*_pmc_FETCH_SIZE.csv")) if os.path.basename(t).count("_") == 3)  # <tag>_pmc_FETCH_SIZE.csv: the MPC bench's tags, not <tag>_planning_.. / <tag>_jointbatch_..
    if not tags:
        return None, None
    tag = tags[-1][: -len("_pmc_FETCH_SIZE.csv")]
    if require_current and not profile_is_current(tag):
        return None, os.path.basename(tag) + " (stale: taken on other kernel sources)"
    tot = 0.0
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            for line in open(f"{tag}_pmc_{c}.csv"):
                f = line.rstrip("\n").split(",")
                if f[0] == kernel and f[1] == c:
                    tot += float(f[4].split()[-1]) * 1024.0
    except (OSError, ValueError, IndexError):
        return None, None
    return (tot or None), os.path.basename(tag)


def profiled_sq(kernel, require_current=True):
    """VALU-active fraction of `kernel` from the committed SQ pass of this same command (profiles/*_pmc_SQ.csv):
    SQ_ACTIVE_INST_VALU (quad-cycles in which a SIMD executes a vector instruction, summed over SIMDs) x 4 /
    (GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / 8 x 1024 SIMDs), timed (last) dispatch.  (None, None) if absent."""
    import glob

    here = os.path.dirname(os.path.abspath(__file__))
    tags = sorted(t for t in glob.glob(os.path.join(here, "profiles", "*_pmc_SQ.csv")) if os.path.basename(t).count("_") == 2)
    if not tags:
        return None, None
    if require_current and not profile_is_current(tags[-1][: -len("_pmc_SQ.csv")]):
        return None, os.path.basename(tags[-1])[: -len("_pmc_SQ.csv")] + " (stale: taken on other kernel sources)"
    vals = {}
    try:
        for line in open(tags[-1]):
            f = line.rstrip("\n").split(",")
            if f[0] == kernel:
                vals[f[1]] = float(f[4].split()[-1])
        raw = vals["SQ_ACTIVE_INST_VALU"] / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    except (OSError, ValueError, IndexError, KeyError, ZeroDivisionError):
        return None, None
    # MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles, so busy cycles = 4 x the counter; `raw` is the plain quotient
    return {"frac": 4.0 * raw, "raw_quotient": raw}, os.path.basename(tags[-1])[: -len("_pmc_SQ.csv")]


def fp64_from_counters(c):
    """Counter values of one dispatch (profiles/*_pmc_FP64.csv, *_pmc_MFMA.csv) -> the executed FP64 work.  The SQ_INSTS_VALU_* counters
    count WAVE-level instructions: flops = (2 FMA + ADD + MUL + TRANS) x 64 lane-slots (an upper bound on useful flops: lanes that an exec
    mask switches off, and values the four lanes of a stage compute redundantly, are counted) + SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 (the
    counter's own unit, rocprofv3 -L: MfmaFlopsF64).  mfma_busy_frac: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) over
    GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024 SIMDs.  valu_useful_frac: FP64 arithmetic instructions (vector and matrix) among
    all vector instructions issued."""
    f64 = c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
    vflop = (2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) * 64.0
    mflop = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
    out = {"flops": vflop + mflop, "vector_flops": vflop, "mfma_flops": mflop, "mfma_instructions": c.get("SQ_INSTS_VALU_MFMA_F64", 0.0),
           "valu_instructions": c["SQ_INSTS_VALU"],
           "valu_useful_frac": (f64 + c.get("SQ_INSTS_VALU_MFMA_F64", 0.0)) / c["SQ_INSTS_VALU"] if c["SQ_INSTS_VALU"] else None,
           "mfma_busy_frac": None}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE_MFMA"):
        out["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE_MFMA"] / 8.0 * 1024.0)
    return out


def _pmc_rows(fn, kernel):
    """{counter: [value per dispatch]} of `kernel` from one profiles/*_pmc_*.csv."""
    vals = {}
    for line in open(fn):
        f = line.rstrip("\n").split(",")
        if f[0] == kernel and len(f) >= 5:
            vals[f[1]] = [float(x) for x in f[4].split()]
    return vals


def profiled_fp64(kernel="loop_kernel", require_current=True):
    """Executed FP64 work of the timed (last) dispatch of `kernel` from the committed FP64 / MFMA counter passes of the default command
    (profiles/<tag>_pmc_FP64.csv, <tag>_pmc_MFMA.csv; tools/gpu_profile_job.sh), with the interior-point iterations of that launch from
    <tag>_meta.json -> (dict of fp64_from_counters + flop_per_ipm_iteration, tag) or (None, reason)."""
    import glob

    here = os.path.dirname(os.path.abspath(__file__))
    tags = sorted(t for t in glob.glob(os.path.join(here, "profiles", "*_pmc_FP64.csv")) if os.path.basename(t).count("_") == 2)
    if not tags:
        return None, None
    tag = tags[-1][: -len("_pmc_FP64.csv")]
    if require_current and not profile_is_current(tag):
        return None, os.path.basename(tag) + " (stale: taken on other kernel sources)"
    try:
        c = {k: v[-1] for k, v in _pmc_rows(tag + "_pmc_FP64.csv", kernel).items()}
        try:
            mf = {k: v[-1] for k, v in _pmc_rows(tag + "_pmc_MFMA.csv", kernel).items()}
            c["SQ_VALU_MFMA_BUSY_CYCLES"] = mf["SQ_VALU_MFMA_BUSY_CYCLES"]; c["GRBM_GUI_ACTIVE_MFMA"] = mf["GRBM_GUI_ACTIVE"]
        except (OSError, KeyError, IndexError):
            pass
        out = fp64_from_counters(c)
        its = json.load(open(tag + "_meta.json")).get("ipm_iterations_timed_launch")
        out["ipm_iterations"] = its
        out["flop_per_ipm_iteration"] = out["flops"] / its if its else None
    except (OSError, ValueError, IndexError, KeyError):
        return None, None
    return out, os.path.basename(tag)


def profiled_extras(dispatch, kernel="colloc_kernel", require_current=True):
    """HBM bytes and VALU-active fraction of ONE dispatch of `kernel` from the committed PMC passes of `python bench.py --extras-only`
    (profiles/<tag>_extras_pmc_*.csv, made by tools/gpu_profile_extras.sh; hash-gated like the headline's): dispatch = the ordinal of the
    launch among the kernel's dispatches of that command (planning_extras records it).  -> dict(traffic, valu_active_frac, source) or None."""
    import glob

    here = os.path.dirname(os.path.abspath(__file__))
    tags = sorted(glob.glob(os.path.join(here, "profiles", "*_extras_pmc_FETCH_SIZE.csv")))
    if not tags:
        return None
    tag = tags[-1][: -len("_pmc_FETCH_SIZE.csv")]
    if require_current and not profile_is_current(tag):
        return {"traffic": None, "valu_active_frac": None, "source": os.path.basename(tag) + " (stale: taken on other kernel sources)"}
    # the dispatch ordinals below are those of THIS file's sequence of planning launches: a profile taken with another bench.py may have
    # another sequence, so the meta file carries bench.py's hash too and a mismatch is stale like another kernel source
    try:
        import hashlib

        want = json.load(open(tag + "_meta.json")).get("bench_sha16")
        if require_current and want is not None and want != hashlib.sha256(open(os.path.abspath(__file__), "rb").read()).hexdigest()[:16]:
            return {"traffic": None, "valu_active_frac": None, "source": os.path.basename(tag) + " (stale: taken with another bench.py, the dispatch ordinals may differ)"}
    except (OSError, ValueError):
        pass

    def per_dispatch(fn):
        vals = {}
        for line in open(fn):
            f = line.rstrip("\n").split(",")
            if f[0] == kernel and len(f) >= 5:
                vals[f[1]] = [float(x) for x in f[4].split()]
        return vals

    try:
        tot = sum(per_dispatch(f"{tag}_pmc_{c}.csv")[c][dispatch] for c in ("FETCH_SIZE", "WRITE_SIZE")) * 1024.0
        sq = per_dispatch(f"{tag}_pmc_SQ.csv")
        valu = 4.0 * sq["SQ_ACTIVE_INST_VALU"][dispatch] / (sq["GRBM_GUI_ACTIVE"][dispatch] / 8.0 * 1024.0)
    except (OSError, ValueError, IndexError, KeyError, ZeroDivisionError):
        return None
    out = {"traffic": tot, "valu_active_frac": valu, "source": os.path.basename(tag)}
    try:  # the FP64 / MFMA passes (round 6): executed flops, the matrix pipe's busy fraction, FP64 arithmetic among the vector instructions
        c = {k: v[dispatch] for k, v in per_dispatch(f"{tag}_pmc_FP64.csv").items()}
        mf = {k: v[dispatch] for k, v in per_dispatch(f"{tag}_pmc_MFMA.csv").items()}
        c["SQ_VALU_MFMA_BUSY_CYCLES"] = mf["SQ_VALU_MFMA_BUSY_CYCLES"]; c["GRBM_GUI_ACTIVE_MFMA"] = mf["GRBM_GUI_ACTIVE"]
        fp = fp64_from_counters(c)
        out.update(fp64_flops=fp["flops"], mfma_flops=fp["mfma_flops"], mfma_busy_frac=fp["mfma_busy_frac"], valu_useful_frac=fp["valu_useful_frac"])
    except (OSError, ValueError, IndexError, KeyError, ZeroDivisionError):
        pass
    return out


TAU5 = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])  # Radau-5 nodes


def planning_extras(device=0, B=256, cpu=True):
    """BASELINE.json configs[1] and configs[3], measured OUTSIDE the timed region of the headline (same process, same GPU):
    configs[1]  B single-vehicle plans (`Vehicle.state_ws` -> collocation plan, vehicle.py:99-231, :360-661): the four vehicles of the
                synthetic strategy in turn, start poses scattered by +-3 cm; one `cfz_state_ws` and one `cfz_colloc` launch -- with
                BASELINE.json's FOUR polytope obstacles (the object's own line), with the reference's six (`six_obstacles`), and stopped
                at the iteration count 95 % of the plans need (`p95`);
    configs[3]  B four-vehicle joint plans (`solve_final_problem_obca`, multi_vehicle_planner.py:343-480) from those single plans,
                one `cfz_joint_colloc` launch (one workgroup per plan).
    Times are wall-clock around the C-ABI calls (host buffers in and out: the transfers are megabytes, the launches seconds).
    `roofline`: algorithmic HBM bytes = what the structured elimination of one Newton system moves between its phases
    (`cfz_colloc_elimination_info`: every array written once and read where another phase consumes it) per interior-point iteration,
    summed over the plans' iteration counts, over the launch time, against the 8 TB/s roof; beside it, from the committed counter passes of
    `python bench.py --extras-only` on the same sources and the same bench.py: measured HBM bytes, VALU-active fraction, executed FP64
    flops, the matrix pipe's busy fraction.
    `cpu_baseline` ("port": the CPU build of the same kernel source, tests/emu, one core): one plan of every vehicle / one joint plan."""
    import tempfile

    from conflict_rez_amd import engine, scenarios, strategy as strat
    from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
    from conflict_rez_amd.vehicle_types import VehicleBody

    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    agents = sorted(hist)
    tubes = {a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents}
    fh = {a: float(paths[a][-1, 2]) for a in agents}
    sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
    rng = np.random.default_rng(0)
    who = [agents[i % 4] for i in range(4 * B)]  # B scenarios x 4 vehicles; the first B entries are configs[1]'s plans
    init = [paths[a][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]

    def guess_of(ws, n_sets, nps=5):  # interp_ws_for_collocation (vehicle.py:298-358) + dt0 = t_end / N (:388)
        N = nps * (n_sets - 1)
        t = 0.1 * np.arange(len(ws))
        ti = (np.arange(N)[:, None] + TAU5[None, :]).ravel() / N * t[-1]
        return np.stack([np.interp(ti, t, ws[:, c]) for c in range(7)], 1), t[-1] / N

    launches = {"colloc_kernel": 0}  # ordinal of the next colloc_kernel dispatch of this process (profiled_extras looks it up)

    def single_plans(idx):
        t0 = time.perf_counter()
        ws = engine.state_ws([init[i] for i in idx], [tubes[who[i]] for i in idx], [paths[who[i]] for i in idx], [fh[who[i]] for i in idx],
                             shrink_tube=0.5, device=device)
        t1 = time.perf_counter()
        good = [k for k, w in enumerate(ws) if w["status"] == 0]
        gs = {k: guess_of(ws[k]["traj"], len(tubes[who[idx[k]]]) + 1) for k in good}
        t2 = time.perf_counter()
        rg = engine.colloc(sp0, [init[idx[k]] for k in good], [tubes[who[idx[k]]] for k in good], [gs[k][0] for k in good],
                           [gs[k][1] for k in good], [fh[who[idx[k]]] for k in good], max_iter=400, device=device)
        t3 = time.perf_counter()
        launches["colloc_kernel"] += 1
        return ws, good, dict(zip(good, rg)), t1 - t0, t3 - t2

    out = {}
    # ---- configs[1] ------------------------------------------------------------------------------------------------------------
    single_plans(list(range(8)))  # warm-up: module load, workspace allocation
    ws, good, plans, t_ws, t_col = single_plans(list(range(B)))
    d6 = launches["colloc_kernel"] - 1
    info1 = {a: engine.colloc_elimination_info([len(tubes[a]) + 1]) for a in agents}
    alg = float(sum(info1[who[k]]["alg_bytes"] * plans[k]["iters"] for k in plans))
    ok = sum(r["status"] == 0 for r in plans.values())

    def roofline_of(alg_bytes, seconds, dispatch, what):
        pe = profiled_extras(dispatch)
        return {"bound": "hbm", "kernel": "colloc_kernel", "achieved": alg_bytes / seconds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / seconds / 1e9 / HBM_PEAK_GBS, "alg_bytes": alg_bytes, "alg_bytes_definition": what,
                "traffic": pe["traffic"] if pe else None, "valu_active_frac": pe["valu_active_frac"] if pe else None,
                "mfma_busy_frac": pe.get("mfma_busy_frac") if pe else None, "valu_useful_frac": pe.get("valu_useful_frac") if pe else None,
                "fp64_tflops": (pe["fp64_flops"] / seconds / 1e12) if pe and pe.get("fp64_flops") else None,
                "mfma_flops_share": (pe["mfma_flops"] / pe["fp64_flops"]) if pe and pe.get("fp64_flops") else None,
                "traffic_source": (pe["source"] + f", colloc_kernel dispatch {dispatch} of `python bench.py --extras-only`") if pe else None}
    what1 = ("bytes the structured elimination of one Newton system moves between its phases (cfz_colloc_elimination_info: "
             "%s per vehicle) x interior-point iterations of every plan" % {a: info1[a]["alg_bytes"] for a in agents})
    six = {
        "workload": f"the same {B} plans on the reference's own map: all SIX obstacles of compute_obstacles (compute_sets.py:259-330)",
        "plans_per_s": B / (t_ws + t_col), "colloc_s": t_col, "colloc_converged": ok,
        "colloc_iters_mean": float(np.mean([r["iters"] for r in plans.values()])), "colloc_iters_max": int(max(r["iters"] for r in plans.values())),
        "colloc_iters_top3": sorted((int(r["iters"]) for r in plans.values()), reverse=True)[:3],
        "ms_per_iteration_of_the_slowest_plan": 1e3 * t_col / max(1, max(r["iters"] for r in plans.values())),
        "roofline": roofline_of(alg, t_col, d6, what1)}
    # configs[1] as BASELINE.json words it: FOUR polytope obstacles (0, 1, 3, 4 of the reference's six) -- the same B plans'
    # collocation refinement on that map (state_ws does not see the obstacles: the tube keeps the vehicle off them).  This is the object's
    # own line since round 6; the six-obstacle run is nested under it.
    sp4 = scenarios.parking_lot_spec(n_nbr=0, N=2, n_obs=4)
    gs = {k: guess_of(ws[k]["traj"], len(tubes[who[k]]) + 1) for k in good}

    def colloc4(sel, max_iter=400):
        t0 = time.perf_counter()
        r = engine.colloc(sp4, [init[k] for k in sel], [tubes[who[k]] for k in sel], [gs[k][0] for k in sel], [gs[k][1] for k in sel],
                          [fh[who[k]] for k in sel], max_iter=max_iter, device=device)
        launches["colloc_kernel"] += 1
        return r, time.perf_counter() - t0
    colloc4(good[:8])  # warm-up, as the six-obstacle launches had
    r4, t_col4 = colloc4(good)
    d4 = launches["colloc_kernel"] - 1
    it4 = np.array([r["iters"] for r in r4])
    # the launch lasts as long as its slowest plan: the same launch stopped at the iteration count that 95 % of the plans need says when
    # 95 % of the batch was done (a measured time, not an extrapolation; the stragglers end with status 1 there)
    it95 = int(np.sort(it4)[int(np.ceil(0.95 * len(it4))) - 1])
    r95, t_col95 = colloc4(good, max_iter=it95)
    done95 = sum(r["status"] == 0 for r in r95)
    info4o = {a: engine.colloc_elimination_info([len(tubes[a]) + 1], n_obs=4) for a in agents}
    alg4o = float(sum(info4o[who[k]]["alg_bytes"] * r["iters"] for k, r in zip(good, r4)))
    out["configs[1]"] = {
        "workload": f"BASELINE.json configs[1]: {B} independent single-vehicle OBCA plans (state_ws -> collocation plan, N_per_set 5, K 5) with "
                    "BASELINE's 4 polytope obstacles (0, 1, 3, 4 of the reference's six)",
        "plans_per_s": B / (t_ws + t_col4), "state_ws_s": t_ws, "colloc_s": t_col4, "state_ws_converged": len(good),
        "colloc_converged": sum(r["status"] == 0 for r in r4),
        "colloc_iters_mean": float(it4.mean()), "colloc_iters_max": int(it4.max()), "colloc_iters_top3": sorted((int(x) for x in it4), reverse=True)[:3],
        "ms_per_iteration_of_the_slowest_plan": 1e3 * t_col4 / max(1, int(it4.max())),
        "p95": {"what": f"the same launch with max_iter = {it95}, the iteration count 95 % of the plans need: the rate at which 95 % of the batch is done",
                "max_iter": it95, "colloc_s": t_col95, "converged": int(done95), "plans_per_s": done95 / (t_ws + t_col95)},
        "state_ws_iters_mean": float(np.mean([w_["iters"] for w_ in ws])), "state_ws_iters_max": int(max(w_["iters"] for w_ in ws)),
        "note": "one plan of the batch has several minimisers: replayed on the CPU build with its guess perturbed by 1e-13 (relative) it takes 61-212 "
                "iterations and ends at one of three plans (docs/notebook.md); its count here is a draw from that range, the launch lasts as long as it",
        "roofline": roofline_of(alg4o, t_col4, d4, what1.replace("%s per vehicle" % {a: info1[a]["alg_bytes"] for a in agents}, "%s per vehicle" % {a: info4o[a]["alg_bytes"] for a in agents})),
        "six_obstacles": six}
    # ---- configs[3] ------------------------------------------------------------------------------------------------------------
    idx = list(range(4 * B))
    ws4, good4, plans4, _, _ = single_plans(idx)
    scen = []
    for b in range(B):
        ks = [4 * b + i for i in range(4)]
        if not all(k in plans4 and plans4[k]["status"] == 0 for k in ks):
            continue
        scen.append(dict(init_poses=[init[k] for k in ks], tubes=[tubes[a] for a in agents], guesses=[plans4[k]["traj"].reshape(-1, 7) for k in ks],
                         dt0=float(np.mean([plans4[k]["dt"] for k in ks])), final_headings=[fh[a] for a in agents]))
    # warm-up (as configs[1]'s launches had): one iteration of the same batch, so that the workspace arena (7 GB for 256 plans) exists when
    # the timed call starts -- on a freshly acquired box its first allocation took 0.6 s in three of eight runs (docs/notebook.md)
    engine.joint_colloc_batch(sp0, scen, max_iter=1, device=device)
    launches["colloc_kernel"] += 1
    t0 = time.perf_counter()
    rj = engine.joint_colloc_batch(sp0, scen, max_iter=300, device=device)
    t_joint = time.perf_counter() - t0
    d3 = launches["colloc_kernel"]
    launches["colloc_kernel"] += 1
    # ... and, as for configs[1], the same launch stopped at the iteration count 95 % of the plans need: when 95 % of the batch was done
    itj = np.array([r["iters"] for r in rj])
    itj95 = int(np.sort(itj)[int(np.ceil(0.95 * len(itj))) - 1])
    t0 = time.perf_counter()
    rj95 = engine.joint_colloc_batch(sp0, scen, max_iter=itj95, device=device)
    t_joint95 = time.perf_counter() - t0
    launches["colloc_kernel"] += 1
    info4 = engine.colloc_elimination_info([len(tubes[a]) + 1 for a in agents])
    nk4, kb4, bb4 = info4["nk"], info4["kb"], info4["band_bytes"]
    alg4 = float(sum(info4["alg_bytes"] * r["iters"] for r in rj))
    out["configs[3]"] = {
        "workload": f"BASELINE.json configs[3]: {len(scen)} centralised four-vehicle joint plans (six pairs, one shared dt) in one launch, one workgroup each",
        "plans_per_s": len(scen) / t_joint, "joint_s": t_joint, "converged": sum(r["status"] == 0 for r in rj),
        "iters_mean": float(np.mean([r["iters"] for r in rj])), "iters_max": int(max(r["iters"] for r in rj)),
        "iters_top3": sorted((int(r["iters"]) for r in rj), reverse=True)[:3],
        # the launch lasts as long as its slowest plan, and which plan wanders between minimisers (and for how long) is decided in the last
        # digits of its guess: the time per iteration of that plan is the figure that compares builds
        "ms_per_iteration_of_the_slowest_plan": 1e3 * t_joint / max(1, max(r["iters"] for r in rj)),
        "p95": {"what": f"the same launch with max_iter = {itj95}, the iteration count 95 % of the plans need: the rate at which 95 % of the batch is done",
                "max_iter": itj95, "joint_s": t_joint95, "converged": int(sum(r["status"] == 0 for r in rj95)),
                "plans_per_s": sum(r["status"] == 0 for r in rj95) / t_joint95},
        "unknowns": nk4, "half_bandwidth": kb4, "band_bytes": bb4, "workspace_bytes_per_plan": info4["workspace_bytes"],
        "elimination": "structured (cfz_jstruct.inl): vehicle-major ordering, per-vehicle band of half-bandwidth 51, no band across the vehicles",
        "roofline": roofline_of(alg4, t_joint, d3, f"bytes the structured elimination of one joint Newton system moves between its phases "
                                f"(cfz_colloc_elimination_info: {info4['alg_bytes']}) x interior-point iterations of every plan")}
    # BASELINE.md section 4 config 2 as drawn there: lane poses, default_rng(1234), MPC form, four obstacles, no neighbours
    try:
        out["configs[1]"]["lane_sampler"] = lane_sampler_line(device)
    except Exception as e:  # noqa: BLE001 - reporting only
        out["configs[1]"]["lane_sampler"] = {"error": f"{type(e).__name__}: {e}"}
    engine.trim_default_workspaces()  # a 256-plan joint launch leaves 25 GB in the calling thread's workspace
    if cpu:
        try:
            out["configs[1]"]["cpu_baseline"], out["configs[3]"]["cpu_baseline"] = planning_cpu_baseline(agents, sets, paths, fh)
        except Exception as e:  # noqa: BLE001 - reporting only
            out["configs[1]"]["cpu_baseline"] = out["configs[3]"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def lane_sampler_line(device=0, B=256):
    """B = 256 independent single-vehicle OBCA problems in MPC form (N = 30, the four
    obstacles 0, 1, 3, 4 of the reference's six, no neighbours) at lane poses drawn with default_rng(1234): one cold `cfz_mpc_solve` of the
    whole batch (host buffers in and out; the kernel time beside it), and the same batch on one host core through the C port."""
    from conflict_rez_amd import engine, scenarios

    spec = scenarios.parking_lot_spec(n_obs=4, n_nbr=0)
    x0, ref, zu = scenarios.lane_sampler(spec, B=B, seed=1234)
    eng = engine.Engine(spec, max_batch=B, device=device)
    eng.solve(x0, ref, None, zu, want_duals=False)  # warm-up
    t0 = time.perf_counter()
    out = eng.solve(x0, ref, None, zu, want_duals=False)
    wall = time.perf_counter() - t0
    eng.close()
    it = out["iters"]
    line = {"workload": f"BASELINE.md section 4 config 2: {B} single-vehicle MPC-form problems, N = 30, 4 obstacles, lane poses x0~U[5,30], y0~U[15,20], "
                        "psi0 in {0,pi}+N(0,0.05), v0~U[-1,1], default_rng(1234); one cold solve of the batch",
            "solves_per_s": B / wall, "solves_per_s_kernel": B / (out["solve_ms"] * 1e-3), "wall_ms": wall * 1e3, "kernel_ms": out["solve_ms"],
            "converged": int((out["status"] == 0).sum()), "iters_mean": float(it.mean()), "iters_max": int(it.max()),
            "status_counts": {int(k): int(v) for k, v in zip(*np.unique(out["status"], return_counts=True))}}
    try:
        from oracle import port
        from oracle.mpc_nlp import MpcSpec

        ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=0)
        t0 = time.perf_counter()
        same = 0
        for b in range(B):
            r = port.solve(ospec, x0[b], ref[b], np.zeros((0, 3, spec.N)), zu[b].T)
            same += int(r["status"] == out["status"][b] and r["iters"] == out["iters"][b])
        tc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": B / tc, "unit": "solves/s", "cores": 1, "kind": "port", "sample": f"the same {B} problems, {tc:.2f} s",
                                "same_status_and_iterations": same}
    except Exception as e:  # noqa: BLE001
        line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    return line


def planning_cpu_baseline(agents, sets, paths, fh):
    """One core, the CPU build of the planning kernels' own source (tests/emu, "kind": "port"): the four vehicles' single plans
    (state_ws + collocation) and ONE four-vehicle joint plan from them -- a bounded sample of configs[1] / configs[3]."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import colloc_emu_binding as ce
    import plan_emu_binding as pe
    from conflict_rez_amd import scenarios
    from oracle import ipm
    from oracle.colloc_nlp import CollocNlp, JointCollocNlp
    from oracle.plan_nlp import StateWsNlp, speed_guess
    from scipy.interpolate import interp1d

    sp = scenarios.parking_lot_spec()
    otubes = {a: [dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in sets[a]] for a in agents}
    opt = ipm.IpmOptions(max_iter=400, reg_dual=1e-7, tol=1e-2, constr_viol_tol=1e-2, mu_init=0.1)
    opt1 = ipm.IpmOptions(max_iter=400, reg_dual=1e-7, tol=1e-2, constr_viol_tol=1e-2, mu_init=0.1)
    opt1.no_prox = 4  # single plans: the structured elimination (cfz_jstruct.inl's single-vehicle scheme), as on the GPU -- also the faster one on the CPU
    t0 = time.perf_counter()
    singles = []
    for a in agents:
        p = paths[a]
        ws = StateWsNlp(p[0], otubes[a], final_heading=fh[a], shrink_tube=0.5)
        r = pe.solve(ws, ws.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, ws.dt)), ipm.IpmOptions(max_iter=500, hessian="exact", reg_dual=1e-9, stall_iters=0, mu_init=0.1))
        z = ws.unpack(r["X"])
        nlp = CollocNlp(p[0], otubes[a], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh[a])
        N = nlp.N[0]
        t_i = np.concatenate([i + nlp.tau for i in range(N)]) / N * z["t"][-1]
        X0 = nlp.pack({k: interp1d(z["t"], z[k])(t_i) for k in ("x", "y", "psi", "v", "delta", "a", "w")}, z["t"][-1] / N)
        singles.append(nlp.unpack(ce.solve(nlp, X0, opt1)["X"]))
    t1 = time.perf_counter()
    jn = JointCollocNlp([dict(init_pose=paths[a][0], tube=otubes[a], final_heading=fh[a]) for a in agents], sp.A_obs, sp.b_obs, N_per_set=5)
    rj = ce.solve(jn, jn.pack(singles, float(np.mean([s["dt"] for s in singles]))), opt1)  # the structured elimination (cfz_jstruct.inl), as on the GPU
    t2 = time.perf_counter()
    return ({"value": 4 / (t1 - t0), "unit": "plans/s", "cores": 1, "kind": "port", "sample": f"the four vehicles' plans (state_ws + collocation), {t1 - t0:.1f} s"},
            {"value": 1 / (t2 - t1), "unit": "plans/s", "cores": 1, "kind": "port",
             "sample": f"one four-vehicle joint plan, {rj['iters']} iterations, status {rj['status']}, {t2 - t1:.1f} s"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scenarios", type=int, default=1024, help="scenarios per GPU (x4 vehicles)")
    ap.add_argument("--max-iter", type=int, default=600, help="IPM iteration limit (reference: 600)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=2024, help="scenario sampler seed of rank 0 (rank r uses seed + r)")
    ap.add_argument("--n-obs", type=int, default=6, help="static obstacles (reference map: 6); fewer = experiments only")
    ap.add_argument("--extras-only", action="store_true", help="only the planning extras (configs[1], configs[3]): the command tools/gpu_profile_extras.sh profiles")
    ap.add_argument("--workload", choices=["mpc4", "single"], default="mpc4",
                    help="mpc4: BASELINE.json configs[2], 4-vehicle distributed MPC (the metric); single: configs[1] in MPC form, "
                         "independent single-vehicle problems, 4 obstacles, no neighbours (use --scenarios 256)")
    ap.add_argument("--mode", choices=["persistent", "step"], default="persistent",
                    help="persistent: K iterations in one launch, scenarios advance independently (cfz_loop_run); "
                         "step: one launch per iteration with a device-wide barrier in between (cfz_loop_step)")
    ap.add_argument("--parallelism", choices=["scenario", "vehicle"], default="scenario",
                    help="scenario: every GPU owns whole scenarios, no data-path collective (default); vehicle: every GPU owns "
                         "vehicles of all scenarios and all-gathers the predictions over RCCL every iteration (the reference's ROS "
                         "deployment; needs torch.distributed, --gpus dividing 4, one launch per iteration)")
    ap.add_argument("--count-iters", action="store_true", help="step mode: also sum the IPM iterations (adds a read-back)")
    ap.add_argument("--reference", choices=["planned", "state_ws", "replan"], default="planned",
                    help="planned (default): conflict_rez_amd/data/refs_4v_planned.npz, the build's own single-vehicle plans (state_ws -> "
                         "collocation plan, VehicleFollower.plan_single_path) of the synthetic strategy, SURVEY.md 8d config 3; state_ws: "
                         "data/refs_4v.npz, the slower state_ws warm-start plans (the table of rounds 1-2); replan: build the planned table "
                         "at start-up with the GPU planning chain instead of loading it")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[1] / configs[3] planning measurements after the timed region")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="experiments only: a field of cfz_options for the MPC engine (e.g. carry_duals=0, restoration=0); noted in config.options")
    ap.add_argument("--no-seeds", action="store_true", help="skip the three-seed and all-moving repetitions of the headline after the timed region")
    ap.add_argument("--raw-starts", action="store_true",
                    help="take the sampler's starts as they come (rounds 1-2); default: a scenario whose noisy start state is already "
                         "inside a clearance (an infeasible first NLP, status 4) is drawn again")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.extras_only:  # (one process, one GPU: what tools/gpu_profile_extras.sh profiles; not a bench line)
        print(json.dumps({"extras_only": True, "extra": planning_extras(device=local_rank, cpu=not args.no_cpu_baseline)}), flush=True)
        return

    dist = None
    # CFZ_BENCH_FORCE_DIST=1: go through torch.distributed even with one rank (exercises the N > 1 code path on one GPU)
    if world > 1 or os.environ.get("CFZ_BENCH_FORCE_DIST") == "1" or (args.parallelism == "vehicle" and "RANK" in os.environ):
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")

    from conflict_rez_amd import engine, scenarios

    single = args.workload == "single"
    spec = scenarios.parking_lot_spec(n_obs=4 if single else args.n_obs, n_nbr=0 if single else 3)
    V = spec.n_nbr + 1
    if args.reference == "replan":
        table, lengths, plan_info = scenarios.planned_reference_table(device=local_rank)
        ref_desc = ("built at start-up by the GPU planning chain (cfz_state_ws -> cfz_colloc, VehicleFollower.plan_single_path): "
                    + ", ".join(f"{a} {i['t_end']:.1f} s" for a, i in plan_info.items()))
    elif args.reference == "planned":
        table, lengths = scenarios.load_reference_table(kind="planned")
        ref_desc = ("conflict_rez_amd/data/refs_4v_planned.npz: the package's own single-vehicle plans of the synthetic strategy (cfz_state_ws -> "
                    "cfz_colloc as in VehicleFollower.plan_single_path, free dt; planned by the round-3 build, kept since so that the workload stays "
                    "the same: tests/test_plan.py compares today's plans with it), sampled every 0.1 s: "
                    + ", ".join(f"vehicle_{i} {0.1 * (n - 1):.1f} s" for i, n in enumerate(lengths)) + " (SURVEY.md 8d config 3)")
    else:
        table, lengths = scenarios.load_reference_table(kind="state_ws")
        ref_desc = ("conflict_rez_amd/data/refs_4v.npz: the four vehicles' Vehicle.state_ws plans of the synthetic strategy "
                    "(tube-constrained warm-start trajectories, 18-30 s)")
    if single:
        table = table[rank % table.shape[0]][None].copy()  # every scenario follows one vehicle's plan, alone on the map
    S = args.scenarios
    fspec = None if (args.raw_starts or single) else spec
    k0, noise = scenarios.sample_scenarios(S, table, seed=args.seed + rank, spec=fspec)
    lengths_v = lengths  # samples of every vehicle's plan (it is parked at its goal after that)
    opts = {kv.split("=", 1)[0]: (float(kv.split("=", 1)[1]) if "." in kv.split("=", 1)[1] or "e" in kv.split("=", 1)[1] else int(kv.split("=", 1)[1]))
            for kv in args.option}
    vehicle_sharded = args.parallelism == "vehicle"
    if vehicle_sharded:
        if dist is None or single:
            raise SystemExit("--parallelism vehicle needs torch.distributed (torchrun, or CFZ_BENCH_FORCE_DIST=1) and the mpc4 workload")
        from conflict_rez_amd.distributed import VehicleShardedExchange, VehicleShardedLoop

        # scenarios x world in total: every rank steps its vehicles (V / world of them, or one vehicle of a scenario shard when
        # there are more ranks than vehicles) -> the same number of solves per GPU as in scenario sharding
        S_total = args.scenarios * world
        k0, noise = scenarios.sample_scenarios(S_total, table, seed=args.seed, spec=fspec)  # (every rank holds all scenarios' starts)
        ex = VehicleShardedExchange(V)
        S = len(range(S_total)[ex.scenarios(S_total)])
        eng = engine.Engine(spec, max_batch=S * len(ex.owned), device=local_rank, max_iter=args.max_iter, **opts)
        vloop = VehicleShardedLoop(eng, ex, table, k0, noise, device=f"cuda:{local_rank}")
        args.mode = "step"
    else:
        eng = engine.Engine(spec, max_batch=S * V, device=local_rank, max_iter=args.max_iter, **opts)
        eng.loop_init(table, k0, noise)

    # what the sampler handed out, counted on the starts that are actually used
    infeasible_starts = None if single else int(((scenarios.start_clearances(spec, table, k0, noise) < spec.dmin - 0.01).any(1)
                                                 | (scenarios.start_box_excess(spec, table, k0, noise) > 1e-2).any(1)).sum())

    redrawn = None
    if fspec is not None and not vehicle_sharded:
        k0_raw, nz_raw = scenarios.sample_scenarios(len(k0), table, seed=args.seed + rank)
        redrawn = int(((k0_raw != k0) | (np.abs(nz_raw - noise).max((1, 2)) > 0.0)).sum())  # scenarios of the raw sampler that were drawn again (rounds 1-2 took them as they came)

    def parked_fraction(k0_, t_first, t_count):
        """Share of the (scenario, vehicle, MPC iteration) triples of a closed-loop window whose vehicle has reached the end of its
        plan (it then holds its goal pose: one or two interior-point iterations per solve)."""
        if lengths_v is None or single:
            return None
        tt = np.arange(t_first, t_first + t_count)
        return float((np.asarray(k0_)[:, None, None] + tt[None, None, :] >= (np.asarray(lengths_v)[None, :, None] - 1)).mean())

    def barrier():
        if dist is not None:
            import torch

            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()

    persistent = args.mode == "persistent"
    cold = None
    if vehicle_sharded:
        for _ in range(args.warmup):
            vloop.step(sync=True)
    elif persistent:
        if args.warmup > 0:
            # the first iteration on its own: cold multipliers, nothing carried (report step 1 separately)
            it1 = eng.loop_run(1)
            cold = {"value": S * V / (eng.last_solve_ms() / 1e3), "unit": "solves/s", "mean_ipm_iters": it1 / (S * V),
                    "kernel_ms": eng.last_solve_ms(), "converged": eng.loop_last_converged() / (S * V),
                    "what": "first MPC iteration after loop_init on this GPU (cold multipliers), one launch, kernel time"}
            if args.warmup > 1:
                eng.loop_run(args.warmup - 1)  # blocks until all scenarios have done `warmup` iterations
    else:
        for _ in range(args.warmup):
            eng.loop_step()  # blocks until the step is complete on the device
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    n_ok = 0
    n_conv = None  # converged solves of the timed region (persistent mode counts them on the device)
    status_counts = None  # ... and how all of them ended
    if vehicle_sharded:
        ipm_iterations = 0
        for _ in range(args.steps):
            vloop.step(sync=True)
            kernel_ms += vloop.solve_ms
            ipm_iterations += int(vloop.iters.sum())
        launches = args.steps
    elif persistent:
        ipm_iterations = eng.loop_run(args.steps)  # K iterations of every scenario, one launch
        kernel_ms = eng.last_solve_ms()
        n_conv = eng.loop_last_converged()
        status_counts = eng.loop_last_status_counts()
        launches = 1
    else:
        ipm_iterations = 0
        for _ in range(args.steps):
            eng.loop_step()
            kernel_ms += eng.last_solve_ms()
            ipm_iterations += int(eng.loop_get()["iters"].sum()) if args.count_iters else 0
        launches = args.steps
    barrier()
    elapsed = time.perf_counter() - t0
    if vehicle_sharded:
        n_ok = int((vloop.status == 0).sum())
        iters_mean = float(vloop.iters.double().mean())
        V_local = len(ex.owned)
    else:
        got = eng.loop_get()
        n_ok = int((got["status"] == 0).sum())
        iters_mean = float(got["iters"].mean())
        V_local = V

    if dist is not None:
        import torch

        t = torch.tensor([elapsed, kernel_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
        c = torch.tensor([n_ok, -1 if n_conv is None else n_conv, 1] + ([0] * 6 if status_counts is None else [int(x) for x in status_counts]),
                         device="cuda", dtype=torch.int64)
        dist.all_reduce(c)
        n_ok = int(c[0])
        n_conv = None if n_conv is None else int(c[1])
        rccl_ranks = int(c[2])  # an all-reduce of ones over RCCL: the ranks that took part in this measurement
        status_counts = None if status_counts is None else c[3:9].cpu().numpy()
    else:
        rccl_ranks = None

    out_line = None
    if rank == 0:
        B = S * V_local  # solves per iteration on one GPU
        solves = B * world * args.steps
        kern_s = kernel_ms / 1e3 / launches  # average solver-kernel duration per launch
        solves_per_launch = B * args.steps // launches
        achieved = solves_per_launch * ALG_BYTES_PER_SOLVE / kern_s / 1e9
        traffic, traffic_src = (None, None)
        valu_frac, valu_src = (None, None)
        if persistent and not vehicle_sharded and args.steps == 20 and S == 1024 and not single:  # the committed PMC passes are of the default command
            traffic, traffic_src = profiled_traffic("loop_kernel")
            valu_frac, valu_src = profiled_sq("loop_kernel")
        # flops per interior-point iteration: MEASURED (FP64 instruction counters of the committed pass of this same command on this same
        # library: executed lane-slots, an upper bound on useful flops) when there is such a pass, SURVEY's estimate of the algorithm otherwise
        fp64_pass, fp64_src = (None, None)
        if persistent and not vehicle_sharded and args.steps == 20 and S == 1024 and not single:
            fp64_pass, fp64_src = profiled_fp64("loop_kernel")
        flop_per_it = fp64_pass["flop_per_ipm_iteration"] if fp64_pass and fp64_pass.get("flop_per_ipm_iteration") else FLOP_PER_IPM_ITERATION
        fp64_tflops = (ipm_iterations * flop_per_it / (kernel_ms / 1e3) / 1e12) if ipm_iterations else None
        line = {
            "metric": "OBCA MPC-step solves/sec (4 vehicles, N=30)",
            # converged solves only (status 0); persistent mode counts them on the device over the whole timed region, the other
            # modes know the last iteration's share only and scale by it
            "value": (n_conv / elapsed) if n_conv is not None else solves / elapsed * (n_ok / (B * world)),
            "value_all": solves / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1] in MPC form: independent single-vehicle problems, N=30, 4 obstacles, "
                                    "no neighbours, closed loop on device") if single else
                                   ("BASELINE.json configs[2]: 4-vehicle distributed MPC (VehicleFollower.step), "
                                    "N=30, 6 obstacles, closed loop on device"), "scenarios_per_gpu": S,
                       "solves_per_step_per_gpu": B,
                       "parallelism": (f"vehicle-sharded x{world} ({len(ex.owned)} vehicle(s) x 1/{ex.n_shards} of the scenarios per rank), "
                                       f"RCCL all-gather of predictions inside groups of {ex.group_size} ranks per iteration" if vehicle_sharded
                                       else f"scenario-sharded x{world}"),
                       "rccl_ranks": rccl_ranks, "options": opts or None,
                       "status_counts": None if status_counts is None else
                                        dict(zip(("0 converged", "1 iteration limit", "2 line search", "3 non-finite",
                                                  "4 measured state infeasible (no iteration)", "5 stalled / locally infeasible"),
                                                 (int(x) for x in status_counts))),
                       "parked_fraction": parked_fraction(k0[ex.scenarios(len(k0))] if vehicle_sharded else k0, args.warmup, args.steps),
                       "max_iter": args.max_iter, "mode": args.mode, "converged_last_step": n_ok / (B * world),
                       "converged_timed_region": (n_conv / solves) if n_conv is not None else None,
                       "reference_plan": ref_desc,
                       "infeasible_starts": infeasible_starts, "redrawn_scenarios": redrawn,
                       "starts": "raw sampler draws" if fspec is None else
                                 "scenarios whose noisy start state violates a clearance or lies outside the NLP's state boxes (an infeasible first NLP) are drawn again",
                       "ipm_iterations_rank0": ipm_iterations,
                       "mean_ipm_iters_timed_region": (ipm_iterations / (B * args.steps)) if ipm_iterations else None,
                       "mean_ipm_iters_last_step": iters_mean, "scenario_steps_per_s": solves / elapsed / V,
                       "lds_bytes_per_instance": eng.kernel_info()[0], "instances_per_cu": eng.kernel_info()[1]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "loop_kernel" if persistent and not vehicle_sharded else "solve_kernel",
                         "kernel_ms_per_launch": kern_s * 1e3, "solves_per_launch": solves_per_launch,
                         "alg_bytes_per_solve": ALG_BYTES_PER_SOLVE,
                         "fp64_tflops": fp64_tflops, "fp64_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                         "fp64_frac": (fp64_tflops / FP64_VECTOR_PEAK_TFLOPS) if fp64_tflops else None,
                         "flop_per_ipm_iteration": flop_per_it,
                         "flop_source": (f"measured: FP64 instruction counters x 64 lanes + MFMA operations of {fp64_src} (executed lane-slots)"
                                         if flop_per_it is not FLOP_PER_IPM_ITERATION else "SURVEY.md 8(d) estimate of the algorithm's flops (no current counter pass)"),
                         "flop_per_ipm_iteration_estimate": FLOP_PER_IPM_ITERATION,
                         "mfma_busy_frac": fp64_pass["mfma_busy_frac"] if fp64_pass else None,
                         "mfma_flops_share": (fp64_pass["mfma_flops"] / fp64_pass["flops"]) if fp64_pass and fp64_pass["flops"] else None,
                         "valu_useful_frac": fp64_pass["valu_useful_frac"] if fp64_pass else None,
                         "valu_active_frac": valu_frac["frac"] if valu_frac else None,
                         "valu_active_raw_quotient": valu_frac["raw_quotient"] if valu_frac else None, "valu_active_source": valu_src,
                         "note": "latency/FP64-issue bound: the iterate lives in LDS, so algorithmic HBM bytes "
                                 "are ~1e-5 of peak by construction (SURVEY.md 8d); see DESIGN.md"},
        }
        if cold is not None:
            line["cold_step"] = cold
        if world == 1 and persistent and not single and not vehicle_sharded and not args.no_seeds:
            # ---- the same measurement on other samples, OUTSIDE the timed region (same process, same engine) ------------------------
            def closed_loop_sample(k0_, noise_):
                eng.loop_init(table, k0_, noise_)
                if args.warmup > 0:
                    eng.loop_run(args.warmup)
                t1 = time.perf_counter()
                its = eng.loop_run(args.steps)
                dt_ = time.perf_counter() - t1
                sc, n = eng.loop_last_status_counts(), S * V * args.steps
                return {"value": float(sc[0] / dt_), "value_all": n / dt_, "ms_per_step": dt_ / args.steps * 1e3, "converged": float(sc[0] / n),
                        "mean_ipm_iters": its / n, "status_counts": [int(x) for x in sc],
                        "parked_fraction": parked_fraction(k0_, args.warmup, args.steps)}

            extra = line.setdefault("extra", {})
            try:
                seeds = [2024, 2025, 2026]
                runs = [closed_loop_sample(*scenarios.sample_scenarios(S, table, seed=sd, spec=fspec)) for sd in seeds]
                vals = sorted(r["value"] for r in runs)
                extra["seeds"] = {"what": "the headline's measurement (warm-up + timed persistent launch, converged solves per second) on three "
                                          "sampler seeds: a launch lasts as long as its slowest scenario's chain of interior-point iterations, "
                                          "the maximum of 1024 draws",
                                  "seeds": seeds, "median": vals[1], "min": vals[0], "max": vals[2], "runs": runs}
                if lengths_v is not None:
                    margin = table.shape[1] - (int(min(lengths_v)) - 30 - (args.warmup + args.steps))
                    k0m, nzm = scenarios.sample_scenarios(S, table, seed=args.seed, horizon_margin=margin, spec=fspec)
                    extra["all_moving"] = dict(closed_loop_sample(k0m, nzm),
                                               what=f"start samples k0 < {table.shape[1] - margin}: no vehicle reaches the end of its plan "
                                                    f"(the shortest has {int(min(lengths_v))} samples) within the {args.warmup + args.steps} MPC iterations")
            except Exception as e:  # noqa: BLE001 - the headline must go out whatever happens here
                extra["seeds_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_extras and not single and not vehicle_sharded:
            try:
                line.setdefault("extra", {}).update(planning_extras(device=local_rank, cpu=not args.no_cpu_baseline))
            except Exception as e:  # noqa: BLE001 - the headline must go out whatever happens here
                line.setdefault("extra", {})["error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_cpu_baseline and not single:
            ref_kind = args.reference
            if args.reference == "replan":  # the workers get the table this run built, not the packaged one
                import tempfile

                ref_kind = os.path.join(tempfile.mkdtemp(), "table.npz")
                np.savez(ref_kind, table=table)
            line["cpu_baseline"] = cpu_baseline(max(args.warmup, 1), args.steps, ref_kind=ref_kind, feasible=fspec is not None)
            cb = line["cpu_baseline"]
            if ipm_iterations:  # like for like: IPM iterations per second, GPU : all usable host cores : one core
                line["gpu_vs_cpu"] = {"ipm_iterations_per_s_gpu": ipm_iterations * world / elapsed,
                                      "ipm_iterations_per_s_cpu": cb["ipm_iterations_per_s"],
                                      "ratio_same_cores": ipm_iterations * world / elapsed / cb["ipm_iterations_per_s"],
                                      "ratio_per_core": ipm_iterations * world / elapsed / (cb["ipm_iterations_per_s"] / cb["cores"])}
        out_line = json.dumps(line)
    if dist is not None:
        dist.destroy_process_group()
    # the ONE line goes out last: RCCL writes its version banner through C stdio, which a pipe only sees at exit unless it is
    # flushed here
    import ctypes

    ctypes.CDLL(None).fflush(None)
    if out_line is not None:
        print(out_line, flush=True)


if __name__ == "__main__":
    main()
