#!/usr/bin/env python3
"""Headline benchmark: OBCA MPC-step solves/sec (4 vehicles, N=30) -- BASELINE.json config 3.

A "step" is one closed-loop iteration of the distributed MPC (`MultiDistributedFollower.solve`,
reference vehicle_follower.py:630-663) for S = 1024 scenarios x 4 vehicles = 4096 NLP solves
per GPU, executed entirely on the device (`cfz_loop_step`: parameters + shifted warm start,
solve, read-back / fallback, plant integration).  Inputs are resident in HBM before the timed
region.  N GPUs = N independent shards of scenarios (weak scaling, no data-path collective).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit, `roofline`
(algorithmic HBM bytes of SURVEY.md 8d / measured solver-kernel time) and `cpu_baseline`
(oracle/cfz_port.c, the plain-C port, timed on this host; N=1 only).
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


def _cpu_worker(args):
    """One process of the all-core CPU baseline: solves its slice with the oracle's C port (its buffers are static,
    hence processes, not threads)."""
    lo, hi, seconds = args
    from conflict_rez_amd import scenarios
    from oracle import port
    from oracle.mpc_nlp import MpcSpec

    spec = scenarios.parking_lot_spec()
    table, _ = scenarios.load_reference_table()
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=spec.n_nbr)
    k0, noise = scenarios.sample_scenarios(4096 // (spec.n_nbr + 1), table, seed=2024)
    x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
    port.solve(ospec, x0[0], ref[0], nbr[0], zu[0].T)  # load + warm
    n, its, t0 = 0, 0, time.perf_counter()
    for b in range(lo, hi):
        if time.perf_counter() - t0 > seconds:
            break
        its += port.solve(ospec, x0[b], ref[b], nbr[b], zu[b].T)["iters"]
        n += 1
    return n, its, time.perf_counter() - t0


def cpu_baseline(spec, table, seconds=12.0, max_solves=4096):
    """Same workload (cold first step of the same scenario sampler) through the oracle's C port: one core, then one
    process per host core (at most 64) over the same 4096 instances.  Bounded: every leg stops after `seconds`."""
    import concurrent.futures as cf
    import multiprocessing as mp

    n1, its1, dt1 = _cpu_worker((0, max_solves, seconds / 2))
    single = {"value": n1 / dt1, "unit": "solves/s", "cores": 1, "kind": "port",
              "sample": f"{n1} cold first-step solves of the same scenario sampler in {dt1:.1f} s, single thread "
                        f"({os.cpu_count()} host cores present), mean {its1 / max(n1, 1):.1f} IPM iterations"}
    # The box reports 256 logical cores but the job may own fewer (cgroup quota): try 8, 16, 32, 64 processes over the
    # same 4096 instances, a few seconds each, and report the best.
    best = None
    try:  # spawn: this process already holds a GPU context; a worker that dies breaks the pool instead of hanging it
        for cores in [c for c in (8, 16, 32, 64) if c <= (os.cpu_count() or 1)] or [1]:
            per = (max_solves + cores - 1) // cores
            jobs = [(i * per, min((i + 1) * per, max_solves), seconds / 4) for i in range(cores) if i * per < max_solves]
            with cf.ProcessPoolExecutor(len(jobs), mp_context=mp.get_context("spawn")) as pool:
                res = list(pool.map(_cpu_worker, jobs, timeout=seconds + 120))
            rate = sum(r[0] for r in res) / max(r[2] for r in res)  # slowest worker; start-up (imports) not counted
            if best is None or rate > best[0]:
                best = (rate, len(jobs), per, res)
    except Exception as e:  # noqa: BLE001 - the baseline is reporting only; fall back to the single-core figure
        single["note"] = f"all-core leg failed ({type(e).__name__}); single core only"
        return single
    _, ncores, per, res = best
    jobs = [None] * ncores
    wall = max(r[2] for r in res)  # slowest worker; process start-up (imports) is not counted
    n = sum(r[0] for r in res)
    return {"value": n / wall, "unit": "solves/s", "cores": len(jobs), "kind": "port",
            "single_core": n1 / dt1,
            "sample": f"{n} cold first-step solves of the same scenario sampler, {len(jobs)} processes x "
                      f"{per} instances, slowest worker {wall:.1f} s (mean {sum(r[1] for r in res) / max(n, 1):.1f} IPM "
                      f"iterations); single core: {n1} solves in {dt1:.1f} s; {os.cpu_count()} host cores present",
            "note": "CasADi/IPOPT (the reference's CPU path) is not installable here; its implied range is "
                    "10-90 ms per solve = 11-100 solves/s per core (BASELINE.md, unpublished)"}


def profiled_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summaries of this same command (profiles/, separate
    --pmc passes of rocprofv3): raw FETCH_SIZE + WRITE_SIZE in KiB of the timed (last) dispatch.  None if absent."""
    import glob

    here = os.path.dirname(os.path.abspath(__file__))
    tags = sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_FETCH_SIZE.csv")))
    if not tags:
        return None, None
    tag = tags[-1][: -len("_pmc_FETCH_SIZE.csv")]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scenarios", type=int, default=1024, help="scenarios per GPU (x4 vehicles)")
    ap.add_argument("--max-iter", type=int, default=600, help="IPM iteration limit (reference: 600)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-obs", type=int, default=6, help="static obstacles (reference map: 6); fewer = experiments only")
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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    dist = None
    # CFZ_BENCH_FORCE_DIST=1: go through torch.distributed even with one rank (exercises the N > 1 code path on one GPU)
    if world > 1 or os.environ.get("CFZ_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")

    from conflict_rez_amd import engine, scenarios

    single = args.workload == "single"
    spec = scenarios.parking_lot_spec(n_obs=4 if single else args.n_obs, n_nbr=0 if single else 3)
    V = spec.n_nbr + 1
    table, _ = scenarios.load_reference_table()
    if single:
        table = table[rank % table.shape[0]][None].copy()  # every scenario follows one vehicle's plan, alone on the map
    S = args.scenarios
    k0, noise = scenarios.sample_scenarios(S, table, seed=2024 + rank)
    vehicle_sharded = args.parallelism == "vehicle"
    if vehicle_sharded:
        if dist is None or single:
            raise SystemExit("--parallelism vehicle needs torch.distributed (torchrun, or CFZ_BENCH_FORCE_DIST=1) and the mpc4 workload")
        from conflict_rez_amd.distributed import VehicleShardedExchange, VehicleShardedLoop

        S = args.scenarios * world  # all scenarios on every rank, a share of the vehicles each: same solves per GPU
        k0, noise = scenarios.sample_scenarios(S, table, seed=2024)
        ex = VehicleShardedExchange(V)
        eng = engine.Engine(spec, max_batch=S * len(ex.owned), device=local_rank, max_iter=args.max_iter)
        vloop = VehicleShardedLoop(eng, ex, table, k0, noise, device=f"cuda:{local_rank}")
        args.mode = "step"
    else:
        eng = engine.Engine(spec, max_batch=S * V, device=local_rank, max_iter=args.max_iter)
        eng.loop_init(table, k0, noise)

    def barrier():
        if dist is not None:
            import torch

            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()

    persistent = args.mode == "persistent"
    if vehicle_sharded:
        for _ in range(args.warmup):
            vloop.step()
    elif persistent:
        if args.warmup > 0:
            eng.loop_run(args.warmup)  # blocks until all scenarios have done `warmup` iterations
    else:
        for _ in range(args.warmup):
            eng.loop_step()  # blocks until the step is complete on the device
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    n_ok = 0
    if vehicle_sharded:
        ipm_iterations = 0
        for _ in range(args.steps):
            vloop.step()
            kernel_ms += vloop.solve_ms
        launches = args.steps
    elif persistent:
        ipm_iterations = eng.loop_run(args.steps)  # K iterations of every scenario, one launch
        kernel_ms = eng.last_solve_ms()
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
        c = torch.tensor([n_ok], device="cuda", dtype=torch.int64)
        dist.all_reduce(c)
        n_ok = int(c[0])

    if rank == 0:
        B = S * V_local  # solves per iteration on one GPU
        solves = B * world * args.steps
        kern_s = kernel_ms / 1e3 / launches  # average solver-kernel duration per launch
        solves_per_launch = B * args.steps // launches
        achieved = solves_per_launch * ALG_BYTES_PER_SOLVE / kern_s / 1e9
        traffic, traffic_src = (None, None)
        if persistent and not vehicle_sharded and args.steps == 20 and S == 1024:  # the committed PMC passes are of the default command
            traffic, traffic_src = profiled_traffic("loop_kernel")
        line = {
            "metric": "OBCA MPC-step solves/sec (4 vehicles, N=30)",
            "value": solves / elapsed,
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
                       "parallelism": (f"vehicle-sharded x{world}, all-gather of predictions per iteration" if vehicle_sharded
                                       else f"scenario-sharded x{world}"),
                       "max_iter": args.max_iter, "mode": args.mode, "converged_last_step": n_ok / (B * world),
                       "ipm_iterations_rank0": ipm_iterations,
                       "mean_ipm_iters_last_step": iters_mean, "scenario_steps_per_s": solves / elapsed / V,
                       "lds_bytes_per_instance": eng.kernel_info()[0], "instances_per_cu": eng.kernel_info()[1]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "loop_kernel" if persistent and not vehicle_sharded else "solve_kernel",
                         "kernel_ms_per_launch": kern_s * 1e3, "solves_per_launch": solves_per_launch,
                         "alg_bytes_per_solve": ALG_BYTES_PER_SOLVE,
                         "note": "latency/FP64-issue bound: the iterate lives in LDS, so algorithmic HBM bytes "
                                 "are ~1e-5 of peak by construction (SURVEY.md 8d); see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline and not single:
            line["cpu_baseline"] = cpu_baseline(spec, table)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
