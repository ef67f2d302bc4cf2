"""bench.py on the GPU through its N > 1 code paths with one rank (`CFZ_BENCH_FORCE_DIST=1`: torch.distributed / RCCL initialised,
barriers and all-reduces taken): scenario sharding and the vehicle-sharded exchange (`--parallelism vehicle`, the layout of
BASELINE.json configs[4]).  The driver's own SCALE runs need an 8-GPU node; this keeps the code those runs take from rotting."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CFZ_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--scenarios", "64",
                          "--no-cpu-baseline", "--no-extras"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--parallelism", "vehicle"]])
def test_bench_line_through_torch_distributed(extra):
    """The bench line through torch.distributed on one GPU, both sharding modes: the contract's fields and what round 4 added --
    `value` counts CONVERGED solves (`value_all` every solve), `config.status_counts` says how the timed solves ended,
    `config.parked_fraction` how many of them are for a vehicle at the end of its plan, `config.rccl_ranks` (an all-reduce of ones)
    how many RCCL ranks took part, `extra.seeds` the same measurement on three sampler seeds and `extra.all_moving` a sample without
    parked vehicles (persistent mode)."""
    b = _run(extra)
    assert b["metric"].startswith("OBCA MPC-step solves/sec") and b["unit"] == "solves/s" and b["n_gpus"] == 1 and b["steps"] == 3 and b["warmup"] == 2
    assert b["value"] > 1e3 and b["scaling"] == "weak" and b["dtype"] == "f64" and b["vs_baseline"] is None
    assert abs(b["value_all"] - 64 * 4 * 3 / (b["ms_per_step"] * 3e-3)) < 1e-6 * b["value_all"]  # value_all = solves / elapsed
    assert 0.5 * b["value_all"] < b["value"] <= b["value_all"]  # value = converged solves / elapsed
    c = b["config"]
    assert c["scenarios_per_gpu"] == 64 and c["solves_per_step_per_gpu"] == 256 and c["infeasible_starts"] == 0
    assert ("vehicle-sharded" in c["parallelism"]) == bool(extra) and "refs_4v_planned" in c["reference_plan"]
    assert c["rccl_ranks"] == 1 and 0.2 < c["parked_fraction"] < 0.7
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0 and r["kernel"] == ("solve_kernel" if extra else "loop_kernel")
    # round 6: flops per interior-point iteration from the FP64 instruction counters when a pass of these sources is
    # committed (otherwise SURVEY's estimate, and the line says which), the matrix pipe's busy fraction and the share of FP64 arithmetic
    # among the vector instructions
    assert {"flop_per_ipm_iteration", "flop_source", "flop_per_ipm_iteration_estimate", "mfma_busy_frac", "mfma_flops_share", "valu_useful_frac"} <= set(r)
    assert r["flop_per_ipm_iteration_estimate"] == 0.75e6 and ("measured" in r["flop_source"] or "estimate" in r["flop_source"])
    if not extra:  # persistent launch: counted on the device over the whole timed region
        sc = c["status_counts"]
        assert sum(sc.values()) == 64 * 4 * 3 and abs(b["value"] / b["value_all"] - sc["0 converged"] / (64 * 4 * 3)) < 1e-9
        assert abs(c["converged_timed_region"] - sc["0 converged"] / (64 * 4 * 3)) < 1e-12
        sd, am = b["extra"]["seeds"], b["extra"]["all_moving"]
        assert sd["seeds"] == [2024, 2025, 2026] and sd["min"] <= sd["median"] <= sd["max"] and len(sd["runs"]) == 3
        assert all(sum(r_["status_counts"]) == 64 * 4 * 3 and r_["value"] <= r_["value_all"] for r_ in sd["runs"])
        assert am["parked_fraction"] == 0.0 and am["value"] > 1e3 and am["mean_ipm_iters"] > sd["runs"][0]["mean_ipm_iters"]


def test_long_persistent_launch_at_4096_scenarios():
    """Regression of the round-3 memory fault: 25 closed-loop iterations of 4096 scenarios on the planned table in ONE persistent
    launch.  Long solves fill the filter; shifting a full filter while the other wavefront still compared against it made the two
    disagree on a step, run different numbers of reductions and read the work item of the persistent loop out of a reduction's
    exchange words (a wild instance index: `Memory access fault`, process aborted -- hence the child process).  Also: the launch
    does the same arithmetic as the stepwise path (equal iteration totals on a smaller batch are tested in test_gpu_parity.py)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fault_probe.py"), "4096", "25"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]
    assert "Memory access fault" not in out.stderr


def test_planning_extras_of_the_bench_line():
    """`bench.py`'s `extra` objects (configs[1]: single plans, configs[3]: four-vehicle joint plans) at a small batch: every plan
    converges, the objects carry their own roofline (band bytes from `cfz_colloc_band_info`) and -- with `cpu=True` in the real run
    -- a CPU baseline; here the GPU half only (the CPU build of the planning source takes a minute per joint plan set)."""
    sys.path.insert(0, ROOT)
    import bench

    ex = bench.planning_extras(device=0, B=8, cpu=False)
    c1, c3 = ex["configs[1]"], ex["configs[3]"]
    # configs[1] is BASELINE.json's wording since round 6 (FOUR polytope obstacles); the reference's own six-obstacle map is nested under it
    assert "4 polytope obstacles" in c1["workload"] and "SIX obstacles" in c1["six_obstacles"]["workload"]
    assert c1["state_ws_converged"] == 8 and c1["colloc_converged"] == 8 and c1["plans_per_s"] > 1.0
    c6 = c1["six_obstacles"]
    assert c6["colloc_converged"] == 8 and c6["plans_per_s"] > 1.0
    assert 5 <= c1["state_ws_iters_mean"] <= c1["state_ws_iters_max"] <= 60 and c6["colloc_iters_mean"] <= c6["colloc_iters_max"] <= 150
    assert c1["colloc_iters_max"] >= c1["colloc_iters_mean"]
    # the launch lasts as long as its slowest plan, and which plan wanders is decided in the last digits of its guess: the lines carry the
    # three longest plans, the time per iteration of the slowest one (the figure that compares builds) and the rate at which 95 % of the
    # batch was done -- measured: the same launch stopped at the iteration count that 95 % of the plans need
    assert c1["colloc_iters_top3"][0] == c1["colloc_iters_max"] and c1["colloc_iters_top3"] == sorted(c1["colloc_iters_top3"], reverse=True)
    assert abs(c1["ms_per_iteration_of_the_slowest_plan"] - 1e3 * c1["colloc_s"] / c1["colloc_iters_max"]) < 1e-9 and "several minimisers" in c1["note"]
    assert abs(c6["ms_per_iteration_of_the_slowest_plan"] - 1e3 * c6["colloc_s"] / c6["colloc_iters_max"]) < 1e-9
    p95 = c1["p95"]
    assert p95["max_iter"] <= c1["colloc_iters_max"] and 0.95 * 8 <= p95["converged"] <= 8 and p95["colloc_s"] <= 1.5 * c1["colloc_s"] and p95["plans_per_s"] > 1.0
    # configs[3] goes through the structured elimination (cfz_jstruct.inl): vehicle-major ordering, tube rows condensed, half-bandwidth 51,
    # no band across the vehicles (round 4: 12,350 unknowns in a band of half-bandwidth 298, 88 MB per plan)
    from conflict_rez_amd import engine

    assert c3["converged"] == 8 and c3["unknowns"] == 12350 - 16 * 30 and c3["half_bandwidth"] == 51 and c3["band_bytes"] == c3["unknowns"] * (2 * 51 + 1) * 8
    assert c3["iters_max"] <= 120 and "structured" in c3["elimination"] and c3["workspace_bytes_per_plan"] < 60e6
    assert c3["iters_top3"][0] == c3["iters_max"] and abs(c3["ms_per_iteration_of_the_slowest_plan"] - 1e3 * c3["joint_s"] / c3["iters_max"]) < 1e-9
    assert c3["p95"]["max_iter"] <= c3["iters_max"] and 0.95 * 8 <= c3["p95"]["converged"] <= 8 and c3["p95"]["plans_per_s"] > 0.5
    info4 = engine.colloc_elimination_info([11, 7, 7, 9])  # the four vehicles' strategy lengths
    assert (info4["nk"], info4["kb"]) == (c3["unknowns"], 51)
    band4 = engine.colloc_elimination_info([11, 7, 7, 9], structured=0)
    assert (band4["nk"], band4["kb"], band4["alg_bytes"]) == (12350, 298, 3 * 12350 * (3 * 298 + 1) * 8) and info4["alg_bytes"] < band4["alg_bytes"] / 4
    for c in (c1, c6, c3):
        r = c["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0 and r["unit"] == "GB/s" and r["kernel"] == "colloc_kernel"
        # `frac` follows from the line's own numbers: algorithmic bytes of the eliminations run / the launch time / the roof
        secs = c["joint_s"] if c is c3 else c["colloc_s"]
        assert abs(r["frac"] - r["alg_bytes"] / secs / 1e9 / 8000.0) < 1e-12 and "cfz_colloc_elimination_info" in r["alg_bytes_definition"]
        assert "traffic" in r and "valu_active_frac" in r and "traffic_source" in r  # (filled when profiles/<tag>_extras_* of these sources exist)
        assert {"mfma_busy_frac", "valu_useful_frac", "fp64_tflops", "mfma_flops_share"} <= set(r)  # (round 6: the FP64 / MFMA counter passes)
    # BASELINE.md section 4's config-2 draw (lane poses, default_rng(1234), MPC form, four obstacles): its own line, checked against the port
    ls = c1["lane_sampler"]
    assert "default_rng(1234)" in ls["workload"] and ls["converged"] >= 250 and ls["solves_per_s_kernel"] >= ls["solves_per_s"] > 1e3
    assert ls["cpu_baseline"]["kind"] == "port" and ls["cpu_baseline"]["same_status_and_iterations"] == 256 and 3 <= ls["iters_mean"] <= ls["iters_max"] <= 200
