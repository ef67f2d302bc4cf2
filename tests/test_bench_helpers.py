"""bench.py's host-side pieces that need no GPU: the committed profile summaries parse into the `roofline` fields, the CPU
baseline leg (the same closed loop through the oracle's C port) runs on a tiny sample, the CasADi probe reports a real import
outcome.  (The GPU leg of bench.py is what the driver runs.)"""
import bench
import pytest


def test_fp64_counters_to_flops():
    """`bench.fp64_from_counters`: wave-level FP64 instruction counts x 64 lane-slots (FMA twice) + MFMA operations x 512; the matrix pipe's
    busy cycles over the SIMD-cycles of the dispatch; FP64 arithmetic among the vector instructions.  And the committed passes parse."""
    c = {"SQ_INSTS_VALU": 1000.0, "SQ_INSTS_VALU_FMA_F64": 100.0, "SQ_INSTS_VALU_ADD_F64": 50.0, "SQ_INSTS_VALU_MUL_F64": 30.0, "SQ_INSTS_VALU_TRANS_F64": 4.0,
         "SQ_INSTS_VALU_MFMA_F64": 10.0, "SQ_INSTS_VALU_MFMA_MOPS_F64": 40.0, "SQ_VALU_MFMA_BUSY_CYCLES": 512.0, "GRBM_GUI_ACTIVE_MFMA": 8.0 * 100.0}
    f = bench.fp64_from_counters(c)
    assert f["vector_flops"] == (2 * 100 + 50 + 30 + 4) * 64 and f["mfma_flops"] == 40 * 512 and f["flops"] == f["vector_flops"] + f["mfma_flops"]
    assert abs(f["valu_useful_frac"] - (100 + 50 + 30 + 4 + 10) / 1000.0) < 1e-15 and abs(f["mfma_busy_frac"] - 512.0 / (100.0 * 1024.0)) < 1e-15
    got, tag = bench.profiled_fp64("loop_kernel", require_current=False)
    if got is not None:  # (profiles of round 6 on: the FP64 / MFMA passes of tools/gpu_profile_job.sh)
        assert 0.2e6 < got["flop_per_ipm_iteration"] < 10e6 and 0.0 < got["valu_useful_frac"] < 1.0 and 0.0 < got["mfma_busy_frac"] < 0.5 and tag


def test_profile_summaries_parse(monkeypatch):
    traffic, tag = bench.profiled_traffic("loop_kernel", require_current=False)
    assert tag is not None and 1e8 < traffic < 1e11   # bytes per launch of the default command
    sq, tag2 = bench.profiled_sq("loop_kernel", require_current=False)
    assert tag2 == tag and 0.0 < sq["raw_quotient"] < 1.0 and abs(sq["frac"] - 4.0 * sq["raw_quotient"]) < 1e-12
    assert bench.profiled_traffic("no_such_kernel", require_current=False) == (None, tag)
    # counters are quoted only when the profile was taken on a library built from the sources loaded now (profiles/<tag>_meta.json
    # against cfz_source_hash): with another hash they are dropped and the line says why
    from conflict_rez_amd import engine

    monkeypatch.setattr(engine, "source_hash", lambda: "0123456789abcdef")
    t2, why = bench.profiled_traffic("loop_kernel")
    assert t2 is None and "stale" in why and bench.profiled_sq("loop_kernel")[0] is None


def test_cpu_closed_loop_worker_and_probe():
    n, its, ok, secs, cold = bench._cpu_closed_loop_worker((0, 2, 1, 2, "planned", True))   # seed 0, 2 scenarios, 1 warm-up + 2 timed iterations
    assert n == 2 * 4 * 2 and 0 < ok <= n and its >= ok and secs > 0.0
    assert cold[0] == 8 and cold[1] >= 8
    assert "casadi" in bench.casadi_probe().lower()


def test_cpu_baseline_reports_the_cores_it_used():
    """`cpu_baseline` of the bench line: one process per usable core (affinity mask capped by the cgroup quota), the CPU model string,
    a bounded sample of the same closed loop (here 2 + 2 iterations)."""
    import bench

    n, info = bench.usable_cores()
    assert 1 <= n <= info["affinity"] <= info["logical"] and isinstance(info["model"], str) and info["model"]
    cb = bench.cpu_baseline(2, 2, n_scen_per_core=4, max_cores=4)
    assert cb["kind"] == "port" and cb["unit"] == "solves/s" and 1 <= cb["cores"] <= min(4, n) and cb["value"] > 10.0
    assert cb["cpu"]["model"] == info["model"] and cb["value_converged"] <= cb["value"] and "usable" in cb["sample"]


def test_lane_sampler_and_elimination_info():
    """BASELINE.md section 4's config-2 draw (`scenarios.lane_sampler`): the poses are the stated distributions under default_rng(1234),
    the references stay in the lane; `cfz_colloc_elimination_info` (host arithmetic) prices the eliminations that are actually run."""
    import numpy as np

    from conflict_rez_amd import engine, scenarios

    spec = scenarios.parking_lot_spec(n_obs=4, n_nbr=0)
    x0, ref, zu = scenarios.lane_sampler(spec)
    assert x0.shape == (256, 5) and ref.shape == (256, 3, 30) and zu.shape == (256, 7, 30)
    rng = np.random.default_rng(1234)
    assert np.array_equal(x0[:, 0], rng.uniform(5.0, 30.0, 256)) and np.array_equal(x0[:, 1], rng.uniform(15.0, 20.0, 256))
    assert 5.0 <= x0[:, 0].min() and x0[:, 0].max() <= 30.0 and np.abs(x0[:, 3]).max() <= 1.0 and np.all(x0[:, 4] == 0.0)
    off = np.minimum(np.abs(x0[:, 2]), np.abs(x0[:, 2] - np.pi))
    assert off.max() < 0.25 and 14.9 < ref[:, 1].min() and ref[:, 1].max() < 20.1
    assert np.allclose(np.hypot(np.diff(ref[:, 0], axis=1), np.diff(ref[:, 1], axis=1)), 0.1, atol=1e-3)  # 1 m/s along the lane
    one = engine.colloc_elimination_info([11])
    band = engine.colloc_elimination_info([11], structured=0)
    assert (band["nk"], band["kb"]) == (engine.colloc_band_info([11])[0], 51)
    assert (one["nk"], one["kb"]) == (band["nk"] - 16 * 10, 51) and one["alg_bytes"] < band["alg_bytes"]   # (tube rows condensed: 16 per strategy step)
    with pytest.raises(RuntimeError, match="structured"):   # round 4's scheme (2) left the library in round 6
        engine.colloc_elimination_info([11], structured=2)
    # a plan of more than 255 intervals per vehicle does not fit the structured elimination's packed interval counts: the host routes it to
    # the band elimination where the slab is sized, instead of the device finding out and failing every Newton system
    long_s, long_b = engine.colloc_elimination_info([53]), engine.colloc_elimination_info([53], structured=0)
    assert long_s == long_b and long_s["nk"] == engine.colloc_band_info([53])[0]
    assert engine.colloc_elimination_info([52])["nk"] == engine.colloc_band_info([52])[0] - 16 * 51   # 255 intervals: still structured
    four = engine.colloc_elimination_info([11, 7, 7, 9])
    assert four["kb"] == 51 and four["nk"] == engine.colloc_band_info([11, 7, 7, 9])[0] - 16 * 30 and four["workspace_bytes"] < 60e6
    assert bench.profiled_extras(0, require_current=False) is None or "traffic" in bench.profiled_extras(0, require_current=False)
