"""Multi-process paths on CPU (`gloo`): the vehicle-sharded neighbour exchange reproduces the single-process Jacobi
iteration -- world 2 (two vehicles per rank, one group) and world 4 over two vehicles (vehicle x scenario grid: two
exchange groups of two ranks, the layout of BASELINE.json configs[4] in small) and world 8 over four vehicles (that layout itself);
scenario sharding covers the batch exactly once."""
import os
import socket

import numpy as np
import pytest

from conflict_rez_amd.distributed import advance_one_step, owned_vehicles, scenario_shard, vehicle_grid


def test_scenario_shards_partition_the_batch():
    for S, W in ((1024, 8), (10, 3), (7, 8)):
        got = np.concatenate([np.arange(S)[scenario_shard(S, r, W)] for r in range(W)])
        assert np.array_equal(got, np.arange(S))
    assert owned_vehicles(4, 1, 2) == [1, 3] and owned_vehicles(4, 3, 4) == [3]


def test_vehicle_grid_covers_every_vehicle_and_scenario_once():
    """SURVEY.md 8e partitioning B: 8 ranks x 4 vehicles -> the pair (2v, 2v+1) owns vehicle v, half the scenarios each;
    the exchange group of a rank is the four ranks holding the other vehicles of the SAME scenarios."""
    assert vehicle_grid(4, 5, 8) == ([2], 1, 2, [1, 3, 5, 7]) and vehicle_grid(4, 4, 8) == ([2], 0, 2, [0, 2, 4, 6])
    for V, W in ((4, 1), (4, 2), (4, 4), (4, 8), (4, 16), (2, 4)):
        seen = {}
        for r in range(W):
            own, shard, n_shards, members = vehicle_grid(V, r, W)
            assert r in members and len(members) * len(own) == V
            for v in own:
                for sc in range(64)[scenario_shard(64, shard, n_shards)]:
                    assert (v, sc) not in seen
                    seen[(v, sc)] = r
            for q in members:  # same shard, and together all vehicles
                assert vehicle_grid(V, q, W)[1] == shard
            assert sorted(v for q in members for v in vehicle_grid(V, q, W)[0]) == list(range(V))
        assert len(seen) == V * 64
    with pytest.raises(ValueError):
        vehicle_grid(4, 0, 3)


def test_advance_one_step_matches_reference_semantics():
    a = np.arange(12.0).reshape(3, 4)
    assert np.array_equal(advance_one_step(a), [[1, 2, 3, 3], [5, 6, 7, 7], [9, 10, 11, 11]])


def _worker(rank, world, port, S, steps, q, V=4):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conflict_rez_amd import scenarios
    from conflict_rez_amd.distributed import VehicleShardedExchange
    from oracle import port as cport
    from oracle.dynamics import plant_step
    from oracle.mpc_nlp import MpcSpec

    spec = scenarios.parking_lot_spec(n_nbr=V - 1)
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=V - 1)
    table, _ = scenarios.load_reference_table()
    table = table[:V]
    k0, noise = scenarios.sample_scenarios(S, table, seed=9)
    ex = VehicleShardedExchange(V)
    mine = list(range(S)[ex.scenarios(S)])  # this rank's shard of the scenarios (all of them when world <= V)
    N, T = spec.N, table.shape[1]
    idx = lambda s, t: np.minimum(k0[s] + t + np.arange(N), T - 1)
    state = {(s, v): table[v, k0[s], :5] + noise[s, v] for s in mine for v in ex.owned}
    pred = {(s, v): table[v, idx(s, 0)].T.copy() for s in mine for v in ex.owned}
    for t in range(steps):
        local = torch.tensor(np.stack([[pred[(s, v)][:3] for v in ex.owned] for s in mine]))
        nbr = ex.neighbour_params(ex.gather(local)).numpy().reshape(len(mine), len(ex.owned), V - 1, 3, N)
        for si, s in enumerate(mine):
            for i, v in enumerate(ex.owned):
                warm = advance_one_step(pred[(s, v)])
                r = cport.solve(ospec, state[(s, v)], table[v, idx(s, t), :3].T.copy(), nbr[si, i], warm.T)
                pred[(s, v)] = r["p"].T.copy() if r["status"] == 0 else warm
                state[(s, v)] = plant_step(state[(s, v)], pred[(s, v)][5:7, 0], spec.dt, spec.wb)
    q.put((rank, {k: v.copy() for k, v in state.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,V", [(2, 4), (4, 2), (8, 4)])
def test_vehicle_sharded_exchange_gloo(world, V):
    """(8, 4): the grid of BASELINE.json configs[4] itself -- eight ranks, four vehicles, two scenario shards: rank pair (2v, 2v+1)
    owns vehicle v, the exchange groups are the four ranks holding the same scenarios (`dist.new_group` x 2, all-gather inside)."""
    import torch.multiprocessing as mp

    S, steps = (4, 2) if world == 8 else (2, 3)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, steps, q, V)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        got.update(q.get(timeout=300)[1])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process replay of the same Jacobi iteration
    from conflict_rez_amd import scenarios
    from oracle import port as cport
    from oracle.dynamics import plant_step
    from oracle.mpc_nlp import MpcSpec

    assert len(got) == S * V  # every (scenario, vehicle) stepped by exactly one rank
    spec = scenarios.parking_lot_spec(n_nbr=V - 1)
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=V - 1)
    table, _ = scenarios.load_reference_table()
    table = table[:V]
    k0, noise = scenarios.sample_scenarios(S, table, seed=9)
    N, T = spec.N, table.shape[1]
    for s in range(S):
        idx = lambda t: np.minimum(k0[s] + t + np.arange(N), T - 1)
        state = [table[v, k0[s], :5] + noise[s, v] for v in range(V)]
        pred = [table[v, idx(0)].T.copy() for v in range(V)]
        for t in range(steps):
            old = [p.copy() for p in pred]
            for v in range(V):
                nbr = np.stack([advance_one_step(old[u])[:3] for u in range(V) if u != v])
                warm = advance_one_step(old[v])
                r = cport.solve(ospec, state[v], table[v, idx(t), :3].T.copy(), nbr, warm.T)
                pred[v] = r["p"].T.copy() if r["status"] == 0 else warm
                state[v] = plant_step(state[v], pred[v][5:7, 0], spec.dt, spec.wb)
        for v in range(V):
            assert np.array_equal(got[(s, v)], state[v]), (s, v)
