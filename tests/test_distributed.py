"""Multi-process paths on CPU (`gloo`, world_size 2): the vehicle-sharded neighbour exchange reproduces the
single-process Jacobi iteration; scenario sharding covers the batch exactly once."""
import os
import socket

import numpy as np
import pytest

from conflict_rez_amd.distributed import advance_one_step, owned_vehicles, scenario_shard


def test_scenario_shards_partition_the_batch():
    for S, W in ((1024, 8), (10, 3), (7, 8)):
        got = np.concatenate([np.arange(S)[scenario_shard(S, r, W)] for r in range(W)])
        assert np.array_equal(got, np.arange(S))
    assert owned_vehicles(4, 1, 2) == [1, 3] and owned_vehicles(4, 3, 4) == [3]


def test_advance_one_step_matches_reference_semantics():
    a = np.arange(12.0).reshape(3, 4)
    assert np.array_equal(advance_one_step(a), [[1, 2, 3, 3], [5, 6, 7, 7], [9, 10, 11, 11]])


def _worker(rank, world, port, S, steps, q):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conflict_rez_amd import scenarios
    from conflict_rez_amd.distributed import VehicleShardedExchange
    from oracle import port as cport
    from oracle.dynamics import plant_step
    from oracle.mpc_nlp import MpcSpec

    spec = scenarios.parking_lot_spec()
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=3)
    table, _ = scenarios.load_reference_table()
    k0, noise = scenarios.sample_scenarios(S, table, seed=9)
    ex = VehicleShardedExchange(4)
    N, T = spec.N, table.shape[1]
    idx = lambda s, t: np.minimum(k0[s] + t + np.arange(N), T - 1)
    state = {(s, v): table[v, k0[s], :5] + noise[s, v] for s in range(S) for v in ex.owned}
    pred = {(s, v): table[v, idx(s, 0)].T.copy() for s in range(S) for v in ex.owned}
    for t in range(steps):
        local = torch.tensor(np.stack([[pred[(s, v)][:3] for v in ex.owned] for s in range(S)]))
        nbr = ex.neighbour_params(ex.gather(local)).numpy().reshape(S, len(ex.owned), 3, 3, N)
        for s in range(S):
            for i, v in enumerate(ex.owned):
                warm = advance_one_step(pred[(s, v)])
                r = cport.solve(ospec, state[(s, v)], table[v, idx(s, t), :3].T.copy(), nbr[s, i], warm.T)
                pred[(s, v)] = r["p"].T.copy() if r["status"] == 0 else warm
                state[(s, v)] = plant_step(state[(s, v)], pred[(s, v)][5:7, 0], spec.dt, spec.wb)
    q.put((rank, {k: v.copy() for k, v in state.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_vehicle_sharded_exchange_gloo_world2():
    import torch.multiprocessing as mp

    S, steps = 2, 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, S, steps, q)) for r in range(2)]
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

    spec = scenarios.parking_lot_spec()
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=3)
    table, _ = scenarios.load_reference_table()
    k0, noise = scenarios.sample_scenarios(S, table, seed=9)
    N, T = spec.N, table.shape[1]
    for s in range(S):
        idx = lambda t: np.minimum(k0[s] + t + np.arange(N), T - 1)
        state = [table[v, k0[s], :5] + noise[s, v] for v in range(4)]
        pred = [table[v, idx(0)].T.copy() for v in range(4)]
        for t in range(steps):
            old = [p.copy() for p in pred]
            for v in range(4):
                nbr = np.stack([advance_one_step(old[u])[:3] for u in range(4) if u != v])
                warm = advance_one_step(old[v])
                r = cport.solve(ospec, state[v], table[v, idx(t), :3].T.copy(), nbr, warm.T)
                pred[v] = r["p"].T.copy() if r["status"] == 0 else warm
                state[v] = plant_step(state[v], pred[v][5:7, 0], spec.dt, spec.wb)
        for v in range(4):
            assert np.array_equal(got[(s, v)], state[v]), (s, v)
