"""Multi-GPU operation of the batched distributed MPC (one process per GPU, `torch.distributed`).

Two partitionings:

A. by scenario (default, what bench.py runs): every rank owns whole scenarios, the neighbour
   exchange of `MultiDistributedFollower.solve` (vehicle_follower.py:636-637) is a device-local
   gather inside `cfz_loop_step` -- no collective on the data path.  `scenario_shard` gives the slice.

B. by vehicle (the ROS deployment of the reference runs one process per vehicle and exchanges
   `VehiclePredictionMsg` x,y,psi arrays over DDS, ros2_ws/.../vehicle_node.py:111-189): every rank
   owns a subset of the vehicles of all scenarios; once per MPC iteration the ranks all-gather the
   predicted x,y,psi [S, V_local, 3, N] of the vehicles they own (RCCL over xGMI with the `nccl`
   backend on GPUs, `gloo` on CPUs), then each builds the neighbour parameters of its own solves.
   `VehicleShardedExchange` implements that exchange; the solver is whatever callable the caller
   passes (the HIP engine's `solve_device` on GPUs).
"""
from typing import List

import numpy as np


def scenario_shard(n_scenarios: int, rank: int, world: int):
    """Contiguous slice of scenarios owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_scenarios, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def owned_vehicles(n_vehicles: int, rank: int, world: int) -> List[int]:
    """Vehicles owned by `rank`: v with v % world == rank (world <= n_vehicles), else rank % n_vehicles
    (several ranks then split the scenarios of one vehicle, see `scenario_shard`)."""
    if world <= n_vehicles:
        return [v for v in range(n_vehicles) if v % world == rank]
    return [rank % n_vehicles]


def advance_one_step(arr, axis=-1):
    """`_adv_onestep` (vehicle_follower.py:413-426) along `axis` for numpy arrays or torch tensors."""
    if isinstance(arr, np.ndarray):
        return np.concatenate([np.take(arr, range(1, arr.shape[axis]), axis), np.take(arr, [-1], axis)], axis)
    import torch

    n = arr.shape[axis]
    idx = torch.clamp(torch.arange(1, n + 1, device=arr.device), max=n - 1)
    return arr.index_select(axis if axis >= 0 else arr.dim() + axis, idx)


def vehicle_grid(n_vehicles: int, rank: int, world: int):
    """Partitioning B as a grid of ranks: `world` <= V -> every rank owns V / world vehicles of ALL scenarios,
    one shard; `world` = V x n_shards -> ranks (v n_shards .. v n_shards + n_shards - 1) own vehicle v, each a contiguous
    shard of the scenarios (BASELINE.json configs[4]: 8 ranks, 4 vehicles: the GPU pair (2v, 2v+1) owns vehicle v, half the
    scenarios each).  Returns (owned vehicles, shard index, n_shards, ranks of this rank's exchange group)."""
    if world <= n_vehicles:
        if n_vehicles % world:
            raise ValueError("world size must divide the vehicle count or be a multiple of it")
        return owned_vehicles(n_vehicles, rank, world), 0, 1, list(range(world))
    if world % n_vehicles:
        raise ValueError("world size must divide the vehicle count or be a multiple of it")
    n_shards = world // n_vehicles
    shard = rank % n_shards
    return [rank // n_shards], shard, n_shards, [v * n_shards + shard for v in range(n_vehicles)]


class VehicleShardedExchange:
    """All-gather of the predictions of a vehicle-sharded batch (partitioning B).  With more ranks than vehicles the
    ranks form one process group per scenario shard (the V ranks that hold the V vehicles of the same scenarios) and
    the all-gather runs inside it: 4 ranks x [S / n_shards, 1, 3, N] on 8 GPUs instead of 8 x [S, ...]."""

    def __init__(self, n_vehicles: int, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.V = n_vehicles
        self.owned, self.shard, self.n_shards, members = vehicle_grid(n_vehicles, self.rank, self.world)
        self.group = group
        if self.n_shards > 1:
            # every rank creates every group, in the same order (torch.distributed's rule); it keeps its own
            for sh in range(self.n_shards):
                g = dist.new_group([v * self.n_shards + sh for v in range(n_vehicles)])
                if sh == self.shard:
                    self.group = g
        self.members = members
        self.group_size = len(members)
        # global vehicle index of every gathered slot: group-rank-major, then local order
        self.slot_vehicle = [v for r in members for v in vehicle_grid(n_vehicles, r, self.world)[0]]

    def scenarios(self, n_scenarios: int):
        """Slice of the scenarios this rank steps (all of them with one shard)."""
        return scenario_shard(n_scenarios, self.shard, self.n_shards)

    def gather(self, local_pred):
        """local_pred [S, V_local, 3, N] (torch tensor, this rank's vehicles) -> [S, V, 3, N] in vehicle order."""
        import torch

        S, Vl, _, N = local_pred.shape
        G = self.group_size
        buf = torch.empty((G,) + tuple(local_pred.shape), dtype=local_pred.dtype, device=local_pred.device)
        if hasattr(self.dist, "all_gather_into_tensor") and local_pred.is_cuda:
            self.dist.all_gather_into_tensor(buf, local_pred.contiguous(), group=self.group)
        else:
            parts = [torch.empty_like(local_pred) for _ in range(G)]
            self.dist.all_gather(parts, local_pred.contiguous(), group=self.group)
            buf = torch.stack(parts, 0)
        flat = buf.permute(1, 0, 2, 3, 4).reshape(S, G * Vl, 3, N)  # slots group-rank-major
        if self.slot_vehicle == sorted(self.slot_vehicle):
            return flat.contiguous()
        order = torch.tensor(np.argsort(self.slot_vehicle), device=flat.device)
        return flat.index_select(1, order).contiguous()

    def neighbour_params(self, all_pred):
        """[S, V, 3, N] gathered predictions -> nbr [S * V_local, V-1, 3, N] for this rank's solves:
        the other vehicles in ascending order, each advanced one step (vehicle_follower.py:444-456)."""
        import torch

        adv = advance_one_step(all_pred, axis=-1)
        rows = []
        for v in self.owned:
            others = [u for u in range(self.V) if u != v]
            rows.append(adv[:, others])
        return torch.stack(rows, 1).reshape(-1, self.V - 1, 3, all_pred.shape[-1])


def rk4_plant(z, u, dt, wb, substeps=10):
    """Plant step (`simulator`, dynamic_model.py:61-93) on torch tensors z [B,5], u [B,2]: RK4, `substeps` sub-steps."""
    import torch

    def f(zz):
        psi, v, de = zz[:, 2], zz[:, 3], zz[:, 4]
        return torch.stack([v * torch.cos(psi), v * torch.sin(psi), v / wb * torch.tan(de), u[:, 0], u[:, 1]], 1)

    h = dt / substeps
    for _ in range(substeps):
        k1 = f(z); k2 = f(z + 0.5 * h * k1); k3 = f(z + 0.5 * h * k2); k4 = f(z + h * k3)
        z = z + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return z


class VehicleShardedLoop:
    """Closed loop of partitioning B on GPUs: this rank owns `exchange.owned` vehicles of its shard of the scenarios; every
    MPC iteration all-gathers the owned predictions inside the rank's exchange group (RCCL) and then runs ONE call of
    `cfz_vsl_step` on torch's current stream: parameters and shifted warm start from the gathered predictions, solve from
    the carried multipliers, read-back / shift fallback, plant -- HIP kernels, no host synchronisation, no torch glue.
    Same Jacobi semantics as `cfz_loop_step` (which keeps all V vehicles of a scenario on one GPU).
    `table`, `k0`, `noise` describe ALL scenarios; the rank keeps its shard."""

    def __init__(self, engine, exchange, table, k0, noise, device="cuda"):
        import torch

        self.torch, self.eng, self.ex = torch, engine, exchange
        self.N, self.dt, self.wb = engine.spec.N, engine.spec.dt, engine.spec.wb
        V, T = table.shape[0], table.shape[1]
        sl = exchange.scenarios(len(k0))
        k0, noise = np.asarray(k0)[sl], np.asarray(noise)[sl]
        self.S, self.T, self.V, self.own = len(k0), T, V, list(exchange.owned)
        dev = torch.device(device)
        self.table = torch.tensor(np.ascontiguousarray(table[self.own]), dtype=torch.float64, device=dev)  # [Vl, T, 7]
        self.k0 = torch.tensor(k0, dtype=torch.int32, device=dev)
        self.d_own = torch.tensor(self.own, dtype=torch.int32, device=dev)
        self.t = 0
        idx = np.minimum(k0[:, None] + np.arange(self.N)[None, :], T - 1)  # [S, N]
        pred0 = np.stack([table[v][idx] for v in self.own], 1).transpose(0, 1, 3, 2)  # [S, Vl, 7, N], as get_current_ref seeds it
        state0 = np.stack([table[v][k0, :5] for v in self.own], 1) + noise[:, self.own]
        self.pred = torch.tensor(np.ascontiguousarray(pred0), dtype=torch.float64, device=dev)
        self.state = torch.tensor(np.ascontiguousarray(state0), dtype=torch.float64, device=dev)
        B = self.S * len(self.own)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(B, dtype=torch.int32, device=dev)
        self.stats = torch.zeros(B * 3, dtype=torch.float64, device=dev)
        self.carry = torch.zeros(B, dtype=torch.int32, device=dev)
        self.solve_ms = 0.0

    def step(self, sync=False):
        """One MPC iteration.  Everything is enqueued on torch's current stream; `sync=True` waits and reads the solver
        kernel's time (`solve_ms`)."""
        torch = self.torch
        allpred = self.ex.gather(self.pred[:, :, :3, :].contiguous())  # [S, V, 3, N], vehicle order
        self.eng.vsl_step(self.S, self.V, self.d_own, self.T, self.table, self.k0, self.t, allpred, self.pred, self.state,
                          self.status, self.iters, self.stats, self.carry, stream=torch.cuda.current_stream().cuda_stream)
        self._keep = allpred  # stays alive until the kernels that read it have run
        self.t += 1
        if sync:
            torch.cuda.synchronize()
            self.solve_ms = self.eng.last_solve_ms()
