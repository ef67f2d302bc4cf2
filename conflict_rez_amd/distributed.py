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


class VehicleShardedExchange:
    """All-gather of the predictions of a vehicle-sharded batch (partitioning B)."""

    def __init__(self, n_vehicles: int, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if n_vehicles % self.world and self.world % n_vehicles:
            raise ValueError("world size must divide the vehicle count or be a multiple of it")
        if self.world > n_vehicles:
            raise NotImplementedError("vehicle x scenario sharding: use one group per scenario shard")
        self.V = n_vehicles
        self.owned = owned_vehicles(n_vehicles, self.rank, self.world)
        # global vehicle index of every gathered slot: rank-major, then local order
        self.slot_vehicle = [v for r in range(self.world) for v in owned_vehicles(n_vehicles, r, self.world)]

    def gather(self, local_pred):
        """local_pred [S, V_local, 3, N] (torch tensor, this rank's vehicles) -> [S, V, 3, N] in vehicle order."""
        import torch

        S, Vl, _, N = local_pred.shape
        buf = torch.empty((self.world,) + tuple(local_pred.shape), dtype=local_pred.dtype, device=local_pred.device)
        if hasattr(self.dist, "all_gather_into_tensor") and local_pred.is_cuda:
            self.dist.all_gather_into_tensor(buf, local_pred.contiguous(), group=self.group)
        else:
            parts = [torch.empty_like(local_pred) for _ in range(self.world)]
            self.dist.all_gather(parts, local_pred.contiguous(), group=self.group)
            buf = torch.stack(parts, 0)
        flat = buf.permute(1, 0, 2, 3, 4).reshape(S, self.world * Vl, 3, N)  # slots rank-major
        order = torch.tensor(np.argsort(self.slot_vehicle), device=flat.device)
        return flat.index_select(1, order)

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
    """Closed loop of partitioning B on GPUs: this rank owns `exchange.owned` vehicles of all S scenarios; every MPC
    iteration all-gathers the owned predictions (RCCL), builds the neighbour parameters, solves its S * V_local NLPs
    with the HIP engine on device tensors (`Engine.solve_device`), applies read-back / shift fallback and the plant.
    Same Jacobi semantics as `cfz_loop_step` (which keeps all V vehicles of a scenario on one GPU)."""

    def __init__(self, engine, exchange, table, k0, noise, device="cuda"):
        import torch

        self.torch, self.eng, self.ex = torch, engine, exchange
        self.N, self.dt, self.wb = engine.spec.N, engine.spec.dt, engine.spec.wb
        V, T = table.shape[0], table.shape[1]
        self.S, self.T, self.own = len(k0), T, list(exchange.owned)
        dev = torch.device(device)
        self.table = torch.tensor(table[self.own], dtype=torch.float64, device=dev)  # [Vl, T, 7]
        self.k0 = torch.tensor(np.asarray(k0), dtype=torch.long, device=dev)
        self.t = 0
        self.state = self._rows(torch.zeros(self.N, dtype=torch.long, device=dev)[:1] * 0)[..., 0, :5] + torch.tensor(
            np.asarray(noise)[:, self.own], dtype=torch.float64, device=dev)
        self.pred = self._rows(torch.arange(self.N, device=dev)).permute(0, 1, 3, 2).contiguous()  # [S, Vl, 7, N]
        B = self.S * len(self.own)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(B, dtype=torch.int32, device=dev)
        self.stats = torch.zeros(B * 3, dtype=torch.float64, device=dev)
        self.solve_ms = 0.0

    def _rows(self, offs):
        """Reference rows k0[s] + t + offs (clipped to the table) of the owned vehicles -> [S, Vl, len(offs), 7]."""
        idx = (self.k0[:, None] + self.t + offs[None, :]).clamp(max=self.T - 1)  # [S, n]
        return self.table[:, idx].permute(1, 0, 2, 3)  # table[Vl, S, n, 7] -> [S, Vl, n, 7]

    def step(self):
        torch, S, Vl, N = self.torch, self.S, len(self.own), self.N
        nbr = self.ex.neighbour_params(self.ex.gather(self.pred[:, :, :3, :].contiguous())).contiguous()  # [S*Vl, V-1, 3, N]
        ref = self._rows(torch.arange(N, device=self.pred.device))[..., :3].permute(0, 1, 3, 2).reshape(S * Vl, 3, N).contiguous()
        warm = advance_one_step(self.pred, axis=-1).reshape(S * Vl, 7, N).contiguous()
        zu = warm.clone()
        x0 = self.state.reshape(S * Vl, 5).contiguous()
        torch.cuda.synchronize()
        if self.t > 0:
            self.eng.set_carry(np.ones(S * Vl, np.int32))  # slot b is the same vehicle as in the previous iteration
        self.eng.solve_device(S * Vl, x0, ref, nbr, zu, self.status, self.iters, self.stats,
                              stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        self.solve_ms = self.eng.last_solve_ms()
        ok = (self.status == 0)[:, None, None]
        new = torch.where(ok, zu, warm)
        self.pred = new.reshape(S, Vl, 7, N)
        self.state = rk4_plant(x0, new[:, 5:7, 0], self.dt, self.wb).reshape(S, Vl, 5)
        self.t += 1
