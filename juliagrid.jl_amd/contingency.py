"""Batched N-1 contingency screening on top of the batched Newton-Raphson analysis.

The reference has no batch API: users loop `updateBranch!(analysis; label, status = 0)` ->
`setInitialPoint!` -> `powerFlow!` -> `updateBranch!(...; status = 1)`
(/root/reference/src/powerSystem/branch.jl:453-459, src/powerFlow/acPowerFlow.jl:1226-1249, 1389-1433;
SURVEY.md 3.5).  Here every scenario of that loop is one lane of the batch: the outage is expressed as
the 4 Ybus edits acNodalUpdate! would make (model.jl:93-101) on top of the shared base matrix, the
Jacobian pattern is shared (stored zeros, model.jl:70-71), and all scenarios advance together.
Scenarios shard across GPUs contiguously with no communication (see shard()).
"""
from __future__ import annotations

import numpy as np

import threading

from .powerflow import AcPowerFlow, newtonRaphson, powerFlow_, setOutage_, setOutages_, _push_voltage
from .system import PowerSystem


def bridges(system: PowerSystem) -> np.ndarray:
    """Boolean mask over branches: True where removing the (in-service) branch islands the grid.
    Iterative DFS low-link; parallel in-service branches are never bridges."""
    n, nb = system.bus.number, system.branch.number
    f = system.branch.layout.from_ - 1
    t = system.branch.layout.to - 1
    on = np.flatnonzero(system.branch.layout.status == 1)
    adj = [[] for _ in range(n)]
    for k in on:
        adj[f[k]].append((int(t[k]), int(k)))
        adj[t[k]].append((int(f[k]), int(k)))
    disc = np.full(n, -1)
    low = np.zeros(n, dtype=np.int64)
    is_bridge = np.zeros(nb, dtype=bool)
    timer = 0
    for root in range(n):
        if disc[root] >= 0:
            continue
        stack = [(root, -1, 0)]
        disc[root] = low[root] = timer
        timer += 1
        while stack:
            v, pe, idx = stack.pop()
            if idx < len(adj[v]):
                stack.append((v, pe, idx + 1))
                u, e = adj[v][idx]
                if e == pe:
                    continue
                if disc[u] < 0:
                    disc[u] = low[u] = timer
                    timer += 1
                    stack.append((u, e, 0))
                else:
                    low[v] = min(low[v], disc[u])
            elif stack:
                p = stack[-1][0]
                low[p] = min(low[p], low[v])
                if low[v] > disc[p]:
                    is_bridge[pe] = True
    return is_bridge


def outageList(system: PowerSystem, count: int, seed: int = 512) -> np.ndarray:
    """`count` branch labels (1-based) drawn by a seeded shuffle of the non-bridge in-service branches
    (BASELINE config 5); wraps around if the grid has fewer candidates."""
    ok = np.flatnonzero((system.branch.layout.status == 1) & ~bridges(system)
                        & (system.branch.layout.from_ != system.branch.layout.to))
    rng = np.random.default_rng(seed)
    rng.shuffle(ok)
    reps = -(-count // ok.size)
    return (np.tile(ok, reps)[:count] + 1).astype(np.int64)


def shard(count: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block of scenarios owned by `rank`: [lo, hi)."""
    per = -(-count // world)
    lo = min(rank * per, count)
    return lo, min(lo + per, count)


def contingencyAnalysis(system: PowerSystem, labels, device: int = 0) -> AcPowerFlow:
    """Batched analysis with scenario s = outage of branch labels[s] (None / 0 = base case)."""
    labels = list(labels)
    an = newtonRaphson(system, batch=len(labels), device=device, max_patch=4)
    setOutages_(an, [int(lab) if lab else 0 for lab in labels])
    return an


class ContingencyPipeline:
    """`inflight` batches of the same grid in flight on one GPU, each on its own handle, HIP stream and host thread.

    Why: the sparse LU replays ~100 dependency levels per iteration and most of them occupy a fraction of the chip for
    a few microseconds (latency bound), and the last iterations of a batch run on the few scenarios that have not
    converged yet.  A second and third batch fill those holes: kernels of different streams run concurrently.  Measured
    on MI355X, case_ACTIVSg10k, 512 scenarios per batch: 131k NR iterations/s with 1 batch in flight, 215k with 3.

    jobs are processed in order; `on_done(job, analysis)` (optional) is called on the CALLER's thread in job order while
    the batch's results are still resident (this is where a sharded run issues its RCCL gather: collectives must be
    issued in the same order on every rank)."""

    def __init__(self, system: PowerSystem, batch: int, inflight: int = 3, device: int = 0, start=None):
        self.system, self.batch = system, int(batch)
        self.handles = [newtonRaphson(system, batch=self.batch, device=device, max_patch=4) for _ in range(max(1, int(inflight)))]
        if start is not None:
            self.setStart(*start)
        else:                                            # default restart point of every solve: the start newtonRaphson() built
            for an in self.handles:
                an.snapshot_voltage()

    def setStart(self, magnitude, angle):
        """Start point of every solve (e.g. the base-case solution), kept in HBM."""
        for an in self.handles:
            _push_voltage(an, magnitude, angle)
            an.snapshot_voltage()

    def close(self):
        for an in self.handles:
            an.close()
        self.handles = []

    def run(self, jobs, iteration: int = 20, tolerance: float = 1e-8, on_done=None, fetch: bool = False):
        """jobs: sequence of label lists (one batch each; None = keep the handle's current outages).
        Returns per-job (iterations, status) arrays."""
        jobs = list(jobs)
        results = [None] * len(jobs)
        done = [threading.Event() for _ in jobs]
        released = [threading.Event() for _ in jobs]
        nh = len(self.handles)
        errors = []

        def worker(k):
            try:
                for j in range(k, len(jobs), nh):
                    if j - nh >= 0:
                        released[j - nh].wait()                  # the caller has consumed this handle's previous results
                    an = self.handles[k]
                    if jobs[j] is not None:
                        labels = [int(x) if x else 0 for x in jobs[j]]
                        setOutages_(an, labels + [0] * (self.batch - len(labels)))
                    an.restore_voltage()
                    powerFlow_(an, iteration=iteration, tolerance=tolerance, fetch=fetch)
                    results[j] = (np.array(an.method.iteration), np.array(an.status))
                    done[j].set()
            except BaseException as e:                             # surface in the caller, never hang it
                errors.append(e)
                for ev in done:
                    ev.set()

        threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(nh)]
        for t in threads:
            t.start()
        for j in range(len(jobs)):
            done[j].wait()
            if errors:
                for ev in released:
                    ev.set()
                break
            if on_done is not None:
                on_done(j, self.handles[j % nh])
            released[j].set()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results

    def screen(self, labels, iteration: int = 20, tolerance: float = 1e-8):
        """N-1 screen of an arbitrary list of branch labels: (iterations, status) per label, in order."""
        labels = [int(x) for x in labels]
        jobs = [labels[i:i + self.batch] for i in range(0, len(labels), self.batch)]
        res = self.run(jobs, iteration, tolerance)
        it = np.concatenate([r[0][:len(j)] for r, j in zip(res, jobs)])
        st = np.concatenate([r[1][:len(j)] for r, j in zip(res, jobs)])
        return it, st


def gatherResults(dist, packed):
    """Final gather of a sharded batch: ONE collective.  `packed` is this rank's [scenarios, 2 n + 2] result block
    (AcPowerFlow.pack_results_device: V | theta | iterations | status); every rank contributes its contiguous block of
    scenarios and receives the global block in scenario order.  `dist` is an initialised torch.distributed module (backend
    "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests); the tensor must live on the backend's device.
    Returns (iterations, status, magnitude, angle) views of the gathered block."""
    import torch
    world = dist.get_world_size()
    packed = packed.contiguous()
    g = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(g, packed)
    return unpackResults(g)


def unpackResults(g):
    """(iterations, status, magnitude, angle) of a [scenarios, 2 n + 2] result block."""
    n = (g.shape[1] - 2) // 2
    return g[:, 2 * n].long(), g[:, 2 * n + 1].long(), g[:, :n], g[:, n:2 * n]
