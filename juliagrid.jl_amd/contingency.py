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

from . import _lib
from .powerflow import AcPowerFlow, BaseCase, newtonRaphson, powerFlow_, setOutage_, setOutages_, startFromBase_, setFirstIteration_, _push_voltage
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


def recommendedLanes(buses: int) -> int:
    """Scenarios per device batch by grid size: what keeps a launch around 5 million bus-lanes.  The 10 000-bus grid fills the chip at 512 lanes; a smaller grid at
    512 lanes is launch-bound (case1354pegase, 50 launches per iteration over 1 354 buses: factorisation at 0.16 of the HBM roofline) and gains from wider batches
    until the state arrays leave the Infinity Cache -- measured on one MI355X (profiles/r06_bench_1354.txt): 512 / 1 024 / 2 048 / 4 096 lanes = 1.61 / 1.82 / 2.47 /
    2.80 million NR it/s, factorisation 0.156 / 0.241 / 0.289 / 0.322 of peak.  512 lanes x floor(10 000 / buses), at least 512, at most 4 096."""
    return 512 * max(1, min(8, 10000 // max(1, int(buses))))


def deviceBatching(share: int, steps: int, lanes: int = 512) -> int:
    """How many consecutive steps a rank solves together as one device batch (strong scaling: a rank's share of a step shrinks
    with the number of ranks, the path is at its best around `lanes` scenarios per launch -- recommendedLanes(buses)).  A rank whose whole run is at most
    1.25 x lanes scenarios solves it as ONE batch, at most 2.5 x lanes as TWO that are both in flight (batches of up to 640 / 1 280 lanes at the default 512);
    otherwise at most lanes // share steps: among the upper half of that range the count that leaves the fewest spare lanes in the last batch of a run of
    `steps` steps, the larger one on a tie.  1 when the share already fills the lanes."""
    share, steps = max(1, int(share)), max(1, int(steps))
    if share < lanes:
        # (round 5) a rank whose WHOLE run is a batch or two -- N = 8 at the driver's K = 20: 20 x 64 = 1 280 scenarios -- solves it as ONE batch of up to 5/4 lanes resp.
        # as TWO batches that are both in flight: measured on one GPU (profiles/r05_n8_shape.txt, 64 scenarios per step, K = 20) 2 x 640 lanes 260k NR it/s, 1 x 1 280
        # 247k, 5 x 256 242k, 512 + 512 + 256 230k, 4 x 320 (the old choice: no spare lanes) 197k
        total, wide = share * steps, lanes + lanes // 4
        if lanes < total <= wide:
            return steps
        if wide < total <= 2 * wide:
            return -(-steps // 2)
    m_max = min(max(1, lanes // share), steps)
    if m_max <= 1:
        return 1
    return min(range(max(1, (m_max + 1) // 2), m_max + 1), key=lambda m: (-(-steps // m) * m - steps, -m))


def contingencyAnalysis(system: PowerSystem, labels, device: int = 0, method: str = "nr") -> AcPowerFlow:
    """Batched analysis with scenario s = outage of branch labels[s] (None / 0 = base case).  method: "nr" Newton-Raphson, "bx" / "xb" fast
    Newton-Raphson (constant matrices with per-scenario edits, ONE factorisation for the batch: jg_nr_fast_patch_batch)."""
    labels = list(labels)
    if method in ("bx", "xb"):
        from .powerflow import fastNewtonRaphsonBX, fastNewtonRaphsonXB
        an = (fastNewtonRaphsonBX if method == "bx" else fastNewtonRaphsonXB)(system, batch=len(labels), device=device, max_patch=4)
    elif method == "nr":
        an = newtonRaphson(system, batch=len(labels), device=device, max_patch=4)
    else:
        raise ValueError("method: nr | bx | xb")
    setOutages_(an, [int(lab) if lab else 0 for lab in labels])
    return an


class _Pool:
    """A handle that collects the stragglers of several batches (ContingencyPipeline, straggler hand-off)."""

    def __init__(self, handle):
        self.handle = handle
        self.fill = 0
        self.routes = []                    # (job, scenario numbers in the job's batch, first lane in the pool)
        self.idle = threading.Event()
        self.idle.set()
        self.queued = False


class ContingencyPipeline:
    """`inflight` batches of the same grid in flight on one GPU, each on its own handle, HIP stream and host thread.

    Why: the sparse LU replays dozens of dependency levels per iteration and most of them occupy a fraction of the chip for
    a few microseconds (latency bound).  A second and third batch fill those holes: kernels of different streams run
    concurrently.  Measured on MI355X, case_ACTIVSg10k, 512 scenarios per batch: 131k NR iterations/s with 1 batch in flight,
    215k with 3.

    Straggler hand-off (`pool` > 0): a batch advances in lockstep, so its last iterations run on the few scenarios that have
    not converged yet (36 of 512 N-1 scenarios of case_ACTIVSg10k need a 4th iteration, one a 5th) at the latency of a full
    pass.  With a pool, a batch stops as soon as at most `defer_at` (<= 64) scenarios are active; those move -- state,
    injections, outage patch, iteration count -- into a pool handle of `pool` lanes that collects the stragglers of several
    batches and finishes them together, while the batch's handle starts its next job.  Lanes never interact, so every
    scenario's result is bitwise what an undisturbed batch gives.

    jobs are processed in order; `on_done(job, analysis)` (optional) is called on the CALLER's thread in job order (this is
    where a sharded run issues its RCCL gather: collectives must be issued in the same order on every rank).  Without a pool
    the batch's results are still resident in `analysis` at that point.  With a pool the handle may already be running its next
    job: results are delivered through `record` (see run)."""

    def __init__(self, system: PowerSystem, batch: int, inflight: int = 3, device: int = 0, start=None, pool: int = 0, defer_at: int = 64,
                 shared_first: bool = True, top_cap: int = 0):
        """shared_first (with a `start`): the start is a BASE CASE -- its Jacobian is factorised once (BaseCase) and the first Newton iteration of every
        scenario is a sweep pair on that shared factor plus a 4 x 4 correction instead of a batched refactorisation (jgrid.h: jg_nr_base_*; the reference
        refactorises per scenario, branch.jl:453-459 + acPowerFlow.jl:890-897).  The handles decide per run whether the conditions hold."""
        self.system, self.batch = system, int(batch)
        self.base = None
        self._shared_first, self._top_cap, self._device = bool(shared_first), int(top_cap), int(device)
        self.handles = [newtonRaphson(system, batch=self.batch, device=device, max_patch=4) for _ in range(max(1, int(inflight)))]
        self.defer_at = max(0, min(64, int(defer_at)))
        self.pools = []
        if pool and self.batch > 64 and self.defer_at > 0:
            lanes = max(2 * self.defer_at, -(-int(pool) // 64) * 64)
            # the engine picks its plan (where the multifrontal top starts, front caps: another summation order) by the side of 256 lanes
            # a handle is on (Engine::create): the pool must sit on the same side as the batches it serves, or a straggler that finishes
            # there would no longer be bitwise the scenario of a lockstep batch
            # (there are more classes than two -- up to 32 scenarios, one lane group, 65-255, 256 and more -- and a pool of 64 lanes
            # beside batches of 128 / 192 would run the one-lane-group plan; jg_nr_move_lanes now refuses a hand-off between different plans)
            lanes = max(lanes, 256) if -(-self.batch // 64) * 64 >= 256 else min(max(lanes, 128), 192)
            self.pools = [_Pool(newtonRaphson(system, batch=lanes, device=device, max_patch=4)) for _ in range(2)]
        if len(self.handles) + len(self.pools) > 1:      # several batches share the GPU: the top launches leave room for the others' workgroups
            for an in self.handles + [p.handle for p in self.pools]:
                _lib.check(_lib.lib().jg_nr_set_shared(an._h, 1))
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
        for p in self.pools:                             # idle pool lanes compute along in their lane group: give them a sane state
            _push_voltage(p.handle, magnitude, angle)
        if self.base is not None:
            self.base.close()
            self.base = None
        if self._shared_first and np.ndim(magnitude) == 1:
            single = newtonRaphson(self.system, batch=1, device=self._device)     # the system's own injections and nodal matrix, the common start
            _push_voltage(single, magnitude, angle)
            self.base = BaseCase(single, top_cap=self._top_cap)
            single.close()
            for an in self.handles:
                self.base.attach(an)

    def setFirstIteration(self, shared: bool = True):
        """A/B switch: shared=False makes every handle refactorise in its first iteration as well (the base stays attached)."""
        for an in self.handles:
            setFirstIteration_(an, shared)

    def close(self):
        for an in self.handles:
            an.close()
        for p in self.pools:
            p.handle.close()
        if self.base is not None:
            self.base.close()
            self.base = None
        self.handles, self.pools = [], []

    def setRating(self, rating):
        """Branch ratings (pu of apparent power, 0 = no limit) of the screen summaries (`run(..., summary=True)`), on every handle."""
        from .powerflow import _upload_branches
        r = None if rating is None else np.ascontiguousarray(np.asarray(rating, dtype=np.float64))
        if r is not None and r.shape != (self.system.branch.number,):
            raise ValueError("rating: one value per branch")
        for an in self.handles + [p.handle for p in self.pools]:
            if not an._branches_on_device:
                _upload_branches(an)
            _lib.check(_lib.lib().jg_nr_set_screen(an._h, None if r is None else r.ctypes.data))
            an._screen_rating = r

    def run(self, jobs, iteration: int = 20, tolerance: float = 1e-8, on_done=None, fetch: bool = False, record=None, records: int = 0,
            summary: bool = False):
        """jobs: sequence of label lists (one batch each; None = keep the handle's current outages), or dicts {"labels": [...], "active": P, "reactive": Q}
        for Monte-Carlo jobs: net injections supply - demand per scenario ([batch, n], or [n] for all) -- the independent instances the north star names beside
        contingencies; a pool receives a straggler's injections with its state (jg_nr_move_lanes).
        record: optional callable job -> DEVICE pointer of a [batch, 2 n + 2] float64 buffer; the job's result record
        (V | theta | iterations | status per scenario) is complete in it when on_done(job, .) is called.  The caller owns a
        ring of `records` such buffers (record(j) and record(j + records) may be the same memory): job j + records is not
        written before on_done(j) has returned.  Returns per-job (iterations, status) arrays.
        summary: the record is the SCREEN SUMMARY instead ([batch, 10] float64 per job: powerflow.screenSummary_ -- worst branch loading against
        setRating's limits, largest flow, voltage extremes, iterations, status), reduced on the device by the handle that finished the scenario:
        what a sharded screen gathers is 80 bytes per scenario, not 16 n + 16."""
        jobs = list(jobs)
        nj = len(jobs)
        results = [None] * nj
        main_done = [threading.Event() for _ in jobs]
        pool_done = [None] * nj                                   # Event of the jobs that handed scenarios to a pool
        released = [threading.Event() for _ in jobs]              # the handle of job j may start job j + nh
        delivered = [threading.Event() for _ in jobs]             # on_done(j) has returned
        nh = len(self.handles)
        use_pool = bool(self.pools)
        if use_pool and on_done is not None and record is None:
            raise ValueError("a pipeline with a straggler pool delivers results through `record`")
        ring = int(records) if (record is not None and records) else 0
        errors = []
        lock = threading.Lock()
        state = {"fill": 0}                                       # index of the pool that is being filled
        import queue
        flush_q = queue.Queue()

        def submit(p):                                            # under `lock`
            if p.fill > 0 and not p.queued:
                p.queued = True
                p.idle.clear()
                flush_q.put(p)

        def flush_filling():
            with lock:
                submit(self.pools[state["fill"]])

        def pool_worker():
            try:
                while True:
                    p = flush_q.get()
                    if p is None:
                        return
                    it, st = p.handle.resume(p.fill, iteration, tolerance)
                    for j, home, off in p.routes:
                        main_done[j].wait()                       # the batch's own record / arrays are written first
                        if errors:
                            break
                        results[j][0][home] = it[off:off + home.size]
                        results[j][1][home] = st[off:off + home.size]
                        if record is not None:
                            if summary:
                                p.handle.screen_rows_device(record(j), off, home)
                            else:
                                p.handle.pack_rows_device(record(j), off, home)
                        pool_done[j].set()
                    p.routes = []                                 # no lock: a worker may hold it while it waits for this pool; submit() sees
                    p.fill = 0                                    # either queued (skips) or an empty pool (skips)
                    p.queued = False
                    p.idle.set()
            except BaseException as e:
                errors.append(e)
                for ev in main_done + delivered + [x for x in pool_done if x is not None]:
                    ev.set()
                for p in self.pools:
                    p.idle.set()

        def worker(k):
            try:
                for j in range(k, nj, nh):
                    if j - nh >= 0:
                        released[j - nh].wait()
                    if ring and j - ring >= 0 and not delivered[j - ring].is_set():
                        if use_pool:
                            flush_filling()                       # what the caller is waiting for may sit in the pool that is filling
                        delivered[j - ring].wait()
                    if errors:
                        return
                    an = self.handles[k]
                    job = jobs[j]
                    if isinstance(job, dict):                     # a Monte-Carlo job: per-scenario injections (load / generation variations), outages optional
                        if job.get("labels") is not None:
                            labels = [int(x) if x else 0 for x in job["labels"]]
                            setOutages_(an, labels + [0] * (self.batch - len(labels)))
                        if job.get("active") is not None or job.get("reactive") is not None:
                            from .powerflow import setInjection_
                            setInjection_(an, job.get("active"), job.get("reactive"))
                    elif job is not None:
                        labels = [int(x) if x else 0 for x in job]
                        setOutages_(an, labels + [0] * (self.batch - len(labels)))
                    if self.base is not None:
                        startFromBase_(an)                        # device-side broadcast of the base state; the run takes its first iteration on the shared factor when it can
                    else:
                        an.restore_voltage()
                    # the LAST job of a handle finishes its own stragglers in lockstep: the hand-off pays when it frees the handle for its next job; at the end of a
                    # run the pools' turns (one worker resumes them one after the other) would only queue the tails of the batches that end together (round 6:
                    # a rank of the 8-GPU run at the driver's K = 20 -- two 640-lane batches, both the last of their handle)
                    if use_pool and j + nh < nj:
                        left = an.run_defer(iteration, tolerance, self.defer_at)
                        if left > 0:
                            with lock:
                                p = self.pools[state["fill"]]
                                if p.queued or p.fill + left > p.handle.batch:
                                    submit(p)
                                    state["fill"] ^= 1
                                    p = self.pools[state["fill"]]
                                p.idle.wait()                     # the other pool has long finished in a steady pipeline
                                if errors:
                                    return
                                home = p.handle.take_lanes(an, p.fill)
                                p.handle._outage_labels[p.fill:p.fill + home.size] = an._outage_labels[home]     # (the screen summary leaves the branch that is out aside)
                                pool_done[j] = threading.Event()
                                p.routes.append((j, home, p.fill))
                                p.fill += home.size
                        an.finish()
                        if fetch:
                            an._pull_voltage()
                    else:
                        powerFlow_(an, iteration=iteration, tolerance=tolerance, fetch=fetch)
                    results[j] = (np.array(an.method.iteration), np.array(an.status))
                    if record is not None:
                        if summary:
                            an.screen_device(record(j))
                        else:
                            an.pack_results_device(record(j))
                    main_done[j].set()
                    if use_pool or record is not None and on_done is None:
                        released[j].set()                         # nothing of this job lives in the handle any more
                if use_pool:
                    flush_filling()                               # this worker adds nothing more
            except BaseException as e:                             # surface in the caller, never hang it
                errors.append(e)
                for ev in main_done + delivered + released + [x for x in pool_done if x is not None]:
                    ev.set()
                for p in self.pools:
                    p.idle.set()

        threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(nh)]
        pthread = threading.Thread(target=pool_worker, daemon=True) if use_pool else None
        if pthread:
            pthread.start()
        for t in threads:
            t.start()
        try:
            for j in range(nj):
                main_done[j].wait()
                if not errors and pool_done[j] is not None:
                    if j == nj - 1 or all(main_done[i].is_set() for i in range(j, nj)):
                        flush_filling()                           # nobody is left to fill the pool
                    while not pool_done[j].wait(0.05):
                        if errors:
                            break
                        if all(ev.is_set() for ev in main_done):
                            flush_filling()
                if errors:
                    break
                if on_done is not None:
                    on_done(j, self.handles[j % nh])
                delivered[j].set()
                released[j].set()
        except BaseException as e:                                 # the caller's own on_done failed: the workers must not wait for deliveries that never come
            errors.insert(0, e)
        finally:
            if errors:
                for ev in released + delivered + [x for x in pool_done if x is not None]:
                    ev.set()
                for p in self.pools:
                    p.idle.set()
            for t in threads:
                t.join()
            if pthread:
                flush_q.put(None)
                pthread.join()
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


def gatherResultsDevice(an: AcPowerFlow, comm, out_ptr: int):
    """The same gather through the C ABI alone (jg_nr_allgather_results: pack + ncclAllGather of RCCL on the handle's stream): `comm` is a
    juliagrid.jl_amd._lib.Comm, `out_ptr` a device pointer to [world x batch][2 n + 2] doubles owned by the caller.  Every rank calls
    it with the same batch; on return the record of the whole screen is in scenario order on every rank."""
    from . import _lib
    _lib.check(_lib.lib().jg_nr_allgather_results(an._h, comm.h, _lib.VP(int(out_ptr))))


def unpackResults(g):
    """(iterations, status, magnitude, angle) of a [scenarios, 2 n + 2] result block."""
    n = (g.shape[1] - 2) // 2
    return g[:, 2 * n].long(), g[:, 2 * n + 1].long(), g[:, :n], g[:, n:2 * n]
