"""Sharded Monte-Carlo state estimation: noisy realisations of ONE measurement set as the lanes of Gauss-Newton batches.

The reference draws a realisation inside `add<Meter>!(...; noise = true)` (/root/reference/src/measurement/utility.jl:70-73:
`mean + variance^(1/2) * randn`) and estimates one at a time (`stateEstimation!`, src/stateEstimation/acStateEstimation.jl:1286-1329);
a Monte-Carlo study is a user-level loop over both.  Realisations share rows, type codes, Jacobian and gain pattern and the symbolic
analysis: here `batch` of them advance together in one handle, several handles keep the GPU busy, the realisations of a study shard across
GPUs contiguously (contingency.shard) and ONE collective carries the result record of every batch (for the estimator what
ContingencyPipeline is for the power flow).
"""
from __future__ import annotations

import threading

import numpy as np

from . import _lib
from .measurement import Measurement
from .stateestimation import AcStateEstimation, drawNoise_, gaussNewton, setNoise_, stateEstimation_

RECORD_TAIL = 3         # iterations | status | objective behind magnitude[n] | angle[n]


class MonteCarloPipeline:
    """`inflight` batches of `batch` realisations each in flight on one GPU: a handle, a HIP stream and a host thread per batch (the narrow
    launches of one batch's factorisation are filled by the others').

    A job is one batch of realisations: an integer SEED or a pair (seed, first realisation) -- the handle draws realisations first .. first + batch - 1 of
    that seed ON THE DEVICE (stateestimation.drawNoise_: counter-based generator + the acWLS value rules, nothing over PCIe; a rank of a sharded study passes
    its own `first`) --, `host_noise=True` draws them with numpy's PCG64(seed) on the host instead (setNoise_: [batch, m] arrays over PCIe), or None (the
    handle keeps the realisations it holds: a timing loop).  Every job restarts from the pipeline's start point, which stays in HBM.  `on_done(job, analysis)` runs on the CALLER's thread in job order -- where a sharded
    run issues its gather (collectives must be issued in the same order on every rank)."""

    def __init__(self, monitoring: Measurement, batch: int, inflight: int = 2, device: int = 0, start=None, method=None, scale: float = 1.0, host_noise: bool = False):
        self.monitoring, self.batch, self.scale, self.host_noise = monitoring, int(batch), float(scale), bool(host_noise)
        kw = {} if method is None else {"method": method}
        self.handles = [gaussNewton(monitoring, batch=self.batch, device=device, **kw) for _ in range(max(1, int(inflight)))]
        n = monitoring.system.bus.number
        vm, va = (np.ones(n), np.zeros(n)) if start is None else start          # flat start (BASELINE config 4) unless told otherwise
        for h in self.handles:
            h.setVoltage(vm, va)
            h.snapshot_voltage()

    @property
    def record_width(self) -> int:
        return 2 * self.monitoring.system.bus.number + RECORD_TAIL

    def close(self):
        for h in self.handles:
            h.close()
        self.handles = []

    def run(self, jobs, iteration: int = 40, tolerance: float = 1e-8, on_done=None, record=None, records: int = 0):
        """jobs: sequence of seeds / None.  record: optional callable job -> DEVICE pointer of a [batch, 2 n + 3] float64 buffer, complete when
        on_done(job, .) is called; the caller owns a ring of `records` buffers (job j + records is not written before on_done(j) has returned).
        Returns per-job (iterations, status) arrays."""
        jobs = list(jobs)
        nj, nh = len(jobs), len(self.handles)
        results = [None] * nj
        done = [threading.Event() for _ in jobs]
        delivered = [threading.Event() for _ in jobs]
        ring = int(records) if (record is not None and records) else 0
        errors = []

        def worker(k):
            try:
                for j in range(k, nj, nh):
                    if j - nh >= 0 and on_done is not None and record is None:
                        delivered[j - nh].wait()                  # without a record the results live in the handle until the caller has seen them
                    if ring and j - ring >= 0:
                        delivered[j - ring].wait()
                    if errors:
                        return
                    h = self.handles[k]
                    if jobs[j] is not None:
                        seed, first = (jobs[j] if isinstance(jobs[j], (tuple, list)) else (jobs[j], 0))
                        if self.host_noise:
                            setNoise_(h, np.random.Generator(np.random.PCG64(int(seed))), scale=self.scale)
                        else:
                            drawNoise_(h, int(seed), scale=self.scale, first=int(first))
                    h.restore_voltage()
                    stateEstimation_(h, iteration=iteration, tolerance=tolerance, fetch=False)
                    results[j] = (np.array(h.method.iteration), np.array(h.status))
                    if record is not None:
                        h.pack_results_device(record(j))
                    done[j].set()
            except BaseException as e:                             # surface in the caller, never hang it
                errors.append(e)
                for ev in done + delivered:
                    ev.set()

        threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(nh)]
        for t in threads:
            t.start()
        try:
            for j in range(nj):
                done[j].wait()
                if errors:
                    break
                if on_done is not None:
                    on_done(j, self.handles[j % nh])
                delivered[j].set()
        except BaseException as e:                                 # the caller's own on_done failed: the workers must not wait for deliveries that never come
            errors.insert(0, e)
        finally:
            if errors:
                for ev in delivered:
                    ev.set()
            for t in threads:
                t.join()
        if errors:
            raise errors[0]
        return results


def gatherEstimates(dist, packed):
    """Final gather of a sharded Monte-Carlo batch: ONE collective.  `packed` is this rank's [realisations, 2 n + 3] record
    (AcStateEstimation.pack_results_device); returns (iterations, status, objective, magnitude, angle) of the global block in realisation order."""
    import torch
    world = dist.get_world_size()
    packed = packed.contiguous()
    g = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(g, packed)
    return unpackEstimates(g)


def gatherEstimatesDevice(an: AcStateEstimation, comm, out_ptr: int):
    """The same gather through the C ABI alone (jg_gn_allgather_results: pack + ncclAllGather of RCCL on the handle's stream); `out_ptr`: device
    memory for [world x batch][2 n + 3] doubles."""
    _lib.check(_lib.lib().jg_gn_allgather_results(an._h, comm.h, _lib.VP(int(out_ptr))))


def unpackEstimates(g):
    """(iterations, status, objective, magnitude, angle) of a [realisations, 2 n + 3] record."""
    n = (g.shape[1] - RECORD_TAIL) // 2
    return g[:, 2 * n].long(), g[:, 2 * n + 1].long(), g[:, 2 * n + 2], g[:, :n], g[:, n:2 * n]
