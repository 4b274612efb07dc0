#!/usr/bin/env python3
"""Round 5 probe: does the chain-bound TOP of one batch's factorisation overlap with the bandwidth-bound BOTTOM levels of another's, and does a stream priority change it?
Handle A launches the bottom levels only (JG_PROBE_FACT_PART=1), handle B the top only (=2); each is timed alone, then both at once on a thread each (wall against the sum).
python tools/r05_overlap_probe.py [batch] [case]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
case = sys.argv[2] if len(sys.argv) > 2 else "case_ACTIVSg10k"
s = jg.powerSystem(case)


def handle(part, prio=None, shared=False):
    os.environ["JG_PROBE_FACT_PART"] = str(part)
    if prio is not None:
        os.environ["JG_STREAM_PRIORITY"] = str(prio)
    an = jg.contingencyAnalysis(s, jg.outageList(s, batch, seed=512))
    os.environ.pop("JG_PROBE_FACT_PART"); os.environ.pop("JG_STREAM_PRIORITY", None)
    if shared:
        jg._lib.check(jg._lib.lib().jg_nr_set_shared(an._h, 1))
    for _ in range(2):
        an.time_kernel(1, 5)
    return an


def together(hs, reps):
    """every handle repeats its part reps[i] times on its own thread; wall of all"""
    bar = threading.Barrier(len(hs) + 1)
    ths = [threading.Thread(target=lambda h=h, r=r: (bar.wait(), h.time_kernel(1, r))) for h, r in zip(hs, reps)]
    for t in ths:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    return (time.perf_counter() - t0) * 1e3


for label, pa, pb in (("no priorities", None, None), ("top stream urgent (-1)", None, -1), ("bottom stream relaxed (1)", 1, None), ("both", 1, -1)):
    for shared in (False, True):
        A, B = handle(1, pa, shared), handle(2, pb, shared)
        ta, tb = min(A.time_kernel(1, 20) for _ in range(3)), min(B.time_kernel(1, 20) for _ in range(3))
        ra, rb = 60, max(1, round(60 * ta / tb))                       # about the same time each
        alone = ta * ra + tb * rb
        both = min(together([A, B], [ra, rb]) for _ in range(3))
        print(f"{label:28s} shared={int(shared)}: bottom {ta:.3f} ms x {ra}, top {tb:.3f} ms x {rb}: one after the other {alone:.1f} ms, at once {both:.1f} ms = {both / alone:.3f}", flush=True)
        A.close(); B.close()
F = [handle(0), handle(0)]
t1 = min(F[0].time_kernel(1, 20) for _ in range(3))
both = min(together(F, [40, 40]) for _ in range(3))
print(f"two whole factorisations: alone {t1:.3f} ms x 80 = {t1 * 80:.1f} ms, at once {both:.1f} ms = {both / (t1 * 80):.3f}")
