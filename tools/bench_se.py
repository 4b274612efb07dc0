#!/usr/bin/env python3
"""BASELINE config 4 on its own: Gauss-Newton WLS state estimation (PMU + legacy) on the 9241-bus PEGASE-shaped grid, 1 GPU
(the same measurement bench.py reports as `config4_se`).

  python tools/bench_se.py [--batch 512] [--steps 12] [--inflight 2] [--case case9241synth] [--no-cpu]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import se_config4      # noqa: E402  (the oracle is only ever touched from bench.py's CPU legs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--case", default="case9241synth")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight (own handle, stream and host thread each)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import juliagrid.jl_amd as jg
    print(json.dumps(se_config4(jg, args.case, args.batch, args.steps, args.warmup, args.inflight, cpu=not args.no_cpu, cpu_budget_s=15.0)))


if __name__ == "__main__":
    main()
