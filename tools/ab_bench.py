#!/usr/bin/env python
"""Developer aid: A/B comparison of environment-variable configurations of bench.py on ONE box, interleaved and repeated (single
runs differ by +-4 % between boxes and by a few % with the order they run in).

    python tools/ab_bench.py [--reps 5] [--args "--batch-users 256"] NAME=VAR1=x,VAR2=y NAME2= ...

Prints the median / min / max ms_per_step per configuration.
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--args", default="")
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    cfgs = []
    for c in a.configs:
        name, _, rest = c.partition("=")
        env = dict(kv.split("=", 1) for kv in rest.split(",") if kv)
        cfgs.append((name, env))
    res = {name: [] for name, _ in cfgs}
    for r in range(a.reps):
        for name, env in cfgs:
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + a.args.split(), env=e,
                                 capture_output=True, text=True)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[name].append(d["ms_per_step"])
    for name, _ in cfgs:
        v = np.array(res[name])
        print(f"{name:24s} median {np.median(v):.5f} ms  min {v.min():.5f}  max {v.max():.5f}  ({len(v)} runs)  -> {256 / np.median(v):.0f}K users/s at 256" if "--batch-users" not in a.args else
              f"{name:24s} median {np.median(v):.5f} ms  min {v.min():.5f}  max {v.max():.5f}  ({len(v)} runs)")


if __name__ == "__main__":
    main()
