#!/usr/bin/env python
"""Developer aid: wavefront timeline of one training batch (CDAE_WAVE_TRACE build-in, cdae_kernels.hpp trace_begin/trace_end).

    python tools/wave_trace.py [batch_users] [batches]      # writes /tmp/cdae_wave_trace.bin, prints a per-role summary

Every traced wavefront records {role, id, start, end} in 100 MHz device time (10 ns ticks).  Per role the summary gives the
number of wavefronts, when the first / median / last one STARTED and ENDED relative to the first record of the batch, and the
distribution of wavefront lifetimes — enough to tell a launch bound by dispatch (late starts), by a few long wavefronts (chains)
or by throughput (everything long).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = os.environ.setdefault("CDAE_WAVE_TRACE", "/tmp/cdae_wave_trace.bin")
import cdae_amd  # noqa: E402
from cdae_amd import synth  # noqa: E402

ROLES = {1: "encode_partial", 2: "encode_finish", 3: "decode hot row", 4: "decode 4 rows", 5: "hidden_gather", 6: "hidden_finish",
         7: "input: bias b", 8: "input: hot row", 9: "input: row", 10: "decode: blocker"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    d = synth.generate_shape("ml10m")
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=B))
    m.set_interactions(d.num_users, d.num_items, d.train_ptr, d.train_col)
    m.init_params(1)
    m.train_users(1, 0, 0, N * B)
    m.close() if hasattr(m, "close") else None
    del m
    import gc
    gc.collect()
    rec = np.fromfile(PATH, dtype=np.uint64).reshape(-1, 4)
    rec = rec[rec[:, 1] != 0]
    if os.environ.get("CDAE_WAVE_TRACE_DUMP"):
        np.save(os.environ["CDAE_WAVE_TRACE_DUMP"], rec)
    # slots keep the last batch that wrote them: drop stragglers of earlier batches (ids the last batch did not reach)
    last = rec[:, 2].astype(np.int64).max()
    rec = rec[last - rec[:, 1].astype(np.int64) < 26_000]            # the last two batches (260 us)
    tag = (rec[:, 0] >> np.uint64(32)).astype(np.int64)
    role, odd = tag % 16, tag // 16
    t0 = rec[:, 1].astype(np.int64)
    t1 = rec[:, 2].astype(np.int64)
    base = t0.min()
    us = lambda x: 0.01 * x          # 100 MHz ticks -> us
    print(f"batch_users {B}: {len(rec)} wavefront records of the last two batches, span {us(t1.max() - base):.1f} us")
    print(f"{'batch role':24s} {'waves':>6s} | start first/median/last (us) | end first/median/last (us) | lifetime min/median/p90/max (us)")
    order = sorted(set(zip(odd.tolist(), role.tolist())), key=lambda pr: t0[(odd == pr[0]) & (role == pr[1])].min())
    prev_end = None
    for o, r in order:
        k = (odd == o) & (role == r)
        s, e, life = t0[k] - base, t1[k] - base, t1[k] - t0[k]
        gap = "" if prev_end is None else f"  (+{us(s.min() - prev_end):.1f} after the previous role's last end)"
        print(f"{'odd ' if o else 'even'} {ROLES.get(r, str(r)):18s} {k.sum():6d} | {us(s.min()):7.1f} {us(np.median(s)):7.1f} {us(s.max()):7.1f}     | "
              f"{us(e.min()):7.1f} {us(np.median(e)):7.1f} {us(e.max()):7.1f}   | "
              f"{us(life.min()):6.1f} {us(np.median(life)):6.1f} {us(np.percentile(life, 90)):6.1f} {us(life.max()):6.1f}{gap}")
        prev_end = e.max() if r not in (7,) else prev_end
        if r == 9:
            prev_end = max(e.max(), (t1[(odd == o) & (role == 7)] - base).max()) if ((odd == o) & (role == 7)).any() else e.max()
    # the ten longest-lived and ten last-finishing wavefronts of every role (id = the kernel's own numbering)
    for r in sorted(set(role.tolist())):
        k = np.where((role == r) & (odd == odd[np.argmax(t1)]))[0]
        life = t1[k] - t0[k]
        top = k[np.argsort(-life)[:6]]
        print(f"  {ROLES.get(r, str(r))}: longest " + ", ".join(f"#{int(rec[i, 0] & np.uint64(0xffffffff))}:{us(t0[i]-base):.1f}->{us(t1[i]-base):.1f}" for i in top))


if __name__ == "__main__":
    main()
