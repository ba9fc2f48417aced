#!/usr/bin/env python
"""Developer aid: timeline of one item row inside decode_rows_kernel (cycle stamps at row start, loop start, around
every 64-example chunk's G store, row end).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DCDAE_DECODE_TIMING \\
          cdae_amd/csrc/cdae_hip.hip -o /tmp/libcdae_hip_timing.so
    CDAE_DEBUG_RANK=<popularity rank> python tools/decode_timeline.py /tmp/libcdae_hip_timing.so [batch_users]

Round-1 history (profiles/r01_decode_timeline.txt, then profiles/r01_decode_bisect.txt): the first timelines (2000-5000
cycles per example on every row while the launch ran) were read as chip-wide VALU-issue pressure; stripping the loop body
showed the time was in the duplicate-negative path instead (an fp32 atomic + s_waitcnt vmcnt(0) stalled a wavefront ~25 us,
and almost every row has a duplicate per batch).  With that fixed a popular row walks ~500 cycles per example and the
others ~1400 cycles per step of four rows.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdae_amd import binding, synth  # noqa: E402
import cdae_amd  # noqa: E402

binding.load_library(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
d = synth.generate_shape("ml10m")
m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=B))
m.set_interactions(d.num_users, d.num_items, d.train_ptr, d.train_col)
m.init_params(1)
for i in range(4):
    m.train_users(1, 0, i * B, (i + 1) * B)      # the timing build prints one "[decode timing]" line per call
