#!/bin/bash
# developer measurement (profiles/rNN_exchange_step_components.txt): the training step alone and with the synchronous exchange machinery of a
# ONE-RANK library-owned RCCL communicator (stage kernel + identity all-reduce + merge kernel + their ordering), by users per step and shape
export TMPDIR=/tmp
R=${1:-r06}
O=gpurun_out/exchange
mkdir -p $O
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
{
echo "# one MI355X, K=200, bench.py --no-cpu-baseline --batch-users B [--layout users --exchange-every 0 with a ONE-RANK library-owned RCCL communicator]"
echo "# plain: the training step alone; sync: + stage kernel + one-rank ncclAllReduce (identity) + merge kernel, every step (round 6: the collective on the main stream)"
echo "# shape     B   plain ms    sync ms  exchange machinery us   [sync ms with round 5's hand-off to the collective stream]"
for shape in ml10m netflix; do
  for B in 64 128 256; do
    p=$(python bench.py --no-cpu-baseline --shape $shape --batch-users $B --steps 300 --warmup 40 2>/dev/null | val)
    s=$(CDAE_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 python bench.py --no-cpu-baseline --shape $shape --batch-users $B --layout users --exchange-every 0 --steps 300 --warmup 40 2>/dev/null | val)
    o=""
    if [ "$B" = "64" ]; then o=$(CDAE_HIP_LIBRARY=$PWD/build/libcdae_hip_dev.so CDAE_XCHG_COLLECTIVE_STREAM=1 CDAE_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29542 python bench.py --no-cpu-baseline --shape $shape --batch-users $B --layout users --exchange-every 0 --steps 300 --warmup 40 2>/dev/null | val); fi
    python -c "print('%-8s %4d %10.4f %10.4f %22.1f   %s' % ('$shape', $B, $p, $s, 1e3*($s-$p), '$o'))"
  done
done
} | tee $O/${R}_exchange_step_components.txt
