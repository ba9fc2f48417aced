mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_multi.py tests/test_bench_contract.py -m gpu -q 2>&1 | tail -n 40 > gpurun_out/r02/gputest8.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r02/gputest8.log | tail -n 12
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline $EXTRA 2>/dev/null | tail -n 1 > gpurun_out/r02/bench8_$name.json; }
EXTRA="--exchange-every 2" run rccl_p2 CDAE_BENCH_FORCE_DIST=1
EXTRA="--exchange-every 8" run rccl_p8 CDAE_BENCH_FORCE_DIST=1
EXTRA="--exchange-every 1000000" run rccl_never CDAE_BENCH_FORCE_DIST=1
EXTRA="--exchange-every 2" run nocomm_p2 CDAE_BENCH_FORCE_DIST=1 CDAE_BENCH_NO_COMM=1
EXTRA="--exchange-every 0" run nocomm_sync CDAE_BENCH_FORCE_DIST=1 CDAE_BENCH_NO_COMM=1
EXTRA="--exchange-every 2" run rccl_p2_q16 CDAE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=16
EXTRA="--exchange-every 2" run rccl_p2_q4 CDAE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/bench8_*.json')):
    d=json.load(open(f))
    print(f.split('bench8_')[1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})
PY
