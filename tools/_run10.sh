mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_multi.py tests/test_bench_contract.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -n 40 > gpurun_out/r02/gputest10.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r02/gputest10.log | tail -n 12
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline $EXTRA 2>/dev/null | tail -n 1 > gpurun_out/r02/bench10_$name.json; }
EXTRA="" run auto CDAE_BENCH_FORCE_DIST=1
EXTRA="--exchange-every 2" run p2 CDAE_BENCH_FORCE_DIST=1
EXTRA="--exchange-every 4" run p4 CDAE_BENCH_FORCE_DIST=1
EXTRA="--exchange-every 0" run sync CDAE_BENCH_FORCE_DIST=1
EXTRA="" run none
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/bench10_*.json')):
    d=json.load(open(f))
    print(f.split('bench10_')[1], round(d['value']), round(d['ms_per_step'],4), d['config']['exchange'][:150])
PY
