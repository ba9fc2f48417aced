mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_multi.py tests/test_bench_contract.py -m gpu -q -x 2>&1 | tail -n 40 > gpurun_out/r02/gputest7.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r02/gputest7.log | tail -n 25
python tools/accuracy_envelope.py --shards 8 --batch-users 64 --period 0 --seeds 20141119 2>&1 | tail -n 2
python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/r02/bench7_default.json
CDAE_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/r02/bench7_onerank.json
python - <<'PY'
import json
for f in ('bench7_default','bench7_onerank'):
    d=json.load(open(f'gpurun_out/r02/{f}.json'))
    print(f, round(d['value']), round(d['ms_per_step'],4), d['config']['exchange'][:80], json.dumps(d['roofline'])[:600])
PY
