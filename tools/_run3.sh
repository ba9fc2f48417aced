mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -n 5
python bench.py --no-cpu-baseline --batch-users 256 2>/dev/null | tail -n 1 > gpurun_out/r02/bench_b256_spec.json
python bench.py --no-cpu-baseline --batch-users 512 2>/dev/null | tail -n 1 > gpurun_out/r02/bench_b512_spec.json
python - <<'PY'
import json
for b in (256,512):
    d=json.load(open(f'gpurun_out/r02/bench_b{b}_spec.json'))
    print(b, round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])
PY
