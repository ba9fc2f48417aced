mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_integer.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -n 15 > gpurun_out/r02/gputest6.log
tail -n 3 gpurun_out/r02/gputest6.log | head -n 2
for up in 64; do for b in 256 512; do
  CDAE_UNIT_POS=$up python bench.py --no-cpu-baseline --batch-users $b 2>/dev/null | tail -n 1 > gpurun_out/r02/bench6_b${b}_u${up}.json
  CDAE_SORT_ROCPRIM=1 CDAE_UNIT_POS=$up python bench.py --no-cpu-baseline --batch-users $b 2>/dev/null | tail -n 1 > gpurun_out/r02/bench6_b${b}_u${up}_rocprim.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/bench6_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})
PY
