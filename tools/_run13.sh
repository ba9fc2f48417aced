mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py tests/test_gpu_accuracy.py tests/test_gpu_edge.py -m gpu -q 2>&1 | tail -n 30 > gpurun_out/r02/gputest13.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r02/gputest13.log | tail -n 8
python bench.py --no-cpu-baseline --full-output --batch-users 2048 --steps 60 --warmup 10 2>/dev/null | tail -n 1 > gpurun_out/r02/bench13_full_ml10m.json
python bench.py --no-cpu-baseline --full-output --shape yelp --num-dim 50 --batch-users 512 --steps 60 --warmup 10 2>/dev/null | tail -n 1 > gpurun_out/r02/bench13_full_yelp.json
python bench.py --no-cpu-baseline --full-output --shape cfg5_items --num-dim 512 --batch-users 1024 --steps 10 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r02/bench13_full_cfg5.json
python - <<'PY'
import json
for f in ('full_ml10m','full_yelp','full_cfg5'):
    d=json.load(open(f'gpurun_out/r02/bench13_{f}.json'))
    print(f, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['achieved'],1), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})
PY
bash tools/profile_round.sh r02
python bench.py > gpurun_out/r02/bench13_default_full.json 2>/dev/null
tail -c 1500 gpurun_out/r02/bench13_default_full.json
