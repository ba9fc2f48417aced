mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_multi.py tests/test_bench_contract.py tests/test_host_cpp.py -m gpu -q 2>&1 | tail -n 30 > gpurun_out/r02/gputest16.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r02/gputest16.log | tail -n 12
for s in 1 2 8; do
python bench.py --full-output --shape cfg5_items --num-dim 512 --batch-users 1024 --steps 6 --warmup 2 --layout item-rows --logical-shards $s --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/r02/bench16_cfg5_itemrows_s$s.json
done
python bench.py --full-output --batch-users 2048 --steps 40 --warmup 5 --layout item-rows --logical-shards 4 --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/r02/bench16_ml10m_itemrows_s4.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/bench16_*.json')):
    try:
        d=json.load(open(f)); print(f.split('bench16_')[1], round(d['value']), round(d['ms_per_step'],3), d['config']['parallelism'], round(d['roofline']['achieved'],1))
    except Exception as e: print(f, 'ERR', e)
PY
