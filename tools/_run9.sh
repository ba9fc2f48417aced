mkdir -p gpurun_out/r02
python tools/_det.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --exchange-every 2 2>/dev/null | tail -n 1 > gpurun_out/r02/bench9_$name.json; }
run own_q8 CDAE_BENCH_FORCE_DIST=1 CDAE_BENCH_NO_COMM=1
run aux_q8 CDAE_BENCH_FORCE_DIST=1 CDAE_BENCH_NO_COMM=1 CDAE_XCHG_STREAM=aux
run main_q8 CDAE_BENCH_FORCE_DIST=1 CDAE_BENCH_NO_COMM=1 CDAE_XCHG_STREAM=main
run aux_q4 CDAE_BENCH_FORCE_DIST=1 CDAE_BENCH_NO_COMM=1 CDAE_XCHG_STREAM=aux GPU_MAX_HW_QUEUES=4
run own_q4 CDAE_BENCH_FORCE_DIST=1 CDAE_BENCH_NO_COMM=1 GPU_MAX_HW_QUEUES=4
run rccl_aux_q8 CDAE_BENCH_FORCE_DIST=1 CDAE_XCHG_STREAM=aux
run rccl_aux_q4 CDAE_BENCH_FORCE_DIST=1 CDAE_XCHG_STREAM=aux GPU_MAX_HW_QUEUES=4
run rccl_own_q4 CDAE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4
run rccl_own_q6 CDAE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=6
run none_q4 GPU_MAX_HW_QUEUES=4
run none_q8 GPU_MAX_HW_QUEUES=8
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/bench9_*.json')):
    d=json.load(open(f))
    print(f.split('bench9_')[1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})
PY
