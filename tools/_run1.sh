mkdir -p gpurun_out/s3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_rows or wide_gemm or three_gemm or 65536" > gpurun_out/s3/t_fused.log 2>&1; tail -5 gpurun_out/s3/t_fused.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "CDAE_DEBUG_SKIP_ROLES=0" "CDAE_FULL_ROWS_KH=1" "CDAE_DEBUG_SKIP_ROLES=32" "CDAE_FULL_ROWS_SEPARATE=1"; do
rm -rf /tmp/pf_c5
env $v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_c5 -o res -- python $R/bench.py --no-cpu-baseline --full-output --shape cfg5_items --num-dim 512 --batch-users 1024 --steps 6 --warmup 2 > /tmp/c5_prof.log 2>&1
echo "== $v"; grep '"metric"' /tmp/c5_prof.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; python $R/tools/rocpd_summary.py stats /tmp/pf_c5/res_results.db | grep -E "gemm3_rows|full_rows|gemm_nt|transpose" | cut -c1-60,108-170
done
