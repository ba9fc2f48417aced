#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_full_output.sh'): rocprofv3 summaries of the full-output (matrix-core) path into
# gpurun_out/profile_full/: kernel stats at ML-10M shape (fused kernel) and at the configs[4] item space (three LDS-staged GEMMs),
# then PMC passes (separate runs, kernel trace only) for MFMA busy cycles and LDS bank conflicts of the GEMM kernels.
RND=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile_full
mkdir -p $O
ML="python $R/bench.py --no-cpu-baseline --full-output --batch-users 2048 --steps 40 --warmup 5"
C5="python $R/bench.py --no-cpu-baseline --full-output --shape cfg5_items --num-dim 512 --batch-users 1024 --steps 6 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_ml -o res -- $ML > $O/ml.log 2>&1
python $R/tools/rocpd_summary.py stats /tmp/pf_ml/res_results.db > $O/${RND}_full_output_ml10m_kernel_stats.txt
grep '"metric"' $O/ml.log > $O/${RND}_full_output_ml10m.json
rm -rf /tmp/pf_ml
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_c5 -o res -- $C5 > $O/c5.log 2>&1
python $R/tools/rocpd_summary.py stats /tmp/pf_c5/res_results.db > $O/${RND}_full_output_cfg5_kernel_stats.txt
grep '"metric"' $O/c5.log > $O/${RND}_full_output_cfg5.json
rm -rf /tmp/pf_c5
: > $O/${RND}_full_output_cfg5_pmc.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pf_pmc -o res -- $C5 > $O/pmc.log 2>&1
  for c in $set; do python $R/tools/rocpd_summary.py pmc /tmp/pf_pmc/res_results.db $c | grep -E "counter|gemm_nt|gemm_tn|gemm1_loss|gemm3_rows|full_rows|to_bf16|bf16_transpose" >> $O/${RND}_full_output_cfg5_pmc.txt; done
  rm -rf /tmp/pf_pmc
done
cat $O/${RND}_full_output_cfg5_pmc.txt | cut -c1-60,108-180
