#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r02'): writes the round's rocprofv3 summaries to gpurun_out/profile/;
# copy <round>_*.txt from there into profiles/ and update profiles/<round>_decode_traffic.json from the decode rows.
# kernel trace + stats, then FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, never with a trace domain other than kernel-trace),
# of the DEFAULT bench command (bench.py with no flags but --no-cpu-baseline and a shorter step count)
RND=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile
mkdir -p $O
CMD="python $R/bench.py --no-cpu-baseline --steps 120 --warmup 20"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o res -- $CMD > $O/kt_bench.log 2>&1
python $R/tools/rocpd_summary.py stats /tmp/p_kt/res_results.db > $O/${RND}_bench_kernel_stats.txt
rm -rf /tmp/p_kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o res -- $CMD > $O/pmc_$c.log 2>&1
  python $R/tools/rocpd_summary.py pmc /tmp/p_$c/res_results.db $c > $O/${RND}_bench_pmc_$(echo $c | tr A-Z a-z).txt
  rm -rf /tmp/p_$c
done
tail -1 $O/kt_bench.log | cut -c1-400
