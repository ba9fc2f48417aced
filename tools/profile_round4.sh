#!/bin/bash
# Run on the GPU box at the FINAL state of a round: gpurun -- 'bash tools/profile_round4.sh r04'
# default bench: kernel trace + FETCH/WRITE PMC passes (tools/profile_round.sh); K = 512 full-output step: kernel trace + PMC passes
# (tools/profile_full_output.sh); then the traffic JSONs bench.py reads back.  Everything lands in gpurun_out/profiles_<round>/ —
# copy it into profiles/.
RND=${1:-r04}
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/profiles_$RND; mkdir -p $D
bash $R/tools/profile_round.sh $RND > $D/profile_round.log 2>&1
cp $R/gpurun_out/profile/${RND}_* $D/
bash $R/tools/profile_full_output.sh $RND > $D/profile_full.log 2>&1
cp $R/gpurun_out/profile_full/${RND}_* $D/
cd $R
mkdir -p profiles_tmp
python tools/traffic_json.py decode $RND $D/${RND}_bench_pmc_fetch_size.txt $D/${RND}_bench_pmc_write_size.txt $D/${RND}_bench_kernel_stats.txt --shape ml10m --num-dim 200 --batch-users 256
python tools/traffic_json.py full $RND $D/${RND}_full_output_cfg5_pmc.txt - $D/${RND}_full_output_cfg5_kernel_stats.txt --shape cfg5_items --num-dim 512 --batch-users 1024
cp profiles/${RND}_decode_traffic.json profiles/${RND}_full_traffic.json $D/
ls $D
