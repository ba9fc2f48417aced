set -x
mkdir -p gpurun_out/r02
python tools/accuracy_envelope.py --batch-users 16 32 64 128 > gpurun_out/r02/env_single_small.log 2>&1
python tools/accuracy_envelope.py --batch-users 1 --seeds 1234 > gpurun_out/r02/env_single_b1.log 2>&1
python tools/accuracy_envelope.py --shards 2 8 --batch-users 64 128 --period 0 --rule 1 --seeds 20141119 1234 > gpurun_out/r02/env_touchmean.log 2>&1
