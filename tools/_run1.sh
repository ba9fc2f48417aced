set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -n 80 > gpurun_out/r02/gputest1.log
python tools/accuracy_envelope.py --batch-users 192 256 320 384 448 512 640 > gpurun_out/r02/env_single.log 2>&1
python tools/accuracy_envelope.py --shards 8 --batch-users 32 64 128 --period 0 1 2 > gpurun_out/r02/env_shards8.log 2>&1
python tools/accuracy_envelope.py --shards 2 4 --batch-users 128 256 --period 0 2 --seeds 20141119 > gpurun_out/r02/env_shards24.log 2>&1
tail -n 5 gpurun_out/r02/gputest1.log
