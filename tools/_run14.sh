mkdir -p gpurun_out/r02
python tools/accuracy_envelope.py --shards 8 --batch-users 32 64 128 --period 0 2 --warm-epochs 2 > gpurun_out/r02/env_warm2_shards8.log 2>&1
python tools/accuracy_envelope.py --shards 2 4 --batch-users 64 128 --period 0 2 > gpurun_out/r02/env_multi_shards24.log 2>&1
python tools/accuracy_envelope.py --shards 2 8 --batch-users 128 --period 2 --warm-epochs 1 > gpurun_out/r02/env_warm1.log 2>&1
tail -n 2 gpurun_out/r02/env_warm1.log
