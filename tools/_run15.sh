mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_multi.py -m gpu -q -s -k "item_rows" 2>&1 | tail -n 60 > gpurun_out/r02/gputest15.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r02/gputest15.log | tail -n 40
