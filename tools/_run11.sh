mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q 2>&1 | tail -n 40 > gpurun_out/r02/gputest11.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r02/gputest11.log | tail -n 14
