#!/bin/bash
# developer aid: build libcdae_hip.so variants with extra -D flags into build/variants/<name>/ (loaded with CDAE_HIP_LIBRARY=...)
# usage: tools/build_variant.sh name -DFOO=1 ...      (always with -DCDAE_DEVELOPER: the environment switches are compiled in)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=build/variants/$name; mkdir -p $d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -w -DCDAE_DEVELOPER"
rm -f $d/cdae_hip.o $d/cdae_multi.o $d/libcdae_hip.so            # (never link a stale object behind a failed compile)
/opt/rocm/bin/hipcc $F "$@" -c cdae_amd/csrc/cdae_hip.hip -o $d/cdae_hip.o & p1=$!
/opt/rocm/bin/hipcc $F "$@" -c cdae_amd/csrc/cdae_multi.hip -o $d/cdae_multi.o & p2=$!
wait $p1; wait $p2                                                 # (set -e: either failure ends the script)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/cdae_hip.o $d/cdae_multi.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $d/libcdae_hip.so
echo built $d/libcdae_hip.so
