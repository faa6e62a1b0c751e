#!/bin/bash
# Dev helper: build the COMMITTED sources (HEAD) into stereoformer_amd/ab/lib_head.so so that a working-tree change can be
# A/B-timed inside ONE gpurun call (boxes differ by ~10 %):  STEREO_HIP_LIB=$PWD/stereoformer_amd/ab/lib_head.so python ...
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/ab_head && mkdir -p /tmp/ab_head/stereoformer_amd /tmp/ab_head/include
git archive HEAD stereoformer_amd/csrc include | tar -x -C /tmp/ab_head
cd /tmp/ab_head
for f in stereoformer_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-fast-math -fno-slp-vectorize -w -c $f -o $(basename $f .hip).o &
done
wait
mkdir -p "$OLDPWD/stereoformer_amd/ab"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OLDPWD/stereoformer_amd/ab/lib_head.so" *.o
ls -la "$OLDPWD/stereoformer_amd/ab/lib_head.so"
