#!/bin/bash
# A/B build of the HIP library with extra compiler flags, for same-box comparisons through SA_HIP_LIB:
#   bash scripts/ab_build.sh <tag> <flags...>    ->  scripts/microbench/bin/libssl_audio_hip_<tag>.so   (git-ignored, travels with gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
B=/tmp/sa_ab_$TAG; rm -rf $B; mkdir -p $B/ssl_audio_amd $B/include
cp -r $ROOT/ssl_audio_amd/csrc $B/ssl_audio_amd/csrc; cp $ROOT/include/*.h $B/include/
rm -f $B/ssl_audio_amd/csrc/*.o $B/ssl_audio_amd/csrc/*.so
make -C $B/ssl_audio_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -fno-gpu-rdc $*" > $B/build.log 2>&1 || { tail -20 $B/build.log; exit 1; }
mkdir -p $ROOT/scripts/microbench/bin
cp $B/ssl_audio_amd/csrc/libssl_audio_hip.so $ROOT/scripts/microbench/bin/libssl_audio_hip_$TAG.so
echo "built scripts/microbench/bin/libssl_audio_hip_$TAG.so with: $*"
