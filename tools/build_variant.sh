#!/bin/bash
# dev helper: build a variant of libwm_hip.so with extra compiler flags into watermarking-gpu_amd/libwm_ab_<name>.so
# (git-ignored, travels to the GPU box); compare with tools/ab.py.   usage: tools/build_variant.sh <name> "<flags>"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/wm_variant_$1
rm -rf "$B" && mkdir -p "$B/watermarking-gpu_amd" "$B/include"
cp -r "$ROOT/watermarking-gpu_amd/csrc" "$B/watermarking-gpu_amd/" && cp "$ROOT"/include/*.h* "$B/include/"
rm -f "$B"/watermarking-gpu_amd/csrc/*.o
make -s -j4 -C "$B/watermarking-gpu_amd/csrc" CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result --offload-arch=gfx950 $2"
cp "$B/watermarking-gpu_amd/libwm_hip.so" "$ROOT/watermarking-gpu_amd/libwm_ab_$1.so"
echo "built watermarking-gpu_amd/libwm_ab_$1.so with: $2"
