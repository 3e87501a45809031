#!/bin/bash
# dev helper (GPU box): A/B of the one-image-per-call path in C++ (wm_single) between builds of libwm_hip.so, alternating
# runs.  usage: tools/single_ab.sh "<variant> [<variant> ...]" [rounds] [rows cols dtype mask]
# (variant: watermarking-gpu_amd/libwm_ab_<name>.so, built by tools/build_variant.sh; "tree" = the in-tree libwm_hip.so)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P="$ROOT/watermarking-gpu_amd"
VS="$1 tree"; N=${2:-4}; R=${3:-2160}; C=${4:-3840}; DT=${5:-f32}; MK=${6:-ME}
for V in $1; do mkdir -p "$P/ab_$V" && cp "$P/libwm_ab_$V.so" "$P/ab_$V/libwm_hip.so" || exit 1; done
for i in $(seq 1 "$N"); do
  for V in $VS; do
    if [ "$V" = tree ]; then L=""; else L="$P/ab_$V"; fi
    printf "%-8s" "$V:"; LD_LIBRARY_PATH="$L" timeout -k 10 60 "$P/wm_single" "$R" "$C" 300 "$DT" "$MK" /tmp | grep -o '"embed_us.*"pair_us": [0-9.]*'
  done
done
