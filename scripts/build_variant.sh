#!/bin/bash
# Both libraries built with extra compiler flags into scripts/variants/<name>/ (git-ignored, travels with gpurun):
#   bash scripts/build_variant.sh probe -DAGP_POTRF_TIMING
#   gpurun -- 'cp scripts/variants/probe/*.so albatross_amd/ && python3 scripts/probe_potrf.py 512'
# (the copy happens on the GPU box's snapshot only; scripts/ab.sh swaps the product library alone)
set -e
NAME=${1:?usage: build_variant.sh <name> [flags...]}
shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/agp_variant_build_$NAME
rm -rf "$B" && mkdir -p "$B/albatross_amd" "$B/include" && cp -r "$ROOT/albatross_amd/csrc" "$B/albatross_amd/" && cp "$ROOT/include/albatross_amd.h" "$B/include/"
rm -rf "$B/albatross_amd/csrc/build"
# (EXTRA, not HIPFLAGS: a HIPFLAGS given on the command line would switch off the per-file -ffp-contract=off of gram.o / ldlt.o)
make -s -j8 -C "$B/albatross_amd/csrc" EXTRA="$*"
mkdir -p "$ROOT/scripts/variants/$NAME" && cp "$B/albatross_amd/"libalbatross_amd*.so "$ROOT/scripts/variants/$NAME/"
rm -rf "$B"
