#!/bin/bash
# -DAGP_POTRF_TIMING builds of both libraries into scripts/variants/probe/ (git-ignored, travels with gpurun):
#   bash scripts/build_probe_libs.sh && gpurun -- 'cp scripts/variants/probe/*.so albatross_amd/ && python3 scripts/probe_potrf.py 512'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/agp_probe_build
rm -rf "$B" && mkdir -p "$B/albatross_amd" "$B/include" && cp -r "$ROOT/albatross_amd/csrc" "$B/albatross_amd/" && cp "$ROOT/include/albatross_amd.h" "$B/include/"
rm -rf "$B/albatross_amd/csrc/build"
make -s -j8 -C "$B/albatross_amd/csrc" HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -DAGP_POTRF_TIMING"
mkdir -p "$ROOT/scripts/variants/probe" && cp "$B/albatross_amd/"libalbatross_amd*.so "$ROOT/scripts/variants/probe/"
