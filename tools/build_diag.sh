#!/bin/bash
# diagnostic builds of the library (diag/*.so travel to the GPU box; git ignores *.so):
#   tools/build_diag.sh stats -DR2S_ISO_STATS        -> diag/stats.so   (tools/iso_phase_stats.py)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/diag
C=$ROOT/rho2sdf.jl_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 "$@" -o $ROOT/diag/$NAME.so \
    $C/rho2sdf_hip.hip $C/r2s_pre.hip $C/r2s_post.hip $C/r2s_io.hip $C/r2s_host.hip -lz
echo built diag/$NAME.so
