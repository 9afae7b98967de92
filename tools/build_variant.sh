#!/bin/bash
# tools/build_variant.sh <name> <source.hip> [flags...]: libbcbf with ONE translation unit recompiled under extra flags
# (development: timing variants through BCBF_LIB_PATH=tools/_variants/libbcbf_<name>.so).  Run `python -m bayesian_cbf_amd.build` first.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
name=$1; src=$2; shift 2
mkdir -p $ROOT/tools/_variants
obj=$ROOT/tools/_variants/${src%.hip}_$name.o
extra=""; [ "$src" = refit_wave64.hip ] && extra="-fno-slp-vectorize"   # (build.py EXTRA_FLAGS)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$ROOT/include -I$ROOT/bayesian_cbf_amd/csrc $extra "$@" -c $ROOT/bayesian_cbf_amd/csrc/$src -o $obj
others=$(ls $ROOT/bayesian_cbf_amd/csrc/_obj/*.o | grep -v "/${src%.hip}\(_[a-z0-9]*\)\?\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_variants/libbcbf_$name.so $obj $others
echo built tools/_variants/libbcbf_$name.so
