#!/bin/bash
# A/B library: tools/build_variant.sh <name> <file.hip> <extra hipcc flags...>  ->  pointrcnn_amd/libprcnn_<name>.so
# (the named source compiled with the extra flags, every other object as built in-tree; select with PRCNN_LIB)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m pointrcnn_amd.build > /dev/null
obj=/tmp/prcnn_variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c pointrcnn_amd/csrc/$src -o $obj
objs=$(ls pointrcnn_amd/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pointrcnn_amd/libprcnn_$name.so $objs $obj
echo pointrcnn_amd/libprcnn_$name.so
