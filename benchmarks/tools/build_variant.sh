#!/bin/bash
# Build a variant of the library beside the default one (build container):
#   benchmarks/tools/build_variant.sh <name> "<unit prefix>=<flags>[;<unit prefix>=<flags>...]"
# -> ssspy_amd/lib/libssspy_amd_<name>.so; then A / B it on the GPU box with benchmarks/tools/ab_lib.sh.
set -e
name=$1; spec=$2
cd "$(dirname "$0")/../.."
SSSPY_AMD_UNIT_FLAGS="$spec" python -m ssspy_amd._build > /dev/null
cp ssspy_amd/lib/libssspy_amd.so ssspy_amd/lib/libssspy_amd_$name.so
python -m ssspy_amd._build > /dev/null
echo ssspy_amd/lib/libssspy_amd_$name.so
