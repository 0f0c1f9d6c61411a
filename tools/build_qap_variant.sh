#!/bin/bash
# tools/build_qap_variant.sh NAME [FLAGS...]: libfrw.so with frw_qap.hip compiled with FLAGS, as tools/variants/libfrw_qap_NAME.so
# (for tools/ab_qap.py: the variants are built here, next to the sources, and travel to the GPU box with the snapshot)
set -euo pipefail
cd "$(dirname "$0")/../falcon-r1cs_amd/csrc"
name=$1; shift
mkdir -p ../../tools/variants build
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -c frw_qap.hip -o /tmp/frw_qap_$name.o
objs=$(ls build/*.o | grep -v frw_qap.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/libfrw_qap_$name.so $objs /tmp/frw_qap_$name.o
echo tools/variants/libfrw_qap_$name.so
