#!/bin/bash
# usage: tools/mkvariant.sh <name> <act_dim> <extra hipcc flags...>
# builds mppi_gpu_amd/lib/alt_<name>/libmppi_gpu_amd.so: the product library with ONLY the packed
# rollout unit of one act_dim recompiled with the given flags (A/B experiments, see tools/abn.sh)
name=$1; A=$2; shift 2
cd "$(dirname "$0")/../mppi_gpu_amd/csrc" || exit 1
mkdir -p ../lib/alt_$name
objs=$(ls ../lib/obj/*.o | grep -v "rollout_packed_a$A.o")
/opt/rocm/bin/hipcc -O3 -ffp-contract=off -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c -o ../lib/alt_$name/rollout_packed_a$A.o rollout_packed_a$A.hip || exit 1
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o ../lib/alt_$name/libmppi_gpu_amd.so $objs ../lib/alt_$name/rollout_packed_a$A.o
