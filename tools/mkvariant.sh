#!/bin/bash
# usage: tools/mkvariant.sh <name> <act_dim | unit> <extra hipcc flags...>
# builds mppi_gpu_amd/lib/alt_<name>/libmppi_gpu_amd.so: the product library with ONLY one rollout
# unit recompiled with the given flags (A/B experiments, see tools/abn.sh).  <act_dim> alone means
# the packed unit rollout_packed_a<act_dim>; a unit name (e.g. rollout_fused_a2) is taken as is.
name=$1; A=$2; shift 2
case $A in [1-4]) unit=rollout_packed_a$A;; *) unit=$A;; esac
cd "$(dirname "$0")/../mppi_gpu_amd/csrc" || exit 1
mkdir -p ../lib/alt_$name
objs=$(ls ../lib/obj/*.o | grep -v "$unit.o")
/opt/rocm/bin/hipcc -O3 -ffp-contract=off -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c -o ../lib/alt_$name/$unit.o $unit.hip || exit 1
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o ../lib/alt_$name/libmppi_gpu_amd.so $objs ../lib/alt_$name/$unit.o
