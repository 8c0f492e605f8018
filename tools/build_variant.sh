#!/bin/bash
# Build a tuning variant of libppo_amd.so: one source recompiled with extra -D flags, linked with the current objects.
# usage: tools/build_variant.sh <name> <file.hip> -DFLAG=...   -> ppo_amd/lib/libppo_amd_<name>.so  (run with PPO_AMD_LIB=...)
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
OBJ=$ROOT/ppo_amd/lib/obj
mkdir -p $OBJ/var_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I $ROOT/include "$@" -c $ROOT/ppo_amd/csrc/$SRC -o $OBJ/var_$NAME/${SRC%.hip}.o
OBJS=$(ls $OBJ/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/ppo_amd/lib/libppo_amd_$NAME.so $OBJS $OBJ/var_$NAME/${SRC%.hip}.o
echo $ROOT/ppo_amd/lib/libppo_amd_$NAME.so
