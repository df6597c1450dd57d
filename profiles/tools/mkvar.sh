#!/bin/bash
# usage: mkvar.sh NAME SRCDIR [extra flags] -> build/variants/lib_NAME.so
name=$1; src=$2; shift 2
mkdir -p /root/repo/build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I/root/repo/include -I/root/repo/terrarium.jl_amd/csrc "$@" -shared -o /root/repo/build/variants/lib_$name.so $src/terrarium_hip.hip 2>&1 | grep -E "error" -A5
