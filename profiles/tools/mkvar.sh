#!/bin/bash
# usage: mkvar.sh NAME SRCDIR [extra compiler flags] -> build/variants/lib_NAME.so
# SRCDIR: a copy of terrarium.jl_amd/csrc (with ../../include beside it as in the tree), or the tree's own csrc.  The translation
# units compile in parallel into build/variants/obj_NAME/.
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/build/variants
make -C $src -s -j8 OUT=$root/build/variants/lib_$name.so OBJDIR=$root/build/variants/obj_$name EXTRA="$*" 2>&1 | grep -E "error" -A5
ls -la $root/build/variants/lib_$name.so
