#!/bin/bash
# build_variant.sh <name> <source.hip> <object dir name> [-D...]: libolsr_<name>.so = libolsr.so with one translation unit
# recompiled under extra defines (kernel experiments; select with OLSR_LIB=online_lang_splatting_amd/libolsr_<name>.so)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; objdir=$3; shift 3
C=online_lang_splatting_amd/csrc
mkdir -p /tmp/olsr_variant_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function "$@" -c $C/$src -o /tmp/olsr_variant_$name/v.o
objs=""
for d in $C/_obj/*/; do
  b=$(basename $d)
  if [ "$b" == "$objdir" ]; then objs="$objs /tmp/olsr_variant_$name/v.o"; else objs="$objs $d$b.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o online_lang_splatting_amd/libolsr_$name.so $objs
echo online_lang_splatting_amd/libolsr_$name.so
