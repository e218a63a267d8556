#!/bin/bash
# Dev tool: rebuild ONE translation unit of libionode with extra flags into variants/<name>/libionode.so, then restore the default build.
# usage: tools/ab_one_unit.sh name unit.o "EXTRA flags"      (e.g. tools/ab_one_unit.sh m6one ionode_capi.o "-DIONODE_M6_TWO_FROM=1073741824")
set -e
cd "$(dirname "$0")/../neural-ode-ion-channels_amd/csrc"
name=$1; unit=$2; extra=$3
make -s -j8 2>&1 | grep -v "argument unused" || true
rm -f "$unit"
make -s -j8 EXTRA="$extra" 2>&1 | grep -v "argument unused" || true
mkdir -p ../variants/$name
cp ../libionode.so ../variants/$name/
rm -f "$unit"
make -s -j8 2>&1 | grep -v "argument unused" || true
ls -la ../libionode.so ../variants/$name/libionode.so
