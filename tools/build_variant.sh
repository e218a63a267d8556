#!/bin/bash
# Dev tool: build a variant of libionode.so from the CURRENT sources in a scratch copy (the in-tree build is untouched):
#   tools/build_variant.sh name "EXTRA flags"   ->  neural-ode-ion-channels_amd/variants/<name>/libionode.so
# (IONODE_LIB=<that path> selects it; tools/gpu/*_ab.sh compare every variant with the in-tree library on one box)
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2
T=/tmp/ionode_variant_$name; rm -rf $T; mkdir -p $T/neural-ode-ion-channels_amd $T/tools
cp -r include $T/; cp tools/gen_mlp_asm.py $T/tools/
mkdir -p $T/neural-ode-ion-channels_amd/csrc
cp neural-ode-ion-channels_amd/csrc/*.hip neural-ode-ion-channels_amd/csrc/*.hpp neural-ode-ion-channels_amd/csrc/Makefile $T/neural-ode-ion-channels_amd/csrc/
make -C $T/neural-ode-ion-channels_amd/csrc -s -j8 EXTRA="$extra" 2>&1 | grep -v "argument unused\|warning generated\|unused variable\|^ *[0-9]* |\|^ *|\|In file included" || true
mkdir -p neural-ode-ion-channels_amd/variants/$name
cp $T/neural-ode-ion-channels_amd/libionode.so neural-ode-ion-channels_amd/variants/$name/
ls -la neural-ode-ion-channels_amd/variants/$name/libionode.so
