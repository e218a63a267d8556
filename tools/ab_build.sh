#!/bin/bash
# Dev tool: build libionode variants with different -D flags into neural-ode-ion-channels_amd/variants/<name>/libionode.so
# usage: tools/ab_build.sh name "EXTRA flags"
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2
make -C neural-ode-ion-channels_amd/csrc -s clean
make -C neural-ode-ion-channels_amd/csrc -s -j8 EXTRA="$extra"
mkdir -p neural-ode-ion-channels_amd/variants/$name
cp neural-ode-ion-channels_amd/libionode.so neural-ode-ion-channels_amd/variants/$name/
