#!/bin/bash
# Dev tool: build libionode.so from a git ref into neural-ode-ion-channels_amd/variants/<name>/ (for same-box A/B runs)
# usage: tools/ab_build_ref.sh <git-ref> <name> ["EXTRA flags"]
set -e
cd "$(dirname "$0")/.."
ref=$1; name=$2; extra=$3
tmp=$(mktemp -d /tmp/ionode_ab.XXXX)
git archive "$ref" include neural-ode-ion-channels_amd/csrc | tar -x -C "$tmp"
make -C "$tmp/neural-ode-ion-channels_amd/csrc" -s -j8 EXTRA="$extra"
mkdir -p neural-ode-ion-channels_amd/variants/$name
cp "$tmp/neural-ode-ion-channels_amd/libionode.so" neural-ode-ion-channels_amd/variants/$name/
rm -rf "$tmp"
