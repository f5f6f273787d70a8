#!/bin/bash
# Builds platinum_amd/csrc/libptamd_stamps.so = the library with tools/shade_stamps.patch applied to a scratch copy of the sources.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
work=$(mktemp -d /tmp/ptamd_stamps.XXXXXX)
mkdir -p $work/platinum_amd $work/include
cp -r $root/platinum_amd/csrc $work/platinum_amd/csrc
cp -r $root/include/. $work/include/
(cd $work && patch -s -p1 < $root/tools/shade_stamps.patch)
make -s -C $work/platinum_amd/csrc libptamd.so
cp $work/platinum_amd/csrc/libptamd.so $root/platinum_amd/csrc/libptamd_stamps.so
echo built platinum_amd/csrc/libptamd_stamps.so
