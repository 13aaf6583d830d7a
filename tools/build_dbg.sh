#!/bin/bash
# tools/build_dbg.sh [NAME]: a -DVISO_DEBUG_VARIANTS build of the working tree's libviso_hip.so into build_ab/NAME.so (default dbg), compiled in a
# scratch copy of the sources so that no debug object ever sits beside the product's objects
set -e
name=${1:-dbg}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/libviso_amd $tmp/tools $root/build_ab
cp -r $root/libviso_amd/csrc $tmp/libviso_amd/csrc
cp -r $root/include $tmp/include
cp -r $root/tools/experiments $tmp/tools/experiments
rm -f $tmp/libviso_amd/csrc/*.o
make -C $tmp/libviso_amd/csrc DEBUG_VARIANTS=1 >/dev/null 2>&1 || make -C $tmp/libviso_amd/csrc DEBUG_VARIANTS=1 2>&1 | grep -E "error" -A3 | head -20
cp $tmp/libviso_amd/libviso_hip.so $root/build_ab/$name.so
rm -rf $tmp
ls -la $root/build_ab/$name.so
