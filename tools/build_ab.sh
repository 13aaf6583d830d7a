#!/bin/bash
# tools/build_ab.sh NAME [REV]: build libviso_hip.so of git revision REV (default: the working tree) into build_ab/NAME.so
# (build_ab/ is git-ignored and travels with gpurun), for tools/ab_so.sh
set -e
name=$1; rev=$2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_ab
if [ -z "$rev" ]; then
  make -C $root/libviso_amd/csrc >/dev/null 2>&1
  cp $root/libviso_amd/libviso_hip.so $root/build_ab/$name.so
else
  tmp=$(mktemp -d)
  git -C $root archive $rev libviso_amd/csrc include | tar -x -C $tmp
  make -C $tmp/libviso_amd/csrc >/dev/null 2>&1
  cp $tmp/libviso_amd/libviso_hip.so $root/build_ab/$name.so
  rm -rf $tmp
fi
ls -la $root/build_ab/$name.so
