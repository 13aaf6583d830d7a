#!/bin/bash
# tools/experiments/dropin_swap.sh A.so B.so ...: the per-call loop (tools/dropin_probe.py, whose C++ host library is LINKED against
# libviso_amd/libviso_hip.so and does not look at $VISO_HIP_SO) with each build copied over that file -- on the GPU box's
# disposable copy of the tree only
cd $GRAFT_REPO_ROOT
cp libviso_amd/libviso_hip.so /tmp/libviso_hip.keep
for so in "$@" "$@"; do
  cp $so libviso_amd/libviso_hip.so
  printf "%s: " $so; python3 tools/dropin_probe.py 257 2000 2>&1 | grep "^frames" | cut -c1-70
done
cp /tmp/libviso_hip.keep libviso_amd/libviso_hip.so
