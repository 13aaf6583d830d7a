#!/bin/bash
# Round 6, VERDICT r5 item 8: does pack_desc_kernel (HBM bound) run beside another batch's match_union8_kernel (VALU bound)?  Kernel
# traces of the matcher-only leg, three batches in flight: the product build, and a -DVISO_DEBUG_VARIANTS build whose union8
# launches carry unused dynamic LDS so that only 6 workgroups fit a CU (room for a pack workgroup's 15.5 KB and four waves).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
Q="--no-cpu --no-e2e --no-streaming --no-images --no-i16 --steps 60"
run() {  # name, env...
  name=$1; shift
  rm -rf gpurun_out/cores_$name
  env "$@" rocprofv3 --kernel-trace -d gpurun_out/cores_$name -o s --output-format csv -- python3 bench.py $Q > gpurun_out/cores_$name.json 2>gpurun_out/cores_$name.err
  echo "== $name ($*)"; python3 -c "import json;d=json.load(open('gpurun_out/cores_$name.json'));print('matcher only %.0f frames/s, %.4f ms per step' % (d['value'], d['ms_per_step']))"
  python3 tools/experiments/pack_coresidency.py gpurun_out/cores_$name/s_kernel_trace.csv
}
run product X=0
run dbg_pad0 VISO_HIP_SO=$GRAFT_REPO_ROOT/build_ab/dbg.so
run dbg_pad700 VISO_HIP_SO=$GRAFT_REPO_ROOT/build_ab/dbg.so VISO_EXP_U8_LDS_PAD=700
run dbg_pad4000 VISO_HIP_SO=$GRAFT_REPO_ROOT/build_ab/dbg.so VISO_EXP_U8_LDS_PAD=4000
run product_again X=0
