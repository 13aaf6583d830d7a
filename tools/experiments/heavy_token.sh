#!/bin/bash
# Round 6: serialise match_union8_kernel across the batches in flight (a device-wide event chain, $VISO_EXP_HEAVY_TOKEN=1 in a
# -DVISO_DEBUG_VARIANTS build) so that the other batches' pack / stereo / sort kernels run BESIDE it instead of a second
# union8; with and without LDS padding that keeps one workgroup slot per CU free of union8.  Matcher-only leg, kernel traces.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=. VISO_HIP_SO=$GRAFT_REPO_ROOT/build_ab/dbg.so
Q="--no-cpu --no-e2e --no-streaming --no-images --no-i16 --steps 60"
run() {  # name, streams, env...
  name=$1; st=$2; shift; shift
  rm -rf gpurun_out/ht_$name
  env "$@" rocprofv3 --kernel-trace -d gpurun_out/ht_$name -o s --output-format csv -- python3 bench.py $Q --streams $st > gpurun_out/ht_$name.json 2>gpurun_out/ht_$name.err
  echo "== $name (streams $st; $*)"; python3 -c "import json;d=json.load(open('gpurun_out/ht_$name.json'));print('matcher only %.0f frames/s, %.4f ms per step' % (d['value'], d['ms_per_step']))"
  python3 tools/experiments/pack_coresidency.py gpurun_out/ht_$name/s_kernel_trace.csv
}
run base3 3 X=0
run tok2 2 VISO_EXP_HEAVY_TOKEN=1
run tok3 3 VISO_EXP_HEAVY_TOKEN=1
run tok4 4 VISO_EXP_HEAVY_TOKEN=1
run tok2_pad 2 VISO_EXP_HEAVY_TOKEN=1 VISO_EXP_U8_LDS_PAD=700
run tok3_pad 3 VISO_EXP_HEAVY_TOKEN=1 VISO_EXP_U8_LDS_PAD=700
run tok4_pad 4 VISO_EXP_HEAVY_TOKEN=1 VISO_EXP_U8_LDS_PAD=700
run base3_again 3 X=0
