#!/bin/bash
# Timing-experiment builds of one kernel file (default conv_f32_wino32.hip, -DX_DBG=<bits>, see that file; ABFILE=<name> for
# another translation unit, TILES=<forced tile ids> for the sweep): one
# library per value under tools/ab/ (git-ignored, travels with gpurun), then `run` times them side by side on ONE
# GPU box on the Winograd shapes of yolov3-608 at batch 64.  Results of X_DBG != 0 builds are garbage by design.
#   bash tools/ab_builds.sh build "0 7 64 128 256"       (here, cross-compiles)
#   bash tools/ab_builds.sh run   "0 7 64 128 256" [variant]   (on the GPU box)
# A value that is not a number is a tag whose compiler flags come from the environment: ABFLAGS_<tag>="-DXGT=4 ..."
set -e
cd "$(dirname "$0")/.."
MODE=$1; VALS=${2:-0}; VARIANT=${3:-30}
SRC=${ABFILE:-conv_f32_wino32}          # the translation unit the experiment flags apply to
if [ "$MODE" = build ]; then
  mkdir -p tools/ab yolo2_light_amd/csrc/build_repro
  for v in $VALS; do
    ( cd yolo2_light_amd/csrc
      case $v in ''|*[!0-9]*) fl_var=ABFLAGS_$v; FL=${!fl_var};; *) FL="-DYL_LAB -DX_DBG=$v";; esac
      SLP=""; [ "$SRC" = conv_f32_wino32 ] && SLP="-fno-slp-vectorize"       # as csrc/Makefile builds that file
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $SLP $FL -c $SRC.hip -o build_repro/${SRC}_x$v.o
      objs=$(ls build/*.o | grep -v "build/$SRC.o")
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libyolo2hip_x$v.so $objs build_repro/${SRC}_x$v.o -ldl -lpthread )
    echo "built tools/ab/libyolo2hip_x$v.so"
  done
else
  for v in $VALS; do
    echo "== build $v variant=$VARIANT"
    YOLO2HIP_LIB=$PWD/tools/ab/libyolo2hip_x$v.so timeout 300 python tools/sweep_conv.py --batch 64 --tiles ${TILES:-31} --only ${SHAPES:-6,9,12,15} --iters ${ITERS:-5} --variant $VARIANT 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['M'], r['C'], r['H'], r['kernel'], '%.3f ms %.1f TF' % (r['ms'], r['tflops']))
"
  done
fi
