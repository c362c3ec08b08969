#!/bin/bash
# Instrumented builds of the library for tools/exp_rroi_ablate.py / exp_rroi_stamps.py (OBB_RROI_ABLATE = 1..5 in
# aidet_amd/csrc/obb_rroi.hip): aidet_amd/lib/libobbops_abl<n>.so, and for tools/exp_gprobe_stamps.py
# (OBB_GPROBE_STAMPS in obb_nms.hip): libobbops_gpst.so.  Built in scratch copies of csrc/ (the objects of
# the product build are left alone).  The libraries are git-ignored like every *.so and travel with gpurun snapshots.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for a in ${ABL-1 3 4 5 6}; do     # (ABL= bash tools/build_ablations.sh: the probe-stamp library only)
  d=$(mktemp -d /tmp/obb_abl${a}_XXXX)
  cp $ROOT/aidet_amd/csrc/*.hip $ROOT/aidet_amd/csrc/*.h $ROOT/aidet_amd/csrc/Makefile $d/
  sed -i "s#\.\./\.\./include#$ROOT/include#g" $d/Makefile $d/obb_common.h
  make -s -C $d -j8 EXTRA=-DOBB_RROI_ABLATE=$a OUT=$ROOT/aidet_amd/lib/libobbops_abl$a.so
  rm -rf $d
done
# stamps of the grid route's probe kernel (tools/exp_gprobe_stamps.py)
d=$(mktemp -d /tmp/obb_gpst_XXXX)
cp $ROOT/aidet_amd/csrc/*.hip $ROOT/aidet_amd/csrc/*.h $ROOT/aidet_amd/csrc/Makefile $d/
sed -i "s#\.\./\.\./include#$ROOT/include#g" $d/Makefile $d/obb_common.h
make -s -C $d -j8 EXTRA=-DOBB_GPROBE_STAMPS OUT=$ROOT/aidet_amd/lib/libobbops_gpst.so
rm -rf $d
# stamps of the position-free Soft-NMS kernel (tools/exp_soft_stamps.py)
d=$(mktemp -d /tmp/obb_sst_XXXX)
cp $ROOT/aidet_amd/csrc/*.hip $ROOT/aidet_amd/csrc/*.h $ROOT/aidet_amd/csrc/Makefile $d/
sed -i "s#\.\./\.\./include#$ROOT/include#g" $d/Makefile $d/obb_common.h
make -s -C $d -j8 EXTRA=-DOBB_SOFT_STAMPS OUT=$ROOT/aidet_amd/lib/libobbops_sst.so
rm -rf $d
# stamps of the grid route's frame and insert kernels (tools/exp_frame_stamps.py)
d=$(mktemp -d /tmp/obb_frst_XXXX)
cp $ROOT/aidet_amd/csrc/*.hip $ROOT/aidet_amd/csrc/*.h $ROOT/aidet_amd/csrc/Makefile $d/
sed -i "s#\.\./\.\./include#$ROOT/include#g" $d/Makefile $d/obb_common.h
make -s -C $d -j8 EXTRA=-DOBB_FRAME_STAMPS OUT=$ROOT/aidet_amd/lib/libobbops_frst.so
rm -rf $d
# stamps of the channels-last RoI-Align forward (tools/exp_nhwc_stamps.py)
d=$(mktemp -d /tmp/obb_nhst_XXXX)
cp $ROOT/aidet_amd/csrc/*.hip $ROOT/aidet_amd/csrc/*.h $ROOT/aidet_amd/csrc/Makefile $d/
sed -i "s#\.\./\.\./include#$ROOT/include#g" $d/Makefile $d/obb_common.h
make -s -C $d -j8 EXTRA=-DOBB_NHWC_STAMPS OUT=$ROOT/aidet_amd/lib/libobbops_nhst.so
rm -rf $d
# stamps of the backward's plan kernel (tools/exp_plan_stamps.py)
d=$(mktemp -d /tmp/obb_plst_XXXX)
cp $ROOT/aidet_amd/csrc/*.hip $ROOT/aidet_amd/csrc/*.h $ROOT/aidet_amd/csrc/Makefile $d/
sed -i "s#\.\./\.\./include#$ROOT/include#g" $d/Makefile $d/obb_common.h
make -s -C $d -j8 EXTRA=-DOBB_PLAN_STAMPS OUT=$ROOT/aidet_amd/lib/libobbops_plst.so
rm -rf $d
# per-sweep stamps of the resolve kernel (tools/exp_resolve_stamps.py)
d=$(mktemp -d /tmp/obb_rsst_XXXX)
cp $ROOT/aidet_amd/csrc/*.hip $ROOT/aidet_amd/csrc/*.h $ROOT/aidet_amd/csrc/Makefile $d/
sed -i "s#\.\./\.\./include#$ROOT/include#g" $d/Makefile $d/obb_common.h
make -s -C $d -j8 EXTRA=-DOBB_RESOLVE_STAMPS OUT=$ROOT/aidet_amd/lib/libobbops_rsst.so
rm -rf $d
ls -la $ROOT/aidet_amd/lib/libobbops_rsst.so $ROOT/aidet_amd/lib/libobbops_plst.so $ROOT/aidet_amd/lib/libobbops_nhst.so $ROOT/aidet_amd/lib/libobbops_abl*.so $ROOT/aidet_amd/lib/libobbops_gpst.so $ROOT/aidet_amd/lib/libobbops_sst.so
