#!/bin/bash
# Same-box A/B of library variants (round 4): the shape-specialised kernels against the generic ones (CN_NO_SHAPE_KERNELS=1) and a
# build without MachineLICM (lib/libcrowdnav_nolicm.so: hipcc ... -mllvm -disable-machine-licm), one-launch / groups / sequence legs.
cd "$(dirname "$0")/.."
PKG=drl-based-mapless-crowd-navigation-with-perceived-risk_amd
{ python tools/quick_perf.py new; CN_NO_SHAPE_KERNELS=1 python tools/quick_perf.py new_generic_kernels
  [ -f $PKG/lib/libcrowdnav_nolicm.so ] && CN_LIB=$PKG/lib/libcrowdnav_nolicm.so python tools/quick_perf.py nolicm
  python tools/quick_perf.py new_again
  [ -f $PKG/lib/libcrowdnav_nolicm.so ] && CN_LIB=$PKG/lib/libcrowdnav_nolicm.so python tools/quick_perf.py nolicm_again
  python tools/seq_perf.py 4096
  [ -f $PKG/lib/libcrowdnav_nolicm.so ] && CN_LIB=$PKG/lib/libcrowdnav_nolicm.so python tools/seq_perf.py 4096; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_ab1.txt
cat gpurun_out/r04_ab1.txt
