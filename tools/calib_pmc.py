#!/usr/bin/env python
"""Known-byte streaming kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box at the
access widths cn_env_kernel uses (4 and 8 bytes per lane).  Run under rocprofv3 --pmc FETCH_SIZE (or
WRITE_SIZE); tools/profile.sh does, and tools/summarize_prof.py turns the result into factors."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch  # noqa: E402
import crowdnav  # noqa: E402

L = C.CDLL(crowdnav._abi.build_timing())   # the calibration kernels live in the profiling build only
L.cn_calib_launch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
L.cn_calib_launch.restype = None
BYTES = 1 << 30   # 1 GiB: well past the 256 MiB Infinity Cache
buf = torch.zeros(BYTES // 8, dtype=torch.float64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for write in (0, 1):
    for width in (4, 8):
        for _ in range(3):
            L.cn_calib_launch(buf.data_ptr(), BYTES, width, write, st)
torch.cuda.synchronize()
print("calibration launches done: 1 GiB per launch, reads then writes, widths 4 and 8")
