#!/bin/bash
# Everything under profiles/<tag>/ except the rocprofv3 passes (tools/profile.sh) and the parity sweep (tools/parity_sweep.sh),
# in one go on the GPU box:  tools/regen_profiles.sh r02   -> gpurun_out/<tag>/*; copy what should be judged into profiles/<tag>/.
TAG="${1:-r05}"; cd "$(dirname "$0")/.."; OUT="gpurun_out/$TAG"; mkdir -p "$OUT"
F='amdgpu.ids'
python bench.py 2>/dev/null | tail -1 > "$OUT/bench_n1.json"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/bench_n1_driver_command.json"
python tools/stage_timing.py 256 2>&1 | grep -v $F > "$OUT/stage_timing.txt"
{ CN_ARB=oldest_first python tools/wave_tail.py; CN_ARB=fair python tools/wave_tail.py; } 2>&1 | grep -v $F > "$OUT/wave_tail.txt"
python tools/tail_corr.py 2>&1 | grep -v $F > "$OUT/tail_corr.txt"
python tools/bench_configs.py 2>&1 | grep -v $F > "$OUT/configs_1gpu.txt"
{ python tools/quick_perf.py lidar-tracker; CN_RISK=1 python tools/quick_perf.py gt; CN_LAYOUT=1 python tools/quick_perf.py layout1; CN_LAYOUT=2 python tools/quick_perf.py layout2; CN_PY2=1 CN_GEOS=1 python tools/quick_perf.py py2+geos38; CN_PED_MODE=2 python tools/quick_perf.py social-force; } 2>&1 | grep -v $F > "$OUT/quick_perf.txt"
python tools/startup_transient.py 24 3 2>&1 | grep -v $F > "$OUT/startup_transient.txt"
python tools/cfg3_perf.py 2>&1 | grep -v $F > "$OUT/config3_decompositions.txt"
{ python tools/seq_perf.py 4096; python tools/seq_perf.py 16384; } 2>&1 | grep -v $F > "$OUT/step_sequence.txt"
{ python tools/actor_bench.py 4096; python tools/actor_bench.py 1024; python tools/actor_bench.py 16384; } 2>&1 | grep -v $F > "$OUT/actor.txt"
python tools/actor_timing.py 2>&1 | grep -v $F > "$OUT/actor_timing.txt"
CN_ARB=oldest_first python tools/wave_fairness.py 2>&1 | grep -v $F > "$OUT/wave_fairness_oldest_first.txt"
CN_ARB=fair python tools/wave_fairness.py 2>&1 | grep -v $F > "$OUT/wave_fairness_fair.txt"
[ -x tools/micro/bin/l2_stream ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/l2_stream tools/micro/l2_stream.hip 2>/dev/null
tools/micro/bin/l2_stream > "$OUT/l2_stream.txt" 2>&1
tail -n 3 "$OUT"/bench_n1_driver_command.json | cut -c1-300; cat "$OUT/quick_perf.txt"
