#!/bin/bash
# tools/fuzz_parity.py for a while, then every world that differed once more with details:  tools/fuzz_and_repro.sh <seed> <seconds> [tag] [extra fuzz_parity.py arguments]
SEED="${1:-3}"; SECS="${2:-120}"; TAG="${3:-r06}"; EXTRA="${4:-}"; cd "$(dirname "$0")/.."; OUT="gpurun_out/$TAG"; mkdir -p "$OUT"
timeout $((SECS + 200)) python tools/fuzz_parity.py --seconds "$SECS" --seed "$SEED" $EXTRA 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_seed$SEED.txt"
grep -v "^refused" "$OUT/fuzz_seed$SEED.txt" | cut -c1-400 | tail -30
grep "bad world" "$OUT/fuzz_seed$SEED.txt" | head -8 | sed 's/.* | //' > "$OUT/fuzz_repro_args.txt"
i=0
while IFS= read -r line; do
  i=$((i+1)); echo "=== repro $i: $(echo "$line" | cut -c1-120)"
  eval "timeout 120 python tools/fuzz_parity.py $line" 2>&1 | grep -v amdgpu.ids | cut -c1-600
done < "$OUT/fuzz_repro_args.txt" > "$OUT/fuzz_repro_seed$SEED.txt" 2>&1
cat "$OUT/fuzz_repro_seed$SEED.txt"
