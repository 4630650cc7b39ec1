#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONPATH=$PWD/drl-based-mapless-crowd-navigation-with-perceived-risk_amd
D=/tmp/trainprof; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python -m crowdnav.train --scenario training_as_logged --envs 16 --updates 16 --waypoint-reward 0 --learner fused --launches 3000 --log-every 1000 --seed 0 --out /tmp/runp > /dev/null 2>&1
F=$(find $D -name '*kernel_stats.csv' | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over 3000 launches = %.1f us per launch" % (tot/1e6, tot/3e6))
for r in rows[:28]:
    print(f"{r['Name'][:90]:<90} {int(r['Calls']):>8} {float(r['AverageNs'])/1e3:>8.2f} us {100*float(r['TotalDurationNs'])/tot:>5.1f}%  per launch {int(r['Calls'])/3000:.2f}")
PY
