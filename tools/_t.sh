set -e
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -o t -- python3 $R/bench.py --train --batch 20 --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, subprocess
f = glob.glob('/tmp/tt/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot / 1e6 / 3)
for r in rows[:26]:
    n = r['Name']
    if n.startswith('_Z'):
        pass
    print(f"{n[:100]:100s} {int(r['Calls'])//3:6d}/step {float(r['TotalDurationNs'])/1e6/3:8.2f} ms/step {float(r['AverageNs'])/1e3:8.1f} us")
PY
