set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/corrprof
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp -o c -- python3 $R/tools/bench_corr_diag.py > $R/gpurun_out/corrprof/out.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/cp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
