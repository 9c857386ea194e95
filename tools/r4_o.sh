set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4o
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || true
tail -8 $O/tests.txt
python tools/harness_bench.py 100 f16 > $O/harness.json 2> $O/harness.err || true
grep -A3 '"value"' $O/harness.json | head -8
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/htrace -o h -- python3 $R/tools/harness_bench.py 60 f16 > $O/harness_traced.json 2> $O/harness_traced.err || true
cd $R
TRACE_COMPACT=$O/harness_trace.csv.gz python tools/harness_timeline.py /tmp/htrace > $O/harness_timeline.txt 2>&1 || true
head -40 $O/harness_timeline.txt
