set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4w
mkdir -p $O
for l in 1 2 3; do python tools/harness_bench.py 100 f16 --lanes $l > $O/h$l.json 2>/dev/null || true; echo "lanes $l: $(grep '"value"' $O/h$l.json) $(grep 'timing clip0' $O/h$l.json)"; done
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/htrace -o h -- python3 $R/tools/harness_bench.py 60 f16 --lanes 3 > $O/harness_traced.json 2> $O/harness_traced.err || true
cd $R
python tools/harness_timeline.py /tmp/htrace > $O/harness_timeline.txt 2>&1 || true
head -30 $O/harness_timeline.txt
