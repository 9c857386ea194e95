set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4a
mkdir -p $O
python bench.py --cpu-baseline none --no-extras --no-harness --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python tools/bench_swin.py > $O/bench_swin.txt 2>&1
python tools/harness_bench.py 60 f16 > $O/harness.json 2> $O/harness.err
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/htrace -o h -- python3 $R/tools/harness_bench.py 60 f16 > $O/harness_traced.json 2> $O/harness_traced.err
cd $R
python tools/harness_timeline.py /tmp/htrace > $O/harness_timeline.txt 2>&1
cat $O/bench.json | cut -c1-400; cat $O/harness.json; cat $O/harness_timeline.txt | head -70
