set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4m
mkdir -p $O
cd /tmp
BENCH="python3 $R/bench.py --cpu-baseline none --no-harness --no-extras --steps 12 --warmup 3"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/t_new -o n -- $BENCH > $O/new.json 2> $O/new.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/t_old -o o -- $BENCH --knobs '{"attn_win4": false}' > $O/old.json 2> $O/old.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/t_new1 -o n -- $BENCH --inflight 1 --streams 1 > $O/new1.json 2> $O/new1.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/t_old1 -o o -- $BENCH --inflight 1 --streams 1 --knobs '{"attn_win4": false}' > $O/old1.json 2> $O/old1.err
cd $R
for t in new old new1 old1; do python tools/harness_timeline.py /tmp/t_$t 2 > $O/tl_$t.txt 2>&1 || true; echo "== $t"; head -22 $O/tl_$t.txt; done
