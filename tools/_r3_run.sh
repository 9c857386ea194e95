export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/r3tt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r3tt -o t -- python3 $R/bench.py --train --batch 20 --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/r3_trace.err
cd $R
python3 tools/stats_md.py /tmp/r3tt $R/gpurun_out/r3_train_stats "wip" "x" "y" 4
sed -n 7,34p $R/gpurun_out/r3_train_stats.md | cut -c1-150
