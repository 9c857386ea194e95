set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/tr
cd /tmp
for b in 8 20; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt$b -o t -- python3 $R/bench.py --train --batch $b --steps 3 --warmup 1 > $R/gpurun_out/tr/bench$b.json 2> $R/gpurun_out/tr/err$b.txt
cd $R; python3 tools/stats_md.py /tmp/tt$b $R/gpurun_out/tr/train$b "train batch $b" "x" "y" 4; cd /tmp
done
