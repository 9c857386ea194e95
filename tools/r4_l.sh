set -e
O=gpurun_out/r4l
mkdir -p $O
python bench.py --cpu-baseline none --no-extras --no-harness --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || true
cut -c1-330 $O/bench.json; tail -3 $O/bench.err
python bench.py --cpu-baseline none --no-extras --no-harness --steps 10 --warmup 3 --knobs '{"attn_win4": false}' > $O/bench_old.json 2> $O/bench_old.err || true
cut -c1-330 $O/bench_old.json
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || true
tail -15 $O/tests.txt
