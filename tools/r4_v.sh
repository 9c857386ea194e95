set -e
O=gpurun_out/r4v
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_harness.py -x -q > $O/tests.txt 2>&1 || true
tail -8 $O/tests.txt
python tools/harness_bench.py 100 f16 > $O/harness.json 2> $O/harness.err || true
grep -A3 '"value"' $O/harness.json | head -6; tail -3 $O/harness.err
python bench.py --cpu-baseline none --no-extras --no-harness --no-families --steps 24 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'fps bench')"
