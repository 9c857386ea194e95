set -e
O=gpurun_out/r4s
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py -x -q -k "forward or golden or harness or 720p or window" > $O/tests.txt 2>&1 || true
grep -i "dPSNR\|passed\|failed\|Error" $O/tests.txt | tail -40
python bench.py --cpu-baseline none --no-extras --no-harness --no-families --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'fps')"
