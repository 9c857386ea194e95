set -e
O=gpurun_out/r4n
mkdir -p $O
B="python bench.py --cpu-baseline none --no-extras --no-harness --steps 16 --warmup 4"
for v in "" "--free-overlap" "--inflight 3 --free-overlap" "--inflight 3" "--inflight 1"; do
  echo "== new $v"; $B $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'fps', round(d['ms_per_step'],2), 'ms; corr', round(d['roofline']['launch_ms'],2))"
done
for v in "" "--free-overlap"; do
  echo "== old $v"; $B $v --knobs '{"attn_win4": false}' 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'fps', round(d['ms_per_step'],2), 'ms; corr', round(d['roofline']['launch_ms'],2))"
done
