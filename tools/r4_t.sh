set -e
B="python bench.py --cpu-baseline none --no-extras --no-harness --no-families --steps 20 --warmup 4"
run() { echo "== $1"; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'fps', round(d['ms_per_step'],2))"; }
run "default (glue+dec2 split)" $B
run "glue1 split too (round 3)" $B --knobs '{"stage": {"glue1": {"precision": "bf16x3", "commute_any": true}}}'
run "no split" $B --knobs '{"split_decode": false}'
run "default again" $B
run "default, 3 lanes free" $B --inflight 3 --free-overlap
run "default, 2 lanes free" $B --free-overlap
