set -e
B="python bench.py --cpu-baseline none --no-extras --no-harness --no-families --steps 24 --warmup 6"
run() { echo "== $1"; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'fps', round(d['ms_per_step'],2), 'corr', round(d['roofline']['correlation_kernel']['launch_ms'],2))"; }
run "2 lanes corr_end" $B
run "3 lanes none" $B --inflight 3 --gate none
run "4 lanes none" $B --inflight 4 --gate none
run "2 lanes corr_start" $B --inflight 2 --gate corr_start
run "3 lanes corr_start" $B --inflight 3 --gate corr_start
run "3 lanes corr_end" $B --inflight 3 --gate corr_end
run "3 lanes none, 1 stream" $B --inflight 3 --gate none --streams 1
run "3 lanes none again" $B --inflight 3 --gate none
