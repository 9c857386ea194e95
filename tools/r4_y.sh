set -e
run() { python - "$@" <<'PY' 2>/dev/null | grep "^frames"
import sys, json, torch, os
sys.path.insert(0, '.')
from speinet_amd import inference
args = sys.argv[1:]
r = inference.harness_throughput(100, "f16", extra_args=args)
print(f"frames 100 {' '.join(args)} HWQ={os.environ.get('GPU_MAX_HW_QUEUES')}: {r['value']:.2f} fps; {r['timing'][0]}")
PY
}
run --lanes 2 --streams 1
run --lanes 3 --streams 1
GPU_MAX_HW_QUEUES=8 run --lanes 3 --streams 1
GPU_MAX_HW_QUEUES=8 run --lanes 3 --streams 2
GPU_MAX_HW_QUEUES=8 run --lanes 2 --streams 2
GPU_MAX_HW_QUEUES=8 run --lanes 4 --streams 1
