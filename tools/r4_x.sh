set -e
O=gpurun_out/r4x
mkdir -p $O
for cfg in "100 3" "100 2" "100 4" "100 1"; do set -- $cfg; python - $1 $2 <<'PY' 2>/dev/null
import sys, json, torch
sys.path.insert(0, '.')
from speinet_amd import inference
n, l = int(sys.argv[1]), sys.argv[2]
r = inference.harness_throughput(n, "f16", extra_args=["--lanes", l])
ms = torch.cuda.memory_stats()
print(f"frames {n} lanes {l}: {r['value']:.2f} fps; {r['timing'][0]}; reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB, alloc retries {ms.get('num_alloc_retries')}, cudaMalloc calls {ms.get('num_device_alloc')}")
PY
done
