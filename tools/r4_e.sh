set -e
O=gpurun_out/r4e
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -x -q -k "attn_fused_vs_oracle" > $O/tests.txt 2>&1 || true
tail -12 $O/tests.txt
timeout -k 10 200 python tools/bench_swin.py 2>&1 | grep -v amdgpu.ids > $O/bench_swin.txt || true
cat $O/bench_swin.txt
