set -e
O=gpurun_out/r4q
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -k "attn_fused_vs_oracle" > $O/tests.txt 2>&1 || true
tail -5 $O/tests.txt
timeout -k 10 200 python tools/bench_swin.py 2>&1 | grep "attn_" > $O/bench_swin.txt || true
cat $O/bench_swin.txt
python -m speinet_amd.build --tuning > $O/build.txt 2>&1
python tools/stamp_phases.py attn4 > $O/stamp_attn4_w8.txt 2>&1 || true
cat $O/stamp_attn4_w8.txt
