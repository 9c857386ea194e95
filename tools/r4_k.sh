set -e
O=gpurun_out/r4k
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -k "mlp_fused_vs_oracle or mlp_tok or swin_block" > $O/tests.txt 2>&1 || true
tail -5 $O/tests.txt
timeout -k 10 200 python tools/bench_swin.py 2>&1 | grep "mlp" > $O/bench_swin.txt || true
cat $O/bench_swin.txt
python -m speinet_amd.build --tuning > $O/build.txt 2>&1
python tools/stamp_phases.py mlpold > $O/stamp_mlp.txt 2>&1 || true
cat $O/stamp_mlp.txt
