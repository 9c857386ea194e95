set -e
O=gpurun_out/r4d
mkdir -p $O
python -m pytest tests/test_gpu_bf16.py -x -q -k "mlp_fused_vs_oracle" > $O/tests.txt 2>&1 || true
tail -5 $O/tests.txt
python tools/bench_swin.py 2>&1 | grep mlp > $O/bench_swin.txt || true
cat $O/bench_swin.txt
python -m speinet_amd.build --tuning > $O/build.txt 2>&1
python tools/stamp_phases.py mlp > $O/stamp_mlp.txt 2>&1 || true
python tools/stamp_phases.py mlpold > $O/stamp_mlpold.txt 2>&1 || true
cat $O/stamp_mlp.txt $O/stamp_mlpold.txt
