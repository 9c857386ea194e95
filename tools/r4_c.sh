set -e
O=gpurun_out/r4c
mkdir -p $O
python -m pytest tests/test_gpu_bf16.py -x -q -k "mlp_fused_vs_oracle" > $O/tests.txt 2>&1 || true
tail -8 $O/tests.txt
python tools/bench_swin.py > $O/bench_swin.txt 2>&1 || true
cat $O/bench_swin.txt
python -m pytest tests/test_gpu_grad.py tests/test_gpu_train.py -x -q -k "conv_forward_backward or linear_forward or resblock_backward or inplace_data_edit" > $O/tests2.txt 2>&1 || true
tail -8 $O/tests2.txt
