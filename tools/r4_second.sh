set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4b
mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/htrace -o h -- python3 $R/tools/harness_bench.py 60 f16 > $O/harness_traced.json 2> $O/harness_traced.err
cd $R
TRACE_COMPACT=$O/harness_trace.csv.gz python tools/harness_timeline.py /tmp/htrace > $O/harness_timeline.txt 2>&1 || true
head -60 $O/harness_timeline.txt
python -m pytest tests/test_gpu_grad.py tests/test_gpu_train.py -x -q -k "conv_forward_backward or linear_forward or resblock_backward or inplace_data_edit" > $O/tests.txt 2>&1 || true
tail -15 $O/tests.txt
