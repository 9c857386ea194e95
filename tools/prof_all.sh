# Round-end profiling pass (run on the GPU box: `gpurun -- bash tools/prof_all.sh <commit>`): kernel trace of the bench, the three
# PMC passes (separate runs, as MI355X_MICROARCH.md prescribes), kernel trace of the training step.  The raw traces exceed what
# gpurun copies back, so the summaries are produced here; copy gpurun_out/r02b/* into profiles/ afterwards.
set -e
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r02b
HEAD=${1:-n/a}
rm -rf $O /tmp/r02b; mkdir -p $O /tmp/r02b
T=/tmp/r02b
cd /tmp
BENCH="python3 $R/bench.py --cpu-baseline none --no-harness"
echo "== kernel trace, bench"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/bench_trace -o b -- $BENCH --steps 10 --warmup 3 > $O/r02_f16_bench_traced.json 2> $O/bench_trace.err
echo "== pmc FETCH_SIZE"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/pmc_fetch -o f -- $BENCH --steps 3 --warmup 1 --no-graph --streams 1 > /dev/null 2> $O/pmc_fetch.err
echo "== pmc WRITE_SIZE"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $T/pmc_write -o w -- $BENCH --steps 3 --warmup 1 --no-graph --streams 1 > /dev/null 2> $O/pmc_write.err
echo "== pmc SQ"; timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $T/pmc_sq -o s -- $BENCH --steps 3 --warmup 1 --no-graph --streams 1 > /dev/null 2> $O/pmc_sq.err
echo "== kernel trace, training step"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/train_trace -o t -- python3 $R/bench.py --train --batch 8 --steps 3 --warmup 1 > $O/r02_train_bench_traced.json 2> $O/train_trace.err
cd $R
python3 tools/stats_md.py $T/bench_trace $O/r02_f16_kernel_stats "Round 2 — rocprofv3 --kernel-trace --stats, f16 / top2 (default bench), 720p _forwardbs, 2 HIP streams, 2 graph segments per frame" "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --cpu-baseline none --no-harness --steps 10 --warmup 3" "bench line of the same run (profiler attached): profiles/r02_f16_bench_traced.json; the tracked kernel's average below must agree with its roofline.launch_ms.  Commit $HEAD." 14
python3 tools/stats_md.py $T/train_trace $O/r02_train_kernel_stats "Round 2 — rocprofv3 --kernel-trace --stats, training step of the swint model (f32, batch 8 of 200x200 crops, n_sequence 3): forward in train() mode, 1*L1+2*HEM, backward, Adam" "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --train --batch 8 --steps 3 --warmup 1" "4 steps in the trace (1 warm-up + 3 timed); bench line of the same run: profiles/r02_train_bench_traced.json.  Commit $HEAD." 4
python3 tools/pmc_traffic.py $T/pmc_fetch $T/pmc_write $O/r02_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --cpu-baseline none --no-harness --steps 3 --warmup 1 --no-graph --streams 1" "f16/top2" $HEAD
python3 tools/pmc_summary.py $T/pmc_sq > $O/r02_pmc_sq_kernels.body.md
python3 tools/pmc_summary.py $T/pmc_fetch $T/pmc_write > $O/r02_pmc_hbm_kernels.body.md
ls -la $O
