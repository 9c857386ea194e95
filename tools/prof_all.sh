# Round-end profiling pass (run on the GPU box: `gpurun -- bash tools/prof_all.sh <commit>`): kernel traces of the bench (the default
# command, and one frame in flight on one stream for exclusive per-kernel times), the three PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes; --kernel-trace only beside them), kernel trace of the training step.  The raw traces exceed what gpurun
# copies back, so the summaries are produced here; copy gpurun_out/r04/* into profiles/ afterwards.
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04
HEAD=${1:-n/a}
rm -rf $O /tmp/r04; mkdir -p $O /tmp/r04
T=/tmp/r04
cd /tmp
BENCH="python3 $R/bench.py --cpu-baseline none --no-harness --no-extras --no-families"
echo "== kernel trace, default bench"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/bench_trace -o b -- $BENCH --steps 10 --warmup 3 > $O/r04_f16_bench_traced.json 2> $O/bench_trace.err
echo "== kernel trace, one frame in flight, one stream"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/excl_trace -o e -- $BENCH --steps 10 --warmup 3 --inflight 1 --streams 1 > $O/r04_f16_bench_exclusive_traced.json 2> $O/excl_trace.err
PM="$BENCH --steps 3 --warmup 1 --inflight 1 --streams 1 --no-graph"
echo "== pmc FETCH_SIZE"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/pmc_fetch -o f -- $PM > /dev/null 2> $O/pmc_fetch.err
echo "== pmc WRITE_SIZE"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $T/pmc_write -o w -- $PM > /dev/null 2> $O/pmc_write.err
echo "== pmc SQ"; timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $T/pmc_sq -o s -- $PM > /dev/null 2> $O/pmc_sq.err
echo "== kernel trace, training step"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/train_trace -o t -- python3 $R/bench.py --train --batch 8 --steps 3 --warmup 1 > $O/r04_train_bench_traced.json 2> $O/train_trace.err
echo "== kernel trace, harness (PNG in -> PNG out, 60-frame 720p clip, cross-window reuse)"; timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $T/harness_trace -o h -- python3 $R/tools/harness_bench.py 60 f16 > $O/r04_harness_bench_traced.json 2> $O/harness_trace.err
cd $R
python3 tools/harness_timeline.py $T/harness_trace > $O/r04_harness_timeline.txt 2>&1 || true
python3 tools/stats_md.py $T/bench_trace $O/r04_f16_kernel_stats "Round 4 — rocprofv3 --kernel-trace --stats, f16 / top2 (default bench): 720p _forwardbs, encoder passes batched per layer, 2 HIP streams, 2 frames in flight" "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --cpu-baseline none --no-harness --no-extras --steps 10 --warmup 3" "bench line of the same run (profiler attached): profiles/r04_f16_bench_traced.json; the correlation kernel's average below must agree with its roofline.launch_ms.  Kernels of two frames and two streams overlap in this run: per-kernel averages include the time a kernel shares the chip (exclusive times: r04_kernel_table.md).  Commit $HEAD." 14
python3 tools/stats_md.py $T/train_trace $O/r04_train_kernel_stats "Round 4 — rocprofv3 --kernel-trace --stats, training step of the swint model (batch 8 of 200x200 crops, n_sequence 3): forward in train() mode, 1*L1+2*HEM, backward, Adam" "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --train --batch 8 --steps 3 --warmup 1" "4 steps in the trace (1 warm-up + 3 timed); bench line of the same run: profiles/r04_train_bench_traced.json.  Commit $HEAD." 4
python3 tools/pmc_traffic.py $T/pmc_fetch $T/pmc_write $O/r04_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --cpu-baseline none --no-harness --no-extras --steps 3 --warmup 1 --inflight 1 --streams 1 --no-graph" "f16/top2" $HEAD
(echo "# Round 4 — per-kernel roofline table, f16 / top2 default arithmetic, 720p _forwardbs (commit $HEAD)"; echo; echo "Durations: rocprofv3 --kernel-trace of \`bench.py --inflight 1 --streams 1\` (exclusive per-kernel times; hipGraph replay; frames in the trace = launches of the correlation kernel).  Bytes and SQ counters: three separate --pmc passes of \`bench.py --inflight 1 --streams 1 --no-graph --steps 3 --warmup 1\` (4 frames each; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Top 24 kernels by time."; echo; python3 tools/kernel_table.py $T/excl_trace $T/pmc_fetch $T/pmc_write $T/pmc_sq 24) > $O/r04_kernel_table.md
python3 tools/pmc_summary.py --top=30 $T/pmc_sq > $O/r04_pmc_sq_kernels.body.md
python3 tools/pmc_summary.py --top=30 $T/pmc_fetch $T/pmc_write > $O/r04_pmc_hbm_kernels.body.md
python3 tools/frame_timeline.py $T/excl_trace > $O/r04_frame_timeline.txt 2>&1 || true
python3 tools/frame_timeline.py $T/bench_trace > $O/r04_frame_timeline_default.txt 2>&1 || true
ls -la $O
