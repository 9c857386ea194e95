echo "== default"; python tools/bench_batching.py 7 2>&1 | grep ResBlocks
for c in 1 2 3; do echo "== N32 cfg $c"; SPEI_SLAB_CFG_N32=$c python tools/bench_batching.py 7 2>&1 | grep "lv1"; done
for c in 1 2 3 4; do echo "== N64 cfg $c"; SPEI_SLAB_CFG_N64=$c python tools/bench_batching.py 7 2>&1 | grep "lv2"; done
for c in 1 2 3 4; do echo "== N128 cfg $c"; SPEI_SLAB_CFG_N128=$c python tools/bench_batching.py 7 2>&1 | grep "lv3"; done
