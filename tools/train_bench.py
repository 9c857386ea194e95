#!/usr/bin/env python3
"""Training-step throughput: a wrapper around `python bench.py --train` (the measurement and its CPU-baseline leg live in bench.py).

    python tools/train_bench.py [--model swint|speinet] [--batch 4] [--patch 200] [--steps 5] [--warmup 2] [--cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402

if __name__ == "__main__":
    bench.train_main(sys.argv[1:])
