#!/usr/bin/env python
"""Replays of the pipelined hipGraph and nothing else (bench.py: shard_leg), for tracing:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/graph_replay.py bf16 64 20
    python tools/graph_timeline.py 'OUT/**/t_kernel_trace.csv'
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if dtype == "bf16" else 8)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
print(json.dumps(bench.shard_leg(dtype, B, steps, torch.device("cuda:0"))))
