"""bench.py's `batched` block alone, for K images per launch (development aid; run under rocprofv3 by
tools/profile_rounds.sh so that the stored counters of profiles/traffic.json belong to the scenes -- and the tile-pass form --
bench.py reports).  usage: batched_bench_scene.py [K]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
print(json.dumps(bench.batched_rate(50000, 512, 768, torch.device("cuda:0"), ks=(k,))))
