#!/usr/bin/env python3
"""bench.py --workload tim_vit_small_mask with the nt tile configuration forced (argument: -1 automatic, 0 = 128 x 128,
1 = 256 x 128, 2 = 256 x 256, 3 = 256 x 256 phased)."""
import os
import runpy
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from visiontransformer_amd._lib import lib  # noqa: E402

lib.vitmi_debug_set_nt_config(int(sys.argv[1]))
sys.argv = ["bench.py", "--workload", "tim_vit_small_mask", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
