#!/usr/bin/env python3
"""A/B of bench.py variants in one session on one box (same clocks): prints one line per variant.
    python tools/ab_headline.py "--col-chunks 1" "--col-chunks 2" "--mode 0" ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for variant in sys.argv[1:] or [""]:
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "5", "--no-cpu-baseline"] + variant.split(),
                       capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print("%-28s GCUPS %8.1f  ms/step %.4f  sweep %.4f  traceback %.4f  materialised %.4f" % (
            variant or "(default)", d["value"], d["ms_per_step"], d["roofline"]["kernel_avg_ms"],
            d["roofline"]["traceback_avg_ms"], d["ms_per_step_materialised"]), flush=True)
    except Exception as e:      # noqa: BLE001
        print(variant, "FAILED", e, p.stderr[-400:], flush=True)
