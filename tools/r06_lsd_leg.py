#!/usr/bin/env python3
"""bench.py's LSD leg alone (device-resident batches of 4096 + one image per call).  python tools/r06_lsd_leg.py [B]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
r = bench.lsd_leg(0, B=int(sys.argv[1]), B2=0) if len(sys.argv) > 1 else bench.lsd_leg(0)
r.pop("note", None); r.pop("workload", None)
print(json.dumps(r))
