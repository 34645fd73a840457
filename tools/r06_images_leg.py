#!/usr/bin/env python3
"""bench.py's images -> poses leg alone.  python tools/r06_images_leg.py [streams] [lines 0/1] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lines = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
r = bench.images_leg(0, B=B, steps=steps, lines=lines)
r.pop("workload", None)
print(json.dumps(r))
