#!/usr/bin/env python3
"""bench.py's end_to_end block alone (host-buffer operators through the C ABI: pageable and pinned buffers, one and two submitting threads).  One JSON line."""
import json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench
from bbtools_amd import bbduk as B
duk = B.BBDuk("ktrim=r k=23 mink=11 hdist=1 ref=adapters")
print(json.dumps(bench.measure_end_to_end(B, duk, int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 2)))
