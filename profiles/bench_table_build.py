#!/usr/bin/env python3
"""Secondary measurement: building the k-mer map on the device (bbduk_build_table_device) against the
host-side C++ index build + upload + finalize, for the benchmark references and a seeded random reference.
One JSON line per case."""
import json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    from bbtools_amd import bbduk as B
    rng = np.random.default_rng(40)
    big = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20_000_000, dtype=np.uint8)].tobytes()
    cases = [("adapters hdist=1 mink=11", "ktrim=r k=23 mink=11 hdist=1 ref=adapters", None),
             ("phix hdist=1", "k=31 hdist=1 ref=phix", None),
             ("random 20 Mbase hdist=0", "k=31 hdist=0", [big]),
             ("random 1 Mbase hdist=1", "k=31 hdist=1", [big[:1_000_000]])]
    B.BBDuk("k=31 ref=phix", build="device").close()                      # warm-up: module load, first launches
    for name, args, refs in cases:
        t = time.perf_counter(); d = B.BBDuk(args, refs=refs, build="device"); t_dev = time.perf_counter() - t
        n_dev = d.gpu.table_size; d.close()
        t = time.perf_counter(); h = B.BBDuk(args, refs=refs, build="host"); t_host = time.perf_counter() - t
        n_host = h.gpu.table_size; h.close()
        print(json.dumps({"case": name, "args": args, "keys": int(n_dev), "same_key_count": bool(n_dev == n_host),
                          "device_build_s": round(t_dev, 3), "host_build_upload_finalize_s": round(t_host, 3)}))


if __name__ == "__main__":
    main()
