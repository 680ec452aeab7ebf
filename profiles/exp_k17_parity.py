#!/usr/bin/env python3
"""GPU box: k=17 against a 4.6 Mbase genome -- cache-resident vs big layout vs the CPU oracle, per read."""
import json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from bbtools_amd import bbduk as B
from oracle.oracle_ffi import Oracle
K = int(sys.argv[1]) if len(sys.argv) > 1 else 17
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4_600_000
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
genome = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(G).integers(0, 4, G)].tobytes()
o = Oracle(k=K); o.add_ref(genome)
b, off = B.synth_generate_host(B.synth_params(3, contam=genome, contam_frac=0.01), 0, NP)
oa, oi, of = o.process_batch(b, off, True, nthreads=16)
for layout, hooks in (("cache", {B.HOOK_BIG_LAYOUT: -1}), ("default", None), ("big_forced_no_twin", {B.HOOK_BIG_LAYOUT: 1})):
    d = B.BBDuk("k=%d" % K, refs=[genome], build="device", hooks=hooks)
    ga, gi, gf = d.gpu.process_batch(b, off, True)
    bad = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
    ex = []
    for i in bad[:4]:
        r = b[off[i]:off[i + 1]].tobytes()
        ex.append({"read": int(i), "gpu": [int(ga[i]), int(gi[i]), int(gf[i])], "oracle": [int(oa[i]), int(oi[i]), int(of[i])], "hasN": r.count(b"N"), "len": len(r)})
    print(json.dumps({"k": K, "layout": layout, "keys": int(d.stored_kmers), "oracle_keys": int(o.stored_kmers), "reads": int(len(oa)), "oracle_hits": int((oa > 0).sum()), "differ": int(len(bad)), "examples": ex}), flush=True)
    d.close()
