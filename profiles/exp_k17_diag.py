#!/usr/bin/env python3
"""GPU box: where do the big layout's k=17 misses come from -- the map (table_lookup of the read's own windows) or the candidate scan
(the tile kernel's exact per-key path on the same map)?"""
import json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import numpy as np
from bbtools_amd import bbduk as B
from oracle.oracle_ffi import Oracle, pack_reads
from bench_c4_full import keys_of
K = int(sys.argv[1]) if len(sys.argv) > 1 else 17
G, NP = 4_600_000, 1_000_000
genome = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(G).integers(0, 4, G)].tobytes()
o = Oracle(k=K); o.add_ref(genome)
b, off = B.synth_generate_host(B.synth_params(3, contam=genome, contam_frac=0.01), 0, NP)
oa, oi, of = o.process_batch(b, off, True, nthreads=16)
d = B.BBDuk("k=%d" % K, refs=[genome], build="device", hooks={B.HOOK_BIG_LAYOUT: 1})
ga, gi, gf = d.gpu.process_batch(b, off, True)
bad = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
print(json.dumps({"differ": int(len(bad)), "spilled": int(d.gpu.table_spilled), "line_hist_tail": d.gpu.line_histogram()[24:].tolist()}))
pairs = sorted({int(i) & ~1 for i in bad})
for p in pairs[:12]:
    info = {"pair": p}
    for r in (p, p + 1):
        rd = b[off[r]:off[r + 1]]
        W = np.lib.stride_tricks.sliding_window_view(rd, K)
        okw = np.isin(W, np.frombuffer(b"ACGT", np.uint8)).all(axis=1)
        ks = keys_of(W[okw], k=K, mm=True)
        got = d.gpu.table_lookup(ks)
        hitpos = np.nonzero(okw)[0][got > 0]
        info["r%d" % (r - p)] = {"gpu": [int(ga[r]), int(gi[r]), int(gf[r])], "oracle": [int(oa[r]), int(oi[r]), int(of[r])], "windows_in_map": hitpos.tolist()[:6], "n_in_map": int((got > 0).sum())}
    # the same pair alone, and the pair through the tile kernel's exact path
    sub = [b[off[p]:off[p + 1]].tobytes(), b[off[p + 1]:off[p + 2]].tobytes()]
    bb, oo = pack_reads(sub)
    a1, i1, f1 = d.gpu.process_batch(bb, oo, True)
    info["alone"] = [a1.tolist(), f1.tolist()]
    print(json.dumps(info), flush=True)
d.close()
d = B.BBDuk("k=%d" % K, refs=[genome], build="device", hooks={B.HOOK_BIG_LAYOUT: 1, B.HOOK_FORCE_TILE: 1})
ga2, gi2, gf2 = d.gpu.process_batch(b[:off[400000]], off[:400001], True)
m = 400000
print(json.dumps({"tile_exact_path_differ_first_400k": int(((ga2 != oa[:m]) | (gi2 != oi[:m]) | (gf2 != of[:m])).sum()), "wave_differ_first_400k": int(((ga[:m] != oa[:m]) | (gf[:m] != of[:m])).sum())}))
d.close()
