import random, sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from bbtools_amd import bbduk as B
from oracle.oracle_ffi import Oracle, pack_reads
from tests import util
from tests.test_gpu_parity import QX
ci = 8
args, okw = QX[ci]
rng = random.Random(51000 + ci)
base = util.rand_seq(rng, 300)
refs = [base, util.rand_seq(rng, 200) + base[50:120], util.revcomp(base[150:260]), util.rand_seq(rng, 90, 0.02), util.rand_seq(rng, 14), util.rand_seq(rng, 400)]
twin = bytearray(refs[5][100:160]); twin[30] = ord("A") if twin[30] != ord("A") else ord("C"); refs.append(bytes(twin))
d = B.BBDuk(args, refs=refs, build="host")
o = Oracle(**okw)
for r in refs: o.add_ref(r)
reads = util.fuzz_reads(rng, refs, 600, [0, 1, 7, 8, 9, 17, 19, 20, 21, 22, 25, 30, 40, 75, 150, 151, 300], junk=True)
for i in range(500):
    n = rng.choice([60, 100, 150, 151])
    r = bytearray(util.rand_seq(rng, n))
    src = refs[rng.choice([0, 1, 2, 5, 6])]; cut = rng.randint(8, 50); a = rng.randrange(len(src) - cut); piece = bytearray(src[a:a + cut])
    if rng.random() < 0.5: piece = bytearray(util.revcomp(bytes(piece)))
    for _ in range(rng.choice([0, 1, 1, 2, 3])): piece[rng.randrange(len(piece))] = rng.choice(b"ACGTNN.")
    w = rng.choice([0, n - cut, rng.randrange(n - cut + 1)]); r[w:w + cut] = piece
    for _ in range(rng.choice([0, 0, 1, 2])): r[rng.randrange(n)] = ord("N")
    reads.append(bytes(r))
if len(reads) % 2: reads.pop()
for paired in (True, False):
    b, off = pack_reads(reads)
    gr, gl, gi, gf = d.gpu.ktrimtips_batch(b, off, paired)
    xr, xl, oi, of = o.process_batch_tips(b, off, paired, nthreads=4)
    bad = np.nonzero((gr != xr) | (gl != xl) | (gi != oi) | (gf != of))[0]
    print("paired", paired, "bad", bad.tolist()[:10])
    for i in bad[:4]:
        print(i, reads[i], "gpu", (int(gr[i]), int(gl[i]), int(gi[i]), int(gf[i])), "oracle", (int(xr[i]), int(xl[i]), int(oi[i]), int(of[i])))
        # alone, and through the tiled kernel
        for tile in (0, 1):
            d.gpu.test_hook(B.HOOK_FORCE_TILE, tile)
            pr = [reads[i & ~1], reads[i | 1]] if paired else [reads[i]]
            bb, oo = pack_reads(pr)
            print("  alone tile=%d" % tile, [x.tolist() for x in d.gpu.ktrimtips_batch(bb, oo, paired)])
        d.gpu.test_hook(B.HOOK_FORCE_TILE, 0)
print("---- variants of read 1027")
r0 = reads[1027]
for name, rd in [("orig", r0), ("first 40", r0[:40]), ("first 33", r0[:33]), ("first 32", r0[:32]), ("N20->A", r0[:20] + b"A" + r0[21:]), ("N32->A", r0[:32] + b"A" + r0[33:]), ("N7->A", r0[:7] + b"A" + r0[8:])]:
    bb, oo = pack_reads([rd])
    res = {}
    for tile in (0, 1):
        d.gpu.test_hook(B.HOOK_FORCE_TILE, tile)
        res[tile] = [x.tolist() for x in d.gpu.ktrimtips_batch(bb, oo, False)]
    d.gpu.test_hook(B.HOOK_FORCE_TILE, 0)
    print(name, "wave", res[0], "tile", res[1], "oracle", [x.tolist() for x in o.process_batch_tips(bb, oo, False, nthreads=1)])
