#!/usr/bin/env python3
"""
Randomised parity soak (GPU box): for --seconds S keeps drawing flag combinations (every mode served: ktrim r/l/rl, ktrim=n,
ksplit, kfilter with mbk/mkf/mcf, k>31, findbestmatch; host- or device-built map; rskip), references and reads (lengths
0..6000, junk bytes, Ns, chimeras) and compares the device operators -- ASCII and packed boundary -- with the CPU oracle,
per read and counter by counter.  Prints one JSON line; exits non-zero at the first mismatch (config and seed in the message).
"""
import argparse, json, os, random, sys, time, traceback
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def ingest_roundtrip(B, d, o, okw, reads, rng):
    import numpy as np, torch
    from oracle import fastq_oracle as FQ
    from oracle.oracle_ffi import pack_reads
    eol = b"\r\n" if rng.random() < 0.3 else b"\n"
    recs = []
    for i, r in enumerate(reads):
        r = bytes(c if c not in (10, 13) else 78 for c in r)
        q = bytes(33 + rng.randrange(41) for _ in range(len(r)))
        recs.append(eol.join([b"@r%d x" % i, r, b"+", q]))
    final = eol if (rng.random() < 0.7 or not reads[-1]) else b""     # an empty last quality line only exists with its terminator
    text = np.frombuffer(eol.join(recs) + final, np.uint8).copy()
    dt = torch.from_numpy(np.concatenate([text, np.zeros(16, np.uint8)])).cuda()[:len(text)]
    fb = B.fastq_ingest_device(dt, None, True, len(reads))
    want = FQ.ingest([text], True, len(reads))
    n = fb.n
    assert n == want["n"] == len(reads) and np.array_equal(fb.offsets[:n + 1].cpu().numpy(), want["offsets"])
    b, off = pack_reads([bytes(want["bases"][want["offsets"][i]:want["offsets"][i + 1]]) for i in range(n)])
    o.reset_counters(); oa, oi, of = o.process_batch(b, off, True, nthreads=4)
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_i = torch.empty_like(d_a); d_f = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(d.gpu.counters_len, dtype=torch.int64, device="cuda")
    d.gpu.process_batch_packed_device(fb.codes, fb.undef, fb.offsets[:n + 1].contiguous(), fb.total_bases, True, d_a, d_i, d_f, d_c)
    torch.cuda.synchronize()
    assert np.array_equal(d_a.cpu().numpy(), oa) and np.array_equal(d_i.cpu().numpy(), oi) and np.array_equal(d_f.cpu().numpy(), of), "ingest->operator"
    assert np.array_equal(d_c.cpu().numpy(), o.counters())
    d_out = torch.empty(len(text) + 64, dtype=torch.uint8, device="cuda")
    left = d_a if okw.get("ktrimLeft") else None; right = d_a if okw.get("ktrimRight") else None
    z = np.zeros(n, np.int32)
    for rem in (False, True):
        nb = B.fastq_write_device(dt, fb, left, right, d_f, rem, d_out)
        exp = FQ.write_records([text], want["lines"], n, oa if left is not None else z, oa if right is not None else z, of, rem)
        assert bytes(d_out[:nb].cpu().numpy()) == exp, "writer"


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=300.0); ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=None, help="re-run the one configuration a FAILED line names by its seed")
    ap.add_argument("--families", default=None, help="comma-separated subset of the families to draw from (e.g. tails,base,kmask,ksplit)")
    a = ap.parse_args()
    from bbtools_amd import bbduk as B
    from oracle.oracle_ffi import Oracle
    from tests import util
    from tests import test_gpu_parity as T
    t0 = time.time(); n_cfg = 0; n_reads = 0; fam_count = {}
    it = 0
    while time.time() - t0 < a.seconds and not (a.only is not None and it):
        seed = a.seed * 1_000_003 + it if a.only is None else a.only; it += 1
        rng = random.Random(seed)
        fam = rng.choice(a.families.split(",") if a.families else ["base", "base", "base", "tails", "kmask", "tips", "ksplit", "big", "fbm", "seal", "tf1bp", "bigmap", "bigmap", "seedmap"])
        bigmap = fam == "bigmap"                               # base configurations with the HBM-resident map layout asked for (round 4: bbduk_bigs_kernel for
        if bigmap: fam = "base"                                # kfilter and ktrim=r, the 52-bit variant's pair scan, the exact scans for the rest)
        # round 5: every other operator family over a big-layout map too (bbduk_bigs_every_kernel's facts, the GENERAL flags, k > 31: bbduk_bigs_kbig.hip)
        if fam in ("kmask", "tips", "ksplit", "big", "fbm") and rng.random() < 0.35: bigmap = True
        tails = fam == "tails"                                 # base configurations on reads that overshoot the pair scan's blocks by 1..32 positions:
        if tails: fam = "base"                                 # (block borders of the stream scan / the every-hit scans' tail pass)
        if fam == "tf1bp":                                     # trimfailuresto1bp over every operator family (units within the main kernel's planes)
            try:
                tfam, targs, tokw = T.tf1bp_config(rng)
                trefs = [util.rand_seq(rng, rng.randint(12, 90), 0.01) for _ in range(rng.randint(1, 8))]
                td, to = T.make_pair(targs, tokw, refs=trefs)
                treads = util.fuzz_reads(rng, trefs, rng.choice([60, 400, 1500]), [0, 1, 2, 3, 9, 10, 12, 22, 23, 31, 40, 64, 75, 150, 151, 300, 1000], junk=True)
                if rng.random() < 0.4:                             # units beyond a wave's planes / beyond the tiles: the tiled and long-unit kernels' record stages (round 5)
                    for _ in range(rng.randint(1, 3)):
                        L = rng.choice([3000, 5000, 45000]); g = bytearray(util.rand_seq(rng, L, 0.0005))
                        r = rng.choice(trefs); pos = rng.choice([0, L - len(r), rng.randrange(L - len(r))]); g[pos:pos + len(r)] = r
                        treads.insert(rng.randrange(len(treads) + 1), bytes(g))
                if len(treads) % 2: treads.append(b"")
                if tfam == "n": T.check_kmask(td, to, treads, True); T.check_kmask(td, to, treads[:101], False)
                elif tfam == "rl": T.check_tips(td, to, treads, True); T.check_tips(td, to, treads[:101], False)
                elif tfam == "ksplit": T.check_split(td, to, treads)
                else: T.check_batch(td, to, treads, True); T.check_batch(td, to, treads[:101], False)
                td.close()
            except Exception as e:
                print(json.dumps({"FAILED": True, "seed": seed, "family": fam, "args": locals().get("targs"), "error": (str(e) or traceback.format_exc())[-2000:]}))
                sys.exit(1)
            n_cfg += 1; n_reads += len(treads); fam_count[fam] = fam_count.get(fam, 0) + 1
            continue
        if fam == "seal":                                      # Seal's k-mer path (include/seal_gpu.h) against oracle/seal_oracle.c
            from tests import test_seal
            try:
                test_seal.run_case(seed, nreads=rng.choice([60, 300, 1200]))
            except Exception as e:
                print(json.dumps({"FAILED": True, "seed": seed, "family": fam, "error": (str(e) or traceback.format_exc())[-2000:]}))
                sys.exit(1)
            n_cfg += 1; n_reads += 300; fam_count[fam] = fam_count.get(fam, 0) + 1
            continue
        args, okw = T.random_config(rng)
        seedfam = fam == "seedmap"                             # round 5: the seed layout with forbidn / maxbadkmers / mm=f, device-built, forced at any size
        if seedfam:
            fam = "base"
            k = rng.choice([16, 17, 20, 21, 23, 24, 25, 27, 31]); okw, ar = dict(k=k, hdist=1), ["k=%d" % k, "hdist=1"]
            if rng.random() < 0.3: okw["maskMiddle"] = 0; ar.append("mm=f")
            if rng.random() < 0.4: okw["forbidN"] = 1; ar.append("forbidn=t")
            if rng.random() < 0.4: v = rng.randint(1, 4); okw["maxBadKmers0"] = v; ar.append("mbk=%d" % v)
            if rng.random() < 0.2: okw["requireBothBad"] = 1; ar.append("rieb=f")
            if rng.random() < 0.3: v = rng.choice([1, 20, 40]); okw["minReadLength"] = v; ar.append("minlen=%d" % v)
            args = " ".join(ar)
        toks = [t for t in args.split() if not t.startswith("ktrim=") and t != "tpe"]
        for key in ("ktrimRight", "ktrimLeft", "trimPairsEvenly"):
            if fam != "base":
                okw.pop(key, None)
        if fam == "base":
            toks = args.split()
        elif fam == "kmask":
            toks.append("ktrim=n"); okw["ktrimN"] = 1
            if rng.random() < 0.4: toks.append("mfc=t"); okw["kmaskFullyCovered"] = 1
            for kk in ("maxBadKmers0", "minKmerFraction", "minCoveredFraction", "ktrimExclusive"):
                okw.pop(kk, None)
            toks = [t for t in toks if not t.startswith(("mbk=", "mkf=", "mcf=", "ktrimexclusive"))]
        elif fam == "tips":
            toks.append("ktrim=rl"); okw["ktrimRight"] = okw["ktrimLeft"] = 1
            for kk in ("maxBadKmers0", "minKmerFraction", "minCoveredFraction"):
                okw.pop(kk, None)
            toks = [t for t in toks if not t.startswith(("mbk=", "mkf=", "mcf="))]
            if rng.random() < 0.4: toks.append("tpe"); okw["trimPairsEvenly"] = 1
        elif fam == "ksplit":
            toks.append("ksplit=t"); okw["ksplit"] = 1
            for kk in ("maxBadKmers0", "minKmerFraction", "minCoveredFraction", "ktrimExclusive", "skipR1"):
                okw.pop(kk, None)
            toks = [t for t in toks if not t.startswith(("mbk=", "mkf=", "mcf=", "ktrimexclusive", "skipr1"))]
            if okw.get("trimPad", 0) > 0:
                okw["trimPad"] = -okw["trimPad"]; toks = [("tp=%d" % okw["trimPad"]) if t.startswith("tp=") else t for t in toks]
        elif fam in ("big", "fbm"):
            for kk in ("mink", "hdist2", "trimPad", "ktrimExclusive", "qSkip", "speed") + (("minCoveredFraction",) if fam == "fbm" else ()):
                okw.pop(kk, None)                                # k>31 keeps mcf: countCoveredBases then runs on the 31-mers
            toks = [t for t in toks if not t.startswith(("mink=", "hdist2=", "tp=", "ktrimexclusive", "qskip=", "speed=") + (("mcf=",) if fam == "fbm" else ()))]
            if fam == "big":
                kb = rng.randint(32, 60); okw["k"] = 31; okw["kbig"] = kb
                okw.pop("maskMiddle", None); okw.pop("qhdist", None)
                toks = [("k=%d" % kb) if t.startswith("k=") else t for t in toks if not t.startswith(("mm=", "qhdist="))]
                if okw.get("hdist", 0) > 1: okw["hdist"] = 1; toks = [t if not t.startswith("hdist=") else "hdist=1" for t in toks]
            else:
                okw["findBestMatch"] = 1; toks.append("fbm=t")
                okw.pop("maxBadKmers0", None); okw.pop("minKmerFraction", None)
                toks = [t for t in toks if not t.startswith(("mbk=", "mkf="))]
        if "mink" in okw and fam in ("base",) and not (okw.get("ktrimRight") or okw.get("ktrimLeft")):
            okw.pop("mink"); toks = [t for t in toks if not t.startswith("mink=")]
        if rng.random() < 0.15 and fam not in ("big",) and not seedfam:
            sk = rng.randint(2, 4); okw["minSkip"] = okw["maxSkip"] = sk; toks.append("rskip=%d" % sk)
        args = " ".join(toks)
        k = okw["k"]
        refs = [util.rand_seq(rng, rng.randint(10, 260 if fam in ("big", "fbm", "ksplit") else 90), 0.01) for _ in range(rng.randint(1, 10))]
        lengths = [0, 1, 5, 10, 22, 23, 31, 40, 64, 75, 150, 151, 300, 1000] + ([2500, 6000] if rng.random() < 0.3 else [])
        if tails:
            lengths = [k - 1 + 128 * rng.choice([1, 1, 1, 2]) + rng.randint(1, 32) for _ in range(12)] + [150, 100, k - 1 + 128, k + 160, 0, 30]
        reads = util.fuzz_reads(rng, refs, rng.choice([50, 400, 1500]), lengths, junk=True)
        if tails:                                               # reference pieces that end in the tail zone or right in front of it
            for _ in range(60):
                L = rng.choice(lengths[:12]); r = rng.choice(refs); g = bytearray(util.rand_seq(rng, L, 0.002))
                end = rng.choice([L, L - 1, L - rng.randint(0, 34), k - 1 + 128 + rng.randint(-2, 3)])
                piece = r[:rng.randint(min(len(r), 8), len(r))]
                if len(piece) <= end <= L: g[end - len(piece):end] = piece
                reads.insert(rng.randrange(len(reads) + 1), bytes(g))
        if fam in ("big", "fbm", "ksplit"):
            reads += util.long_refs_and_reads(rng)[1][:0]
            reads += [util.rand_seq(rng, rng.randint(0, 60)) + rng.choice(refs)[:rng.randint(20, 260)] + util.rand_seq(rng, rng.randint(0, 60)) for _ in range(100)]
        uni = 0
        if rng.random() < 0.3:
            # every read the same length (round 6: the read-aligned forms of the stream scans -- one slot pair per read where L > 128 and L - k + 1 <= 128,
            # two reads per block of the big layout's scan where L > 120 and L - k + W <= 128 -- and the lengths right beside their limits): reads are cut
            # or padded at either end, so reference pieces, Ns and junk land anywhere, the first and the last window included
            uni = rng.choice([150, 150, 151, 151, 152, 129, 130, k + 127, k + 126, k + 128, 121, 125, 140, 200, 250, 100, 36, 50, 75, 40, 64, k, k + 1, k + 6, 76])
            made = []
            for r in reads:
                if len(r) >= uni:
                    o0 = rng.choice([0, len(r) - uni, rng.randrange(len(r) - uni + 1)]); r = r[o0:o0 + uni]
                else:
                    padl = rng.choice([0, uni - len(r), rng.randrange(uni - len(r) + 1)])
                    r = util.rand_seq(rng, padl, 0.004) + r + util.rand_seq(rng, uni - len(r) - padl, 0.004)
                made.append(bytes(r))
            reads = made
        if rng.random() < (0.2 if not uni else 0.08):           # a few reads beyond the LDS tiles: the chunked long-read kernels take over
            for _ in range(rng.randint(1, 4)):
                L = rng.randint(41000, 110000)
                g = bytearray(util.rand_seq(rng, L, 0.0005))
                for _ in range(rng.randint(0, 3)):
                    r = rng.choice(refs); pos = rng.choice([0, L - len(r), rng.randrange(L - len(r)), 2432 * rng.randint(1, 15) - rng.randint(0, 40)])
                    g[pos:pos + len(r)] = r
                reads.insert(rng.randrange(len(reads) + 1), bytes(g))
        if len(reads) % 2: reads.append(b"")
        build = "device" if ((rng.random() < 0.4 or seedfam) and okw.get("hdist", 0) <= 2 and "minSkip" not in okw) else "host"
        try:
            o = Oracle(**okw)
            for r in refs: o.add_ref(r)
            hooks = {}
            squeeze = ""
            if build == "host" and rng.random() < 0.35 and o.stored_kmers > 64:
                # crowd the map (about 3.4 keys per 4-way bucket: long overflow chains, continuation flags everywhere) and / or
                # shrink or drop the LDS presence filter: the rarely taken lookup paths
                import math
                bb = max(4, math.ceil(math.log2(o.stored_kmers / 3.4)))
                if rng.random() < 0.7: hooks[B.HOOK_BUCKET_BITS] = bb; squeeze += " bucketbits=%d" % bb
                if rng.random() < 0.6: lb = rng.choice([0, 10, 12]); hooks[B.HOOK_LDS_BITS] = lb; squeeze += " ldsbits=%d" % lb
            seedmap = (not bigmap and fam == "base" and build == "device" and okw.get("hdist", 0) == 1 and not (okw.get("ktrimRight") or okw.get("ktrimLeft")) and (rng.random() < 0.5 or seedfam))
            if seedmap:                                         # the seed layout (parents under their halves) where the library has it: forbidn, mbk > 0 too
                hooks = {B.HOOK_SEED_LAYOUT: 1}; squeeze = " seedmap"
            if bigmap:
                hooks = {B.HOOK_BIG_LAYOUT: rng.choice([1, 1, 1, 2, 3, 3]) if fam == "base" else rng.choice([1, 1, 3]), B.HOOK_BIG_LOAD: rng.choice([0, 0, 30, 60, 95])}
                if rng.random() < 0.5: hooks[B.HOOK_LDS_BITS] = rng.choice([4, 6, 7, 8, 9, 10, 12])      # (m of the minimizer lines)
                squeeze = " bigmap=%r" % (sorted(hooks.items()),)
            if os.environ.get("SOAK_DUMP"):
                import pickle; pickle.dump(dict(args=args, okw=okw, refs=refs, reads=reads, build=build), open(os.environ["SOAK_DUMP"], "wb"))
            if os.environ.get("SOAK_HOOKS"): hooks = {int(k_): int(v_) for k_, v_ in (kv.split(":") for kv in os.environ["SOAK_HOOKS"].split(",") if kv)}      # (debugging a --only run)
            d = B.BBDuk(args, refs=refs, build=build, hooks=hooks)
            if squeeze: fam_count["squeezed"] = fam_count.get("squeezed", 0) + 1
            lay = d.gpu.table_layout
            fam_count["layout_%d" % (lay & 3)] = fam_count.get("layout_%d" % (lay & 3), 0) + 1        # 0 cache-resident, 1 big, 2 seed
            if lay & 8: fam_count["qx_tabulated"] = fam_count.get("qx_tabulated", 0) + 1
            assert d.stored_kmers == o.stored_kmers or (d.gpu.table_layout & 3) == 2, ("stored", d.stored_kmers, o.stored_kmers)      # (a seed-layout map counts its parents)
            if fam == "kmask":
                T.check_kmask(d, o, reads, True); T.check_kmask(d, o, reads[:101], False)
            elif fam == "tips":
                T.check_tips(d, o, reads, True); T.check_tips(d, o, reads[:101], False)
            elif fam == "ksplit":
                T.check_split(d, o, reads)
            else:
                T.check_batch(d, o, reads, True); T.check_batch(d, o, reads[:101], False)
                if fam == "base" and rng.random() < 0.2:        # the same reads as FASTQ text through ingest -> packed operator -> writer
                    ingest_roundtrip(B, d, o, okw, reads, rng); fam_count["fastq_text"] = fam_count.get("fastq_text", 0) + 1
            d.close()
        except Exception as e:
            print(json.dumps({"FAILED": True, "seed": seed, "family": fam, "args": args, "okw": okw, "build": build, "squeeze": locals().get("squeeze", ""), "error": (str(e) or traceback.format_exc())[-2000:]}))
            sys.exit(1)
        n_cfg += 1; n_reads += len(reads); fam_count["tails" if tails else fam] = fam_count.get("tails" if tails else fam, 0) + 1
        if bigmap and fam != "base": fam_count["bigmap_" + fam] = fam_count.get("bigmap_" + fam, 0) + 1
        if uni: fam_count["uniform_length"] = fam_count.get("uniform_length", 0) + 1
    print(json.dumps({"soak_seconds": round(time.time() - t0, 1), "configs": n_cfg, "reads_checked": n_reads * 3, "families": fam_count, "mismatches": 0, "seed": a.seed}))


if __name__ == "__main__":
    main()
