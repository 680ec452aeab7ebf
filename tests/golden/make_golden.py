"""
Generates the golden fixtures in this directory from oracle/spec.py (the string-based restatement).
Run from the repo root:  python tests/golden/make_golden.py
Inputs: data/adapters.fa, data/phix2.fa.gz (benchmark inputs named by BASELINE.json: ref=adapters, ref=phix)
and reads drawn from the repo's own deterministic generator plus hand-written edge cases.  Each fixture
stores the reads themselves, so nothing but this JSON is needed to replay it.
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle.spec import Spec, Args  # noqa: E402
from bbtools_amd.bbduk import synth_params, synth_generate_host, synth_pair_inserts, read_fasta  # noqa: E402

ADAPTERS = os.path.join(ROOT, "data", "adapters.fa")
PHIX = os.path.join(ROOT, "data", "phix2.fa.gz")


def synth_reads(seed, n_pairs, **kw):
    sp = synth_params(seed, **kw)
    b, off = synth_generate_host(sp, 0, n_pairs)
    return [bytes(b[off[i]:off[i + 1]]).decode() for i in range(2 * n_pairs)]


def run_case(name, cli, spec_args, ref, reads, paired):
    s = Spec(Args(**spec_args))
    if isinstance(ref, str):
        s.load_fasta(ref)
        ref_desc = {"fasta": os.path.relpath(ref, ROOT)}
    else:
        for r in ref:
            s.add_ref(r.encode())
        ref_desc = {"literal": ref}
    out = s.process_batch([r.encode() for r in reads], paired)
    fx = {"name": name, "cli": cli, "spec_args": spec_args, "ref": ref_desc, "paired": paired, "reads": reads,
          "expected": [list(t) for t in out], "counters": s.all_counters(), "stored_kmers": len(s.table),
          "num_scaffolds": s.nscaf}
    with open(os.path.join(os.path.dirname(__file__), name + ".json"), "w") as f:
        json.dump(fx, f, indent=0)
    hit = sum(1 for t in out if t[0])
    print(name, "reads", len(reads), "with_hits", hit, "keys", len(s.table))


def main():
    phix = read_fasta(PHIX)[0][1]
    # C1: ktrim=r k=23 hdist=0 ref=adapters
    reads = synth_reads(1, 60, ins_min=30, ins_max=260)
    run_case("c1_ktrimr_k23_hdist0_adapters", "ktrim=r k=23 hdist=0 ref=adapters",
             dict(k=23, ktrimRight=True), ADAPTERS, reads, True)
    # C2: ktrim=r k=23 mink=11 hdist=1 ref=adapters
    reads = synth_reads(2, 60, ins_min=30, ins_max=260, sub_rate=0.02, n_rate=0.004)
    run_case("c2_ktrimr_k23_mink11_hdist1_adapters", "ktrim=r k=23 mink=11 hdist=1 ref=adapters",
             dict(k=23, mink=11, hdist=1, ktrimRight=True), ADAPTERS, reads, True)
    # C2 + trimpairsevenly (the canonical adapter-trimming command line: ktrim=r k=23 mink=11 hdist=1 tpe)
    run_case("c2_ktrimr_k23_mink11_hdist1_tpe_adapters", "ktrim=r k=23 mink=11 hdist=1 tpe ref=adapters",
             dict(k=23, mink=11, hdist=1, ktrimRight=True, trimPairsEvenly=True), ADAPTERS, reads, True)
    # C2 flags with ktrim=n (kmask): expected = (masked bases, id0, flags)
    run_case("c2_kmask_k23_mink11_hdist1_adapters", "ktrim=n k=23 mink=11 hdist=1 ref=adapters",
             dict(k=23, mink=11, hdist=1, ktrimN=True), ADAPTERS, reads, True)
    # C3: kfilter k=31 hdist=1 ref=phix
    reads = synth_reads(3, 60, contam=phix, contam_frac=0.3, sub_rate=0.02, n_rate=0.004)
    run_case("c3_kfilter_k31_hdist1_phix", "k=31 hdist=1 ref=phix", dict(k=31, hdist=1), PHIX, reads, True)
    # C4-shaped: kfilter k=31 hdist=0 (forbidNs true) against phiX, unpaired
    reads = synth_reads(4, 40, contam=phix, contam_frac=0.3, sub_rate=0.01, n_rate=0.01)
    run_case("c4_kfilter_k31_hdist0_phix_unpaired", "k=31 hdist=0 ref=phix", dict(k=31), PHIX, reads, False)
    # hand-written known-answer cases
    ad = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCACATCACGATCTCGTATGCCGTCTTCTGCTTG"
    g = "ACGGTCAGTCCATGGACTTAGCATCAGGATACCATTGACCAGTAGGACCATAGCAGTTACGATCAGGACTTAACGGATTACAGGCATTAG"
    kat = [
        ad + "ACGT" * 5,                       # whole-read adapter: trimmed to 1 base -> discarded at minlength=10
        g[:60] + ad[:40],                      # adapter read-through at 60
        g[:80] + ad[:14],                      # only a short k-mer (14 >= mink) matches at the 3' end
        g[:84] + ad[:10],                      # 10 < mink: no match
        g[:50] + ad[:12] + "N" + ad[13:40],    # N inside the adapter: under hdist=1 N reads as A forward, 0 reverse
        g,                                     # clean
        "",                                    # empty read
        "ACGTACGTAC",                          # shorter than mink
        ad[:23],                               # exactly one k-mer
        g[:5] + ad[:30],                       # adapter at position 5 -> trimmed length 5 < minlength
    ]
    run_case("kat_ktrimr_k23_mink11_hdist1", "ktrim=r k=23 mink=11 hdist=1 ref=adapters",
             dict(k=23, mink=11, hdist=1, ktrimRight=True), ADAPTERS, kat, False)
    run_case("kat_ktriml_k23_mink11", "ktrim=l k=23 mink=11 ref=adapters",
             dict(k=23, mink=11, ktrimLeft=True), ADAPTERS,
             [ad[-30:] + g[:60], ad[-12:] + g[:60], g, ad, g[:40] + ad[:30] + g[40:70]], False)

    # k>31 (kbig): the C3 reads, k=40 counted as runs of 31-mer hits; expected = (countSetKmersBig return, credited id, flags)
    reads = synth_reads(3, 60, contam=phix, contam_frac=0.3, sub_rate=0.02, n_rate=0.004)
    run_case("c3_kfilter_k40_hdist1_phix", "k=40 hdist=1 ref=phix", dict(k=31, kbig=40, hdist=1), PHIX, reads, True)
    # findBestMatch over scaffolds that share sequence: expected = (hits, best id, flags)
    import random
    rng = random.Random(31)
    rs = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    sc = [rs(120) for _ in range(4)]
    sc.append(sc[0][:70] + sc[1][50:])                      # shares k-mers with scaffolds 1 and 2: first-writer-wins ids, ties
    fb = [sc[0][10:100], sc[1][:60] + sc[2][30:110], sc[2][:40] + rs(20) + sc[3][60:], rs(150), sc[4], sc[3][5:30] + "N" + sc[3][31:80],
          sc[1][80:] + sc[0][:45], sc[0][:26] + sc[1][:26] + sc[2][:26] + sc[3][:26], sc[2][50:74], ""]
    run_case("fbm_k25_literal", "k=25 fbm=t", dict(k=25, findBestMatch=True), sc, fb, True)
    # ksplit: adapters in the middle (split), at either end (trim), short k-mers at the 3' end, nothing
    ad = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCACATCACGATCTCGTATGCCGTCTTCTGCTTG"
    g = rs(400)
    ks = [g[:70] + ad[:40] + g[70:150], ad[:35] + g[:90], g[:90] + ad[:35], g[:100] + ad[:13], g[:100] + ad[:9], g[:150],
          g[:60] + ad[:30] + g[60:61], g[:1] + ad[5:40] + g[100:180], ad, g[:40] + ad[:25] + g[40:80] + ad[10:40] + g[80:120], "", "ACGT"]
    run_case("ksplit_k23_mink11_hdist1_adapters", "ksplit=t k=23 mink=11 hdist=1 ref=adapters",
             dict(k=23, mink=11, hdist=1, ksplit=True), ADAPTERS, ks, False)

    cli_fixtures(phix, ks)


def write_stats_files(here, stem, s, scaffolds):
    """stats= (3 and 5 columns) and rpkm= as BBDukProcessorS.writeStats / writeRPKM print them (:572-655); the #File line
    carries placeholders because it echoes the command line's paths."""
    import numpy as np
    names = [""] + [n for n, _ in scaffolds]; lens = [0] + [len(q) for _, q in scaffolds]
    readsIn, basesIn = s.counters[0], s.counters[1]
    rows = [(names[i], s.scafReads[i], s.scafBases[i]) for i in range(1, s.nscaf) if s.scafReads[i] > 0]
    rows.sort(key=lambda t: (-t[2], -t[1], t[0]))                # structures/StringCount.java:36-40
    rsum = sum(t[1] for t in rows)
    rmult, bmult = 100.0 / max(readsIn, 1), 100.0 / max(basesIn, 1)
    with open(os.path.join(here, stem + "_stats3.txt"), "w") as f:
        f.write("#File\tIN1\tIN2\n#Total\t%d\n#Matched\t%d\t%.5f%%\n#Name\tReads\tReadsPct\n" % (readsIn, rsum, rmult * rsum))
        for nm, r, b in rows:
            f.write("%s\t%d\t%.5f%%\n" % (nm, r, r * rmult))
    with open(os.path.join(here, stem + "_stats5.txt"), "w") as f:
        f.write("#File\tIN1\tIN2\n#Total\t%d\t%d\n#Matched\t%d\t%.5f%%\n#Name\tReads\tReadsPct\tBases\tBasesPct\n" % (readsIn, basesIn, rsum, rmult * rsum))
        for nm, r, b in rows:
            f.write("%s\t%d\t%.5f%%\t%d\t%.5f%%\n" % (nm, r, r * rmult, b, b * bmult))
    mapped = sum(s.scafReads)
    mult = np.float32(1000000000.0) / np.float32(max(1, mapped))
    with open(os.path.join(here, stem + "_rpkm.txt"), "w") as f:
        f.write("#File\tIN1\tIN2\n#Reads\t%d\n#Mapped\t%d\n#RefSequences\t%d\n#Name\tLength\tBases\tCoverage\tReads\tRPKM\n" % (readsIn, mapped, max(0, s.nscaf - 1)))
        for i in range(1, s.nscaf):
            if s.scafReads[i] > 0:
                invlen = 1.0 / max(1, lens[i])
                f.write("%s\t%d\t%d\t%.4f\t%d\t%.4f\n" % (names[i], lens[i], s.scafBases[i], s.scafBases[i] * invlen, s.scafReads[i], s.scafReads[i] * (float(mult) * invlen)))


    # refstats= (BBDukIndexMod.writeRefStats :196-245): the scaffold counters summed per reference FILE; here one file, "adapters"
    tot_len = sum(lens[1:]); tot_r = sum(s.scafReads[1:]); tot_b = sum(s.scafBases[1:])
    invlen = 1.0 / max(1, tot_len)
    with open(os.path.join(here, stem + "_refstats.txt"), "w") as f:
        f.write("#File\tIN1\tIN2\n#Reads\t%d\n#Mapped\t%d\n#References\t1\n#Name\tLength\tScaffolds\tBases\tCoverage\tReads\tRPKM\n" % (readsIn, mapped))
        if tot_r > 0:
            f.write("%s\t%d\t%d\t%d\t%.4f\t%d\t%.4f\n" % ("adapters", tot_len, s.nscaf - 1, tot_b, tot_b * invlen, tot_r, tot_r * (float(mult) * invlen)))


def truth_names(seed, n_pairs, reads, **kw):
    """Read names in the AddAdapters truth convention (jgi/AddAdapters.java:485: r.id = initial + "_" + remaining): the read's length
    and the number of its bases that are genome, i.e. before the adapter read-through starts -- so that the real bbduk.sh's output on
    these files can be graded (addadapters.sh grade) and compared (tools/verify_with_reference.sh).  A running pair index and the
    mate number follow after a blank, which keeps the names unique."""
    ins = synth_pair_inserts(synth_params(seed, **kw), 0, n_pairs)
    return ["%d_%d %d /%d" % (len(reads[i]), min(len(reads[i]), int(ins[i // 2])), i // 2, i % 2 + 1) for i in range(len(reads))]


def write_fastq(path, names, reads):
    with open(path, "w") as f:
        for i, (nm, r) in enumerate(zip(names, reads)):
            q = "".join(chr(33 + (7 * i + 3 * j) % 41) for j in range(len(r)))
            f.write("@%s\n%s\n+\n%s\n" % (nm, r, q))


def cli_fixtures(phix, ks):
    """FASTQ inputs + the per-read TSV bbduk_cli has to print for them (the non-JVM caller is pinned by
    fixture FASTQs and the spec's per-read results).  Read names follow the AddAdapters truth convention."""
    here = os.path.dirname(__file__)

    def tsv(path, names, reads, out, ktrim):
        with open(path, "w") as f:
            for nm, r, t in zip(names, reads, out):
                f.write("%s\t%d\t%d\t%d\t%d\t%d\n" % (nm, len(r), t[0], t[1], len(r) - (t[0] if ktrim else 0), t[2]))

    # paired, two files, ktrim=r (configs[1] flags)
    reads = synth_reads(12, 100, ins_min=30, ins_max=260, sub_rate=0.01, n_rate=0.002)
    names = truth_names(12, 100, reads, ins_min=30, ins_max=260, sub_rate=0.01, n_rate=0.002)
    write_fastq(os.path.join(here, "cli_c2_r1.fq"), names[0::2], reads[0::2])
    write_fastq(os.path.join(here, "cli_c2_r2.fq"), names[1::2], reads[1::2])
    s = Spec(Args(k=23, mink=11, hdist=1, ktrimRight=True)); s.load_fasta(ADAPTERS)
    tsv(os.path.join(here, "cli_c2.tsv"), names, reads, s.process_batch([r.encode() for r in reads], True), True)
    write_stats_files(here, "cli_c2", s, read_fasta(ADAPTERS))
    # the same pairs through ktrim=n: per-read TSV and the masked FASTQ (symbol N, quality '!')
    s = Spec(Args(k=23, mink=11, hdist=1, ktrimN=True)); s.load_fasta(ADAPTERS)
    out = s.process_batch([r.encode() for r in reads], True)
    tsv(os.path.join(here, "cli_kmask.tsv"), names, reads, out, False)
    with open(os.path.join(here, "cli_kmask_expected.fq"), "w") as f:
        for i, (nm, r, m) in enumerate(zip(names, reads, s.masks)):
            q = [chr(33 + (7 * (i // 2) + 3 * j) % 41) for j in range(len(r))]      # write_fastq's qualities (index within its file)
            b = list(r)
            for j in range(len(r)):
                if (m >> j) & 1:
                    b[j] = "N"; q[j] = "!"
            f.write("@%s\n%s\n+\n%s\n" % (nm, "".join(b), "".join(q)))
    # interleaved, kfilter against phiX (configs[2] flags)
    reads = synth_reads(13, 100, contam=phix, contam_frac=0.3, sub_rate=0.01, n_rate=0.002)
    names = truth_names(13, 100, reads, contam=phix, contam_frac=0.3, sub_rate=0.01, n_rate=0.002)
    write_fastq(os.path.join(here, "cli_c3_interleaved.fq"), names, reads)
    s = Spec(Args(k=31, hdist=1)); s.load_fasta(PHIX)
    tsv(os.path.join(here, "cli_c3.tsv"), names, reads, s.process_batch([r.encode() for r in reads], True), False)
    # the same file with k=40: K is split into k=31 + runs (countSetKmersBig)
    s = Spec(Args(k=31, kbig=40, hdist=1)); s.load_fasta(PHIX)
    tsv(os.path.join(here, "cli_c3_k40.tsv"), names, reads, s.process_batch([r.encode() for r in reads], True), False)
    # ksplit: unpaired input; kept reads (trimmed at an end or untouched) and the split pairs that leave through outm
    names = ["split%d" % i for i in range(len(ks))]
    write_fastq(os.path.join(here, "cli_ksplit.fq"), names, ks)
    s = Spec(Args(k=23, mink=11, hdist=1, ksplit=True)); s.load_fasta(ADAPTERS)
    out = s.process_batch([r.encode() for r in ks], False)
    tsv(os.path.join(here, "cli_ksplit.tsv"), names, ks, out, True)
    with open(os.path.join(here, "cli_ksplit_expected_out.fq"), "w") as fo, open(os.path.join(here, "cli_ksplit_expected_outm.fq"), "w") as fm:
        for i, (nm, r, t, (lm, rm)) in enumerate(zip(names, ks, out, s.splits)):
            q = "".join(chr(33 + (7 * i + 3 * j) % 41) for j in range(len(r)))
            if t[2] & 2:                                         # split: [0,leftmost) and [rightmost+1, len-1)
                fm.write("@%s\n%s\n+\n%s\n" % (nm, r[:lm], q[:lm]))
                fm.write("@%s\n%s\n+\n%s\n" % (nm, r[rm + 1:len(r) - 1], q[rm + 1:len(r) - 1]))
            elif t[0] > 0 and lm == 0:
                fo.write("@%s\n%s\n+\n%s\n" % (nm, r[t[0]:], q[t[0]:]))
            else:
                fo.write("@%s\n%s\n+\n%s\n" % (nm, r[:len(r) - t[0]], q[:len(r) - t[0]]))
    print("cli fixtures written")


if __name__ == "__main__":
    main()
