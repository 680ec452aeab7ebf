"""
Golden fixtures over the reference's own sample reads: resources/sample1.fq.gz / sample2.fq.gz of the BBTools tree (100 pairs of 100-base
reads drawn from phiX with substitutions, N's and real quality strings), kept byte-identical as data/sample1.fq.gz / data/sample2.fq.gz.
Expected results come from oracle/spec.py, like every fixture here; tools/verify_with_reference.sh runs the real bbduk.sh over the same two
files where a JVM exists.  Run from the repo root:  python tests/golden/make_sample_golden.py
"""
import gzip
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))
from make_golden import run_case, ADAPTERS, PHIX  # noqa: E402


def sample_pairs():
    """The two files interleaved (mate 1, mate 2, ...), bases only."""
    def bases(path):
        with gzip.open(path, "rt") as f:
            lines = f.read().split("\n")
        return [lines[i + 1] for i in range(0, len(lines) - 1, 4)]
    r1, r2 = bases(os.path.join(ROOT, "data", "sample1.fq.gz")), bases(os.path.join(ROOT, "data", "sample2.fq.gz"))
    assert len(r1) == len(r2) == 100
    return [x for p in zip(r1, r2) for x in p]


def sample_names():
    def names(path):
        with gzip.open(path, "rt") as f:
            lines = f.read().split("\n")
        return [lines[i][1:] for i in range(0, len(lines) - 1, 4)]
    n1, n2 = names(os.path.join(ROOT, "data", "sample1.fq.gz")), names(os.path.join(ROOT, "data", "sample2.fq.gz"))
    return [x for p in zip(n1, n2) for x in p]


def cli_tsv(stem, spec_args, ref, ktrim):
    """The per-read table bbduk_cli prints (tsv=) for in=data/sample1.fq.gz in2=data/sample2.fq.gz: name, length, result, scaffold id, new
    length, flags -- what tools/compare_with_reference.py holds the real bbduk.sh's output files against."""
    from oracle.spec import Spec, Args
    s = Spec(Args(**spec_args)); s.load_fasta(ref)
    reads, names = sample_pairs(), sample_names()
    out = s.process_batch([r.encode() for r in reads], True)
    with open(os.path.join(os.path.dirname(__file__), stem + ".tsv"), "w") as f:
        for nm, r, t in zip(names, reads, out):
            f.write("%s\t%d\t%d\t%d\t%d\t%d\n" % (nm, len(r), t[0], t[1], len(r) - (t[0] if ktrim else 0), t[2]))


def len151_fixture():
    """2x151 pairs from the repo's generator (the commonest Illumina length: one k-mer end position more than the pair scan's block holds with
    k=23, so the tail pass serves it), AddAdapters-style names, interleaved FASTQ + the per-read table of ktrim=r k=23 mink=11 hdist=1."""
    from make_golden import synth_reads, truth_names, write_fastq
    from oracle.spec import Spec, Args
    kw = dict(read_len=151, ins_min=100, ins_max=175, sub_rate=0.01, n_rate=0.002)
    reads = synth_reads(21, 150, **kw)
    names = truth_names(21, 150, reads, **kw)
    here = os.path.dirname(__file__)
    write_fastq(os.path.join(here, "cli_len151_interleaved.fq"), names, reads)
    s = Spec(Args(k=23, mink=11, hdist=1, ktrimRight=True)); s.load_fasta(ADAPTERS)
    out = s.process_batch([r.encode() for r in reads], True)
    with open(os.path.join(here, "cli_len151.tsv"), "w") as f:
        for nm, r, t in zip(names, reads, out):
            f.write("%s\t%d\t%d\t%d\t%d\t%d\n" % (nm, len(r), t[0], t[1], len(r) - t[0], t[2]))
    print("cli_len151", len(reads), "reads,", sum(1 for t in out if t[0]), "trimmed,", sum(1 for r, t in zip(reads, out) if t[0] == 23), "with their only 23-mer hit at the last (tail) position")


def main():
    len151_fixture()
    reads = sample_pairs()
    cli_tsv("cli_sample_ktrimr_k25_phix", dict(k=25, ktrimRight=True), PHIX, True)
    cli_tsv("cli_sample_kfilter_k31_hdist1_phix", dict(k=31, hdist=1), PHIX, False)
    cli_tsv("cli_sample_kfilter_k31_mkf_phix", dict(k=31, minKmerFraction=0.5), PHIX, False)
    run_case("sample_kfilter_k31_hdist1_phix", "k=31 hdist=1 ref=phix", dict(k=31, hdist=1), PHIX, reads, True)
    run_case("sample_kfilter_k31_mkf_phix", "k=31 mkf=0.5 ref=phix", dict(k=31, minKmerFraction=0.5), PHIX, reads, True)
    run_case("sample_kfilter_k27_mcf_phix", "k=27 mcf=0.9 ref=phix", dict(k=27, minCoveredFraction=0.9), PHIX, reads, True)
    run_case("sample_kfilter_k40_phix", "k=40 ref=phix", dict(k=31, kbig=40), PHIX, reads, True)
    run_case("sample_ktrimr_k25_phix", "ktrim=r k=25 ref=phix", dict(k=25, ktrimRight=True), PHIX, reads, True)
    run_case("sample_ktriml_k25_restrictleft40_phix", "ktrim=l k=25 restrictleft=40 ref=phix", dict(k=25, ktrimLeft=True, restrictLeft=40), PHIX, reads, True)
    run_case("sample_kmask_k27_phix", "ktrim=n k=27 ref=phix", dict(k=27, ktrimN=True), PHIX, reads, True)
    run_case("sample_ktrimr_k23_mink11_hdist1_adapters", "ktrim=r k=23 mink=11 hdist=1 ref=adapters",
             dict(k=23, mink=11, hdist=1, ktrimRight=True), ADAPTERS, reads, True)


if __name__ == "__main__":
    main()
