#!/bin/bash
# Pins this repository's oracle against the REAL BBDuk (the reference is Java: there is no JVM in the build image, so this
# cannot run there -- it is meant to run, unmodified, on any machine with a JDK and a BBTools checkout).
#
#   tools/verify_with_reference.sh /path/to/bbtools [workdir]
#
# For every fixture under tests/golden/cli_* it runs the reference's own bbduk.sh with the fixture's flags and compares reads,
# trimmed lengths, kept / removed routing and the stats files with the expectations committed beside the fixture (written by
# tests/golden/make_golden.py from oracle/spec.py, and reproduced bit for bit by the HIP path in tests/test_cli.py).  The fixture
# reads carry AddAdapters-style names (<initial>_<remaining>, jgi/AddAdapters.java:485), so `addadapters.sh in=... grade` can
# grade the trimmed output as well.  Exit status 0 = the real BBDuk agrees with every expectation.
set -u
BB=${1:?usage: verify_with_reference.sh /path/to/bbtools [workdir]}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
G=$ROOT/tests/golden
W=${2:-$(mktemp -d)}
mkdir -p "$W"
DUK="$BB/bbduk.sh"
[ -x "$DUK" ] || { echo "no bbduk.sh under $BB"; exit 2; }
REFA=$ROOT/data/adapters.fa
REFP=$ROOT/data/phix2.fa.gz
COMMON="ordered=t overwrite=t -Xmx2g t=4"
fail=0
run() { echo "+ $*"; "$@" > "$W/last.log" 2>&1 || { echo "  bbduk.sh failed:"; tail -5 "$W/last.log"; fail=1; }; }
cmp_stats() {   # the reference's stats / rpkm file vs the committed one, without the "#File" line (it names the input paths)
  if diff <(grep -v '^#File' "$1") <(grep -v '^#File' "$2") > "$W/stats.diff"; then echo "  $(basename "$2"): identical"; else echo "  $(basename "$2"): DIFFERS"; head -10 "$W/stats.diff"; fail=1; fi
}

# 1. configs[1] flags: paired, two files, ktrim=r k=23 mink=11 hdist=1 (+ stats=, 5-column stats, rpkm=)
run "$DUK" in="$G/cli_c2_r1.fq" in2="$G/cli_c2_r2.fq" out="$W/c2_o1.fq" out2="$W/c2_o2.fq" outm="$W/c2_m1.fq" outm2="$W/c2_m2.fq" \
    stats="$W/c2_stats.txt" rpkm="$W/c2_rpkm.txt" refstats="$W/c2_refstats.txt" ktrim=r k=23 mink=11 hdist=1 ref="$REFA" $COMMON
python3 "$ROOT/tools/compare_with_reference.py" --tsv "$G/cli_c2.tsv" --out "$W/c2_o1.fq" "$W/c2_o2.fq" --outm "$W/c2_m1.fq" "$W/c2_m2.fq" || fail=1
cmp_stats "$W/c2_stats.txt" "$G/cli_c2_stats3.txt"
cmp_stats "$W/c2_rpkm.txt" "$G/cli_c2_rpkm.txt"
cmp_stats "$W/c2_refstats.txt" "$G/cli_c2_refstats.txt"
run "$DUK" in="$G/cli_c2_r1.fq" in2="$G/cli_c2_r2.fq" stats="$W/c2_stats5.txt" statscolumns=5 ktrim=r k=23 mink=11 hdist=1 ref="$REFA" $COMMON
cmp_stats "$W/c2_stats5.txt" "$G/cli_c2_stats5.txt"

# 2. the same pairs through ktrim=n: the masked FASTQ byte for byte (interleaved output)
run "$DUK" in="$G/cli_c2_r1.fq" in2="$G/cli_c2_r2.fq" out="$W/kmask.fq" ktrim=n k=23 mink=11 hdist=1 ref="$REFA" $COMMON
if diff "$W/kmask.fq" "$G/cli_kmask_expected.fq" > "$W/kmask.diff"; then echo "  cli_kmask_expected.fq: identical"; else echo "  cli_kmask_expected.fq: DIFFERS"; head -8 "$W/kmask.diff"; fail=1; fi

# 3. configs[2] flags: interleaved, kfilter k=31 hdist=1 against phiX; and k=40 (k>31: runs of 31-mers)
run "$DUK" in="$G/cli_c3_interleaved.fq" int=t out="$W/c3_o.fq" outm="$W/c3_m.fq" k=31 hdist=1 ref="$REFP" $COMMON
python3 "$ROOT/tools/compare_with_reference.py" --kfilter --tsv "$G/cli_c3.tsv" --out "$W/c3_o.fq" --outm "$W/c3_m.fq" || fail=1
run "$DUK" in="$G/cli_c3_interleaved.fq" int=t out="$W/c3k40_o.fq" outm="$W/c3k40_m.fq" k=40 hdist=1 ref="$REFP" $COMMON
python3 "$ROOT/tools/compare_with_reference.py" --kfilter --tsv "$G/cli_c3_k40.tsv" --out "$W/c3k40_o.fq" --outm "$W/c3k40_m.fq" || fail=1

# 3b. the reference's own sample reads (resources/sample1.fq.gz, sample2.fq.gz: kept byte-identical under data/), gzipped, two files
S1=$ROOT/data/sample1.fq.gz; S2=$ROOT/data/sample2.fq.gz
run "$DUK" in="$S1" in2="$S2" out="$W/s_tr_o.fq" outm="$W/s_tr_m.fq" ktrim=r k=25 ref="$REFP" $COMMON
python3 "$ROOT/tools/compare_with_reference.py" --tsv "$G/cli_sample_ktrimr_k25_phix.tsv" --out "$W/s_tr_o.fq" --outm "$W/s_tr_m.fq" || fail=1
run "$DUK" in="$S1" in2="$S2" out="$W/s_kf_o.fq" outm="$W/s_kf_m.fq" k=31 hdist=1 ref="$REFP" $COMMON
python3 "$ROOT/tools/compare_with_reference.py" --kfilter --tsv "$G/cli_sample_kfilter_k31_hdist1_phix.tsv" --out "$W/s_kf_o.fq" --outm "$W/s_kf_m.fq" || fail=1
run "$DUK" in="$S1" in2="$S2" out="$W/s_mkf_o.fq" outm="$W/s_mkf_m.fq" k=31 mkf=0.5 ref="$REFP" $COMMON
python3 "$ROOT/tools/compare_with_reference.py" --kfilter --tsv "$G/cli_sample_kfilter_k31_mkf_phix.tsv" --out "$W/s_mkf_o.fq" --outm "$W/s_mkf_m.fq" || fail=1

# 3c. 2x151 pairs (one k-mer end position more than 128 with k=23), interleaved
run "$DUK" in="$G/cli_len151_interleaved.fq" int=t out="$W/l151_o.fq" outm="$W/l151_m.fq" ktrim=r k=23 mink=11 hdist=1 ref="$REFA" $COMMON
python3 "$ROOT/tools/compare_with_reference.py" --tsv "$G/cli_len151.tsv" --out "$W/l151_o.fq" --outm "$W/l151_m.fq" || fail=1

# 4. ksplit: kept reads and the split pairs byte for byte
run "$DUK" in="$G/cli_ksplit.fq" out="$W/ks_o.fq" outm="$W/ks_m.fq" ksplit=t k=23 mink=11 hdist=1 ref="$REFA" $COMMON
for pair in "ks_o.fq cli_ksplit_expected_out.fq" "ks_m.fq cli_ksplit_expected_outm.fq"; do
  set -- $pair
  if diff "$W/$1" "$G/$2" > "$W/ks.diff"; then echo "  $2: identical"; else echo "  $2: DIFFERS"; head -8 "$W/ks.diff"; fail=1; fi
done

# 5. Seal (jgi/Seal.java; include/seal_gpu.h): pairs binned by scaffold, the four ambig= modes, per-scaffold files and the stats file
SEAL="$BB/seal.sh"
if [ -x "$SEAL" ]; then
  for mode in all first toss random; do
    run "$SEAL" in="$G/seal_reads.fq" int=t ref="$G/seal_refs.fa" pattern="$W/seal_${mode}_%.fq" outu="$W/seal_${mode}_u.fq" stats="$W/seal_${mode}_stats.txt" \
        ambig=$mode k=31 mm=t statscolumns=5 $COMMON
    python3 "$ROOT/tools/compare_seal_with_reference.py" --tsv "$G/seal_$mode.tsv" --pattern "$W/seal_${mode}_%.fq" --outu "$W/seal_${mode}_u.fq" \
        --names scafA scafB scafC scafD scafE scafF || fail=1
    cmp_stats "$W/seal_${mode}_stats.txt" "$G/seal_${mode}_stats.txt"
  done
else echo "no seal.sh under $BB: the Seal fixtures were not checked"; fi

# 5b. the guides' own worked examples (tests/test_documented_behaviour.py: expectations derived by hand from BBDukGuide.txt / bbduk.sh, which the
#     restatements and the HIP operators already meet): the real tool on the same inputs
D=$W/doc
for name in $(python3 "$ROOT/tools/doc_cases.py" export "$D"); do
  run "$DUK" in="$D/$name.fq" ref="$D/$name.fa" out="$D/$name.out.fq" outm="$D/$name.outm.fq" $(cat "$D/$name.args") $COMMON
  python3 "$ROOT/tools/doc_cases.py" check "$D" "$name" "$D/$name.out.fq" "$D/$name.outm.fq" || fail=1
done

# 6. optional: the reference's own grader on the trimmed reads (names are <initial>_<remaining>)
if [ -x "$BB/addadapters.sh" ]; then "$BB/addadapters.sh" in="$W/c2_o1.fq" in2="$W/c2_o2.fq" grade 2>&1 | tail -15; fi

if [ $fail -eq 0 ]; then echo "REFERENCE AGREES with every committed expectation: parity pinned."; else echo "DIFFERENCES found (see above; work dir $W)."; fi
exit $fail
