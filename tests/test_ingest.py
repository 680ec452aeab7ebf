"""Device-side FASTQ ingest (bbduk_fastq_ingest_device) against oracle/fastq_oracle.py: line offsets, record count,
consumed bytes, base offsets, packed codes / undefined bits; then end to end through the packed operators."""
import os
import random

import numpy as np
import pytest

from bbtools_amd import bbduk as B
from oracle import fastq_oracle as FQ
from oracle.oracle_ffi import Oracle, pack_reads
from tests import util

GOLD = os.path.join(util.ROOT, "tests", "golden")


def make_fastq(rng, nrec, crlf=False, final_newline=True, lengths=(0, 1, 15, 16, 17, 31, 50, 100, 150, 151, 300, 1000)):
    eol = b"\r\n" if crlf else b"\n"
    recs, reads = [], []
    for i in range(nrec):
        L = rng.choice(lengths)
        seq = bytes(rng.choice(b"ACGTACGTACGTNacgtnRY.") for _ in range(L))
        qual = bytes(33 + rng.randrange(41) for _ in range(L))              # may contain '@' and '+': only line position decides
        name = b"@r%d %s" % (i, bytes(rng.choice(b"ab:/ 12@+") for _ in range(rng.randint(0, 30))))
        plus = b"+" + (name[1:] if rng.random() < 0.2 else b"")
        recs.append(eol.join([name, seq, plus, qual]))
        reads.append(seq)
    text = eol.join(recs) + (eol if final_newline and recs else b"")
    return np.frombuffer(text, np.uint8).copy(), reads


def test_oracle_splits_like_a_line_reader():
    rng = random.Random(1)
    for crlf in (False, True):
        for fin in (True, False):
            text, reads = make_fastq(rng, 23, crlf, fin)
            got = FQ.ingest([text], True, 10 ** 6)
            assert got["n"] == 23 and got["first_bad"] == -1 and got["consumed"] == [len(text)]
            assert [bytes(got["bases"][got["offsets"][i]:got["offsets"][i + 1]]) for i in range(23)] == reads
            # a chunk (not final) that ends in the middle of a record: that record stays with the caller
            cut = len(text) - 7
            part = FQ.ingest([text[:cut]], False, 10 ** 6)
            assert part["n"] == 22 and bytes(text[part["consumed"][0]:part["consumed"][0] + 3]) == b"@r2"
    bad = np.frombuffer(b"@a\nACGT\n+\nIIII\n@b\nAC\nIIII\n+\n", np.uint8)
    assert FQ.ingest([bad], True, 100)["first_bad"] == 1
    assert FQ.ingest([np.frombuffer(b"@a\nACGT\n+\nIII\n", np.uint8)], True, 100)["first_bad"] == 0


def _check(texts, is_final, max_reads=None, shift=0):
    import torch
    want = FQ.ingest(texts, is_final, max_reads if max_reads is not None else 10 ** 9)
    # shift: the text starts `shift` bytes into its allocation (round 5: any alignment -- a chunk's carried-over tail starts wherever the record does)
    dts = [torch.from_numpy(np.concatenate([np.full(shift, 10, np.uint8), t, np.zeros(16, np.uint8)])).cuda()[shift:shift + len(t)] for t in texts]
    fb = B.fastq_ingest_device(dts[0], dts[1] if len(dts) == 2 else None, is_final, max_reads)
    assert fb.n == want["n"] and list(fb.consumed[:len(texts)]) == want["consumed"]
    n = fb.n
    rec = n // len(texts)
    assert np.array_equal(fb.lines1[:4 * rec + 1].cpu().numpy(), want["lines"][0])
    if len(texts) == 2:
        assert np.array_equal(fb.lines2[:4 * rec + 1].cpu().numpy(), want["lines"][1])
    assert np.array_equal(fb.offsets[:n + 1].cpu().numpy(), want["offsets"])
    total = int(want["offsets"][-1])
    assert fb.total_bases == total
    hc, hu = B.pack_bases_host(want["bases"])
    gc = fb.codes[:len(hc)].cpu().numpy().view(np.uint32); gu = fb.undef[:len(hu)].cpu().numpy().view(np.uint32)
    assert np.array_equal(gc, hc)                                # undefined bases carry code 0 in both packers
    assert np.array_equal(gu, hu)                                # incl. the all-ones tail of the last word
    return fb, want


@pytest.mark.gpu
def test_ingest_matches_oracle_on_random_fastq():
    rng = random.Random(7)
    for nrec in (0, 1, 2, 5, 64, 1000):
        for crlf in (False, True):
            for fin in (True, False):
                text, _ = make_fastq(rng, nrec, crlf, fin)
                _check([text], True)
                if nrec >= 2:
                    cut = len(text) - rng.randint(1, min(len(text) - 1, 400))
                    _check([text[:cut]], False)                  # chunk mode: the tail record is left to the caller
                    _check([text], True, max_reads=nrec - 1)     # capacity-limited: consumed stops at the last record taken
    a, _ = make_fastq(rng, 300); b, _ = make_fastq(rng, 280, crlf=True, final_newline=False)
    _check([a, b], True)                                         # two files: mates interleave, the shorter file bounds the batch
    _check([a, b], True, max_reads=101)
    big, _ = make_fastq(rng, 60000, lengths=(100, 150, 151))     # > one scan block of reads, ~20 MB of text
    _check([big], True)
    for shift in (1, 3, 7, 8, 13, 15):                          # texts that start at any byte
        t, _ = make_fastq(rng, 700, crlf=shift == 7, final_newline=shift != 13)
        _check([t], True, shift=shift)
        _check([a, b], True, shift=shift)
    _check([big], True, shift=5)


@pytest.mark.gpu
def test_ingest_reports_malformed_records():
    import torch
    rng = random.Random(3)
    text, _ = make_fastq(rng, 50)
    lines = bytes(text).split(b"\n")
    lines[4 * 17 + 2] = b"-"                                     # record 17 loses its '+'
    bad = np.frombuffer(b"\n".join(lines), np.uint8).copy()
    with pytest.raises(B.BBDukError, match="first bad read 17"):
        B.fastq_ingest_device(torch.from_numpy(bad).cuda())
    lines = bytes(text).split(b"\n")
    lines[4 * 30 + 3] = lines[4 * 30 + 3] + b"I"                 # qualities longer than bases
    bad = np.frombuffer(b"\n".join(lines), np.uint8).copy()
    with pytest.raises(B.BBDukError, match="first bad read 30"):
        B.fastq_ingest_device(torch.from_numpy(bad).cuda())


@pytest.mark.gpu
def test_ingest_to_ktrim_end_to_end():
    """FASTQ text -> device ingest -> packed ktrim operator == host-parsed reads through the ASCII operator == oracle."""
    import torch
    t1 = np.fromfile(os.path.join(GOLD, "cli_c2_r1.fq"), np.uint8); t2 = np.fromfile(os.path.join(GOLD, "cli_c2_r2.fq"), np.uint8)
    fb = B.fastq_ingest_device(torch.from_numpy(t1).cuda(), torch.from_numpy(t2).cuda())
    r1 = bytes(t1).split(b"\n")[1::4]; r2 = bytes(t2).split(b"\n")[1::4]
    reads = [x for pair in zip(r1, r2) for x in pair]
    assert fb.n == len(reads) == 200
    args, okw, ref = util.CONFIGS["c2"]
    d = B.BBDuk(args); o = Oracle(**okw); o.load_fasta(ref)
    n = fb.n
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_i = torch.empty(n, dtype=torch.int32, device="cuda")
    d_f = torch.empty(n, dtype=torch.uint8, device="cuda"); d_c = torch.zeros(d.gpu.counters_len, dtype=torch.int64, device="cuda")
    d.gpu.process_batch_packed_device(fb.codes, fb.undef, fb.offsets[:n + 1].contiguous(), fb.total_bases, True, d_a, d_i, d_f, d_c)
    torch.cuda.synchronize()
    b, off = pack_reads(reads)
    oa, oi, of = o.process_batch(b, off, True)
    assert np.array_equal(d_a.cpu().numpy(), oa) and np.array_equal(d_i.cpu().numpy(), oi) and np.array_equal(d_f.cpu().numpy(), of)
    assert np.array_equal(d_c.cpu().numpy(), o.counters())
    assert int((oa > 0).sum()) > 20
    d.close()


@pytest.mark.gpu
def test_writer_matches_oracle_and_the_cli_files(tmp_path):
    """ingest -> ktrim -> device writer: the kept / removed FASTQ texts equal the oracle writer's, and equal what bbduk_cli
    (host parser, host writer) puts into out= / outm= for the same input."""
    import subprocess
    import torch
    from bbtools_amd import _build
    rng = random.Random(11)
    # (a) random text, arbitrary trims and flags, one and two files, CRLF input
    for crlf in (False, True):
        a, _ = make_fastq(rng, 400, crlf); b, _ = make_fastq(rng, 400, crlf)
        for texts in ([a], [a, b]):
            dts = [torch.from_numpy(t).cuda() for t in texts]
            fb = B.fastq_ingest_device(dts[0], dts[1] if len(dts) == 2 else None)
            want = FQ.ingest(texts, True, 10 ** 9)
            n = fb.n
            lens = np.diff(want["offsets"])
            left = np.array([rng.randint(-2, max(0, l // 2)) for l in lens], np.int32)
            right = np.array([rng.randint(0, max(0, l - max(x, 0))) for l, x in zip(lens, left)], np.int32)
            flags = np.array([rng.choice([0, 0, 1, 2, 3]) for _ in range(n)], np.uint8)
            d_out = torch.empty(sum(len(t) for t in texts) + 64, dtype=torch.uint8, device="cuda")
            for removed in (False, True):
                nb = B.fastq_write_device(dts[0], fb, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), torch.from_numpy(flags).cuda(),
                                          removed, d_out, dts[1] if len(dts) == 2 else None)
                assert bytes(d_out[:nb].cpu().numpy()) == FQ.write_records(texts, want["lines"], n, left, right, flags, removed)
            nb = B.fastq_write_device(dts[0], fb, None, None, None, False, d_out, dts[1] if len(dts) == 2 else None)      # no trims, no flags: everything
            assert bytes(d_out[:nb].cpu().numpy()) == FQ.write_records(texts, want["lines"], n, np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint8), False)
    # (b) the CLI's files
    f1, f2 = os.path.join(GOLD, "cli_c2_r1.fq"), os.path.join(GOLD, "cli_c2_r2.fq")
    out, outm = str(tmp_path / "clean.fq"), str(tmp_path / "removed.fq")
    r = subprocess.run([_build.cli_path(), "in=" + f1, "in2=" + f2, "out=" + out, "outm=" + outm, "ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters", "deviceingest=f"],
                       cwd=util.ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    t1, t2 = np.fromfile(f1, np.uint8), np.fromfile(f2, np.uint8)
    d1, d2 = torch.from_numpy(t1).cuda(), torch.from_numpy(t2).cuda()
    fb = B.fastq_ingest_device(d1, d2)
    n = fb.n
    d = B.BBDuk("ktrim=r k=23 mink=11 hdist=1 ref=adapters")
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_i = torch.empty(n, dtype=torch.int32, device="cuda")
    d_f = torch.empty(n, dtype=torch.uint8, device="cuda"); d_c = torch.zeros(d.gpu.counters_len, dtype=torch.int64, device="cuda")
    d.gpu.process_batch_packed_device(fb.codes, fb.undef, fb.offsets[:n + 1].contiguous(), fb.total_bases, True, d_a, d_i, d_f, d_c)
    d_out = torch.empty(len(t1) + len(t2) + 64, dtype=torch.uint8, device="cuda")
    nb = B.fastq_write_device(d1, fb, None, d_a, d_f, False, d_out, d2)
    assert bytes(d_out[:nb].cpu().numpy()) == open(out, "rb").read()
    nb = B.fastq_write_device(d1, fb, None, d_a, d_f, True, d_out, d2)
    assert bytes(d_out[:nb].cpu().numpy()) == open(outm, "rb").read()
    d.close()


@pytest.mark.gpu
def test_memory_and_stream_helpers_round_trip():
    """The plumbing a caller without HIP bindings uses (bbduk_cli's pipeline): pinned and device buffers, a non-blocking stream, asynchronous copies
    host -> device -> device -> host and the wait; the synchronous forms beside them."""
    import ctypes as C
    L = B.lib()
    vp = C.c_void_p
    for f, at in ((L.bbduk_pinned_malloc, [C.c_int64, C.POINTER(vp)]), (L.bbduk_device_malloc, [C.c_int32, C.c_int64, C.POINTER(vp)]), (L.bbduk_stream_create, [C.c_int32, C.POINTER(vp)]),
                  (L.bbduk_copy_async, [C.c_int32, vp, vp, C.c_int64, C.c_int32, vp]), (L.bbduk_stream_synchronize, [C.c_int32, vp]), (L.bbduk_stream_destroy, [C.c_int32, vp]),
                  (L.bbduk_copy_from_device, [C.c_int32, vp, vp, C.c_int64, vp]), (L.bbduk_device_memset, [C.c_int32, vp, C.c_int32, C.c_int64, vp]),
                  (L.bbduk_pinned_free, [vp]), (L.bbduk_device_free, [C.c_int32, vp])):
        f.argtypes = at; f.restype = C.c_int
    n = 3_000_001
    h_in, h_out, d_a, d_b, st = vp(), vp(), vp(), vp(), vp()
    assert L.bbduk_pinned_malloc(n, C.byref(h_in)) == 0 and L.bbduk_pinned_malloc(n, C.byref(h_out)) == 0
    assert L.bbduk_device_malloc(0, n, C.byref(d_a)) == 0 and L.bbduk_device_malloc(0, n, C.byref(d_b)) == 0 and L.bbduk_stream_create(0, C.byref(st)) == 0 and st.value
    src = np.random.default_rng(3).integers(0, 256, n, dtype=np.uint8)
    C.memmove(h_in.value, src.ctypes.data, n)
    assert L.bbduk_device_memset(0, d_b, 0, n, None) == 0
    assert L.bbduk_copy_async(0, d_a, h_in, n, 0, st) == 0 and L.bbduk_copy_async(0, vp(d_b.value + 5), vp(d_a.value + 5), n - 5, 2, st) == 0       # (any alignment)
    assert L.bbduk_copy_async(0, h_out, d_b, n, 1, st) == 0 and L.bbduk_stream_synchronize(0, st) == 0
    got = np.ctypeslib.as_array(C.cast(h_out.value, C.POINTER(C.c_uint8)), (n,))
    assert not got[:5].any() and np.array_equal(got[5:], src[5:])
    back = np.empty(n, np.uint8)
    assert L.bbduk_copy_from_device(0, back.ctypes.data, d_a, n, None) == 0 and np.array_equal(back, src)
    assert L.bbduk_copy_async(0, d_a, h_in, -1, 0, st) != 0 and L.bbduk_copy_async(0, d_a, h_in, 8, 7, st) != 0                               # bad size / kind
    assert L.bbduk_stream_destroy(0, st) == 0 and L.bbduk_device_free(0, d_a) == 0 and L.bbduk_device_free(0, d_b) == 0
    assert L.bbduk_pinned_free(h_in) == 0 and L.bbduk_pinned_free(h_out) == 0
