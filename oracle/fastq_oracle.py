"""
fastq_oracle.py -- CPU restatement (numpy) of the reference's FASTQ record splitting, the checker for the device-side
ingest (bbtools_amd/csrc/bbduk_ingest.hip).  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED like the rest of oracle/ (no JVM,
no golden vectors for stream/ in the reference).

Follows stream/FASTQ.java:778-853 (toReadList: every four lines are one read: header, bases, plus line, qualities; the
'@' and '+' are asserted in quadToRead_slow :1047-1049) over the lines fileIO/ByteFile.nextLine returns (a line ends at
'\\n', one preceding '\\r' is dropped, an unterminated last line is still a line).
"""
import numpy as np


def split_lines(buf: np.ndarray, is_final: bool) -> np.ndarray:
    """Byte offset of every line start; entry j+1 is one past the newline that ends line j."""
    nl = np.flatnonzero(buf == 10)
    lines = np.concatenate([[0], nl + 1]).astype(np.int64)
    if is_final and len(buf) > 0 and buf[-1] != 10:
        lines = np.append(lines, len(buf) + 1)          # as if a newline sat at the end of the file
    return lines


def _line(buf, lines, j):
    s, e = int(lines[j]), int(lines[j + 1]) - 1
    e = min(e, len(buf))
    if e > s and buf[e - 1] == 13:
        e -= 1
    return s, e


def ingest(texts, is_final: bool, max_reads: int):
    """texts: one or two uint8 arrays.  Returns dict(n, consumed[], lines[], offsets, bases (ASCII, concatenated),
    first_bad) with reads interleaved 2i/2i+1 from the two texts."""
    ns = len(texts)
    if ns == 2:
        max_reads -= max_reads & 1
    lines = [split_lines(t, is_final) for t in texts]
    rec = min(min((len(l) - 1) // 4 for l in lines), max_reads // ns)
    n = rec * ns
    consumed = [int(min(l[4 * rec], len(t))) if rec else 0 for l, t in zip(lines, texts)]
    offsets = [0]
    parts = []
    first_bad = -1
    for i in range(n):
        t, l, r = texts[i % ns], lines[i % ns], i // ns
        h = _line(t, l, 4 * r); s = _line(t, l, 4 * r + 1); p = _line(t, l, 4 * r + 2); q = _line(t, l, 4 * r + 3)
        ok = h[1] > h[0] and t[h[0]] == ord("@") and p[1] > p[0] and t[p[0]] == ord("+") and (s[1] - s[0]) == (q[1] - q[0])
        if not ok and first_bad < 0:
            first_bad = i
        parts.append(t[s[0]:s[1]])
        offsets.append(offsets[-1] + s[1] - s[0])
    bases = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return dict(n=n, consumed=consumed, lines=[l[:4 * rec + 1] for l in lines], offsets=np.asarray(offsets, np.int64),
                bases=bases.astype(np.uint8), first_bad=first_bad)


def write_records(texts, lines, n, left, right, flags, want_removed: bool) -> bytes:
    """stream/FASTQ.java:474-490 toFASTQ on the reads of ingest(): '@' id, bases, a bare '+', qualities, trimmed by
    left[i]/right[i] (shared/TrimRead.trimByAmount), for the reads whose REMOVED flag equals want_removed."""
    ns = len(texts)
    out = bytearray()
    for i in range(n):
        if bool(flags[i] & 2) != bool(want_removed):
            continue
        t, l, r = texts[i % ns], lines[i % ns], i // ns
        h = _line(t, l, 4 * r); s = _line(t, l, 4 * r + 1); q = _line(t, l, 4 * r + 3)
        a, b = max(int(left[i]), 0), max(int(right[i]), 0)
        L = s[1] - s[0]
        nl = max(L - a - b, 0)
        out += bytes(t[h[0]:h[1]]) + b"\n" + bytes(t[s[0] + a:s[0] + a + nl]) + b"\n+\n" + bytes(t[q[0] + a:q[0] + a + nl]) + b"\n"
    return bytes(out)
