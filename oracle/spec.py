"""
spec.py -- SECOND, structurally different restatement of the BBDuk k-mer matching path.
TEST INFRASTRUCTURE ONLY (used by tests/ and by tests/golden/make_golden.py).  PARITY UNPINNED
(no JVM in the image; the reference ships no golden vectors for this path -- SURVEY.md §8c).

Where oracle/bbduk_oracle.c follows the reference's sequential rolling loops line by line, this file
works on *strings* and uses the position-parallel closed form: every k-mer
is re-derived from a slice of the read, the reference table is a plain dict.  The two restatements share
no code and no data structure, so agreement between them (tests/test_oracle.py differential fuzz) is
evidence that both read the reference the same way.

Citations are relative to /root/reference/current/ .
"""
from dataclasses import dataclass, field

# dna/AminoAcid.java:1284-1311
_FWD = {c: i for i, c in enumerate("ACGT")}
_FWD.update({c.lower(): i for c, i in list(_FWD.items())})
_FWD.update({"U": 3, "u": 3})
_COMP = {c: 3 - v for c, v in _FWD.items()}


def defined(ch: int) -> bool:
    return ch < 128 and chr(ch) in _FWD


def fwd_code(ch: int) -> int:           # baseToNumber0: undefined -> 0
    return _FWD.get(chr(ch), 0) if ch < 128 else 0


def comp_code(ch: int) -> int:          # baseToComplementNumber0: undefined -> 0
    return _COMP.get(chr(ch), 0) if ch < 128 else 0


def rcomp_int(kmer: int, k: int) -> int:
    """dna/AminoAcid.java:585-601, done the slow obvious way (digit reversal)."""
    out = 0
    for _ in range(k):
        out = (out << 2) | (3 - (kmer & 3))
        kmer >>= 2
    return out


@dataclass
class Args:
    k: int = 27
    mink: int = -1
    hdist: int = 0
    hdist2: int = -1
    qhdist: int = 0
    qhdist2: int = -1
    maskMiddle: bool = True
    midMaskLen: int = 0
    rcomp: bool = True
    forbidN: bool = False
    ktrimRight: bool = False
    ktrimLeft: bool = False
    maxBadKmers0: int = 0
    minReadLength: int = 10
    minLenFraction: float = 0.0
    requireBothBad: bool = False
    trimPad: int = 0
    ktrimExclusive: bool = False
    restrictLeft: int = 0
    restrictRight: int = 0
    skipR1: bool = False
    skipR2: bool = False
    trimPairsEvenly: bool = False   # tpe
    qSkip: int = 1                  # qskip=
    speed: int = 0                  # speed=
    minKmerFraction: float = 0.0    # mkf=
    minCoveredFraction: float = 0.0 # mcf=
    ktrimN: bool = False            # ktrim=n / kmask=  (mask instead of trim; kmaskfullycovered is not restated)
    kbig: int = -1                  # k>31 on the command line: k is then 31 and kbig the requested k (BBDukParser.java:164-165)
    findBestMatch: bool = False     # findbestmatch / fbm (kfilter only; rename is not restated)
    ksplit: bool = False            # ksplit=t (unpaired reads only)
    kmaskFullyCovered: bool = False # kmaskfullycovered / mfc: mask only bases covered by matching k-mers in every window
    trimFailuresTo1bp: bool = False # trimfailures: discarded reads are cut to one base instead; nothing is evicted (:1431, 1464-1488)
    minSkip: int = 1                # rskip / minskip / maxskip: reference-side k-mer skipping (BBDukLoader.java:417, 432-449)
    maxSkip: int = 1


COUNTER_NAMES = ["readsIn", "basesIn", "readsKTrimmed", "basesKTrimmed", "readsKFiltered", "basesKFiltered",
                 "readsOutu", "basesOutu", "readsOutm", "basesOutm"]
NCOUNTERS = 16
FLAG_DISCARDED, FLAG_REMOVED = 1, 2


class Spec:
    def __init__(self, a: Args):
        self.a = a
        k = a.k
        assert 1 <= k <= 31
        # bbduk/BBDukParser.java:130-150
        self.hdist = a.hdist
        self.hdist2 = a.hdist if a.hdist2 == -1 else a.hdist2
        self.qhdist = a.qhdist
        self.qhdist2 = a.qhdist if a.qhdist2 == -1 else a.qhdist2
        self.forbidNs = a.forbidN or self.hdist < 1
        self.k = k
        mm = a.maskMiddle and not a.kbig > k                                      # :237-243 (before minlen2 is derived)
        mml = (a.midMaskLen if a.midMaskLen > 0 else 2 - (k & 1)) if mm else 0     # :230-236
        self.mink = min(a.mink, k)                                                 # :245
        self.minlen = k - 1                                                        # :274
        self.minlen2 = (k - mml) // 2 if mm else k                                 # :276 (before mink switches mm off)
        self.kmask = 1 << (2 * k)
        self.useShortKmers = 0 < self.mink < k                                     # :289
        if self.useShortKmers:
            mm, mml = False, 0                                                     # :290-296
            assert a.ktrimLeft or a.ktrimRight or a.ktrimN or a.ksplit            # :301
        if mm:                                                                     # :303-312
            self.middleMask = ~(((1 << (2 * mml)) - 1) << (((k - mml) // 2) * 2))
        else:
            self.middleMask = -1
        self.kfilter = not (a.ktrimLeft or a.ktrimRight or a.ktrimN or a.ksplit)
        self.kbig = a.kbig if a.kbig > k else k
        self.keff = max(k, self.kbig)                                              # :231
        assert not (self.kbig > k and not self.kfilter) and not (self.kbig > k and (a.speed > 0 or a.qSkip > 1))   # :207-223 reduce kbig to k
        assert not (a.findBestMatch and self.kbig > k)                             # :299
        self.splits = []           # ksplit: per-read (leftmost, rightmost) or (-1, -1), process order
        self.rieb = (not a.requireBothBad) and (not a.trimFailuresTo1bp)           # :109
        self.table = {}            # key -> id  (first writer wins == smallest id; ids ascend in file order)
        self.nscaf = 1             # scaffoldNames[0] reserved (bbduk/BBDukIndex.java:105-107)
        self.counters = [0] * NCOUNTERS
        self.scafReads = [0]
        self.scafBases = [0]
        self.masks = []            # ktrim=n: per-read base masks of the reads processed so far (process order)
        self.tips = []             # ktrim=rl: per-read (right, left) trim amounts, process order

    # ---------------------------------------------------------------- keys
    def key_of(self, kmer: int, rkmer: int, length: int) -> int:
        """bbduk/BBDukIndexMod.java:532-544 toValue.  kmer, rkmer < 2^62 so signed max == int max."""
        v = max(kmer, rkmer) if self.a.rcomp else kmer
        return (v & self.middleMask) | (1 << (2 * length))

    def _neighbours(self, kmer: int, length: int, dist: int, out: list):
        """bbduk/BBDukIndexMod.java:383-413 mutate (substitutions only), as a plain enumeration."""
        out.append(kmer)
        if dist > 0:
            for j in range(4):
                for i in range(length):
                    t = (kmer & ~(3 << (2 * i))) | (j << (2 * i))
                    if t != kmer:
                        self._neighbours(t, length, dist - 1, out)

    def _store(self, kmer: int, length: int, dist: int, sid: int) -> None:
        vs = []
        self._neighbours(kmer, length, dist, vs)
        for v in vs:
            key = self.key_of(v, rcomp_int(v, length), length)
            self.table.setdefault(key, sid)

    # ---------------------------------------------------------------- reference
    def add_ref(self, seq: bytes) -> None:
        """bbduk/BBDukLoader.java:416-494 (both branches: all k-mers, or every skip-th of a run), by slicing instead of rolling."""
        sid = self.nscaf
        self.nscaf += 1
        self.scafReads.append(0)
        self.scafBases.append(0)
        k, n = self.k, len(seq)
        if n < k:
            return
        ok = [defined(c) for c in seq]
        a = self.a
        mn = max(1, min(a.minSkip, a.maxSkip)); mx = max(mn, a.maxSkip)             # BBDukParser.java:148-149
        if self.kbig > k:
            mn = mx = 0                                                               # :238
        skip = max(mn, min(mx, k if n > 20000000 else 11 if n > 5000000 else 2 if n > 500000 else 0))   # BBDukLoader.java:397, 417
        for i in range(k - 1, n):
            win = seq[i - k + 1:i + 1]
            if not all(ok[i - k + 1:i + 1]):          # len>=k <=> the last k bases are all defined
                continue
            if skip > 1:                              # len = defined bases ending at i; only every skip-th k-mer of a run
                run = 0
                while run <= i and ok[i - run]:
                    run += 1
                if run % skip != 0:
                    continue
            kmer = 0
            for c in win:
                kmer = (kmer << 2) | fwd_code(c)
            self._store(kmer, k, self.hdist, sid)
            if self.useShortKmers:
                if i == k - 1:                         # addToMapRightShift: prefixes of the first k-mer
                    for L in range(k - 1, self.mink - 1, -1):
                        self._store(kmer >> (2 * (k - L)), L, self.hdist2, sid)
                if i == n - 1:                         # addToMapLeftShift: suffixes of the last k-mer
                    for L in range(k - 1, self.mink - 1, -1):
                        self._store(kmer & ((1 << (2 * L)) - 1), L, self.hdist2, sid)

    def load_fasta(self, path: str) -> int:
        import gzip
        op = gzip.open if path.endswith(".gz") else open
        seq, have, nrec = bytearray(), False, 0
        with op(path, "rb") as f:
            for line in f:
                line = line.rstrip(b"\r\n")
                if line.startswith(b">"):
                    if have and len(seq) > 0:
                        self.add_ref(bytes(seq)); nrec += 1
                    have, seq = True, bytearray()
                else:
                    seq += bytes(c for c in line if c > 13)
        if have and len(seq) > 0:
            self.add_ref(bytes(seq)); nrec += 1
        return nrec

    # ---------------------------------------------------------------- lookup
    def lookup(self, kmer: int, rkmer: int, length: int, qhdist: int, qpos: int) -> int:
        """bbduk/BBDukIndexMod.java:462-520 getValue / getValueInner (qskip gate :494, passesSpeed :506,562:
        with this index the speed gate sits on the query side only)."""
        if self.a.qSkip > 1 and qpos % self.a.qSkip != 0:
            return -1
        key = self.key_of(kmer, rkmer, length)
        vid = -1 if (self.a.speed > 0 and key % 17 < self.a.speed) else self.table.get(key, -1)
        if vid < 1 and qhdist > 0:
            for j in range(4):
                for i in range(length):
                    t = (kmer & ~(3 << (2 * i))) | (j << (2 * i))
                    if t != kmer:
                        vid = self.lookup(t, rcomp_int(t, length), length, qhdist - 1, qpos)
                        if vid >= 1:
                            return vid
        return vid

    # ---------------------------------------------------------------- scans (closed form)
    def _span(self, n: int):
        a = self.a
        start = 0 if a.restrictRight < 1 else max(0, n - a.restrictRight)
        stop = n if a.restrictLeft < 1 else min(n, a.restrictLeft)
        return start, stop

    def _main_hits(self, read: bytes, start: int, stop: int):
        """Yield (i, id) for every position of the main scan whose lookup returns id>0, in increasing i.
        bbduk/BBDukProcessorS.java:2009-2029 == :1547-1591."""
        for i, vid in self._main_lookups(read, start, stop):
            if vid > 0:
                yield i, vid

    def _main_lookups(self, read: bytes, start: int, stop: int):
        """Yield (i, id) for every position of the main scan that is looked up at all (id may be -1)."""
        k = self.k
        for i in range(max(start, k - 1), stop):
            lo = max(start, i - k + 1)
            lastN = start - 1
            if self.forbidNs:
                for j in range(i, start - 1, -1):
                    if not defined(read[j]):
                        lastN = j
                        break
            length = i - lastN
            if length < self.minlen2:
                continue
            kmer = 0
            for j in range(lo, i + 1):
                kmer |= fwd_code(read[j]) << (2 * (i - j))
            rk = 0
            for j in range(max(lo, lastN + 1), i + 1):
                rk |= comp_code(read[j]) << (2 * (k - 1 - (i - j)))
            yield i, self.lookup(kmer, rk, k, self.qhdist, i)

    @staticmethod
    def _trim_by_amount(n, left, right, minres):
        """shared/TrimRead.java:304-345 on lengths."""
        left, right = max(left, 0), max(right, 0)
        if n < 1:
            return 0, n
        minres = min(n, max(minres, 0))
        if left + right + minres > n:
            right, left = max(1, n - minres), 0
        return left + right, n - (left + right)

    def ktrim(self, read: bytes, pairnum: int):
        """Returns (x, id0, newLen).  bbduk/BBDukProcessorS.java:1806-1811, 1993-2140."""
        start, stop = self._span(len(read))
        return self._ktrim_span(read, pairnum, start, stop, self.a.ktrimLeft, self.a.ktrimRight)

    def ktrim_tips(self, read: bytes, pairnum: int):
        """ktrim=rl / ktrimtips.  Returns (x, id0, newLen, xRight, xLeft).  bbduk/BBDukProcessorS.java:1813-1985: a right
        pass over [start, len) and then a left pass over [0, stop) of the already right-trimmed read."""
        a, k, n = self.a, self.k, len(read)
        mid = n // 2 - (k - 1) // 2
        xr = xl = 0; id0 = -1; cur = read
        if a.ktrimRight:
            start = max(0, mid if a.restrictRight < 1 else n - a.restrictRight)
            xr, id0, nl = self._ktrim_span(cur, pairnum, start, n, False, True)
            cur = cur[:nl]
        if a.ktrimLeft:
            stop = min(len(cur), mid + k - 1 if a.restrictLeft < 1 else a.restrictLeft)
            xl, idl, nl = self._ktrim_span(cur, pairnum, 0, stop, True, False)
            cur = cur[len(cur) - nl:]
            if id0 < 0: id0 = idl
        return xr + xl, id0, len(cur), xr, xl

    def _ktrim_span(self, read: bytes, pairnum: int, start: int, stop: int, left: bool, right: bool):
        """ktrim(Read,start,stop) (:1993-2140) == ktrimTip(Read,start,stop,right,left) (:1832-1985)."""
        a, k, n = self.a, self.k, len(read)
        if n < max(1, min(k, self.mink) if self.useShortKmers else k) or not self.table:
            return 0, -1, n
        if (a.skipR1 and pairnum == 0) or (a.skipR2 and pairnum == 1):
            return 0, -1, n
        hits = list(self._main_hits(read, start, stop))
        BIG = 999999999
        found, id0 = len(hits), (hits[0][1] if hits else -1)
        minLoc = min((i - k + 1 for i, _ in hits), default=BIG)
        maxLoc = max((i for i, _ in hits), default=-1)
        minLocEx = minLoc + k if hits else BIG
        maxLocEx = maxLoc - k if hits else -1
        if self.useShortKmers and found == 0:
            if left:                                          # :2037-2069
                for i in range(start, min(k, stop)):
                    s = read[start:i + 1]
                    L = len(s)
                    if L < self.mink:
                        continue
                    km = 0
                    for c in s:
                        km = ((km << 2) | fwd_code(c)) & ((1 << (2 * k)) - 1)
                    rk = 0
                    for t, c in enumerate(s):
                        rk |= comp_code(c) << (2 * t)
                    vid = self.lookup(km, rk, L, self.qhdist2, i)
                    if vid > 0:
                        if id0 < 0: id0 = vid
                        minLoc = 0
                        minLocEx = min(minLocEx, i + 1)
                        maxLoc = max(maxLoc, i)
                        maxLocEx = max(maxLocEx, 0)
                        found += 1
            if right:                                         # :2072-2102
                for L in range(1, (k - 1 if stop >= k else stop) + 1):
                    i = stop - L
                    if L < self.mink:
                        continue
                    s = read[i:stop]
                    km = 0
                    for c in s:
                        km = (km << 2) | fwd_code(c)
                    rk = 0
                    for t, c in enumerate(s):
                        rk |= comp_code(c) << (2 * t)
                    rk &= (1 << (2 * k)) - 1
                    vid = self.lookup(km, rk, L, self.qhdist2, i)
                    if vid > 0:
                        if id0 < 0: id0 = vid
                        minLoc = i
                        minLocEx = min(minLocEx, n)
                        maxLoc = n - 1
                        maxLocEx = max(maxLocEx, i - 1)
                        found += 1
        if found == 0:
            return 0, -1, n
        self.scafReads[id0] += 1
        self.scafBases[id0] += n
        if a.trimPad != 0:
            mid = lambda lo, x, hi: lo if x < lo else (hi if x > hi else x)
            maxLoc = mid(0, maxLoc + a.trimPad, n)
            minLoc = mid(0, minLoc - a.trimPad, n)
            maxLocEx = mid(0, maxLocEx + a.trimPad, n)
            minLocEx = mid(0, minLocEx - a.trimPad, n)
        if left:
            leftLoc, rightLoc = (maxLocEx + 1 if a.ktrimExclusive else maxLoc + 1), n - 1
        else:
            leftLoc, rightLoc = 0, (minLocEx - 1 if a.ktrimExclusive else minLoc - 1)
        x, newLen = self._trim_by_amount(n, leftLoc, n - rightLoc - 1, 1)   # TrimRead.java:273-276
        return x, id0, newLen

    def kmask_read(self, read: bytes, pairnum: int):
        """Returns (cardinality, id0, mask) with mask = int whose bit b says base b is masked.
        bbduk/BBDukProcessorS.java:2149-2323 with kmaskFullyCovered=false."""
        a, k, n = self.a, self.k, len(read)
        if n < max(1, min(k, self.mink) if self.useShortKmers else k) or not self.table:
            return 0, -1, 0
        if (a.skipR1 and pairnum == 0) or (a.skipR2 and pairnum == 1):
            return 0, -1, 0
        if n < k:                                                                  # :2154
            return 0, -1, 0
        if a.kmaskFullyCovered:
            return self._kmask_fully_covered(read)
        bs = 0                                                                     # BitSet(n + trimPad + 1)
        def bset(lo, hi):
            nonlocal bs
            if hi > lo:
                bs |= ((1 << (hi - lo)) - 1) << lo
        minus, plus = k - 1 - a.trimPad, a.trimPad + 1
        start, stop = self._span(n)
        found, id0 = 0, -1
        for i, vid in self._main_hits(read, start, stop):                          # :2171-2200
            if id0 < 0: id0 = vid
            bset(max(0, i - minus), i + plus)
            found += 1
        if self.useShortKmers:                                                     # :2203-2291 (always, not only when found==0)
            for i in range(start, min(k, stop)):                                   # left side
                sq = read[start:i + 1]
                L = len(sq)
                if L < self.mink:
                    continue
                km = 0
                for c in sq:
                    km = ((km << 2) | fwd_code(c)) & ((1 << (2 * k)) - 1)
                rk = 0
                for t, c in enumerate(sq):
                    rk |= comp_code(c) << (2 * t)
                vid = self.lookup(km, rk, L, self.qhdist2, i)
                if vid > 0:
                    if id0 < 0: id0 = vid
                    bset(0, min(n, i + a.trimPad + 1))
                    found += 1
            for L in range(1, (k - 1 if stop >= k else stop) + 1):                 # right side: i = stop-1 .. > max(-1, stop-k)
                i = stop - L
                if L < self.mink:
                    continue
                sq = read[i:stop]
                km = 0
                for c in sq:
                    km = (km << 2) | fwd_code(c)
                rk = 0
                for t, c in enumerate(sq):
                    rk |= comp_code(c) << (2 * t)
                rk &= (1 << (2 * k)) - 1
                vid = self.lookup(km, rk, L, self.qhdist2, i)
                if vid > 0:
                    if id0 < 0: id0 = vid
                    bset(max(0, i - a.trimPad), n)
                    found += 1
        if found == 0:
            return 0, -1, 0
        self.scafReads[id0] += 1
        self.scafBases[id0] += n
        return bin(bs).count("1"), id0, bs & ((1 << n) - 1)

    def _kmask_fully_covered(self, read: bytes):
        """kmask with kmaskFullyCovered=true (:2163, 2193-2195, 2243-2245, 2286-2288): the bit set starts full and every k-mer
        position that does NOT match (looked up or not) clears its window; short k-mer positions that do not match clear
        their end of the read, the length mink-1 included (it is examined, len2>=minminlen, but never looked up).  Only
        clears happen, so the result is the complement of a union of ranges."""
        a, k, n = self.a, self.k, len(read)
        minus, plus = k - 1 - a.trimPad, a.trimPad + 1
        start, stop = self._span(n)
        cleared = [False] * (n + max(plus, 0) + 2)
        def clr(lo, hi):
            for b in range(max(lo, 0), min(hi, len(cleared))):
                cleared[b] = True
        found, id0 = 0, -1
        hit_at = dict(self._main_hits(read, start, stop))
        for i in range(max(start, k - 1), stop):                                   # if(i>=minlen): a hit, or a clear
            if i in hit_at:
                if id0 < 0: id0 = hit_at[i]
                found += 1
            else:
                clr(max(0, i - minus), i + plus)
        if self.useShortKmers:
            for i in range(start, min(k, stop)):                                   # left side
                L = i - start + 1
                if L < self.mink - 1:
                    continue
                vid = self._short_lookup_left(read[start:i + 1], L, i) if L >= self.mink else -1
                if vid > 0:
                    if id0 < 0: id0 = vid
                    found += 1
                else:
                    clr(0, min(n, i + a.trimPad + 1))
            for L in range(1, (k - 1 if stop >= k else stop) + 1):                 # right side
                i = stop - L
                if L < self.mink - 1:
                    continue
                vid = self._short_lookup(read[i:stop], L, i) if L >= self.mink else -1
                if vid > 0:
                    if id0 < 0: id0 = vid
                    found += 1
                else:
                    clr(max(0, i - a.trimPad), n)
        if found == 0:
            return 0, -1, 0
        self.scafReads[id0] += 1
        self.scafBases[id0] += n
        bs = sum(1 << b for b in range(n) if not cleared[b])
        return bin(bs).count("1"), id0, bs

    def _short_lookup_left(self, sub: bytes, L: int, qpos: int) -> int:
        km = 0
        for c in sub:
            km = ((km << 2) | fwd_code(c)) & ((1 << (2 * self.k)) - 1)
        rk = 0
        for t, c in enumerate(sub):
            rk |= comp_code(c) << (2 * t)
        return self.lookup(km, rk, L, self.qhdist2, qpos)

    def count_set_kmers(self, read: bytes, pairnum: int, maxBad: int):
        """Returns (found, id).  bbduk/BBDukProcessorS.java:1534-1593."""
        a, k, n = self.a, self.k, len(read)
        if n < k or not self.table:
            return 0, -1
        if (a.skipR1 and pairnum == 0) or (a.skipR2 and pairnum == 1):
            return 0, -1
        start, stop = self._span(n)
        found = 0
        for _, vid in self._main_hits(read, start, stop):
            if found == maxBad:
                self.scafReads[vid] += 1
                self.scafBases[vid] += n
                return found + 1, vid
            found += 1
        return found, -1

    def count_set_kmers_big(self, read: bytes, pairnum: int, maxBad: int):
        """Returns (found, id).  bbduk/BBDukProcessorS.java:1726-1804: k-mers longer than 31 are emulated by runs of
        consecutive matching 31-mers; positions that are not looked up neither extend nor close a run."""
        from itertools import groupby
        a, k, n = self.a, self.k, len(read)
        if n < self.kbig or not self.table:
            return 0, -1
        if (a.skipR1 and pairnum == 0) or (a.skipR2 and pairnum == 1):
            return 0, -1
        sub = self.kbig - k - 1
        start, stop = self._span(n)
        found = 0
        for hit, grp in groupby(self._main_lookups(read, start, stop), key=lambda t: t[1] > 0):
            if not hit:
                continue
            grp = list(grp)
            dif = grp[-1][0] - grp[0][0] - sub
            if dif > 0:
                old, found = found, found + dif
                if found > maxBad and old <= maxBad:            # both exits credit the run's last id; the early return
                    vid = grp[-1][1]                             # (:1763-1773) and the tail (:1783-1800) return the same count
                    self.scafReads[vid] += 1
                    self.scafBases[vid] += n
                    return found, vid
        return found, -1

    def find_best_match(self, read: bytes, pairnum: int, maxBad: int):
        """Returns (found, id).  bbduk/BBDukProcessorS.java:1659-1719 (length gate k, no early exit; the id with the most
        hits, the earliest-seen one among equals).  Only maxBad==0 is restated: with found<=maxBad the reference leaves its
        per-thread countArray dirty, which makes later answers depend on the thread's history."""
        a, k, n = self.a, self.k, len(read)
        assert maxBad == 0
        self.last_matches = []
        if n < k or not self.table:
            return 0, -1
        if (a.skipR1 and pairnum == 0) or (a.skipR2 and pairnum == 1):
            return 0, -1
        start, stop = self._span(n)
        counts = {}
        for _, vid in self._main_hits(read, start, stop):
            counts[vid] = counts.get(vid, 0) + 1               # dicts keep first-insertion order == idList order
        if not counts:
            return 0, -1
        best = max(counts.values())
        vid = next(i for i, c in counts.items() if c == best)
        self.last_matches = list(counts.items())               # idList / countList as rename() prints them (:2508-2522)
        self.scafReads[vid] += 1
        self.scafBases[vid] += n
        return sum(counts.values()), vid

    def ksplit_read(self, read: bytes):
        """Returns (trimmed, id0, leftmost, rightmost, split, newPairLength).  bbduk/BBDukProcessorS.java:2332-2506."""
        a, k, n = self.a, self.k, len(read)
        none = (0, -1, -1, -1, False, n)
        if n < max(1, min(k, self.mink) if self.useShortKmers else k) or not self.table or n < k:
            return none
        assert a.trimPad <= 0          # a positive trimPad can push rightmost past the read end, where subRead throws
        minus, plus = k - 1 - a.trimPad, a.trimPad
        start, stop = self._span(n)
        hits = list(self._main_hits(read, start, stop))
        leftmost, rightmost, id0 = 1 << 31, -1, -1
        if hits:
            id0 = hits[0][1]
            leftmost = max(0, hits[0][0] - minus)
            rightmost = hits[-1][0] + plus
        found = len(hits)
        if self.useShortKmers and id0 == -1:
            # right side (:2391-2431): suffix read[i:stop] of every length >= mink, longest i first in the loop but all hits count
            for i in range(stop - 1, max(-1, stop - k), -1):
                L = stop - i
                if L < self.mink:
                    continue
                vid = self._short_lookup(read[i:stop], L, i)
                if vid > 0:
                    if id0 < 0: id0 = vid
                    leftmost = min(leftmost, max(0, i - a.trimPad)); rightmost = n - 1; found += 1
            if id0 == -1:                                       # left side (:2434-2473)
                for i in range(start, min(k, stop)):
                    L = i - start + 1
                    if L < self.mink:
                        continue
                    vid = self._short_lookup(read[start:i + 1], L, i)
                    if vid > 0:
                        if id0 < 0: id0 = vid
                        leftmost = 0; rightmost = max(rightmost, i + a.trimPad); found += 1
        if found == 0:
            return none
        self.scafReads[id0] += 1
        self.scafBases[id0] += n
        if leftmost == 0:
            x, n1 = self._trim_by_amount(n, rightmost + 1, n - (n - 1) - 1, 1)
            return n - n1, id0, leftmost, rightmost, False, n1
        if rightmost == n - 1:
            x, n1 = self._trim_by_amount(n, 0, n - (leftmost - 1) - 1, 1)
            return n - n1, id0, leftmost, rightmost, False, n1
        n2 = (n - 1) - (rightmost + 1)                          # subRead(rightmost+1, n-1): copyOfRange excludes index n-1
        x, n1 = self._trim_by_amount(n, 0, n - (leftmost - 1) - 1, 1)
        return n - (n1 + n2), id0, leftmost, rightmost, True, n1 + n2

    def _short_lookup(self, sub: bytes, L: int, qpos: int) -> int:
        kmer = rk = 0
        for j, c in enumerate(sub):
            kmer |= fwd_code(c) << (2 * (L - 1 - j))
            rk |= comp_code(c) << (2 * j)
        return self.lookup(kmer, rk, L, self.qhdist2, qpos)

    def num_valid_kmers(self, read: bytes, k: int) -> int:
        """stream/Read.java:1673-1683."""
        ln = counted = 0
        for c in read:
            ln = ln + 1 if defined(c) else 0
            if ln >= k:
                counted += 1
        return counted

    def count_covered_bases(self, read: bytes, pairnum: int, minCovered: int):
        """Returns (found, id).  bbduk/BBDukProcessorS.java:1602-1651."""
        a, k, n = self.a, self.k, len(read)
        if n < k or not self.table:
            return 0, -1
        if (a.skipR1 and pairnum == 0) or (a.skipR2 and pairnum == 1):
            return 0, -1
        start, stop = self._span(n)
        found, last = 0, -1
        for i, vid in self._main_hits(read, start, stop):
            found += min(k, i - last)
            last = i
            if found >= minCovered:
                self.scafReads[vid] += 1
                self.scafBases[vid] += n
                return found, vid
        return found, -1

    # ---------------------------------------------------------------- pair stage
    def process_pair(self, r1: bytes, r2):
        """bbduk/BBDukProcessorS.java:807-818, 948-1093, 1431-1443.  Returns [(a, id, flags), ...] per mate."""
        a, C = self.a, self.counters
        has2 = r2 is not None
        l1, l2 = len(r1), (len(r2) if has2 else 0)
        pc = 2 if has2 else 1
        import numpy as np
        f32 = np.float32
        minlen1 = int(max(f32(l1) * f32(a.minLenFraction), f32(a.minReadLength)))
        minlen2 = int(max(f32(l2) * f32(a.minLenFraction), f32(a.minReadLength)))
        C[0] += pc; C[1] += l1 + l2
        d1 = d2 = remove = False
        n1, n2 = l1, l2
        res = []

        def tf(d, n):
            """setDiscarded / isDiscarded under trimfailuresto1bp (:1464-1482): cut to one base, and 'discarded' = one base long"""
            if not a.trimFailuresTo1bp:
                return d, n
            if d and n > 1:
                n = 1
            return n == 1, n
        if self.table and a.ktrimN and not (a.ktrimLeft or a.ktrimRight):              # :984-998, 1009-1016
            x1, i1, m1 = self.kmask_read(r1, 0)
            xsum, rkt = x1, int(x1 > 0)
            d1 = n1 < minlen1
            x2, i2, m2 = 0, -1, 0
            if has2:
                x2, i2, m2 = self.kmask_read(r2, 1)
                xsum += x2; rkt += int(x2 > 0)
                d2 = n2 < minlen2
            d1, n1 = tf(d1, n1)
            if has2: d2, n2 = tf(d2, n2)
            if (self.rieb and (d1 or d2)) or (d1 and (not has2 or d2)):
                remove = True
            C[3] += xsum; C[2] += rkt
            res = [(x1, i1)] + ([(x2, i2)] if has2 else [])
            self.masks += [m1] + ([m2] if has2 else [])
        elif self.table and a.ksplit:                                                  # :999-1013
            assert not has2
            x1, i1, lm, rm, split, n1 = self.ksplit_read(r1)
            C[3] += x1; C[2] += int(x1 > 0)
            remove = split                                                             # remove=(r1.mate!=null): both pieces go to outm
            res = [(x1, i1)]
            self.splits.append((lm, rm))
        elif self.table and (a.ktrimLeft or a.ktrimRight):
            tips = a.ktrimLeft and a.ktrimRight                                        # :771, 954-967
            if tips:
                x1, i1, n1, xr1, xl1 = self.ktrim_tips(r1, 0)
            else:
                x1, i1, n1 = self.ktrim(r1, 0)
            xsum, rkt = x1, int(x1 > 0)
            d1 = n1 < minlen1
            x2 = 0; i2 = -1
            xr2 = xl2 = 0
            if has2:
                if tips:
                    x2, i2, n2, xr2, xl2 = self.ktrim_tips(r2, 1)
                else:
                    x2, i2, n2 = self.ktrim(r2, 1)
                xsum += x2; rkt += int(x2 > 0)
                d2 = n2 < minlen2
            rl1, rl2 = n1, n2                                                          # rlen1 / rlen2: what ktrim left
            d1, n1 = tf(d1, n1)
            if has2: d2, n2 = tf(d2, n2)
            if (self.rieb and (d1 or d2)) or (d1 and (not has2 or d2)):
                xsum += rl1 + rl2; rkt = pc; remove = True
            elif a.ktrimRight and a.trimPairsEvenly and xsum > 0 and has2 and n1 != n2:        # :1021-1031
                if n1 > n2:
                    x, n1 = self._trim_by_amount(n1, 0, n1 - (n2 - 1) - 1, 1); x1 += x
                    if tips: xr1 += x
                else:
                    x, n2 = self._trim_by_amount(n2, 0, n2 - (n1 - 1) - 1, 1); x2 += x
                    if tips: xr2 += x
                if rkt < 2: rkt += 1
                xsum += x
            C[3] += xsum; C[2] += rkt
            res = [(x1, i1)] + ([(x2, i2)] if has2 else [])
            if tips:
                self.tips += [(xr1, xl1)] + ([(xr2, xl2)] if has2 else [])
        elif self.table and a.minCoveredFraction > 0:                                  # :1038-1049
            import math
            mc1 = int(math.ceil(float(f32(a.minCoveredFraction) * f32(l1))))
            f1, i1 = self.count_covered_bases(r1, 0, mc1)
            d1 = f1 >= mc1
            res = [(f1, i1)]
            if has2:
                mc2 = int(math.ceil(float(f32(a.minCoveredFraction) * f32(l2))))
                f2, i2 = self.count_covered_bases(r2, 1, mc2)
                d2 = f2 >= mc2
                res.append((f2, i2))
            d1, n1 = tf(d1, n1)
            if has2: d2, n2 = tf(d2, n2)
            if (self.rieb and (d1 or d2)) or (d1 and (not has2 or d2)):
                remove = True
                C[4] += pc; C[5] += l1 + l2
        elif self.table:
            mb1 = mb2 = a.maxBadKmers0                                                 # :1055-1062
            if a.minKmerFraction != 0:
                vk1, vk2 = self.num_valid_kmers(r1, self.keff), (self.num_valid_kmers(r2, self.keff) if has2 else 0)
                mb1 = max(a.maxBadKmers0, int(f32(vk1 - 1) * f32(a.minKmerFraction)))
                mb2 = max(a.maxBadKmers0, int(f32(vk2 - 1) * f32(a.minKmerFraction)))
            count = self.find_best_match if a.findBestMatch else (self.count_set_kmers_big if self.kbig > self.k else self.count_set_kmers)
            f1, i1 = count(r1, 0, mb1)                                                 # :1064-1076
            d1 = i1 > 0 if a.findBestMatch else f1 > mb1
            res = [(f1, i1)]
            if has2:
                f2, i2 = count(r2, 1, mb2)
                d2 = i2 > 0 if a.findBestMatch else f2 > mb2
                res.append((f2, i2))
            d1, n1 = tf(d1, n1)
            if has2: d2, n2 = tf(d2, n2)
            if (self.rieb and (d1 or d2)) or (d1 and (not has2 or d2)):
                remove = True
                C[4] += pc; C[5] += l1 + l2
        else:
            res = [(0, -1)] * pc
        if remove and a.trimFailuresTo1bp:
            remove = False                                                             # :1431: flagged and counted, not evicted
        if remove:
            C[8] += pc; C[9] += n1 + n2
        else:
            C[6] += pc; C[7] += n1 + n2
        fl = [(FLAG_DISCARDED if d1 else 0) | (FLAG_REMOVED if remove else 0)]
        if has2:
            fl.append((FLAG_DISCARDED if d2 else 0) | (FLAG_REMOVED if remove else 0))
        return [(r[0], r[1], f) for r, f in zip(res, fl)]

    def process_batch(self, reads, paired: bool):
        out = []
        if paired:
            assert len(reads) % 2 == 0
            for i in range(0, len(reads), 2):
                out += self.process_pair(reads[i], reads[i + 1])
        else:
            for r in reads:
                out += self.process_pair(r, None)
        return out

    def all_counters(self):
        return list(self.counters) + list(self.scafReads) + list(self.scafBases)
