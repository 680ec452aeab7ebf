// bbduk_host.cpp -- host-side mirror of the reference's BBDukParser / BBDukLoader+BBDukIndexMod roles
// (see include/bbduk_host.h).  Product code; independent of oracle/.
//
// The index builder is deliberately NOT the reference's algorithm (7 threads x recursive mutate() x
// hash-insert with first-writer-wins): it enumerates every (key, scaffold id) candidate into a flat vector
// and reduces it with sort + unique keeping the smallest id, which is the same map because scaffold ids
// ascend in file order and every key a scaffold produces carries that scaffold's id
// (bbduk/BBDukLoader.java:219-251; kmer/HashArray.java:221-239).
#include "../../include/bbduk_host.h"
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct KV { uint64_t key; int32_t id; };

struct Parsed {                      // bbduk/BBDukParser.java fields (subset that reaches this path)
    int k = 27; bool setk = false;   // :163
    int mink = -1;                   // :1230
    int hdist = 0, hdist2 = -1, edist = 0, edist2 = -1, qhdist = 0, qhdist2 = -1;
    bool maskMiddle = true; int midMaskLen = 0;   // :1091
    bool rcomp = true;               // :1208
    bool forbidNs = false;
    bool ktrimLeft = false, ktrimRight = false, ktrimN = false;
    int maxBadKmers0 = 0;            // :1232
    int minReadLength = 10;          // :437
    float minLenFraction = 0.f;      // :439
    bool requireBothBad = false;
    int trimPad = 0; bool ktrimExclusive = false;
    int restrictLeft = 0, restrictRight = 0;
    bool skipR1 = false, skipR2 = false;
    int minSkip = 1, maxSkip = 1;
    bool trimPairsEvenly = false;    // tpe
    int qSkip = 1, speed = 0;        // :1373-1377
    float minKmerFraction = 0.f, minCoveredFraction = 0.f;   // :1234-1236
    bool ksplit = false, findBestMatch = false, rename = false;   // :1313, 1323, 1367
    bool kmaskFullyCovered = false;                          // :1317
    bool trimFailuresTo1bp = false;                          // BBDukParser.java:774, 1359
    int kbig = -1;                                           // :1226 (derived: the requested k when it exceeds 31)
    std::vector<std::string> ref, literal;
    // derived (:130-312)
    int minlen = 0, minlen2 = 0;
    bool useShortKmers = false;
    int64_t middleMask = -1;
};

}  // namespace

struct bbduk_host {
    Parsed p;
    std::vector<std::vector<uint8_t>> scaffolds;     // id = index+1
    std::vector<std::string> names;                  // scaffoldNames[id] (BBDukLoader.java:224, 275): FASTA header, or the id as text
    std::vector<std::string> refNames;               // one per ref= file, then "literal" (BBDukParser.java:330-340); refstats= groups by these
    std::vector<int> refScafCounts;                  // scaffolds each of them contributed (BBDukLoader.java:224, 275)
    std::vector<KV> cand;
    std::vector<int64_t> keys;
    std::vector<int32_t> vals;
    bool built = false;
};

namespace {

bool parse_bool(const std::string& b, bool& ok) {    // parse/Parse.java parseBoolean
    ok = true;
    if (b.empty()) return true;
    std::string s = b; for (auto& c : s) c = (char)tolower(c);
    if (s == "t" || s == "true" || s == "1") return true;
    if (s == "f" || s == "false" || s == "0") return false;
    ok = false; return false;
}
bool parse_int(const std::string& b, int& v) {
    if (b.empty()) return false;
    char* e = nullptr; long x = strtol(b.c_str(), &e, 10);
    if (*e) return false; v = (int)x; return true;
}

int derive(Parsed& p, std::string& err) {            // bbduk/BBDukParser.java:130-312, same order of evaluation
    if (p.hdist2 == -1) p.hdist2 = p.hdist;
    if (p.qhdist2 == -1) p.qhdist2 = p.qhdist;
    if (p.edist2 == -1) p.edist2 = p.edist;
    if (p.edist > 1 || p.edist2 > 1) { err = "edist>1 is not supported by this path (the expansion is quadratic per k-mer)"; return BBDUK_ERR_ARG; }
    p.hdist = std::max(p.edist, p.hdist);
    p.hdist2 = std::max(p.edist2, p.hdist2);
    p.minSkip = std::max(1, std::min(p.minSkip, p.maxSkip));
    p.maxSkip = std::max(p.minSkip, p.maxSkip);
    p.forbidNs = (p.forbidNs || p.hdist < 1);
    p.restrictLeft = std::max(p.restrictLeft, 0);
    p.restrictRight = std::max(p.restrictRight, 0);
    if (!p.setk) p.k = 27;
    if (p.k < 1) { err = "k must be positive"; return BBDUK_ERR_ARG; }
    p.kbig = (p.k > 31 ? p.k : -1);                                      // :164-165
    p.k = std::min(p.k, 31);
    if ((p.ktrimLeft || p.ktrimRight || p.ktrimN || p.ksplit) && p.kbig > p.k) p.kbig = p.k;   // :207-215 "K has been reduced"
    if ((p.speed > 0 || p.qSkip > 1) && p.kbig > p.k) p.kbig = p.k;     // :217-223
    if (p.kbig > p.k) { p.maskMiddle = false; p.minSkip = p.maxSkip = 0; }   // :237-243, before minlen2 is derived
    if (p.maskMiddle) p.midMaskLen = (p.midMaskLen > 0 ? p.midMaskLen : 2 - (p.k & 1));
    else p.midMaskLen = 0;
    p.mink = std::min(p.mink, p.k);
    p.minlen = p.k - 1;
    p.minlen2 = (p.maskMiddle ? (p.k - p.midMaskLen) / 2 : p.k);        // before mink turns maskMiddle off
    if (p.mink > 0 && p.mink < p.k) p.useShortKmers = true;
    if (p.useShortKmers && p.maskMiddle) { p.maskMiddle = false; p.midMaskLen = 0; }
    p.findBestMatch = p.rename || p.findBestMatch;                                                   // :153
    const bool kfilter = !(p.ktrimLeft || p.ktrimRight || p.ktrimN || p.ksplit);                     // :298
    if (p.rename && kfilter && p.minCoveredFraction > 0.f) { err = "rename with mincoveredfraction is not supported by this path"; return BBDUK_ERR_ARG; }   // :1049-1052
    if (p.findBestMatch && kfilter && p.kbig > p.k) { err = "K must be less than 32 in 'findBestMatch' mode"; return BBDUK_ERR_ARG; }   // :299
    if (p.findBestMatch && kfilter && (p.maxBadKmers0 != 0 || p.minKmerFraction != 0.f)) {
        err = "findbestmatch with maxbadkmers>0 or minkmerfraction>0 is not supported by this path (the reference's answer depends on thread history there)";
        return BBDUK_ERR_ARG;
    }
    if (p.ksplit && p.trimPad > 0) { err = "ksplit with a positive trimpad is not supported by this path"; return BBDUK_ERR_ARG; }
    if (p.useShortKmers && !(p.ktrimLeft || p.ktrimRight || p.ktrimN || p.ksplit)) { err = "Setting mink also requires setting a ktrim mode, such as 'r' or 'l'"; return BBDUK_ERR_ARG; }
    // ktrim=rl (tips): both flags stay set, the device runs the two passes (BBDUK_MODE_KTRIM_TIPS)
    if (p.maskMiddle) {
        if (!(p.k > p.midMaskLen + 1)) { err = "k too small for maskmiddle"; return BBDUK_ERR_ARG; }
        const int bits = p.midMaskLen * 2;
        const int shift = ((p.k - p.midMaskLen) / 2) * 2;
        p.middleMask = (int64_t)~((~(~0ULL << bits)) << shift);
    } else p.middleMask = -1;
    if (p.hdist < 0 || p.hdist > 3 || p.hdist2 < 0 || p.hdist2 > 3 || p.qhdist < 0 || p.qhdist > 3 || p.qhdist2 < 0 || p.qhdist2 > 3) {
        err = "hamming distance must be between 0 and 3"; return BBDUK_ERR_ARG;
    }
    return BBDUK_OK;
}

inline int code_of(uint8_t b) {      // dna/AminoAcid.java:1284-1298 baseToNumber; -1 undefined
    switch (b) {
        case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3; default: return -1;
    }
}
inline uint64_t rcomp_bits(uint64_t v, int len) {   // digit-reversal with complement
    uint64_t x = ~v;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);
    return x >> (64 - 2 * len);
}

// all sequences within `dist` substitutions of v (duplicates allowed; reduced later by sort/unique)
void expand(uint64_t v, int len, int dist, std::vector<uint64_t>& out) {
    out.clear();
    out.push_back(v);
    size_t lo = 0;
    for (int d = 0; d < dist; d++) {
        const size_t hi = out.size();
        for (size_t q = lo; q < hi; q++) {
            const uint64_t b = out[q];
            for (int i = 0; i < len; i++) {
                const uint64_t cleared = b & ~(3ULL << (2 * i));
                const uint64_t cur = (b >> (2 * i)) & 3ULL;
                for (uint64_t j = 0; j < 4; j++) if (j != cur) out.push_back(cleared | (j << (2 * i)));
            }
        }
        lo = hi;
    }
}

void emit(bbduk_host* h, uint64_t v, int len, int dist, int id, std::vector<uint64_t>& tmp) {
    const Parsed& p = h->p;
    expand(v, len, dist, tmp);
    const uint64_t lmask = 1ULL << (2 * len);
    for (uint64_t t : tmp) {
        const uint64_t r = rcomp_bits(t, len);
        const uint64_t mx = p.rcomp ? std::max(t, r) : t;
        h->cand.push_back(KV{(mx & (uint64_t)p.middleMask) | lmask, id});
    }
}

// Reference-side EDIT expansion (bbduk/BBDukIndexMod.java:383-445 mutate with editDistance>0): substitutions, deletions (the
// base behind the k-mer moves in: extraBase) and insertions (the last base falls out and becomes the next level's extraBase).
// indels: the reference tests the GLOBAL editDistance there (:415), also for the short k-mers that run with editDistance2.
void edits(uint64_t kmer, int len, int dist, int64_t extraBase, bool indels, std::vector<uint64_t>& out) {
    out.push_back(kmer);
    if (dist <= 0) return;
    const int d2 = dist - 1;
    for (uint64_t j = 0; j < 4; j++)                                  // Sub (:404-412)
        for (int i = 0; i < len; i++) {
            const uint64_t t = (kmer & ~(3ULL << (2 * i))) | (j << (2 * i));
            if (t != kmer) edits(t, len, d2, extraBase, indels, out);
        }
    if (!indels) return;
    if (extraBase >= 0 && extraBase <= 3)                             // Del (:415-425)
        for (int i = 1; i < len; i++) {
            const uint64_t left = ~0ULL << (2 * i), right = ~left;
            const uint64_t t = (kmer & left) | ((kmer << 2) & right) | (uint64_t)extraBase;
            if (t != kmer) edits(t, len, d2, -1, indels, out);
        }
    const int64_t eb2 = (int64_t)(kmer & 3ULL);                       // Ins (:427-439)
    for (int i = 1; i < len; i++) {
        const uint64_t left = ~0ULL << (2 * i), right = ~left;
        const uint64_t t0 = (kmer & left) | ((kmer & right) >> 2);
        for (uint64_t j = 0; j < 4; j++) {
            const uint64_t t = t0 | (j << (2 * (i - 1)));
            if (t != kmer) edits(t, len, d2, eb2, indels, out);
        }
    }
}
// addToMap's dispatch (:351-373): hdist==0 -> the k-mer itself; editDistance>0 -> mutate(edist, extraBase); else mutate(hdist)
void emit_any(bbduk_host* h, uint64_t v, int len, int hdist, int edist, int64_t extraBase, int id, std::vector<uint64_t>& tmp) {
    if (hdist == 0 || edist <= 0) { emit(h, v, len, hdist, id, tmp); return; }
    const Parsed& p = h->p;
    tmp.clear();
    edits(v, len, edist, extraBase, p.edist > 0, tmp);
    std::sort(tmp.begin(), tmp.end()); tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    const uint64_t lmask = 1ULL << (2 * len);
    for (uint64_t t : tmp) {
        const uint64_t r = rcomp_bits(t, len);
        const uint64_t mx = p.rcomp ? std::max(t, r) : t;
        h->cand.push_back(KV{(mx & (uint64_t)p.middleMask) | lmask, id});
    }
}

}  // namespace

extern "C" int bbduk_host_parse(const char* args, bbduk_host** out, char* errbuf, int errlen) {
    auto seterr = [&](const std::string& m) { if (errbuf && errlen > 0) { snprintf(errbuf, (size_t)errlen, "%s", m.c_str()); } };
    if (!args || !out) { seterr("null argument"); return BBDUK_ERR_ARG; }
    *out = nullptr;
    bbduk_host* h = new bbduk_host();
    Parsed& p = h->p;
    std::istringstream ss(args);
    std::string tok;
    while (ss >> tok) {
        const size_t eq = tok.find('=');
        std::string a = tok.substr(0, eq), b = (eq == std::string::npos ? "" : tok.substr(eq + 1));
        for (auto& c : a) c = (char)tolower(c);
        bool ok = true; int iv = 0;
        auto need_int = [&](int& dst) { if (parse_int(b, iv)) dst = iv; else ok = false; };
        auto need_bool = [&](bool& dst) { bool o; bool v = parse_bool(b, o); if (o) dst = v; else ok = false; };
        if (a == "k") { need_int(p.k); p.setk = true; }
        else if (a == "mink" || a == "kmin") need_int(p.mink);
        else if (a == "hdist" || a == "hammingdistance") need_int(p.hdist);
        else if (a == "hdist2" || a == "hammingdistance2") need_int(p.hdist2);
        else if (a == "qhdist" || a == "queryhammingdistance") need_int(p.qhdist);
        else if (a == "qhdist2" || a == "queryhammingdistance2") need_int(p.qhdist2);
        else if (a == "edist" || a == "edits" || a == "editdistance") need_int(p.edist);
        else if (a == "edist2" || a == "edits2" || a == "editdistance2") need_int(p.edist2);
        else if (a == "mm" || a == "maskmiddle") {
            if (b.empty() || isalpha((unsigned char)b[0])) need_bool(p.maskMiddle);
            else { need_int(p.midMaskLen); p.maskMiddle = p.midMaskLen > 0; }
        }
        else if (a == "rcomp") need_bool(p.rcomp);
        else if (a == "forbidns" || a == "forbidn" || a == "fn") need_bool(p.forbidNs);
        else if (a == "ktrim") {
            std::string v = b; for (auto& c : v) c = (char)tolower(c);
            if (v == "rl" || v == "lr" || v == "tips") { p.ktrimLeft = p.ktrimRight = true; p.ktrimN = p.ksplit = false; }
            else if (v == "l" || v == "left") { p.ktrimLeft = true; p.ktrimRight = false; p.ktrimN = p.ksplit = false; }
            else if (v == "r" || v == "right") { p.ktrimLeft = false; p.ktrimRight = true; p.ktrimN = p.ksplit = false; }
            else if (v == "n" || (v.size() == 1 && v != "t" && v != "f")) { p.ktrimLeft = p.ktrimRight = p.ksplit = false; p.ktrimN = true; }   // the symbol stays with the caller
            else if (v == "f" || v == "false") { p.ktrimLeft = p.ktrimRight = false; }
            else { seterr("Invalid setting for ktrim - values must be f (false), l (left), r (right), or n"); delete h; return BBDUK_ERR_ARG; }
        }
        else if (a == "trimtips" || a == "ktrimtips") {             // BBDukParser.java:626-632
            if (!b.empty()) { int v = 0; need_int(v); if (ok) { p.ktrimLeft = p.ktrimRight = true; p.ktrimN = p.ksplit = false; p.restrictLeft = p.restrictRight = v; } }
        }
        else if (a == "kmask" || a == "mask") {                      // BBDukParser.java:635-651: lc | t | a symbol | f
            std::string v = b; for (auto& c : v) c = (char)tolower(c);
            if (v == "f" || v == "false") p.ktrimN = false;
            else { p.ktrimN = true; p.ktrimLeft = p.ktrimRight = p.ksplit = false; }
        }
        else if (a == "ksplit") { bool x = false; need_bool(x); if (ok) { p.ksplit = x; if (x) p.ktrimLeft = p.ktrimRight = p.ktrimN = false; } }   // :599-603
        else if (a == "findbestmatch" || a == "fbm") need_bool(p.findBestMatch);                                                                       // :594-595
        else if (a == "rename") need_bool(p.rename);                                                                                                   // :694-695
        else if (a == "ktrimn") { need_bool(p.ktrimN); if (ok) p.ktrimLeft = p.ktrimRight = !p.ktrimN; }
        else if (a == "kmaskfullycovered" || a == "maskfullycovered" || a == "mfc") {
            need_bool(p.kmaskFullyCovered);
        }
        else if (a == "kfilter") { bool x = false; need_bool(x); if (x) { p.ktrimLeft = p.ktrimRight = false; p.ktrimN = p.ksplit = false; } }
        else if (a == "maxbadkmers" || a == "mbk") need_int(p.maxBadKmers0);
        else if (a == "minhits" || a == "minkmerhits" || a == "mkh") { need_int(p.maxBadKmers0); p.maxBadKmers0 -= 1; }
        else if (a == "ml" || a == "minlen" || a == "minlength") need_int(p.minReadLength);
        else if (a == "mlf" || a == "minlenfrac" || a == "minlenfraction" || a == "minlengthfraction") { char* e; p.minLenFraction = strtof(b.c_str(), &e); ok = !b.empty() && !*e; }
        else if (a == "trimfailures" || a == "trimfailuresto1bp") need_bool(p.trimFailuresTo1bp);                   // BBDukParser.java:773-774
        else if (a == "requirebothbad" || a == "rbb") need_bool(p.requireBothBad);
        else if (a == "removeifeitherbad" || a == "rieb") { bool x = true; need_bool(x); p.requireBothBad = !x; }
        else if (a == "trimextra" || a == "trimpad" || a == "tp") need_int(p.trimPad);
        else if (a == "ktrimexclusive") need_bool(p.ktrimExclusive);
        else if (a == "restrictleft") need_int(p.restrictLeft);
        else if (a == "restrictright") need_int(p.restrictRight);
        else if (a == "skipr1") need_bool(p.skipR1);
        else if (a == "skipr2") need_bool(p.skipR2);
        else if (a == "tpe" || a == "trimpairsevenly") need_bool(p.trimPairsEvenly);
        else if (a == "qskip") need_int(p.qSkip);
        else if (a == "minkmerfraction" || a == "minfraction" || a == "mkf" || a == "mincoveredfraction" || a == "mincovfraction" || a == "mcf") {
            char* e; float v = strtof(b.c_str(), &e); ok = !b.empty() && !*e;
            if (ok && v > 1.f) { seterr(a + " must range from 0 to 1; value=" + b); delete h; return BBDUK_ERR_ARG; }    // :284,287
            if (v < 0.f) v = 0.f;                                                                                      // :283,286
            if (a == "minkmerfraction" || a == "minfraction" || a == "mkf") p.minKmerFraction = v; else p.minCoveredFraction = v;
        }
        else if (a == "speed") { need_int(p.speed); if (ok && (p.speed < 0 || p.speed > 16)) { seterr("Speed range is 0 to 16.  Value: " + b); delete h; return BBDUK_ERR_ARG; } }
        else if (a == "maxskip" || a == "maxrskip" || a == "mxs") need_int(p.maxSkip);
        else if (a == "minskip" || a == "minrskip" || a == "mns") need_int(p.minSkip);
        else if (a == "skip" || a == "refskip" || a == "rskip") { need_int(p.minSkip); p.maxSkip = p.minSkip; }
        else if (a == "ref" || a == "adapters") { std::stringstream rs(b); std::string r; while (std::getline(rs, r, ',')) if (!r.empty()) p.ref.push_back(r); }
        else if (a == "literal") { std::stringstream rs(b); std::string r; while (std::getline(rs, r, ',')) if (!r.empty()) p.literal.push_back(r); }
        else { seterr("Unknown parameter " + tok); delete h; return BBDUK_ERR_ARG; }      // BBDukParser.java:870-872
        if (!ok) { seterr("Bad value in " + tok); delete h; return BBDUK_ERR_ARG; }
    }
    std::string err;
    const int rc = derive(p, err);
    if (rc != BBDUK_OK) { seterr(err); delete h; return rc; }
    *out = h;
    return BBDUK_OK;
}

extern "C" void bbduk_host_destroy(bbduk_host* h) { delete h; }

extern "C" int bbduk_host_add_ref(bbduk_host* h, const uint8_t* seq, int64_t len) {
    if (!h || len < 0 || (len > 0 && !seq) || h->built) return BBDUK_ERR_ARG;
    h->scaffolds.emplace_back(seq, seq + len);
    h->names.push_back(std::to_string(h->scaffolds.size()));
    return BBDUK_OK;
}

static int load_stream(bbduk_host* h, FILE* f) {
    std::vector<uint8_t> seq; bool have = false; int nrec = 0; int ch; bool bol = true, inhdr = false;
    std::string name;                                    // header of the record being read (without '>')
    auto flush = [&]() {                                 // scaffoldNames.add(r1.id==null ? id.toString() : r1.id), BBDukLoader.java:224
        h->scaffolds.push_back(seq);
        h->names.push_back(name.empty() ? std::to_string(h->scaffolds.size()) : name);
        nrec++;
    };
    while ((ch = fgetc(f)) != EOF) {
        if (inhdr) { if (ch == '\n') { inhdr = false; bol = true; } else if (ch != '\r') name.push_back((char)ch); continue; }
        if (bol && ch == '>') {
            if (have && !seq.empty()) flush();           // records shorter than 1 base are dropped
            have = true; seq.clear(); name.clear(); inhdr = true; continue;
        }
        if (ch == '\n' || ch == '\r') { bol = true; continue; }
        bol = false;
        if (ch > '\r') seq.push_back((uint8_t)ch);
    }
    if (have && !seq.empty()) flush();
    return nrec;
}

extern "C" int bbduk_host_load_fasta(bbduk_host* h, const char* path) {
    if (!h || !path || h->built) return BBDUK_ERR_ARG;
    const std::string p(path);
    const bool gz = p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0;
    if (p.find('\'') != std::string::npos) return BBDUK_ERR_ARG;
    FILE* f = gz ? popen(("gzip -dc '" + p + "'").c_str(), "r") : fopen(path, "rb");
    if (!f) return BBDUK_ERR_ARG;
    const int n = load_stream(h, f);
    if (gz) { if (pclose(f) != 0) return BBDUK_ERR_ARG; } else fclose(f);
    return n;
}

extern "C" int bbduk_host_load_refs(bbduk_host* h, const char* resource_dir) {
    if (!h || h->built) return BBDUK_ERR_ARG;
    const std::string dir = resource_dir ? resource_dir : ".";
    int total = 0;
    for (const std::string& r : h->p.ref) {            // bbduk/BBDukParser.java:899-936 modifyRefPath
        std::string path = r;
        FILE* t = fopen(path.c_str(), "rb");
        if (t) fclose(t);
        else {
            std::string low = r; for (auto& c : low) c = (char)tolower(c);
            if (low == "adapters") path = dir + "/adapters.fa";
            else if (low == "phix") path = dir + "/phix2.fa.gz";
            else return BBDUK_ERR_ARG;
        }
        const int n = bbduk_host_load_fasta(h, path.c_str());
        if (n < 0) return n;
        total += n;
        h->refNames.push_back(path); h->refScafCounts.push_back(n);
    }
    if (!h->p.literal.empty()) { h->refNames.push_back("literal"); h->refScafCounts.push_back((int)h->p.literal.size()); }
    for (const std::string& l : h->p.literal) {                      // BBDukLoader.java:272-277: a literal's name is its id
        h->scaffolds.emplace_back(l.begin(), l.end()); h->names.push_back(std::to_string(h->scaffolds.size())); total++;
    }
    return total;
}

extern "C" int bbduk_host_num_refs(const bbduk_host* h) { return h ? (int)h->refNames.size() : BBDUK_ERR_ARG; }
extern "C" int bbduk_host_ref_info(const bbduk_host* h, int32_t r, const char** name, int32_t* num_scaffolds) {
    if (!h || r < 0 || r >= (int)h->refNames.size()) return BBDUK_ERR_ARG;
    if (name) *name = h->refNames[(size_t)r].c_str();
    if (num_scaffolds) *num_scaffolds = h->refScafCounts[(size_t)r];
    return BBDUK_OK;
}

extern "C" int64_t bbduk_host_build_index(bbduk_host* h) {
    if (!h || h->built) return BBDUK_ERR_ARG;
    const Parsed& p = h->p;
    const int k = p.k;
    const uint64_t mask = (2 * k > 63) ? ~0ULL : ~(~0ULL << (2 * k));
    std::vector<uint64_t> tmp;
    for (size_t s = 0; s < h->scaffolds.size(); s++) {
        const std::vector<uint8_t>& b = h->scaffolds[s];
        const int id = (int)s + 1;
        const int64_t n = (int64_t)b.size();
        if (n < k) continue;
        // reference-side skipping (BBDukLoader.java:397, 417, 432-449): with skip > 1 only every skip-th k-mer of a run of
        // defined bases is stored (len%skip==0), the short k-mers of the ends included
        const int heur = n > 20000000 ? k : n > 5000000 ? 11 : n > 500000 ? 2 : 0;
        const int skip = std::max(p.minSkip, std::min(p.maxSkip, heur));
        uint64_t fwd = 0; int64_t run = 0;               // run = defined bases ending here
        for (int64_t i = 0; i < n; i++) {
            const int c = code_of(b[i]);
            fwd = ((fwd << 2) | (uint64_t)(c < 0 ? 0 : c)) & mask;
            run = (c < 0) ? 0 : run + 1;
            if (run < k) continue;
            if (skip > 1 && run % skip != 0) continue;
            const int64_t extraBase = (i >= n - 1) ? -1 : (int64_t)code_of(b[i + 1]);    // BBDukLoader.java:443, 481
            emit_any(h, fwd, k, p.hdist, p.edist, extraBase, id, tmp);
            if (p.useShortKmers) {
                if (i == k - 1)                          // prefixes of the scaffold's first k-mer (addToMapRightShift :320-341): the
                    for (int L = k - 1; L >= p.mink; L--)                                  // base shifted out is the extra base
                        emit_any(h, fwd >> (2 * (k - L)), L, p.hdist2, p.edist2, (int64_t)((fwd >> (2 * (k - L - 1))) & 3ULL), id, tmp);
                if (i == n - 1)                          // suffixes of its last k-mer (addToMapLeftShift :289-310)
                    for (int L = k - 1; L >= p.mink; L--) emit_any(h, fwd & ((1ULL << (2 * L)) - 1), L, p.hdist2, p.edist2, extraBase, id, tmp);
            }
        }
    }
    std::sort(h->cand.begin(), h->cand.end(), [](const KV& a, const KV& b) { return a.key != b.key ? a.key < b.key : a.id < b.id; });
    h->keys.clear(); h->vals.clear();
    for (size_t i = 0; i < h->cand.size(); i++) {
        if (i == 0 || h->cand[i].key != h->cand[i - 1].key) { h->keys.push_back((int64_t)h->cand[i].key); h->vals.push_back(h->cand[i].id); }
    }
    h->cand.clear(); h->cand.shrink_to_fit();
    h->built = true;
    return (int64_t)h->keys.size();
}

extern "C" int bbduk_host_index_pairs(const bbduk_host* h, const int64_t** keys, const int32_t** values, int64_t* n) {
    if (!h || !h->built || !keys || !values || !n) return BBDUK_ERR_ARG;
    *keys = h->keys.data(); *values = h->vals.data(); *n = (int64_t)h->keys.size();
    return BBDUK_OK;
}
extern "C" int bbduk_host_num_scaffolds(const bbduk_host* h) { return h ? (int)h->scaffolds.size() + 1 : BBDUK_ERR_ARG; }
extern "C" int bbduk_host_scaffold_info(const bbduk_host* h, int32_t id, const char** name, int64_t* length) {
    if (!h || id < 1 || id > (int32_t)h->scaffolds.size()) return BBDUK_ERR_ARG;
    if (name) *name = h->names[(size_t)id - 1].c_str();
    if (length) *length = (int64_t)h->scaffolds[(size_t)id - 1].size();
    return BBDUK_OK;
}

extern "C" int bbduk_host_params(const bbduk_host* h, int32_t device, bbduk_params* out) {
    if (!h || !out) return BBDUK_ERR_ARG;
    const Parsed& p = h->p;
    memset(out, 0, sizeof *out);
    out->abi_version = BBDUK_ABI_VERSION;
    out->mode = p.ksplit ? BBDUK_MODE_KSPLIT : (p.ktrimRight && p.ktrimLeft) ? BBDUK_MODE_KTRIM_TIPS :
                p.ktrimRight ? BBDUK_MODE_KTRIM_R : (p.ktrimLeft ? BBDUK_MODE_KTRIM_L : (p.ktrimN ? BBDUK_MODE_KMASK : BBDUK_MODE_KFILTER));
    out->k = p.k; out->mink = p.mink; out->rcomp = p.rcomp; out->forbidNs = p.forbidNs;
    out->minlen = p.minlen; out->minlen2 = p.minlen2; out->middleMask = p.middleMask;
    out->qhdist = p.qhdist; out->qhdist2 = p.qhdist2; out->maxBadKmers = p.maxBadKmers0;
    out->minReadLength = p.minReadLength; out->minLenFraction = p.minLenFraction;
    out->removePairsIfEitherBad = (!p.requireBothBad) && (!p.trimFailuresTo1bp);   // BBDukParser.java:109
    out->trimFailuresTo1bp = p.trimFailuresTo1bp ? 1 : 0;
    out->trimPad = p.trimPad; out->ktrimExclusive = p.ktrimExclusive;
    out->restrictLeft = p.restrictLeft; out->restrictRight = p.restrictRight;
    out->skipR1 = p.skipR1; out->skipR2 = p.skipR2;
    out->trimPairsEvenly = p.trimPairsEvenly; out->qSkip = p.qSkip; out->speed = p.speed;
    out->minKmerFraction = p.minKmerFraction; out->minCoveredFraction = p.minCoveredFraction;
    out->kmaskFullyCovered = (p.kmaskFullyCovered && out->mode == BBDUK_MODE_KMASK) ? 1 : 0;
    out->kbig = p.kbig > p.k ? p.kbig : 0;
    out->findBestMatch = (p.findBestMatch && out->mode == BBDUK_MODE_KFILTER) ? 1 : 0;
    out->numScaffolds = (int32_t)h->scaffolds.size() + 1;
    out->device = device;
    return BBDUK_OK;
}

// The other way to fill the device map: hand the loaded scaffolds over and let the GPU build it (bbduk_build_table_device).
extern "C" int bbduk_host_build_on_device(const bbduk_host* h, bbduk_handle* dev) {
    if (!h || !dev) return BBDUK_ERR_ARG;
    const Parsed& p = h->p;
    if (p.edist > 1 || p.edist2 > 1 || p.hdist > 3 || p.hdist2 > 3) return BBDUK_ERR_ARG;      // (edist = 1 goes to bbduk_build_table_device_edits since round 4)
    for (const auto& sc : h->scaffolds) {                                                       // so does it serve reference-side skipping
        const int64_t n = (int64_t)sc.size();
        const int heur = n > 20000000 ? p.k : n > 5000000 ? 11 : n > 500000 ? 2 : 0;
        if (std::max(p.minSkip, std::min(p.maxSkip, heur)) > 1) return BBDUK_ERR_ARG;
    }
    std::vector<uint8_t> cat; std::vector<int64_t> off(1, 0);
    for (const auto& s : h->scaffolds) { cat.insert(cat.end(), s.begin(), s.end()); off.push_back((int64_t)cat.size()); }
    if (cat.empty()) cat.push_back(0);
    return bbduk_build_table_device_edits(dev, cat.data(), off.data(), (int32_t)h->scaffolds.size(), p.hdist, p.hdist2, p.edist, p.edist2);
}

extern "C" int bbduk_host_upload_index(const bbduk_host* h, bbduk_handle* dev) {
    if (!h || !h->built || !dev) return BBDUK_ERR_ARG;
    int rc = bbduk_upload_pairs(dev, h->keys.data(), h->vals.data(), (int64_t)h->keys.size());
    if (rc != BBDUK_OK) return rc;
    return bbduk_finalize_table(dev);
}
