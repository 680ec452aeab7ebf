// bbduk_cli -- the non-JVM caller of SURVEY §8(b): plays the part of BBDukProcessorS.processList around the batch
// operators of include/bbduk_gpu.h.  It owns nothing of the algorithm: flags go to bbduk_host_parse, the k-mer map
// comes from bbduk_host_build_index, every per-read decision comes back from bbduk_ktrim_batch / bbduk_kfilter_batch.
// What it does itself is what the Java caller keeps doing: read FASTQ, group mates, apply
// TrimRead.trimByAmount to bases+qualities with the returned amounts, route removed pairs, print the counters.
//
//   bbduk_cli in=r1.fq [in2=r2.fq | int=t] [out=clean.fq] [outm=removed.fq] [tsv=per_read.tsv] [resources=DIR]
//             [batch=N] [device=D | devices=0,1,..] [devicebuild=t] [deviceingest=t|f [chunk=BYTES] [pipeline=f] [timeline=t] [readthreads=N]]
//             <BBDuk flags: ktrim= k= mink= hdist= ref= literal= ...>
//
// devices=0,1,..: one handle per listed device (a device may be listed twice), the map replicated on each; every batch
// is cut into contiguous blocks of whole pairs, one block per handle, submitted from one host thread per handle; output keeps the
// input order; at the end ONE counter all-reduce (bbduk_comm_create_local + bbduk_allreduce_counters_local: RCCL) merges the
// per-device counters, as BBDukProcessorS.add merges the per-thread processors (bbduk/BBDukProcessorS.java:300-342).
//
// deviceingest=t (the default wherever it serves the run: one device, not ksplit, no renaming, no trimfailuresto1bp): the FASTQ text itself goes to
// the GPU in chunks; record splitting, 2-bit packing,
// matching and the writing of the trimmed records all happen there (bbduk_fastq_ingest_device, bbduk_*_batch_packed_device,
// bbduk_fastq_write_device); the host only moves bytes between the files and pinned buffers.
//
// tsv columns: name, length, result (ktrim: bases removed | kfilter: k-mer hits counted | ktrim=n: bases masked),
// scaffold id or -1, length after trimming, flags (1 = read discarded, 2 = pair removed).  With ktrim=n / kmask= the
// masked bases are replaced by the trim symbol (quality '!') or lower-cased (kmask=lc), BBDukProcessorS.java:2309-2320.
// Exit status 0 = OK, 1 = error (message on stderr) -- like the reference, which sets errorState and exits non-zero
// (bbduk/BBDukS.java:199-202).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>
#include "../../include/bbduk_gpu.h"
#include "../../include/bbduk_host.h"

namespace {

struct Reader {                      // FASTQ, plain or .gz (through `gzip -dc`, as ByteFile does for .gz)
    FILE* f = nullptr; bool piped = false; char* line = nullptr; size_t cap = 0;
    bool open(const std::string& path) {
        if (path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0) {
            if (path.find('\'') != std::string::npos) return false;       // the path goes through a shell: no quote can be allowed in it
            const std::string cmd = "gzip -dc '" + path + "'";
            f = popen(cmd.c_str(), "r"); piped = true;
        } else f = fopen(path.c_str(), "r");
        return f != nullptr;
    }
    bool getline_(std::string& out) {
        const ssize_t n = getline(&line, &cap, f);
        if (n < 0) return false;
        size_t m = (size_t)n;
        while (m > 0 && (line[m - 1] == '\n' || line[m - 1] == '\r')) m--;
        out.assign(line, m);
        return true;
    }
    // one record; returns 0 = ok, 1 = clean EOF, -1 = malformed
    int next(std::string& name, std::string& bases, std::string& quals) {
        std::string plus;
        if (!getline_(name)) return 1;
        if (name.empty() || name[0] != '@') return -1;
        if (!getline_(bases) || !getline_(plus) || !getline_(quals)) return -1;
        if (plus.empty() || plus[0] != '+' || quals.size() != bases.size()) return -1;
        name.erase(0, 1);
        return 0;
    }
    // false: the decompressor failed (truncated or corrupt .gz): what was read so far is not the whole file
    bool close() { bool ok = true; if (f) { if (piped) ok = pclose(f) == 0; else fclose(f); f = nullptr; } free(line); line = nullptr; return ok; }
};

struct Rec { std::string name, bases, quals; };

int fail(const char* what, const char* detail) { fprintf(stderr, "bbduk_cli: %s%s%s\n", what, detail ? ": " : "", detail ? detail : ""); return 1; }

bool parse_bool(const std::string& v) { return v.empty() || v == "t" || v == "true" || v == "1" || v == "T"; }

// stats= and rpkm= (bbduk/BBDukProcessorS.java:572-655): per-scaffold hit statistics from the counter vector.
struct StatsOut { std::string stats, rpkm, refstats; int columns = 3; bool nonZeroOnly = true; };   // BBDukParser.java:1508, 1363
int write_stats(const StatsOut& so, const bbduk_host* host, const std::vector<int64_t>& c, const std::string& in1, const std::string& in2) {
    const int ns = bbduk_host_num_scaffolds(host);
    const int64_t readsIn = c[BBDUK_READS_IN], basesIn = c[BBDUK_BASES_IN];
    const int64_t* reads = c.data() + BBDUK_NCOUNTERS; const int64_t* bases = reads + ns;
    const std::string fileLine = "#File\t" + in1 + (in2.empty() ? "" : "\t" + in2) + "\n";
    if (!so.stats.empty()) {                                                       // writeStats :572-617
        FILE* f = fopen(so.stats.c_str(), "w");
        if (!f) return fail("cannot open", so.stats.c_str());
        struct SC { std::string name; int64_t reads, bases; };
        std::vector<SC> list; int64_t rsum = 0, bsum = 0;
        for (int i = 1; i < ns; i++) {
            if (reads[i] > 0 || !so.nonZeroOnly) {
                const char* nm = ""; bbduk_host_scaffold_info(host, i, &nm, nullptr);
                rsum += reads[i]; bsum += bases[i];
                list.push_back(SC{nm, reads[i], bases[i]});
            }
        }
        std::sort(list.begin(), list.end(), [](const SC& a, const SC& b) {         // structures/StringCount.java:36-40
            if (a.bases != b.bases) return a.bases > b.bases;
            if (a.reads != b.reads) return a.reads > b.reads;
            return a.name < b.name;
        });
        const double rmult = 100.0 / (readsIn > 0 ? readsIn : 1), bmult = 100.0 / (basesIn > 0 ? basesIn : 1);
        fputs(fileLine.c_str(), f);
        if (so.columns == 3) {
            fprintf(f, "#Total\t%lld\n", (long long)readsIn);
            fprintf(f, "#Matched\t%lld\t%.5f%%\n", (long long)rsum, rmult * rsum);
            fputs("#Name\tReads\tReadsPct\n", f);
            for (const SC& x : list) fprintf(f, "%s\t%lld\t%.5f%%\n", x.name.c_str(), (long long)x.reads, x.reads * rmult);
        } else {
            fprintf(f, "#Total\t%lld\t%lld\n", (long long)readsIn, (long long)basesIn);
            fprintf(f, "#Matched\t%lld\t%.5f%%\n", (long long)rsum, rmult * rsum);  // the reference's format string ends here too (:602)
            fputs("#Name\tReads\tReadsPct\tBases\tBasesPct\n", f);
            for (const SC& x : list) fprintf(f, "%s\t%lld\t%.5f%%\t%lld\t%.5f%%\n", x.name.c_str(), (long long)x.reads, x.reads * rmult, (long long)x.bases, x.bases * bmult);
        }
        fclose(f);
    }
    if (!so.rpkm.empty()) {                                                        // writeRPKM :622-655
        FILE* f = fopen(so.rpkm.c_str(), "w");
        if (!f) return fail("cannot open", so.rpkm.c_str());
        int64_t mapped = 0;
        for (int i = 0; i < ns; i++) mapped += reads[i];
        fputs(fileLine.c_str(), f);
        fprintf(f, "#Reads\t%lld\n#Mapped\t%lld\n#RefSequences\t%d\n#Name\tLength\tBases\tCoverage\tReads\tRPKM\n", (long long)readsIn, (long long)mapped, ns - 1 > 0 ? ns - 1 : 0);
        const float mult = 1000000000.0f / (float)(mapped > 1 ? mapped : 1);        // float arithmetic in the reference (:640)
        for (int i = 1; i < ns; i++) {
            const char* nm = ""; int64_t len = 0; bbduk_host_scaffold_info(host, i, &nm, &len);
            const double invlen = 1.0 / (double)(len > 1 ? len : 1);
            const double mult2 = (double)mult * invlen;
            if (reads[i] > 0 || !so.nonZeroOnly) fprintf(f, "%s\t%lld\t%lld\t%.4f\t%lld\t%.4f\n", nm, (long long)len, (long long)bases[i], bases[i] * invlen, (long long)reads[i], reads[i] * mult2);
        }
        fclose(f);
    }
    if (!so.refstats.empty()) {                                                    // writeRefStats, bbduk/BBDukIndexMod.java:196-245
        FILE* f = fopen(so.refstats.c_str(), "w");
        if (!f) return fail("cannot open", so.refstats.c_str());
        int64_t mapped = 0;
        for (int i = 0; i < ns; i++) mapped += reads[i];
        const int nr = bbduk_host_num_refs(host);
        fputs(fileLine.c_str(), f);
        fprintf(f, "#Reads\t%lld\n#Mapped\t%lld\n#References\t%d\n#Name\tLength\tScaffolds\tBases\tCoverage\tReads\tRPKM\n", (long long)readsIn, (long long)mapped, nr > 0 ? nr : 0);
        const float mult = 1000000000.0f / (float)(mapped > 1 ? mapped : 1);
        static const char* const exts[] = {"fq", "fastq", "fa", "fasta", "fas", "fna", "ffn", "frn", "seq", "fsa", "faa", "gz", "bz2", "zip", "xz", "zst", "txt", nullptr};
        for (int r = 0, sidx = 1; r < nr; r++) {
            const char* nm = ""; int32_t scafs = 0; bbduk_host_ref_info(host, r, &nm, &scafs);
            int64_t rr = 0, rb = 0, len = 0;
            for (const int lim = sidx + scafs; sidx < lim; sidx++) { int64_t L = 0; bbduk_host_scaffold_info(host, sidx, nullptr, &L); rr += reads[sidx]; rb += bases[sidx]; len += L; }
            std::string core = nm;                                                 // ReadWrite.stripToCore: no directories, no (stacked) extensions
            const size_t sl = core.find_last_of("/\\"); if (sl != std::string::npos) core.erase(0, sl + 1);
            for (bool again = true; again;) {
                again = false;
                for (int e = 0; exts[e]; e++) { const std::string x = std::string(".") + exts[e]; if (core.size() > x.size() && core.compare(core.size() - x.size(), x.size(), x) == 0) { core.erase(core.size() - x.size()); again = true; } }
            }
            const double invlen = 1.0 / (double)(len > 1 ? len : 1), mult2 = (double)mult * invlen;
            if (rr > 0 || !so.nonZeroOnly) fprintf(f, "%s\t%lld\t%d\t%lld\t%.4f\t%lld\t%.4f\n", core.c_str(), (long long)len, scafs, (long long)rb, rb * invlen, (long long)rr, rr * mult2);
        }
        fclose(f);
    }
    return 0;
}

struct TextIn {                      // one input file streamed through a pinned buffer; `have` bytes wait at the front
    FILE* f = nullptr; bool piped = false, eof = false; uint8_t* h = nullptr; uint8_t* d = nullptr; int64_t have = 0; int64_t* d_lines = nullptr;
    bool open(const std::string& path) {
        if (path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0) {
            if (path.find('\'') != std::string::npos) return false;
            f = popen(("gzip -dc '" + path + "'").c_str(), "r"); piped = true;
        } else f = fopen(path.c_str(), "rb");
        return f != nullptr;
    }
    void fill(int64_t cap) { while (!eof && have < cap) { const size_t g = fread(h + have, 1, (size_t)(cap - have), f); if (g == 0) eof = true; have += (int64_t)g; } }
    bool close() { bool ok = true; if (f) { if (piped) ok = pclose(f) == 0; else fclose(f); f = nullptr; } return ok; }
};

// The deviceingest=t pipeline.  Returns 0 or an error exit status; prints the same closing lines as the host path.
int run_device_ingest(bbduk_handle* dev, const bbduk_host* host, const StatsOut& so, const bbduk_params& P, const std::string& in1, const std::string& in2, bool paired,
                      const std::string& out, const std::string& outm, const std::string& tsv, int64_t chunk, int device, bool maskLower, char maskSymbol) {
    if (P.mode == BBDUK_MODE_KSPLIT) return fail("deviceingest=t serves ktrim=r|l|rl|n and kfilter, not ksplit", nullptr);
    const int ns = in2.empty() ? 1 : 2;
    TextIn T[2];
    if (!T[0].open(in1)) return fail("cannot open", in1.c_str());
    if (ns == 2 && !T[1].open(in2)) return fail("cannot open", in2.c_str());
    FILE* fout = out.empty() ? nullptr : fopen(out.c_str(), "wb");
    FILE* foutm = outm.empty() ? nullptr : fopen(outm.c_str(), "wb");
    FILE* ftsv = tsv.empty() ? nullptr : fopen(tsv.c_str(), "w");
    if ((!out.empty() && !fout) || (!outm.empty() && !foutm) || (!tsv.empty() && !ftsv)) return fail("cannot open an output file", nullptr);
    const int64_t maxRec = chunk / 40 + 16;                       // records taken per text and round; more just wait for the next round
    const int64_t maxReads = maxRec * ns, capBases = (chunk * ns) / 2 + 64, capOut = chunk * ns + 64;
    void* p = nullptr;
    auto dmal = [&](int64_t bytes) -> void* { return bbduk_device_malloc(device, bytes, &p) == BBDUK_OK ? p : nullptr; };
    auto pmal = [&](int64_t bytes) -> void* { return bbduk_pinned_malloc(bytes, &p) == BBDUK_OK ? p : nullptr; };
    for (int s = 0; s < ns; s++) {
        T[s].h = (uint8_t*)pmal(chunk + 16); T[s].d = (uint8_t*)dmal(chunk + 64); T[s].d_lines = (int64_t*)dmal((4 * maxRec + 1) * 8);
        if (!T[s].h || !T[s].d || !T[s].d_lines) return fail("out of memory (chunk= too large?)", nullptr);
    }
    int64_t* d_off = (int64_t*)dmal((maxReads + 1) * 8); uint32_t* d_codes = (uint32_t*)dmal(capBases / 4 + 64); uint32_t* d_undef = (uint32_t*)dmal(capBases / 8 + 64);
    int32_t* d_a = (int32_t*)dmal(maxReads * 4); int32_t* d_id = (int32_t*)dmal(maxReads * 4); uint8_t* d_fl = (uint8_t*)dmal(maxReads);
    int32_t* d_b = P.mode == BBDUK_MODE_KTRIM_TIPS ? (int32_t*)dmal(maxReads * 4) : nullptr;                 // ktrim=rl: the left amounts
    uint32_t* d_mask = P.mode == BBDUK_MODE_KMASK ? (uint32_t*)dmal((capBases / 32 + 4) * 4) : nullptr;      // ktrim=n: one bit per base
    if ((P.mode == BBDUK_MODE_KTRIM_TIPS && !d_b) || (P.mode == BBDUK_MODE_KMASK && !d_mask)) return fail("out of memory (chunk= too large?)", nullptr);
    std::vector<int32_t> rb;
    const int nctr = bbduk_counters_len(dev);
    int64_t* d_ctr = (int64_t*)dmal((int64_t)nctr * 8);
    uint8_t* d_out = (uint8_t*)dmal(capOut); uint8_t* h_out = (uint8_t*)pmal(capOut);
    if (!d_off || !d_codes || !d_undef || !d_a || !d_id || !d_fl || !d_ctr || !d_out || !h_out) return fail("out of memory (chunk= too large?)", nullptr);
    bbduk_device_memset(device, d_ctr, 0, (int64_t)nctr * 8, nullptr);
    std::vector<int64_t> lines[2]; std::vector<int32_t> ra, rid; std::vector<uint8_t> rfl;
    const bool kfilter = P.mode == BBDUK_MODE_KFILTER;
    long long nread = 0;
    for (;;) {
        for (int s = 0; s < ns; s++) T[s].fill(chunk);
        if (T[0].have == 0 && (ns == 1 || T[1].have == 0)) break;
        const bool fin = T[0].eof && (ns == 1 || T[1].eof);
        for (int s = 0; s < ns; s++) if (bbduk_copy_to_device(device, T[s].d, T[s].h, T[s].have, nullptr) != BBDUK_OK) return fail("host to device copy", nullptr);
        bbduk_fastq_result R;
        const int rc = bbduk_fastq_ingest_device(T[0].d, T[0].have, ns == 2 ? T[1].d : nullptr, ns == 2 ? T[1].have : 0, fin ? 1 : 0, maxReads, capBases,
                                                 T[0].d_lines, ns == 2 ? T[1].d_lines : nullptr, d_off, d_codes, d_undef, device, nullptr, &R);
        if (rc == BBDUK_ERR_FORMAT) { char m[64]; snprintf(m, sizeof m, "read %lld", nread + (long long)R.first_bad_read); return fail("malformed FASTQ record at", m); }
        if (rc != BBDUK_OK) return fail("bbduk_fastq_ingest_device", nullptr);
        int64_t n = R.n_reads;
        if (paired && ns == 1 && (n & 1)) {                       // interleaved text: a pair stays together; its second record comes next round
            if (fin) return fail("unpaired or malformed mate at the end of", in1.c_str());
            n--;
            if (bbduk_copy_from_device(device, &R.consumed1, T[0].d_lines + 4 * n, 8, nullptr) != BBDUK_OK) return fail("device to host copy", nullptr);
        }
        if (n == 0) {
            if (ns == 2 && (T[0].have == 0) != (T[1].have == 0) && (T[0].have == 0 ? T[0].eof : T[1].eof)) return fail("the two input files hold different numbers of reads", nullptr);
            if (fin) return fail("truncated FASTQ record at the end of", in1.c_str());
            if (T[0].have >= chunk || (ns == 2 && T[1].have >= chunk)) return fail("a FASTQ record exceeds chunk=", nullptr);
            continue;
        }
        const int64_t total = [&]() { int64_t v = 0; bbduk_copy_from_device(device, &v, d_off + n, 8, nullptr); return v; }();
        const int orc = kfilter ? bbduk_kfilter_batch_packed_device(dev, d_codes, d_undef, d_off, n, total, paired ? 1 : 0, d_a, d_id, d_fl, d_ctr, nullptr)
                      : P.mode == BBDUK_MODE_KTRIM_TIPS ? bbduk_ktrimtips_batch_packed_device(dev, d_codes, d_undef, d_off, n, total, paired ? 1 : 0, d_a, d_b, d_id, d_fl, d_ctr, nullptr)
                      : P.mode == BBDUK_MODE_KMASK ? bbduk_kmask_batch_packed_device(dev, d_codes, d_undef, d_off, n, total, paired ? 1 : 0, d_a, d_id, d_fl, d_mask, d_ctr, nullptr)
                      : bbduk_ktrim_batch_packed_device(dev, d_codes, d_undef, d_off, n, total, paired ? 1 : 0, d_a, d_id, d_fl, d_ctr, nullptr);
        if (orc != BBDUK_OK) return fail("batch operator", bbduk_last_error(dev));
        const int32_t* dl = P.mode == BBDUK_MODE_KTRIM_L ? d_a : (P.mode == BBDUK_MODE_KTRIM_TIPS ? d_b : nullptr);
        const int32_t* dr = (P.mode == BBDUK_MODE_KTRIM_R || P.mode == BBDUK_MODE_KTRIM_TIPS) ? d_a : nullptr;
        for (int sel = 0; sel < 2; sel++) {
            FILE* dst = sel ? foutm : fout;
            if (!dst) continue;
            int64_t nb = 0;
            if (bbduk_fastq_write_masked_device(T[0].d, T[0].d_lines, ns == 2 ? T[1].d : nullptr, ns == 2 ? T[1].d_lines : nullptr, n, dl, dr, d_fl, sel,
                                                d_mask ? d_off : nullptr, d_mask, maskLower ? -1 : (int)(unsigned char)maskSymbol, d_out, capOut, device, nullptr, &nb) != BBDUK_OK)
                return fail("bbduk_fastq_write_device", nullptr);
            if (bbduk_copy_from_device(device, h_out, d_out, nb, nullptr) != BBDUK_OK) return fail("device to host copy", nullptr);
            if (nb > 0 && fwrite(h_out, 1, (size_t)nb, dst) != (size_t)nb) return fail("write error", nullptr);
        }
        if (ftsv) {                                               // names and lengths come straight out of the text the host still holds
            const int64_t rec = n / ns;
            ra.resize(n); rid.resize(n); rfl.resize(n);
            bbduk_copy_from_device(device, ra.data(), d_a, n * 4, nullptr); bbduk_copy_from_device(device, rid.data(), d_id, n * 4, nullptr);
            bbduk_copy_from_device(device, rfl.data(), d_fl, n, nullptr);
            if (d_b) { rb.resize(n); bbduk_copy_from_device(device, rb.data(), d_b, n * 4, nullptr); for (int64_t i = 0; i < n; i++) ra[i] += rb[i]; }   // the tsv shows right + left
            for (int s = 0; s < ns; s++) { lines[s].resize(4 * rec + 1); bbduk_copy_from_device(device, lines[s].data(), T[s].d_lines, (4 * rec + 1) * 8, nullptr); }
            for (int64_t i = 0; i < n; i++) {
                const int s = ns == 2 ? (int)(i & 1) : 0; const int64_t r = ns == 2 ? (i >> 1) : i;
                const uint8_t* t = T[s].h; const int64_t* l = lines[s].data();
                auto len = [&](int64_t a, int64_t b) { int64_t m = b - a - 1; if (m > 0 && t[b - 2] == '\r') m--; return (int)m; };
                const int hl = len(l[4 * r], l[4 * r + 1]), L = len(l[4 * r + 1], l[4 * r + 2]);
                fprintf(ftsv, "%.*s\t%d\t%d\t%d\t%d\t%d\n", hl - 1, (const char*)t + l[4 * r] + 1, L, ra[i], rid[i], L - ((kfilter || d_mask) ? 0 : ra[i]), (int)rfl[i]);
            }
        }
        nread += n;
        const int64_t used[2] = {R.consumed1, R.consumed2};
        for (int s = 0; s < ns; s++) { memmove(T[s].h, T[s].h + used[s], (size_t)(T[s].have - used[s])); T[s].have -= used[s]; }
    }
    for (int s = 0; s < ns; s++) if (!T[s].close()) return fail("gzip reported an error (truncated or corrupt input?) on", s ? in2.c_str() : in1.c_str());
    if (fout) fclose(fout);
    if (foutm) fclose(foutm);
    if (ftsv) fclose(ftsv);
    std::vector<int64_t> c((size_t)nctr);
    bbduk_copy_from_device(device, c.data(), d_ctr, (int64_t)nctr * 8, nullptr);
    fprintf(stderr, "Input:                  \t%lld reads \t\t%lld bases.\n", (long long)c[BBDUK_READS_IN], (long long)c[BBDUK_BASES_IN]);
    if (d_mask) fprintf(stderr, "KMasked:                \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KTRIMMED], (long long)c[BBDUK_BASES_KTRIMMED]);
    else if (!kfilter) fprintf(stderr, "KTrimmed:               \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KTRIMMED], (long long)c[BBDUK_BASES_KTRIMMED]);
    else fprintf(stderr, "Contaminants:           \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KFILTERED], (long long)c[BBDUK_BASES_KFILTERED]);
    fprintf(stderr, "Total Removed:          \t%lld reads \t%lld bases\n", (long long)(c[BBDUK_READS_IN] - c[BBDUK_READS_OUTU]), (long long)(c[BBDUK_BASES_IN] - c[BBDUK_BASES_OUTU]));
    fprintf(stderr, "Result:                 \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_OUTU], (long long)c[BBDUK_BASES_OUTU]);
    if (c[BBDUK_CTR_STATUS]) return fail("device reported an error status", nullptr);
    return write_stats(so, host, c, in1, in2);
}

// ---- deviceingest=t, pipelined (round 5; plain input files -- one, or in= + in2= -- and no tsv=).  The serial form above reads, copies, runs and writes one chunk after
// the other, so a run costs the SUM of its stages (profiles/r04_cli_31gb.json: 2.7 s for 31 GB, 8.4 s with out=).  Here each stage has its own
// thread and the run costs the slowest one:
//   reader    `rthreads` pread()s per piece of the file into one of NB pinned buffers
//   uploader  host to device on a stream of its own into one of ND text slots, each with room in front for the tail the piece before left over
//   main      tail (a device to device copy of a partial record), bbduk_fastq_ingest_device, the packed operator, bbduk_fastq_write_masked_device
//             into one of NO output slots, device to host queued on that slot's stream
//   writer    waits for that copy, writes the bytes
// Pieces are cut at fixed file offsets, so a record (or the second mate of an interleaved pair) can straddle two pieces: what a piece leaves
// unconsumed is copied in front of the next one on the device, where the text may start at any byte.  Two files: reader, uploader and tail per file;
// a batch takes as many records as BOTH hold, so the file that is ahead keeps a longer tail and skips a round of taking a new piece when it passes chunk.
struct Turnstile {                   // a counter that only grows; other threads wait for it to pass a value
    std::mutex m; std::condition_variable cv; int64_t v = 0; bool dead = false;
    void set(int64_t x) { { std::lock_guard<std::mutex> g(m); v = x; } cv.notify_all(); }
    bool wait_above(int64_t x) { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return v > x || dead; }); return v > x; }   // false: the run was abandoned
    void kill() { { std::lock_guard<std::mutex> g(m); dead = true; } cv.notify_all(); }
};
struct StageClock {                  // seconds a stage spent working, for timeline=t
    double busy = 0; std::chrono::steady_clock::time_point t0;
    void start() { t0 = std::chrono::steady_clock::now(); }
    void stop() { busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool pread_all(int fd, uint8_t* dst, int64_t off, int64_t bytes) {
    while (bytes > 0) { const ssize_t g = pread(fd, dst, (size_t)std::min<int64_t>(bytes, 1LL << 30), (off_t)off); if (g <= 0) return false; dst += g; off += g; bytes -= g; }
    return true;
}
bool pwrite_all(int fd, const uint8_t* src, int64_t off, int64_t bytes) {
    while (bytes > 0) { const ssize_t g = pwrite(fd, src, (size_t)std::min<int64_t>(bytes, 1LL << 30), (off_t)off); if (g <= 0) return false; src += g; off += g; bytes -= g; }
    return true;
}
// `bytes` at `off`, by `nt` threads over 1 MiB-aligned ranges
template <class F> bool in_parallel(int64_t bytes, int nt, F&& part) {
    const int64_t step = std::max<int64_t>(((bytes + nt - 1) / nt + 0xFFFFF) & ~0xFFFFFLL, 1 << 20);
    std::vector<std::thread> th; std::vector<char> ok((size_t)nt, 1);
    int used = 0;
    for (int64_t a = step; a < bytes && used + 1 < nt; a += step) { const int me = ++used; th.emplace_back([&, a, me]() { ok[(size_t)me] = part(a, std::min(step, bytes - a)) ? 1 : 0; }); }
    ok[0] = part(0, std::min(step, bytes)) ? 1 : 0;
    for (auto& t : th) t.join();
    for (char c : ok) if (!c) return false;
    return true;
}

// The buffers of a pipelined run.  They depend on chunk=, the device and the number of input files only, so main() allocates them on a thread of their
// own while the reference is parsed and the k-mer map is built (pinning ~0.6 GB of host memory is ~70 ms of a run that lasts ~1 s).
struct PipeBufs {
    static constexpr int NB = 4, ND = 3, NO = 3;
    int64_t chunk = 0, haveMax = 0, maxRec = 0, maxReads = 0, capBases = 0, capOut = 0; int device = 0, ns = 1; bool ok = false; double seconds = 0;
    uint8_t* H[2][NB] = {}; uint8_t* D[2][ND] = {}; void* upStream[2] = {}; int64_t* d_lines[2] = {};
    uint8_t* dOut[NO] = {}; uint8_t* hOut[NO] = {}; void* outStream[NO] = {};
    int64_t* d_off = nullptr; uint32_t* d_codes = nullptr; uint32_t* d_undef = nullptr; int32_t* d_a = nullptr; int32_t* d_id = nullptr; uint8_t* d_fl = nullptr;
    std::thread worker;
    // Two phases: what the pipeline needs to START (the device side, the streams, one pinned buffer per file) -> `ready`; then the other pinned buffers
    // while the first pieces are already on their way (hostAlloc[s] = buffers of file s that exist, outAlloc = output buffers that exist): pinning host
    // memory runs at 2-3 GB/s, 0.3 s for two files with out= where the map build that hides it takes 0.14 s.
    Turnstile ready, hostAlloc[2], outAlloc;
    void start(int dev, int64_t chunkBytes, int nstreams, bool anyOut) {
        device = dev; chunk = chunkBytes; ns = nstreams;
        haveMax = 2 * chunk;                                      // per text: a tail (<= chunk) + a piece
        maxRec = haveMax / 40 + 16; maxReads = maxRec * ns; capBases = (haveMax * ns) / 2 + 64; capOut = haveMax * ns + 64;
        worker = std::thread([this, anyOut]() {
            const double t0 = now_s();
            void* p = nullptr;
            auto dmal = [&](int64_t bytes) -> void* { return bbduk_device_malloc(device, bytes, &p) == BBDUK_OK ? p : nullptr; };
            auto pmal = [&](int64_t bytes) -> void* { return bbduk_pinned_malloc(bytes, &p) == BBDUK_OK ? p : nullptr; };
            auto give_up = [&]() { ready.kill(); hostAlloc[0].kill(); hostAlloc[1].kill(); outAlloc.kill(); seconds = now_s() - t0; };
            bool good = true;
            for (int s = 0; s < ns; s++) {
                for (int i = 0; i < ND; i++) good = good && (D[s][i] = (uint8_t*)dmal(haveMax + 64));
                good = good && bbduk_stream_create(device, &upStream[s]) == BBDUK_OK && (d_lines[s] = (int64_t*)dmal((4 * maxRec + 1) * 8));
            }
            if (anyOut) for (int i = 0; i < NO; i++) good = good && (dOut[i] = (uint8_t*)dmal(capOut)) && bbduk_stream_create(device, &outStream[i]) == BBDUK_OK;
            good = good && (d_off = (int64_t*)dmal((maxReads + 1) * 8)) && (d_codes = (uint32_t*)dmal(capBases / 4 + 64)) &&
                   (d_undef = (uint32_t*)dmal(capBases / 8 + 64)) && (d_a = (int32_t*)dmal(maxReads * 4)) && (d_id = (int32_t*)dmal(maxReads * 4)) && (d_fl = (uint8_t*)dmal(maxReads));
            for (int s = 0; s < ns; s++) good = good && (H[s][0] = (uint8_t*)pmal(chunk + 16));
            if (!good) { give_up(); return; }
            for (int s = 0; s < ns; s++) hostAlloc[s].set(1);
            ready.set(1);
            for (int i = 1; i < NB; i++) for (int s = 0; s < ns; s++) {
                if (!(H[s][i] = (uint8_t*)pmal(chunk + 16))) { give_up(); return; }
                hostAlloc[s].set(i + 1);
            }
            if (anyOut) for (int i = 0; i < NO; i++) {
                if (!(hOut[i] = (uint8_t*)pmal(capOut))) { give_up(); return; }
                outAlloc.set(i + 1);
            }
            ok = true; seconds = now_s() - t0;
        });
    }
    bool wait() { if (worker.joinable()) worker.join(); return ok; }
    ~PipeBufs() { wait(); }
};

// One input file of a pipelined run: its pieces (fixed file offsets), the reader and the uploader that bring them to the device, and what the main
// stage holds of it -- `have` bytes at `text` (the unconsumed tail of the pieces taken so far, then nothing else).
struct InStream {
    int fd = -1; int64_t bytes = 0, NP = 0, next = 0;             // next: the first piece main has not taken yet
    Turnstile filled, hostFreed, uploaded, devFreed;
    std::thread reader, uploader;
    uint8_t* text = nullptr; int64_t have = 0;
    StageClock cRead, cUp;
};

int run_device_ingest_piped(bbduk_handle* dev, const bbduk_host* host, const StatsOut& so, const bbduk_params& P, const std::string& in1, const std::string& in2, bool paired,
                            const std::string& out, const std::string& outm, PipeBufs& B, bool maskLower, char maskSymbol,
                            int rthreads, int wthreads, bool timeline, double tProcess) {
    if (P.mode == BBDUK_MODE_KSPLIT) return fail("deviceingest=t serves ktrim=r|l|rl|n and kfilter, not ksplit", nullptr);
    const double tBegin = now_s();
    const int ns = B.ns;
    InStream S[2];
    const int64_t chunk = B.chunk; const int device = B.device;
    int fdOut[2] = {-1, -1};
    void* owned[3] = {nullptr, nullptr, nullptr};                 // this call's own device buffers (d_b, d_mask, d_ctr)
    // every way out of this function -- the early returns below too -- closes what it opened and frees what it allocated (an in-process caller
    // inherited the descriptors and buffers; the threads are joined by finish() before any return that follows their start)
    struct Owned {
        InStream* S; int* fdOut; void** owned; int device;
        ~Owned() {
            for (int s = 0; s < 2; s++) if (S[s].fd >= 0) { close(S[s].fd); S[s].fd = -1; }
            for (int f = 0; f < 2; f++) if (fdOut[f] >= 0) { close(fdOut[f]); fdOut[f] = -1; }
            for (int q = 0; q < 3; q++) if (owned[q]) { bbduk_device_free(device, owned[q]); owned[q] = nullptr; }
        }
    } ownedGuard{S, fdOut, owned, device};
    for (int s = 0; s < ns; s++) {
        const std::string& path = s ? in2 : in1;
        S[s].fd = open(path.c_str(), O_RDONLY);
        if (S[s].fd < 0) return fail("cannot open", path.c_str());
        struct stat sb;
        if (fstat(S[s].fd, &sb) != 0) return fail("cannot stat", path.c_str());
        if (!S_ISREG(sb.st_mode)) return fail("the pipelined ingest reads regular files (pipeline=f serves pipes and devices)", path.c_str());
        S[s].bytes = (int64_t)sb.st_size; S[s].NP = (S[s].bytes + B.chunk - 1) / B.chunk;
    }
    if (!out.empty()) fdOut[0] = open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (!outm.empty()) fdOut[1] = open(outm.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if ((!out.empty() && fdOut[0] < 0) || (!outm.empty() && fdOut[1] < 0)) return fail("cannot open an output file", nullptr);
    const double tOpen = now_s();
    if (!B.ready.wait_above(0)) return fail("out of memory (chunk= too large?)", nullptr);
    constexpr int NB = PipeBufs::NB, ND = PipeBufs::ND, NO = PipeBufs::NO;
    const int64_t maxReads = B.maxReads, capBases = B.capBases, capOut = B.capOut;
    uint8_t** dOut = B.dOut; uint8_t** hOut = B.hOut; void** outStream = B.outStream;
    int64_t* d_off = B.d_off; uint32_t* d_codes = B.d_codes; uint32_t* d_undef = B.d_undef; int32_t* d_a = B.d_a; int32_t* d_id = B.d_id; uint8_t* d_fl = B.d_fl;
    void* p = nullptr;
    auto dmal = [&](int64_t bytes) -> void* { return bbduk_device_malloc(device, bytes, &p) == BBDUK_OK ? p : nullptr; };
    int32_t* d_b = P.mode == BBDUK_MODE_KTRIM_TIPS ? (int32_t*)dmal(maxReads * 4) : nullptr;
    owned[0] = d_b;
    uint32_t* d_mask = P.mode == BBDUK_MODE_KMASK ? (uint32_t*)dmal((capBases / 32 + 4) * 4) : nullptr;
    owned[1] = d_mask;
    const int nctr = bbduk_counters_len(dev);
    int64_t* d_ctr = (int64_t*)dmal((int64_t)nctr * 8);
    owned[2] = d_ctr;
    if (!d_ctr || (P.mode == BBDUK_MODE_KTRIM_TIPS && !d_b) || (P.mode == BBDUK_MODE_KMASK && !d_mask)) return fail("out of memory (chunk= too large?)", nullptr);
    bbduk_device_memset(device, d_ctr, 0, (int64_t)nctr * 8, nullptr);
    const double tAlloc = now_s();

    Turnstile outQueued, outFreed;
    StageClock cTail, cIngest, cOp, cWriteK, cD2H, cWrite, cWaitUp, cWaitOut;
    std::atomic<const char*> failed{nullptr};                     // set by a stage thread before it kills the turnstiles
    auto abandon = [&](const char* why) {
        const char* none = nullptr; failed.compare_exchange_strong(none, why);
        for (int s = 0; s < ns; s++) for (Turnstile* t : {&S[s].filled, &S[s].hostFreed, &S[s].uploaded, &S[s].devFreed}) t->kill();
        outQueued.kill(); outFreed.kill();
    };
    const int rt = std::max(1, rthreads / ns);
    for (int s = 0; s < ns; s++) {
        InStream& X = S[s];
        auto piece_bytes = [&X, chunk](int64_t q) { return std::min(chunk, X.bytes - q * chunk); };
        X.reader = std::thread([&, s, piece_bytes]() {
            InStream& Y = S[s];
            for (int64_t q = 0; q < Y.NP; q++) {
                if (q >= NB && !Y.hostFreed.wait_above(q - NB)) return;
                if (q < NB && !B.hostAlloc[s].wait_above(q)) { abandon("out of memory (chunk= too large?)"); return; }      // (the buffer is still being pinned)
                Y.cRead.start();
                const int64_t off = q * chunk;
                const bool ok = in_parallel(piece_bytes(q), rt, [&](int64_t a, int64_t n) { return pread_all(Y.fd, B.H[s][q % NB] + a, off + a, n); });
                Y.cRead.stop();
                if (!ok) { abandon("read error"); return; }
                Y.filled.set(q + 1);
            }
        });
        X.uploader = std::thread([&, s, piece_bytes]() {
            InStream& Y = S[s];
            for (int64_t q = 0; q < Y.NP; q++) {
                if (!Y.filled.wait_above(q)) return;
                if (q >= ND && !Y.devFreed.wait_above(q - ND)) return;
                Y.cUp.start();
                const bool ok = bbduk_copy_async(device, B.D[s][q % ND] + chunk, B.H[s][q % NB], piece_bytes(q), 0, B.upStream[s]) == BBDUK_OK && bbduk_stream_synchronize(device, B.upStream[s]) == BBDUK_OK;
                Y.cUp.stop();
                if (!ok) { abandon("host to device copy"); return; }
                Y.hostFreed.set(q + 1); Y.uploaded.set(q + 1);
            }
        });
    }
    struct OutJob { int slot; int64_t bytes; int sel; };
    std::mutex jobMu; std::deque<OutJob> jobs;
    int64_t outPos[2] = {0, 0};
    std::thread writer([&]() {
        for (int64_t j = 0;; j++) {
            if (!outQueued.wait_above(j)) return;
            OutJob job; { std::lock_guard<std::mutex> g(jobMu); job = jobs.front(); jobs.pop_front(); }
            if (job.slot < 0) return;                             // the end marker
            cD2H.start();
            const bool okc = bbduk_stream_synchronize(device, outStream[job.slot]) == BBDUK_OK;
            cD2H.stop();
            cWrite.start();
            const int f = fdOut[job.sel]; const int64_t at = outPos[job.sel];
            const bool okw = okc && in_parallel(job.bytes, wthreads, [&](int64_t a, int64_t n) { return pwrite_all(f, hOut[job.slot] + a, at + a, n); });
            cWrite.stop();
            outPos[job.sel] += job.bytes;
            if (!okw) { abandon(okc ? "write error" : "device to host copy"); return; }
            outFreed.set(j + 1);
        }
    });
    auto push_job = [&](const OutJob& j, int64_t seq) { { std::lock_guard<std::mutex> g(jobMu); jobs.push_back(j); } outQueued.set(seq + 1); };
    int64_t outSeq = 0;
    auto finish = [&](int rc) {                                   // every exit goes through here: the threads hold references to this frame
        if (rc != 0) abandon("stopped");
        else push_job(OutJob{-1, 0, 0}, outSeq);
        for (int s = 0; s < ns; s++) { S[s].reader.join(); S[s].uploader.join(); }
        writer.join();
        B.wait();                                                 // (a tiny input can be through before the last buffer is pinned)
        return rc;                                                 // (descriptors and buffers: ownedGuard)
    };

    const bool kfilter = P.mode == BBDUK_MODE_KFILTER;
    long long nread = 0; int rounds = 0;
    for (;;) {
        // a text takes its next piece when what it still holds fits the room in front of one (<= chunk): with two files the one that is ahead in
        // records waits a round while the other catches up, as the serial form's fill-to-chunk does
        bool took = false;
        for (int s = 0; s < ns; s++) {
            InStream& X = S[s];
            if (X.next >= X.NP || X.have > chunk) continue;
            const int64_t q = X.next;
            cWaitUp.start();
            const bool okq = X.uploaded.wait_above(q);
            cWaitUp.stop();
            if (!okq) return finish(fail(failed.load() ? failed.load() : "pipeline stopped", nullptr));
            uint8_t* fresh = B.D[s][q % ND] + chunk - X.have;
            cTail.start();
            if (X.have > 0 && (bbduk_copy_async(device, fresh, X.text, X.have, 2, nullptr) != BBDUK_OK || bbduk_stream_synchronize(device, nullptr) != BBDUK_OK)) return finish(fail("device to device copy", nullptr));
            cTail.stop();
            X.devFreed.set(q);                                    // the slots of the pieces before this one can be overwritten
            X.text = fresh; X.have += std::min(chunk, X.bytes - q * chunk); X.next++; took = true;
        }
        const bool fin = S[0].next >= S[0].NP && (ns == 1 || S[1].next >= S[1].NP);
        if (S[0].have == 0 && (ns == 1 || S[1].have == 0)) { if (fin) break; continue; }
        rounds++;
        bbduk_fastq_result R;
        cIngest.start();
        const int rc = bbduk_fastq_ingest_device(S[0].text, S[0].have, ns == 2 ? S[1].text : nullptr, ns == 2 ? S[1].have : 0, fin ? 1 : 0, maxReads, capBases,
                                                 B.d_lines[0], ns == 2 ? B.d_lines[1] : nullptr, d_off, d_codes, d_undef, device, nullptr, &R);
        cIngest.stop();
        if (rc == BBDUK_ERR_FORMAT) { char m[64]; snprintf(m, sizeof m, "read %lld", nread + (long long)R.first_bad_read); return finish(fail("malformed FASTQ record at", m)); }
        if (rc != BBDUK_OK) return finish(fail("bbduk_fastq_ingest_device", nullptr));
        int64_t n = R.n_reads;
        const bool full = n >= maxReads - (ns == 2 ? 1 : 0);      // the record limit, not the end of the text, ended this round
        if (paired && ns == 1 && (n & 1)) {                       // interleaved text: a pair stays together; its second record comes with the next piece
            if (fin && !full) return finish(fail("unpaired or malformed mate at the end of", in1.c_str()));
            n--;
            if (bbduk_copy_from_device(device, &R.consumed1, B.d_lines[0] + 4 * n, 8, nullptr) != BBDUK_OK) return finish(fail("device to host copy", nullptr));
        }
        if (n == 0) {
            if (ns == 2 && (S[0].have == 0) != (S[1].have == 0) && (S[0].have == 0 ? S[0].next >= S[0].NP : S[1].next >= S[1].NP))
                return finish(fail("the two input files hold different numbers of reads", nullptr));
            if (fin) return finish(fail("truncated FASTQ record at the end of", in1.c_str()));
            if (!took) return finish(fail("a FASTQ record exceeds chunk=", nullptr));
            continue;                                             // the next pieces bring the rest
        }
        const int64_t total = [&]() { int64_t v = 0; bbduk_copy_from_device(device, &v, d_off + n, 8, nullptr); return v; }();
        cOp.start();
        const int orc = kfilter ? bbduk_kfilter_batch_packed_device(dev, d_codes, d_undef, d_off, n, total, paired ? 1 : 0, d_a, d_id, d_fl, d_ctr, nullptr)
                      : P.mode == BBDUK_MODE_KTRIM_TIPS ? bbduk_ktrimtips_batch_packed_device(dev, d_codes, d_undef, d_off, n, total, paired ? 1 : 0, d_a, d_b, d_id, d_fl, d_ctr, nullptr)
                      : P.mode == BBDUK_MODE_KMASK ? bbduk_kmask_batch_packed_device(dev, d_codes, d_undef, d_off, n, total, paired ? 1 : 0, d_a, d_id, d_fl, d_mask, d_ctr, nullptr)
                      : bbduk_ktrim_batch_packed_device(dev, d_codes, d_undef, d_off, n, total, paired ? 1 : 0, d_a, d_id, d_fl, d_ctr, nullptr);
        if (orc == BBDUK_OK && timeline) bbduk_stream_synchronize(device, nullptr);            // so that the operator's time is its own in the table
        cOp.stop();
        if (orc != BBDUK_OK) return finish(fail("batch operator", bbduk_last_error(dev)));
        const int32_t* dl = P.mode == BBDUK_MODE_KTRIM_L ? d_a : (P.mode == BBDUK_MODE_KTRIM_TIPS ? d_b : nullptr);
        const int32_t* dr = (P.mode == BBDUK_MODE_KTRIM_R || P.mode == BBDUK_MODE_KTRIM_TIPS) ? d_a : nullptr;
        for (int sel = 0; sel < 2; sel++) {
            if (fdOut[sel] < 0) continue;
            const int slot = (int)(outSeq % NO);
            cWaitOut.start();
            const bool oko = outSeq < NO ? B.outAlloc.wait_above(outSeq) : outFreed.wait_above(outSeq - NO);
            cWaitOut.stop();
            if (!oko) return finish(fail(failed.load() ? failed.load() : "pipeline stopped", nullptr));
            int64_t nb = 0;
            cWriteK.start();
            const int wrc = bbduk_fastq_write_masked_device(S[0].text, B.d_lines[0], ns == 2 ? S[1].text : nullptr, ns == 2 ? B.d_lines[1] : nullptr, n, dl, dr, d_fl, sel,
                                                            d_mask ? d_off : nullptr, d_mask, maskLower ? -1 : (int)(unsigned char)maskSymbol, dOut[slot], capOut, device, nullptr, &nb);
            cWriteK.stop();
            if (wrc != BBDUK_OK) return finish(fail("bbduk_fastq_write_device", nullptr));
            if (bbduk_copy_async(device, hOut[slot], dOut[slot], nb, 1, outStream[slot]) != BBDUK_OK) return finish(fail("device to host copy", nullptr));
            push_job(OutJob{slot, nb, sel}, outSeq); outSeq++;
        }
        nread += n;
        S[0].text += R.consumed1; S[0].have -= R.consumed1;
        if (ns == 2) { S[1].text += R.consumed2; S[1].have -= R.consumed2; }
    }
    const double tLoop = now_s();
    if (finish(0) != 0) return 1;
    if (failed.load()) return fail(failed.load(), nullptr);
    const double tEnd = now_s();
    std::vector<int64_t> c((size_t)nctr);
    bbduk_copy_from_device(device, c.data(), d_ctr, (int64_t)nctr * 8, nullptr);
    fprintf(stderr, "Input:                  \t%lld reads \t\t%lld bases.\n", (long long)c[BBDUK_READS_IN], (long long)c[BBDUK_BASES_IN]);
    if (d_mask) fprintf(stderr, "KMasked:                \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KTRIMMED], (long long)c[BBDUK_BASES_KTRIMMED]);
    else if (!kfilter) fprintf(stderr, "KTrimmed:               \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KTRIMMED], (long long)c[BBDUK_BASES_KTRIMMED]);
    else fprintf(stderr, "Contaminants:           \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KFILTERED], (long long)c[BBDUK_BASES_KFILTERED]);
    fprintf(stderr, "Total Removed:          \t%lld reads \t%lld bases\n", (long long)(c[BBDUK_READS_IN] - c[BBDUK_READS_OUTU]), (long long)(c[BBDUK_BASES_IN] - c[BBDUK_BASES_OUTU]));
    fprintf(stderr, "Result:                 \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_OUTU], (long long)c[BBDUK_BASES_OUTU]);
    if (timeline) {                                               // one JSON line: where the wall clock went (seconds; stages on different threads overlap)
        fprintf(stderr, "{\"timeline\": {\"file_bytes\": %lld, \"pieces\": %lld, \"rounds\": %d, \"chunk\": %lld, \"rthreads\": %d, \"wthreads\": %d, \"files\": %d, "
                        "\"startup_to_pipeline\": %.3f, \"open_files\": %.3f, \"wait_buffers\": %.3f, \"buffers_alloc_thread\": %.3f, \"pipeline_wall\": %.3f, \"drain\": %.3f, "
                        "\"reader_pread\": %.3f, \"uploader_h2d\": %.3f, \"main_wait_upload\": %.3f, \"main_tail_copy\": %.3f, \"main_ingest\": %.3f, \"main_operator\": %.3f, "
                        "\"main_write_kernels\": %.3f, \"main_wait_out_slot\": %.3f, \"writer_wait_d2h\": %.3f, \"writer_pwrite\": %.3f, \"out_bytes\": %lld}}\n",
                (long long)(S[0].bytes + S[1].bytes), (long long)(S[0].NP + S[1].NP), rounds, (long long)chunk, rthreads, wthreads, ns, tBegin - tProcess, tOpen - tBegin, tAlloc - tOpen, B.seconds, tLoop - tAlloc, tEnd - tLoop,
                std::max(S[0].cRead.busy, S[1].cRead.busy), std::max(S[0].cUp.busy, S[1].cUp.busy), cWaitUp.busy, cTail.busy, cIngest.busy, cOp.busy, cWriteK.busy, cWaitOut.busy, cD2H.busy, cWrite.busy, (long long)(outPos[0] + outPos[1]));
    }
    if (c[BBDUK_CTR_STATUS]) return fail("device reported an error status", nullptr);
    return write_stats(so, host, c, in1, in2);
}

}  // namespace

// watchdog=SECONDS (diagnostics; round 5: the intermittent stall of devices=0,0,0 runs had never been looked at from the inside, the images carry no
// debugger): a thread that, if the run has not finished in time, makes EVERY thread of the process print its own call stack (a signal per thread,
// backtrace_symbols_fd in the handler) together with what the kernel says it waits in, and ends the process with status 97.
static volatile int g_wd_done = 0;
static void wd_handler(int) {
    void* fr[64];
    const int n = backtrace(fr, 64);
    char head[96]; const int m = snprintf(head, sizeof head, "---- watchdog: thread %ld\n", (long)syscall(SYS_gettid));
    if (write(2, head, (size_t)m) < 0) {}
    backtrace_symbols_fd(fr, n, 2);
}
static void wd_start(const int seconds) {
    std::thread([seconds]() {
        for (int t = 0; t < seconds * 10 && !g_wd_done; t++) usleep(100000);
        if (g_wd_done) return;
        fprintf(stderr, "==== watchdog: no end after %d s, dumping every thread ====\n", seconds);
        struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = wd_handler; sigaction(SIGUSR2, &sa, nullptr);
        const long pid = (long)getpid(), me = (long)syscall(SYS_gettid);
        if (DIR* d = opendir("/proc/self/task")) {
            while (struct dirent* e = readdir(d)) {
                const long tid = atol(e->d_name);
                if (tid <= 0 || tid == me) continue;
                char path[64], buf[256];
                for (const char* what : {"comm", "wchan", "syscall"}) {
                    snprintf(path, sizeof path, "/proc/self/task/%ld/%s", tid, what);
                    if (FILE* f = fopen(path, "r")) { const size_t k = fread(buf, 1, sizeof buf - 1, f); buf[k] = 0; fclose(f); for (char* c = buf; *c; c++) if (*c == '\n') *c = ' '; fprintf(stderr, "tid %ld %s: %s\n", tid, what, buf); }
                }
                syscall(SYS_tgkill, pid, tid, SIGUSR2);
                usleep(200000);
            }
            closedir(d);
        }
        fflush(stderr);
        _exit(97);
    }).detach();
}

int main(int argc, char** argv) {
    std::string in1, in2, out, outm, tsv, resources = "data", flags;
    bool interleaved = false; long batch = 1000000; int device = 0; std::vector<int> devices;
    bool maskLower = false; char maskSymbol = 'N';
    bool rename = false;                                          // rename=t: matched reads get "\tscaffold=hits" appended (BBDukProcessorS.java:2508-2522)
    StatsOut so;
    int deviceIngestArg = -1; long long chunk = 0;                // deviceingest=t|f; not given: on wherever that path serves the run (see below)
    bool deviceIngest = false;               // deviceingest=t: FASTQ text to the GPU, chunk= bytes per file and round (default 256 MiB; pipelined: 64 MiB over the input files)
    bool pipeline = true, timeline = false; int rthreads = 8, wthreads = 1;      // pipeline=f: the serial form; timeline=t: one JSON line of stage times on stderr
    const double tProcess = now_s();
    bool deviceBuild = false;        // devicebuild=t: the GPU builds the k-mer map from the reference sequences (bbduk_build_table_device)
    for (int i = 1; i < argc; i++) {
        const std::string tok = argv[i];
        const size_t eq = tok.find('=');
        const std::string a = tok.substr(0, eq), b = eq == std::string::npos ? "" : tok.substr(eq + 1);
        if (a == "in" || a == "in1") in1 = b;
        else if (a == "in2") in2 = b;
        else if (a == "int" || a == "interleaved") interleaved = parse_bool(b);
        else if (a == "out" || a == "out1" || a == "outu") out = b;
        else if (a == "outm" || a == "outmatch") outm = b;
        else if (a == "tsv") tsv = b;
        else if (a == "resources") resources = b;
        else if (a == "batch") batch = atol(b.c_str());
        else if (a == "device") device = atoi(b.c_str());
        else if (a == "devices") { devices.clear(); size_t q = 0; while (q <= b.size()) { const size_t c = b.find(',', q); const std::string t = b.substr(q, c == std::string::npos ? std::string::npos : c - q); if (!t.empty()) devices.push_back(atoi(t.c_str())); if (c == std::string::npos) break; q = c + 1; } }
        else if (a == "devicebuild") deviceBuild = parse_bool(b);
        else if (a == "deviceingest") deviceIngestArg = parse_bool(b) ? 1 : 0;
        else if (a == "chunk") chunk = atoll(b.c_str());
        else if (a == "pipeline") pipeline = parse_bool(b);
        else if (a == "timeline") timeline = parse_bool(b);
        else if (a == "readthreads") rthreads = std::max(1, atoi(b.c_str()));
        else if (a == "writethreads") wthreads = std::max(1, atoi(b.c_str()));
        else if (a == "watchdog") { if (atoi(b.c_str()) > 0) wd_start(atoi(b.c_str())); }
        else if (a == "stats" || a == "scafstats") so.stats = b;     // BBDukParser.java:486-494, 689-693
        else if (a == "rpkm" || a == "fpkm" || a == "cov" || a == "coverage") so.rpkm = b;
        else if (a == "refstats") so.refstats = b;                   // BBDukParser.java:493
        else if (a == "statscolumns" || a == "columns" || a == "cols") so.columns = atoi(b.c_str());
        else if (a == "nzo" || a == "nonzeroonly") so.nonZeroOnly = parse_bool(b);
        else {
            if (a == "rename") rename = parse_bool(b);               // also goes to the parser: it implies findbestmatch (BBDukParser.java:153)
            if (a == "ktrim" || a == "kmask" || a == "mask") {       // the replacement symbol is the caller's business (BBDukParser.java:619-644)
                std::string v = b; for (auto& c : v) c = (char)tolower(c);
                if (v == "lc" || v == "lowercase") maskLower = true;
                else if (b.size() == 1 && v != "t" && v != "f" && !((a == "ktrim") && (v == "r" || v == "l" || v == "n"))) maskSymbol = b[0];
            }
            flags += tok; flags += ' ';
        }
    }
    if (in1.empty()) {
        fprintf(stderr, "usage: bbduk_cli in=r1.fq [in2=r2.fq|int=t] [out=clean.fq] [outm=removed.fq] [tsv=reads.tsv] "
                        "[resources=DIR] [batch=N] [device=D] <BBDuk flags>\n");
        return 1;
    }
    if (batch < 2) batch = 2;
    const bool paired = interleaved || !in2.empty();
    if (devices.empty()) devices.push_back(device);
    device = devices[0];
    char err[512] = {0};
    bbduk_host* host = nullptr;
    if (bbduk_host_parse(flags.c_str(), &host, err, sizeof err) != BBDUK_OK) return fail("bad arguments", err);
    {   // deviceingest= not given: the device pipeline wherever it serves the run (round 5: the host parser moves 0.5 Gbases/s, the pipeline 15) --
        // one device, FASTQ text, no read renaming, no cutting of failed reads to one base, not ksplit
        bbduk_params P0;
        const bool can = bbduk_host_params(host, device, &P0) == BBDUK_OK && devices.size() == 1 && P0.mode != BBDUK_MODE_KSPLIT && !(rename && P0.findBestMatch) && !P0.trimFailuresTo1bp;
        deviceIngest = deviceIngestArg < 0 ? can : deviceIngestArg != 0;
    }
    auto gz = [](const std::string& f) { return f.size() > 3 && f.compare(f.size() - 3, 3, ".gz") == 0; };
    // (the pipelined form sizes its pieces from st_size and reads them with pread: regular files only.  /dev/stdin, a named pipe or <(zcat ..) report
    // st_size = 0 -- the run would print "Input: 0 reads" and exit 0 -- so those take the serial form, which reads its FILE* to the end: ADVICE r5)
    auto regular = [](const std::string& f) { struct stat sb; return f.empty() || (stat(f.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)); };
    const bool piped = deviceIngest && pipeline && tsv.empty() && !gz(in1) && !gz(in2) && regular(in1) && regular(in2);     // plain files, one or two
    if (chunk == 0) chunk = piped ? ((64LL << 20) / (in2.empty() ? 1 : 2)) : (256LL << 20);      // (64 MiB of text per round, over one file or two: the pinned buffers are what start-up pays for)
    if (chunk < 4096) chunk = 4096;
    PipeBufs pipeBufs;
    if (piped) pipeBufs.start(device, (int64_t)chunk, in2.empty() ? 1 : 2, !out.empty() || !outm.empty());      // (allocated while the map is built)

    if (bbduk_host_load_refs(host, resources.c_str()) < 0) return fail("cannot load ref=", resources.c_str());
    int64_t stored = 0;
    if (!deviceBuild) { stored = bbduk_host_build_index(host); if (stored < 0) return fail("index build failed", nullptr); }
    { bool several = false; for (int d : devices) several = several || d != devices[0]; if (several) bbduk_comm_preload(); }     // (the collective library loads while the map is built)
    bbduk_params P;
    if (bbduk_host_params(host, device, &P) != BBDUK_OK) return fail("unsupported parameter combination", nullptr);
    std::vector<bbduk_handle*> devs;                               // one handle per entry of devices=, the map replicated on each
    for (size_t q = 0; q < devices.size(); q++) {
        bbduk_params Pq = P; Pq.device = devices[q];
        bbduk_handle* hq = nullptr;
        if (bbduk_create(&Pq, &hq) != BBDUK_OK) return fail("bbduk_create", hq ? bbduk_last_error(hq) : "no usable device (there is no CPU fallback)");
        if (deviceBuild) {
            if (bbduk_host_build_on_device(host, hq) != BBDUK_OK) return fail("device-side table build (hdist <= 2, no edist)", bbduk_last_error(hq));
            stored = bbduk_table_size(hq);
        } else if (bbduk_host_upload_index(host, hq) != BBDUK_OK) return fail("table upload", bbduk_last_error(hq));
        devs.push_back(hq);
    }
    bbduk_handle* dev = devs[0];
    const int ndev = (int)devs.size();
    fprintf(stderr, "Added %lld kmers; %d scaffolds.\n", (long long)stored, bbduk_host_num_scaffolds(host) - 1);

    if (deviceIngest) {
        if (rename && P.findBestMatch) return fail("deviceingest=t does not rewrite read names: use rename=t without it", nullptr);
        if (ndev > 1) return fail("deviceingest=t drives one device: use device=", nullptr);
        if (P.trimFailuresTo1bp) return fail("deviceingest=t does not cut discarded reads to one base: use trimfailuresto1bp without it", nullptr);
        const int rc = piped ? run_device_ingest_piped(dev, host, so, P, in1, in2, paired, out, outm, pipeBufs, maskLower, maskSymbol, rthreads, wthreads, timeline, tProcess)
                             : run_device_ingest(dev, host, so, P, in1, in2, paired, out, outm, tsv, (int64_t)chunk, device, maskLower, maskSymbol);
        if (piped) { pipeBufs.wait(); fflush(stdout); fflush(stderr); g_wd_done = 1; _exit(rc); }   // every file is closed; the buffers go with the process (tearing the runtime down in order costs ~0.1 s of a ~1 s run)
        if (rc == 0) { for (bbduk_handle* hq : devs) bbduk_destroy(hq); bbduk_host_destroy(host); }
        return rc;
    }
    Reader r1, r2;
    if (!r1.open(in1)) return fail("cannot open", in1.c_str());
    if (!in2.empty() && !r2.open(in2)) return fail("cannot open", in2.c_str());
    FILE* fout = out.empty() ? nullptr : fopen(out.c_str(), "w");
    FILE* foutm = outm.empty() ? nullptr : fopen(outm.c_str(), "w");
    FILE* ftsv = tsv.empty() ? nullptr : fopen(tsv.c_str(), "w");
    if ((!out.empty() && !fout) || (!outm.empty() && !foutm) || (!tsv.empty() && !ftsv)) return fail("cannot open an output file", nullptr);

    const bool ktrim = P.mode != BBDUK_MODE_KFILTER;
    struct Shard { int64_t lo = 0, hi = 0; std::vector<uint8_t> bases; std::vector<int64_t> offsets; std::vector<uint32_t> mask; int rc = 0; };
    std::vector<Rec> recs; std::vector<Shard> shards;
    std::vector<int32_t> res, ids, resL, resR, mN, mIds, mCnt; std::vector<uint8_t> fl;
    if (P.mode == BBDUK_MODE_KSPLIT && paired) return fail("ksplit works on unpaired reads (BBDukProcessorS.java:2334)", nullptr);
    bool eof = false; long long nread = 0;
    while (!eof) {
        recs.clear();
        while ((long)recs.size() + (paired ? 2 : 1) <= batch * ndev) {      // mates stay adjacent: reads 2i, 2i+1
            Rec a, b;
            int rc = r1.next(a.name, a.bases, a.quals);
            if (rc == 1) { eof = true; break; }
            if (rc < 0) return fail("malformed FASTQ record in", in1.c_str());
            if (paired) {
                rc = in2.empty() ? r1.next(b.name, b.bases, b.quals) : r2.next(b.name, b.bases, b.quals);
                if (rc != 0) return fail("unpaired or malformed mate for read", a.name.c_str());
            }
            recs.push_back(std::move(a));
            if (paired) recs.push_back(std::move(b));
        }
        if (recs.empty()) break;
        const int64_t n = (int64_t)recs.size();
        res.resize(n); ids.resize(n); fl.resize(n);
        if (P.mode == BBDUK_MODE_KTRIM_TIPS || P.mode == BBDUK_MODE_KSPLIT) resL.resize(n);
        if (P.mode == BBDUK_MODE_KSPLIT) resR.resize(n);
        const bool wantLists = rename && P.findBestMatch;             // rename() :2508-2522 on the lists findBestMatch left
        const int cap = 64;
        if (wantLists) { mN.resize(n); mIds.resize((size_t)n * cap); mCnt.resize((size_t)n * cap); }
        // contiguous blocks of whole pairs, one per handle; a block's buffers are its own, offsets start at 0
        const int64_t units = paired ? n / 2 : n, per = paired ? 2 : 1;
        shards.resize(ndev);
        for (int q = 0; q < ndev; q++) {
            Shard& S = shards[q];
            S.lo = per * (units * q / ndev); S.hi = per * (units * (q + 1) / ndev); S.rc = BBDUK_OK;
            S.bases.clear(); S.offsets.assign(1, 0);
            for (int64_t i = S.lo; i < S.hi; i++) { S.bases.insert(S.bases.end(), recs[i].bases.begin(), recs[i].bases.end()); S.offsets.push_back((int64_t)S.bases.size()); }
            if (S.bases.empty()) S.bases.push_back(0);                // an all-empty block still needs a valid pointer
        }
        auto run_shard = [&](int q) {
            Shard& S = shards[q]; bbduk_handle* h = devs[q];
            const int64_t m = S.hi - S.lo, lo = S.lo;
            if (m == 0) return;
            if (P.mode == BBDUK_MODE_KTRIM_TIPS) S.rc = bbduk_ktrimtips_batch(h, S.bases.data(), S.offsets.data(), m, paired, res.data() + lo, resL.data() + lo, ids.data() + lo, fl.data() + lo);
            else if (P.mode == BBDUK_MODE_KSPLIT) S.rc = bbduk_ksplit_batch(h, S.bases.data(), S.offsets.data(), m, res.data() + lo, resL.data() + lo, resR.data() + lo, ids.data() + lo, fl.data() + lo);
            else if (P.mode == BBDUK_MODE_KMASK) {
                S.mask.assign((size_t)(S.offsets[m] + 31) / 32 + 1, 0u);
                S.rc = bbduk_kmask_batch(h, S.bases.data(), S.offsets.data(), m, paired, res.data() + lo, ids.data() + lo, fl.data() + lo, S.mask.data());
            } else if (wantLists) S.rc = bbduk_kfilter_batch_matches(h, S.bases.data(), S.offsets.data(), m, paired, res.data() + lo, ids.data() + lo, fl.data() + lo, cap,
                                                                  mN.data() + lo, mIds.data() + (size_t)lo * cap, mCnt.data() + (size_t)lo * cap);
            else S.rc = ktrim ? bbduk_ktrim_batch(h, S.bases.data(), S.offsets.data(), m, paired, res.data() + lo, ids.data() + lo, fl.data() + lo)
                              : bbduk_kfilter_batch(h, S.bases.data(), S.offsets.data(), m, paired, res.data() + lo, ids.data() + lo, fl.data() + lo);
        };
        if (ndev == 1) run_shard(0);
        else {
            std::vector<std::thread> th;
            for (int q = 0; q < ndev; q++) th.emplace_back(run_shard, q);
            for (auto& t : th) t.join();
        }
        for (int q = 0; q < ndev; q++) if (shards[q].rc != BBDUK_OK) return fail("batch operator", bbduk_last_error(devs[q]));
        if (wantLists) for (int64_t i = 0; i < n; i++) {
            for (int j = 0; j < mN[i]; j++) {
                const char* nm = ""; bbduk_host_scaffold_info(host, mIds[(size_t)i * cap + j], &nm, nullptr);
                recs[i].name += '\t'; recs[i].name += nm; recs[i].name += '='; recs[i].name += std::to_string(mCnt[(size_t)i * cap + j]);
            }
        }
        int shardOf = 0;
        for (int64_t i = 0; i < n; i++) {
            Rec& r = recs[i];
            while (i >= shards[shardOf].hi) shardOf++;
            const Shard& S = shards[shardOf];
            const int L = (int)r.bases.size();
            int left = 0, right = 0;                                 // TrimRead.trimByAmount(r, left, right, 1) with the returned amount
            if (P.mode == BBDUK_MODE_KTRIM_R) right = res[i]; else if (P.mode == BBDUK_MODE_KTRIM_L) left = res[i];
            else if (P.mode == BBDUK_MODE_KTRIM_TIPS) { right = res[i]; left = resL[i]; res[i] += resL[i]; }     // the tsv shows the sum
            // ksplit: the span lies inside the read (two pieces) iff it touches neither end; the pieces leave through outm -- or stay, with
            // trimfailuresto1bp, where nothing is evicted (:1431)
            const bool split = P.mode == BBDUK_MODE_KSPLIT && resL[i] > 0 && resR[i] >= 0 && resR[i] != L - 1;
            if (P.mode == BBDUK_MODE_KSPLIT && res[i] > 0 && !split) {                              // :2485-2490: the match touches an end
                if (resL[i] == 0) left = res[i]; else right = res[i];
            }
            if (split) {                                                                            // :2491-2498: two pieces, a pair
                const int lm = resL[i], rm = resR[i];
                if (ftsv) fprintf(ftsv, "%s\t%d\t%d\t%d\t%d\t%d\n", r.name.c_str(), L, res[i], ids[i], L - res[i], (int)fl[i]);
                FILE* const foutm0 = foutm;
                FILE* const fpieces = (fl[i] & BBDUK_FLAG_REMOVED) ? foutm0 : fout;
                if (fpieces) {
                    fprintf(fpieces, "@%s\n", r.name.c_str());
                    fwrite(r.bases.data(), 1, (size_t)lm, fpieces); fputs("\n+\n", fpieces);
                    fwrite(r.quals.data(), 1, (size_t)lm, fpieces); fputc('\n', fpieces);
                    const int n2 = (L - 1) - (rm + 1);                                             // subRead(rightmost+1, length-1)
                    fprintf(fpieces, "@%s\n", r.name.c_str());
                    fwrite(r.bases.data() + rm + 1, 1, (size_t)n2, fpieces); fputs("\n+\n", fpieces);
                    fwrite(r.quals.data() + rm + 1, 1, (size_t)n2, fpieces); fputc('\n', fpieces);
                }
                continue;
            }
            int newLen = L - left - right;
            if (P.trimFailuresTo1bp && (fl[i] & BBDUK_FLAG_DISCARDED) && newLen > 1) { right = L - left - 1; newLen = 1; }   // setDiscarded (:1464-1470)
            if (P.mode == BBDUK_MODE_KMASK && res[i] > 0) {            // :2309-2320
                for (int b = 0; b < L; b++) {
                    const int64_t g = S.offsets[i - S.lo] + b;
                    if (!((S.mask[(size_t)(g >> 5)] >> (g & 31)) & 1u)) continue;
                    if (maskLower) r.bases[b] = (char)tolower((unsigned char)r.bases[b]);
                    else { r.bases[b] = maskSymbol; if (maskSymbol == 'N') r.quals[b] = '!'; }
                }
            }
            if (ftsv) fprintf(ftsv, "%s\t%d\t%d\t%d\t%d\t%d\n", r.name.c_str(), L, res[i], ids[i], newLen, (int)fl[i]);
            FILE* dst = (fl[i] & BBDUK_FLAG_REMOVED) ? foutm : fout;
            if (dst) {
                fprintf(dst, "@%s\n", r.name.c_str());
                fwrite(r.bases.data() + left, 1, (size_t)newLen, dst); fputs("\n+\n", dst);
                fwrite(r.quals.data() + left, 1, (size_t)newLen, dst); fputc('\n', dst);
            }
        }
        nread += n;
    }
    if (!r1.close()) return fail("gzip reported an error (truncated or corrupt input?) on", in1.c_str());
    if (!r2.close()) return fail("gzip reported an error (truncated or corrupt input?) on", in2.c_str());
    if (fout) fclose(fout);
    if (foutm) fclose(foutm);
    if (ftsv) fclose(ftsv);

    if (ndev > 1) {       // the path's only exchange: one all-reduce (sum, int64) of the counter vectors over the devices
        if (bbduk_comm_create_local(devs.data(), ndev) != BBDUK_OK || bbduk_allreduce_counters_local(devs.data(), ndev) != BBDUK_OK)
            return fail("counter all-reduce", bbduk_last_error(dev));
    }
    std::vector<int64_t> c((size_t)bbduk_counters_len(dev));
    bbduk_get_counters(dev, c.data(), (int32_t)c.size());
    // the lines BBDukS prints at the end of a run (bbduk/BBDukS.java:350-420), same wording for the shared counters
    fprintf(stderr, "Input:                  \t%lld reads \t\t%lld bases.\n", (long long)c[BBDUK_READS_IN], (long long)c[BBDUK_BASES_IN]);
    if (P.mode == BBDUK_MODE_KMASK) fprintf(stderr, "KMasked:                \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KTRIMMED], (long long)c[BBDUK_BASES_KTRIMMED]);
    else if (ktrim) fprintf(stderr, "KTrimmed:               \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KTRIMMED], (long long)c[BBDUK_BASES_KTRIMMED]);
    else fprintf(stderr, "Contaminants:           \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_KFILTERED], (long long)c[BBDUK_BASES_KFILTERED]);
    fprintf(stderr, "Total Removed:          \t%lld reads \t%lld bases\n", (long long)(c[BBDUK_READS_IN] - c[BBDUK_READS_OUTU]),
            (long long)(c[BBDUK_BASES_IN] - c[BBDUK_BASES_OUTU]));
    fprintf(stderr, "Result:                 \t%lld reads \t%lld bases\n", (long long)c[BBDUK_READS_OUTU], (long long)c[BBDUK_BASES_OUTU]);
    if (c[BBDUK_CTR_STATUS]) return fail("device reported an error status", nullptr);
    if (write_stats(so, host, c, in1, in2) != 0) return 1;
    for (bbduk_handle* hq : devs) bbduk_destroy(hq);
    bbduk_host_destroy(host);
    return 0;                                                     // (a watchdog stays armed through the exit handlers: the HIP / RCCL teardown is a suspect too)
}
