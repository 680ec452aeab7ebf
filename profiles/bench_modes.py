#!/usr/bin/env python3
"""Secondary measurement: the secondary operators (ktrim=n, ktrim=rl, ksplit) on the configs[1] reads, device-resident,
operator time by HIP events around the call.  One JSON line per mode."""
import argparse, json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import ctypes as C


ADAPTER_MODES = (("ktrim=n k=23 mink=11 hdist=1 ref=adapters", "kmask"), ("ktrim=rl k=23 mink=11 hdist=1 ref=adapters", "tips"),
                 ("ksplit=t k=23 mink=11 hdist=1 ref=adapters", "ksplit"),
                 ("k=23 hdist=1 fbm ref=adapters", "fbm"), ("k=23 hdist=1 rename=t ref=adapters", "fbm+matches"),
                 ("k=40 hdist=1 ref=phix", "k40"),
                 # round 6: query-side expansion, tabulated at the end of the table build and looked up by the stream scans
                 ("ktrim=r k=21 mink=11 qhdist=1 ref=adapters", "qhdist_ktrim"), ("k=31 qhdist=1 ref=phix", "qhdist_kfilter_mm"),
                 ("ktrim=r k=21 mink=11 hdist=1 qhdist=1 ref=adapters", "qhdist_x_hdist"))
# round 5: the same operators against the guide's genome-size example (a random 4.6 Mbase genome, BBDukGuide.txt:34), i.e. a
# big-layout map scanned by bbduk_bigs_every_kernel; 1 % of the pairs drawn from the genome
GENOME_MODES = (("ktrim=n k=31", "kmask"), ("ktrim=rl k=31", "tips"), ("ksplit=t k=31", "ksplit"), ("k=31 fbm", "fbm"), ("k=31 mkf=0.1", "mkf"), ("k=31 mcf=0.3", "mcf"),
                ("ktrim=l k=31", "ktriml"), ("k=31 mbk=1", "mbk"), ("k=40", "k40"))


def run_genome(reads=20_000_000, genome_bases=4_600_000, modes=GENOME_MODES):
    import numpy as np
    genome = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(77).integers(0, 4, genome_bases)].tobytes()
    return run(reads, 150, modes=modes, genome=genome)


def run(reads=20_000_000, read_len=150, modes=ADAPTER_MODES, genome=None):
    """-> one dict per operator (bench.py's `operators` block calls this too)"""
    import torch
    from bbtools_amd import bbduk as B
    n = reads // 2 * 2
    RL = read_len
    out = []
    d_bases = torch.empty(n * RL, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    if genome is not None: sp = B.synth_params(9, contam=genome[:100_000], contam_frac=0.01)
    else: sp = B.synth_params(2) if RL == 150 else B.synth_params(2, read_len=RL, ins_min=RL // 3, ins_max=RL * 7 // 3)
    B.synth_generate_device(sp, 0, n // 2, d_bases, d_off, 0)
    L = B.lib()
    for args, mode in modes:
        duk = B.BBDuk(args, refs=[genome], build="device") if genome is not None else B.BBDuk(args)
        d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_b = torch.empty(n, dtype=torch.int32, device="cuda"); d_c2 = torch.empty(n, dtype=torch.int32, device="cuda")
        d_i = torch.empty(n, dtype=torch.int32, device="cuda"); d_f = torch.empty(n, dtype=torch.uint8, device="cuda")
        d_m = torch.zeros(n * RL // 32 + 8, dtype=torch.int32, device="cuda")
        d_mi = torch.zeros(n * 8 if mode == "fbm+matches" else 1, dtype=torch.int32, device="cuda"); d_mc = torch.zeros_like(d_mi)
        d_ctr = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
        h = duk.gpu.h
        ts = []
        for _ in range(4):
            d_ctr.zero_(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            if mode == "kmask":
                rc = L.bbduk_kmask_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_m.data_ptr(), d_ctr.data_ptr(), None)
            elif mode == "tips":
                rc = L.bbduk_ktrimtips_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_b.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_ctr.data_ptr(), None)
            elif mode in ("fbm", "k40", "mkf", "mcf", "mbk", "qhdist_kfilter_mm"):
                rc = L.bbduk_kfilter_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_ctr.data_ptr(), None)
            elif mode in ("ktriml", "qhdist_ktrim", "qhdist_x_hdist"):
                rc = L.bbduk_ktrim_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_ctr.data_ptr(), None)
            elif mode == "fbm+matches":                       # the lists rename=t prints, eight entries per read
                rc = L.bbduk_kfilter_batch_matches_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(),
                                                          8, d_b.data_ptr(), d_mi.data_ptr(), d_mc.data_ptr(), d_ctr.data_ptr(), None)
            else:
                rc = L.bbduk_ksplit_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, d_a.data_ptr(), d_b.data_ptr(), d_c2.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_ctr.data_ptr(), None)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0, rc
            ts.append(e0.elapsed_time(e1))
        ms = min(ts[1:]); c = d_ctr.cpu().numpy()
        out.append({"mode": mode, "args": args if genome is None else args + " vs a random %d-base genome" % len(genome), "keys": int(duk.stored_kmers), "reads": n, "read_len": RL, "operator_ms": round(ms, 2), "Gbases_per_s": round(n * RL / ms / 1e6, 1), "readsKTrimmed": int(c[2]), "readsKFiltered": int(c[4]), "status": int(c[15])})
        duk.close()
    return out


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=0, help="bases of a random genome as the reference (0: the adapter / phiX references)")
    a = ap.parse_args()
    for r in (run_genome(a.reads, a.genome) if a.genome else run(a.reads, a.read_len)):
        print(json.dumps(r))


if __name__ == "__main__":
    main()
