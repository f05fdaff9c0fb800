#!/usr/bin/env python3
"""bench.py -- k-mer ID queries/s of the MI355X-native Blight hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the query hot path (canonical k-mer -> minimizer -> bucket -> BBHash rank -> position field ->
2^b window compare -> ID) over one batch of synthetic query k-mers that is already resident in HBM.  The index is a
seeded synthetic unitig set (BASELINE has no offline genomes: only lambda ships with the reference) built on the GPU by
the same library.  Rank 0 prints ONE JSON line.

Workload naming: C3 = chr1-scale 2.3e8 31-mers (k31 m9 n7 b6), C4 = human-scale 2.5e9 31-mers
(k31 m9 n14 b6), C5 = 5e8 63-mers (k63 m9 n14 b8, 128-bit k-mers); 50 % of the queries are indexed k-mers, 50 % uniform
random k-mers.  --skew draws 10 % of the unitigs from a 2-symbol source (uneven minimizer buckets).

Besides the headline (`value`: the query pipeline on k-mers resident in HBM) a single-GPU run reports, under extra keys:
  roofline      the dominant kernel (k_query) by HIP events on the launching stream, per-kernel times of the whole pipeline,
                fractions of the 8 TB/s spec, of the 6.29 TB/s measured copy rate and of the measured random-line ceiling,
                HBM traffic and VALU issue from the rocprofv3 PMC summary of this source tree (profiles/, when present)
  cpu_baseline  the CPU oracle (restated reference algorithm) on the host cores, IDs compared element-wise (both k-mer widths)
  cpu_reference the REAL reference (-t 1) on a bounded sample of the same input
  reads         the sequence front end on 150-nt reads (half cut from the indexed text, half foreign): reads resident in HBM,
                blmx_query_sequences on host buffers, file_query on a FASTA file, one query_sequence_hash call per read
                from OpenMP threads through the C++ facade (the reference's calling pattern)
  host_buffers  blmx_query_kmers / blmx_query_membership on host arrays (PCIe and host copies included)

N > 1: `--mode sharded` (default) partitions the MPHF groups (contiguous minimizer-bucket ranges) over the ranks and routes
every query k-mer to its owner with an RCCL all-to-all over xGMI (and the IDs back); `--mode replicas` keeps a full index
per GPU with no data-path collective (upper bound).  Per-GPU query batch is fixed: weak scaling.  A sharded run also
measures the replicas layout right after the timed region and reports it under `replicas`; `value` is the sharded number.
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {31: 145, 63: 209}  # algorithmic bytes per query, SURVEY.md section 8(d): B(k=31,b=6)=145, B(k=63,b=8)=209
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
HBM_COPY_GBS = 6290.0            # MI355X_MICROARCH.md: measured float4 copy

WORKLOADS = {
    # name: (k, m, log2_mphf, b, indexed k-mers, queries per step per GPU)
    "C2": (31, 9, 7, 6, 4_600_000, 4_600_000),
    "C3": (31, 9, 7, 6, 230_000_000, 10_000_000),
    "C3n14": (31, 9, 14, 6, 230_000_000, 10_000_000),
    "C4": (31, 9, 14, 6, 2_500_000_000, 100_000_000),
    "C5": (63, 9, 14, 8, 500_000_000, 100_000_000),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("BLIGHT_BENCH_WORKLOAD", "C4"), choices=sorted(WORKLOADS))
    ap.add_argument("--kmers", type=int, default=0, help="override the number of indexed k-mers")
    ap.add_argument("--queries", type=int, default=0, help="override the queries per step per GPU")
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replicas"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip every CPU leg)")
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--no-replicas-leg", action="store_true", help="N > 1, sharded: skip the secondary replicas measurement")
    ap.add_argument("--no-slice-build", action="store_true", help="N > 1, native exchange: skip the construction-from-slices leg")
    ap.add_argument("--rehearse-sharded", action="store_true",
                    help="run the N > 1 code path (native RCCL exchange, construction from slices, replicas leg) with a world of one rank")
    ap.add_argument("--present", type=float, default=0.5, help="fraction of the queries that are indexed k-mers (headline: 0.5)")
    ap.add_argument("--skew", nargs="?", const="dup", default=None, choices=["dup", "dedup"],
                    help="every tenth unitig from a 2-symbol source (giant buckets).  dup (default): as generated -- repeated k-mers, hence "
                         "fallback keys; dedup: repeated k-mers removed first, what BCALM2 unitigs look like (distinct k-mers, giant buckets)")
    ap.add_argument("--light", action="store_true",
                    help="the index at the reference's footprint: no absent-key filter (filter_bits = 0), 16-B MPHF lines")
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads of the sequence front-end leg (0 = skip)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--no-host-legs", action="store_true", help="skip the host-buffer / file / per-read-call measurements")
    ap.add_argument("--batches", type=int, default=4, help="independently seeded query batches rotated through the timed loop")
    ap.add_argument("--no-fasta-leg", action="store_true", help="skip construct_index(file) on files written to tmpfs")
    ap.add_argument("--tune", action="append", default=[], help=argparse.SUPPRESS)  # experiments: name=value passed to blmx_set_tuning before the timed steps
    ap.add_argument("--exp-sort", type=int, default=-1, help=argparse.SUPPRESS)  # experiment: order every batch by (MPHF group, top N bits of the level-0 hash)
    ap.add_argument("--no-excursion", action="store_true", help="skip the all-absent / all-indexed batches after the timed steps (profiling runs: "
                                                                "every launch of the query kernels is then a headline step)")
    return ap.parse_args()


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def synth_unitigs(torch, dev, n_kmers, k, seed, skew=None):
    """Seeded random unitigs in HBM: lengths k-1+Geometric(mean 80), uniform ACGT; with `skew` every tenth unitig
    is drawn from the 2-symbol source {A, C}.  Returns (ascii u8, offsets i64, total)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n_unitigs = max(1, n_kmers // 80)
    extra = torch.empty(n_unitigs, device=dev, dtype=torch.float32).geometric_(1.0 / 80.0, generator=g).to(torch.int64).clamp_(min=1)
    # trim so that the k-mer total is exactly n_kmers
    csum = torch.cumsum(extra, 0)
    keep = int(torch.searchsorted(csum, torch.tensor([n_kmers], device=dev, dtype=torch.int64)).item()) + 1
    keep = min(keep, n_unitigs)
    extra = extra[:keep].clone()
    over = int(csum[keep - 1].item()) - n_kmers
    if over > 0:
        extra[keep - 1] -= over
    if int(extra[keep - 1].item()) < 1:
        extra = extra[:keep - 1]
    lens = extra + (k - 1)
    offs = torch.zeros(lens.numel() + 1, device=dev, dtype=torch.int64)
    torch.cumsum(lens, 0, out=offs[1:])
    total = int(offs[-1].item())
    ascii_ = torch.empty(total + 64, device=dev, dtype=torch.uint8)
    chunk = 1 << 28
    for s in range(0, total, chunk):
        e = min(total, s + chunk)
        c = torch.randint(0, 4, (e - s,), device=dev, dtype=torch.uint8, generator=g)
        if skew:
            u = torch.searchsorted(offs, torch.arange(s, e, device=dev, dtype=torch.int64), right=True) - 1
            c = torch.where((u % 10) == 3, c & 1, c)  # codes 0/1 -> A/C only
            del u
        # 0,1,2,3 -> A C G T  (65 67 71 84)
        ascii_[s:e] = 65 + 2 * c + 2 * (c == 2).to(torch.uint8) + 13 * (c == 3).to(torch.uint8)
    ascii_[total:] = 0
    return ascii_, offs, total


def synth_unitigs_dedup(torch, dev, blight_amd, n_kmers, k, seed):
    """The realistic skew (--skew dedup): nine tenths of the k-mers in ordinary unitigs, one tenth in unitigs over {A, C} from which every
    REPEATED k-mer has been removed -- a compacted de Bruijn graph (BCALM2, the reference's input, blight.cpp:212-229) never holds a
    k-mer twice, but low-complexity sequence still piles millions of DISTINCT k-mers under a few minimizers: giant buckets and MPHF
    groups without the duplicate keys (fallback entries) that the plain --skew makes.  The {A, C} unitigs are cut wherever a k-mer
    that occurred before is dropped; runs of kept k-mers become the unitigs.  Returns (ascii u8, offsets i64, total)."""
    n_ac = n_kmers // 10
    a1, o1, t1 = synth_unitigs(torch, dev, n_kmers - n_ac, k, seed, None)
    # {A, C} unitigs with ~8 % more k-mers than needed (a 31-mer over two symbols has 2^31 values: ~6 % of 2.5e8 draws repeat)
    g = torch.Generator(device=dev)
    g.manual_seed(seed * 31 + 5)
    want = int(n_ac * 1.08) + 1000
    nu = max(1, want // 80)
    extra = torch.empty(nu, device=dev, dtype=torch.float32).geometric_(1.0 / 80.0, generator=g).to(torch.int64).clamp_(min=1)
    lens = extra + (k - 1)
    o2 = torch.zeros(nu + 1, device=dev, dtype=torch.int64)
    torch.cumsum(lens, 0, out=o2[1:])
    t2 = int(o2[-1].item())
    c = torch.randint(0, 2, (t2,), device=dev, dtype=torch.uint8, generator=g)
    a2 = (65 + 2 * c).contiguous()  # A / C
    del c
    # every k-mer start, its canonical k-mer, first occurrences
    koff = torch.zeros(nu + 1, device=dev, dtype=torch.int64)
    torch.cumsum(extra, 0, out=koff[1:])
    K = int(koff[-1].item())
    which = torch.repeat_interleave(torch.arange(nu, device=dev), extra)
    ar = torch.arange(K, device=dev, dtype=torch.int64)
    pos = (o2[:-1][which] + (ar - koff[:-1][which])).contiguous()
    first_of_unitig = ar == koff[:-1][which]
    del which
    helper = blight_amd.kmer_Set_Light(k, 9 if k >= 11 else 3, 4, 2, 1, 6, device=dev.index or 0)
    km = torch.empty(K, device=dev, dtype=torch.int64)
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    helper.kmers_at_device(a2.data_ptr(), pos.data_ptr(), K, km.data_ptr(), err.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    helper.close()
    vals, idx = torch.sort(km, stable=True)
    dup_sorted = torch.zeros(K, device=dev, dtype=torch.bool)
    dup_sorted[1:] = vals[1:] == vals[:-1]
    keep = torch.ones(K, device=dev, dtype=torch.bool)
    keep[idx] = ~dup_sorted
    del vals, idx, dup_sorted, km
    # trim to exactly n_ac kept k-mers
    ck = torch.cumsum(keep.to(torch.int64), 0)
    keep &= ck <= n_ac
    del ck
    prev_kept = torch.zeros(K, device=dev, dtype=torch.bool)
    prev_kept[1:] = keep[:-1]
    start_run = keep & (first_of_unitig | ~prev_kept)
    run_of = torch.cumsum(start_run.to(torch.int64), 0) - 1
    nruns = int(start_run.sum().item())
    run_len = torch.bincount(run_of[keep], minlength=nruns)
    run_pos = pos[start_run]
    del prev_kept, first_of_unitig, run_of, keep, start_run, pos, ar
    new_lens = run_len + (k - 1)
    o3 = torch.zeros(nruns + 1, device=dev, dtype=torch.int64)
    torch.cumsum(new_lens, 0, out=o3[1:])
    t3 = int(o3[-1].item())
    r_of = torch.repeat_interleave(torch.arange(nruns, device=dev), new_lens)
    src = run_pos[r_of] + (torch.arange(t3, device=dev, dtype=torch.int64) - o3[:-1][r_of])
    a3 = a2[src]
    del r_of, src, a2
    ascii_ = torch.empty(t1 + t3 + 64, device=dev, dtype=torch.uint8)
    ascii_[:t1] = a1[:t1]
    ascii_[t1:t1 + t3] = a3
    ascii_[t1 + t3:] = 0
    offs = torch.cat([o1, o3[1:] + t1])
    return ascii_, offs, t1 + t3


def synth_queries(torch, ix, dev, ascii_, offs, total, k, nq, seed, kmer_words, present=0.5):
    """`present` (default 50 %) indexed k-mers (uniform over positions), the rest uniform random k-mers, shuffled;
    canonical, in HBM."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    npos = int(nq * present)
    p = torch.randint(0, total, (max(npos, 1),), device=dev, dtype=torch.int64, generator=g)
    u = torch.searchsorted(offs, p, right=True) - 1
    p = torch.minimum(p, offs[u + 1] - k)
    q = torch.empty((nq, kmer_words), device=dev, dtype=torch.int64)
    pos_k = torch.empty((max(npos, 1), kmer_words), device=dev, dtype=torch.int64)
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    ix.kmers_at_device(ascii_.data_ptr(), p.data_ptr(), max(npos, 1), pos_k.data_ptr(), err.data_ptr(), torch.cuda.current_stream().cuda_stream)
    neg = torch.randint(-(1 << 63), (1 << 63) - 1, (max(nq - npos, 1), kmer_words), device=dev, dtype=torch.int64, generator=g)
    ix.canonicalize_device(neg.data_ptr(), max(nq - npos, 1), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    q[:npos] = pos_k[:npos]
    q[npos:] = neg[: nq - npos]
    perm = torch.randperm(nq, device=dev, generator=g)
    # shuffle word column by word column on contiguous 1-D tensors (a 2-D row gather of a (1e8, 2) int64 tensor returned
    # wrong rows on this torch/ROCm build; the 1-D gather is the well-trodden path)
    cols = [q[:, w].contiguous()[perm] for w in range(kmer_words)]
    q = torch.stack(cols, dim=1).contiguous()
    is_pos = (perm < npos)
    return q, is_pos


def synth_reads(torch, dev, ascii_, total, k, nreads, read_len, present, seed):
    """`nreads` reads of `read_len` characters in HBM: `present` of them are windows of the indexed unitig text (a window may
    run over a unitig end: the k-mers across the junction are then foreign, as with real reads over graph junctions), the rest
    uniform random ACGT; shuffled.  Returns (chars u8 [nreads * read_len], offsets i64 [nreads + 1], nreads, number of k-mers)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    npos = int(nreads * present)
    rows = torch.empty((nreads, read_len), device=dev, dtype=torch.uint8)
    ar = torch.arange(read_len, device=dev, dtype=torch.int64)
    if npos:
        starts = torch.randint(0, max(1, total - read_len), (npos,), device=dev, dtype=torch.int64, generator=g)
        step = 1 << 18
        for a in range(0, npos, step):
            z = min(npos, a + step)
            rows[a:z] = ascii_[(starts[a:z, None] + ar[None, :]).reshape(-1)].reshape(z - a, read_len)
    if nreads > npos:
        c = torch.randint(0, 4, (nreads - npos, read_len), device=dev, dtype=torch.uint8, generator=g)
        rows[npos:] = 65 + 2 * c + 2 * (c == 2).to(torch.uint8) + 13 * (c == 3).to(torch.uint8)
    perm = torch.randperm(nreads, device=dev, generator=g)
    chars = rows[perm].reshape(-1).contiguous()
    offs = torch.arange(nreads + 1, device=dev, dtype=torch.int64) * read_len
    return chars, offs, nreads, nreads * max(0, read_len - k + 1)


def footprint(st):
    """bits per indexed k-mer, array by array, next to the figures the reference's recap prints (blight.cpp:371-378: the packed graph,
    the positions, and "+ 4" for the BBHash it does not measure there)"""
    nk = max(1, st["local_number_kmer"])
    seq = 2.0 * st["total_nt"] / nk
    pos = st["positions_total_size"] / nk
    mphf = 8.0 * st["mphf_bytes"] / nk
    total = 8.0 * st["device_bytes"] / nk
    return {"bits_per_kmer": {"packed_graph": round(seq, 3), "positions_reference_layout": round(pos, 3),
                              "mphf_lines_incl_ranks_and_inline_filter": round(mphf, 3), "everything_in_hbm": round(total, 3)},
            "reference_recap_bits_per_kmer": {"packed_graph": round(seq, 3), "positions": round(pos, 3), "with_bbhash_estimate": round(seq + pos + 4.0, 3)},
            "what": "everything_in_hbm = all device arrays of the finished index (descriptors, fallback table, filter, padding); the reference's "
                    "own figure is its recap's `TOTAL Bits per kmer (with bbhash)` = packed graph + positions + 4"}


def host_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box gives one GPU's
    share of the host, not all of its cores)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def export_index(ix, st, k, b):
    """D2H the finished index as numpy arrays for the oracle's CPU query path."""
    import numpy as np
    arrays = {name: ix.export(name) for name in ("bucket_nt", "bucket_start", "positions", "g_nkeys", "g_prefix", "g_posstart",
                                                   "g_nbits", "level_sizes", "g_wbase", "mphf_bits", "fb_keys", "fb_ranks", "fb_group")}
    seq = ix.export("seq")
    pad = ((1 << b) + k) // 32 + 8
    arrays["seq"] = np.concatenate([seq, np.zeros(pad, np.uint64)])
    return arrays


def import_oracle(ix, st, k, m, n, b):
    """The CPU oracle's query path over the index image exported from HBM (test infrastructure: the checker and the CPU leg)."""
    from oracle import oracle as O
    t0 = time.time()
    arrays = export_index(ix, st, k, b)
    orc = O.ImportedOracle(k, m, n, b, st["total_nt"], st["number_kmer"], st["positions_total_size"], arrays, kmer_bytes=st["kmer_bytes"])
    return orc, time.time() - t0


def cpu_baseline(orc, t_import, q_host, gpu_ids_host, budget_s):
    """Time the CPU oracle (restated reference algorithm) on the box's host cores over a bounded sample of the same
    queries on the same index and cross-check the IDs element-wise with the GPU's.  q_host: (nq,) uint64 or (nq, 2)."""
    threads = host_cores()
    batch = 200_000
    done = 0
    elapsed = 0.0
    mismatches = 0
    # 1-thread rate on a small slice (the reference is only valid at -t 1), then all cores
    t = time.time()
    ids1 = orc.query_kmers(q_host[:batch], threads=1)
    t1 = time.time() - t
    mismatches += int((ids1 != gpu_ids_host[:batch]).sum())
    while elapsed < budget_s and done < q_host.shape[0]:
        e = min(q_host.shape[0], done + batch * threads)
        t = time.time()
        ids = orc.query_kmers(q_host[done:e], threads=threads)
        elapsed += time.time() - t
        mismatches += int((ids != gpu_ids_host[done:e]).sum())
        done = e
    return {
        "value": done / elapsed if elapsed > 0 else 0.0, "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": "first %d queries of the same batch on the same index (image exported from HBM, import %.1f s); "
                  "oracle = plain-C restatement of the reference, OpenMP over queries" % (done, t_import),
        "single_thread_value": batch / t1, "ids_equal_to_gpu": mismatches == 0, "mismatches": mismatches,
    }


def cpu_reference_calibration(sample_ascii, sample_offs, k, m, n, b, nq=2_000_000):
    """The REAL reference (oracle/_ref: Piezoid/Blight + the three one-token fixes, compiled from /root/reference in the dev
    container, -t 1) timed on this host next to the oracle port, both on a bounded sample of the workload: the first unitigs
    of the same synthetic input and the same 50/50 query mix.  Calibrates the `port` numbers against the reference itself."""
    import numpy as np
    from oracle import oracle as O
    if not O.ref_available() or k > 32:
        return None
    o = sample_offs
    seqs = [sample_ascii[o[i]:o[i + 1]].tobytes().decode() for i in range(len(o) - 1)]
    t = time.time()
    ref = O.Reference(k, m, n, b, seqs)
    t_ref_build = time.time() - t
    t = time.time()
    orc = O.Oracle(k, m, n, b, seqs)
    t_orc_build = time.time() - t
    nk = ref.stats["number_kmer"]
    rng = np.random.default_rng(5)
    pos = np.concatenate([orc.canonical_kmers(s) for s in seqs[: max(1, nq // 2 // 80)]])[: nq // 2]
    neg = rng.integers(0, 1 << 62, size=nq - pos.size, dtype=np.uint64)
    neg = np.array([min(int(x), O.p_rc(int(x), k)) for x in neg[:20000]] * ((nq - pos.size) // 20000 + 1), dtype=np.uint64)[: nq - pos.size]
    q = np.concatenate([pos, neg])
    q = np.ascontiguousarray(q[rng.permutation(q.size)])
    t = time.time()
    ids_ref = ref.query_kmers(q)
    t_ref = time.time() - t
    t = time.time()
    ids_orc = orc.query_kmers(q, threads=1)
    t_orc = time.time() - t
    out = {"value": q.size / t_ref, "unit": "queries/s", "cores": 1, "kind": "reference",
           "sample": "patched reference (-t 1) on the first %d unitigs (%d k-mers) of the same synthetic input, %d queries 50/50; "
                     "index built by the reference itself in %.1f s (%.2f M k-mers/s)" % (len(seqs), nk, q.size, t_ref_build, nk / t_ref_build / 1e6),
           "port_single_thread_same_sample": q.size / t_orc, "port_build_kmers_per_s": nk / t_orc_build,
           "ids_equal_to_port": bool(np.array_equal(ids_ref, ids_orc))}
    ref.close()
    orc.close()
    return out


def source_digest():
    """sha256 over the kernel sources: ties a PMC summary under profiles/ to the source tree it was measured on"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "blight_amd", "csrc", "*.h*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc(workload):
    """rocprofv3 PMC summary of the bench command of this workload (tools/pmc_collect.sh -> profiles/r*_pmc_<workload>.json),
    newest round first; None when there is none."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload.lower()))):
        best = f
    if not best:
        return None
    try:
        d = json.load(open(best))
        d["file"] = os.path.relpath(best, ROOT)
        return d
    except Exception:
        return None


def stage_profile(ix, fn, reps=5):
    """average HIP-event stage times (ms) of `reps` calls of fn (a device-resident query call of this index)"""
    ix.set_profiling(1)
    acc = {}
    for _ in range(reps):
        fn()
        for key, v in ix.last_profile().items():
            acc[key] = acc.get(key, 0.0) + v / reps
    ix.set_profiling(0)
    return acc


def roofline_block(args, k, kw, b, nq, stages, step_ms, st):
    """`roofline` of the JSON line: the kernel that takes the largest share of a step (by HIP events on the launching stream), with ITS
    algorithmic bytes; the query kernel with the reference's bytes next to it whichever dominates; the whole pipeline; and
    what the PMC summary adds (counter traffic and its fraction of the peak)."""
    algo = ALGO_BYTES.get(k, 145)
    kin = 8 * kw
    # algorithmic bytes per k-mer of the passes around the query kernel: k-mer in + k-mer, minimizer, 2-B tile position and 1.5 B of run
    # table out (partition); tile position, run table and a 4-B ID in, 8-B ID out (gather)
    stage_info = {
        "k_query": ("k_query_slots<%d,%d>" % (kw, b), algo, "the query body on the partitioned batch (bucket, filter, MPHF probes, position, 2^b window): latency-bound gathers"),
        "k_bin_scatter": ("k_bin_onepass<%d>" % kw, 2 * kin + 4 + 3.5, "one-pass partition: %d m-mer hashes per k-mer, tile sort in LDS -- VALU-issue-bound, not HBM-bound" % (k - 9 + 1)),
        "k_gather_ids": ("k_gather_runs<%d>" % (8192 // kw), 3.5 + 4 + 8, "IDs back into the caller's order through LDS"),
    }
    timed = {a: v for a, v in stages.items() if a in stage_info and v > 0}
    dom = max(timed, key=timed.get) if timed else "k_query"
    dom_ms = timed.get(dom, step_ms)
    dname, dbytes, dwhat = stage_info[dom]
    kq_ms = stages.get("k_query", 0.0) or step_ms
    achieved = dbytes * nq / (dom_ms * 1e-3) / 1e9
    kq_ach = algo * nq / (kq_ms * 1e-3) / 1e9
    pipe_ach = algo * nq / (step_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "kernel": "%s: %s (the stage with the largest share of a step: %.3f of %.3f ms)" % (dname, dwhat, dom_ms, step_ms), "kernel_ms": dom_ms,
           "algorithmic_bytes_per_query": dbytes, "frac_of_measured_copy_rate": achieved / HBM_COPY_GBS,
           "per_kernel_ms": {a: round(v, 4) for a, v in stages.items()},
           "query_kernel": {"kernel": "k_query_slots<%d,%d>" % (kw, b), "kernel_ms": kq_ms, "algorithmic_bytes_per_query": algo, "achieved": kq_ach,
                            "frac": kq_ach / HBM_PEAK_GBS, "note": "the reference's bytes per query (SURVEY 8d) over the query kernel's time; the absent-key "
                            "filter makes the kernel move fewer bytes than that: see traffic_frac"},
           "pipeline": {"kernels": "k_plan_make + k_bin_onepass [stage k_bin_scatter] + k_query_slots [stage k_query] + k_gather_runs [stage k_gather_ids] + "
                                   "k_plan_learn (what `value` times; stages k_bin_hist / scan are empty once the bin shares are learnt)", "ms": step_ms,
                        "achieved": pipe_ach, "frac": pipe_ach / HBM_PEAK_GBS, "frac_of_measured_copy_rate": pipe_ach / HBM_COPY_GBS}}
    out["which_fraction_is_which"] = ("frac = the REFERENCE's bytes per query (SURVEY 8d) x queries / kernel time / 8 TB/s: work done per second in the "
                                      "reference's currency; traffic_frac (when a PMC summary of this source tree is under profiles/) = bytes the "
                                      "counters saw the kernel move / kernel time / 8 TB/s: what the memory system really carried -- lower, because "
                                      "the absent-key filter answers the absent half after one line; all_indexed.frac = frac on a batch where every "
                                      "query does the reference's full work")
    pmc = load_pmc(args.workload)
    if pmc and not args.skew and not args.kmers and not args.queries and args.present == 0.5 and not args.light:
        kern = pmc.get("kernels", {})
        out["pmc"] = {"file": pmc["file"], "source_digest_matches": pmc.get("source_digest") == source_digest()}
        cands = [(v.get("trace_avg_ns", 0.0) * v.get("trace_calls", 1), name) for name, v in kern.items() if name.startswith("k_query")]
        kq = kern[max(cands)[1]] if cands else None
        dk = [v for name, v in kern.items() if name.startswith(dname.split("<")[0])]
        if dk:  # counter traffic of the dominant kernel: coalesced streams count at half their bytes in FETCH_SIZE on gfx950
            corr = 2.0 if dom == "k_bin_scatter" else 1.0
            out["traffic"] = dk[0]["fetch_bytes"] * corr + dk[0]["write_bytes"]
            out["traffic_frac"] = out["traffic"] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if kq:
            out["pmc"]["kernel"] = max(cands)[1]
            # FETCH_SIZE counts a coalesced 16-B-per-lane stream at half its bytes on gfx950 (MI355X_MICROARCH.md) and random
            # 64-B lines exactly (tools/gather_ceiling): k_query's traffic is divergent line accesses -> no correction
            out["query_kernel"]["traffic"] = kq["fetch_bytes"] + kq["write_bytes"]
            out["query_kernel"]["traffic_frac"] = out["query_kernel"]["traffic"] / (kq_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["pmc"]["k_query_hbm_lines_per_query"] = kq["fetch_bytes"] / 64.0 / nq
            if kq.get("l2_read_requests"):
                out["pmc"]["k_query_l1_to_l2_requests_per_query"] = kq["l2_read_requests"] / nq
            if kq.get("valu_insts"):
                # VALU issue: wave-instructions x 4 cycles (measured integer rate, profiles/r01_valu_rate.jsonl: 3.1-4.65e13
                # lane-ops/s chip-wide) over the SIMD cycles of the kernel
                lane_ops = kq["valu_insts"] * 64.0
                out["valu"] = {"wave_insts_per_launch": kq["valu_insts"], "insts_per_query": kq["valu_insts"] * 64.0 / nq,
                               "lane_ops_per_s": lane_ops / (kq_ms * 1e-3),
                               "frac_of_measured_valu_rate": [lane_ops / (kq_ms * 1e-3) / 4.65e13, lane_ops / (kq_ms * 1e-3) / 3.1e13]}
            ceil = pmc.get("gather_ceiling_lines_per_s")
            if ceil:
                out["frac_of_gather_ceiling"] = kq["fetch_bytes"] / 64.0 / (kq_ms * 1e-3) / ceil
        tot = 0.0
        for name, v in kern.items():  # the partition passes read wide coalesced streams: FETCH_SIZE counts those at half their bytes
            if not name.startswith(("k_bin_onepass", "k_query_slots", "k_gather_runs", "k_plan_")):
                continue  # kernels of the first (exact) batch are not part of a steady-state step
            tot += v["fetch_bytes"] * (2.0 if name.startswith(("k_bin_hist", "k_bin_scatter", "k_bin_onepass")) else 1.0) + v["write_bytes"]
        out["pipeline"]["traffic"] = tot
    return out


def reads_leg(torch, B, ix, dev, ascii_, total, k, args, orc):
    """the sequence / read front end on 150-nt reads: half cut from the indexed text, half foreign"""
    import numpy as np
    stream = torch.cuda.current_stream().cuda_stream
    chars, roffs, nreads, nk = synth_reads(torch, dev, ascii_, total, k, args.reads, args.read_len, 0.5, args.seed * 7 + 1)
    ids = torch.empty(nk, device=dev, dtype=torch.int64)
    err = torch.zeros(16, device=dev, dtype=torch.int32)

    def dev_call():
        ix.query_sequences_device(chars.data_ptr(), roffs.data_ptr(), nreads, int(chars.numel()), nk, ids.data_ptr(), err.data_ptr(), stream)

    for _ in range(2):
        dev_call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        dev_call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out = {"workload": "%d reads x %d nt (50 %% windows of the indexed text, 50 %% random), %d k-mers per pass" % (nreads, args.read_len, nk),
           "device_resident": {"kmers_per_s": nk / (ms * 1e-3), "ms": ms, "stages_ms": {a: round(v, 4) for a, v in stage_profile(ix, dev_call).items()}},
           "found_fraction": float((ids >= 0).float().mean().item()), "invalid_char_flag": int(err[0].item())}
    ids_dev = ids.cpu().numpy()
    if orc is not None:  # a sample of the reads against the CPU oracle, element-wise
        ns = 3000
        raw = chars[: ns * args.read_len].cpu().numpy().tobytes()
        want = np.concatenate([orc.query_sequence(raw[i * args.read_len:(i + 1) * args.read_len]) for i in range(ns)])
        out["oracle_sample_equal"] = bool(np.array_equal(want, ids_dev[: want.size]))
    if args.no_host_legs:
        return out
    # ---- host buffers through blmx_query_sequences (PCIe + host copies included)
    h_chars = chars.cpu().numpy()
    h_offs = roffs.cpu().numpy().astype(np.uint64)
    h_ids = np.empty(nk, dtype=np.int64)
    u64p, i64p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int64)
    best = None
    for _ in range(3):
        t = time.perf_counter()
        rc = ix._lib.blmx_query_sequences(ix._h, h_chars.ctypes.data_as(ctypes.c_char_p), h_offs.ctypes.data_as(u64p), nreads, h_ids.ctypes.data_as(i64p))
        dt = time.perf_counter() - t
        assert rc == 0
        best = dt if best is None else min(best, dt)
    out["host_buffers"] = {"entry": "blmx_query_sequences", "kmers_per_s": nk / best, "seconds": best, "ids_equal_to_device": bool(np.array_equal(h_ids, ids_dev))}
    # ---- file_query on a FASTA file of the same reads
    tmpdir = tempfile.mkdtemp(prefix="blight_bench_", dir=os.environ.get("TMPDIR") or None)
    path = os.path.join(tmpdir, "reads.fa")
    rec = np.empty((nreads, 11 + args.read_len + 1), dtype=np.uint8)  # ">%09d\n<read>\n"
    digits = (np.arange(nreads, dtype=np.int64)[:, None] // (10 ** np.arange(8, -1, -1, dtype=np.int64))[None, :]) % 10
    rec[:, 0] = ord(">")
    rec[:, 1:10] = digits.astype(np.uint8) + ord("0")
    rec[:, 10] = ord("\n")
    rec[:, 11:11 + args.read_len] = h_chars.reshape(nreads, args.read_len)
    rec[:, -1] = ord("\n")
    rec.tofile(path)
    fsize = os.path.getsize(path)
    good = ctypes.c_uint64()
    bad = ctypes.c_uint64()
    best = None
    for _ in range(3):
        t = time.perf_counter()
        rc = ix._lib.blmx_query_file(ix._h, path.encode(), ctypes.byref(good), ctypes.byref(bad))
        dt = time.perf_counter() - t
        assert rc == 0
        best = dt if best is None else min(best, dt)
    out["file_query"] = {"entry": "blmx_query_file", "file_bytes": fsize, "kmers_per_s": nk / best, "seconds": best,
                         "counts_equal_to_device": (good.value, bad.value) == (int((ids_dev >= 0).sum()), int((ids_dev < 0).sum()))}
    os.unlink(path)
    os.rmdir(tmpdir)
    # ---- one query_sequence_hash call per read from OpenMP threads through the C++ facade (the reference's calling pattern)
    libpath = os.path.join(ROOT, "tools", "libbench_calls.so" if k <= 32 else "libbench_calls128.so")  # kmer_t = uint64_t / __uint128_t
    if os.path.exists(libpath):
        lib = ctypes.CDLL(libpath)
        lib.blight_bench_read_calls.argtypes = [ctypes.c_void_p, ctypes.c_char_p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
        ncall = min(nreads, 200_000)
        nk_call = ncall * max(0, args.read_len - k + 1)
        calls = {}
        for threads in sorted({1, min(8, host_cores())}):
            secs = ctypes.c_double()
            r = [ctypes.c_uint64() for _ in range(4)]
            rc = lib.blight_bench_read_calls(ix._h, h_chars.ctypes.data_as(ctypes.c_char_p), h_offs.ctypes.data_as(u64p), ncall, threads, 1,
                                             ctypes.byref(secs), *[ctypes.byref(x) for x in r])
            assert rc == 0
            calls["threads_%d" % threads] = {"calls_per_s": ncall / secs.value, "kmers_per_s": r[0].value / secs.value,
                                             "counts_equal_to_device": (r[1].value, r[2].value) == (int((ids_dev[:nk_call] >= 0).sum()), int((ids_dev[:nk_call] < 0).sum()))}
        out["per_read_calls"] = {"entry": "kmer_Set_Light::query_sequence_hash(std::string) via include/blight.h, OpenMP callers", "reads": ncall, **calls}
    return out


def scalar_calls_leg(ix, q, ids, kw, k):
    """kmer_Set_Light::query_kmer_hash(kmer_t): ONE k-mer per call (reference blight.cpp:545-550) from 1 and 8 OpenMP threads through the
    C++ facade; concurrent calls are answered together by one launch (the combiner of blmx_front.hip)"""
    import numpy as np
    libpath = os.path.join(ROOT, "tools", "libbench_calls.so" if k <= 32 else "libbench_calls128.so")
    if not os.path.exists(libpath):
        return None
    lib = ctypes.CDLL(libpath)
    u64p, i64p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int64)
    lib.blight_bench_kmer_calls.argtypes = [ctypes.c_void_p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64p]
    n = 100_000
    h_q = np.ascontiguousarray(q[:n].cpu().numpy().view(np.uint64).reshape(-1))
    want = ids[:n].cpu().numpy()
    out = {"entry": "kmer_Set_Light::query_kmer_hash(kmer_t) via include/blight.h, one k-mer per call, OpenMP callers", "kmers": n}
    for threads in sorted({1, min(8, host_cores())}):
        got = np.empty(n, dtype=np.int64)
        secs, found, cs = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        rc = lib.blight_bench_kmer_calls(ix._h, h_q.ctypes.data_as(u64p), n, threads, ctypes.byref(secs), ctypes.byref(found), ctypes.byref(cs), got.ctypes.data_as(i64p))
        assert rc == 0
        out["threads_%d" % threads] = {"queries_per_s": n / secs.value, "ids_equal_to_device": bool(np.array_equal(got, want))}
    return out


def write_fasta(torch, dev, ascii_, offs, nu, path):
    """the first `nu` unitigs as 2-line FASTA records (">\n<sequence>\n": the reference's reader skips the header line, blight.cpp:212-229),
    assembled on the device chunk by chunk and written to `path`; returns the file size"""
    import numpy as np
    nl, gt = 10, 62
    size = 0
    step = 1 << 21  # unitigs per chunk (~230 MB of text)
    with open(path, "wb") as f:
        for u0 in range(0, nu, step):
            u1 = min(nu, u0 + step)
            o = offs[u0:u1 + 1]
            base = int(o[0].item())
            L = int(o[-1].item()) - base
            out = torch.empty(L + 3 * (u1 - u0), device=dev, dtype=torch.uint8)
            rel = (o - base)                                         # start of every unitig inside the chunk's text
            idx = torch.arange(u1 - u0, device=dev, dtype=torch.int64)
            d0 = rel[:-1] + 3 * idx                                  # where the record of unitig i starts in the output
            j = torch.arange(L, device=dev, dtype=torch.int64)
            u = torch.searchsorted(rel, j, right=True) - 1
            out[j + 3 * u + 2] = ascii_[base:base + L]
            out[d0] = gt
            out[d0 + 1] = nl
            out[rel[1:] + 3 * idx + 2] = nl
            del j, u
            a = out.cpu().numpy()
            a.tofile(f)
            size += a.size
            del out
    return size


def fasta_leg(torch, B, dev, ascii_, offs, k, m, n, b, kw, args):
    """construct_index(file) (reference blight.cpp:108-125 is the reference's ONLY construction entry): plain files of the
    C3-size prefix of the input and of the whole input on tmpfs, plus the prefix gzip-compressed; the finished index must equal the one
    built from the same unitigs resident in HBM (counters, per-group tables, IDs of a sample of its k-mers)."""
    import subprocess
    import numpy as np
    nu_all = offs.numel() - 1
    nu_c3 = min(nu_all, WORKLOADS["C3"][4] // 80)
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    tmpdir = tempfile.mkdtemp(prefix="blight_fa_", dir=tmp)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"where": tmp}
    cases = [("c3_size_plain", nu_c3, False), ("c3_size_gzip", nu_c3, True)]
    if nu_all > nu_c3:
        cases.append(("whole_input_plain", nu_all, False))
    refs = {}
    try:
        for name, nu, gz in cases:
            path = os.path.join(tmpdir, "u%d.fa" % nu)
            if not os.path.exists(path):
                write_fasta(torch, dev, ascii_, offs, nu, path)
            fpath = path
            if gz:
                subprocess.check_call(["gzip", "-1", "-k", "-f", path])
                fpath = path + ".gz"
            if nu not in refs:  # the same unitigs built from HBM
                r = B.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=dev.index or 0)
                t0 = time.perf_counter()
                r.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), nu, int(offs[nu].item()))
                t_dev = time.perf_counter() - t0
                r.set_tuning("absent_filter", 1)
                total_n = int(offs[nu].item())
                sq, _ = synth_queries(torch, r, dev, ascii_, offs[:nu + 1], total_n, k, 1_000_000, 99, kw, 1.0)
                sid = torch.empty(sq.shape[0], device=dev, dtype=torch.int64)
                r.query_kmers_device(sq.data_ptr(), sq.shape[0], sid.data_ptr(), stream)
                torch.cuda.synchronize()
                refs[nu] = (r.stats(), {a: r.export(a) for a in ("g_nkeys", "g_prefix", "g_posstart", "bucket_nt")}, sq, sid, t_dev)
                r.close()
            rst, rarr, sq, sid, t_dev = refs[nu]
            each = []
            for rep in range(1 if gz else 3):  # a fresh index per run; plain files: the median of three
                if rep:
                    f.close()
                f = B.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=dev.index or 0)
                t0 = time.perf_counter()
                f.construct_index(fpath)
                each.append(time.perf_counter() - t0)
            secs = sorted(each)[len(each) // 2]
            f.set_tuning("absent_filter", 1)
            fst = f.stats()
            got = torch.empty_like(sid)
            f.query_kmers_device(sq.data_ptr(), sq.shape[0], got.data_ptr(), stream)
            torch.cuda.synchronize()
            same = all(fst[x] == rst[x] for x in ("number_kmer", "number_super_kmer", "total_nt", "largest_mphf", "positions_total_size", "fallback_keys")) and \
                all(np.array_equal(f.export(a), rarr[a]) for a in rarr) and bool(torch.equal(got, sid)) and bool((sid >= 0).all().item())
            fbytes = os.path.getsize(fpath)
            out[name] = {"file_bytes": fbytes, "unitigs": nu, "kmers": fst["number_kmer"], "seconds": secs, "seconds_each": each, "file_GB_per_s": fbytes / secs / 1e9,
                         "text_GB_per_s": os.path.getsize(path) / secs / 1e9, "kmers_per_s": fst["number_kmer"] / secs,
                         "same_unitigs_from_hbm_seconds": t_dev, "equal_to_build_from_hbm": bool(same)}
            f.close()
            if gz:
                # the denominator of that figure: zlib's inflate alone on one host thread over the same file (the library reads it with
                # gzread; one DEFLATE stream cannot be inflated by two threads), nothing parsed, nothing copied to the card
                import zlib
                t0 = time.perf_counter()
                dobj, nout = zlib.decompressobj(31), 0
                with open(fpath, "rb") as fh:
                    while True:
                        chunk = fh.read(1 << 20)
                        if not chunk:
                            break
                        nout += len(dobj.decompress(chunk))
                t_z = time.perf_counter() - t0
                out[name]["zlib_inflate_alone"] = {"seconds": t_z, "text_GB_per_s": nout / t_z / 1e9, "threads": 1,
                                                   "share_of_the_build": t_z / secs}
                os.unlink(fpath)
    finally:
        for fn in os.listdir(tmpdir):
            os.unlink(os.path.join(tmpdir, fn))
        os.rmdir(tmpdir)
    return out


def host_kmer_legs(torch, ix, q, ids, kw):
    """blmx_query_kmers / blmx_query_membership on host arrays: PCIe and the host copies included"""
    import numpy as np
    nq = min(q.shape[0], 100_000_000)
    h_q = q[:nq].cpu().numpy().view(np.uint64).reshape(-1)
    want = ids[:nq].cpu().numpy()
    out = {"queries": nq, "note": "caller-owned host buffers as in the C ABI; best of 4 calls into the same result buffer (the first call also "
                                  "pays for the fresh pages of that buffer: first_call_seconds)"}
    for name, call, dtype in (("blmx_query_kmers", ix.query_kmer_hash, np.int64), ("blmx_query_membership", ix.query_kmer_bool, np.uint8)):
        res = np.empty(nq, dtype=dtype)
        best = first = None
        for _ in range(4):
            t = time.perf_counter()
            got = call(h_q, out=res)
            dt = time.perf_counter() - t
            first = dt if first is None else first
            best = dt if best is None else min(best, dt)
        same = bool(np.array_equal(got, want)) if dtype is np.int64 else bool(np.array_equal(got, want >= 0))
        out[name] = {"queries_per_s": nq / best, "seconds": best, "first_call_seconds": first, "equal_to_device_path": same}
    return out


def main():
    args = parse_args()
    # the bench builds human-scale indexes in a row (query index, file legs, replicas): the library parks freed device blocks up to a
    # cap per device (default 1/16 of the card); raised here, and said so in the line (config.scratch_cache_gb)
    os.environ.setdefault("BLIGHT_SCRATCH_CACHE_GB", "48")
    # stdout carries exactly ONE line, the result: everything else that writes to file descriptor 1 (RCCL prints its version banner
    # there when a communicator is created) is sent to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch  # first: one HIP runtime for torch and the library
    import torch.distributed as dist
    import blight_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Blight hot path has no CPU fallback")
    # rehearsal on a ONE-GPU box: all ranks share cuda:0 and the collectives run over gloo through host memory
    rehearsal = os.environ.get("BLIGHT_DIST_HOST_STAGING") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.rehearse_sharded  # a process group exists and the sharded code path runs
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL has no timeout: a collective that a peer never joins would leave this run hanging until somebody kills it, and the
        # native exchange has never met a second GPU.  A run of N ranks takes one to two minutes; after
        # BLIGHT_BENCH_WATCHDOG_S (default 900) seconds every rank gives up by itself, and rank 0 leaves a line that says so.
        import threading
        limit_s = float(os.environ.get("BLIGHT_BENCH_WATCHDOG_S", "900"))

        def give_up():
            log("rank %d: no result after %.0f s: a collective is stuck (RCCL has no timeout); giving up" % (rank, limit_s))
            if rank == 0:
                os.write(result_fd, (json.dumps({"metric": "kmer_id_queries_per_sec", "value": None, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
                                                 "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
                                                 "data": "synthetic",
                                                 "error": "watchdog: the run did not finish within %.0f s (a stuck collective); no number was measured" % limit_s}) + "\n").encode())
            os._exit(4)
        if limit_s > 0:
            wd = threading.Timer(limit_s, give_up)
            wd.daemon = True
            wd.start()
        if world == 1:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        elif rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    effective_mode = args.mode if multi else "single"
    k, m, n, b, n_kmers, nq = WORKLOADS[args.workload]
    n_kmers = args.kmers or n_kmers
    nq = args.queries or nq
    kw = 1 if k <= 32 else 2
    stream = torch.cuda.current_stream().cuda_stream

    # ---- index: seeded synthetic unitigs generated in HBM, built by the HIP pipeline
    t0 = time.time()
    if args.skew == "dedup":
        ascii_, offs, total = synth_unitigs_dedup(torch, dev, blight_amd, n_kmers, k, args.seed)
    else:
        ascii_, offs, total = synth_unitigs(torch, dev, n_kmers, k, args.seed, args.skew)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    ref_sample = None
    if not multi and args.cpu_seconds > 0 and kw == 1:
        # host copy of the first unitigs (~1.5e7 k-mers) for the reference calibration leg
        nu = min(offs.numel() - 1, 190_000)
        so = offs[: nu + 1].cpu().numpy()
        ref_sample = (ascii_[: int(so[-1])].cpu().numpy(), so)
    ix = blight_amd.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=local_rank)
    if args.light:  # the reference's footprint: its arrays and nothing else (no absent-key filter, 16-B MPHF lines, its position fields)
        ix.set_tuning("filter_bits", 0)
        ix.set_tuning("fat_lines", 0)
        ix.set_tuning("pos_hint", 0)
    if effective_mode == "sharded":
        ix.set_shard(rank, world)  # this rank keeps the MPHF groups it owns; IDs stay global
    t0 = time.time()
    ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
    t_build = time.time() - t0
    for tv in args.tune:
        ix.set_tuning(tv.split("=")[0], int(tv.split("=")[1]))
    ix.set_tuning("absent_filter", 1)  # the filter's passes run in the background after the build: the timed steps start with it armed
    t_build_armed = time.time() - t0   # build + the filter's background passes: the state the timed queries run on
    t_filter_bg = t_build_armed - t_build
    st = ix.stats()
    t_build_again = None
    if not multi:
        # the same build once more in this process: its temporaries and arrays now come from the blocks the first one parked (hipMalloc
        # of VRAM the process has used before costs 25-100 ms per GB on this driver: the first build of a process that generated its
        # input with torch pays that for ~40 GB)
        ix_again = blight_amd.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=local_rank)
        if args.light:
            ix_again.set_tuning("filter_bits", 0)
            ix_again.set_tuning("fat_lines", 0)
            ix_again.set_tuning("pos_hint", 0)
        t0 = time.time()
        ix_again.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
        t_build_again = time.time() - t0
        ix_again.close()
    if rank == 0:
        log("workload %s%s: %d k-mers, %d super-k-mers, %d unitigs, %.2f GB index, gen %.1f s, build %.2f s (%.1f M k-mers/s)" % (
            args.workload, (" (skewed, %s)" % args.skew) if args.skew else "", st["number_kmer"], st["number_super_kmer"], offs.numel() - 1, st["device_bytes"] / 1e9,
            t_gen, t_build, st["number_kmer"] / t_build / 1e6))

    # ---- queries (per rank, resident in HBM)
    # several independently seeded batches rotate through the warm-up and the timed loop: the one-pass partition sizes its bin regions
    # after the PREVIOUS batch and the filter gate follows the previous batch's answers, so neither is ever applied to the batch it was
    # learnt from
    nbatch = max(1, args.batches)
    batches = [synth_queries(torch, ix, dev, ascii_, offs, total, k, nq, args.seed * 11 + rank + 1000 * j, kw, args.present) for j in range(nbatch)]
    if args.exp_sort >= 0 and kw == 1:
        # (experiment, profiles/r04_exp_sorted_batch_c4.txt: what would a partition that also orders a bin's k-mers by the place of their
        # level-0 line buy?  The order is made here, outside the timed region: the pipeline keeps it at tile granularity.)
        M = (1 << 64) - 1
        seed = 0xAAAAAAAA55555555

        def s64(v):
            v &= M
            return v - (1 << 64) if v >> 63 else v

        def shr(x, sft):
            return (x >> sft) & ((1 << (64 - sft)) - 1)

        def hash_bis0(key):  # SingleHashFunctor::hash_bis with the level-0 seed (bbhash.h:172-185), int64 wrap-around arithmetic
            h = (key * s64(seed >> 3)) ^ s64(seed ^ ((seed << 7) & M))
            h = h ^ ~(s64((seed << 11) & M) + (key ^ s64(seed >> 5)))
            h = (~h) + (h << 21)
            h = h ^ shr(h, 24)
            h = (h + (h << 3)) + (h << 8)
            h = h ^ shr(h, 14)
            h = (h + (h << 2)) + (h << 4)
            h = h ^ shr(h, 28)
            return h + (h << 31)
        sorted_batches = []
        for (qq, pp) in batches:
            grp = torch.empty(nq, device=dev, dtype=torch.int32)
            ix.kmer_groups_device(qq.data_ptr(), nq, grp.data_ptr(), stream)
            torch.cuda.synchronize()
            key = grp.to(torch.int64) << args.exp_sort
            if args.exp_sort:
                key |= shr(hash_bis0(qq[:, 0]), 64 - args.exp_sort)
            order = torch.argsort(key)
            sorted_batches.append((qq[:, 0].contiguous()[order].reshape(-1, 1).contiguous(), pp[order]))
            del grp, key, order
        batches = sorted_batches
    q, is_pos = batches[0]
    ids_all = [torch.empty(nq, device=dev, dtype=torch.int64) for _ in range(nbatch)]
    ids = ids_all[0]
    torch.cuda.synchronize()

    querier = None   # torch.distributed exchange (blight_amd/dist.py): fallback of the native one
    comm = None      # native exchange: RCCL behind the C ABI (blmx_comm_*, blmx_query_kmers_sharded_device)
    if effective_mode == "sharded":
        def all_failed(flag):
            t = torch.tensor([1 if flag else 0], device=("cpu" if rehearsal else dev), dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return bool(t.item())
        # 1. the library's own RCCL communicator (rank 0's unique id travels through the launcher's process group)
        failed = rehearsal  # several ranks on ONE GPU: RCCL refuses duplicate devices
        if not failed:
            try:
                uid = torch.zeros(blight_amd.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(blight_amd.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid, 0)
                comm = blight_amd.Comm(bytes(uid.cpu().numpy().tobytes()), rank, world, device=local_rank)
                probe = torch.empty(4096, device=dev, dtype=torch.int64)
                comm.query_kmers_sharded_device(ix, q[:4096].contiguous().data_ptr(), 4096, probe.data_ptr(), stream)
                torch.cuda.synchronize()
            except Exception as e:  # pragma: no cover - only on a broken collective
                log("rank %d: native sharded exchange failed (%r)" % (rank, e))
                failed = True
        if all_failed(failed):
            comm = None
            # 2. the same choreography over torch.distributed
            from blight_amd.dist import DeviceOps, ShardedQuerier
            bounds = torch.from_numpy(ix.shard_bounds().astype("int32")).to(dev)
            querier = ShardedQuerier(DeviceOps(ix, dev), bounds)
            failed = False
            try:
                querier.query(q[:4096].contiguous())
                torch.cuda.synchronize()
            except Exception as e:  # pragma: no cover
                log("rank %d: torch.distributed sharded exchange failed (%r); falling back to replicas" % (rank, e))
                failed = True
            if all_failed(failed):  # 3. full replicas, no data-path collective
                effective_mode = "replicas (sharded exchange failed)"
                querier = None
                ix.close()
                ix = blight_amd.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=local_rank)
                t0 = time.time()
                ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
                t_build = time.time() - t0
                st = ix.stats()
            else:
                effective_mode = "sharded (torch.distributed exchange)"
        else:
            effective_mode = "sharded (native RCCL exchange behind the C ABI)"
    sharded = comm is not None or querier is not None
    slice_leg = comm is not None and not args.no_slice_build
    keep_input = (sharded and not args.no_replicas_leg) or slice_leg or not multi  # single GPU: the excursion, reads and FASTA legs draw from it
    if not keep_input:
        del ascii_

    if comm is not None:
        def step(j=0):  # partition -> all-to-all-v (RCCL/xGMI) -> owner's query kernel -> all-to-all-v back -> gather
            comm.query_kmers_sharded_device(ix, batches[j][0].data_ptr(), nq, ids_all[j].data_ptr(), stream)
    elif querier is not None:
        def step(j=0):
            ids_all[j] = querier.query(batches[j][0])
    else:
        def step(j=0):
            ix.query_kmers_device(batches[j][0].data_ptr(), nq, ids_all[j].data_ptr(), stream)

    for i in range(args.warmup):
        step(i % nbatch)
    torch.cuda.synchronize()
    ix.partition_stats(reset=True)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step((args.warmup + i) % nbatch)  # a batch other than the one before it
        evs[i + 1].record()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = evs[0].elapsed_time(evs[-1]) / args.steps
    step_ms_list = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    pstats = ix.partition_stats()
    for j in range(nbatch):  # every batch answered at least once (their IDs are checked below), outside the timed region
        step(j)
    torch.cuda.synchronize()
    ids = ids_all[0]
    # ---- excursion (single GPU): a batch with no indexed k-mer, then one with only indexed k-mers, each followed by a headline batch
    # that is timed alone: the filter gate and the learnt plan were set by a batch of a different kind
    excursion = None
    if not multi and not args.no_excursion:
        excursion = {}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        scratch_ids = torch.empty(nq, device=dev, dtype=torch.int64)
        for name, frac in (("after_a_batch_without_indexed_kmers", 0.0), ("after_a_batch_of_indexed_kmers_only", 1.0)):
            qx, _ = synth_queries(torch, ix, dev, ascii_, offs, total, k, nq, args.seed * 13 + int(frac) + 7, kw, frac)
            ms_x = []
            for _ in range(3):  # the first one meets the state the headline batches left; the third is that workload's steady state
                e0.record()
                ix.query_kmers_device(qx.data_ptr(), nq, scratch_ids.data_ptr(), stream)
                e1.record()
                torch.cuda.synchronize()
                ms_x.append(e0.elapsed_time(e1))
            # per-kernel times of that workload's steady state (HIP events at the stage boundaries; two more batches of the same kind)
            stages_x = {a: round(v, 4) for a, v in stage_profile(ix, lambda: ix.query_kmers_device(qx.data_ptr(), nq, scratch_ids.data_ptr(), stream), reps=2).items()}
            ms_h = []
            for j in range(2):
                e0.record()
                step((1 + j) % nbatch)
                e1.record()
                torch.cuda.synchronize()
                ms_h.append(e0.elapsed_time(e1))
            excursion[name] = {"those_batches_ms": [round(x, 4) for x in ms_x], "steady_queries_per_s": nq / (ms_x[-1] * 1e-3),
                               "next_headline_batches_ms": [round(x, 4) for x in ms_h], "stages_ms": stages_x}
            del qx
        step(0)  # back to steady state (and ids_all[0] holds batch 0 again)
        torch.cuda.synchronize()
        del scratch_ids
    if multi:
        tt = torch.tensor([dt], device=("cpu" if rehearsal else dev), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- N > 1, sharded: also report the communication-free alternative (full index on every GPU, queries split), measured the
    # same way right after the sharded run ("report both").  `value` stays the sharded number.
    # ---- N > 1, native exchange: the same shard built from THIS RANK'S SLICE of the unitigs only (consecutive pieces in rank order;
    # super-k-mers travel to the owner of their MPHF group: blmx_build_sharded_from_device_sequences) must answer identically.
    slice_build = None
    if slice_leg:
        cpu_or_dev = "cpu" if rehearsal else dev
        nu = offs.numel() - 1
        u0, u1 = nu * rank // world, nu * (rank + 1) // world
        so = offs[u0:u1 + 1]
        c0, c1 = int(so[0].item()), int(so[-1].item())
        so = (so - so[0]).contiguous()
        chars = ascii_[c0:c1]
        ix2 = blight_amd.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=local_rank)
        ix2.set_shard(rank, world)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        comm.build_sharded_from_device(ix2, chars.data_ptr(), so.data_ptr(), u1 - u0, c1 - c0)
        torch.cuda.synchronize()
        ts = torch.tensor([time.perf_counter() - t0], device=cpu_or_dev, dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        ids2 = torch.empty(nq, device=dev, dtype=torch.int64)
        comm.query_kmers_sharded_device(ix2, q.data_ptr(), nq, ids2.data_ptr(), stream)
        torch.cuda.synchronize()
        st2 = ix2.stats()
        same = torch.tensor([1 if torch.equal(ids2, ids) and st2["local_number_kmer"] == st["local_number_kmer"] else 0], device=cpu_or_dev,
                            dtype=torch.int32)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        slice_build = {"seconds": float(ts.item()), "whole_input_build_seconds_rank0": t_build, "unitigs_of_rank0": u1 - u0,
                       "ids_equal_to_whole_input_build": bool(same.item()),
                       "build_scratch_peak_bytes_rank0": int(st2["build_scratch_peak_bytes"]),
                       "whole_input_build_scratch_peak_bytes_rank0": int(st["build_scratch_peak_bytes"]),
                       "what": "every rank chops 1/%d of the unitigs; all-reduce of the group histograms, all-to-all-v of the super-k-mers" % world}
        del ids2, chars, so
        ix2.close()

    replicas = None
    if sharded and keep_input and not args.no_replicas_leg:
        ids_sharded = ids
        full = blight_amd.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=local_rank)
        full.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
        del ascii_
        ids_r = torch.empty(nq, device=dev, dtype=torch.int64)
        for _ in range(args.warmup):
            full.query_kmers_device(q.data_ptr(), nq, ids_r.data_ptr(), stream)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full.query_kmers_device(q.data_ptr(), nq, ids_r.data_ptr(), stream)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        tr = torch.tensor([time.perf_counter() - t0], device=("cpu" if rehearsal else dev), dtype=torch.float64)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        same = torch.tensor([1 if torch.equal(ids_r, ids_sharded) else 0], device=("cpu" if rehearsal else dev), dtype=torch.int32)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        replicas = {"value": world * nq * args.steps / float(tr.item()), "unit": "queries/s", "ms_per_step": float(tr.item()) / args.steps * 1e3,
                    "parallelism": "replicas x%d (full index per GPU, no data-path collective)" % world,
                    "ids_equal_to_sharded": bool(same.item())}
        full.close()

    # ---- sanity, every batch: every indexed query found, no random query found, IDs in range
    n_pos_found = n_neg_found = n_pos = 0
    ok = True
    for j in range(nbatch):
        found = ids_all[j] >= 0
        pj = batches[j][1]
        n_pos_found += int((found & pj).sum().item())
        n_neg_found += int((found & ~pj).sum().item())
        n_pos += int(pj.sum().item())
        ok = ok and int(ids_all[j].max().item()) < st["number_kmer"]
    ok = ok and n_pos_found == n_pos
    if slice_build is not None:
        ok = ok and slice_build["ids_equal_to_whole_input_build"]
    if multi:
        okt = torch.tensor([1 if ok else 0], device=("cpu" if rehearsal else dev), dtype=torch.int32)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    if rank == 0:
        value = world * nq * args.steps / dt
        stages = stage_profile(ix, step) if not sharded else {}
        out = {
            "metric": "kmer_id_queries_per_sec", "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64" if kw == 1 else "u128", "data": "synthetic",
            "config": {"workload": "%s%s: %d indexed %d-mers (seeded synthetic unitigs, k%d m%d n%d b%d), %d queries/step/GPU, "
                                   "%g%% indexed / %g%% random" % (args.workload, (" skewed (a tenth of the k-mers in unitigs over {A,C}%s)" % (", repeated k-mers removed" if args.skew == "dedup" else "")) if args.skew else "",
                                                                    st["number_kmer"], k, k, m, n, b, nq, 100 * args.present, 100 * (1 - args.present)),
                       "parallelism": "1 GPU" if not multi else ("%s x%d" % (effective_mode, world)),
                       "index_bytes_per_gpu": st["device_bytes"], "kmers_per_gpu": st["local_number_kmer"],
                       "footprint": footprint(st), "light": bool(args.light),
                       "scratch_cache_gb": float(os.environ.get("BLIGHT_SCRATCH_CACHE_GB", "0")),
                       "build_scratch_peak_bytes": st["build_scratch_peak_bytes"],
                       "build_seconds": t_build, "build_seconds_filter_armed": t_build_armed, "build_seconds_again": t_build_again,
                       "build_kmers_per_sec": st["number_kmer"] / t_build_armed,
                       "build_note": "build_seconds_again: the same build a second time in this process, on recycled blocks (the first one also "
                                     "pays the driver for fresh VRAM); build_seconds: until construct_index returned (queries are answered from then on, unfiltered); "
                                     "build_seconds_filter_armed: until the absent-key filter's background passes were done too -- the state the "
                                     "timed queries run on; build_kmers_per_sec is derived from the latter",
                       "query_batches": "%d independently seeded batches rotate through warm-up and timed steps" % nbatch,
                       "absent_filter": "blocked Bloom filter over the keys, %s bits/key; its passes ran %.2f s in the background after the build returned" % (
                           os.environ.get("BLIGHT_FILTER_BITS", "8, or 16 at b >= 7"), t_filter_bg), "largest_mphf": st["largest_mphf"], "largest_bucket": st["largest_bucket"],
                       "mphf_groups_lds_built": st["mphf_groups_lds_built"], "mphf_groups": st["ngroups"], "fallback_keys": st["fallback_keys"]},
            "roofline": roofline_block(args, k, kw, b, nq, stages, kern_ms, st) if not multi else
            {"bound": "hbm", "achieved": ALGO_BYTES.get(k, 145) * nq * world / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
             "frac": ALGO_BYTES.get(k, 145) * nq / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
             "kernel": "whole sharded step (partition, exchange, k_query, exchange, gather) against the aggregate HBM peak"},
            "checks": {"indexed_queries_found": n_pos_found, "indexed_queries": n_pos, "random_queries_found": n_neg_found,
                       "batches_checked": nbatch, "ok": bool(ok)},
            "ms_per_step_each": [round(x, 4) for x in step_ms_list],
            "partition": {"what": "one-pass partition of the timed steps: k-mers that did not fit their bin's region (sized after the previous, "
                                  "different batch) and went through the overflow region", "batches": pstats["batches"], "kmers": pstats["kmers"],
                          "overflow_kmers": pstats["overflow_kmers"], "max_overflow_of_a_batch": pstats["max_overflow_of_a_batch"]},
        }
        if excursion is not None:
            out["excursion"] = excursion
            ai = excursion.get("after_a_batch_of_indexed_kmers_only")
            if ai and "roofline" in out and "query_kernel" in out["roofline"] and ai.get("stages_ms"):
                kq = ai["stages_ms"].get("k_query", 0.0)
                if kq > 0:
                    algo = ALGO_BYTES.get(k, 145)
                    out["roofline"]["all_indexed"] = {
                        "kernel_ms": kq, "step_ms": ai["those_batches_ms"][-1], "achieved": algo * nq / (kq * 1e-3) / 1e9,
                        "frac": algo * nq / (kq * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "what": "the query kernel on a batch of indexed k-mers only (steady state of the excursion; the reference's own benchmark queries "
                                "the indexed file): every query then does the reference's work, so this is the fraction to read as `bytes of the "
                                "reference's algorithm per second`; `frac` above is the same quotient on the 50/50 batch, where the absent-key filter "
                                "ends half the queries after one line, and `traffic_frac` is what the counters say the kernel really moved"}
        if multi:
            out["exchange"] = "native-rccl" if comm is not None else ("torch.distributed" if querier is not None else "replicas")
            if comm is not None:
                # what a slot costs on the wire (the prediction a SCALE record is read against): regions are 6 % + 256 slots
                # per bin larger than their content; (N - 1) / N of every batch travels, over N - 1 links
                ship = os.environ.get("BLIGHT_DIST_SHIP_MINIMIZERS", "0") not in ("", "0")
                out["exchange_wire"] = {"out_bytes_per_slot": 8 * kw + (4 if ship else 0), "back_bytes_per_slot": 4 if st["number_kmer"] < (1 << 32) - 1 else 8,
                                        "minimizers_travel": ship, "region_slack": "6 % + 256 slots per bin",
                                        "fraction_of_a_batch_that_travels": (world - 1) / world if world > 1 else 0.0}
        if args.mode == "sharded" and multi and not sharded:
            out["sharded_fallback"] = effective_mode
        orc = None
        out["cpu_baseline"] = None
        if not multi and args.cpu_seconds > 0:
            try:
                import numpy as np
                orc, t_import = import_oracle(ix, st, k, m, n, b)
                q_host = q.cpu().numpy().view(np.uint64)
                q_host = q_host.reshape(-1) if kw == 1 else q_host.reshape(-1, 2)
                out["cpu_baseline"] = cpu_baseline(orc, t_import, q_host, ids.cpu().numpy(), args.cpu_seconds)
            except Exception as e:  # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
            if kw == 1:
                try:
                    out["cpu_reference"] = cpu_reference_calibration(ref_sample[0], ref_sample[1], k, m, n, b) if ref_sample else None
                except Exception as e:
                    out["cpu_reference"] = {"value": None, "kind": "reference", "sample": "failed: %r" % (e,)}
        if not multi and args.reads > 0:
            try:
                out["reads"] = reads_leg(torch, blight_amd, ix, dev, ascii_, total, k, args, orc)
            except Exception as e:
                out["reads"] = {"failed": repr(e)}
            if not args.no_host_legs:
                try:
                    out["host_buffers"] = host_kmer_legs(torch, ix, q, ids, kw)
                except Exception as e:
                    out["host_buffers"] = {"failed": repr(e)}
                try:
                    out["scalar_calls"] = scalar_calls_leg(ix, q, ids, kw, k)
                except Exception as e:
                    out["scalar_calls"] = {"failed": repr(e)}
        if orc is not None:
            orc.close()
        if not multi and not args.no_fasta_leg and not args.skew:
            try:
                out["build_from_fasta"] = fasta_leg(torch, blight_amd, dev, ascii_, offs, k, m, n, b, kw, args)
            except Exception as e:
                out["build_from_fasta"] = {"failed": repr(e)}
        if replicas is not None:
            out["replicas"] = replicas
        if slice_build is not None:
            out["build_from_slices"] = slice_build
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench sanity check failed")
    if args.mode == "sharded" and multi and comm is None:
        raise SystemExit("bench: --mode sharded was asked for, but the native RCCL exchange did not run (%s; `exchange` in the line above says what "
                         "did): the line above is NOT the number of the native sharded path" % effective_mode)


if __name__ == "__main__":
    main()
