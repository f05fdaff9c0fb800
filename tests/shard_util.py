"""Helpers of the sharding tests: every k-mer readable in a packed sequence, virtual ranks on one GPU, answers through shards."""
import ctypes as C

import numpy as np

from test_dist_gloo import shard_bounds_from_counts


def unpack_nts(words, total_nt):
    """2-bit packed words (first nucleotide most significant) -> uint8 codes"""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    return ((words[:, None] >> np.arange(62, -2, -2, dtype=np.uint64)[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)[:total_nt]


def canonical_kmers_at_all_positions(nts, k, kb):
    """canonical k-mer (min of forward and reverse complement, numeric order, reference blight.cpp:86-91) starting at every position of
    the code array `nts`: (len - k + 1, kb / 8) uint64, low word first"""
    nts = np.asarray(nts, dtype=np.uint64)
    n = len(nts) - k + 1
    if n <= 0:
        return np.zeros((0, kb // 8), np.uint64)
    lo_f = np.zeros(n, np.uint64); hi_f = np.zeros(n, np.uint64); lo_r = np.zeros(n, np.uint64); hi_r = np.zeros(n, np.uint64)
    two = np.uint64(2)
    for i in range(k):
        c = nts[i:i + n]
        fs = 2 * (k - 1 - i)               # forward: nucleotide i at bit 2 (k - 1 - i)
        rs = 2 * i                         # reverse complement: nucleotide i, complemented, at bit 2 i
        if fs >= 64:
            hi_f |= c << np.uint64(fs - 64)
        else:
            lo_f |= c << np.uint64(fs)
        if rs >= 64:
            hi_r |= (c ^ two) << np.uint64(rs - 64)
        else:
            lo_r |= (c ^ two) << np.uint64(rs)
    fwd = (hi_f < hi_r) | ((hi_f == hi_r) & (lo_f <= lo_r))
    lo = np.where(fwd, lo_f, lo_r)
    hi = np.where(fwd, hi_f, hi_r)
    return lo.reshape(-1, 1) if kb == 8 else np.stack([lo, hi], axis=1)


def readable_kmers(index, k, kb, b):
    """every canonical k-mer readable anywhere in the packed sequence of `index` (a single, unsharded index), the zero padding a
    window of the last buckets can reach included: (unique k-mers as (n, kb/8) uint64, nts)"""
    st = index.stats()
    nts = unpack_nts(index.export("seq"), st["total_nt"])
    nts = np.concatenate([nts, np.zeros((1 << b) + k, np.uint8)])
    allk = canonical_kmers_at_all_positions(nts, k, kb)
    return np.unique(allk, axis=0), nts


def hip_runtime():
    hip = C.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return hip


def build_from_slices(B, seqs, cuts, params, halo=True, before_halo=None):
    """`world` virtual ranks on one GPU: rank r feeds seqs[cuts[r]:cuts[r + 1]] through blmx_shard_build_begin / _plan / _finish; the
    all-reduce is a numpy sum, the all-to-all-v a set of device copies, the halo all-gather a concatenation.  before_halo(ixs) runs
    between finish and the halo exchange."""
    import torch
    k, m, n, b = params
    world = len(cuts) - 1
    hip = hip_runtime()
    ixs, hists = [], []
    for r in range(world):
        x = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
        x.set_shard(r, world)
        hists.append(x.shard_build_begin(seqs[cuts[r]:cuts[r + 1]]))
        ixs.append(x)
    gh = np.sum(hists, axis=0, dtype=np.uint64)
    plans = [x.shard_build_plan(gh, world) for x in ixs]
    dev = torch.device("cuda", 0)
    incoming = []
    for d in range(world):
        recv_sk = np.array([plans[r][0][d] for r in range(world)], dtype=np.uint64)
        recv_w = np.array([plans[r][1][d] for r in range(world)], dtype=np.uint64)
        meta = torch.zeros(int(recv_sk.sum()) + 1, dtype=torch.int64, device=dev)
        words = torch.zeros(int(recv_w.sum()) + 1, dtype=torch.int64, device=dev)
        om = ow = 0
        for r in range(world):
            sk, w, pm, pw = plans[r]
            sm, sw = int(sk[:d].sum()), int(w[:d].sum())
            if recv_sk[r]:
                assert hip.hipMemcpy(meta.data_ptr() + 8 * om, pm + 8 * sm, 8 * int(recv_sk[r]), 3) == 0
            if recv_w[r]:
                assert hip.hipMemcpy(words.data_ptr() + 8 * ow, pw + 8 * sw, 8 * int(recv_w[r]), 3) == 0
            om += int(recv_sk[r])
            ow += int(recv_w[r])
        incoming.append((meta, words, recv_sk, recv_w))
    torch.cuda.synchronize()
    for d, (meta, words, recv_sk, recv_w) in enumerate(incoming):     # the outgoing buffers of a rank live until ITS finish
        ixs[d].shard_build_finish(meta.data_ptr(), words.data_ptr(), recv_sk, recv_w)
    if before_halo:
        before_halo(ixs)
    if halo:
        recs = np.concatenate([x.shard_halo_get() for x in ixs])
        for x in ixs:
            x.shard_halo_set(recs)
    return ixs, gh


def answers_through_shards(shards, q):
    """every shard is asked for every k-mer (host path): a shard answers only k-mers whose bucket it owns, so at most one answer per
    k-mer is an ID; returns the combined answers"""
    got = np.stack([ix.query_kmer_hash(q.reshape(-1)) for ix in shards])
    assert ((got >= 0).sum(axis=0) <= 1).all(), "two shards answered the same k-mer"
    return got.max(axis=0)


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def make_cut_case(O, k, kb, m, n, b, nseq, world, seed):
    """sequences, oracle, the k-mers to ask, the oracle's answers, and which of them are false positives of the reference that can
    only be read across a shard cut of a `world`-way split"""
    rng = np.random.default_rng(seed)
    seqs = [_rand_seq(rng, k + int(rng.integers(0, 60))) for _ in range(nseq)]
    seqs += [seqs[1], max(seqs[:40], key=len)[3:], "A" * (k + 9)]
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    A = orc.arrays()
    total_nt = orc.stats["total_nt"]
    nts = np.concatenate([unpack_nts(A["seq"], total_nt), np.zeros((1 << b) + k, np.uint8)])
    allk = canonical_kmers_at_all_positions(nts, k, kb)
    q, inverse = np.unique(allk, axis=0, return_inverse=True)
    want = orc.query_kmers(q.reshape(-1) if kb == 8 else q)
    keys = np.unique(np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs]), axis=0)
    keyset = set(map(tuple, keys.tolist()))
    is_key = np.array([row in keyset for row in map(tuple, q.tolist())])
    bounds = shard_bounds_from_counts(A["g_nkeys"], world)
    gshift = 2 * m - 1 - n
    cuts = sorted({int(A["bucket_start"][int(bounds[r]) << gshift]) for r in range(1, world) if (int(bounds[r]) << gshift) < len(A["bucket_start"])} - {0, total_nt})
    starts = np.arange(len(allk))
    straddles = np.zeros(len(allk), bool)
    for c in cuts:
        straddles |= (starts < c) & (starts + k > c)
    only_across = np.ones(len(q), bool)
    np.logical_and.at(only_across, inverse.reshape(-1), straddles)
    cut_fp = only_across & ~is_key & (want >= 0)
    # ... and, of those, the ones a shard WITHOUT its halo cannot read either: behind its end it sees padding ('A' = code 0), so a
    # k-mer whose nucleotides behind the cut are all 'A' reads the same with and without the halo
    needs_halo = np.zeros(len(allk), bool)
    for c in cuts:
        for st in range(max(0, c - k + 1), c):
            needs_halo[st] = nts[c:st + k].any()
    strict = np.ones(len(q), bool)
    np.logical_and.at(strict, inverse.reshape(-1), needs_halo)
    return seqs, orc, q, want, is_key, cut_fp, cut_fp & strict, bounds


CUT_CASES = [(31, 8, 7, 5, 6, 400, 2, 1), (31, 8, 7, 5, 6, 400, 3, 1), (31, 8, 7, 5, 6, 400, 8, 1), (21, 8, 5, 3, 8, 300, 2, 8), (21, 8, 5, 3, 8, 300, 8, 1),
         (63, 16, 9, 7, 8, 300, 2, 1), (63, 16, 9, 7, 8, 300, 3, 1), (63, 16, 9, 7, 8, 300, 8, 1)]


