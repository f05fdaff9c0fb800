"""World-size-2 `gloo` test of the sharded query choreography (blight_amd/dist.py ShardedQuerier) on CPU: the exchange logic
(counts all-to-all, k-mer all-to-all-v, IDs all-to-all-v back, un-permute) is the code that runs on GPUs; only the per-rank
ops are swapped for numpy + the CPU oracle (a rank answers ONLY the MPHF groups it owns, everything else is -1)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard_bounds_from_counts(nkeys, world):
    """same rule as the library (blmx_build.hip): boundary r = first group whose key prefix >= r*N/world"""
    pre = np.concatenate([[0], np.cumsum(nkeys.astype(np.uint64))]).astype(np.uint64)
    n = int(pre[-1])
    b = [0]
    for r in range(1, world):
        b.append(int(np.searchsorted(pre[:-1], (n * r) // world, side="left")))
    b.append(len(nkeys))
    return np.array(b, dtype=np.uint32)


class OracleOps:
    def __init__(self, orc, O, rank, bounds):
        self.orc, self.O, self.rank, self.bounds = orc, O, rank, bounds
        self.gshift = 2 * orc.m - 1 - orc.n

    def empty_kmers(self, n):
        return torch.empty((n, 1), dtype=torch.int64)

    def empty_ids(self, n):
        return torch.empty(n, dtype=torch.int64)

    def _groups(self, kmers):
        ku = kmers.numpy().view(np.uint64).reshape(-1)
        return np.array([self.O.p_minimizer(int(x), self.orc.k, self.orc.m) >> self.gshift for x in ku], dtype=np.uint32)

    def route(self, kmers, bounds_t, world):
        g = self._groups(kmers)
        dest = np.searchsorted(self.bounds[1:-1], g, side="right")
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=world).astype(np.int64)
        return kmers[torch.from_numpy(order)], torch.from_numpy(order.astype(np.int32)), torch.from_numpy(counts)

    def query(self, kmers):
        if kmers.shape[0] == 0:
            return self.empty_ids(0)
        g = self._groups(kmers)
        owned = (g >= self.bounds[self.rank]) & (g < self.bounds[self.rank + 1])
        ids = self.orc.query_kmers(kmers.numpy().view(np.uint64).reshape(-1).copy())
        ids[~owned] = -1  # a shard holds nothing of foreign groups
        return torch.from_numpy(ids)

    def unpermute(self, vals, order):
        out = torch.empty_like(vals)
        out[order.long()] = vals
        return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from blight_amd.dist import ShardedQuerier
        c = G.Case("lambda_k31_m9_n7_b6")
        orc = O.Oracle(c.k, c.m, c.n, c.b, c.seqs)
        bounds = shard_bounds_from_counts(orc.arrays()["g_nkeys"], world)
        rng = np.random.default_rng(100 + rank)
        pos = orc.canonical_kmers(c.seqs[1 + rank][:3000])
        kmers = np.concatenate([pos, c.neg_kmers[rank * 500:(rank + 1) * 500]])
        kmers = kmers[rng.permutation(len(kmers))]
        want = orc.query_kmers(kmers)
        sq = ShardedQuerier(OracleOps(orc, O, rank, bounds), torch.from_numpy(bounds.astype(np.int32)))
        got = sq.query(torch.from_numpy(kmers.view(np.int64).reshape(-1, 1).copy()))
        ok = bool(np.array_equal(got.numpy(), want)) and int((want >= 0).sum()) > 2000
        # uneven / empty batches
        got0 = sq.query(torch.from_numpy(kmers[: (5 if rank == 0 else 0)].view(np.int64).reshape(-1, 1).copy()))
        ok = ok and np.array_equal(got0.numpy(), want[: (5 if rank == 0 else 0)])
        # the software pipeline with 1 (synchronous path) and 3 pieces gives the same answers
        for chunks in (1, 3):
            gc = sq.query(torch.from_numpy(kmers.view(np.int64).reshape(-1, 1).copy()), chunks=chunks)
            ok = ok and bool(np.array_equal(gc.numpy(), want))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_query_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(0, True), (1, True)]


# ---- construction from slices (blight_amd/dist.py SlicedBuilder) ------------------------------------------------------------------
class OracleSliceOps:
    """CPU stand-in of the three build phases: the oracle's chopper cuts the slice into super-k-mers, they travel as (meta word,
    2-bit packed nucleotides) exactly like on the GPU, and the owner builds an ORACLE over the super-k-mers it received, in the
    order it received them."""

    def __init__(self, O, k, m, n, b, rank):
        self.O, self.k, self.m, self.n, self.b, self.rank = O, k, m, n, b, rank
        self.gshift = 2 * m - 1 - n

    def begin(self, seqs):
        self.sks = []  # (minimizer, nucleotides) in input order
        hist = np.zeros((1 << self.n) + 1, dtype=np.int64)
        for s in seqs:
            for off, nt, mini in self.O.p_chop(s, self.k, self.m):  # (offset, length in nucleotides, minimizer)
                self.sks.append((mini, s[off:off + nt]))
                hist[mini >> self.gshift] += nt - self.k + 1
        hist[-1] = len(self.sks)
        return torch.from_numpy(hist)

    def plan(self, global_hist, world):
        gh = global_hist.numpy()
        self.bounds = shard_bounds_from_counts(gh[:-1], world)
        self.id_base = int(gh[:self.bounds[self.rank]].sum())
        self.global_n = int(gh[:-1].sum())
        owner = np.searchsorted(self.bounds[1:-1], [mi >> self.gshift for mi, _ in self.sks], side="right") if self.sks else np.zeros(0, int)
        send_sk, send_words, meta, words = [], [], [], []
        for r in range(world):
            mine = [self.sks[i] for i in np.flatnonzero(owner == r)]       # stable: input order inside a destination
            nts = "".join(s for _, s in mine)
            nw = (len(nts) + 31) // 32
            codes = np.array([self.O.p_nuc2int(c) for c in nts.encode()] + [0] * (32 * nw - len(nts)), dtype=np.uint64).reshape(nw, 32)
            shifts = np.arange(62, -2, -2, dtype=np.uint64)                    # first nucleotide most significant
            words.append((codes << shifts).sum(axis=1, dtype=np.uint64) if nw else np.zeros(0, np.uint64))
            meta.append(np.array([(mi << 32) | len(s) for mi, s in mine], dtype=np.uint64))
            send_sk.append(len(mine))
            send_words.append(nw)
        cat = lambda parts: torch.from_numpy(np.concatenate(parts).view(np.int64).copy()) if parts else torch.empty(0, dtype=torch.int64)
        return send_sk, send_words, cat(meta), cat(words)

    def empty(self, n):
        return torch.empty(int(n), dtype=torch.int64)

    def finish(self, meta_in, words_in, recv_sk, recv_words):
        meta, words = meta_in.numpy().view(np.uint64), words_in.numpy().view(np.uint64)
        seqs, mo, wo = [], 0, 0
        for src in range(len(recv_sk)):
            w = words[wo:wo + recv_words[src]]
            nts = "".join("ACTG"[int((x >> np.uint64(sh)) & np.uint64(3))] for x in w for sh in range(62, -2, -2))
            pos = 0
            for e in meta[mo:mo + recv_sk[src]]:
                ln = int(e & np.uint64(0xffffffff))
                seqs.append(nts[pos:pos + ln])
                pos += ln
            mo += recv_sk[src]
            wo += recv_words[src]
        self.received = seqs
        self.shard = self.O.Oracle(self.k, self.m, self.n, self.b, seqs) if seqs else None


def _slice_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from blight_amd.dist import SlicedBuilder
        assert "ACTG"[O.p_nuc2int(ord("T"))] == "T" and O.p_nuc2int(ord("G")) == 3  # the reference's code: A0 C1 T2 G3 (kmer.h:56-69)
        k, m, n, b = 31, 9, 7, 6
        rng = np.random.default_rng(77)                                   # the same input on every rank; each takes its slice
        seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, k - 1 + int(rng.geometric(1 / 60)))) for _ in range(400)]
        seqs += ["A" * 70, seqs[3], seqs[3][2:]]                            # duplicates: equal k-mers from two slices in one bucket
        cut = [0, 150, len(seqs)] if world == 2 else [len(seqs) * r // world for r in range(world + 1)]
        ops = OracleSliceOps(O, k, m, n, b, rank)
        SlicedBuilder(ops).build(seqs[cut[rank]:cut[rank + 1]])
        whole = O.Oracle(k, m, n, b, seqs)
        arrays = whole.arrays()
        ok = np.array_equal(ops.bounds, shard_bounds_from_counts(arrays["g_nkeys"], world)) and ops.global_n == int(arrays["g_nkeys"].sum())
        # every k-mer of the WHOLE input: the owner's shard answers with the global ID minus its base, the other shard with -1
        kmers = np.unique(np.concatenate([whole.canonical_kmers(s) for s in seqs]))
        want = whole.query_kmers(kmers)
        groups = np.array([O.p_minimizer(int(x), k, m) >> ops.gshift for x in kmers])
        owned = (groups >= ops.bounds[rank]) & (groups < ops.bounds[rank + 1])
        got = ops.shard.query_kmers(kmers) if ops.shard else np.full(len(kmers), -1, np.int64)
        ok = ok and bool(np.all(want >= 0)) and np.array_equal(got[owned] + ops.id_base, want[owned]) and bool(np.all(got[~owned] == -1))
        ok = ok and int(owned.sum()) > 1000
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sliced_build_world2_gloo():
    """Two ranks, each with its own slice of the sequences: all-reduce of the histograms, all-to-all-v of the super-k-mers, and
    the owner's index over what it received (in source-rank order) reproduces the GLOBAL IDs of the whole-input index."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slice_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(0, True), (1, True)]
