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
