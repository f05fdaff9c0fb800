#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite: minutes of GPU time): random (k, m, n, b, width) and random inputs with
low-complexity and duplicated stretches; every run compares the GPU with the CPU oracle on counters, all exported arrays, IDs
of all k-mers, negatives, near misses, positions, and the partition -> query -> gather pipeline.  Usage: python tests/fuzz_parity.py [runs] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # test infrastructure: may use the oracle
import numpy as np
import blight_amd as B
from oracle import oracle as O

def sweep(runs, seed, verbose=True):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for it in range(runs):
        kb = 8 if rng.random() < 0.6 else 16
        k = int(rng.integers(9, 33)) if kb == 8 else int(rng.integers(9, 64))
        m = int(rng.choice([x for x in (3, 5, 7, 9, 11) if x <= k]))
        n = int(rng.integers(0, 2 * m))          # <= 2m - 1
        b = int(rng.integers(0, 11))
        alpha = "ACGT" if rng.random() < 0.7 else ("AC" if rng.random() < 0.5 else "ACG")
        nseq = int(rng.integers(1, 400))
        seqs = []
        for _ in range(nseq):
            L = k + int(rng.geometric(1 / 50)) - 1
            s = "".join(alpha[i] for i in rng.integers(0, len(alpha), L))
            if rng.random() < 0.1:
                s = s[: max(k, L // 2)] * 2      # tandem duplication
            if rng.random() < 0.05:
                s = "A" * (k + int(rng.integers(0, 30)))
            seqs.append(s)
        if rng.random() < 0.3:
            seqs += seqs[: max(1, nseq // 5)]    # duplicated unitigs
        tag = "run %d: k=%d m=%d n=%d b=%d kb=%d nseq=%d alpha=%s" % (it, k, m, n, b, kb, len(seqs), alpha)
        try:
            ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
            if rng.random() < 0.5:
                ix.set_tuning("build_passes", int(rng.integers(1, 9)))  # passes over ranges of MPHF groups: the arrays must not depend on them
            if rng.random() < 0.5:
                ix.set_tuning("fat_lines", int(rng.integers(0, 2)))     # separate filter array / fat level-0 lines instead of level-0 records
            ix.construct_index_from_sequences(seqs)
            if rng.random() < 0.8:
                ix.set_tuning("absent_filter", 1)  # queries through the armed absent-key filter (else: maybe before its passes are done)
            orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
            st = ix.stats()
            for f, g in (("number_kmer", "number_kmer"), ("number_super_kmer", "number_super_kmer"), ("total_nt", "total_nt"),
                         ("largest_mphf", "largest_mphf"), ("positions_total_size", "positions_bits")):
                assert st[f] == orc.stats[g], (tag, f, st[f], orc.stats[g])
            a = orc.arrays()
            for key in ("bucket_nt", "bucket_start", "seq", "positions", "g_nkeys", "g_prefix", "g_posstart"):
                assert np.array_equal(ix.export(key), a[key]), (tag, key)
            q = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
            neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(2000)]
            neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
            allq = np.ascontiguousarray(np.concatenate([q, neg]))
            want = orc.query_kmers(allq.reshape(-1) if kb == 8 else allq)
            for pol in (-1, 1, 1):  # direct kernel; partitioned pipeline: exact pass, then one pass with the learnt regions
                ix.set_query_binning(pol)
                got = ix.query_kmer_hash(allq.reshape(-1))
                assert np.array_equal(got, want), (tag, "ids", pol, int((got != want).sum()))
            assert (want[: len(q)] >= 0).all(), tag
            # every k-mer readable anywhere in the packed sequence (junctions of super-k-mers and of buckets included): the
            # reference answers some of those although they are not keys, and so must the filter
            words = ix.export("seq")
            nts = ((words[:, None] >> np.arange(62, -2, -2, dtype=np.uint64)[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)[: st["total_nt"]]
            nts = np.concatenate([nts, np.zeros((1 << b) + k, np.uint8)])  # reads past the end are defined as 'A' (windows of the last buckets)
            vals, v, mask = set(), 0, (1 << (2 * k)) - 1
            for p, c in enumerate(nts.tolist()):
                v = ((v << 2) | c) & mask
                if p >= k - 1:
                    vals.add(min(v, O.p_rc(v, k, kb)))
            rq = O.kmers_to_array(sorted(vals), kb)
            got = ix.query_kmer_hash(rq.reshape(-1))
            rwant = orc.query_kmers(rq if kb == 8 else rq.reshape(-1, 2))
            if not np.array_equal(got, rwant):
                bad = np.flatnonzero(got != rwant)
                ix.set_tuning("absent_filter", 0)
                nofilter = ix.query_kmer_hash(rq.reshape(-1))
                ix.set_query_binning(-1)
                direct = ix.query_kmer_hash(rq.reshape(-1))
                raise AssertionError((tag, "readable k-mers", len(bad), [(int(i), [hex(int(x)) for x in rq.reshape(len(got), -1)[i]], int(got[i]), int(rwant[i]),
                                                                         int(nofilter[i]), int(direct[i])) for i in bad[:5]]))
            # the same k-mers through whole-input SHARDS of a random world: the halo makes every shard cut invisible
            if rng.random() < 0.15 and n >= 1:
                import shard_util
                world = int(rng.integers(2, 7))
                shards = []
                for r in range(world):
                    sx = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
                    sx.set_shard(r, world)
                    sx.construct_index_from_sequences(seqs)
                    sx.set_tuning("absent_filter", int(rng.integers(0, 2)))
                    shards.append(sx)
                sgot = shard_util.answers_through_shards(shards, rq.reshape(len(rwant), -1))
                assert np.array_equal(sgot, rwant), (tag, "shards", world, int((sgot != rwant).sum()))
                for sx in shards:
                    sx.close()
            # reads cut from the packed sequence itself (junctions and padding included) through the sequence front end
            step = max(1, len(nts) // 40)
            rd = ["".join("ACTG"[c] for c in nts[a0:a0 + k + int(rng.integers(0, 2 * k))].tolist()) for a0 in range(0, max(1, len(nts) - k), step)]
            assert np.array_equal(ix.query_sequences_hash(rd), np.concatenate([orc.query_sequence(r_) for r_ in rd])), (tag, "reads from the packed sequence")
            mut = []
            for s in seqs[:50]:
                p = int(rng.integers(0, len(s)))
                mut.append(s[:p] + "ACGT"[("ACGT".index(s[p]) + 1) % 4] + s[p + 1:])
            assert np.array_equal(ix.query_sequences_hash(mut), np.concatenate([orc.query_sequence(s) for s in mut])), (tag, "mut")
            ix.close(); orc.close()
            if rng.random() < 0.12:  # the same parameters through the sharded construction from slices (virtual ranks on this GPU)
                import test_gpu_shards
                test_gpu_shards.test_sharded_build_from_slices(int(rng.integers(2, 7)), k, m, n, b)
        except Exception as e:
            print("FAIL", tag, repr(e)[:400], flush=True)
            raise
        if it % 10 == 9 and verbose:
            print("ok through", tag, "%.0f s" % (time.time() - t0), flush=True)
    if verbose:
        print("FUZZ-OK %d runs" % runs)


if __name__ == "__main__":
    sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
