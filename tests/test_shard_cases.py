"""CPU side of the sharding tests: the helpers are pinned against the oracle, and the premise of tests/test_gpu_halo.py holds -- for
every case the oracle (= the reference's algorithm) answers, with an ID, a k-mer that is NOT indexed and that can only be read across
the cut between two shards (reference blight.cpp:729-739: the 2^b scan runs past the end of a bucket into the next one)."""
import numpy as np
import pytest

import shard_util as S


def test_canonical_kmers_helper_equals_the_oracle():
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    for k, kb in ((31, 8), (21, 8), (63, 16), (40, 16), (32, 8), (9, 8)):
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, 500))
        orc = O.Oracle(k, 7 if k >= 7 else 3, 3, 4, [s], kmer_bytes=kb)
        nts = np.array([(ord(c) >> 1) & 3 for c in s], np.uint8)
        assert np.array_equal(S.canonical_kmers_at_all_positions(nts, k, kb), orc.canonical_kmers(s).reshape(-1, kb // 8)), (k, kb)


@pytest.mark.parametrize("k,kb,m,n,b,nseq,world,seed", S.CUT_CASES)
def test_cases_hold_a_false_positive_across_a_shard_cut(k, kb, m, n, b, nseq, world, seed):
    from oracle import oracle as O
    seqs, orc, q, want, is_key, cut_fp, cut_fp_strict, bounds = S.make_cut_case(O, k, kb, m, n, b, nseq, world, seed)
    assert (want[is_key] >= 0).all()
    assert is_key.sum() == len(np.unique(np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs]), axis=0))
    assert cut_fp.sum() > 0 and cut_fp_strict.sum() > 0
    assert len(bounds) == world + 1
