"""Live array-level comparison of the CPU oracle with the REAL reference (oracle/_ref/*.so).

Runs wherever the prebuilt oracle/_ref libraries exist (they are built from /root/reference by oracle/build_ref.sh in
the dev container and travel, git-ignored, to the GPU box); skipped otherwise.  /root/reference itself is never read.
"""
import numpy as np
import pytest

import golden_util as G
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), n))


def _compare(k, m, n, b, seqs, rng, nneg=5000):
    orc = O.Oracle(k, m, n, b, seqs)
    ref = O.Reference(k, m, n, b, seqs)
    a, r = orc.arrays(), ref.arrays()
    for key in r:
        assert np.array_equal(a[key], r[key]), key
    for f in ("number_kmer", "number_super_kmer", "total_nt", "largest_mphf", "largest_bucket", "positions_bits"):
        assert orc.stats[f] == ref.stats[f], f
    for g in range(orc.stats["ngroups"]):
        mo, mr = orc.mphf(g), ref.mphf(g)
        assert mo["nwords"] == mr["nwords"]
        if mr["nwords"]:
            assert np.array_equal(mo["sizes"], mr["sizes"]) and np.array_equal(mo["bits"], mr["bits"])
        assert mo["fallback_keys"] == mr["fallback_keys"] and mo["fallback_ranks"] == mr["fallback_ranks"]
    for s in seqs[:50]:
        assert np.array_equal(orc.query_sequence(s), ref.query_sequence(s))
    kb = orc.kmer_bytes
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(nneg)]
    neg = [min(x, O.p_rc(x, k, kb)) for x in neg]
    assert np.array_equal(orc.query_kmers(neg), ref.query_kmers(neg))
    return orc, ref


@pytest.mark.parametrize("k,m,n,b", [(31, 7, 5, 6), (31, 9, 7, 2), (63, 9, 7, 8), (15, 5, 4, 3), (31, 13, 9, 5), (19, 9, 17, 1)])
def test_lambda_live(k, m, n, b):
    seqs = O.read_fasta(G.LAMBDA)
    _compare(k, m, n, b, seqs, np.random.default_rng(k * 1000 + m))


def test_random_and_lowcomplexity_live():
    rng = np.random.default_rng(99)
    seqs = [_rand_seq(rng, int(rng.integers(31, 400))) for _ in range(300)]
    seqs += [_rand_seq(rng, int(rng.integers(31, 200)), "AC") for _ in range(100)]  # heavy duplicates, phantoms, fallback
    seqs += ["A" * 64, "ACACACACACACACACACACACACACACACACACACACAC", seqs[3], seqs[3].lower()]
    orc, ref = _compare(31, 7, 5, 4, seqs, rng)
    # near-miss negatives: one substitution in indexed k-mers (exercises the 2^b scan and phantom junction k-mers)
    qs = []
    for s in seqs[:200]:
        pos = int(rng.integers(0, len(s)))
        t = s[:pos] + "ACGT"[(("ACGT".index(s[pos].upper())) + 1) % 4] + s[pos + 1:]
        assert np.array_equal(orc.query_sequence(t), ref.query_sequence(t))


def test_k63_low64_collisions_live():
    """k > 32: the MPHF hashes only the low 64 bits (SURVEY C-7) -> k-mers sharing a 32-nt suffix fall to the fallback map."""
    rng = np.random.default_rng(5)
    suffix = _rand_seq(rng, 40)
    seqs = [_rand_seq(rng, 40) + suffix for _ in range(30)] + [_rand_seq(rng, 300) for _ in range(20)]
    orc, ref = _compare(63, 9, 7, 8, seqs, rng, nneg=500)
    assert sum(len(orc.mphf(g)["fallback_keys"]) for g in range(orc.stats["ngroups"])) > 0
