"""The CPU oracle (oracle/blight_oracle.c) against the golden vectors produced from the REAL reference.

This pins the oracle: every committed output of the (patched, -t 1) reference -- IDs of all indexed k-mers, negatives,
positional queries, wrong-minimizer queries, index statistics, the digests of every internal array and the primitive
known-answer vectors -- must be reproduced bit for bit.
"""
import hashlib

import numpy as np
import pytest

import golden_util as G
from oracle import oracle as O


@pytest.fixture(scope="module")
def prim():
    return G.primitives()


def test_primitives_revhash_rc_parity(prim):
    for x, y in zip(prim["revhash_in"], prim["revhash_out"]):
        assert O.p_revhash32(int(x)) == int(y)
    for m in (5, 7, 9, 11, 13, 15):
        for x, r, p in zip(prim["mmer_in_m%d" % m], prim["rc32_m%d" % m], prim["parity_m%d" % m]):
            assert O.p_rc32(int(x), m) == int(r)
            assert O.p_parity_canon(int(x), m) == int(p)


def test_primitives_survey_kats():
    # SURVEY.md App. B-4 (values printed by the reference's own functions)
    assert [O.p_nuc2int(c) for c in "ACGTa"] == [0, 1, 3, 2, 0]
    assert [O.p_revhash32(x) for x in (0, 1, 4660, 8191)] == [0, 1378302561, 66229113, -1169167018]
    assert O.p_rc32(0x1234, 7) == 0x2D8B
    assert O.p_parity_canon(0x1234, 7) == 2330 and O.p_parity_canon(0x1235, 7) == 7877
    x = O.p_str2num("ACGTTGCAACGTTGCAACGTTGCAACGTTGC")
    assert x == 0x07AD07AD07AD07AD
    assert O.p_rc(x, 31) == 0x341EB41EB41EB41E
    assert O.p_minimizer(x, 31, 7) == 6671
    assert O.p_hash_bis(0, 0xAAAAAAAA55555555) == 0x6E1BCCDB7AA2BC25
    assert O.p_hash_bis(0, 0x33333333CCCCCCCC) == 0x37F1A230C4E4AE8E
    assert O.p_hash_bis(1, 0xAAAAAAAA55555555) == 0x054676A7B01425B7
    assert O.p_hash_bis(1, 0x33333333CCCCCCCC) == 0x168C5E59DEC857BC
    assert O.p_level_hashes(0x0123456789ABCDEF, 6) == [0xDCEB9F6436030DFA, 0x21A88DFDF6AD3C27, 0x7100B26D0FA96748,
                                                      0xDF63B525DF662977, 0x78EB87392836BB4E, 0x0C972DC84B48D17A]


def test_primitives_nuc2int(prim):
    for c in range(256):
        assert O.p_nuc2int(bytes([c])) == int(prim["nuc2int"][c])


def test_primitives_kmer_rc_minimizer(prim):
    for kb, ks, ms in ((8, (15, 21, 25, 31, 32), (5, 7, 9)), (16, (33, 47, 63), (7, 9, 11))):
        for k in ks:
            xs = G.to_ints(prim["kmer_in_k%d" % k], kb)
            rs = G.to_ints(prim["kmer_rc_k%d" % k], kb)
            for x, r in zip(xs, rs):
                assert O.p_rc(x, k, kb) == r
            if k == 32:
                continue
            for m in ms:
                for x, mi in zip(xs, prim["mini_k%d_m%d" % (k, m)]):
                    assert O.p_minimizer(x, k, m, kb) == int(mi)


def test_primitives_level_hashes(prim):
    for key, lv in zip(prim["hash_keys"], prim["hash_levels"]):
        assert O.p_level_hashes(int(key), 16) == [int(x) for x in lv]
    # 128-bit keys are hashed by their low 64 bits only (bbhash.h:160-170, SURVEY C-7)
    for key, lv in zip(prim["hash_keys128"], prim["hash_levels128"]):
        assert O.p_level_hashes(int(key[0]), 16) == [int(x) for x in lv]


def test_primitives_str2num_and_chop(prim):
    for s, v in zip(prim["str2num_in"], prim["str2num_out"]):
        assert O.p_str2num(str(s)) == int(v)
    with pytest.raises(O.OracleError):
        O.p_str2num("ACGTNACGTACGTACGTACGTACGTACGTACG"[:31])
    for i in range(6):
        k, m = (int(x) for x in prim["chop%d_km" % i])
        got = O.p_chop(str(prim["chop%d_seq" % i]), k, m)
        assert got == [tuple(int(x) for x in row) for row in prim["chop%d_out" % i]]


@pytest.mark.parametrize("name", G.case_names())
def test_index_case(name):
    c = G.Case(name)
    orc = O.Oracle(c.k, c.m, c.n, c.b, c.seqs)
    st = orc.stats
    assert (st["number_kmer"], st["number_super_kmer"], st["total_nt"], st["largest_mphf"], st["largest_bucket"],
            st["positions_bits"]) == (c.number_kmer, c.number_super_kmer, c.total_nt, c.largest_mphf, c.largest_bucket,
                                      c.positions_bits)
    ids = np.concatenate([orc.query_sequence(s) for s in c.seqs])
    assert np.array_equal(ids, c.ids)
    assert np.array_equal(orc.query_kmers(c.neg_kmers), c.neg_ids)
    a = orc.arrays()
    assert G.sha(a["bucket_nt"]) == str(c["bucket_nt_sha"])
    assert G.sha(a["bucket_start"]) == str(c["bucket_start_sha"])
    assert G.sha(a["seq"]) == str(c["seq_sha"])
    assert G.sha(a["positions"]) == str(c["positions_sha"])
    for key in ("g_nkeys", "g_prefix", "g_posstart", "g_nbits"):
        assert np.array_equal(a[key], c[key]), key
    h = hashlib.sha256()
    fbk, fbr, fbg = [], [], []
    for g in range(st["ngroups"]):
        mp = orc.mphf(g)
        if mp["nwords"]:
            assert np.array_equal(mp["sizes"], c["level_sizes"][g])
            h.update(mp["bits"].tobytes())
        fbk += mp["fallback_keys"]
        fbr += mp["fallback_ranks"]
        fbg += [g] * len(mp["fallback_keys"])
    assert h.hexdigest() == str(c["mphf_bits_sha"])
    assert fbk == G.to_ints(c["fb_keys"], c.kmer_bytes) and fbr == [int(x) for x in c["fb_ranks"]]
    assert fbg == [int(x) for x in c["fb_group"]]
    # query_get_pos_unitig + query_get_hash with caller-supplied minimizers
    assert np.array_equal(orc.query_kmers_mini(c["pos_kmers"], c["pos_minis"], want_pos=True), c["pos_out"])
    assert np.array_equal(orc.query_kmers_mini(c["pos_kmers"], c["wrongmini"], want_pos=False), c["wrongmini_out"])
    for _, q, qids in c.extra_queries():
        assert np.array_equal(orc.query_sequence(q), qids)


def test_survey_digests():
    """md5 of 'one decimal ID per line' over the lambda file: SURVEY.md App. B-2 (printed by the reference)."""
    seqs = O.read_fasta(G.LAMBDA)
    for (k, m, n, b), md5 in (((31, 7, 5, 6), "c410cd4f9ed2aa4ed1bce95e64598bfb"), ((31, 9, 7, 6), "34671292f208e3e8ca82679da26080b8"),
                              ((31, 7, 10, 6), "8299d8411d0b0fa0b5e6fa5a903908b0"), ((31, 9, 14, 6), "f939b967b44f9ca298da07d72a8454be"),
                              ((31, 11, 7, 6), "e45099c04cd63a713bba7b706619e22e"), ((21, 7, 5, 4), "51b3e144d8fae5cf91e79f83ad02bbf4"),
                              ((25, 5, 3, 3), "63dd1753c8005a76c1d2351625f59283"), ((31, 7, 5, 0), "c410cd4f9ed2aa4ed1bce95e64598bfb"),
                              ((63, 7, 5, 6), "bd5b4d4bb61db773b82f70a790363ddc")):
        orc = O.Oracle(k, m, n, b, seqs)
        ids = np.concatenate([orc.query_sequence(s) for s in seqs])
        assert hashlib.md5("".join("%d\n" % i for i in ids).encode()).hexdigest() == md5
    # B-3 golden statistics
    orc = O.Oracle(31, 7, 5, 6, seqs)
    a = orc.arrays()
    assert orc.stats["largest_mphf"] == 3036 and orc.stats["largest_bucket"] == 1337
    assert orc.stats["total_nt"] == 159702 and orc.stats["positions_bits"] == 188140
    assert int((a["bucket_nt"] > 0).sum()) == 1098
    assert a["g_prefix"][:4].tolist() == [0, 1177, 2009, 3600] and a["g_nkeys"][:4].tolist() == [1177, 832, 1591, 1187]
    assert a["g_nbits"][:4].tolist() == [4, 3, 4, 4] and a["g_posstart"][:4].tolist() == [0, 4716, 7220, 13592]
    first30 = [25641, 25851, 26464, 26128, 26184, 26357, 25912, 26004, 26420, 25642, 25809, 26457, 26191, 26393, 26380, 25861,
               40404, 40892, 40564, 42107, 6123, 7045, 7163, 6537, 5997, 6094, 7016, 4880, 4874, 5112]
    assert orc.query_sequence(seqs[0])[:30].tolist() == first30


def test_param_validation_and_errors():
    assert O.check_params(33, 7, 5, 3) == (-1, "kmer size too large")
    assert O.check_params(31, 10, 5, 3) == (-2, "minimizer_length must be odd")
    assert O.check_params(31, 17, 5, 3) == (-3, "minimizer_length size too large")
    assert O.check_params(31, 7, 14, 3)[0] == -4
    assert O.check_params(31, 7, 5, 6)[0] == -5
    assert O.check_params(31, 7, 5, 3)[0] == 0 and O.check_params(63, 9, 7, 3, 16)[0] == 0
    with pytest.raises(O.OracleError) as e:
        O.Oracle(31, 7, 5, 6, ["ACGT" * 10 + "N" + "ACGT" * 10])
    assert e.value.code == -3
    with pytest.raises(O.OracleError) as e:
        O.Oracle(31, 7, 5, 6, ["ACGTACGT"])
    assert e.value.code == -2
    orc = O.Oracle(31, 7, 5, 6, ["ACGTTGCAACGTTGCAACGTTGCAACGTTGCA"])
    assert orc.query_sequence("ACGT").size == 0  # shorter than k: empty (blight.cpp:577-579)


def test_murmur3_x64_128_known_answers():
    """MurmurHash3 is dead code in the reference but named by the task: the oracle's restatement reproduces
    the known answers of SURVEY App. B-4, and -- where oracle/_ref is present -- the reference's own MurmurHash3.cpp."""
    import os
    import struct
    import numpy as np
    from oracle import oracle as O
    assert O.p_murmur3_x64_128(struct.pack("<Q", 0x0123456789ABCDEF), 0) == (0x995bb6c03277035a, 0xa51c3d420fcd7479)
    assert O.p_murmur3_x64_128(b"hello", 42) == (0xc4b8b3c960af6f08, 0x2334b875b0efbc7a)
    if os.path.exists(O.MURMUR_REF):
        rng = np.random.default_rng(9)
        for n in list(range(0, 40)) + [63, 64, 65, 100]:
            data = rng.bytes(n)
            seed = int(rng.integers(0, 1 << 32))
            assert O.p_murmur3_x64_128(data, seed) == O.ref_murmur3_x64_128(data, seed), n
