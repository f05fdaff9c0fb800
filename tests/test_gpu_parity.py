"""GPU parity: the HIP build + query path (through the C ABI) against the golden vectors of the real reference and
against the CPU oracle, bit for bit -- IDs, membership, positions, counters and every internal array."""
import gzip
import hashlib

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import blight_amd
    return blight_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _build(B, k, m, n, b, seqs, kmer_bytes=0):
    ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kmer_bytes)
    ix.construct_index_from_sequences(seqs)
    return ix


def test_build_shuffle_is_a_stable_sort(B):
    """The build's shuffle (blight.cpp:234-277: super-k-mers go to their minimizer's bucket in input order) is a hand-written stable LSD
    radix sort (blmx_build.hip: k_rs_hist / k_scan_* / k_rs_scatter) -- on its own through blmx_primitive op 8 against numpy's stable
    argsort: every pass count (1 to 32 key bits), sizes around the tile and scan block boundaries, few and many equal keys."""
    rng = np.random.default_rng(77)
    for bits, n in ((1, 70000), (5, 8192), (5, 8193), (9, 1), (9, 63), (9, 65), (9, 8191), (10, 16385), (17, 300001), (18, 1 << 20),
                    (19, 123457), (27, 500000), (28, 99999), (31, 250000), (32, 8192 * 3 + 5), (17, 5_000_003)):
        hi = (1 << bits) - 1
        for kind in ("uniform", "few values", "sorted already", "reversed"):
            if kind == "uniform":
                keys = rng.integers(0, hi + 1, size=n, dtype=np.uint64).astype(np.uint32)
            elif kind == "few values":
                keys = rng.choice(rng.integers(0, hi + 1, size=3, dtype=np.uint64), size=n).astype(np.uint32)
            elif kind == "sorted already":
                keys = np.sort(rng.integers(0, hi + 1, size=n, dtype=np.uint64)).astype(np.uint32)
            else:
                keys = np.sort(rng.integers(0, hi + 1, size=n, dtype=np.uint64))[::-1].astype(np.uint32)
            got = B.primitive(8, keys, bits)
            order = np.argsort(keys, kind="stable").astype(np.uint32)
            assert np.array_equal(got[1], order), (bits, n, kind)
            assert np.array_equal(got[0], keys[order]), (bits, n, kind)
    # bits above the stated width are ignored by the sort (the build passes 2m - 1) but travel with the key
    keys = rng.integers(0, 1 << 32, size=50000, dtype=np.uint64).astype(np.uint32)
    got = B.primitive(8, keys, 11)
    order = np.argsort(keys & 0x7ff, kind="stable").astype(np.uint32)
    assert np.array_equal(got[1], order) and np.array_equal(got[0], keys[order])


def test_device_primitives(B):
    prim = G.primitives()
    assert np.array_equal(B.primitive(0, prim["revhash_in"]), prim["revhash_out"])
    for m in (5, 7, 9, 11, 13, 15):
        assert np.array_equal(B.primitive(1, prim["mmer_in_m%d" % m], m), prim["rc32_m%d" % m])
        assert np.array_equal(B.primitive(2, prim["mmer_in_m%d" % m], m), prim["parity_m%d" % m])
    codes = B.primitive(6, np.arange(256, dtype=np.uint8))
    want = np.where(prim["nuc2int"] < 0, 4, prim["nuc2int"]).astype(np.uint8)
    assert np.array_equal(codes, want)
    for kb, ks, ms in ((8, (15, 21, 25, 31, 32), (5, 7, 9)), (16, (33, 47, 63), (7, 9, 11))):
        for k in ks:
            x = prim["kmer_in_k%d" % k]
            assert np.array_equal(B.primitive(3, x, k, kmer_bytes=kb).reshape(x.shape), prim["kmer_rc_k%d" % k])
            if k == 32:
                continue
            for m in ms:
                assert np.array_equal(B.primitive(4, x, k, m, kmer_bytes=kb), prim["mini_k%d_m%d" % (k, m)])
    assert np.array_equal(B.primitive(5, prim["hash_keys"]), prim["hash_levels"])
    assert np.array_equal(B.primitive(5, np.ascontiguousarray(prim["hash_keys128"][:, 0])), prim["hash_levels128"])


def _check_against_oracle_arrays(ix, orc):
    a = orc.arrays()
    st = ix.stats()
    for key in ("bucket_nt", "bucket_start", "seq", "positions", "g_nkeys", "g_prefix", "g_posstart"):
        assert np.array_equal(ix.export(key), a[key]), key
    assert np.array_equal(ix.export("g_nbits").astype(np.uint8), a["g_nbits"])
    sizes = ix.export("level_sizes").reshape(-1, 16)
    wbase = ix.export("g_wbase")
    bits = ix.export("mphf_bits")
    fbk, fbr, fbg = [], [], []
    for g in range(st["ngroups"]):
        mp = orc.mphf(g)
        if mp["nwords"]:
            assert np.array_equal(sizes[g], mp["sizes"]), g
            assert np.array_equal(bits[int(wbase[g]):int(wbase[g + 1])], mp["bits"]), g
        else:
            assert wbase[g] == wbase[g + 1]
        fbk += mp["fallback_keys"]
        fbr += mp["fallback_ranks"]
        fbg += [g] * len(mp["fallback_keys"])
    # the device table keeps every occurrence of a duplicated key; winners (last of an equal run) = the map
    dk = G.to_ints(ix.export("fb_keys"), ix.kmer_bytes)
    dr = [int(x) for x in ix.export("fb_ranks")]
    dg = [int(x) for x in ix.export("fb_group")]
    win = [(dg[i], dk[i], dr[i]) for i in range(len(dk)) if i + 1 == len(dk) or (dg[i + 1], dk[i + 1]) != (dg[i], dk[i])]
    assert win == list(zip(fbg, fbk, fbr))


@pytest.mark.parametrize("name", G.case_names())
def test_golden_case(B, O, name):
    c = G.Case(name)
    ix = _build(B, c.k, c.m, c.n, c.b, c.seqs)
    st = ix.stats()
    assert (st["number_kmer"], st["number_super_kmer"], st["total_nt"], st["largest_mphf"], st["largest_bucket"],
            st["positions_total_size"]) == (c.number_kmer, c.number_super_kmer, c.total_nt, c.largest_mphf,
                                            c.largest_bucket, c.positions_bits)
    # IDs of every indexed k-mer, in file order (query_sequence_hash over all unitigs)
    ids = ix.query_sequences_hash(c.seqs)
    assert np.array_equal(ids, c.ids)
    # negatives, positional variant, caller-supplied (wrong) minimizers, lower case / reverse complement queries
    assert np.array_equal(ix.query_kmer_hash(c.neg_kmers), c.neg_ids)
    assert np.array_equal(ix.query_kmer_bool(c.neg_kmers), c.neg_ids >= 0)
    assert np.array_equal(ix.query_get_pos_unitig(c["pos_kmers"], c["pos_minis"]).astype(np.int64), c["pos_out"])
    assert np.array_equal(ix.query_get_hash(c["pos_kmers"], c["wrongmini"]), c["wrongmini_out"])
    for _, q, qids in c.extra_queries():
        assert np.array_equal(ix.query_sequence_hash(q), qids)
    # internal arrays: golden digests (from the real reference) ...
    assert G.sha(ix.export("bucket_nt")) == str(c["bucket_nt_sha"])
    assert G.sha(ix.export("bucket_start")) == str(c["bucket_start_sha"])
    assert G.sha(ix.export("seq")) == str(c["seq_sha"])
    assert G.sha(ix.export("positions")) == str(c["positions_sha"])
    for key in ("g_nkeys", "g_prefix", "g_posstart"):
        assert np.array_equal(ix.export(key), c[key]), key
    assert np.array_equal(ix.export("g_nbits").astype(np.uint8), c["g_nbits"])
    sizes = ix.export("level_sizes").reshape(-1, 16)
    assert np.array_equal(sizes, c["level_sizes"])
    assert hashlib.sha256(ix.export("mphf_bits").tobytes()).hexdigest() == str(c["mphf_bits_sha"])
    # ... and element-wise against the oracle (incl. fallback tables)
    orc = O.Oracle(c.k, c.m, c.n, c.b, c.seqs)
    _check_against_oracle_arrays(ix, orc)
    assert ix.number_query > 0
    ix.close()


def test_fasta_plain_gzip_and_reader_quirks(B, O, tmp_path):
    """construct_index(file): plain and gzip, empty lines tolerated like blight.cpp:212-229."""
    c = G.Case("lambda_k31_m7_n5_b6")
    ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    ix.construct_index(G.LAMBDA)
    assert np.array_equal(ix.query_sequences_hash(c.seqs), c.ids)
    gz = tmp_path / "l.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(G.LAMBDA, "rb").read())
    ix2 = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    ix2.construct_index(str(gz))
    assert np.array_equal(ix2.query_sequences_hash(c.seqs), c.ids)
    quirk = tmp_path / "q.fa"
    quirk.write_text(">a\n%s\n\nskipped-line\n>b\n\n>c\n%s" % (c.seqs[0], c.seqs[2]))
    want = O.read_fasta(str(quirk))
    ix3 = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    ix3.construct_index(str(quirk))
    orc = O.Oracle(31, 7, 5, 6, want)
    assert ix3.number_kmer == orc.stats["number_kmer"]
    assert np.array_equal(ix3.query_sequences_hash(want), np.concatenate([orc.query_sequence(s) for s in want]))
    tp, fp = ix.file_query(G.LAMBDA)
    assert (tp, fp) == (48462, 0)  # "Good kmer" / "Erroneous kmers" of the reference's self-query


def test_errors(B, tmp_path):
    ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    with pytest.raises(ValueError, match="Invalid char in DNA"):
        ix.construct_index_from_sequences(["ACGT" * 10 + "N" + "ACGT" * 10])
    ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    with pytest.raises(B.BlightError) as e:
        ix.construct_index_from_sequences(["ACGTACGT"])
    assert e.value.code == B.BLMX_ESHORT
    ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    with pytest.raises(OSError):
        ix.construct_index(str(tmp_path / "missing.fa"))
    with pytest.raises(B.BlightError) as e:
        ix.query_kmer_hash([1, 2, 3])
    assert e.value.code == B.BLMX_ESTATE
    ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    ix.construct_index_from_sequences(["ACGTTGCAACGTTGCAACGTTGCAACGTTGCA"])
    assert ix.query_sequence_hash("ACGT").size == 0 and ix.query_sequence_bool("ACGT") == (0, 0)
    with pytest.raises(ValueError, match="Invalid char in DNA"):
        ix.query_sequence_hash("ACGTTGCAACGTTGCAACGTNGCAACGTTGCAAA")
    # empty input: everything absent
    ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    ix.construct_index_from_sequences([])
    assert ix.number_kmer == 0 and ix.query_kmer_hash([0, 5, 77]).tolist() == [-1, -1, -1]
    # an empty file through the streaming ingest, then a file made of blank lines only
    for content in ("", "\n\n\n"):
        f = tmp_path / "empty.fa"
        f.write_text(content)
        ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
        ix.construct_index(str(f))
        assert ix.number_kmer == 0 and ix.query_kmer_bool([123]).tolist() == [False]
        assert ix.file_query(str(f)) == (0, 0)


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), n))


@pytest.mark.parametrize("k,m,n,b,kb", [(31, 9, 7, 6, 8), (31, 7, 3, 4, 8), (27, 11, 14, 5, 8), (63, 9, 7, 8, 16), (47, 7, 5, 3, 16),
                                        (21, 5, 0, 9, 8), (31, 9, 7, 6, 16)])
def test_random_vs_oracle(B, O, k, m, n, b, kb):
    """seeded random unitigs incl. low-complexity ones (duplicates -> fallback, phantoms): everything vs the oracle."""
    rng = np.random.default_rng(k * 100 + m)
    seqs = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 80))) for _ in range(1500)]
    seqs += [_rand_seq(rng, k + int(rng.integers(0, 150)), "AC") for _ in range(60)]
    seqs += ["A" * (k + 40), seqs[5], seqs[5].lower(), seqs[7][:k]]
    ix = _build(B, k, m, n, b, seqs, kmer_bytes=kb)
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    st = ix.stats()
    for f, g in (("number_kmer", "number_kmer"), ("number_super_kmer", "number_super_kmer"), ("total_nt", "total_nt"),
                 ("largest_mphf", "largest_mphf"), ("largest_bucket", "largest_bucket"), ("positions_total_size", "positions_bits")):
        assert st[f] == orc.stats[g], f
    _check_against_oracle_arrays(ix, orc)
    want = np.concatenate([orc.query_sequence(s) for s in seqs])
    assert np.array_equal(ix.query_sequences_hash(seqs), want)
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(20000)]
    neg = [min(x, O.p_rc(x, k, kb)) for x in neg]
    assert np.array_equal(ix.query_kmer_hash(neg), orc.query_kmers(neg))
    # near misses: one substitution per sequence (2^b scan, junction phantoms)
    mut = []
    for s in seqs[:400]:
        p = int(rng.integers(0, len(s)))
        mut.append(s[:p] + "ACGT"[("ACGT".index(s[p].upper()) + 1) % 4] + s[p + 1:])
    assert np.array_equal(ix.query_sequences_hash(mut), np.concatenate([orc.query_sequence(s) for s in mut]))
    # non-canonical k-mers can never match (blight.cpp:733-735)
    km = orc.canonical_kmers(seqs[0])
    kl = G.to_ints(km, kb)[:50]
    nonc = [O.p_rc(x, k, kb) for x in kl if O.p_rc(x, k, kb) != x]
    assert np.array_equal(ix.query_kmer_hash(nonc), orc.query_kmers(nonc))


def test_build_is_deterministic(B):
    rng = np.random.default_rng(3)
    seqs = [_rand_seq(rng, 31 + int(rng.integers(0, 300))) for _ in range(800)] + [_rand_seq(rng, 60, "AC") for _ in range(40)]
    a = _build(B, 31, 9, 7, 6, seqs)
    b = _build(B, 31, 9, 7, 6, seqs)
    for key in ("seq", "positions", "mphf_bits", "fb_keys", "fb_ranks", "bucket_start"):
        assert np.array_equal(a.export(key), b.export(key)), key


@pytest.mark.parametrize("k,m,n,b,kb", [(31, 9, 7, 6, 8), (63, 9, 7, 8, 16), (21, 3, 2, 3, 8), (31, 7, 13, 0, 8)])
def test_binned_batch_query_equals_direct(B, O, k, m, n, b, kb):
    """The locality-binned batch path (partition by minimizer prefix -> query -> gather back) returns exactly the direct
    kernel's IDs, in the caller's order, and both equal the oracle."""
    rng = np.random.default_rng(k + m)
    seqs = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 60))) for _ in range(2500)]
    seqs += [_rand_seq(rng, k + 30, "AC") for _ in range(30)]
    ix = _build(B, k, m, n, b, seqs, kmer_bytes=kb)
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    pos = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(30000)]
    neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
    q = np.concatenate([pos, neg])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    ix.set_query_binning(-1)
    direct = ix.query_kmer_hash(q.reshape(-1))
    ix.set_query_binning(1)
    binned = ix.query_kmer_hash(q.reshape(-1))
    assert np.array_equal(direct, binned)
    assert np.array_equal(direct, orc.query_kmers(q.reshape(-1) if kb == 8 else q))
    assert int((direct >= 0).sum()) >= len(pos)
    # tiny and empty batches through the binned path
    assert np.array_equal(ix.query_kmer_hash(q[:3].reshape(-1)), direct[:3])
    assert ix.query_kmer_hash(np.zeros(0, np.uint64)).size == 0


def test_all_mphf_build_paths(B, O, monkeypatch):
    """BBHash levels are built by one workgroup per group in LDS (groups of up to ~1.18 M keys, in 1-4 slices), by one workgroup
    per (level, slice) for larger groups, or by the global level kernels (test hook BLIGHT_BUILD_NO_LDS: 1 = every group, 2 = the
    large groups); all must give the oracle's arrays, alone and mixed inside one index."""
    c = G.Case("dup_k31_m7_n5_b6")  # exercises 16 levels + fallback
    for no_lds in ("0", "1"):
        monkeypatch.setenv("BLIGHT_BUILD_NO_LDS", no_lds)
        ix = _build(B, c.k, c.m, c.n, c.b, c.seqs)
        assert (ix.stats()["mphf_groups_lds_built"] > 0) == (no_lds == "0")
        assert np.array_equal(ix.query_sequences_hash(c.seqs), c.ids)
        _check_against_oracle_arrays(ix, O.Oracle(c.k, c.m, c.n, c.b, c.seqs))
    # a 3-letter source skews the minimizers, so some of the 4 groups exceed the single-workgroup budget and some do not;
    # duplicated unitigs push keys through all 16 levels into the fallback table of the large groups too
    rng = np.random.default_rng(21)
    seqs = [_rand_seq(rng, 31 + int(rng.integers(20, 400)), "ACG") for _ in range(20000)]
    seqs += seqs[:300]
    k, m, n, b = 31, 9, 2, 6
    orc = O.Oracle(k, m, n, b, seqs)
    nk = orc.arrays()["g_nkeys"]
    nonempty = int((nk > 0).sum())
    want = np.concatenate([orc.query_sequence(s) for s in seqs[:500]])
    for no_lds, lds_built in (("0", nonempty), ("2", None), ("1", 0)):
        monkeypatch.setenv("BLIGHT_BUILD_NO_LDS", no_lds)
        ix = _build(B, k, m, n, b, seqs)
        st = ix.stats()
        if lds_built is None:
            assert 0 < st["mphf_groups_lds_built"] < nonempty, (st["mphf_groups_lds_built"], nk.tolist())
        else:
            assert st["mphf_groups_lds_built"] == lds_built
        assert st["fallback_keys"] > 0
        _check_against_oracle_arrays(ix, orc)
        assert np.array_equal(ix.query_sequences_hash(seqs[:500]), want)
    monkeypatch.setenv("BLIGHT_BUILD_NO_LDS", "0")


@pytest.mark.parametrize("k,m,n,b,kb", [(31, 9, 7, 6, 8), (63, 9, 7, 8, 16)])
def test_index_image_round_trip(B, O, tmp_path, k, m, n, b, kb):
    """dump_disk -> load into a fresh object (and from_file): every exported array, every counter and every answer is
    identical; wrong parameters, truncated and foreign files are refused with the documented errors."""
    rng = np.random.default_rng(k)
    seqs = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 60))) for _ in range(1500)] + [_rand_seq(rng, k + 40, "AC") for _ in range(20)]
    a = _build(B, k, m, n, b, seqs, kmer_bytes=kb)
    path = str(tmp_path / "index.blmx")
    a.dump_disk(path)
    c = B.kmer_Set_Light.from_file(path)
    d = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
    d.load_disk(path)
    for other in (c, d):
        for key in B.EXPORT:
            assert np.array_equal(a.export(key), other.export(key)), key
        sa, so = a.stats(), other.stats()
        sa.pop("number_query"), so.pop("number_query")
        assert sa == so
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    q = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs[:400]]).reshape(-1)
    want = orc.query_kmers(q if kb == 8 else q.reshape(-1, 2))
    assert np.array_equal(c.query_kmer_hash(q), want) and np.array_equal(d.query_kmer_hash(q), want)
    assert np.array_equal(c.query_sequence_hash(seqs[0]), orc.query_sequence(seqs[0]))
    # refusals
    with pytest.raises(ValueError):
        B.kmer_Set_Light(k, m, n - 1, min(n - 1, 3), 1, b, kmer_bytes=kb).load_disk(path)
    with pytest.raises(B.BlightError):
        d.load_disk(path)  # already built
    raw = open(path, "rb").read()
    cut = str(tmp_path / "cut.blmx")
    open(cut, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(OSError):
        B.kmer_Set_Light.from_file(cut)
    with pytest.raises(OSError):
        B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb).load_disk(str(tmp_path / "nope.blmx"))


@pytest.mark.parametrize("k,kb,b", [(31, 8, 6), (63, 16, 8), (21, 8, 3)])
def test_scan_filter_false_candidates(B, O, k, kb, b):
    """The query kernel scans the 2^b window with a filter on the 16 leading nucleotides and verifies the first candidate
    exactly; a DIFFERENT k-mer with the same 16-nt prefix earlier in the window must send it to the exact scan.  The data
    plants such pairs (a unit repeated after p nucleotides with a mutation behind the shared prefix); the test first
    proves from the exported index that false candidates really precede matches, then compares with the oracle."""
    rng = np.random.default_rng(100 + k)
    m, n = 9, 7
    seqs = []
    for _ in range(600):
        if (1 << b) > 27:
            p = int(rng.integers(17, 28))          # distance between the two copies
            u = _rand_seq(rng, p)
            v = list(u)
            mut = int(rng.integers(16, p)) if p > 16 else 16
            v[mut % p] = "ACGT"[("ACGT".index(v[mut % p]) + 1 + int(rng.integers(0, 3))) % 4]
            core = u + "".join(v)
        else:                                      # small windows: a short period, k-mers p apart share k-2 nucleotides
            p = int(rng.integers(2, 1 << b))
            core = (_rand_seq(rng, p) * k)[: k - 2 + p]
        seqs.append(_rand_seq(rng, int(rng.integers(0, 40))) + core + _rand_seq(rng, k + int(rng.integers(0, 30))))
    ix = _build(B, k, m, n, b, seqs, kmer_bytes=kb)
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    q = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
    ids = ix.query_kmer_hash(q.reshape(-1))
    assert np.array_equal(ids, orc.query_kmers(q.reshape(-1) if kb == 8 else q))
    assert (ids >= 0).all()
    # positions (first match in the window) through the same scan
    ints = O.array_to_kmers(q, kb)
    minis = np.array([O.p_minimizer(x, k, m, kb) for x in ints], dtype=np.uint32)
    pos = ix.query_get_pos_unitig(q.reshape(-1), minis)
    assert np.array_equal(pos.astype(np.int64), orc.query_kmers_mini(q.reshape(-1) if kb == 8 else q, minis, want_pos=True))
    # proof that the filter met false candidates: decode the bucket sequences and look at the windows
    words = ix.export("seq")
    nts = ((words[:, None] >> np.arange(62, -2, -2, dtype=np.uint64)[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
    bstart = ix.export("bucket_start")
    P = 1 << b
    false_first = 0
    for i in range(0, len(pos), 7 if P > 16 else 1):
        g = int(bstart[minis[i]]) + int(pos[i])
        w = g - (int(pos[i]) & (P - 1))
        tgt = nts[g:g + 16]
        for j in range(w, g):
            if np.array_equal(nts[j:j + 16], tgt):
                false_first += 1
                break
    assert false_first > 0


def test_device_murmur3_x64_128(B, O):
    """device MurmurHash3_x64_128 (blmx_primitive op 7) == oracle restatement: SURVEY App. B-4 known answers and random
    8- and 16-byte records (the k-mer widths), full and partial lengths."""
    import struct
    rec = np.frombuffer(struct.pack("<Q", 0x0123456789ABCDEF), dtype=np.uint8)
    assert [int(x) for x in B.primitive(7, rec, 0, 0, 8)[0]] == [0x995bb6c03277035a, 0xa51c3d420fcd7479]
    hello = np.frombuffer(b"hello\0\0\0", dtype=np.uint8)
    assert [int(x) for x in B.primitive(7, hello, 42, 5, 8)[0]] == [0xc4b8b3c960af6f08, 0x2334b875b0efbc7a]
    rng = np.random.default_rng(4)
    for kb in (8, 16):
        data = rng.integers(0, 256, size=(500, kb), dtype=np.uint8)
        for length in (0, 3, kb - 1):
            seed = int(rng.integers(0, 1 << 31))
            got = B.primitive(7, data, seed, length, kb)
            n = length or kb
            want = [O.p_murmur3_x64_128(bytes(r[:n]), seed) for r in data]
            assert [(int(a), int(b)) for a, b in got] == want


@pytest.mark.parametrize("k,kb,m,n,b,nseq", [(31, 8, 7, 5, 6, 400), (21, 8, 5, 3, 8, 300), (63, 16, 9, 7, 8, 300)])
def test_absent_filter_keeps_every_answer_of_the_reference(B, O, k, kb, m, n, b, nseq):
    """The absent-key filter (a blocked Bloom filter over the keys, index_device.h) must never change an answer.  The hard case are
    the reference's own false positives: its 2^b scan compares whatever it reads from the bucket's concatenated sequence, so a k-mer
    read across the junction of two super-k-mers -- not a key of anybody -- is answered with an ID when the MPHF happens to send it to
    a rank whose window covers that place.  Every k-mer readable ANYWHERE in the packed sequence is queried: the answers equal the
    oracle's with the filter on, off, and before its second pass has finished; the reference's false positives really occur in this
    data, and a filter without the second pass (test hook) loses exactly those."""
    rng = np.random.default_rng(1000 + k + b)
    seqs = [_rand_seq(rng, k + int(rng.integers(0, 60))) for _ in range(nseq)]
    seqs += [seqs[1], seqs[2][3:], "A" * (k + 9)]
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    ix = _build(B, k, m, n, b, seqs, kmer_bytes=kb)
    assert ix.stats()["device_bytes"] > 0
    words = ix.export("seq")
    total_nt = ix.stats()["total_nt"]
    nts = ((words[:, None] >> np.arange(62, -2, -2, dtype=np.uint64)[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)[:total_nt]
    # the windows of the last buckets run past the end of the sequence: those reads are defined as 'A' (here and in the oracle; the
    # reference reads its vector<bool> out of range), so k-mers that are only readable with that padding belong to the set too
    nts = np.concatenate([nts, np.zeros((1 << b) + k, np.uint8)])
    vals = set()
    v = 0
    mask = (1 << (2 * k)) - 1
    for p, c in enumerate(nts.tolist()):
        v = ((v << 2) | c) & mask
        if p >= k - 1:
            vals.add(min(v, O.p_rc(v, k, kb)))
    keys = set()
    for s in seqs:
        keys.update(O.array_to_kmers(orc.canonical_kmers(s), kb))
    assert keys <= vals
    q = O.kmers_to_array(sorted(vals), kb)
    want = orc.query_kmers(q if kb == 8 else q.reshape(-1, 2))
    is_key = np.array([x in keys for x in sorted(vals)])
    assert (want[is_key] >= 0).all()
    ref_false_positives = int((want[~is_key] >= 0).sum())
    assert ref_false_positives > 0, "the data must contain answers the reference gives for k-mers that are not keys"
    got_early = ix.query_kmer_hash(q.reshape(-1))          # maybe before the background pass has finished: unfiltered then
    ix.set_tuning("absent_filter", 1)                        # waits for the pass
    got_on = ix.query_kmer_hash(q.reshape(-1))
    ix.set_tuning("query_binning", 1)                        # the partitioned pipeline
    got_binned = ix.query_kmer_hash(np.tile(q.reshape(-1, kb // 8), (3, 1)).reshape(-1))
    ix.set_tuning("query_binning", 0)
    ix.set_tuning("absent_filter", 0)
    got_off = ix.query_kmer_hash(q.reshape(-1))
    for got in (got_early, got_on, got_off, got_binned[:len(want)], got_binned[len(want):2 * len(want)]):
        assert np.array_equal(got, want)
    # sequences through the read front end (fused kernel): junction k-mers inside longer reads
    reads = ["".join("ACTG"[c] for c in nts[a:a + 3 * k].tolist()) for a in range(0, max(1, total_nt - 3 * k), 97)]
    reads.append("".join("ACTG"[c] for c in nts[max(0, total_nt - 2 * k):total_nt + k].tolist()))  # runs into the padding
    ix.set_tuning("absent_filter", 1)
    assert np.array_equal(ix.query_sequences_hash(reads), np.concatenate([orc.query_sequence(r) for r in reads]))
    # the image carries the finished filter
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        ix.dump_disk(td + "/i.blmx")
        ld = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
        ld.load_disk(td + "/i.blmx")
        assert np.array_equal(ld.query_kmer_hash(q.reshape(-1)), want)
        assert ld.stats()["device_bytes"] == ix.stats()["device_bytes"]
    # an image written right after the build (the writer waits for the background passes), an index destroyed right after its build
    with tempfile.TemporaryDirectory() as td:
        fresh = _build(B, k, m, n, b, seqs, kmer_bytes=kb)
        fresh.dump_disk(td + "/j.blmx")
        fresh.close()
        ld = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
        ld.load_disk(td + "/j.blmx")
        assert np.array_equal(ld.query_kmer_hash(q.reshape(-1)), want)
    _build(B, k, m, n, b, seqs, kmer_bytes=kb).close()
    # without the second pass the filter rejects (some of) the reference's false positives and nothing else
    bare = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
    bare.set_tuning("filter_junctions", 0)
    bare.construct_index_from_sequences(seqs)
    got_bare = bare.query_kmer_hash(q.reshape(-1))
    wrong = got_bare != want
    assert wrong.sum() > 0 and not wrong[is_key].any() and (got_bare[wrong] == -1).all()
    none = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
    none.set_tuning("filter_bits", 0)
    none.construct_index_from_sequences(seqs)
    assert np.array_equal(none.query_kmer_hash(q.reshape(-1)), want)
    assert none.stats()["device_bytes"] < ix.stats()["device_bytes"]


@pytest.mark.parametrize("k,kb,m,n,b", [(31, 8, 9, 5, 6), (27, 8, 7, 3, 0), (63, 16, 9, 7, 8), (21, 8, 5, 0, 2)])
def test_level0_layouts_answer_alike(B, O, k, kb, m, n, b):
    """The MPHF's level 0 has three physical layouts (index_device.h): 16-B lines + a separate filter array (0), 64-B fat lines that carry
    the filter (1), 64-B records of 32 level bits that carry the filter AND the position fields of the keys they rank (2, the default
    while the index is small against the card).  None of them is observable: the exported arrays (the reference's logical arrays) are
    equal among the three and to the oracle's, every k-mer and read gets the oracle's answer -- also positional queries, queries with
    a caller's minimizer, the partitioned pipeline and a reloaded image.  b = 0 with a long low-complexity unitig makes position
    fields of 18+ bits, so that a record holds fewer fields than it can rank keys and the rest is read from `positions`."""
    rng = np.random.default_rng(4000 + k + b)
    seqs = [_rand_seq(rng, k + int(rng.integers(0, 120))) for _ in range(1200)]
    seqs += [_rand_seq(rng, 150000 if b == 0 else 3000, "AC"), seqs[3], "A" * (k + 20), seqs[10][1:]]
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    a = orc.arrays()
    q = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs[:1200] + seqs[1201:]])
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(3000)]
    neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
    allq = np.ascontiguousarray(np.concatenate([q, neg, q[::-1][:5000]]))
    flat = allq.reshape(-1)
    want = orc.query_kmers(flat if kb == 8 else allq)
    sub = allq[::17]  # positional queries and queries with the caller's minimizer (python minimizers: a sample)
    sub_minis = np.array([O.p_minimizer(x, k, m, kb) for x in O.array_to_kmers(sub, kb)], dtype=np.uint32)
    want_pos = orc.query_kmers_mini(sub.reshape(-1) if kb == 8 else sub, sub_minis, want_pos=True)
    want_mini = orc.query_kmers_mini(sub.reshape(-1) if kb == 8 else sub, sub_minis)
    reads = [s[i:i + 2 * k + 7] for s in seqs[:200] for i in (0, 5)] + [_rand_seq(rng, 3 * k) for _ in range(50)]
    want_reads = np.concatenate([orc.query_sequence(r) for r in reads])
    exported = {}
    sizes = {}
    for layout in (0, 1, 2):
        ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
        ix.set_tuning("fat_lines", layout)
        if b > 6:
            ix.set_tuning("filter_bits", 8)  # (the automatic 16 bits of the wide windows keep the filter a separate array)
        ix.construct_index_from_sequences(seqs)
        ix.set_tuning("absent_filter", 1)
        sizes[layout] = ix.stats()["device_bytes"]
        exported[layout] = {name: ix.export(name) for name in B.EXPORT}
        for key in ("bucket_nt", "bucket_start", "seq", "positions", "g_nkeys", "g_prefix", "g_posstart", "g_nbits"):
            assert np.array_equal(exported[layout][key], a[key]), (layout, key)
        for pol in (-1, 1, 1):  # direct kernel; partitioned pipeline: exact pass, then one pass with the learnt regions
            ix.set_query_binning(pol)
            assert np.array_equal(ix.query_kmer_hash(flat), want), (layout, pol)
        ix.set_query_binning(0)
        ix.set_tuning("absent_filter", 0)
        assert np.array_equal(ix.query_kmer_hash(flat), want), (layout, "unfiltered")
        ix.set_tuning("absent_filter", 1)
        assert np.array_equal(ix.query_get_pos_unitig(np.ascontiguousarray(sub).reshape(-1), sub_minis).astype(np.int64), want_pos), (layout, "pos")
        assert np.array_equal(ix.query_get_hash(np.ascontiguousarray(sub).reshape(-1), sub_minis), want_mini), (layout, "caller's minimizer")
        assert np.array_equal(ix.query_sequences_hash(reads), want_reads), (layout, "reads")
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            ix.dump_disk(td + "/i.blmx")
            ld = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
            ld.load_disk(td + "/i.blmx")
            assert np.array_equal(ld.query_kmer_hash(flat), want), (layout, "image")
            assert ld.stats()["device_bytes"] == sizes[layout]
            for name in B.EXPORT:
                assert np.array_equal(ld.export(name), exported[layout][name]), (layout, "image", name)
        ix.close()
    for name in B.EXPORT:
        assert np.array_equal(exported[0][name], exported[1][name]) and np.array_equal(exported[0][name], exported[2][name]), name
    assert sizes[2] > sizes[1]  # what the records cost


@pytest.mark.parametrize("k,kb,m,n,b,fat", [(31, 8, 7, 5, 6, 1), (31, 8, 9, 7, 4, 0), (63, 16, 9, 7, 8, 1), (31, 8, 9, 6, 6, 2), (63, 16, 9, 7, 5, 2)])
def test_build_passes_do_not_change_the_index(B, O, k, kb, m, n, b, fat):
    """The construction works through the MPHF groups in passes (keys, survivor lists and position fields of one range of groups at a
    time: the reference builds one of 2^s super-buckets at a time, blight.cpp:335-356).  Every array, counter and answer is the same for
    1, 2, 4, 8 and 32 passes -- duplicates and low-complexity sequence put fallback keys into several groups, so the fallback table is
    assembled from several passes -- and equals the oracle's."""
    rng = np.random.default_rng(77 + k + b)
    seqs = [_rand_seq(rng, k + int(rng.integers(0, 90))) for _ in range(1500)]
    seqs += [seqs[3], seqs[40], seqs[41][2:] if len(seqs[41]) > k + 2 else seqs[41], "A" * (k + 30), "AC" * (k + 5), "ACG" * (k + 1), seqs[900]] + seqs[1000:1040]
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    base = None
    q = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
    want = orc.query_kmers(q.reshape(-1) if kb == 8 else q)
    for passes in (1, 2, 4, 8, 32):
        ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
        ix.set_tuning("build_passes", passes)
        ix.set_tuning("fat_lines", fat)
        ix.construct_index_from_sequences(seqs)
        ix.set_tuning("absent_filter", 1)
        arrays = {name: ix.export(name) for name in B.EXPORT}
        st = {f: v for f, v in ix.stats().items() if f not in ("build_seconds", "number_query", "build_scratch_peak_bytes")}
        assert np.array_equal(ix.query_kmer_hash(q.reshape(-1)), want), passes
        if base is None:
            base = (arrays, st)
            assert st["fallback_keys"] > 0 and len(set(arrays["fb_group"].tolist())) > 1  # fallback keys in more than one group
            a = orc.arrays()
            for key in ("bucket_nt", "bucket_start", "seq", "positions", "g_nkeys", "g_prefix", "g_posstart"):
                assert np.array_equal(arrays[key], a[key]), key
        else:
            for name in B.EXPORT:
                assert np.array_equal(arrays[name], base[0][name]), (passes, name)
            assert st == base[1], (passes, [(f, st[f], base[1][f]) for f in st if st[f] != base[1][f]])
        rq = q[rng.integers(0, len(q), 2000)]
        import tempfile
        if passes == 4:
            with tempfile.TemporaryDirectory() as td:
                ix.dump_disk(td + "/p.blmx")
                ld = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
                ld.load_disk(td + "/p.blmx")
                assert np.array_equal(ld.query_kmer_hash(rq.reshape(-1)), ix.query_kmer_hash(rq.reshape(-1)))
        ix.close()


def test_scratch_cache_is_reused_and_released(B, O):
    """blmx_release_scratch: the scratch blocks of a finished build and the pinned window buffers of a destroyed index wait for the next
    build of the process; the call hands them back to the driver, and an index built from recycled blocks is the same index."""
    rng = np.random.default_rng(5)
    seqs = [_rand_seq(rng, 200) for _ in range(3000)]
    B.release_scratch()
    arrays = []
    for rep in range(2):
        ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
        ix.construct_index_from_sequences(seqs)
        ix.set_tuning("absent_filter", 1)  # waits for the filter's passes: the build's temporaries are parked afterwards
        ix.query_sequences_hash(seqs[:1500])  # host-buffer calls take pinned buffers: window slots ...
        ix.query_sequences_hash(seqs[:1])     # ... and the 1-MiB slot of the small calls
        arrays.append({name: ix.export(name) for name in B.EXPORT})
        ix.close()
    for name in B.EXPORT:
        assert np.array_equal(arrays[0][name], arrays[1][name]), name
    dev, pinned = B.release_scratch()
    assert dev > 0 and pinned > 0
    assert B.release_scratch() == (0, 0)


@pytest.mark.parametrize("env", [{"BLIGHT_SCRATCH_CACHE_GB": "0"}, {"BLIGHT_POISON": "1"}, {"BLIGHT_POISON": "1", "BLIGHT_BUILD_PASSES": "3"}])
def test_allocation_hooks_do_not_change_answers(env):
    """A fresh process with the block cache switched off (every buffer straight from / back to the driver), and one in which every
    device buffer that is not asked to be zero starts as 0xA5 bytes (recycled blocks hold arbitrary data, fresh ones happen to be
    zero): a golden case builds and answers like the reference, through the k-mer, sequence and file paths."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np
import blight_amd as B
import golden_util as G
c = G.Case("lambda_k31_m7_n5_b6")
for rep in range(3):
    ix = B.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    if rep == 1:
        ix.construct_index(G.LAMBDA)
    else:
        ix.construct_index_from_sequences(c.seqs)
    ix.set_tuning("absent_filter", 1)
    assert np.array_equal(ix.query_sequences_hash(c.seqs), c.ids)
    assert np.array_equal(ix.query_kmer_hash(c.neg_kmers.reshape(-1)), c.neg_ids)
    assert ix.file_query(G.LAMBDA) == (48462, 0)
    ix.close()
dev, pinned = B.release_scratch()
if os.environ.get("BLIGHT_SCRATCH_CACHE_GB") == "0":
    assert dev == 0, dev
print("HOOKS-OK")
'''
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=e, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "HOOKS-OK" in out.stdout, out.stderr[-2000:]


def test_a_kmer_without_a_slot_fails_the_call(B):
    """The one-pass partition places every k-mer in its bin's region or in the overflow region, which holds a whole batch: nothing can be
    left over.  If something ever were (a bug), the k-mer would come back as "absent" -- so the kernel counts it and the call fails
    instead (the next call, when the batch was answered asynchronously).  Test hook: an overflow region of 256 slots under a batch
    that needs thousands (one bin's region is sized for a uniform batch, the batch is one k-mer repeated)."""
    rng = np.random.default_rng(5)
    seqs = [_rand_seq(rng, 200) for _ in range(3000)]
    ix = _build(B, 31, 9, 6, 6, seqs)
    ix.set_query_binning(1)
    ix.set_tuning("partition", 1)
    from oracle import oracle as O
    orc = O.Oracle(31, 9, 6, 6, seqs)
    q = np.concatenate([orc.canonical_kmers(s).reshape(-1) for s in seqs[:600]])
    big = np.tile(q, 12)[:1 << 20]
    ids = ix.query_kmer_hash(big)
    assert (ids >= 0).all()
    ix.set_tuning("test_ovf_cap", 256)
    skew = np.full(1 << 20, q[0], dtype=np.uint64)
    with pytest.raises(Exception) as e:
        ix.query_kmer_hash(skew)
        ix.query_kmer_hash(big)  # (at the latest)
    assert "found no slot" in str(e.value)
    ix.set_tuning("test_ovf_cap", 0)
    assert np.array_equal(ix.query_kmer_hash(skew), np.full(1 << 20, ids[0]))
    assert np.array_equal(ix.query_kmer_hash(big), ids)


def test_device_batches_of_any_size_are_answered_in_pieces(B, O):
    """A device-resident batch above 2^30 k-mers is answered piece by piece (the partition numbers slots with 32 bits); with the piece
    size lowered to 2^20 (test hook) a batch of 3 * 2^20 + 17 k-mers goes through four pieces, the last one ragged, with and without
    the partitioned pipeline, and equals the oracle."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(9)
    seqs = [_rand_seq(rng, 300) for _ in range(4000)]
    ix = _build(B, 31, 9, 6, 6, seqs)
    orc = O.Oracle(31, 9, 6, 6, seqs)
    q = np.concatenate([orc.canonical_kmers(s).reshape(-1) for s in seqs[:1500]])
    neg = rng.integers(0, 1 << 62, size=len(q) // 2, dtype=np.uint64)
    neg = np.array([min(int(x), O.p_rc(int(x), 31, 8)) for x in neg[:20000]], dtype=np.uint64)
    allq = np.concatenate([q, neg])
    n = 3 * (1 << 20) + 17
    allq = np.tile(allq, n // len(allq) + 1)[:n]
    want = orc.query_kmers(allq)
    dq = torch.from_numpy(allq.view(np.int64)).cuda()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    ix.set_tuning("max_batch", 1 << 20)
    for pol in (1, 1, -1):
        ix.set_query_binning(pol)
        out.fill_(-7)
        ix.query_kmers_device(dq.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want), pol


@pytest.mark.parametrize("k,kb,m,n,b", [(31, 8, 7, 3, 4), (63, 16, 9, 5, 8), (15, 8, 5, 0, 2)])
def test_fallback_hash_table_on_and_off(B, O, k, kb, m, n, b):
    """Keys that survive all 16 MPHF levels (duplicated k-mers do, always) live in the reference's unordered_map (bbhash.h:567-575,
    711-728).  Here: the sorted entries (exported, part of the image) and, derived from them, one open-addressing table per MPHF group
    that lookups probe (fb_table = 1, default) instead of searching the sorted entries (0).  Same answers either way, for every key
    incl. the duplicated ones ("last assignment wins"), for absent k-mers that reach the fallback stage, and after an image round trip."""
    rng = np.random.default_rng(31 + k)
    seqs = [_rand_seq(rng, k + int(rng.integers(0, 80))) for _ in range(400)]
    seqs += [_rand_seq(rng, 4000, "AC"), _rand_seq(rng, 3000, "AC"), "A" * (k + 50), "AC" * (k + 9)] + seqs[:120] + [seqs[7][2:]]
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    q = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(4000)]
    neg += [int("".join(rng.choice(["00", "01"], size=k)), 2) for _ in range(4000)]  # {A, C} k-mers: the groups that hold the fallback keys
    neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
    allq = np.ascontiguousarray(np.concatenate([q, neg]))
    want = orc.query_kmers(allq.reshape(-1) if kb == 8 else allq)
    sizes = {}
    for table in (1, 0):
        ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
        ix.set_tuning("fb_table", table)
        ix.construct_index_from_sequences(seqs)
        ix.set_tuning("absent_filter", 1)
        st = ix.stats()
        assert st["fallback_keys"] > 100
        sizes[table] = st["device_bytes"]
        for pol in (-1, 1, 1):
            ix.set_query_binning(pol)
            assert np.array_equal(ix.query_kmer_hash(allq.reshape(-1)), want), (table, pol)
        ix.set_tuning("absent_filter", 0)  # every absent k-mer now walks the levels, many of them into the fallback stage
        assert np.array_equal(ix.query_kmer_hash(allq.reshape(-1)), want), (table, "unfiltered")
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            ix.dump_disk(td + "/i.blmx")
            ld = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
            ld.set_tuning("fb_table", table)
            ld.load_disk(td + "/i.blmx")
            ld.set_tuning("absent_filter", 0)
            assert np.array_equal(ld.query_kmer_hash(allq.reshape(-1)), want), (table, "image")
        ix.close()
    assert sizes[1] > sizes[0]
