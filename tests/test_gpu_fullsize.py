"""Size-independent properties at BASELINE scale (chr1-scale C3: 2.3e8 31-mers; the human-scale C4 run is checked by
bench.py itself on every invocation): the oracle cannot build these in test time, so parity is asserted through properties
the domain offers --
  * construct -> self-query round trip: every indexed k-mer is found and the IDs are a PERMUTATION of [0, N)
  * absent k-mers (uniform random 31-mers) are absent; membership == (ID >= 0)
  * strand symmetry: the IDs of a reverse-complemented read are the forward IDs reversed
  * sharding: the union of S shard answers equals the unsharded answers
  * a bounded sample is compared element-wise with the CPU oracle's query path running on the exported index image
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    import torch
    import blight_amd as B
    import bench
    k, m, n, b = 31, 9, 7, 6
    dev = torch.device("cuda", 0)
    ascii_, offs, total = bench.synth_unitigs(torch, dev, 230_000_000, k, 3)
    torch.cuda.synchronize()
    ix = B.kmer_Set_Light(k, m, n, 6, 1, b)
    ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
    ix.set_tuning("absent_filter", 1)  # wait for the filter's background passes: every query below goes through the armed filter
    return dict(torch=torch, B=B, bench=bench, dev=dev, ascii=ascii_, offs=offs, total=total, ix=ix, k=k, m=m, n=n, b=b)


def _all_kmer_positions(torch, offs, k):
    lens = offs[1:] - offs[:-1]
    nk = lens - (k - 1)
    starts = torch.repeat_interleave(offs[:-1] - (torch.cumsum(nk, 0) - nk), nk)  # position - running k-mer index
    return starts + torch.arange(int(nk.sum().item()), device=offs.device, dtype=torch.int64)


def test_self_query_ids_are_a_permutation(big):
    torch, ix, dev, k = big["torch"], big["ix"], big["dev"], big["k"]
    st = ix.stats()
    N = st["number_kmer"]
    assert N == 230_000_000
    pos = _all_kmer_positions(torch, big["offs"], k)
    assert pos.numel() == N
    s = torch.cuda.current_stream().cuda_stream
    kmers = torch.empty(N, device=dev, dtype=torch.int64)
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    ix.kmers_at_device(big["ascii"].data_ptr(), pos.data_ptr(), N, kmers.data_ptr(), err.data_ptr(), s)
    ids = torch.empty(N, device=dev, dtype=torch.int64)
    ix.query_kmers_device(kmers.data_ptr(), N, ids.data_ptr(), s)
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    assert int(ids.min().item()) >= 0 and int(ids.max().item()) == N - 1
    # random 31-mers over 2.3e8 positions collide with probability ~N^2/2/4^31 (< 1e-2 expected pairs): duplicates would
    # share an ID and consume a rank, so allow the count of distinct IDs to fall short by the number of duplicate k-mers
    counts = torch.bincount(ids, minlength=N)
    dup_ids = int((counts > 1).sum().item())
    missing = int((counts == 0).sum().item())
    assert missing == dup_ids and dup_ids <= 2  # each duplicated k-mer: one ID seen twice, one rank consumed and unseen
    if dup_ids == 0:
        assert bool((counts == 1).all().item())
    big["kmers"], big["ids"] = kmers, ids


def test_random_kmers_absent_and_membership(big):
    torch, ix, dev = big["torch"], big["ix"], big["dev"]
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    nq = 20_000_000
    q = torch.randint(-(1 << 63), (1 << 63) - 1, (nq,), device=dev, dtype=torch.int64, generator=g)
    s = torch.cuda.current_stream().cuda_stream
    ix.canonicalize_device(q.data_ptr(), nq, s)
    ids = torch.empty(nq, device=dev, dtype=torch.int64)
    ix.query_kmers_device(q.data_ptr(), nq, ids.data_ptr(), s)
    torch.cuda.synchronize()
    assert int((ids >= 0).sum().item()) <= 2  # P(hit) ~ 2.3e8 / 4^31 * 2 per query
    # host-buffer membership entry point agrees
    sample = q[:100000].cpu().numpy().view(np.uint64)
    assert np.array_equal(ix.query_kmer_bool(sample), ids[:100000].cpu().numpy() >= 0)


def test_strand_symmetry_and_oracle_sample(big):
    torch, ix, bench, k = big["torch"], big["ix"], big["bench"], big["k"]
    offs = big["offs"][:2001].cpu().numpy()
    raw = bytes(big["ascii"][: int(offs[-1])].cpu().numpy())
    seqs = [raw[offs[i]:offs[i + 1]].decode() for i in range(2000)]
    fwd = ix.query_sequences_hash(seqs)
    comp = str.maketrans("ACGT", "TGCA")
    rc = ix.query_sequences_hash([s[::-1].translate(comp) for s in seqs])
    o = 0
    for s_ in seqs:
        nk = len(s_) - k + 1
        assert np.array_equal(rc[o:o + nk], fwd[o:o + nk][::-1])
        o += nk
    assert (fwd >= 0).all()
    # element-wise against the CPU oracle's query path on the exported index image
    from oracle import oracle as O
    st = ix.stats()
    arrays = bench.export_index(ix, st, k, big["b"])
    orc = O.ImportedOracle(k, big["m"], big["n"], big["b"], st["total_nt"], st["number_kmer"], st["positions_total_size"], arrays)
    want = np.concatenate([orc.query_sequence(s_) for s_ in seqs[:300]])
    assert np.array_equal(fwd[: want.size], want)
    rng = np.random.default_rng(1)
    neg = rng.integers(0, 1 << 62, size=200000, dtype=np.uint64)
    neg = np.array([min(int(x), O.p_rc(int(x), k)) for x in neg[:20000]], dtype=np.uint64)
    assert np.array_equal(ix.query_kmer_hash(neg), orc.query_kmers(neg))
    # k-mers read from the packed sequence ITSELF at random places and at its very end -- junctions of super-k-mers and buckets and
    # the zero padding included: what the reference's scan can find although it is not a key must survive the absent-key filter
    words = arrays["seq"]
    total_nt = st["total_nt"]
    starts = np.concatenate([rng.integers(0, total_nt - 4 * k, size=3000), [total_nt - 2 * k]])
    readable = []
    for a0 in starts.tolist():
        w = words[a0 // 32: a0 // 32 + 8]
        nts = ((w[:, None] >> np.arange(62, -2, -2, dtype=np.uint64)[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
        nts = np.concatenate([nts, np.zeros(4 * k, np.uint8)])[a0 % 32: a0 % 32 + 4 * k]
        readable.append("".join("ACTG"[c] for c in nts.tolist()))
    got = ix.query_sequences_hash(readable)
    want = np.concatenate([orc.query_sequence(r) for r in readable])
    assert np.array_equal(got, want)
    assert (want >= 0).sum() > 0.15 * want.size and (want < 0).sum() > 1000  # keys inside super-k-mers (11 of 41 positions), absent k-mers across junctions


def test_shards_union_equals_unsharded(big):
    torch, B, ix, dev = big["torch"], big["B"], big["ix"], big["dev"]
    if "kmers" not in big:
        pytest.skip("needs the self-query test")
    world = 4
    nq = 30_000_000
    q = big["kmers"][:nq]
    want = big["ids"][:nq]
    s = torch.cuda.current_stream().cuda_stream
    got = torch.full((nq,), -1, device=dev, dtype=torch.int64)
    local = 0
    for r in range(world):
        sh = B.kmer_Set_Light(big["k"], big["m"], big["n"], 6, 1, big["b"])
        sh.set_shard(r, world)
        sh.construct_index_from_device(big["ascii"].data_ptr(), big["offs"].data_ptr(), big["offs"].numel() - 1, big["total"])
        sh.set_tuning("absent_filter", 1)
        local += sh.stats()["local_number_kmer"]
        part = torch.empty(nq, device=dev, dtype=torch.int64)
        sh.query_kmers_device(q.data_ptr(), nq, part.data_ptr(), s)
        torch.cuda.synchronize()
        assert int(((part >= 0) & (got >= 0)).sum().item()) == 0  # owners are disjoint
        got = torch.maximum(got, part)
        sh.close()
    assert local == ix.number_kmer
    assert torch.equal(got, want)


def test_streaming_file_query_multi_batch(tmp_path):
    """blmx_query_file on a 130 MB FASTA (several pinned batches in flight): the recap counters must equal the counts
    computed from the known composition of the file (indexed records, foreign records, records shorter than k, blank lines)."""
    import blight_amd as B
    k = 31
    rng = np.random.default_rng(12)
    n_idx, n_other, L = 300_000, 900_000, 100
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = letters[rng.integers(0, 4, size=(n_idx + n_other, L), dtype=np.uint8)]
    fa = tmp_path / "big.fa"
    with open(fa, "wb") as f:
        rows = np.empty((n_idx + n_other, 3 + L + 1), dtype=np.uint8)  # ">x\n" + seq + "\n"
        rows[:, 0] = ord(">")
        rows[:, 1] = ord("u")
        rows[:, 2] = ord("\n")
        rows[:, 3:3 + L] = seqs
        rows[:, 3 + L] = ord("\n")
        perm = rng.permutation(n_idx + n_other)
        f.write(rows[perm].tobytes())
        f.write(b">short\nACGTACGT\n\n\n>tail-without-newline\n" + bytes(seqs[0]))  # short record, blank header, no final newline
    ix = B.kmer_Set_Light(k, 9, 7, 6, 1, 6)
    ix.construct_index_from_sequences([bytes(s) for s in seqs[:n_idx]])
    per = L - k + 1
    assert ix.number_kmer == n_idx * per
    good, bad = ix.file_query(str(fa))
    assert good == (n_idx + 1) * per  # + the tail record, a copy of an indexed sequence
    assert bad == n_other * per
    assert ix.number_query == good + bad
    # the same 130 MB through the streaming INGEST (several parse windows and pinned chunks, parallel bulk parse, a last
    # line without newline): every record becomes part of the index, so a self-query finds everything
    fb = tmp_path / "big2.fa"
    with open(fb, "wb") as f:
        f.write(rows[perm].tobytes())
        f.write(b">tail-without-newline\n" + bytes(seqs[0]))
    ix2 = B.kmer_Set_Light(k, 9, 7, 6, 1, 6)
    ix2.construct_index(str(fb))
    assert ix2.number_kmer == (n_idx + n_other + 1) * per
    good2, bad2 = ix2.file_query(str(fb))
    assert (good2, bad2) == ((n_idx + n_other + 1) * per, 0)
    sample = [bytes(s_) for s_ in seqs[n_idx:n_idx + 50]]
    assert (ix2.query_sequences_hash(sample) >= 0).all() and (ix.query_sequences_hash(sample) < 0).all()


# ---- k = 63 (128-bit k-mers, b = 8) at BASELINE scale: C5 = 5e8 63-mers --------------------------------------------------------
@pytest.fixture(scope="module")
def big63():
    import torch
    import blight_amd as B
    import bench
    k, m, n, b = 63, 9, 14, 8
    dev = torch.device("cuda", 0)
    ascii_, offs, total = bench.synth_unitigs(torch, dev, 500_000_000, k, 5)
    torch.cuda.synchronize()
    ix = B.kmer_Set_Light(k, m, n, 6, 1, b)
    ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
    ix.set_tuning("absent_filter", 1)  # wait for the filter's background passes: every query below goes through the armed filter
    return dict(torch=torch, B=B, bench=bench, dev=dev, ascii=ascii_, offs=offs, total=total, ix=ix, k=k, m=m, n=n, b=b)


def test_k63_self_query_negatives_and_fallback(big63):
    """Every indexed 63-mer is found, the IDs are a permutation of [0, N) up to duplicated k-mers, random 63-mers are absent,
    and the fallback table -- populated for real at this scale by k-mers that agree in their low 64 bits (SURVEY C-7: the MPHF
    hashes 128-bit keys by their low half, bbhash.h:160-170) -- answers like the oracle on the exported index image."""
    torch, ix, dev, k = big63["torch"], big63["ix"], big63["dev"], big63["k"]
    st = ix.stats()
    N = st["number_kmer"]
    assert N == 500_000_000 and st["kmer_bytes"] == 16
    assert st["fallback_keys"] > 0
    pos = _all_kmer_positions(torch, big63["offs"], k)
    s = torch.cuda.current_stream().cuda_stream
    kmers = torch.empty((N, 2), device=dev, dtype=torch.int64)
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    ix.kmers_at_device(big63["ascii"].data_ptr(), pos.data_ptr(), N, kmers.data_ptr(), err.data_ptr(), s)
    del pos
    ids = torch.empty(N, device=dev, dtype=torch.int64)
    ix.query_kmers_device(kmers.data_ptr(), N, ids.data_ptr(), s)
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    assert int(ids.min().item()) >= 0 and int(ids.max().item()) == N - 1
    counts = torch.bincount(ids, minlength=N)
    dup_ids = int((counts > 1).sum().item())
    assert int((counts == 0).sum().item()) == dup_ids and dup_ids <= 2
    del counts
    # random 63-mers: absent; membership == (ID >= 0)
    g = torch.Generator(device=dev)
    g.manual_seed(63)
    nq = 20_000_000
    q = torch.randint(-(1 << 63), (1 << 63) - 1, (nq, 2), device=dev, dtype=torch.int64, generator=g)
    ix.canonicalize_device(q.data_ptr(), nq, s)
    nids = torch.empty(nq, device=dev, dtype=torch.int64)
    ix.query_kmers_device(q.data_ptr(), nq, nids.data_ptr(), s)
    torch.cuda.synchronize()
    assert int((nids >= 0).sum().item()) == 0
    sample = q[:100000].cpu().numpy().view(np.uint64)
    assert not ix.query_kmer_bool(sample.reshape(-1)).any()
    # the oracle's query path on the exported image: indexed sample, fallback keys, near misses of fallback keys
    from oracle import oracle as O
    orc, _ = big63["bench"].import_oracle(ix, st, k, big63["m"], big63["n"], big63["b"])
    hs = kmers[:300000].cpu().numpy().view(np.uint64)
    assert np.array_equal(ids[:300000].cpu().numpy(), orc.query_kmers(hs))
    fb = ix.export("fb_keys").reshape(-1, 2)
    got = ix.query_kmer_hash(fb.reshape(-1))
    assert np.array_equal(got, orc.query_kmers(fb)) and (got >= 0).all()
    flipped = fb.copy()
    flipped[:, 1] ^= np.uint64(1)  # same low 64 bits (same MPHF hashes), another k-mer: must be absent or answered like the oracle
    ints = [min(x, O.p_rc(x, k, 16)) for x in O.array_to_kmers(flipped, 16)]
    fa = O.kmers_to_array(ints, 16)
    assert np.array_equal(ix.query_kmer_hash(fa.reshape(-1)), orc.query_kmers(fa))
    orc.close()
    big63["kmers"], big63["ids"] = kmers, ids


def test_k63_strand_symmetry_and_reads(big63):
    torch, ix, k = big63["torch"], big63["ix"], big63["k"]
    offs = big63["offs"][:1501].cpu().numpy()
    raw = bytes(big63["ascii"][: int(offs[-1])].cpu().numpy())
    seqs = [raw[offs[i]:offs[i + 1]].decode() for i in range(1500)]
    fwd = ix.query_sequences_hash(seqs)
    comp = str.maketrans("ACGT", "TGCA")
    rc = ix.query_sequences_hash([s[::-1].translate(comp) for s in seqs])
    o = 0
    for s_ in seqs:
        nk = len(s_) - k + 1
        assert np.array_equal(rc[o:o + nk], fwd[o:o + nk][::-1])
        o += nk
    assert (fwd >= 0).all()
    if "ids" in big63:  # the sequence front end and the k-mer batch path agree
        assert np.array_equal(fwd, big63["ids"][: fwd.size].cpu().numpy())


def test_more_than_2_pow_32_kmers():
    """An index of 4.4e9 31-mers on one GPU: k-mer counts and IDs are 64-bit throughout (32-bit quantities are per MPHF group and
    per batch).  Indexed k-mers are found with IDs in [0, N) -- some beyond 2^32 --, random ones are absent, the host-buffer path
    (which carries such IDs as 64-bit words over PCIe) agrees with the device path, and a sample equals the CPU oracle's query
    path on the exported image."""
    import torch
    import blight_amd as B
    import bench
    k, m, n, b = 31, 9, 14, 6
    N = 4_400_000_000
    dev = torch.device("cuda", 0)
    ascii_, offs, total = bench.synth_unitigs(torch, dev, N, k, 9)
    torch.cuda.synchronize()
    ix = B.kmer_Set_Light(k, m, n, 6, 1, b)
    ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
    ix.set_tuning("absent_filter", 1)  # wait for the filter's background passes: every query below goes through the armed filter
    st = ix.stats()
    N = st["number_kmer"]  # (the generator trims to whole unitigs)
    assert abs(N - 4_400_000_000) < 1_000_000 and N > (1 << 32)
    nq = 20_000_000
    q, is_pos = bench.synth_queries(torch, ix, dev, ascii_, offs, total, k, nq, 91, 1, 0.5)
    del ascii_
    ids = torch.empty(nq, device=dev, dtype=torch.int64)
    ix.query_kmers_device(q.data_ptr(), nq, ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    found = ids >= 0
    assert bool((found == is_pos).all().item())
    assert int(ids.max().item()) < N and int(ids.max().item()) >= (1 << 32)
    h_q = q[:3_000_000].cpu().numpy().view(np.uint64).reshape(-1)
    h_ids = ids[:3_000_000].cpu().numpy()
    assert np.array_equal(ix.query_kmer_hash(h_q), h_ids)
    assert np.array_equal(ix.query_kmer_bool(h_q[:100000]), h_ids[:100000] >= 0)
    orc, _ = bench.import_oracle(ix, st, k, m, n, b)
    assert np.array_equal(orc.query_kmers(h_q[:400000], threads=8), h_ids[:400000])
    orc.close()
    ix.close()


def test_c2_ecoli_shape_few_long_unitigs():
    """BASELINE config 2 (E. coli K-12 BCALM2 unitigs, k = 31, ~4.6 M k-mers, query self): few LONG unitigs -- 9,200 x 530 nt, SURVEY 8d --
    and an index that fits the Infinity Cache.  Self-query of all 4.6 M k-mers is a permutation of [0, N); random 31-mers are absent;
    a 1e5-query sample (indexed, random, near misses) equals the CPU oracle's answers on the exported index image, and the oracle
    built from the same unitigs gives the same IDs for whole unitigs (the oracle can build this size)."""
    import torch
    import blight_amd as B
    import bench
    from oracle import oracle as O
    k, m, n, b = 31, 9, 7, 6
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    nu, L = 9200, 530
    c = torch.randint(0, 4, (nu * L,), device=dev, dtype=torch.uint8, generator=g)
    ascii_ = (65 + 2 * c + 2 * (c == 2).to(torch.uint8) + 13 * (c == 3).to(torch.uint8)).contiguous()
    offs = (torch.arange(nu + 1, device=dev, dtype=torch.int64) * L).contiguous()
    ix = B.kmer_Set_Light(k, m, n, 6, 1, b)
    ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), nu, nu * L)
    ix.set_tuning("absent_filter", 1)
    st = ix.stats()
    N = nu * (L - k + 1)
    assert st["number_kmer"] == N == 4_600_000
    s = torch.cuda.current_stream().cuda_stream
    pos = _all_kmer_positions(torch, offs, k)
    kmers = torch.empty(N, device=dev, dtype=torch.int64)
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    ix.kmers_at_device(ascii_.data_ptr(), pos.data_ptr(), N, kmers.data_ptr(), err.data_ptr(), s)
    ids = torch.empty(N, device=dev, dtype=torch.int64)
    for policy in (0, 1, -1):  # automatic, partitioned pipeline (exact pass, then the one-pass partition), direct kernel
        ix.set_query_binning(policy)
        for _ in range(2 if policy == 1 else 1):
            ix.query_kmers_device(kmers.data_ptr(), N, ids.data_ptr(), s)
            torch.cuda.synchronize()
            assert int(err.item()) == 0
            counts = torch.bincount(ids.clamp(min=0), minlength=N)
            assert int(ids.min().item()) >= 0 and bool((counts == 1).all().item())  # a permutation of [0, N)
    ix.set_query_binning(0)
    # the oracle on the exported image: a 1e5 sample of indexed k-mers, random k-mers and near misses
    arrays = bench.export_index(ix, st, k, b)
    orc = O.ImportedOracle(k, m, n, b, st["total_nt"], st["number_kmer"], st["positions_total_size"], arrays)
    rng = np.random.default_rng(22)
    hk = kmers.cpu().numpy().view(np.uint64)
    sample = hk[rng.integers(0, N, 60000)]
    neg = rng.integers(0, 1 << 62, size=20000, dtype=np.uint64)
    neg = np.array([min(int(x), O.p_rc(int(x), k)) for x in neg], dtype=np.uint64)
    near = sample[:20000] ^ (np.uint64(1) << rng.integers(0, 62, 20000).astype(np.uint64))
    near = np.array([min(int(x), O.p_rc(int(x), k)) for x in near], dtype=np.uint64)
    q = np.concatenate([sample, neg, near])
    want = orc.query_kmers(q)
    assert np.array_equal(ix.query_kmer_hash(q), want)
    assert (want[:60000] >= 0).all() and (want[60000:80000] < 0).all()
    # the oracle BUILT from the same unitigs (not the exported image): whole unitigs through the sequence front end
    raw = bytes(ascii_.cpu().numpy())
    seqs = [raw[i * L:(i + 1) * L].decode() for i in range(nu)]
    built = O.Oracle(k, m, n, b, seqs)
    assert built.stats["number_kmer"] == N
    some = seqs[::460]
    assert np.array_equal(ix.query_sequences_hash(some), np.concatenate([built.query_sequence(x) for x in some]))
    for key in ("g_nkeys", "g_prefix", "bucket_nt", "bucket_start", "seq", "positions"):
        assert np.array_equal(ix.export(key), built.arrays()[key]), key
