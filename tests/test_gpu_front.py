"""GPU parity of the sequence / read / file front end: query_sequence_hash, query_sequence_bool, file_query and
the host-buffer batch entry points, through every path the library can take (one fused launch on a pinned slot, windows
with the front end fused into the query kernel, windows feeding the binned pipeline, raw FASTA windows parsed on the
device), against the CPU oracle.  Bit-exact: integer IDs and counters."""
import ctypes as C
import gzip
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import blight_amd
    return blight_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), n))


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def _index(B, O, k, m, n, b, kb, seed, nseq=1500):
    rng = np.random.default_rng(seed)
    seqs = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 80))) for _ in range(nseq)]
    seqs += [_rand_seq(rng, k + int(rng.integers(0, 150)), "AC") for _ in range(40)]
    ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
    ix.construct_index_from_sequences(seqs)
    return ix, O.Oracle(k, m, n, b, seqs, kmer_bytes=kb), seqs, rng


def _reads(rng, seqs, k, n):
    """reads cut from the indexed sequences (either strand, either case), mutated ones, foreign ones, short and empty ones"""
    out = []
    for i in range(n):
        r = int(rng.integers(0, 8))
        s = seqs[int(rng.integers(0, len(seqs)))]
        if r <= 3:
            a = int(rng.integers(0, max(1, len(s) - k)))
            q = s[a:a + int(rng.integers(k, 151))]
            if r == 1:
                q = _revcomp(q)
            if r == 2:
                q = q.lower()
            if r == 3 and len(q) > 3:
                p = int(rng.integers(0, len(q)))
                q = q[:p] + "ACGT"[("ACGT".index(q[p].upper()) + 1) % 4] + q[p + 1:]
        elif r <= 5:
            q = _rand_seq(rng, int(rng.integers(k, 200)))
        elif r == 6:
            q = _rand_seq(rng, int(rng.integers(0, k)))  # shorter than k (or empty): contributes nothing
        else:
            q = s[:k]
        out.append(q)
    return out


def _want(orc, qs, k):
    parts = [orc.query_sequence(q) for q in qs if len(q) >= k]
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


@pytest.mark.parametrize("k,m,n,b,kb", [(31, 9, 7, 6, 8), (63, 9, 7, 8, 16), (21, 5, 3, 3, 8), (31, 9, 7, 6, 16)])
def test_sequence_queries_every_path(B, O, k, m, n, b, kb):
    ix, orc, seqs, rng = _index(B, O, k, m, n, b, kb, 1000 + k)
    # one call per read: ONE fused launch on a pinned slot (what the reference's callers do)
    for q in _reads(rng, seqs, k, 60):
        assert np.array_equal(ix.query_sequence_hash(q), _want(orc, [q], k)), q
        found, absent = ix.query_sequence_bool(q)
        w = _want(orc, [q], k)
        assert (found, absent) == (int((w >= 0).sum()), int((w < 0).sum()))
    few = _reads(rng, seqs, k, 40)
    assert np.array_equal(ix.query_sequences_hash(few), _want(orc, few, k))
    # many reads: windows; every combination of fused / binned and window sizes (windows split inside sequences)
    many = _reads(rng, seqs, k, 3000) + [seqs[3] * 1, "", _rand_seq(rng, 70000)]
    want = _want(orc, many, k)
    for mode in (0, 1, 2):
        for wkb in (0, 64, 1):
            ix.set_tuning("reads_mode", mode)
            ix.set_tuning("window_kb", wkb)
            ix.set_tuning("narrow_ids", 0 if (mode, wkb) == (1, 64) else 1)
            got = ix.query_sequences_hash(many)
            assert np.array_equal(got, want), (mode, wkb)
    ix.set_tuning("reads_mode", 0)
    ix.set_tuning("window_kb", 0)
    assert ix.number_query > 0
    ix.close()


def test_sequence_offsets_not_starting_at_zero(B, O):
    """blmx_query_sequences with offsets[0] > 0 (the characters before it belong to nobody), small and windowed"""
    k = 31
    ix, orc, seqs, rng = _index(B, O, k, 9, 7, 6, 8, 77, nseq=400)
    for nreads, wkb in ((5, 0), (1500, 16)):
        qs = _reads(rng, seqs, k, nreads)
        junk = _rand_seq(rng, 123)
        buf = (junk + "".join(qs)).encode()
        offs = np.zeros(len(qs) + 1, dtype=np.uint64)
        offs[0] = len(junk)
        offs[1:] = len(junk) + np.cumsum([len(q) for q in qs], dtype=np.uint64)
        want = _want(orc, qs, k)
        out = np.full(want.size, -7, dtype=np.int64)
        ix.set_tuning("window_kb", wkb)
        rc = ix._lib.blmx_query_sequences(ix._h, buf, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(qs), out.ctypes.data_as(C.POINTER(C.c_int64)))
        assert rc == 0
        assert np.array_equal(out, want)


def test_invalid_characters(B, O):
    """Invalid char in DNA (kmer.h:68) only where the reference would look: inside a sequence of at least k characters"""
    k = 31
    ix, orc, seqs, rng = _index(B, O, k, 9, 7, 6, 8, 5, nseq=300)
    good = _reads(rng, seqs, k, 30)
    assert ix.query_sequences_hash(good + ["ACGTN"]).size == _want(orc, good, k).size  # too short: never encoded
    with pytest.raises(ValueError, match="Invalid char in DNA"):
        ix.query_sequence_hash(seqs[0][:40] + "N" + seqs[0][41:])
    big = _reads(rng, seqs, k, 2500)
    big[1700] = big[1700][:20] + "X" + _rand_seq(rng, 40)
    for mode in (1, 2):
        ix.set_tuning("reads_mode", mode)
        with pytest.raises(ValueError, match="Invalid char in DNA"):
            ix.query_sequences_hash(big)
    ix.set_tuning("reads_mode", 0)
    big[1700] = "NNNN"
    assert np.array_equal(ix.query_sequences_hash(big), _want(orc, big, k))


def test_concurrent_callers(B, O):
    """query_sequence_hash / query_kmer_hash from 8 threads at once (the reference's callers query from OpenMP threads,
    Colored_De_Bruijn_graph_snippet.cpp:122-157); number_query counts every k-mer exactly once (blight.cpp:687-688)."""
    k = 31
    ix, orc, seqs, rng = _index(B, O, k, 9, 7, 6, 8, 9, nseq=600)
    reads = _reads(rng, seqs, k, 1600)
    wants = [_want(orc, [q], k) for q in reads]
    kmers = [orc.canonical_kmers(q) if len(q) >= k else np.zeros(0, np.uint64) for q in reads]
    before = ix.number_query
    errors = []

    def worker(t):
        try:
            for i in range(t, len(reads), 8):
                got = ix.query_sequence_hash(reads[i])
                if not np.array_equal(got, wants[i]):
                    errors.append(("seq", i))
                if kmers[i].size and i % 5 == 0:
                    if not np.array_equal(ix.query_kmer_hash(kmers[i]), wants[i]):
                        errors.append(("kmers", i))
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[:5]
    expect = sum(w.size for w in wants) + sum(wants[i].size for i in range(len(reads)) if kmers[i].size and i % 5 == 0)
    assert ix.number_query - before == expect


def test_concurrent_large_batches_on_streams(B, O):
    """Large device-resident batches (the partitioned pipeline: shared scratch, learnt bin regions, filter gate) from four threads, each
    on its own stream, at the same time as sequence calls from the host API: every batch gets its own answers."""
    import torch
    k = 31
    ix, orc, seqs, rng = _index(B, O, k, 9, 7, 6, 8, 19, nseq=1500)
    ix.set_tuning("absent_filter", 1)
    ix.set_query_binning(1)
    pos = np.concatenate([orc.canonical_kmers(s) for s in seqs])
    dev = torch.device("cuda", 0)
    batches, wants = [], []
    for t in range(4):
        neg = np.array([min(int(x), O.p_rc(int(x), k)) for x in rng.integers(0, 1 << 62, size=20000, dtype=np.uint64)], dtype=np.uint64)
        q = np.concatenate([pos[rng.permutation(len(pos))[: 30000 + 7919 * t]], neg])
        q = q[rng.permutation(len(q))]
        batches.append(torch.from_numpy(q.view(np.int64)).to(dev))
        wants.append(orc.query_kmers(q))
    reads = _reads(rng, seqs, k, 200)
    want_reads = [_want(orc, [r], k) for r in reads]
    errors = []

    def device_worker(t):
        try:
            st = torch.cuda.Stream()
            out = torch.empty(batches[t].numel(), dtype=torch.int64, device=dev)
            for rep in range(12):
                out.fill_(-7)
                st.wait_stream(torch.cuda.current_stream())
                ix.query_kmers_device(batches[t].data_ptr(), batches[t].numel(), out.data_ptr(), st.cuda_stream)
                st.synchronize()
                if not np.array_equal(out.cpu().numpy(), wants[t]):
                    errors.append(("device", t, rep))
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    def host_worker():
        try:
            for i, r in enumerate(reads):
                if not np.array_equal(ix.query_sequence_hash(r), want_reads[i]):
                    errors.append(("host", i))
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    th = [threading.Thread(target=device_worker, args=(t,)) for t in range(4)] + [threading.Thread(target=host_worker)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[:5]


def _file_counts(O, orc, path, k):
    good = bad = 0
    for s in O.read_fasta(str(path)):
        if len(s) >= k:
            ids = orc.query_sequence(s)
            good += int((ids >= 0).sum())
            bad += int((ids < 0).sum())
    return good, bad


def test_file_query_raw_windows(B, O, tmp_path):
    """file_query on files that exercise the reference reader's rule (blight.cpp:760-775: lines are consumed in pairs, the
    second of a pair is a sequence iff both are non-empty; shorter than k: skipped), cut into windows of every size"""
    k = 31
    ix, orc, seqs, rng = _index(B, O, k, 9, 7, 6, 8, 31, nseq=500)
    reads = _reads(rng, seqs, k, 1200)
    lines = []
    for i, q in enumerate(reads):
        lines.append(">r%d some description" % i)
        lines.append(q)
        if i % 37 == 5:
            lines.append("")            # empty header: swallows the next line
        if i % 53 == 7:
            lines += [">lonely", ""]    # empty sequence
        if i % 101 == 3:
            lines += ["", ""]
    contents = {
        "regular.fa": "".join(">%d\n%s\n" % (i, q) for i, q in enumerate(reads) if q),
        "quirks.fa": "\n".join(lines) + "\n",
        "noeol.fa": "\n".join(lines),                      # last line without a newline
        "header_last.fa": "\n".join(lines) + "\n>orphan",  # a header with no sequence line
        "blank_start.fa": "\n" + "\n".join(lines) + "\n",  # shifts every pair by one line
        "junk_in_skipped.fa": ">a\nNNNN\n>b\n%s\n\nXYZ-not-DNA-but-swallowed\n>c\n%s\n" % (reads[0], reads[1]),
    }
    for name, text in contents.items():
        p = tmp_path / name
        p.write_text(text)
        gz = tmp_path / (name + ".gz")
        with gzip.open(gz, "wb") as f:
            f.write(text.encode())
        want = _file_counts(O, orc, p, k)
        for wkb in (0, 16, 1):
            for mode in (1, 2):
                ix.set_tuning("window_kb", wkb)
                ix.set_tuning("reads_mode", mode)
                assert ix.file_query(str(p)) == want, (name, wkb, mode)
            assert ix.file_query(str(gz)) == want, (name, "gz", wkb)
    ix.set_tuning("window_kb", 0)
    ix.set_tuning("reads_mode", 0)
    bad = tmp_path / "bad.fa"
    bad.write_text(">a\n%s\n>b\n%sN%s\n" % (reads[0], reads[1], reads[2]))
    with pytest.raises(ValueError, match="Invalid char in DNA"):
        ix.file_query(str(bad))


def test_file_long_line_after_short_records(B, O, tmp_path):
    """short records followed by one line of 20 MB (a chromosome behind a mitochondrion), plain and gzip, with windows
    smaller than the line: the window grows and keeps what it already holds"""
    k = 31
    rng = np.random.default_rng(2)
    codes = rng.integers(0, 4, 20_000_000, dtype=np.uint8)
    long_line = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes().decode()
    shorts = [_rand_seq(rng, int(rng.integers(40, 400))) for _ in range(200)]
    index_seqs = shorts[:100] + [long_line[i:i + 5000] for i in range(0, 2_000_000, 50_000)]
    ix = B.kmer_Set_Light(k, 9, 7, 3, 1, 6)
    ix.construct_index_from_sequences(index_seqs)
    orc = O.Oracle(k, 9, 7, 6, index_seqs)
    text = "".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(shorts)) + ">chr\n" + long_line + "\n>tail\n" + shorts[3] + "\n"
    p = tmp_path / "long.fa"
    p.write_text(text)
    gz = tmp_path / "long.fa.gz"
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(text.encode())
    want = _file_counts(O, orc, p, k)
    assert want[0] > 100_000 and want[1] > 1_000_000
    for wkb in (0, 1024):
        ix.set_tuning("window_kb", wkb)
        assert ix.file_query(str(p)) == want, wkb
        assert ix.file_query(str(gz)) == want, ("gz", wkb)
    # the same line through query_sequence_hash (windows split inside one sequence)
    ix.set_tuning("window_kb", 4096)
    ids = ix.query_sequence_hash(long_line)
    assert (int((ids >= 0).sum()), int((ids < 0).sum())) == _file_counts(O, orc, _one_record(tmp_path, long_line), k)
    assert np.array_equal(ids[:3_000_000], orc.query_sequence(long_line[:3_000_000 + k - 1]))


def _one_record(tmp_path, seq):
    p = tmp_path / "one.fa"
    p.write_text(">x\n" + seq + "\n")
    return p


@pytest.mark.parametrize("k,kb,b", [(31, 8, 6), (63, 16, 8)])
def test_host_batches_in_chunks(B, O, k, kb, b):
    """blmx_query_kmers / _membership / _with_minimizer / _positions on host arrays that span several pinned chunks"""
    ix, orc, seqs, rng = _index(B, O, k, 9, 7, b, kb, 400 + k, nseq=4000)
    pos = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(60000)]
    neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
    q = np.concatenate([pos, neg])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    want = orc.query_kmers(q.reshape(-1) if kb == 8 else q)
    ix.set_tuning("window_kb", 128)  # 2^16 k-mers per chunk
    assert len(q) > 5 * (1 << 16)
    for binning, narrow in ((-1, 1), (1, 1), (1, 0)):
        ix.set_query_binning(binning)
        ix.set_tuning("narrow_ids", narrow)  # IDs cross PCIe as 32-bit words (widened on the host) or as 64-bit words
        assert np.array_equal(ix.query_kmer_hash(q.reshape(-1)), want)
    assert np.array_equal(ix.query_kmer_bool(q.reshape(-1)), want >= 0)
    # caller-supplied minimizers: the right ones for the first 20000 k-mers, then the same values tiled (mostly wrong: defined too)
    sub = q[:300000]
    ints = O.array_to_kmers(sub[:20000], kb)
    minis = np.resize(np.array([O.p_minimizer(x, k, 9, kb) for x in ints], dtype=np.uint32), len(sub))
    wm = orc.query_kmers_mini(sub.reshape(-1) if kb == 8 else sub, minis)
    assert np.array_equal(ix.query_get_hash(sub.reshape(-1), minis), wm)
    wp = orc.query_kmers_mini(sub.reshape(-1) if kb == 8 else sub, minis, want_pos=True)
    assert np.array_equal(ix.query_get_pos_unitig(sub.reshape(-1), minis).astype(np.int64), wp)


@pytest.mark.parametrize("k,kb,b", [(31, 8, 6), (63, 16, 8)])
def test_tuning_knobs_do_not_change_results(B, O, k, kb, b):
    """code paths behind per-index knobs: full group descriptor instead of the compact query head, minimizer hand-off on / off,
    coarse and fine partitions"""
    ix, orc, seqs, rng = _index(B, O, k, 9, 7, b, kb, 900 + k, nseq=3000)
    pos = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(40000)]
    neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
    q = np.concatenate([pos, neg])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    want = orc.query_kmers(q.reshape(-1) if kb == 8 else q)
    reads = _reads(rng, seqs, k, 2000)
    want_reads = _want(orc, reads, k)
    ix.set_query_binning(1)
    for heads in (1, 0):
        for handoff in (-1, 0, 1):
            for bits, narrow in ((0, 1), (3, 0), (10, 1)):
                ix.set_tuning("query_heads", heads)
                ix.set_tuning("mini_handoff", handoff)
                ix.set_tuning("bin_bits", bits)
                ix.set_tuning("narrow_ids", narrow)   # ordered IDs as 32-bit or 64-bit words
                assert np.array_equal(ix.query_kmer_hash(q.reshape(-1)), want), (heads, handoff, bits, narrow)
        ix.set_tuning("narrow_ids", 1)
        ix.set_tuning("reads_mode", 2)
        assert np.array_equal(ix.query_sequences_hash(reads), want_reads)
        ix.set_tuning("reads_mode", 1)
        assert np.array_equal(ix.query_sequences_hash(reads), want_reads)
    with pytest.raises(ValueError):
        ix.set_tuning("no_such_knob", 1)


@pytest.mark.parametrize("k,kb,b", [(31, 8, 6), (63, 16, 8)])
def test_partition_paths_and_skewed_batches(B, O, k, kb, b):
    """The batch partition in front of the query kernel: one pass into bin regions sized after the previous batch (default after the
    first batch; 1 = always) and the exact partition
    give the direct kernel's answers -- also when nearly the whole batch falls into a few bins and spills into the overflow
    region (a few k-mers repeated a thousand times), and for batches that are not a multiple of anything."""
    ix, orc, seqs, rng = _index(B, O, k, 9, 7, b, kb, 1300 + k, nseq=2500)
    pos = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(30000)]
    neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
    uniform = np.concatenate([pos, neg])
    uniform = np.ascontiguousarray(uniform[rng.permutation(len(uniform))])
    few = uniform[:700]
    skewed = np.ascontiguousarray(np.concatenate([np.tile(few, (1500, 1)), uniform[:12345]])[rng.permutation(700 * 1500 + 12345)])
    one = np.ascontiguousarray(np.tile(uniform[5:6], (300001, 1)))
    for name, q in (("uniform", uniform), ("skewed", skewed), ("one k-mer", one)):
        want = orc.query_kmers(q.reshape(-1) if kb == 8 else q)
        ix.set_query_binning(-1)
        assert np.array_equal(ix.query_kmer_hash(q.reshape(-1)), want), name
        ix.set_query_binning(1)
        for part in (0, 1, 2):  # automatic (exact, then one pass with learnt regions) / always one pass / always exact
            for bits in (0, 4, 10):
                ix.set_tuning("partition", part)
                ix.set_tuning("bin_bits", bits)
                assert np.array_equal(ix.query_kmer_hash(q.reshape(-1)), want), (name, part, bits)
        # the one-pass kernel as one workgroup per CU that stages a whole tile (default: two, half a tile at a time)
        ix.set_tuning("partition", 1)
        ix.set_tuning("bin_bits", 0)
        ix.set_tuning("bin_rounds", 1)
        assert np.array_equal(ix.query_kmer_hash(q.reshape(-1)), want), (name, "bin_rounds=1")
        ix.set_tuning("bin_rounds", 2)
    ix.set_tuning("partition", 0)
    ix.set_tuning("bin_bits", 0)
    # automatic mode over a sequence of batches: regions learnt from a batch of the same distribution, then from a very different one
    wants = {name: orc.query_kmers(q.reshape(-1) if kb == 8 else q) for name, q in (("uniform", uniform), ("skewed", skewed), ("one k-mer", one))}
    batches = {"uniform": uniform, "skewed": skewed, "one k-mer": one}
    for name in ("skewed", "skewed", "uniform", "uniform", "one k-mer", "uniform", "skewed"):
        assert np.array_equal(ix.query_kmer_hash(batches[name].reshape(-1)), wants[name]), name
    half = uniform[: len(uniform) // 3]  # another batch size: the shares scale
    assert np.array_equal(ix.query_kmer_hash(half.reshape(-1)), wants["uniform"][: len(half)])


@pytest.mark.parametrize("board", ["pinned", "device"])
def test_serving_burst_answers_small_calls_and_leaves(B, O, board, monkeypatch):
    """Small calls -- one k-mer, a few k-mers, one read: the reference's calling pattern (blight.cpp:537-550, 575-591) -- are answered by
    a BOUNDED serving kernel that polls a request ring in pinned memory (blmx_serve.hip).  Every answer equals the oracle's (64- and
    128-bit k-mers, membership, reads with and without invalid characters, from eight threads at once); the kernel is gone within
    its idle bound once the calls stop, the next call simply starts the next one; with serve_burst = 0 the same calls take one launch
    per round and give the same answers.  `board`: where the kernel looks for posted requests -- a piece of the pinned ring (default), or
    device memory that the host writes through the PCIe BAR (BLIGHT_SERVE_BOARD=device; falls back to pinned memory by itself where
    the process cannot store into device memory)."""
    import time
    monkeypatch.setenv("BLIGHT_SERVE_BOARD", board)  # (read when an index starts its first serving kernel)
    for (k, kb, m, n, b) in ((31, 8, 9, 6, 6), (63, 16, 9, 5, 8), (21, 8, 7, 3, 2))[: 3 if board == "pinned" else 2]:
        ix, orc, seqs, rng = _index(B, O, k, m, n, b, kb, 23 + k, nseq=600)
        ix.set_tuning("absent_filter", 1)
        pos = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs[:40]])
        neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in (int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(300))], kb).reshape(-1, kb // 8)
        q = np.ascontiguousarray(np.concatenate([pos[:300], neg]))
        want = orc.query_kmers(q.reshape(-1) if kb == 8 else q)
        before = ix.serve_stats()
        got = np.array([ix.query_kmer_hash(np.ascontiguousarray(q[i]).reshape(-1))[0] for i in range(len(q))])
        assert np.array_equal(got, want)
        st = ix.serve_stats()
        assert st["calls"] - before["calls"] == len(q) and st["kernels"] >= 1
        # a few k-mers per call, membership
        for a in range(0, len(q), 37):
            part = np.ascontiguousarray(q[a:a + 37]).reshape(-1)
            assert np.array_equal(ix.query_kmer_hash(part), want[a:a + 37])
            assert np.array_equal(ix.query_kmer_bool(part), want[a:a + 37] >= 0)
        # reads: one call each
        reads = _reads(rng, seqs, k, 120)
        # ... and long ones: several k-mers per thread of the serving workgroup's two waves (a slot holds up to 1536 characters),
        # lengths around the multiples of 128 k-mers, a whole indexed sequence and its reverse complement
        long_src = "".join(seqs[:30])
        for L in (k + 127, k + 128, k + 129, k + 255, k + 256, 700, 1535, 1536):
            a = int(rng.integers(0, len(long_src) - L))
            reads.append(long_src[a:a + L])
        reads.append(_revcomp(long_src[:1200]))
        reads.append(_rand_seq(rng, 1500))
        for r in reads:
            assert np.array_equal(ix.query_sequence_hash(r), _want(orc, [r], k))
        with pytest.raises(Exception):
            ix.query_sequence_hash(reads[0][:k + 3] + "N" + reads[0][k + 4:])
        assert np.array_equal(ix.query_sequence_hash(reads[1]), _want(orc, [reads[1]], k))  # the ring is fine after a failed call
        # eight threads at once
        errors = []

        def worker(t):
            try:
                for i in range(t, len(q), 8):
                    if ix.query_kmer_hash(np.ascontiguousarray(q[i]).reshape(-1))[0] != want[i]:
                        errors.append(("kmer", i))
                for i in range(t, len(reads), 8):
                    if not np.array_equal(ix.query_sequence_hash(reads[i]), _want(orc, [reads[i]], k)):
                        errors.append(("read", i))
            except Exception as e:  # pragma: no cover
                errors.append(repr(e))
        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errors, errors[:5]
        # no traffic: the kernel leaves within its bound (60 us idle by default; the check allows for a loaded box)
        t0 = time.time()
        while ix.serve_stats()["alive"] and time.time() - t0 < 1.0:
            time.sleep(0.001)
        assert not ix.serve_stats()["alive"], "the serving kernel is still there %.3f s after the last call" % (time.time() - t0)
        kernels = ix.serve_stats()["kernels"]
        assert ix.query_kmer_hash(np.ascontiguousarray(q[0]).reshape(-1))[0] == want[0]  # the next call starts the next kernel
        assert ix.serve_stats()["kernels"] == kernels + 1
        # the same calls without the burst
        ix.set_tuning("serve_burst", 0)
        assert not ix.serve_stats()["alive"]
        calls = ix.serve_stats()["calls"]
        assert np.array_equal(np.array([ix.query_kmer_hash(np.ascontiguousarray(q[i]).reshape(-1))[0] for i in range(0, len(q), 5)]), want[::5])
        assert np.array_equal(ix.query_sequence_hash(reads[2]), _want(orc, [reads[2]], k))
        assert ix.serve_stats()["calls"] == calls
        ix.close()
