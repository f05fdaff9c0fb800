"""The native multi-GPU path (blmx_comm.hip: construction from slices, halo exchange, the query exchange with its piece pipeline and
multi-round all-to-all-v, failure votes) with MORE THAN ONE RANK on a one-GPU box: the library's loopback transport -- `world`
communicators in this process, one thread per rank, all on the test box's GPU; flows are device copies under the same grouping and
stream ordering as RCCL.  IDs must equal the single index's (and the oracle's) for every rank's batch."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

os.environ.setdefault("BLIGHT_LOOPBACK_TIMEOUT", "90")


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def run_ranks(world, fn):
    """fn(rank) on `world` threads; returns the results, re-raises the first exception (after every thread has ended)"""
    res, err = [None] * world, [None] * world

    def wrap(r):
        try:
            res[r] = fn(r)
        except BaseException as e:  # noqa: BLE001 - reported below
            err[r] = e

    th = [threading.Thread(target=wrap, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return res, err


@pytest.mark.parametrize("world,k,m,n,b", [(2, 31, 9, 7, 6), (3, 31, 7, 5, 4), (8, 31, 9, 9, 6), (3, 63, 9, 7, 8)])
def test_native_exchange_on_the_loopback_transport(world, k, m, n, b, tmp_path):
    import torch
    import blight_amd as B
    import shard_util as S
    from oracle import oracle as O
    rng = np.random.default_rng(world * 13 + k)
    seqs = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 80))) for _ in range(4000)]
    seqs += ["A" * (k + 40), seqs[5]]
    kb = 8 if k <= 32 else 16
    full = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
    full.construct_index_from_sequences(seqs)
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    # queries: every k-mer readable in the single index's packed sequence (false positives at the shard cuts included), the indexed
    # k-mers once more, random ones; shuffled
    rq, _ = S.readable_kmers(full, k, kb, b)
    pos = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs[:1500]])
    neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(30000)]
    neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
    q = np.concatenate([rq, pos, neg])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    want = orc.query_kmers(q.reshape(-1) if kb == 8 else q)
    assert np.array_equal(full.query_kmer_hash(q.reshape(-1)), want)
    # uneven slices of the input and of the query batch; with three or more ranks one rank gets no sequences and one no queries
    cuts = sorted(int(x) for x in rng.integers(0, len(seqs), world - 1))
    qcuts = sorted(int(x) for x in rng.integers(0, len(q), world - 1))
    if world >= 3:
        cuts[1] = cuts[0]
        qcuts[-1] = len(q)
    cuts = [0] + cuts + [len(seqs)]
    qcuts = [0] + qcuts + [len(q)]
    ref = []
    for r in range(world):
        a = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
        a.set_shard(r, world)
        a.construct_index_from_sequences(seqs)
        ref.append(a)
    dev = torch.device("cuda", 0)
    d_q = [torch.from_numpy(q[qcuts[r]:qcuts[r + 1]].view(np.int64).copy()).to(dev) for r in range(world)]
    d_ids = [torch.full((max(1, qcuts[r + 1] - qcuts[r]),), -7, dtype=torch.int64, device=dev) for r in range(world)]
    streams = [torch.cuda.Stream(dev) for _ in range(world)]
    torch.cuda.synchronize()
    uid = B.comm_loopback_id()
    comms, ixs = [None] * world, [None] * world

    def build(r):
        comms[r] = B.Comm(uid, r, world, device=0)
        assert comms[r].transport == "loopback"
        ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
        ix.set_shard(r, world)
        comms[r].build_sharded_from_sequences(ix, seqs[cuts[r]:cuts[r + 1]])
        ixs[r] = ix

    _, err = run_ranks(world, build)
    assert not any(err), err
    for r in range(world):
        for name in B.EXPORT:
            assert np.array_equal(ixs[r].export(name), ref[r].export(name)), (r, name)
        s1, s2 = ixs[r].stats(), ref[r].stats()
        assert not [f for f in s1 if f not in ("build_seconds", "number_query", "build_scratch_peak_bytes") and s1[f] != s2[f]], [(r, f, s1[f], s2[f]) for f in s1 if s1[f] != s2[f] and f not in ("build_seconds", "number_query", "build_scratch_peak_bytes")]
        ixs[r].set_tuning("absent_filter", 1)

    # ---- the query exchange: 1..4 pieces, whole flows and flows cut into many ragged rounds; pieces that travel densely (the exact
    # partition: the first batch of an index, small pieces) and as bin regions of the one-pass partition (slot_piece_min lowered so that
    # these small batches qualify; rank 0 keeps the default, i.e. sends densely while the others send regions: formats mix)
    # ship: do the minimizers of a slot range travel with its k-mers, or does the owner hash what it receives again (-1: by the size of
    # the world, the default)
    for pieces, flow_words, slot_min, ship in ((2, 1 << 25, 0, -1), (1, 997, 2000, 1), (3, 4099, 2000, 0), (4, 1 << 25, 2000, 1), (2, 1 << 25, 2000, 0),
                                               (2, 1 << 25, 2000, -1)):
        for r, c in enumerate(comms):
            c.set_tuning("pieces", pieces)
            c.set_tuning("flow_words", flow_words)
            c.set_tuning("ship_minimizers", ship)
            ixs[r].set_tuning("slot_piece_min", slot_min if (r > 0 or pieces == 2) else 0)

        def query(r):
            n_r = qcuts[r + 1] - qcuts[r]
            comms[r].query_kmers_sharded_device(ixs[r], d_q[r].data_ptr(), n_r, d_ids[r].data_ptr(), streams[r].cuda_stream)
            streams[r].synchronize()
            return d_ids[r][:n_r].cpu().numpy()

        got, err = run_ranks(world, query)
        assert not any(err), err
        for r in range(world):
            assert np.array_equal(got[r], want[qcuts[r]:qcuts[r + 1]]), (pieces, flow_words, r, int((got[r] != want[qcuts[r]:qcuts[r + 1]]).sum()))

    # ---- a batch unlike its predecessor (one k-mer, many times): the learnt regions overflow, that piece falls back to the exact
    # partition on its sender alone; the batch after it goes through the regions again
    nrep = 60000
    first = [min(qcuts[r], len(q) - 1) for r in range(world)]
    d_rep = [torch.from_numpy(np.tile(q[first[r]:first[r] + 1], (nrep, 1)).view(np.int64).copy()).to(dev) for r in range(world)]
    d_rep_ids = [torch.full((nrep,), -7, dtype=torch.int64, device=dev) for _ in range(world)]
    torch.cuda.synchronize()

    def repeated(r):
        comms[r].query_kmers_sharded_device(ixs[r], d_rep[r].data_ptr(), nrep, d_rep_ids[r].data_ptr(), streams[r].cuda_stream)
        streams[r].synchronize()
        return d_rep_ids[r].cpu().numpy()

    got, err = run_ranks(world, repeated)
    assert not any(err), err
    for r in range(world):
        assert (got[r] == want[first[r]]).all(), r
    got, err = run_ranks(world, query)
    assert not any(err), err
    for r in range(world):
        assert np.array_equal(got[r], want[qcuts[r]:qcuts[r + 1]]), r

    # ---- host-buffer collectives: k-mers, and whole sequences through this rank's read front end
    reads = seqs[:200] + [_rand_seq(rng, 200) for _ in range(40)] + ["ACG"]
    rwant = [orc.query_sequence(s) if len(s) >= k else np.zeros(0, np.int64) for s in reads]

    def host_calls(r):
        a = comms[r].query_kmers_sharded(ixs[r], q[qcuts[r]:qcuts[r + 1]].reshape(-1))
        mine = reads[r::world]
        b_ = comms[r].query_sequences_sharded(ixs[r], mine)
        return a, b_

    got, err = run_ranks(world, host_calls)
    assert not any(err), err
    for r in range(world):
        assert np.array_equal(got[r][0], want[qcuts[r]:qcuts[r + 1]]), r
        assert np.array_equal(got[r][1], np.concatenate(rwant[r::world])), r

    # ---- many small collectives back to back: ranks leave a collective at different times and enter the next one at once (the events
    # of a collective must outlive its slowest rank)
    small = [q[qcuts[r]:qcuts[r] + 40 + r] for r in range(world)]

    def many_small(r):
        out = []
        for it in range(60):
            out.append(comms[r].query_kmers_sharded(ixs[r], small[r].reshape(-1)))
        return out

    got, err = run_ranks(world, many_small)
    assert not any(err), err
    for r in range(world):
        for a in got[r]:
            assert np.array_equal(a, want[qcuts[r]:qcuts[r] + len(small[r])]), r

    # ---- construction from ONE FASTA file named on every rank (device parser; each rank keeps its consecutive part of the sequences)
    fa = tmp_path / "u.fa"
    fa.write_text("".join(">u%d\n%s\n" % (i, s_) for i, s_ in enumerate(seqs)))
    from_file = [None] * world

    def build_from_file(r):
        ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
        ix.set_shard(r, world)
        comms[r].build_sharded_from_fasta(ix, str(fa))
        from_file[r] = ix

    _, err = run_ranks(world, build_from_file)
    assert not any(err), err
    for r in range(world):
        for name in B.EXPORT:
            assert np.array_equal(from_file[r].export(name), ref[r].export(name)), (r, name)
        from_file[r].close()

    # ---- failures are collective: one rank's input is rejected / one rank's index is not built -> nobody hangs, everybody fails
    fresh = [B.kmer_Set_Light(k, m, n, min(n, 3), 1, b) for _ in range(world)]

    def bad_build(r):
        fresh[r].set_shard(r, world)
        comms[r].build_sharded_from_sequences(fresh[r], seqs[cuts[r]:cuts[r + 1]] + (["ACGT"] if r == world - 1 else []))

    _, err = run_ranks(world, bad_build)
    assert all(isinstance(e, B.BlightError) for e in err), err
    assert err[world - 1].code == B.BLMX_ESHORT and all(e.code == B.BLMX_ERCCL for e in err[:-1])

    def bad_query(r):
        ix = fresh[r] if r == 0 else ixs[r]     # rank 0 brings an index that was never built
        n_r = qcuts[r + 1] - qcuts[r]
        comms[r].query_kmers_sharded_device(ix, d_q[r].data_ptr(), n_r, d_ids[r].data_ptr(), streams[r].cuda_stream)

    _, err = run_ranks(world, bad_query)
    assert all(isinstance(e, B.BlightError) for e in err), err
    assert err[0].code == B.BLMX_ESTATE and all(e.code == B.BLMX_ERCCL for e in err[1:])

    # ... and the communicators are still usable afterwards
    def again(r):
        comms[r].build_sharded_from_sequences(fresh[r], seqs[cuts[r]:cuts[r + 1]])
        n_r = qcuts[r + 1] - qcuts[r]
        comms[r].query_kmers_sharded_device(fresh[r], d_q[r].data_ptr(), n_r, d_ids[r].data_ptr(), streams[r].cuda_stream)
        streams[r].synchronize()
        return d_ids[r][:n_r].cpu().numpy()

    got, err = run_ranks(world, again)
    assert not any(err), err
    for r in range(world):
        assert np.array_equal(got[r], want[qcuts[r]:qcuts[r + 1]]), r
    torch.cuda.synchronize()
    for c in comms:
        c.close()


def test_loopback_ranks_must_agree_on_the_world():
    import blight_amd as B
    uid = B.comm_loopback_id()
    a = B.Comm(uid, 0, 2)
    with pytest.raises(ValueError):
        B.Comm(uid, 1, 3)
    a.close()


def test_a_rank_that_never_arrives_kills_the_group_not_the_process(monkeypatch):
    """Every wait of the loopback transport has a deadline.  A rank that waits in vain (its peer threw before the collective, or never
    called it) gets BLMX_ERCCL -- and the group is dead from then on: what the ranks that did arrive had posted is stale, so every later
    collective on any communicator of the id fails AT ONCE instead of pulling a stale flow or waiting out another deadline."""
    import time
    import torch
    import blight_amd as B
    monkeypatch.setenv("BLIGHT_LOOPBACK_TIMEOUT", "1.5")
    rng = np.random.default_rng(3)
    k, m, n, b = 31, 9, 6, 6
    seqs = [_rand_seq(rng, 200) for _ in range(800)]
    shards = []
    for r in range(2):
        a = B.kmer_Set_Light(k, m, n, 3, 1, b)
        a.set_shard(r, 2)
        a.construct_index_from_sequences(seqs)
        shards.append(a)
    uid = B.comm_loopback_id()
    comms = [None, None]

    def mk(r):
        comms[r] = B.Comm(uid, r, 2, device=0)
    run_ranks(2, mk)
    dev = torch.device("cuda", 0)
    q = torch.randint(0, 1 << 60, (5000,), dtype=torch.int64, device=dev)
    out = torch.empty(5000, dtype=torch.int64, device=dev)
    t0 = time.time()
    with pytest.raises(Exception, match="did not arrive|gave up"):  # rank 1 never calls
        comms[0].query_kmers_sharded_device(shards[0], q.data_ptr(), 5000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert 1.0 < time.time() - t0 < 20.0
    for r in (1, 0):
        t0 = time.time()
        with pytest.raises(Exception, match="dead|did not arrive|gave up"):
            comms[r].query_kmers_sharded_device(shards[r], q.data_ptr(), 5000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert time.time() - t0 < 1.0  # at once: no second deadline
    torch.cuda.synchronize()
    for c in comms:
        c.close()


def test_ranks_that_disagree_on_a_flow_setting_fail_together():
    """The ID width, the number of pieces and whether minimizers travel size the flows of the exchange; they are per-rank settings.  Ranks
    that disagree would post sends and receives of different lengths (RCCL: a hang).  The count exchange carries the settings: every rank
    sees the same table and fails with the same error before a flow is posted -- and the same communicators answer again once they agree."""
    import torch
    import blight_amd as B
    rng = np.random.default_rng(5)
    k, m, n, b = 31, 9, 6, 6
    seqs = [_rand_seq(rng, 200) for _ in range(800)]
    single = B.kmer_Set_Light(k, m, n, 3, 1, b)
    single.construct_index_from_sequences(seqs)
    shards = []
    for r in range(2):
        a = B.kmer_Set_Light(k, m, n, 3, 1, b)
        a.set_shard(r, 2)
        a.construct_index_from_sequences(seqs)
        a.set_tuning("slot_piece_min", 2000)
        shards.append(a)
    uid = B.comm_loopback_id()
    comms = [None, None]

    def mk(r):
        comms[r] = B.Comm(uid, r, 2, device=0)
    run_ranks(2, mk)
    dev = torch.device("cuda", 0)
    from oracle import oracle as O
    orc = O.Oracle(k, m, n, b, seqs)
    q_np = np.concatenate([orc.canonical_kmers(s).reshape(-1) for s in seqs[:60]] + [rng.integers(0, 1 << 61, 3000, dtype=np.uint64)])
    q = torch.from_numpy(np.ascontiguousarray(q_np).view(np.int64)).to(dev)
    want = single.query_kmer_hash(np.ascontiguousarray(q_np))
    outs = [torch.empty(q.numel(), dtype=torch.int64, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()

    def query(r):
        comms[r].query_kmers_sharded_device(shards[r], q.data_ptr(), q.numel(), outs[r].data_ptr(), streams[r].cuda_stream)
        streams[r].synchronize()
        return outs[r].cpu().numpy()

    for key, values in (("ship_minimizers", (1, 0)), ("pieces", (1, 2))):
        for r in range(2):
            comms[r].set_tuning(key, values[r])
        got, err = run_ranks(2, query)
        assert all(e is not None and "disagree" in str(e) for e in err), err
        for r in range(2):
            comms[r].set_tuning(key, values[0])
        for _ in range(2):  # (the second batch goes through the learnt regions)
            got, err = run_ranks(2, query)
            assert not any(err), err
            for r in range(2):
                assert np.array_equal(got[r], want)
    for c in comms:
        c.close()
