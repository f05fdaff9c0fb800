"""A sharded index must answer EVERY k-mer like the single index -- also the reference's own false positives.  The reference's 2^b
scan does not stop at the end of a bucket (blight.cpp:729-739, 699-709): a window that starts in a shard's last bucket reads on into
what the single index holds next, i.e. the first buckets of the NEXT owner.  Every shard therefore keeps that head behind its own
sequence (the halo).  Here: every k-mer readable anywhere in the single index's packed sequence is queried through virtual shards on
one GPU (whole-input builds and builds from slices), for seeds where the CPU oracle shows that one of the reference's false
positives is readable ONLY across a shard cut."""
import numpy as np
import pytest

import shard_util as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,kb,m,n,b,nseq,world,seed", S.CUT_CASES)
def test_shards_answer_every_readable_kmer_like_the_single_index(k, kb, m, n, b, nseq, world, seed):
    import blight_amd as B
    from oracle import oracle as O
    seqs, orc, q, want, is_key, cut_fp, cut_fp_strict, bounds = S.make_cut_case(O, k, kb, m, n, b, nseq, world, seed)
    assert (want[is_key] >= 0).all()
    assert cut_fp.sum() > 0, "the case must hold a false positive of the reference that is only readable across a shard cut"
    full = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
    full.construct_index_from_sequences(seqs)
    full.set_tuning("absent_filter", 1)
    rq, _ = S.readable_kmers(full, k, kb, b)
    assert np.array_equal(rq, q)                                    # the single index's packed sequence is the oracle's
    assert np.array_equal(full.query_kmer_hash(q.reshape(-1)), want)
    # ---- whole-input shards (blmx_set_shard + the complete input on every rank): the halo is computed during the build
    whole = []
    for r in range(world):
        ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
        ix.set_shard(r, world)
        ix.construct_index_from_sequences(seqs)
        whole.append(ix)
        assert np.array_equal(ix.shard_bounds(), bounds)
    for filt in (1, 0):     # armed filter (its junction pass saw the halo) / no filter
        for ix in whole:
            ix.set_tuning("absent_filter", filt)
        got = S.answers_through_shards(whole, q)
        assert np.array_equal(got, want), (filt, int((got != want).sum()), int((got != want)[cut_fp].sum()))
    # ---- shards built from slices: the halo arrives with one more exchange after every rank has finished
    rng = np.random.default_rng(seed + 77)
    cuts = sorted(int(x) for x in rng.integers(0, len(seqs), world - 1))
    cuts = [0] + cuts + [len(seqs)]
    seen = {}

    def before(ixs):  # without the halo the shards lose exactly (some of) the false positives at the cuts
        for ix in ixs:
            ix.set_tuning("absent_filter", 1)
        g0 = S.answers_through_shards(ixs, q)
        wrong = g0 != want
        seen["wrong"] = int(wrong.sum())
        assert not wrong[is_key].any() and cut_fp_strict.any() and (g0[cut_fp_strict] == -1).all()

    sliced, _ = S.build_from_slices(B, seqs, cuts, (k, m, n, b), before_halo=before)
    assert seen["wrong"] > 0
    for filt in (1, 0):
        for ix in sliced:
            ix.set_tuning("absent_filter", filt)
        got = S.answers_through_shards(sliced, q)
        assert np.array_equal(got, want), (filt, int((got != want).sum()))
    for r in range(world):
        for name in B.EXPORT:
            assert np.array_equal(sliced[r].export(name), whole[r].export(name)), (r, name)
    # a second halo_set is a no-op; the image of a shard carries its halo
    recs = np.concatenate([x.shard_halo_get() for x in sliced])
    sliced[0].shard_halo_set(recs)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for r in (0, world - 1):
            sliced[r].dump_disk(td + "/s.blmx")
            ld = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b, kmer_bytes=kb)
            ld.load_disk(td + "/s.blmx")
            assert np.array_equal(ld.query_kmer_hash(q.reshape(-1)), sliced[r].query_kmer_hash(q.reshape(-1)))
    # ---- the device path of the exchange: partition on shard 0 (bins split at the boundaries), owners answer their segments
    import torch
    from blight_amd.dist import DeviceOps
    dev = torch.device("cuda", 0)
    for ix in whole:
        ix.set_tuning("absent_filter", 1)
    ops = [DeviceOps(ix, dev) for ix in whole]
    qt = torch.from_numpy(np.ascontiguousarray(q).view(np.int64)).to(dev)
    bt = torch.from_numpy(bounds.astype(np.int32)).to(dev)
    sorted_k, order, counts = ops[0].route(qt, bt, world)
    split = [int(x) for x in counts.tolist()]
    parts, o = [], 0
    for r in range(world):
        parts.append(ops[r].query(sorted_k[o:o + split[r]].contiguous()))
        o += split[r]
    got = ops[0].unpermute(torch.cat(parts), order)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
