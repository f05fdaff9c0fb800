"""Sharding on ONE GPU ("virtual shards"): S shard indexes built with blmx_set_shard + the HIP routing
kernels must reproduce the unsharded index's IDs exactly; shard boundaries follow the key-balance rule on every rank."""
import numpy as np
import pytest

import golden_util as G
from test_dist_gloo import shard_bounds_from_counts

pytestmark = pytest.mark.gpu


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


@pytest.mark.parametrize("world,k,m,n,b", [(2, 31, 9, 7, 6), (3, 31, 7, 5, 4), (8, 31, 9, 14, 6), (4, 63, 9, 7, 8)])
def test_virtual_shards(world, k, m, n, b):
    import torch
    import blight_amd as B
    from oracle import oracle as O
    rng = np.random.default_rng(world * 7 + k)
    seqs = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 80))) for _ in range(3000)]
    full = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
    full.construct_index_from_sequences(seqs)
    N = full.number_kmer
    shards = []
    for r in range(world):
        ix = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
        ix.set_shard(r, world)
        ix.construct_index_from_sequences(seqs)
        ix.set_tuning("absent_filter", 1)  # wait for the filter's background passes: the shard's filter regions follow LOCAL IDs (id_base > 0)
        shards.append(ix)
    bounds = shards[0].shard_bounds()
    assert np.array_equal(bounds, shard_bounds_from_counts(full.export("g_nkeys"), world))
    for ix in shards:
        assert np.array_equal(ix.shard_bounds(), bounds)
        assert ix.number_kmer == N
    assert sum(ix.stats()["local_number_kmer"] for ix in shards) == N
    # queries: all indexed k-mers (via the oracle's rolling encoder) + random negatives
    kb = full.kmer_bytes
    orc = O.Oracle(k, m, n, b, seqs[:400], kmer_bytes=kb)
    pos = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs[:400]])
    neg = rng.integers(0, 1 << 62, size=(5000, kb // 8), dtype=np.uint64)
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(np.concatenate([pos, neg]).view(np.int64)).to(dev)
    nq = q.shape[0]
    full.canonicalize_device(q.data_ptr(), nq, torch.cuda.current_stream().cuda_stream)
    want = torch.empty(nq, device=dev, dtype=torch.int64)
    full.query_kmers_device(q.data_ptr(), nq, want.data_ptr(), torch.cuda.current_stream().cuda_stream)
    # route on shard 0's handle, answer segment r on shard r, un-permute
    from blight_amd.dist import DeviceOps
    ops = [DeviceOps(ix, dev) for ix in shards]
    bt = torch.from_numpy(bounds.astype(np.int32)).to(dev)
    sorted_k, order, counts = ops[0].route(q, bt, world)
    split = [int(x) for x in counts.tolist()]
    assert sum(split) == nq
    parts, o = [], 0
    for r in range(world):
        parts.append(ops[r].query(sorted_k[o:o + split[r]].contiguous()))
        o += split[r]
    got = ops[0].unpermute(torch.cat(parts), order)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert int((want >= 0).sum().item()) >= pos.shape[0]
    # a shard answers nothing outside its groups
    alone = ops[0].query(q)
    torch.cuda.synchronize()
    own = alone >= 0
    assert torch.equal(alone[own], want[own]) and int(own.sum().item()) < int((want >= 0).sum().item())


def test_routing_kernels_by_destination():
    """blmx_route_count/scatter/unpermute (destination-major routing without the locality order; kept in the ABI for callers
    that exchange unordered batches): owners computed on the device equal the bounds rule, the permutation round-trips."""
    import torch
    import blight_amd as B
    from oracle import oracle as O
    k, m, n, b, world = 31, 9, 7, 6, 4
    rng = np.random.default_rng(5)
    seqs = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 80))) for _ in range(1500)]
    ix = B.kmer_Set_Light(k, m, n, 3, 1, b)
    ix.set_shard(1, world)
    ix.construct_index_from_sequences(seqs)
    bounds = ix.shard_bounds()
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    q = torch.from_numpy(rng.integers(0, 1 << 62, size=200000, dtype=np.uint64).view(np.int64)).to(dev)
    nq = q.shape[0]
    ix.canonicalize_device(q.data_ptr(), nq, s)
    bt = torch.from_numpy(bounds.astype(np.int32)).to(dev)
    dest = torch.empty(nq, device=dev, dtype=torch.uint8)
    counts = torch.empty(world, device=dev, dtype=torch.int64)
    ix.route_count_device(q.data_ptr(), nq, bt.data_ptr(), world, dest.data_ptr(), counts.data_ptr(), s)
    offsets = torch.cumsum(counts, 0) - counts
    sorted_k = torch.empty_like(q)
    order = torch.empty(nq, device=dev, dtype=torch.int32)
    cursor = torch.empty(world, device=dev, dtype=torch.int64)
    ix.route_scatter_device(q.data_ptr(), dest.data_ptr(), nq, offsets.data_ptr(), world, sorted_k.data_ptr(), order.data_ptr(),
                            cursor.data_ptr(), s)
    groups = torch.empty(nq, device=dev, dtype=torch.int32)
    ix.kmer_groups_device(q.data_ptr(), nq, groups.data_ptr(), s)
    torch.cuda.synchronize()
    want_dest = np.searchsorted(bounds[1:], groups.cpu().numpy(), side="right")
    assert np.array_equal(dest.cpu().numpy(), want_dest)
    assert np.array_equal(counts.cpu().numpy(), np.bincount(want_dest, minlength=world))
    back = torch.empty_like(q)
    ix.unpermute_device(sorted_k.data_ptr(), order.data_ptr(), nq, back.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(back, q)
    # the partitioned exchange groups by the same owners
    ordered = torch.empty_like(q)
    inv = torch.empty(nq, device=dev, dtype=torch.int32)
    pc = torch.zeros(world, device=dev, dtype=torch.int64)
    ix.partition_device(q.data_ptr(), nq, ordered.data_ptr(), inv.data_ptr(), pc.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(pc, counts)
    og = torch.empty(nq, device=dev, dtype=torch.int32)
    ix.kmer_groups_device(ordered.data_ptr(), nq, og.data_ptr(), s)
    torch.cuda.synchronize()
    od = np.searchsorted(bounds[1:], og.cpu().numpy(), side="right")
    assert np.all(np.diff(od) >= 0)  # owner-major
    assert torch.equal(ordered[inv.long()], q)


_NCCL_WORLD1 = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["BLIGHT_ROOT"])
import blight_amd as B
from blight_amd.dist import DeviceOps, ShardedQuerier
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", os.environ["BLIGHT_PORT"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rng = np.random.default_rng(3)
k, m, n, b = 31, 9, 7, 6
seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, 30 + int(rng.geometric(1 / 80)))) for _ in range(4000)]
ix = B.kmer_Set_Light(k, m, n, 3, 1, b)
ix.set_shard(0, 1)
ix.construct_index_from_sequences(seqs)
q = torch.from_numpy(rng.integers(0, 1 << 62, size=(300000, 1), dtype=np.uint64).view(np.int64)).to(dev)
s = torch.cuda.current_stream().cuda_stream
ix.canonicalize_device(q.data_ptr(), q.shape[0], s)
# half of the batch: k-mers of the index itself (first k-mer of every sequence)
ascii_ = torch.frombuffer(bytearray("".join(seqs).encode()), dtype=torch.uint8).to(dev)
starts = torch.from_numpy(np.cumsum([0] + [len(x) for x in seqs[:-1]]).astype(np.int64)).to(dev)
pk = torch.empty((starts.numel(), 1), device=dev, dtype=torch.int64)
err = torch.zeros(1, device=dev, dtype=torch.int32)
ix.kmers_at_device(ascii_.data_ptr(), starts.data_ptr(), starts.numel(), pk.data_ptr(), err.data_ptr(), s)
q = torch.cat([q, pk])[torch.randperm(q.shape[0] + pk.shape[0], device=dev)].contiguous()
want = torch.empty(q.shape[0], device=dev, dtype=torch.int64)
ix.query_kmers_device(q.data_ptr(), q.shape[0], want.data_ptr(), s)
bounds = torch.from_numpy(ix.shard_bounds().astype(np.int32)).to(dev)
sq = ShardedQuerier(DeviceOps(ix, dev), bounds)
for chunks in (1, 2, 3):
    got = sq.query(q, chunks=chunks)
    torch.cuda.synchronize()
    assert torch.equal(got, want), chunks
assert int((want >= 0).sum().item()) >= len(seqs)
# construction from slices with torch.distributed as the transport (blmx_shard_build_* phases, library buffers as zero-copy tensors)
from blight_amd.dist import DeviceSliceOps, SlicedBuilder
ix2 = B.kmer_Set_Light(k, m, n, 3, 1, b)
ix2.set_shard(0, 1)
SlicedBuilder(DeviceSliceOps(ix2, dev)).build(seqs)
for name in B.EXPORT:
    assert np.array_equal(ix.export(name), ix2.export(name)), name
got = torch.empty_like(want)
ix2.query_kmers_device(q.data_ptr(), q.shape[0], got.data_ptr(), s)
torch.cuda.synchronize()
assert torch.equal(got, want)
# the thin caller of the library's own RCCL path
from blight_amd.dist import NativeSharded
ix3 = B.kmer_Set_Light(k, m, n, 3, 1, b)
ns = NativeSharded(ix3, dev)
ns.build(seqs)
assert torch.equal(ns.query(q), want)
ns.close()
dist.destroy_process_group()
print("NCCL-WORLD1-OK")
"""


def test_sharded_querier_on_rccl_world1(tmp_path):
    """The pipelined exchange (asynchronous all_to_all_single on the `nccl` = RCCL backend interleaved with the HIP kernels
    on torch's current stream) with a world of ONE rank on the one GPU of the test box: the collectives degenerate to
    device copies, but the stream choreography is the production one."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "world1.py"
    script.write_text(_NCCL_WORLD1)
    import os
    env = dict(os.environ, BLIGHT_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), BLIGHT_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "NCCL-WORLD1-OK" in r.stdout, r.stderr[-2000:]


_NATIVE_WORLD1 = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["BLIGHT_ROOT"])
os.environ["BLIGHT_NO_TORCH"] = "1"      # no torch in this process: the library loads librccl.so.1 itself
import ctypes as C
import blight_amd as B
rng = np.random.default_rng(5)
k, m, n, b = int(sys.argv[1]), 9, 7, int(sys.argv[2])
kb = 8 if k <= 32 else 16
seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, k - 1 + int(rng.geometric(1 / 80)))) for _ in range(4000)]
ix = B.kmer_Set_Light(k, m, n, 3, 1, b)
ix.set_shard(0, 1)
ix.construct_index_from_sequences(seqs)
comm = B.Comm(B.comm_unique_id(), 0, 1, device=0)
# queries: the k-mers of the index (through the sequence entry point they are canonical) + random ones, shuffled
hip_path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]  # the HIP runtime the library itself loaded
hip = C.CDLL(hip_path)
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
def dmalloc(nbytes):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(max(nbytes, 16))) == 0
    return p
want_pos = ix.query_sequences_hash(seqs)
from oracle import oracle as O
orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
pos = np.concatenate([orc.canonical_kmers(s).reshape(-1, kb // 8) for s in seqs])
neg = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(50000)]
neg = O.kmers_to_array([min(x, O.p_rc(x, k, kb)) for x in neg], kb).reshape(-1, kb // 8)
q = np.concatenate([pos, neg])
q = np.ascontiguousarray(q[rng.permutation(len(q))])
want = ix.query_kmer_hash(q.reshape(-1))
assert np.array_equal(want, orc.query_kmers(q.reshape(-1) if kb == 8 else q))
nq = len(q)
d_q, d_ids = dmalloc(q.nbytes), dmalloc(nq * 8)
assert hip.hipMemcpy(d_q, C.c_void_p(q.ctypes.data), C.c_size_t(q.nbytes), 1) == 0
for pieces in ("1", "2", "3"):
    os.environ["BLIGHT_DIST_CHUNKS"] = pieces
    for n_ in (nq, 12345, 0):
        comm.query_kmers_sharded_device(ix, d_q, n_, d_ids, None)
        assert hip.hipDeviceSynchronize() == 0
        got = np.full(max(n_, 1), -7, dtype=np.int64)
        assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), d_ids, C.c_size_t(n_ * 8), 2) == 0
        assert np.array_equal(got[:n_], want[:n_]), (pieces, n_)
# sharded construction from slices, world 1: chop -> all-reduce -> all-gather -> all-to-all-v (to itself) -> owner's build
ix2 = B.kmer_Set_Light(k, m, n, 3, 1, b)
ix2.set_shard(0, 1)
comm.build_sharded_from_sequences(ix2, seqs)
for name in B.EXPORT:
    assert np.array_equal(ix.export(name), ix2.export(name)), name
assert np.array_equal(ix2.query_kmer_hash(q.reshape(-1)), want)
s1, s2 = ix.stats(), ix2.stats()
assert not [f for f in s1 if f not in ("build_seconds", "number_query", "build_scratch_peak_bytes") and s1[f] != s2[f]], [(f, s1[f], s2[f]) for f in s1 if s1[f] != s2[f] and f not in ("build_seconds", "number_query", "build_scratch_peak_bytes")]
ix3 = B.kmer_Set_Light(k, m, n, 3, 1, b)
ix3.set_shard(0, 1)
try:
    comm.build_sharded_from_sequences(ix3, seqs[:10] + ["ACGT"])      # a sequence shorter than k: every rank fails, nobody hangs
    raise SystemExit("expected a failure")
except B.BlightError as e:
    assert e.code == B.BLMX_ESHORT, e
comm.build_sharded_from_sequences(ix3, seqs)                           # ... and the communicator and the index are still usable
assert np.array_equal(ix3.query_kmer_hash(q.reshape(-1)), want)
# host-buffer collectives: k-mers and whole sequences (this rank's read front end + the exchange)
assert np.array_equal(comm.query_kmers_sharded(ix3, q.reshape(-1)), want)
some = seqs[:300] + ["".join("ACGT"[i] for i in rng.integers(0, 4, 200)) for _ in range(50)] + ["ACG"]
assert np.array_equal(comm.query_sequences_sharded(ix3, some), ix.query_sequences_hash(some))
assert comm.query_sequences_sharded(ix3, []).size == 0
try:
    comm.query_sequences_sharded(ix3, ["ACGTN" * 20])
    raise SystemExit("expected Invalid char in DNA")
except ValueError:
    pass
# the same file on every rank, each indexes its part
fa = os.path.join(os.environ["BLIGHT_TMP"], "u.fa")
open(fa, "w").write("".join(">u%d\n%s\n" % (i, s_) for i, s_ in enumerate(seqs)))
ix4 = B.kmer_Set_Light(k, m, n, 3, 1, b)
ix4.set_shard(0, 1)
comm.build_sharded_from_fasta(ix4, fa)
for name in B.EXPORT:
    assert np.array_equal(ix.export(name), ix4.export(name)), name
ix5 = B.kmer_Set_Light(k, m, n, 3, 1, b)
ix5.set_shard(0, 1)
try:
    comm.build_sharded_from_fasta(ix5, fa + ".missing")
    raise SystemExit("expected OSError")
except OSError:
    pass
comm.close()
print("NATIVE-WORLD1-OK")
"""


@pytest.mark.parametrize("k,b", [(31, 6), (63, 8)])
def test_native_rccl_exchange_world1(tmp_path, k, b):
    """blmx_comm_* + blmx_query_kmers_sharded_device: the native exchange (RCCL loaded by the library itself, grouped
    ncclSend / ncclRecv all-to-all-v on the communicator's stream, pieces pipelined against the query kernel) with a world of
    one rank on the test box's GPU, in a process WITHOUT torch: IDs equal the unsharded path and the oracle, for 1, 2 and 3
    pieces, partial and empty batches."""
    import os
    import subprocess
    import sys
    script = tmp_path / "native_world1.py"
    script.write_text(_NATIVE_WORLD1)
    env = dict(os.environ, BLIGHT_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), HSA_ENABLE_IPC_MODE_LEGACY="0",
               BLIGHT_TMP=str(tmp_path))
    r = subprocess.run([sys.executable, str(script), str(k), str(b)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "NATIVE-WORLD1-OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("world,k,m,n,b", [(2, 31, 9, 7, 6), (3, 31, 7, 5, 4), (8, 31, 9, 14, 6), (4, 63, 9, 7, 8)])
def test_sharded_build_from_slices(world, k, m, n, b):
    """Construction where every rank only sees ITS OWN consecutive slice of the input (blmx_shard_build_begin / _plan / _finish;
    the native collective blmx_build_sharded_from_sequences runs the same three phases around RCCL): `world` virtual ranks on one
    GPU, the all-reduce is a numpy sum and the all-to-all-v a set of device copies.  Every array of every shard must equal the
    whole-input sharded build (which equals the unsharded index up to the shard's group range: test_virtual_shards)."""
    import blight_amd as B
    import shard_util as S
    from oracle import oracle as O
    rng = np.random.default_rng(world * 11 + k)
    seqs = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 80))) for _ in range(3000)]
    longest = max(seqs[:40], key=len)
    seqs += ["A" * (k + 40), "AC" * (k + 3), seqs[5], longest[min(3, len(longest) - k):]]  # duplicates and low complexity: fallback keys, shared buckets
    # uneven consecutive slices; with three or more ranks one of them gets nothing
    cuts = sorted(int(x) for x in rng.integers(0, len(seqs), world - 1))
    if world >= 3:
        cuts[1] = cuts[0]
    cuts = [0] + cuts + [len(seqs)]
    ref = []
    for r in range(world):
        a = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
        a.set_shard(r, world)
        a.construct_index_from_sequences(seqs)
        ref.append(a)
    ixs, gh = S.build_from_slices(B, seqs, cuts, (k, m, n, b))
    assert int(gh[-1]) == ref[0].stats()["number_super_kmer"]
    with pytest.raises(ValueError):
        ixs[0].shard_build_plan(gh, world + 1)       # the caller's world must be the index's (its arrays are sized by it)
    for r in range(world):
        assert np.array_equal(ixs[r].shard_bounds(), ref[r].shard_bounds())
        for name in B.EXPORT:
            assert np.array_equal(ixs[r].export(name), ref[r].export(name)), (r, name)
        s1, s2 = ixs[r].stats(), ref[r].stats()
        assert not [f for f in s1 if f not in ("build_seconds", "number_query", "build_scratch_peak_bytes") and s1[f] != s2[f]], [(r, f, s1[f], s2[f]) for f in s1 if s1[f] != s2[f] and f not in ("build_seconds", "number_query", "build_scratch_peak_bytes")]
        for x in (ixs[r], ref[r]):
            x.set_tuning("absent_filter", 1)  # filter armed: its regions follow the shard's LOCAL IDs
        some = seqs[::7] + [_rand_seq(rng, 3 * k) for _ in range(50)]
        got = ixs[r].query_sequences_hash(some)
        assert np.array_equal(got, ref[r].query_sequences_hash(some)), r
        # (sanity only -- `some` holds a seventh of the input: a shard that owns next to nothing, as happens with fewer groups than ranks
        # or a handful of minimizer values, may well see none of its k-mers there)
        assert (got >= 0).any() or s1["local_number_kmer"] < 2000, r
    # EVERY k-mer readable anywhere in the SINGLE index's packed sequence (junctions of super-k-mers, of buckets -- also the ones
    # that are shard cuts now -- and the padding behind the end): both kinds of shards answer like the single index and the oracle
    full = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
    full.construct_index_from_sequences(seqs)
    kb = full.kmer_bytes
    q, _ = S.readable_kmers(full, k, kb, b)
    orc = O.Oracle(k, m, n, b, seqs, kmer_bytes=kb)
    want = orc.query_kmers(q.reshape(-1) if kb == 8 else q)
    assert np.array_equal(full.query_kmer_hash(q.reshape(-1)), want)
    for shards in (ixs, ref):
        got = S.answers_through_shards(shards, q)
        assert np.array_equal(got, want), int((got != want).sum())
    # phase errors: finish without plan, plan twice
    y = B.kmer_Set_Light(k, m, n, min(n, 3), 1, b)
    y.set_shard(0, world)
    with pytest.raises(B.BlightError):
        y.shard_build_finish(0, 0, np.zeros(world, np.uint64), np.zeros(world, np.uint64))
    y.shard_build_begin(seqs[:50])
    with pytest.raises(B.BlightError):
        y.shard_build_finish(0, 0, np.zeros(world, np.uint64), np.zeros(world, np.uint64))
