"""Sharded multi-GPU queries: one process per GPU, MPHF groups (contiguous minimizer-bucket ranges,
reference blight.cpp:692) partitioned over the ranks by key count, and ONE exchange step per query batch:

    local k-mers --partition by minimizer prefix, bins split at the shard boundaries (= grouped by owner rank AND ordered
                   for locality, one pass)--> all-to-all-v of k-mers  (RCCL over xGMI, `nccl` backend)
                 owner runs the query kernel on the received, already ordered segments (no second partition pass)
                 all-to-all-v of the 8-byte IDs back --> gather through the inverse permutation into the caller's order

The reference has no communication at all (single process, OpenMP); this is the MI355X-native equivalent of its shared
memory.  all-to-all-v uses every point-to-point xGMI link at once (each pairwise flow has its own link), payload is
8 (16) B out + 8 B back per remote query.

The exchange choreography lives in `ShardedQuerier.query()` and only talks to an `ops` object, so that it is exercised by
the world_size-2 `gloo` CPU tests (tests/test_dist_gloo.py, with numpy ops and the CPU oracle as the per-rank engine) and
runs unchanged on GPUs with `DeviceOps` (hand-written HIP routing kernels + the HIP query kernel).
"""
import os

import torch
import torch.distributed as dist


class DeviceOps:
    """Per-rank operations on the GPU: everything is a kernel of libblight_mi355x.so on torch's current stream."""

    def __init__(self, index, device):
        self.ix = index
        self.dev = device
        self.kw = index.kmer_bytes // 8

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def empty_kmers(self, n):
        return torch.empty((n, self.kw), device=self.dev, dtype=torch.int64)

    def empty_ids(self, n):
        return torch.empty(n, device=self.dev, dtype=torch.int64)

    def route(self, kmers, bounds_dev, world):
        """-> (k-mers ordered by partition bin = grouped by owner rank, inverse permutation, counts per owner [device])

        ONE partition pass serves both purposes: it groups the batch by owner (every bin belongs to one owner) and leaves
        every owner's segment ordered by minimizer prefix, which is the order its query kernel wants -- the owner does not
        partition again."""
        nq = kmers.shape[0]
        sorted_k = self.empty_kmers(nq)
        inv = torch.empty(nq, device=self.dev, dtype=torch.int32)
        counts = torch.zeros(world, device=self.dev, dtype=torch.int64)
        self.ix.partition_device(kmers.data_ptr(), nq, sorted_k.data_ptr(), inv.data_ptr(), counts.data_ptr(), self._stream())
        return sorted_k, inv, counts

    def query(self, kmers):
        """answers a batch made of ordered segments (one per peer); IDs in batch order"""
        n = kmers.shape[0]
        ids = self.empty_ids(n)
        self.ix.query_partitioned_device(kmers.data_ptr(), n, ids.data_ptr(), self._stream())
        return ids

    def unpermute(self, vals, inv):
        out = self.empty_ids(vals.shape[0])
        self.unpermute_into(out, vals, inv)
        return out

    def unpermute_into(self, out, vals, inv):
        """out[i] = vals[inv[i]] written straight into `out` (a contiguous slice of the caller's result)"""
        assert out.is_contiguous() and out.shape[0] == vals.shape[0]
        self.ix.gather_ids_device(vals.data_ptr(), inv.data_ptr(), vals.shape[0], out.data_ptr(), self._stream())


class ShardedQuerier:
    """query(kmers) -> IDs in the caller's order, with every k-mer answered by the rank that owns its MPHF group."""

    def __init__(self, ops, bounds, group=None):
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        assert len(bounds) == self.world + 1
        self.bounds = bounds  # tensor on the ops' device, uint32 stored as int32 bit pattern
        # Rehearsal switch for boxes with ONE GPU shared by several ranks (RCCL refuses duplicate devices): run the
        # collectives through host memory on the gloo backend.  Never set in production / by bench.py's driver path.
        self.host_staging = os.environ.get("BLIGHT_DIST_HOST_STAGING") == "1"

    def _a2a(self, out, inp, out_split=None, in_split=None):
        if self.host_staging and out.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_split, input_split_sizes=in_split, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, output_split_sizes=out_split, input_split_sizes=in_split, group=self.group)

    def query(self, kmers, chunks=None):
        """IDs of `kmers` in the caller's order.

        The batch is cut into `chunks` consecutive pieces (default: env BLIGHT_DIST_CHUNKS or 2; 1 when the collectives
        are host-staged) that go through partition -> exchange -> query -> exchange -> gather as a software pipeline:
        all pieces are partitioned first and their per-owner counts travel in ONE small all-to-all (one host sync), then
        every piece's k-mer exchange is issued asynchronously, so that piece c + 1 is on the links while the owner answers
        piece c, and the IDs of piece c travel back while piece c + 1 is being answered.  The collectives are issued in
        the same order on every rank (k-mers 0..C-1 interleaved with IDs 0..C-1 as the loop below shows)."""
        world = self.world
        n = kmers.shape[0]
        if chunks is None:
            chunks = 1 if self.host_staging else int(os.environ.get("BLIGHT_DIST_CHUNKS", "2"))
        chunks = max(1, int(chunks))  # the same on every rank, whatever the local batch size (pieces may be empty)
        cuts = [(n * c) // chunks for c in range(chunks + 1)]
        parts = [self.ops.route(kmers[cuts[c]:cuts[c + 1]], self.bounds, world) for c in range(chunks)]
        # counts of all pieces in one exchange: row r of the send matrix = what I send to rank r, piece by piece
        send_counts = torch.stack([p[2] for p in parts], dim=1).contiguous()  # (world, chunks)
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts.view(-1), send_counts.view(-1))
        send_split = send_counts.tolist()
        recv_split = recv_counts.tolist()
        out = self.ops.empty_ids(n)
        if self.host_staging or chunks == 1:
            for c in range(chunks):
                ss = [int(send_split[r][c]) for r in range(world)]
                rs = [int(recv_split[r][c]) for r in range(world)]
                recv_k = self.ops.empty_kmers(sum(rs))
                self._a2a(recv_k, parts[c][0], rs, ss)
                ids_remote = self.ops.query(recv_k)
                ids_sorted = self.ops.empty_ids(sum(ss))
                self._a2a(ids_sorted, ids_remote, ss, rs)
                self._unpermute_into(out[cuts[c]:cuts[c + 1]], ids_sorted, parts[c][1])
            return out
        ss = [[int(send_split[r][c]) for r in range(world)] for c in range(chunks)]
        rs = [[int(recv_split[r][c]) for r in range(world)] for c in range(chunks)]
        recv_k = [self.ops.empty_kmers(sum(rs[c])) for c in range(chunks)]
        ids_sorted = [self.ops.empty_ids(sum(ss[c])) for c in range(chunks)]
        ids_remote = [None] * chunks
        w_k = [None] * chunks
        w_i = [None] * chunks
        w_k[0] = dist.all_to_all_single(recv_k[0], parts[0][0], rs[0], ss[0], group=self.group, async_op=True)
        for c in range(chunks):
            if c + 1 < chunks:  # the next piece's k-mers travel while this piece is answered
                w_k[c + 1] = dist.all_to_all_single(recv_k[c + 1], parts[c + 1][0], rs[c + 1], ss[c + 1], group=self.group, async_op=True)
            w_k[c].wait()
            ids_remote[c] = self.ops.query(recv_k[c])
            w_i[c] = dist.all_to_all_single(ids_sorted[c], ids_remote[c], ss[c], rs[c], group=self.group, async_op=True)
        for c in range(chunks):
            w_i[c].wait()
            self._unpermute_into(out[cuts[c]:cuts[c + 1]], ids_sorted[c], parts[c][1])
        return out

    def _unpermute_into(self, out_slice, vals, order):
        if hasattr(self.ops, "unpermute_into"):
            self.ops.unpermute_into(out_slice, vals, order)
        else:
            out_slice.copy_(self.ops.unpermute(vals, order))


# ---- sharded CONSTRUCTION from slices --------------------------------------------------------------------------------------------
class _DevicePointer:
    """zero-copy torch view of a device buffer owned by the library (__cuda_array_interface__)"""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


class DeviceSliceOps:
    """The three phases of blmx_shard_build_* on this rank's GPU (the native collective blmx_build_sharded_from_sequences runs the
    same phases around RCCL inside the library; this class is for callers that bring torch.distributed as their transport)."""

    def __init__(self, index, device):
        self.ix, self.dev = index, device

    def begin(self, seqs):
        return torch.from_numpy(self.ix.shard_build_begin(seqs).view("int64").copy())

    def plan(self, global_hist, world):
        sk, words, pm, pw = self.ix.shard_build_plan(global_hist.numpy().view("uint64"), world)
        nsk, nw = int(sk.sum()), int(words.sum())
        meta = torch.as_tensor(_DevicePointer(pm, nsk), device=self.dev) if nsk else torch.empty(0, dtype=torch.int64, device=self.dev)
        packed = torch.as_tensor(_DevicePointer(pw, nw), device=self.dev) if nw else torch.empty(0, dtype=torch.int64, device=self.dev)
        return [int(x) for x in sk], [int(x) for x in words], meta, packed

    def empty(self, n):
        return torch.empty(max(int(n), 1), dtype=torch.int64, device=self.dev)[:int(n)]

    def finish(self, meta_in, words_in, recv_sk, recv_words):
        torch.cuda.synchronize()
        self.ix.shard_build_finish(meta_in.data_ptr(), words_in.data_ptr(), recv_sk, recv_words)

    def halo_get(self):
        """the head of this shard's packed sequence (int64 view of the library's record)"""
        return torch.from_numpy(self.ix.shard_halo_get().view("int64").copy())

    def halo_set(self, records):
        self.ix.shard_halo_set(records.cpu().numpy().view("uint64"))


class SlicedBuilder:
    """build(seqs): every rank passes ITS OWN consecutive slice of the input (rank order = input order).  One all-reduce (per-group
    key histograms -> identical key-balanced boundaries everywhere), one small all-to-all (counts), two all-to-all-v (one u64 of
    (minimizer, length) per super-k-mer; the 2-bit packed nucleotides, every destination's part starting on a word).  What a rank
    receives is concatenated in source-rank order = input order, so the owner's stable minimizer sort reproduces the bucket
    content -- and hence the IDs -- of a single-GPU build."""

    def __init__(self, ops, group=None):
        self.ops, self.group = ops, group
        self.world = dist.get_world_size(group)
        self.host_staging = os.environ.get("BLIGHT_DIST_HOST_STAGING") == "1"
        # the small host-side vectors (histogram, counts) travel through the device when the backend only moves device memory (nccl)
        self.cdev = "cpu" if self.host_staging else getattr(ops, "dev", "cpu")

    def _a2a(self, out, inp, out_split, in_split):
        if self.host_staging and out.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_split, input_split_sizes=in_split, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, output_split_sizes=out_split, input_split_sizes=in_split, group=self.group)

    def build(self, seqs):
        hist = self.ops.begin(seqs).to(self.cdev)  # int64: keys per MPHF group of this slice, then its super-k-mer count
        dist.all_reduce(hist, group=self.group)
        send_sk, send_words, meta, packed = self.ops.plan(hist.cpu(), self.world)
        mine = torch.tensor([[send_sk[r], send_words[r]] for r in range(self.world)], dtype=torch.int64, device=self.cdev)
        theirs = torch.empty_like(mine)
        dist.all_to_all_single(theirs.view(-1), mine.view(-1), group=self.group)
        theirs = theirs.cpu()
        recv_sk = [int(x) for x in theirs[:, 0]]
        recv_words = [int(x) for x in theirs[:, 1]]
        meta_in, words_in = self.ops.empty(sum(recv_sk)), self.ops.empty(sum(recv_words))
        self._a2a(meta_in, meta, recv_sk, send_sk)
        self._a2a(words_in, packed, recv_words, send_words)
        self.ops.finish(meta_in, words_in, recv_sk, recv_words)
        # halo: every shard keeps, behind its own sequence, the head of what follows it in the single index (the reference's scan
        # runs on past the end of a bucket, blight.cpp:729-739) -- one small all-gather of the shards' heads
        if hasattr(self.ops, "halo_get"):
            mine = self.ops.halo_get().to(self.cdev)
            recs = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(recs, mine, group=self.group)
            self.ops.halo_set(torch.cat([r.cpu() for r in recs]))


# ---- the native exchange (RCCL inside libblight_mi355x.so) behind the same two method names --------------------------------------
class NativeSharded:
    """Thin caller of the library's own multi-GPU path (blmx_comm_*, include/blight_mi355x.h): rank 0's communicator id travels through
    the launcher's process group, everything else -- construction from this rank's slice, the query exchange -- runs inside the
    library on RCCL.  ShardedQuerier / SlicedBuilder above are the same choreographies over torch.distributed, kept for the gloo
    CPU tests and as a fallback where the library cannot open RCCL itself."""

    def __init__(self, index, device, group=None):
        import blight_amd
        self.ix, self.dev, self.group = index, device, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        uid = torch.zeros(blight_amd.COMM_ID_BYTES, dtype=torch.uint8, device=device)
        if self.rank == 0:
            uid.copy_(torch.frombuffer(bytearray(blight_amd.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0, group=group)
        self.comm = blight_amd.Comm(bytes(uid.cpu().numpy().tobytes()), self.rank, self.world, device=device.index or 0)

    def build(self, seqs):
        """collective: this rank's consecutive slice of the sequences"""
        self.ix.set_shard(self.rank, self.world)
        self.comm.build_sharded_from_sequences(self.ix, seqs)

    def query(self, kmers):
        """collective: canonical k-mers (int64 tensor on the device, one row per k-mer) -> IDs in the caller's order"""
        n = kmers.shape[0]
        out = torch.empty(n, dtype=torch.int64, device=self.dev)
        self.comm.query_kmers_sharded_device(self.ix, kmers.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out

    def close(self):
        self.comm.close()
