// blmx_comm.hip -- native multi-GPU queries: RCCL over xGMI behind the C ABI (the reference is single-process).
//
// Ownership unit = MPHF group (a contiguous range of minimizer buckets, reference blight.cpp:692): rank r of `world` holds the
// groups [bounds[r], bounds[r + 1]) (blmx_set_shard).  One query batch = one exchange step:
//
//   partition by minimizer prefix, bins split at the shard boundaries  (k_bin_hist / scan / k_bin_scatter: the batch becomes a
//                                                                      concatenation of per-owner segments, each already in
//                                                                      the order the owner's query kernel wants)
//   all-to-all-v of the k-mers     grouped ncclSend / ncclRecv: every pairwise flow has its own xGMI link (8 or 16 B per k-mer)
//   owner: k_query on what it received (no second partition pass)
//   all-to-all-v of the 8-byte IDs back, gather through the inverse permutation into the caller's order
//
// The batch is cut into pieces that run as a software pipeline: the k-mers of piece c + 1 travel (on the communicator's own
// stream) while the owner answers piece c, and the IDs of piece c travel back while piece c + 1 is answered.  The per-owner
// counts of ALL pieces are exchanged first with one ncclAllGather (the only host synchronisations of a batch).
// RCCL is loaded at run time (dlopen "librccl.so.1"): a process that already carries the library (e.g. PyTorch) shares it, and
// single-GPU users of libblight_mi355x.so do not need it at all.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <memory>

#include <rccl/rccl.h>

#include <chrono>

#include "blmx_internal.h"

namespace blmx {

namespace {

struct Rccl {
	void* lib = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string err;
};

Rccl& rccl()
{
	static Rccl r = [] {
		Rccl x;
		for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (x.lib)
				break;
		}
		if (!x.lib) {
			x.err = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "not found");
			return x;
		}
#define BLMX_SYM(field, sym)                                                   \
	x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.lib, #sym));         \
	if (!x.field)                                                              \
		x.err = "librccl: missing symbol " #sym;
		BLMX_SYM(GetUniqueId, ncclGetUniqueId)
		BLMX_SYM(CommInitRank, ncclCommInitRank)
		BLMX_SYM(CommDestroy, ncclCommDestroy)
		BLMX_SYM(GroupStart, ncclGroupStart)
		BLMX_SYM(GroupEnd, ncclGroupEnd)
		BLMX_SYM(Send, ncclSend)
		BLMX_SYM(Recv, ncclRecv)
		BLMX_SYM(AllGather, ncclAllGather)
		BLMX_SYM(AllReduce, ncclAllReduce)
		BLMX_SYM(GetErrorString, ncclGetErrorString)
#undef BLMX_SYM
		return x;
	}();
	if (!r.err.empty())
		throw StatusError{BLMX_ERCCL, r.err};
	return r;
}

#define BLMX_NCCL(expr)                                                                                          \
	do {                                                                                                         \
		const ncclResult_t _r = (expr);                                                                          \
		if (_r != ncclSuccess)                                                                                   \
			throw StatusError{BLMX_ERCCL, std::string(#expr) + ": " + rccl().GetErrorString(_r)};                \
	} while (0)

constexpr int kMaxPieces = 4;

void grow(DevBuf& b, size_t bytes, hipStream_t s)
{
	if (bytes > b.bytes)
		BLMX_HIP(b.alloc(bytes + bytes / 8 + 256, false, s));
}

}  // namespace

struct Comm {
	ncclComm_t comm = nullptr;
	uint32_t rank = 0, world = 1;
	int device = 0;
	hipStream_t cs = nullptr;  // the exchange runs on this stream, next to the caller's compute stream
	hipEvent_t ev_part[kMaxPieces] = {}, ev_k[kMaxPieces] = {}, ev_q[kMaxPieces] = {}, ev_i[kMaxPieces] = {};
	DevBuf sorted[kMaxPieces], inv[kMaxPieces], recv_k[kMaxPieces], ids_remote[kMaxPieces], ids_sorted[kMaxPieces];
	DevBuf counts, all_counts;
	Pinned h_counts;
	std::mutex mu;
	~Comm()
	{
		(void)hipSetDevice(device);
		for (auto* arr : {ev_part, ev_k, ev_q, ev_i})
			for (int i = 0; i < kMaxPieces; i++)
				if (arr[i])
					(void)hipEventDestroy(arr[i]);
		if (comm)
			(void)rccl().CommDestroy(comm);
		if (cs)
			(void)hipStreamDestroy(cs);
	}
};

void comm_unique_id(void* id_out)
{
	ncclUniqueId id;
	BLMX_NCCL(rccl().GetUniqueId(&id));
	static_assert(sizeof(id) == BLMX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
	memcpy(id_out, &id, sizeof id);
}

Comm* comm_create(const void* id_bytes, uint32_t rank, uint32_t world, int device)
{
	if (world == 0 || world > 16 || rank >= world)
		throw StatusError{BLMX_EINVAL, "need rank < world <= 16"};
	BLMX_HIP(hipSetDevice(device));
	std::unique_ptr<Comm> c(new Comm());
	c->rank = rank;
	c->world = world;
	c->device = device;
	ncclUniqueId id;
	memcpy(&id, id_bytes, sizeof id);
	BLMX_NCCL(rccl().CommInitRank(&c->comm, (int)world, id, (int)rank));
	BLMX_HIP(hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking));
	for (auto* arr : {c->ev_part, c->ev_k, c->ev_q, c->ev_i})
		for (int i = 0; i < kMaxPieces; i++)
			BLMX_HIP(hipEventCreateWithFlags(&arr[i], hipEventDisableTiming));
	return c.release();
}

void comm_destroy(Comm* c) { delete c; }

// all-to-all-v as grouped point-to-point transfers; counts in elements of `words` uint64 each.  A pairwise flow larger than
// kFlowWords is cut into rounds (both ends of a flow know its length, so they cut it alike; ranks whose flows are shorter simply
// take part in fewer rounds): one 1.7-GB ncclSend/ncclRecv pair delivered wrong data here (RCCL 2.26/2.27, self flow), 1 GB was fine.
constexpr uint64_t kFlowWords = 1ull << 25;  // 256 MiB

static void all_to_all_v(Comm& c, const void* send, const uint64_t* send_cnt, void* recv, const uint64_t* recv_cnt, size_t words, hipStream_t s)
{
	Rccl& R = rccl();
	uint64_t so[17], ro[17], rounds = 0;
	so[0] = ro[0] = 0;
	for (uint32_t r = 0; r < c.world; r++) {
		so[r + 1] = so[r] + send_cnt[r] * words;
		ro[r + 1] = ro[r] + recv_cnt[r] * words;
		rounds = std::max(rounds, (std::max(send_cnt[r], recv_cnt[r]) * words + kFlowWords - 1) / kFlowWords);
	}
	for (uint64_t j = 0; j < rounds; j++) {
		const uint64_t off = j * kFlowWords;
		BLMX_NCCL(R.GroupStart());
		for (uint32_t r = 0; r < c.world; r++) {
			const uint64_t sn = send_cnt[r] * words, rn = recv_cnt[r] * words;
			if (off < sn)
				BLMX_NCCL(R.Send((const uint64_t*)send + so[r] + off, std::min(kFlowWords, sn - off), ncclUint64, (int)r, c.comm, s));
			if (off < rn)
				BLMX_NCCL(R.Recv((uint64_t*)recv + ro[r] + off, std::min(kFlowWords, rn - off), ncclUint64, (int)r, c.comm, s));
		}
		BLMX_NCCL(R.GroupEnd());
	}
}

// IDs of nq canonical k-mers in HBM, every k-mer answered by the rank that owns its MPHF group; collective: every rank of the
// communicator calls it (with its own batch, possibly empty) in the same order.
void query_kmers_sharded(Index& ix, Comm& c, const void* d_kmers, uint64_t nq, int64_t* d_ids, hipStream_t s)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	if (ix.shard_world != c.world || ix.shard_rank != c.rank)
		throw StatusError{BLMX_EINVAL, "the index was sharded for another rank / world than this communicator"};
	if ((int)ix.prm.device != c.device)
		throw StatusError{BLMX_EINVAL, "index and communicator live on different devices"};
	std::lock_guard<std::mutex> lock(c.mu);
	BLMX_HIP(hipSetDevice(c.device));
	Rccl& R = rccl();
	const uint32_t world = c.world, me = c.rank;
	const char* e = getenv("BLIGHT_DIST_CHUNKS");
	int pieces = e ? atoi(e) : 2;  // must be the same on every rank
	pieces = pieces < 1 ? 1 : (pieces > kMaxPieces ? kMaxPieces : pieces);
	const size_t kw = (size_t)ix.KW;
	uint64_t cut[kMaxPieces + 1];
	for (int p = 0; p <= pieces; p++)
		cut[p] = nq * (uint64_t)p / (uint64_t)pieces;
	// ---- partition every piece; the per-owner counts of all pieces are exchanged in one all-gather
	grow(c.counts, (size_t)pieces * world * 8, s);
	grow(c.all_counts, (size_t)pieces * world * world * 8, s);
	c.h_counts.ensure((size_t)pieces * world * world * 8);
	for (int p = 0; p < pieces; p++) {
		const uint64_t n = cut[p + 1] - cut[p];
		grow(c.sorted[p], n * 8 * kw + 16, s);
		grow(c.inv[p], n * 4 + 16, s);
		launch_partition(ix, (const uint64_t*)d_kmers + cut[p] * kw, n, c.sorted[p].p, c.inv[p].as<uint32_t>(), c.counts.as<uint64_t>() + (size_t)p * world, s);
		BLMX_HIP(hipEventRecord(c.ev_part[p], s));
	}
	BLMX_NCCL(R.AllGather(c.counts.p, c.all_counts.p, (size_t)pieces * world, ncclUint64, c.comm, s));
	BLMX_HIP(hipMemcpyAsync(c.h_counts.p, c.all_counts.p, (size_t)pieces * world * world * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	const uint64_t* all = c.h_counts.as<uint64_t>();  // all[(r * pieces + p) * world + d] = k-mers rank r sends to rank d in piece p
	uint64_t send_cnt[kMaxPieces][16], recv_cnt[kMaxPieces][16], recv_total[kMaxPieces];
	for (int p = 0; p < pieces; p++) {
		recv_total[p] = 0;
		for (uint32_t r = 0; r < world; r++) {
			send_cnt[p][r] = all[((size_t)me * pieces + p) * world + r];
			recv_cnt[p][r] = all[((size_t)r * pieces + p) * world + me];
			recv_total[p] += recv_cnt[p][r];
		}
		grow(c.recv_k[p], recv_total[p] * 8 * kw + 16, s);
		grow(c.ids_remote[p], recv_total[p] * 8 + 16, s);
		grow(c.ids_sorted[p], (cut[p + 1] - cut[p]) * 8 + 16, s);
	}
	BLMX_HIP(hipStreamSynchronize(s));  // buffers that grew must exist before the other stream touches them
	// ---- the pipeline: collectives are issued in the same order on every rank (k-mers 0, k-mers 1, IDs 0, k-mers 2, IDs 1, ...)
	auto send_kmers = [&](int p) {
		BLMX_HIP(hipStreamWaitEvent(c.cs, c.ev_part[p], 0));
		all_to_all_v(c, c.sorted[p].p, send_cnt[p], c.recv_k[p].p, recv_cnt[p], kw, c.cs);
		BLMX_HIP(hipEventRecord(c.ev_k[p], c.cs));
	};
	send_kmers(0);
	for (int p = 0; p < pieces; p++) {
		if (p + 1 < pieces)
			send_kmers(p + 1);  // the next piece travels while this one is answered
		BLMX_HIP(hipStreamWaitEvent(s, c.ev_k[p], 0));
		launch_query_partitioned(ix, c.recv_k[p].p, recv_total[p], c.ids_remote[p].as<int64_t>(), s);
		BLMX_HIP(hipEventRecord(c.ev_q[p], s));
		BLMX_HIP(hipStreamWaitEvent(c.cs, c.ev_q[p], 0));
		all_to_all_v(c, c.ids_remote[p].p, recv_cnt[p], c.ids_sorted[p].p, send_cnt[p], 1, c.cs);
		BLMX_HIP(hipEventRecord(c.ev_i[p], c.cs));
	}
	for (int p = 0; p < pieces; p++) {
		BLMX_HIP(hipStreamWaitEvent(s, c.ev_i[p], 0));
		launch_gather_ids(ix, c.ids_sorted[p].as<int64_t>(), c.inv[p].as<uint32_t>(), cut[p + 1] - cut[p], d_ids + cut[p], s);
	}
	ix.number_query.fetch_add(nq, std::memory_order_relaxed);
}

// Sharded construction from slices: every rank passes ITS OWN consecutive piece of the input (rank order = input order).  Collective.
// Steps: chop locally -> all-reduce of the per-group key histograms (every rank derives the same key-balanced boundaries) ->
// all-gather of the per-destination counts -> all-to-all-v of the super-k-mers (meta words, then packed nucleotides) -> the owner
// finishes its shard.  A rank whose input is invalid still takes part in every collective and all ranks fail together.
void build_sharded(Index& ix, Comm& c, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len, const StatusError* already_failed)
{
	if (ix.built)
		throw StatusError{BLMX_ESTATE, "index already built"};
	if (ix.shard_world != c.world || ix.shard_rank != c.rank)
		throw StatusError{BLMX_EINVAL, "the index was sharded for another rank / world than this communicator"};
	if ((int)ix.prm.device != c.device)
		throw StatusError{BLMX_EINVAL, "index and communicator live on different devices"};
	std::lock_guard<std::mutex> lock(c.mu);
	BLMX_HIP(hipSetDevice(c.device));
	Rccl& R = rccl();
	hipStream_t s = ix.stream;
	const uint32_t world = c.world, me = c.rank, ngroups = ix.ngroups;
	const size_t nh = (size_t)ngroups + 2;  // histogram, super-k-mer count, failed ranks
	const bool trace = getenv("BLIGHT_BUILD_TRACE") && atoi(getenv("BLIGHT_BUILD_TRACE")) != 0;
	auto t_last = std::chrono::steady_clock::now();
	auto mark = [&](const char* what) {
		if (!trace)
			return;
		(void)hipStreamSynchronize(s);
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[blmx build] rank %u %-21s %8.2f ms\n", me, what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	std::vector<uint64_t> hist(nh, 0);
	StatusError local{BLMX_OK, ""};
	try {
		if (already_failed)  // e.g. this rank could not open its input: it still takes part, so that nobody waits for it
			throw *already_failed;
		shard_build_begin(ix, d_ascii, d_offs, nseq, total_len, hist.data());
	} catch (const StatusError& e) {
		local = e;
		std::fill(hist.begin(), hist.end(), 0);
		hist[nh - 1] = 1;
	}
	mark("exchange: chop slice");
	DevBuf d_hist;
	BLMX_HIP(d_hist.alloc(nh * 8, false, s));
	BLMX_HIP(hipMemcpyAsync(d_hist.p, hist.data(), nh * 8, hipMemcpyHostToDevice, s));
	BLMX_NCCL(R.AllReduce(d_hist.p, d_hist.p, nh, ncclUint64, ncclSum, c.comm, s));
	BLMX_HIP(hipMemcpyAsync(hist.data(), d_hist.p, nh * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	if (hist[nh - 1]) {
		shard_build_abort(ix);
		if (local.code != BLMX_OK)
			throw local;
		throw StatusError{BLMX_ERCCL, "sharded build: the input of another rank was rejected"};
	}
	mark("exchange: all-reduce");
	uint64_t send_sk[16], send_words[16], recv_sk[16], recv_words[16];
	const void *d_meta = nullptr, *d_words = nullptr;
	shard_build_plan(ix, hist.data(), send_sk, send_words, &d_meta, &d_words);
	mark("exchange: pack outgoing");
	std::vector<uint64_t> mine(2 * (size_t)world), all(2 * (size_t)world * world);
	for (uint32_t r = 0; r < world; r++) {
		mine[r] = send_sk[r];
		mine[world + r] = send_words[r];
	}
	DevBuf d_mine, d_all;
	BLMX_HIP(d_mine.alloc(mine.size() * 8, false, s));
	BLMX_HIP(d_all.alloc(all.size() * 8, false, s));
	BLMX_HIP(hipMemcpyAsync(d_mine.p, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, s));
	BLMX_NCCL(R.AllGather(d_mine.p, d_all.p, mine.size(), ncclUint64, c.comm, s));
	BLMX_HIP(hipMemcpyAsync(all.data(), d_all.p, all.size() * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	uint64_t in_sk = 0, in_words = 0;
	for (uint32_t r = 0; r < world; r++) {
		recv_sk[r] = all[(size_t)r * 2 * world + me];
		recv_words[r] = all[(size_t)r * 2 * world + world + me];
		in_sk += recv_sk[r];
		in_words += recv_words[r];
	}
	DevBuf d_in_meta, d_in_words;
	BLMX_HIP(d_in_meta.alloc((in_sk + 1) * 8, false, s));
	BLMX_HIP(d_in_words.alloc((in_words + 1) * 8, false, s));
	all_to_all_v(c, d_meta, send_sk, d_in_meta.p, recv_sk, 1, s);
	all_to_all_v(c, d_words, send_words, d_in_words.p, recv_words, 1, s);
	BLMX_HIP(hipStreamSynchronize(s));
	mark("exchange: all-to-all-v");
	shard_build_finish(ix, d_in_meta.p, d_in_words.p, recv_sk, recv_words);
	ix.built = true;
}

// every rank reads the file (the reader's pair rule needs the lines from the start) and keeps the rank-th consecutive part of its
// sequences: rank r of w takes sequences [nseq * r / w, nseq * (r + 1) / w)
void build_sharded_from_fasta(Index& ix, Comm& c, const char* path)
{
	if (ix.built)
		throw StatusError{BLMX_ESTATE, "index already built"};
	BLMX_HIP(hipSetDevice(c.device));
	DevBuf d_ascii, d_offs;
	std::vector<uint64_t> offs(1, 0);
	uint64_t total = 0;
	StatusError local{BLMX_OK, ""};
	try {
		load_fasta_to_device(ix, path, d_ascii, offs, total);
	} catch (const StatusError& e) {
		local = e;
	}
	if (local.code != BLMX_OK) {
		build_sharded(ix, c, nullptr, nullptr, 0, 0, &local);  // throws `local` after the all-reduce
		throw local;
	}
	const uint64_t nseq = offs.size() - 1;
	const uint64_t u0 = nseq * c.rank / c.world, u1 = nseq * (c.rank + 1) / c.world;
	std::vector<uint64_t> mine(u1 - u0 + 1);
	for (uint64_t i = 0; i <= u1 - u0; i++)
		mine[i] = offs[u0 + i] - offs[u0];
	BLMX_HIP(d_offs.alloc(mine.size() * 8, false, ix.stream));
	BLMX_HIP(hipMemcpyAsync(d_offs.p, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, ix.stream));
	BLMX_HIP(hipStreamSynchronize(ix.stream));
	build_sharded(ix, c, d_ascii.as<uint8_t>() + offs[u0], d_offs.as<uint64_t>(), u1 - u0, mine.back(), nullptr);
}

// host buffers in and out around the device exchange (collective)
void query_kmers_sharded_host(Index& ix, Comm& c, const void* kmers, uint64_t nq, int64_t* ids_out)
{
	BLMX_HIP(hipSetDevice(c.device));
	std::lock_guard<std::mutex> lock(ix.mu);
	hipStream_t s = ix.stream;
	const size_t kb = 8 * (size_t)ix.KW;
	grow(ix.q_in, nq * kb + 16, s);
	grow(ix.q_out, nq * 8 + 16, s);
	if (nq)
		BLMX_HIP(hipMemcpyAsync(ix.q_in.p, kmers, nq * kb, hipMemcpyHostToDevice, s));
	query_kmers_sharded(ix, c, ix.q_in.p, nq, ix.q_out.as<int64_t>(), s);
	if (nq)
		BLMX_HIP(hipMemcpyAsync(ids_out, ix.q_out.p, nq * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
}

}  // namespace blmx
