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

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <random>

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
		// BLIGHT_RCCL_LIB names the library to load (a test points it at a file that does not exist to see BLMX_ERCCL)
		const char* forced = getenv("BLIGHT_RCCL_LIB");
		std::vector<const char*> names;
		if (forced && *forced)
			names = {forced};
		else
			names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		std::string why;
		for (const char* name : names) {
			x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (x.lib)
				break;
			const char* e = dlerror();  // ONE call: dlerror() clears the message it returns
			if (why.empty())
				why = e ? e : "not found";
		}
		if (!x.lib) {
			x.err = std::string("cannot load ") + names[0] + ": " + why;
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

// ---- transports -----------------------------------------------------------------------------------------------------------------------
// What the exchange needs from a communicator: grouped point-to-point flows of u64 words (the all-to-all-v) and two small
// collectives, all enqueued on a HIP stream.  Two implementations: RCCL over xGMI (one process per GPU), and a loopback -- `world`
// communicators inside ONE process, one caller thread each, all on the same GPU -- which moves the flows with device copies under
// the same grouping and stream-ordering rules.  The loopback exists so that the multi-rank choreography (count transposition, piece
// pipeline, multi-round flows, failure votes) runs on a one-GPU box; it is not a performance path.
struct Transport {
	virtual ~Transport() {}
	virtual const char* name() const = 0;
	virtual void group_start() = 0;
	virtual void group_end() = 0;
	virtual void send(const uint64_t* p, uint64_t n, uint32_t peer, hipStream_t s) = 0;
	virtual void recv(uint64_t* p, uint64_t n, uint32_t peer, hipStream_t s) = 0;
	virtual void all_gather(const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t s) = 0;  // out = world x n words, rank order
	virtual void all_reduce_sum(uint64_t* inout, uint64_t n, hipStream_t s) = 0;
};

struct RcclTransport final : Transport {
	ncclComm_t comm = nullptr;
	RcclTransport(const void* id_bytes, uint32_t rank, uint32_t world)
	{
		ncclUniqueId id;
		memcpy(&id, id_bytes, sizeof id);
		BLMX_NCCL(rccl().CommInitRank(&comm, (int)world, id, (int)rank));
	}
	~RcclTransport() override
	{
		if (comm)
			(void)rccl().CommDestroy(comm);
	}
	const char* name() const override { return "rccl"; }
	void group_start() override { BLMX_NCCL(rccl().GroupStart()); }
	void group_end() override { BLMX_NCCL(rccl().GroupEnd()); }
	void send(const uint64_t* p, uint64_t n, uint32_t peer, hipStream_t s) override { BLMX_NCCL(rccl().Send(p, n, ncclUint64, (int)peer, comm, s)); }
	void recv(uint64_t* p, uint64_t n, uint32_t peer, hipStream_t s) override { BLMX_NCCL(rccl().Recv(p, n, ncclUint64, (int)peer, comm, s)); }
	void all_gather(const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t s) override { BLMX_NCCL(rccl().AllGather(in, out, n, ncclUint64, comm, s)); }
	void all_reduce_sum(uint64_t* inout, uint64_t n, hipStream_t s) override
	{
		BLMX_NCCL(rccl().AllReduce(inout, inout, n, ncclUint64, ncclSum, comm, s));
	}
};

// ---- loopback -------------------------------------------------------------------------------------------------------------------------
constexpr char kLoopMagic[16] = {'B', 'L', 'M', 'X', '-', 'L', 'O', 'O', 'P', 'B', 'A', 'C', 'K', '-', '1', 0};

__global__ void k_sum_rows(const uint64_t* __restrict__ rows, uint32_t nrows, uint64_t n, uint64_t* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	uint64_t a = 0;
	for (uint32_t r = 0; r < nrows; r++)
		a += rows[(uint64_t)r * n + i];
	out[i] = a;
}

// Meeting point of the `world` communicators of one loopback id.  A flow is posted by its sender (source pointer + an event that
// says "the data is there") and pulled by its receiver with a device copy on the receiver's stream; the sender's stream then waits
// for the copy before it may touch the buffer again -- the ordering RCCL gives a send / recv pair.  Flows of one (source,
// destination) pair match in FIFO order.  Collectives meet in numbered slots.  Every wait has a deadline: a rank that never
// arrives (it threw before its collective) turns into BLMX_ERCCL on the others instead of a hang.
struct LoopHub {
	uint32_t world = 0;
	std::mutex mu;
	std::condition_variable cv;
	struct Flow {
		const uint64_t* p;
		uint64_t n;
		hipEvent_t ready = nullptr, done = nullptr;
		bool consumed = false;
	};
	std::deque<std::shared_ptr<Flow>> q[16][16];
	struct Coll {
		std::vector<const uint64_t*> in;
		std::vector<hipEvent_t> ready, done;
		uint64_t n = 0;
		uint32_t posted = 0, finished = 0, left = 0;
	};
	std::map<uint64_t, Coll> colls;
	double timeout_s = 120.0;
	bool poisoned = false;  // a wait has timed out: flows and collective slots of the ranks that did arrive are stale from then on

	// (caller holds mu)  A rank that has waited in vain leaves posted flows and a collective sequence number behind that its peers will
	// never match: the hub is dead from here on -- every later call on any of its communicators fails at once with BLMX_ERCCL instead
	// of pulling a stale flow (a pointer its owner may have freed) or waiting out another deadline.
	void poison()
	{
		poisoned = true;
		for (auto& row : q)
			for (auto& d : row)
				d.clear();
		colls.clear();
		cv.notify_all();
	}
	void check(const char* what) const
	{
		if (poisoned)
			throw StatusError{BLMX_ERCCL, std::string("loopback transport: an earlier collective timed out, this communicator group is dead (") + what + ")"};
	}
	template <typename Pred> void wait(std::unique_lock<std::mutex>& l, Pred pred, const char* what)
	{
		check(what);
		if (!cv.wait_for(l, std::chrono::duration<double>(timeout_s), [&] { return poisoned || pred(); }) || poisoned) {
			const bool first = !poisoned;
			poison();
			throw StatusError{BLMX_ERCCL, std::string("loopback transport: ") + (first ? "a peer did not arrive (" : "a peer gave up waiting (") + what + ")"};
		}
	}
};

std::mutex g_hub_mu;
std::map<std::string, std::weak_ptr<LoopHub>> g_hubs;

std::shared_ptr<LoopHub> loop_hub(const void* id_bytes, uint32_t world)
{
	const std::string key(reinterpret_cast<const char*>(id_bytes), BLMX_COMM_ID_BYTES);
	std::lock_guard<std::mutex> l(g_hub_mu);
	std::shared_ptr<LoopHub> h = g_hubs[key].lock();
	if (!h) {
		h = std::make_shared<LoopHub>();
		h->world = world;
		if (const char* e = getenv("BLIGHT_LOOPBACK_TIMEOUT"))
			h->timeout_s = atof(e) > 0 ? atof(e) : h->timeout_s;
		g_hubs[key] = h;
	}
	if (h->world != world)
		throw StatusError{BLMX_EINVAL, "loopback communicators of one id must agree on the world size"};
	return h;
}

struct LoopbackTransport final : Transport {
	std::shared_ptr<LoopHub> hub;
	uint32_t me, world;
	uint64_t coll_seq = 0;
	bool in_group = false;
	struct Op {
		bool is_send;
		uint64_t* p;
		uint64_t n;
		uint32_t peer;
		hipStream_t s;
	};
	std::vector<Op> ops;
	std::vector<hipEvent_t> retired;  // events whose last use has been enqueued; destroyed once they have completed
	DevBuf tmp;

	LoopbackTransport(const void* id_bytes, uint32_t rank, uint32_t w) : hub(loop_hub(id_bytes, w)), me(rank), world(w) {}
	~LoopbackTransport() override
	{
		for (hipEvent_t e : retired)
			(void)hipEventDestroy(e);
	}
	const char* name() const override { return "loopback"; }

	hipEvent_t new_event()
	{
		hipEvent_t e;
		BLMX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		return e;
	}
	void sweep()
	{
		size_t keep = 0;
		for (hipEvent_t e : retired) {
			if (hipEventQuery(e) == hipSuccess)
				(void)hipEventDestroy(e);
			else
				retired[keep++] = e;
		}
		retired.resize(keep);
	}

	void group_start() override
	{
		in_group = true;
		ops.clear();
	}
	void send(const uint64_t* p, uint64_t n, uint32_t peer, hipStream_t s) override
	{
		ops.push_back(Op{true, const_cast<uint64_t*>(p), n, peer, s});
		if (!in_group)
			flush();
	}
	void recv(uint64_t* p, uint64_t n, uint32_t peer, hipStream_t s) override
	{
		ops.push_back(Op{false, p, n, peer, s});
		if (!in_group)
			flush();
	}
	void group_end() override
	{
		in_group = false;
		flush();
	}
	// all sends are posted before any receive waits for its peer: every rank does the same, so nobody waits for a post that is
	// stuck behind a wait
	void flush()
	{
		sweep();
		std::vector<std::pair<std::shared_ptr<LoopHub::Flow>, hipStream_t>> mine;
		for (const Op& o : ops) {
			if (!o.is_send)
				continue;
			auto f = std::make_shared<LoopHub::Flow>();
			f->p = o.p;
			f->n = o.n;
			f->ready = new_event();
			BLMX_HIP(hipEventRecord(f->ready, o.s));
			{
				std::lock_guard<std::mutex> l(hub->mu);
				hub->check("post of a flow");
				hub->q[me][o.peer].push_back(f);
			}
			hub->cv.notify_all();
			mine.emplace_back(f, o.s);
		}
		for (const Op& o : ops) {
			if (o.is_send)
				continue;
			std::shared_ptr<LoopHub::Flow> f;
			{
				std::unique_lock<std::mutex> l(hub->mu);
				hub->wait(l, [&] { return !hub->q[o.peer][me].empty(); }, "send of a flow");
				f = hub->q[o.peer][me].front();
				hub->q[o.peer][me].pop_front();
			}
			if (f->n != o.n) {
				std::lock_guard<std::mutex> l(hub->mu);
				hub->poison();
				throw StatusError{BLMX_ERCCL, "loopback transport: send and receive disagree on the length of a flow"};
			}
			BLMX_HIP(hipStreamWaitEvent(o.s, f->ready, 0));
			if (o.n)
				BLMX_HIP(hipMemcpyAsync(o.p, f->p, o.n * 8, hipMemcpyDeviceToDevice, o.s));
			hipEvent_t done = new_event();
			BLMX_HIP(hipEventRecord(done, o.s));
			{
				std::lock_guard<std::mutex> l(hub->mu);
				f->done = done;
				f->consumed = true;
			}
			hub->cv.notify_all();
		}
		for (auto& fs : mine) {
			{
				std::unique_lock<std::mutex> l(hub->mu);
				hub->wait(l, [&] { return fs.first->consumed; }, "receive of a flow");
			}
			BLMX_HIP(hipStreamWaitEvent(fs.second, fs.first->done, 0));
			retired.push_back(fs.first->ready);
			retired.push_back(fs.first->done);
		}
		ops.clear();
	}

	void all_gather(const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t s) override
	{
		sweep();
		const uint64_t seq = coll_seq++;
		hipEvent_t ready = new_event();
		BLMX_HIP(hipEventRecord(ready, s));
		std::vector<const uint64_t*> src;
		std::vector<hipEvent_t> evs;
		{
			std::unique_lock<std::mutex> l(hub->mu);
			LoopHub::Coll& c = hub->colls[seq];
			if (c.in.empty()) {
				c.in.assign(world, nullptr);
				c.ready.assign(world, nullptr);
				c.done.assign(world, nullptr);
				c.n = n;
			}
			if (c.n != n)
				throw StatusError{BLMX_ERCCL, "loopback transport: ranks disagree on the size of a collective"};
			c.in[me] = in;
			c.ready[me] = ready;
			c.posted++;
			hub->cv.notify_all();
			hub->wait(l, [&] { return hub->colls[seq].posted == world; }, "collective");
			src = hub->colls[seq].in;
			evs = hub->colls[seq].ready;
		}
		for (uint32_t r = 0; r < world; r++) {
			BLMX_HIP(hipStreamWaitEvent(s, evs[r], 0));
			if (n)
				BLMX_HIP(hipMemcpyAsync(out + (uint64_t)r * n, src[r], n * 8, hipMemcpyDeviceToDevice, s));
		}
		hipEvent_t done = new_event();
		BLMX_HIP(hipEventRecord(done, s));
		{
			std::unique_lock<std::mutex> l(hub->mu);
			LoopHub::Coll& c = hub->colls[seq];
			c.done[me] = done;
			c.finished++;
			hub->cv.notify_all();
			hub->wait(l, [&] { return hub->colls[seq].finished == world; }, "end of a collective");
			evs = hub->colls[seq].done;
		}
		for (uint32_t r = 0; r < world; r++)  // nobody overwrites its contribution before everybody has copied it
			BLMX_HIP(hipStreamWaitEvent(s, evs[r], 0));
		{
			// every rank waits on every rank's events: they are retired by the LAST rank to have enqueued its waits (a rank that retired
			// its own events on leaving could destroy them -- in the sweep of its next call -- before a slower peer has used them)
			std::lock_guard<std::mutex> l(hub->mu);
			LoopHub::Coll& c = hub->colls[seq];
			if (++c.left == world) {
				for (uint32_t r = 0; r < world; r++) {
					retired.push_back(c.ready[r]);
					retired.push_back(c.done[r]);
				}
				hub->colls.erase(seq);
			}
		}
	}

	void all_reduce_sum(uint64_t* inout, uint64_t n, hipStream_t s) override
	{
		if ((size_t)world * n * 8 + 16 > tmp.bytes) {
			BLMX_HIP(hipStreamSynchronize(s));
			BLMX_HIP(tmp.alloc((size_t)world * n * 8 + 16, false, s));
		}
		all_gather(inout, tmp.as<uint64_t>(), n, s);
		if (n)
			hipLaunchKernelGGL(k_sum_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tmp.as<uint64_t>(), world, n, inout);
		BLMX_HIP(hipGetLastError());
	}
};

}  // namespace

struct Comm {
	std::unique_ptr<Transport> tp;
	uint32_t rank = 0, world = 1;
	int device = 0;
	// A pairwise flow larger than flow_words u64 words is cut into rounds (both ends of a flow know its length, so they cut it alike;
	// ranks whose flows are shorter simply take part in fewer rounds).  See kFlowWords.
	uint64_t flow_words;
	int pieces = 0;            // pieces of a query batch (software pipeline); 0 = BLIGHT_DIST_CHUNKS or 2.  The same on every rank.
	// Do the minimizers of a slot range travel with its k-mers (4 of the 12 bytes a slot costs on the wire), or does the owner hash the
	// k-mers it receives again (query_slot without a minimizer array: ~340 vector instructions per k-mer inside a kernel that waits for
	// memory)?  -1 / 0: they stay home -- even on the loopback transport, whose wire is a device copy, the exchange is faster without
	// them at every world size (C4, world 2 / 4 / 8: 6.13 / 6.85 / 8.9 ms per step against 6.35 / 7.15 / 9.8,
	// profiles/r04_loopback_rehearsal_c4.txt), and a real link is slower than a device copy.  1: they travel.  The same on every rank
	// (checked in the count exchange).
	int ship_minis = -1;
	hipStream_t cs = nullptr;  // the exchange runs on this stream, next to the caller's compute stream
	hipEvent_t ev_part[kMaxPieces] = {}, ev_k[kMaxPieces] = {}, ev_q[kMaxPieces] = {}, ev_i[kMaxPieces] = {};
	DevBuf sorted[kMaxPieces], inv[kMaxPieces], recv_k[kMaxPieces], recv_m[kMaxPieces], recv_h[kMaxPieces], ids_remote[kMaxPieces], ids_sorted[kMaxPieces];
	SlotPiece sp[kMaxPieces];  // a piece in the slot format (one-pass partition into bin regions)
	DevBuf counts, small;
	Pinned h_dense;
	std::mutex mu;
	~Comm()
	{
		(void)hipSetDevice(device);
		(void)hipDeviceSynchronize();
		for (auto* arr : {ev_part, ev_k, ev_q, ev_i})
			for (int i = 0; i < kMaxPieces; i++)
				if (arr[i])
					(void)hipEventDestroy(arr[i]);
		tp.reset();
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

void comm_loopback_id(void* id_out)
{
	char* o = reinterpret_cast<char*>(id_out);
	memset(o, 0, BLMX_COMM_ID_BYTES);
	memcpy(o, kLoopMagic, sizeof kLoopMagic);
	std::random_device rd;
	static std::atomic<uint64_t> serial{0};
	const uint64_t a = ((uint64_t)rd() << 32) ^ rd(), b = serial.fetch_add(1) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
	memcpy(o + 16, &a, 8);
	memcpy(o + 24, &b, 8);
}

// All-to-all-v flows are cut into rounds of at most 256 MiB.  Root cause of round 2's "one 1.7-GB flow delivered wrong data": it is
// not our offset arithmetic.  tools/rccl_flow_probe.hip (no code of this library: one grouped ncclSend / ncclRecv pair of a rank to
// ITSELF, pattern in, compare on the device) shows on RCCL 2.27.7 / ROCm 7.2 that a self flow of MORE THAN 2^30 BYTES arrives only
// in its first half -- 1 GiB: intact; 1.5 GB: every word from byte 750,256,128 on is untouched; 1.9 GB, 2^31 - 8, 2^31, 2.4 GB,
// 3 GiB, 4 GiB + 64: the same, always from (about) the middle -- while the same bytes sent as 256-MiB rounds arrive intact at every
// size (profiles/r03_rccl_flow_probe.jsonl).  The world-1 rehearsal of the construction from slices at C4 size sent 1.9 GB of meta
// words and 2.4 GB of packed nucleotides to itself as one message each: that is this.  Whether flows between DIFFERENT ranks are
// affected cannot be seen on a one-GPU box, so nothing here ever sends a message above 256 MiB; blmx_comm_set_tuning("flow_words")
// shrinks the rounds so that the tests walk through the multi-round branch with ragged last rounds.
constexpr uint64_t kFlowWords = 1ull << 25;

Comm* comm_create(const void* id_bytes, uint32_t rank, uint32_t world, int device)
{
	if (world == 0 || world > 16 || rank >= world)
		throw StatusError{BLMX_EINVAL, "need rank < world <= 16"};
	BLMX_HIP(hipSetDevice(device));
	std::unique_ptr<Comm> c(new Comm());
	c->rank = rank;
	c->world = world;
	c->device = device;
	c->flow_words = kFlowWords;
	if (memcmp(id_bytes, kLoopMagic, sizeof kLoopMagic) == 0)
		c->tp.reset(new LoopbackTransport(id_bytes, rank, world));
	else
		c->tp.reset(new RcclTransport(id_bytes, rank, world));
	BLMX_HIP(hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking));
	for (auto* arr : {c->ev_part, c->ev_k, c->ev_q, c->ev_i})
		for (int i = 0; i < kMaxPieces; i++)
			BLMX_HIP(hipEventCreateWithFlags(&arr[i], hipEventDisableTiming));
	return c.release();
}

void comm_destroy(Comm* c) { delete c; }

const char* comm_transport(const Comm* c) { return c->tp->name(); }

void comm_set_tuning(Comm& c, const char* key, int64_t value)
{
	const std::string k(key);
	if (k == "flow_words") {
		if (value < 1)
			throw StatusError{BLMX_EINVAL, "flow_words must be positive"};
		c.flow_words = (uint64_t)value;
	} else if (k == "pieces") {
		c.pieces = value < 0 ? 0 : (value > kMaxPieces ? kMaxPieces : (int)value);
	} else if (k == "ship_minimizers") {
		c.ship_minis = value < 0 ? -1 : (value ? 1 : 0);
	} else {
		throw StatusError{BLMX_EINVAL, "unknown communicator tuning key"};
	}
}

// all-to-all-v as grouped point-to-point transfers; counts in elements of `words` uint64 each
static void all_to_all_v(Comm& c, const void* send, const uint64_t* send_cnt, void* recv, const uint64_t* recv_cnt, size_t words, hipStream_t s)
{
	Transport& T = *c.tp;
	const uint64_t fw = c.flow_words;
	uint64_t so[17], ro[17], rounds = 0;
	so[0] = ro[0] = 0;
	for (uint32_t r = 0; r < c.world; r++) {
		so[r + 1] = so[r] + send_cnt[r] * words;
		ro[r + 1] = ro[r] + recv_cnt[r] * words;
		rounds = std::max(rounds, (std::max(send_cnt[r], recv_cnt[r]) * words + fw - 1) / fw);
	}
	for (uint64_t j = 0; j < rounds; j++) {
		const uint64_t off = j * fw;
		T.group_start();
		for (uint32_t r = 0; r < c.world; r++) {
			const uint64_t sn = send_cnt[r] * words, rn = recv_cnt[r] * words;
			if (off < sn)
				T.send((const uint64_t*)send + so[r] + off, std::min(fw, sn - off), r, s);
			if (off < rn)
				T.recv((uint64_t*)recv + ro[r] + off, std::min(fw, rn - off), r, s);
		}
		T.group_end();
	}
}

// the same with explicit offsets (in u64 words; nullptr = packed in rank order) and counts in u64 words: slot ranges travel from
// where they lie in the slot arrays, and the IDs come back into them
static void all_to_all_v_words(Comm& c, const void* send, const uint64_t* send_off, const uint64_t* send_words, void* recv, const uint64_t* recv_off,
                               const uint64_t* recv_words, hipStream_t s)
{
	Transport& T = *c.tp;
	const uint64_t fw = c.flow_words;
	uint64_t so[17], ro[17], rounds = 0;
	so[0] = ro[0] = 0;
	for (uint32_t r = 0; r < c.world; r++) {
		so[r + 1] = so[r] + send_words[r];
		ro[r + 1] = ro[r] + recv_words[r];
		rounds = std::max(rounds, (std::max(send_words[r], recv_words[r]) + fw - 1) / fw);
	}
	for (uint64_t j = 0; j < rounds; j++) {
		const uint64_t off = j * fw;
		T.group_start();
		for (uint32_t r = 0; r < c.world; r++) {
			if (off < send_words[r])
				T.send((const uint64_t*)send + (send_off ? send_off[r] : so[r]) + off, std::min(fw, send_words[r] - off), r, s);
			if (off < recv_words[r])
				T.recv((uint64_t*)recv + (recv_off ? recv_off[r] : ro[r]) + off, std::min(fw, recv_words[r] - off), r, s);
		}
		T.group_end();
	}
}

// a few host words summed / gathered over the ranks (failure votes, counts, halo records): staged through a small device buffer
static void all_reduce_host(Comm& c, uint64_t* v, size_t n, hipStream_t s)
{
	grow(c.small, n * 8 + 16, s);
	BLMX_HIP(hipMemcpyAsync(c.small.p, v, n * 8, hipMemcpyHostToDevice, s));
	c.tp->all_reduce_sum(c.small.as<uint64_t>(), n, s);
	BLMX_HIP(hipMemcpyAsync(v, c.small.p, n * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
}
static void all_gather_host(Comm& c, const uint64_t* mine, size_t n, uint64_t* all, hipStream_t s)
{
	grow(c.small, (size_t)(c.world + 1) * n * 8 + 16, s);
	uint64_t* d_mine = c.small.as<uint64_t>();
	uint64_t* d_all = d_mine + n;
	BLMX_HIP(hipMemcpyAsync(d_mine, mine, n * 8, hipMemcpyHostToDevice, s));
	c.tp->all_gather(d_mine, d_all, n, s);
	BLMX_HIP(hipMemcpyAsync(all, d_all, (size_t)c.world * n * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
}

// what must hold before a rank may join a collective on (ix, c): returned, not thrown, so that the caller can still take part and
// fail together with the others instead of leaving them waiting
static StatusError precondition(const Index& ix, const Comm& c, bool want_built)
{
	if (want_built && !ix.built)
		return StatusError{BLMX_ESTATE, "index not built"};
	if (!want_built && ix.built)
		return StatusError{BLMX_ESTATE, "index already built"};
	if (ix.shard_world != c.world || ix.shard_rank != c.rank)
		return StatusError{BLMX_EINVAL, "the index was sharded for another rank / world than this communicator"};
	if ((int)ix.prm.device != c.device)
		return StatusError{BLMX_EINVAL, "index and communicator live on different devices"};
	return StatusError{BLMX_OK, ""};
}

// IDs of nq canonical k-mers in HBM, every k-mer answered by the rank that owns its MPHF group; collective: every rank of the
// communicator calls it (with its own batch, possibly empty) in the same order.  A rank that cannot take part (index not built,
// sharded for another world) still joins the count exchange with a flag, and all ranks fail together.
//
// Every piece of the batch leaves its sender in one of two formats, chosen by the sender alone and announced with the counts:
//   slots  the single-GPU pipeline's one-pass partition (bin regions sized after the previous batch, bins split at the shard
//          boundaries): owner r gets the slot range of its bins as it is -- k-mers, their minimizers, one header word per bin -- and
//          returns one ID word per slot (4 B when every ID fits); what this rank owns itself is answered in place, without a flow;
//   dense  the exact counting sort (first batch of an index, small pieces, a piece whose regions overflowed): per-owner segments
//          without gaps, 8-byte IDs back.
// d_minis (may be null): the minimizers of the k-mers when the caller has them (the read front end).
void query_kmers_sharded(Index& ix, Comm& c, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, int64_t* d_ids, hipStream_t s)
{
	std::lock_guard<std::mutex> lock(c.mu);
	BLMX_HIP(hipSetDevice(c.device));
	StatusError local = precondition(ix, c, true);
	if (local.code == BLMX_OK && nq >= 0xffffffffull)
		local = StatusError{BLMX_ELIMIT, "a partitioned batch holds fewer than 2^32 - 1 k-mers"};
	const uint32_t world = c.world, me = c.rank;
	// BLIGHT_TRACE=3: host wall-clock of the phases of one exchange step, per rank (what the rehearsals under profiles/ print)
	static const bool trace = getenv("BLIGHT_TRACE") && atoi(getenv("BLIGHT_TRACE")) == 3;
	auto t_last = std::chrono::steady_clock::now();
	double ph[6] = {0, 0, 0, 0, 0, 0};
	auto phase = [&](int i) {
		if (!trace)
			return;
		const auto now = std::chrono::steady_clock::now();
		ph[i] += std::chrono::duration<double, std::milli>(now - t_last).count();
		t_last = now;
	};
	const char* e = getenv("BLIGHT_DIST_CHUNKS");
	// pieces overlap the exchange of one with the queries of another; with one rank there is nothing to overlap, and a piece of
	// half the batch loses locality in its bins.  Must be the same on every rank.
	int pieces = c.pieces ? c.pieces : (e ? atoi(e) : (c.world > 1 ? 2 : 1));
	pieces = pieces < 1 ? 1 : (pieces > kMaxPieces ? kMaxPieces : pieces);
	const size_t kw = (size_t)ix.KW;
	uint64_t cut[kMaxPieces + 1];
	for (int p = 0; p <= pieces; p++)
		cut[p] = local.code == BLMX_OK ? nq * (uint64_t)p / (uint64_t)pieces : 0;
	// ---- partition every piece
	bool slots_fmt[kMaxPieces] = {};
	grow(c.counts, (size_t)pieces * world * 8, s);
	c.h_dense.ensure((size_t)pieces * world * 8);
	auto partition_dense = [&](int p) {
		const uint64_t n = cut[p + 1] - cut[p];
		grow(c.sorted[p], n * 8 * kw + 16, s);
		grow(c.inv[p], n * 4 + 16, s);
		grow(c.ids_sorted[p], n * 8 + 16, s);  // (everything sized by this rank's own batch is allocated in front of the vote)
		launch_partition(ix, (const uint64_t*)d_kmers + cut[p] * kw, n, c.sorted[p].p, c.inv[p].as<uint32_t>(), c.counts.as<uint64_t>() + (size_t)p * world, s);
		BLMX_HIP(hipMemcpyAsync(c.h_dense.as<uint64_t>() + (size_t)p * world, c.counts.as<uint64_t>() + (size_t)p * world, 8 * (size_t)world, hipMemcpyDeviceToHost, s));
		BLMX_HIP(hipEventRecord(c.ev_part[p], s));
	};
	if (local.code == BLMX_OK) try {  // (a rank whose partition fails -- out of memory, a HIP error -- still joins the all-gather below, with its flag set)
		for (int p = 0; p < pieces; p++) {
			const uint64_t n = cut[p + 1] - cut[p];
			slots_fmt[p] = slot_piece_possible(ix, n);
			if (slots_fmt[p]) {
				slot_piece_partition(ix, c.sp[p], (const uint64_t*)d_kmers + cut[p] * kw, d_minis ? d_minis + cut[p] : nullptr, n, s);
				BLMX_HIP(hipEventRecord(c.ev_part[p], s));
			} else {
				partition_dense(p);
			}
		}
		BLMX_HIP(hipStreamSynchronize(s));
		bool again = false;
		for (int p = 0; p < pieces; p++)
			if (slots_fmt[p] && c.sp[p].overflow()) {  // the regions did not hold this piece (a batch unlike its predecessor): exact partition instead
				slots_fmt[p] = false;
				partition_dense(p);
				again = true;
			}
		if (again)
			BLMX_HIP(hipStreamSynchronize(s));
	} catch (const StatusError& err) {
		local = err;
	}
	phase(0);  // partition of all pieces + the wait for their per-owner counts
	// ---- formats and per-owner counts of all pieces, what every rank can receive without growing a buffer, the settings that size
	// the flows (ID width, number of pieces: per-rank settings that must agree) and one "cannot take part" flag in one all-gather
	const bool narrow = ix.tune_narrow_ids && ix.st.number_kmer < 0xffffffffull;
	const size_t idw = narrow ? 1 : 2;  // 32-bit words per slot of a returned slot range
	const char* e_ship = getenv("BLIGHT_DIST_SHIP_MINIMIZERS");
	const int ship_set = c.ship_minis >= 0 ? c.ship_minis : (e_ship ? (atoi(e_ship) != 0) : -1);
	const bool ship = ship_set > 0;
	// (the table has room for kMaxPieces pieces whatever `pieces` is here: ranks that disagree on it must still gather tables of ONE size --
	// an all-gather whose ranks name different lengths is a hang on RCCL, not an error)
	const size_t per_piece = (size_t)world + 1 + 4, per_rank = (size_t)kMaxPieces * per_piece + 4;
	std::vector<uint64_t> mine(per_rank, 0), all(per_rank * world, 0);
	for (int p = 0; p < pieces; p++) {
		uint64_t* mp = &mine[(size_t)p * per_piece];
		if (local.code == BLMX_OK) {
			mp[0] = slots_fmt[p] ? 1 : 0;
			for (uint32_t r = 0; r < world; r++)
				mp[1 + r] = slots_fmt[p] ? (uint64_t)(c.sp[p].seg()[r + 1] - c.sp[p].seg()[r]) : c.h_dense.as<uint64_t>()[(size_t)p * world + r];
		}
		mp[world + 1] = c.recv_k[p].bytes;  // capacities of this rank's receive buffers: every rank can tell whether somebody has to grow one
		mp[world + 2] = c.recv_m[p].bytes;
		mp[world + 3] = c.recv_h[p].bytes;
		mp[world + 4] = c.ids_remote[p].bytes;
	}
	mine[per_rank - 4] = ship ? 1 : 0;
	mine[per_rank - 3] = narrow ? 1 : 0;
	mine[per_rank - 2] = (uint64_t)pieces;
	mine[per_rank - 1] = local.code != BLMX_OK;
	all_gather_host(c, mine.data(), per_rank, all.data(), s);
	phase(1);  // the all-gather of counts (includes waiting for the slowest rank's partition)
	for (uint32_t r = 0; r < world; r++)
		if (all[(size_t)r * per_rank + per_rank - 1]) {
			if (local.code != BLMX_OK)
				throw local;
			throw StatusError{BLMX_ERCCL, "sharded query: another rank could not take part"};
		}
	for (uint32_t r = 0; r < world; r++)  // (every rank sees the same table: all of them give up together, before any flow is posted)
		if (all[(size_t)r * per_rank + per_rank - 3] != (narrow ? 1u : 0u) || all[(size_t)r * per_rank + per_rank - 2] != (uint64_t)pieces ||
		    all[(size_t)r * per_rank + per_rank - 4] != (ship ? 1u : 0u))
			throw StatusError{BLMX_EINVAL, "sharded query: the ranks disagree on the ID width (narrow_ids), on the number of pieces (BLIGHT_DIST_CHUNKS / comm tuning) "
			                               "or on whether minimizers travel (BLIGHT_DIST_SHIP_MINIMIZERS / comm tuning)"};
	auto fmt_of = [&](uint32_t r, int p) { return all[(size_t)r * per_rank + (size_t)p * per_piece] != 0; };
	auto cnt_of = [&](uint32_t r, int p, uint32_t d) { return all[(size_t)r * per_rank + (size_t)p * per_piece + 1 + d]; };
	auto cap_of = [&](uint32_t r, int p, int which) { return all[(size_t)r * per_rank + (size_t)p * per_piece + world + 1 + which]; };
	struct Flows {
		uint64_t ks[16], kr[16], ms[16], mr[16], hs[16], hr[16], is[16], ir[16];  // k-mers / minimizers / headers out and in, IDs back out and in (counts)
		uint64_t recv_k_words, recv_m_words, recv_h_words, ids_words;
	} F[kMaxPieces];
	// header words a sender has for owner d: one per bin of d plus the end of its last region; the bin geometry follows from the shard
	// boundaries and the index parameters, so it is the same on every rank (also on one that sends densely itself)
	for (int p = 0; p < pieces; p++) {
		for (uint32_t d = 0; d < world; d++) {
			uint64_t to_d = 0;  // every rank sees the same matrix: all of them give up together
			for (uint32_t r = 0; r < world; r++)
				to_d += cnt_of(r, p, d);
			if (to_d >= 0xffffffffull)
				throw StatusError{BLMX_ELIMIT, "an owner receives fewer than 2^32 - 1 k-mers (or slots) per piece"};
		}
	}
	// the flows of piece p as rank `who` sees them (every rank can work them out for every rank: the table is the same everywhere)
	auto flows_of = [&](uint32_t who, int p, Flows& f) {
		const uint32_t who_nb = slot_owner_bins(ix, who);
		f.recv_k_words = f.recv_m_words = f.recv_h_words = f.ids_words = 0;
		for (uint32_t r = 0; r < world; r++) {
			const bool out_slots = fmt_of(who, p), in_slots = fmt_of(r, p);
			const uint64_t out_n = cnt_of(who, p, r), in_n = cnt_of(r, p, who);
			const bool self = r == who;
			// out: what `who` sends to owner r
			f.ks[r] = (out_slots && self) ? 0 : out_n;                               // k-mers (slots or dense), KW words each
			f.ms[r] = (ship && out_slots && !self) ? out_n / 2 : 0;                  // minimizers, two per word (ship: they travel at all)
			f.hs[r] = (out_slots && !self) ? (uint64_t)slot_owner_bins(ix, r) + 1 : 0;
			f.ir[r] = out_slots ? (self ? 0 : out_n * idw / 2) : out_n;              // IDs that come back from owner r (words)
			// in: what sender r holds for `who`
			f.kr[r] = (in_slots && self) ? 0 : in_n;
			f.mr[r] = (ship && in_slots && !self) ? in_n / 2 : 0;
			f.hr[r] = (in_slots && !self) ? (uint64_t)who_nb + 1 : 0;
			f.is[r] = in_slots ? (self ? 0 : in_n * idw / 2) : in_n;
			f.recv_k_words += f.kr[r] * kw;
			f.recv_m_words += f.mr[r];
			f.recv_h_words += f.hr[r];
			f.ids_words += f.is[r];
		}
	};
	// Somebody has to grow a receive buffer (first batches, a batch larger than its predecessors)?  Then an allocation can fail on one
	// rank alone AFTER the vote above -- its peers would already be inside the grouped send / receive, and RCCL has no timeout -- so
	// such a batch takes a second, one-word vote behind the allocations.  In the steady state nobody grows and nobody votes twice.
	const uint32_t my_nb = slot_owner_bins(ix, me);
	bool somebody_grows = false;
	for (uint32_t r = 0; r < world && !somebody_grows; r++)
		for (int p = 0; p < pieces; p++) {
			Flows f;
			flows_of(r, p, f);
			if (f.recv_k_words * 8 + 16 > cap_of(r, p, 0) || f.recv_m_words * 8 + 16 > cap_of(r, p, 1) || f.recv_h_words * 8 + 16 > cap_of(r, p, 2) ||
			    f.ids_words * 8 + 16 > cap_of(r, p, 3))
				somebody_grows = true;
		}
	try {
		for (int p = 0; p < pieces; p++) {
			Flows& f = F[p];
			flows_of(me, p, f);
			grow(c.recv_k[p], f.recv_k_words * 8 + 16, s);
			grow(c.recv_m[p], f.recv_m_words * 8 + 16, s);
			grow(c.recv_h[p], f.recv_h_words * 8 + 16, s);
			grow(c.ids_remote[p], f.ids_words * 8 + 16, s);
		}
		BLMX_HIP(hipStreamSynchronize(s));  // buffers that grew must exist before the other stream touches them
	} catch (const StatusError& err) {
		if (!somebody_grows)
			throw;  // (not an allocation: the device is gone, and so is the communicator)
		local = err;
	}
	if (somebody_grows) {
		uint64_t my_vote = local.code != BLMX_OK;
		std::vector<uint64_t> votes(world, 0);
		all_gather_host(c, &my_vote, 1, votes.data(), s);
		for (uint32_t r = 0; r < world; r++)
			if (votes[r]) {
				if (local.code != BLMX_OK)
					throw local;
				throw StatusError{BLMX_ERCCL, "sharded query: another rank could not allocate its receive buffers"};
			}
	}
	phase(2);  // receive buffers (+ the second vote when somebody grew)
	// ---- the pipeline: collectives are issued in the same order on every rank (k-mers 0, k-mers 1, IDs 0, k-mers 2, IDs 1, ...)
	auto send_kmers = [&](int p) {
		const Flows& f = F[p];
		BLMX_HIP(hipStreamWaitEvent(c.cs, c.ev_part[p], 0));
		uint64_t kso[16], ksw[16], krw[16], mso[16], hso[16];
		for (uint32_t r = 0; r < world; r++) {
			ksw[r] = f.ks[r] * kw;
			krw[r] = f.kr[r] * kw;
			kso[r] = mso[r] = hso[r] = 0;
			if (slots_fmt[p]) {  // the slot range of owner r lies where the partition put it (my own range does not travel)
				kso[r] = (uint64_t)c.sp[p].seg()[r] * kw;
				mso[r] = (uint64_t)c.sp[p].seg()[r] / 2;
				hso[r] = c.sp[p].first_bin[r];
			}
		}
		if (slots_fmt[p]) {
			all_to_all_v_words(c, c.sp[p].slots.p, kso, ksw, c.recv_k[p].p, nullptr, krw, c.cs);
			all_to_all_v_words(c, c.sp[p].minis.p, mso, f.ms, c.recv_m[p].p, nullptr, f.mr, c.cs);
			all_to_all_v_words(c, c.sp[p].hdr.p, hso, f.hs, c.recv_h[p].p, nullptr, f.hr, c.cs);
		} else {
			all_to_all_v_words(c, c.sorted[p].p, nullptr, ksw, c.recv_k[p].p, nullptr, krw, c.cs);
			all_to_all_v_words(c, nullptr, nullptr, f.ms, c.recv_m[p].p, nullptr, f.mr, c.cs);  // (f.ms / f.hs are all zero for a dense sender)
			all_to_all_v_words(c, nullptr, nullptr, f.hs, c.recv_h[p].p, nullptr, f.hr, c.cs);
		}
		BLMX_HIP(hipEventRecord(c.ev_k[p], c.cs));
	};
	send_kmers(0);
	for (int p = 0; p < pieces; p++) {
		const Flows& f = F[p];
		if (p + 1 < pieces)
			send_kmers(p + 1);  // the next piece travels while this one is answered
		BLMX_HIP(hipStreamWaitEvent(s, c.ev_k[p], 0));
		// the owner's side: slot ranges (mine in place, the others' from the receive buffers) in one launch, dense segments one by one
		ForeignSlots fsrc[16];
		uint32_t nsrc = 0;
		uint64_t ko = 0, mo = 0, ho = 0, io = 0;  // word offsets of source r in the receive buffers / in the IDs that go back
		for (uint32_t r = 0; r < world; r++) {
			const uint64_t in_n = cnt_of(r, p, me);
			if (fmt_of(r, p)) {
				ForeignSlots fs;
				if (r == me) {
					const uint32_t s0 = c.sp[p].seg()[me];
					fs.slots = c.sp[p].slots.as<uint64_t>() + (size_t)s0 * kw;
					fs.minis = c.sp[p].minis.as<uint32_t>() + s0;
					fs.ids = c.sp[p].ids.as<char>() + (size_t)s0 * 4 * idw;
					fs.hdr = c.sp[p].hdr.as<uint64_t>() + c.sp[p].first_bin[me];
				} else {
					fs.slots = c.recv_k[p].as<uint64_t>() + ko;
					fs.minis = ship ? reinterpret_cast<const uint32_t*>(c.recv_m[p].as<uint64_t>() + mo) : nullptr;  // (null: the owner hashes the k-mers again)
					fs.ids = c.ids_remote[p].as<uint64_t>() + io;
					fs.hdr = c.recv_h[p].as<uint64_t>() + ho;
				}
				fs.nslots = (uint32_t)in_n;
				fs.nb = my_nb;
				fsrc[nsrc++] = fs;
			} else if (in_n) {
				launch_query_partitioned(ix, c.recv_k[p].as<uint64_t>() + ko, in_n, reinterpret_cast<int64_t*>(c.ids_remote[p].as<uint64_t>() + io), s);
			}
			ko += f.kr[r] * kw;
			mo += f.mr[r];
			ho += f.hr[r];
			io += f.is[r];
		}
		launch_query_foreign_slots(ix, fsrc, nsrc, narrow, s);
		BLMX_HIP(hipEventRecord(c.ev_q[p], s));
		BLMX_HIP(hipStreamWaitEvent(c.cs, c.ev_q[p], 0));
		if (slots_fmt[p]) {  // the IDs of owner r's range come back into the range itself
			uint64_t ioff[16];
			for (uint32_t r = 0; r < world; r++)
				ioff[r] = (uint64_t)c.sp[p].seg()[r] * idw / 2;
			all_to_all_v_words(c, c.ids_remote[p].p, nullptr, f.is, c.sp[p].ids.p, ioff, f.ir, c.cs);
		} else {
			all_to_all_v_words(c, c.ids_remote[p].p, nullptr, f.is, c.ids_sorted[p].p, nullptr, f.ir, c.cs);
		}
		BLMX_HIP(hipEventRecord(c.ev_i[p], c.cs));
	}
	for (int p = 0; p < pieces; p++) {
		BLMX_HIP(hipStreamWaitEvent(s, c.ev_i[p], 0));
		if (slots_fmt[p])
			slot_piece_gather(ix, c.sp[p], d_ids + cut[p], s);
		else
			launch_gather_ids(ix, c.ids_sorted[p].as<int64_t>(), c.inv[p].as<uint32_t>(), cut[p + 1] - cut[p], d_ids + cut[p], s);
	}
	ix.number_query.fetch_add(nq, std::memory_order_relaxed);
	phase(3);  // flows, owner kernels and gathers ENQUEUED (the device works on them from here on)
	if (trace) {
		BLMX_HIP(hipStreamSynchronize(s));
		phase(4);
		fprintf(stderr, "[blmx exchange] rank %u of %u, %llu k-mers in %d pieces: partition+counts %.3f ms, all-gather %.3f, buffers %.3f, enqueue %.3f, device drain %.3f\n", me,
		        world, (unsigned long long)nq, pieces, ph[0], ph[1], ph[2], ph[3], ph[4]);
	}
}

// Sharded construction from slices: every rank passes ITS OWN consecutive piece of the input (rank order = input order).  Collective.
// Steps: chop locally -> all-reduce of the per-group key histograms (every rank derives the same key-balanced boundaries) ->
// all-gather of the per-destination counts -> all-to-all-v of the super-k-mers (meta words, then packed nucleotides) -> the owner
// finishes its shard -> all-gather of the heads of the shards' sequences: every shard keeps, behind its own sequence, what the single
// index holds there (shard_halo_set).  A rank whose input is invalid -- or whose index / communicator do not fit -- still takes
// part in the first all-reduce with a failure flag, and all ranks fail together.
void build_sharded(Index& ix, Comm& c, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len, const StatusError* already_failed,
                   DevBuf* packed_text)
{
	std::lock_guard<std::mutex> lock(c.mu);
	BLMX_HIP(hipSetDevice(c.device));
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
	StatusError local = already_failed ? *already_failed : precondition(ix, c, false);  // e.g. this rank could not open its input: it still takes part
	if (local.code == BLMX_OK) {
		try {
			shard_build_begin(ix, d_ascii, d_offs, nseq, total_len, hist.data(), packed_text);
		} catch (const StatusError& e) {
			local = e;
		}
	}
	if (local.code != BLMX_OK) {
		std::fill(hist.begin(), hist.end(), 0);
		hist[nh - 1] = 1;
	}
	mark("exchange: chop slice");
	all_reduce_host(c, hist.data(), nh, s);
	if (hist[nh - 1]) {
		shard_build_abort(ix);
		if (local.code != BLMX_OK)
			throw local;
		throw StatusError{BLMX_ERCCL, "sharded build: the input of another rank was rejected"};
	}
	mark("exchange: all-reduce");
	uint64_t send_sk[16], send_words[16], recv_sk[16], recv_words[16];
	const void *d_meta = nullptr, *d_words = nullptr;
	shard_build_plan(ix, hist.data(), world, send_sk, send_words, &d_meta, &d_words);
	mark("exchange: pack outgoing");
	std::vector<uint64_t> mine(2 * (size_t)world), all(2 * (size_t)world * world);
	for (uint32_t r = 0; r < world; r++) {
		mine[r] = send_sk[r];
		mine[world + r] = send_words[r];
	}
	all_gather_host(c, mine.data(), mine.size(), all.data(), s);
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
	// From here on a local failure (out of memory in the owner's build) cannot be hidden from the others: the halo all-gather below
	// carries a failure word, so nobody waits for a rank that has given up.
	StatusError tail{BLMX_OK, ""};
	try {
		shard_build_finish(ix, d_in_meta.p, d_in_words.p, recv_sk, recv_words);
		ix.built = true;
	} catch (const StatusError& e2) {
		tail = e2;
	}
	const uint64_t rw = shard_halo_words(ix);
	std::vector<uint64_t> rec(rw + 1, 0), recs((size_t)world * (rw + 1), 0);
	if (tail.code == BLMX_OK)
		shard_halo_get(ix, rec.data());
	rec[rw] = tail.code != BLMX_OK;
	all_gather_host(c, rec.data(), rw + 1, recs.data(), s);
	bool any_failed = false;
	std::vector<uint64_t> packed((size_t)world * rw);
	for (uint32_t r = 0; r < world; r++) {
		any_failed |= recs[(size_t)r * (rw + 1) + rw] != 0;
		memcpy(&packed[(size_t)r * rw], &recs[(size_t)r * (rw + 1)], rw * 8);
	}
	if (tail.code != BLMX_OK)
		throw tail;
	if (any_failed) {
		ix.built = false;
		throw StatusError{BLMX_ERCCL, "sharded build: another rank could not finish its shard"};
	}
	shard_halo_set(ix, packed.data(), world);
	mark("exchange: halo");
}

// every rank reads the file (the reader's pair rule needs the lines from the start) and keeps the rank-th consecutive part of its
// sequences: rank r of w takes sequences [nseq * r / w, nseq * (r + 1) / w)
void build_sharded_from_fasta(Index& ix, Comm& c, const char* path)
{
	// Every rank streams the whole file through the device parser (the host only moves bytes: ~0.1 s for 3.5 GB) and keeps the rank-th
	// consecutive part of its sequences -- which sequences a rank contributes does not change the index (blmx_build.hip: slices).
	BLMX_HIP(hipSetDevice(c.device));
	DevBuf d_text, d_offs_all, d_mine, d_offs;
	uint64_t nseq = 0, total = 0, mine_total = 0, u0 = 0, u1 = 0;
	StatusError local = precondition(ix, c, false);
	if (local.code == BLMX_OK) {
		try {
			load_fasta_raw_to_device(ix, path, d_text, d_offs_all, nseq, total);
			u0 = nseq * c.rank / c.world;
			u1 = nseq * (c.rank + 1) / c.world;
			slice_packed_text(d_text, d_offs_all.as<uint64_t>(), u0, u1, d_mine, d_offs, mine_total, ix.stream);
			BLMX_HIP(hipStreamSynchronize(ix.stream));
			d_text.release();
			d_offs_all.release();
		} catch (const StatusError& e) {
			local = e;
		}
	}
	if (local.code != BLMX_OK) {
		build_sharded(ix, c, nullptr, nullptr, 0, 0, &local);  // throws `local` after the all-reduce
		throw local;
	}
	build_sharded(ix, c, nullptr, d_offs.as<uint64_t>(), u1 - u0, mine_total, nullptr, &d_mine);
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
	query_kmers_sharded(ix, c, ix.q_in.p, nullptr, nq, ix.q_out.as<int64_t>(), s);
	if (nq)
		BLMX_HIP(hipMemcpyAsync(ids_out, ix.q_out.p, nq * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
}

}  // namespace blmx
