// blmx_serve.hip -- a BOUNDED serving burst for the reference's own calling pattern (r04).
//
// The reference's applications call query_kmer_hash / query_sequence_hash once per k-mer or per read from OpenMP threads
// (blight.cpp:537-550, 575-591; Colored_De_Bruijn_graph_snippet.cpp:122-157).  One kernel launch per call (or per round of the
// combiner, blmx_front.hip) costs a launch plus a completion: 11-13 us per call, 3e5 scalar q/s from 8 threads -- below ONE core of
// the reference.  Here the first caller launches a small serving kernel that STAYS while requests keep arriving:
//
//   * a ring of request slots in pinned host memory (64 slots of 16 KiB); a caller claims a slot, writes its k-mers or its read, then
//     POSTS the request on the request board: one 16-byte line per slot {first k-mer | sequence number, kind} -- a scalar call needs
//     nothing else -- and spins on the slot's `done` word.  The board is a piece of the pinned ring; the host's stop word is one of its
//     lines and comes with the same load instruction (ONE read over PCIe per poll; r04 before: two, one behind the other).
//     BLIGHT_SERVE_BOARD=device puts the board into DEVICE memory that the host writes through the PCIe BAR (two posted 8-byte writes,
//     the k-mer first): the kernel's poll is then a local load.  tools/pingpong_probe says that saves 0.3-1 us of a bare round trip
//     (profiles/r04_pingpong_probe.jsonl); on the real path it does not pay -- three runs each on one box, 8 callers: 1.34-1.49e6
//     scalar q/s with the board in device memory, 1.50-1.63e6 in pinned memory (the host's fenced stores into the write-combining
//     mapping cost the callers more than the wave's PCIe read) -- so it is an option, not the default;
//   * 16 workgroups of two waves: wave 0 polls four slots (system-scope loads) and answers scalar calls by itself; for a read or a
//     k-mer list both waves work -- the query body of the batch kernels (query_device.h: same filter, same MPHF probes, same scan),
//     one thread per k-mer; a read is 2-bit packed into LDS first and every thread takes the k-mers at its positions --, store the
//     IDs into the slot, fence, and wave 0 sends the slot's answer line {ID, done, err} as one 16-byte store;
//   * every wave LEAVES after `idle` microseconds without a request (default 60) or after `max` milliseconds (default 20),
//     whichever comes first, and the last one out says so in pinned memory.  Nothing waits for the kernel without a deadline: a
//     caller whose request is not answered within 200 us looks whether the kernel has gone (it may have decided to leave just
//     before the request arrived), launches the next one, and waits again; after 50 ms without an answer the call fails with
//     BLMX_EHIP.  The kernel cannot outlive its bound, so the grid has always drained before the process ends; an index that is
//     rebuilt, loaded or destroyed tells the waves to leave and waits for them (bounded by the same deadline).
//
// Requests that do not fit a slot (reads above 1536 characters, batches above 128 k-mers) and indexes with windows above 2^8 take the
// launch-per-round path as before; serve_burst = 0 switches the burst off.
#include <chrono>
#include <cstring>

#include <hip/hip_runtime.h>
#include <cstddef>
#include <fcntl.h>
#include <immintrin.h>
#include <unistd.h>

#include "blmx_internal.h"
#include "query_device.h"

namespace blmx {

constexpr uint32_t kBurstSlots = 64;
constexpr uint32_t kBurstWaves = 16;                           // single-wave workgroups
constexpr uint32_t kBurstPerWave = kBurstSlots / kBurstWaves;  // slots a wave polls
constexpr size_t kBurstSlotBytes = 16384;
constexpr size_t kBurstPayloadAt = 64, kBurstPayloadBytes = 2048, kBurstResultsAt = kBurstPayloadAt + kBurstPayloadBytes;
constexpr uint32_t kBurstMaxChars = 1536, kBurstMaxKmers = 128;
constexpr uint32_t kBurstMaxIds = (uint32_t)((kBurstSlotBytes - kBurstResultsAt) / 8);  // 1784
static_assert(kBurstMaxChars <= kBurstPayloadBytes && kBurstMaxKmers * 16 <= kBurstPayloadBytes && kBurstMaxChars <= kBurstMaxIds, "slot layout");

// the first 64 bytes of a slot.  The host writes the payload and then posts the request on the board (below); the device writes the
// results, err, then `done`.  `what` of a board line: bit 31: a read (characters), else k-mers; bit 30: membership bytes wanted; low
// bits: characters / k-mers.
struct BurstHead {
	uint32_t posted_unused;  // (before the board the request was posted here)
	uint32_t what_unused;
	uint64_t kmer0_unused[2];
	uint32_t pad0[2];
	// the answer line: ONE aligned 16-byte store of the device = one write over PCIe, no fence in front of it for a scalar call
	int64_t id0;        // the ID (or 0 / 1 for a membership call) of a scalar call of a 64-bit k-mer; other calls: results area
	uint32_t done;      // = posted once the results are in place (device)
	int32_t err;        // 1: a character outside ACGTacgt (device)
	uint32_t pad[4];
};
static_assert(offsetof(BurstHead, id0) == 32, "the answer is one aligned 16-byte store");
static_assert(sizeof(BurstHead) == 64, "BurstHead");

struct BurstCtrl {  // one line of pinned memory next to the ring
	uint32_t unused;
	uint32_t exited_gen;    // device: generation of the last kernel whose waves have all left
	uint32_t pad[14];
};

// The request board: kBurstSlots + 1 lines of 16 bytes.  Line i < kBurstSlots = {u64 first k-mer (low word) | u32 posted, u32 what} of slot i;
// the last line = {0 | u32 stop, 0}: the host tells the waves to leave.  The host writes a line as TWO 8-byte stores with a store fence
// between them, the half that carries the sequence number last: 8-byte stores are single transactions on every mapping (a 16-byte store
// through a write-combining mapping need not be), and posted writes stay in order -- a poll can see a new k-mer behind an old
// sequence number (not posted yet), never the reverse.
constexpr size_t kBoardLine = 16, kBoardBytes = 4096;
constexpr uint32_t kBoardStopLine = kBurstSlots;
static_assert((kBurstSlots + 1) * kBoardLine <= kBoardBytes, "board");

template <int KW, int LOGP>
__device__ __forceinline__ int64_t serve_kmer(const DeviceIndexView& v, Kmer<KW> x)
{
	const Kmer<KW> xr = kmer_rc<KW>(x, v.k);
	int64_t id = -1;
	int32_t posj;
	if (!(xr < x)) {  // (a non-canonical k-mer matches nothing: blight.cpp:733-735)
		const uint32_t mini = kmer_minimizer<KW>(x, xr, v.k, v.m);
		query_one<KW, LOGP, false>(v, x, xr, mini, id, posj);
	}
	return id;
}

// The minimizer of ONE 64-bit k-mer by 16 lanes (a scalar call has nothing else for them to do): lane `sub` of the group hashes the
// windows sub and sub + 16 (k - m <= 31), then a 16-lane minimum -- ~60 instructions deep instead of the ~340 of kmer_minimizer on
// one lane, half a microsecond of a call's eight.  Same windows, same strand rule, same signed order (device_prims.h: kmer_minimizer).
__device__ __forceinline__ uint32_t kmer_minimizer_by16(Kmer<1> x, Kmer<1> xr, int k, int m, uint32_t sub)
{
	const uint32_t mask = (m >= 16) ? 0xffffffffu : ((1u << (2 * m)) - 1u);
	const int w = k - m;
	int32_t best = 0x7fffffff;
	for (int i = (int)sub; i <= w; i += 16) {
		const uint32_t f = kmer_shr32(x, 2 * i), r = kmer_shr32(xr, 2 * (w - i));
		const uint32_t sel = (__popc(f & mask) & 1) ? f : r;  // the strand with odd parity (ParityCanonical, kmer.h:475-487)
		best = min(best, revhash32(__builtin_amdgcn_ubfe(sel, 1u, (uint32_t)(2 * m - 1))));
	}
	for (int o = 8; o >= 1; o >>= 1)
		best = min(best, __shfl_xor(best, o, 16));
	return unrevhash32((uint32_t)best);
}

// what the waves of one serving kernel share (device memory): they leave TOGETHER -- wave 0 decides, from the time ANY wave last served
// a request -- so that "a kernel is there" means every slot is being watched (a wave that left alone would strand its four slots)
struct BurstShared {
	uint32_t left;              // waves that have left
	uint32_t leaving;           // generation whose waves have been told to leave
	unsigned long long last_work;  // s_memrealtime of the last served request
};

constexpr uint32_t kBurstThreads = 128;  // the polling wave + one helper wave that takes every other 64 k-mers of a read or a list

template <int KW, int LOGP>
__global__ __launch_bounds__(kBurstThreads) void k_serve_burst(DeviceIndexView v0, uint8_t* __restrict__ ring, BurstCtrl* __restrict__ ctrl,
                                                              const uint8_t* __restrict__ board, BurstShared* __restrict__ sh, uint32_t gen,
                                                              uint32_t idle_ticks, uint32_t max_ticks, uint32_t max_polls)
{
	__shared__ uint64_t s_seq[kBurstMaxChars / 32 + 8];  // a read, 2-bit packed, first nucleotide most significant (like the index's text)
	__shared__ uint32_t s_bad;
	// what wave 0 (the poller) tells the helper wave each round: bit 0 leave, bit 1 something is pending; the slots (lanes of wave 0) whose
	// requests need the whole workgroup, and their `what` words.  Every branch both waves take between two barriers depends on these only.
	__shared__ uint32_t s_ctl;
	__shared__ uint64_t s_rest;
	__shared__ uint32_t s_what[kBurstPerWave];
	const DeviceIndexView v = gated_view(v0);
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, wave = blockIdx.x;
	// lane l < kBurstPerWave of wave 0 watches slot wave + l * kBurstWaves
	const bool watcher = wv == 0 && lane < kBurstPerWave;
	const bool poller = watcher || (wv == 0 && lane == 63);  // lane 63 reads the stop line with the same instruction
	const uint32_t my_index = wave + (watcher ? lane : 0) * kBurstWaves;
	uint8_t* my_slot = ring + (size_t)my_index * kBurstSlotBytes;
	const uint8_t* my_line = board + (size_t)(lane == 63 ? kBoardStopLine : my_index) * kBoardLine;
	uint32_t served = 0;
	if (watcher)  // what was answered before this kernel started stays answered
		served = __hip_atomic_load(&reinterpret_cast<BurstHead*>(my_slot)->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
	uint64_t t_mine = t_start;
	if (wave == 0 && tid == 0)
		atomicMax(&sh->last_work, (unsigned long long)t_start);
	for (uint32_t poll = 0; poll < max_polls; poll++) {
		uint32_t posted = served, what = 0;
		uint64_t k0 = 0;
		bool pending = false;
		if (wv == 0) {
			// ---- poll: ONE 16-byte read per watched board line brings the first k-mer (low word), the sequence number and the kind; lane 63
			// reads the host's stop word in the same instruction (one memory round trip per poll, local when the board is in device memory)
			uint4 q = make_uint4(0u, 0u, 0u, 0u);
			if (poller)
				asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(q) : "v"(my_line) : "memory");
			if (watcher) {
				k0 = (uint64_t)q.x | ((uint64_t)q.y << 32);
				posted = q.z;
				what = q.w;
			}
			const uint32_t stop = lane == 63 ? q.z : 0u;
			pending = watcher && posted != served;
			const uint64_t pend = __ballot(pending);
			// Wave 0 of workgroup 0 looks at the clock for all of them (idle since ANY workgroup's last request, or the kernel's own time
			// limit); everybody looks at its verdict and at the host's stop word, and leaves once it has nothing pending.
			uint32_t go = stop;
			if (lane == 0) {
				if (wave == 0) {
					const uint64_t now = __builtin_amdgcn_s_memrealtime();
					const uint64_t lw = __hip_atomic_load(&sh->last_work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((pend == 0 && now > lw && now - lw > idle_ticks) || now - t_start > max_ticks)
						__hip_atomic_store(&sh->leaving, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				go |= __hip_atomic_load(&sh->leaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen ? 1u : 0u;
			}
			const bool leave0 = __ballot(go != 0) != 0;
			const bool whole = pending && !(KW == 1 && (what & 0xbfffffffu) == 1u);  // needs the workgroup (not a scalar call of a 64-bit k-mer)
			if (whole)
				s_what[lane] = what;
			const uint64_t rest0 = __ballot(whole);
			if (lane == 0) {
				s_ctl = (leave0 ? 1u : 0u) | (pend ? 2u : 0u);
				s_rest = rest0;
			}
		}
		__syncthreads();
		const uint32_t ctl = s_ctl;
		uint64_t rest = s_rest;
		__syncthreads();  // (the next round's s_ctl is not written before everybody has read this one)
		const bool leave = (ctl & 1u) != 0;
		if (!(ctl & 2u)) {
			if (leave)
				break;
			if (wv == 0 && __builtin_amdgcn_s_memrealtime() - t_mine > 2000u)  // (20 us after its own last request a wave polls back to back: a
				__builtin_amdgcn_s_sleep(16);                                    // blocking caller's next call comes within microseconds)
			continue;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // (the payload was written before `posted`)
		// ---- scalar calls of 64-bit k-mers: every watcher answers its own from the polled line
		int64_t my_id = 0;
		int32_t my_err = 0;
		if constexpr (KW == 1) {
			const bool scalar = pending && (what & 0xbfffffffu) == 1u;
			if (wv == 0 && __ballot(scalar)) {  // (wave-uniform: the 16 lanes [16 l, 16 l + 16) work on watcher l's k-mer)
				const int src = (int)(lane >> 4);
				Kmer<1> gx;
				gx.lo = (uint64_t)(uint32_t)__shfl((int)(uint32_t)k0, src) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(k0 >> 32), src) << 32);
				const Kmer<1> gxr = kmer_rc<1>(gx, v.k);
				const uint32_t gmini = kmer_minimizer_by16(gx, gxr, v.k, v.m, lane & 15u);
				const uint32_t mini = (uint32_t)__shfl((int)gmini, (int)((lane & 3u) << 4));  // watcher l takes its group's
				if (scalar) {
					Kmer<1> x;
					x.lo = k0;
					const Kmer<1> xr = kmer_rc<1>(x, v.k);
					int64_t id = -1;
					int32_t posj;
					if (!(xr < x))  // (a non-canonical k-mer matches nothing: blight.cpp:733-735)
						query_one<1, LOGP, false>(v, x, xr, mini, id, posj);
					my_id = (what & 0x40000000u) ? (int64_t)(id >= 0 ? 1 : 0) : id;
				}
			}
		}
		// ---- everything else, one request after the other, both waves on each: thread t takes k-mers t, t + 128, ...
		while (rest) {
			const int src = __ffsll((unsigned long long)rest) - 1;
			rest &= rest - 1;
			const uint32_t w = s_what[src];
			uint8_t* slot = ring + (size_t)(wave + (uint32_t)src * kBurstWaves) * kBurstSlotBytes;
			const uint32_t n = w & 0x3fffffffu;
			if (w & 0x80000000u) {  // a read of n characters: 2-bit pack it into LDS, 16 characters per thread
				if (tid == 0)
					s_bad = 0;
				for (uint32_t i = tid; i < sizeof(s_seq) / 8; i += kBurstThreads)
					s_seq[i] = 0;
				__syncthreads();
				uint32_t* s32 = reinterpret_cast<uint32_t*>(s_seq);
				const uint8_t* chars = slot + kBurstPayloadAt;
				uint32_t bad = 0;
				for (uint32_t c = tid; c * 16 < n; c += kBurstThreads) {
					typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
					const u32x4 q = *reinterpret_cast<const u32x4*>(chars + (size_t)c * 16);  // (payload bytes behind the read are the host's zeros)
					uint32_t packed = 0;
#pragma unroll
					for (int j = 0; j < 16; j++) {
						const uint32_t word = j < 4 ? q.x : (j < 8 ? q.y : (j < 12 ? q.z : q.w));
						const uint8_t ch = (uint8_t)(word >> (8 * (j & 3)));
						const uint32_t code = nuc_code(ch);
						if (c * 16 + (uint32_t)j < n)
							bad |= code > 3 ? 1u : 0u;
						packed |= (code & 3u) << (30 - 2 * j);
					}
					s32[c ^ 1u] = packed;  // (little-endian u64 words: the first 16 nucleotides of a word are its high half)
				}
				if (bad)
					s_bad = 1;
				__syncthreads();
				const uint32_t nk = n >= (uint32_t)v.k ? n - (uint32_t)v.k + 1 : 0;
				int64_t* ids = reinterpret_cast<int64_t*>(slot + kBurstResultsAt);
				const uint32_t was_bad = s_bad;
				if (!was_bad)
					for (uint32_t p = tid; p < nk; p += kBurstThreads) {
						const Kmer<KW> y = packed_kmer<KW>(s_seq, p, v.k);
						const Kmer<KW> yr = kmer_rc<KW>(y, v.k);
						ids[p] = serve_kmer<KW, LOGP>(v, yr < y ? yr : y);  // min_k, blight.cpp:86-91
					}
				if (wv == 0 && (int)lane == src)
					my_err = (int32_t)was_bad;
			} else {  // n canonical k-mers
				const bool want_bool = (w & 0x40000000u) != 0;
				for (uint32_t i = tid; i < n; i += kBurstThreads) {
					const Kmer<KW> x = load_kmer<KW>(reinterpret_cast<const uint64_t*>(slot + kBurstPayloadAt), i);
					const int64_t id = serve_kmer<KW, LOGP>(v, x);
					if (want_bool)
						reinterpret_cast<uint8_t*>(slot + kBurstResultsAt)[i] = id >= 0 ? 1 : 0;
					else
						reinterpret_cast<int64_t*>(slot + kBurstResultsAt)[i] = id;
				}
			}
			__threadfence_system();  // this thread's results are on their way before ...
			__syncthreads();         // ... the poller learns that the request is done (and before s_seq is packed again)
		}
		// ---- the answer line {ID, done, err}: one 16-byte store per slot.  The results areas were fenced by the threads that wrote them
		// (above); a wave that had only scalar calls has nothing to fence.
		if (pending) {
			typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
			const u32x4 ans = {(uint32_t)my_id, (uint32_t)((uint64_t)my_id >> 32), posted, (uint32_t)my_err};
			asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(my_slot + offsetof(BurstHead, id0)), "v"(ans) : "memory");
			served = posted;
		}
		if (wv == 0) {
			t_mine = __builtin_amdgcn_s_memrealtime();
			if (lane == 0)
				atomicMax(&sh->last_work, (unsigned long long)t_mine);
		}
		if (leave)  // (told to leave while requests were pending: they are answered, the next ones are the next kernel's)
			break;
	}
	// ---- leave: the last workgroup out publishes the generation
	if (tid == 0) {
		__threadfence_system();
		if (atomicAdd(&sh->left, 1u) + 1u == gridDim.x) {
			sh->left = 0;
			__threadfence();
			__hip_atomic_store(&ctrl->exited_gen, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
struct BurstState {
	Pinned ring;   // kBurstSlots slots + one BurstCtrl line
	DevBuf left;   // BurstShared of the current kernel
	hipStream_t s = nullptr;
	std::mutex launch_mu;
	std::atomic<uint32_t> launched_gen{0};
	// what the callers of one slot share, on a line of its own: eight OpenMP callers at a million calls a second must not pass one cache
	// line around (r04: the slots' busy words sat side by side, and every call also took the index's mutex twice and bumped two shared
	// counters -- the callers spent more time on each other than on the GPU)
	struct alignas(64) SlotState {
		std::atomic<uint32_t> busy{0};
		uint32_t seq = 0;
		std::atomic<uint64_t> calls{0};
		std::atomic<uint64_t> kmers{0};  // k-mers answered through this slot: part of the index's number_query (burst_number_query)
	};
	SlotState st[kBurstSlots];
	std::atomic<uint32_t> next_hint{0};
	std::atomic<uint64_t> launches{0}, relaunch_after_timeout{0}, relaunch_after_exit{0};
	uint64_t calls_total() const
	{
		uint64_t n = 0;
		for (const auto& x : st)
			n += x.calls.load(std::memory_order_relaxed);
		return n;
	}
	uint32_t idle_us = 60, max_ms = 20;
	uint8_t* board = nullptr;     // the request board: device memory behind the BAR, or (fallback) a piece of the pinned ring
	void* board_dev = nullptr;    // the device allocation, when the board is there
	BurstState() = default;
	~BurstState()
	{
		if (board_dev)
			(void)hipFree(board_dev);
	}
	// one board line: the half with the sequence number (or the stop word) last, a store fence in front of either half -- the payload in
	// the pinned slot is globally visible before the k-mer, the k-mer before the sequence number (see kBoardLine)
	void board_write(uint32_t line, uint64_t w0, uint64_t w1)
	{
		volatile uint64_t* p = reinterpret_cast<volatile uint64_t*>(board + (size_t)line * kBoardLine);
		_mm_sfence();
		p[0] = w0;
		_mm_sfence();
		p[1] = w1;
		_mm_sfence();
	}
	uint8_t* slot(uint32_t i) { return ring.as<uint8_t>() + (size_t)i * kBurstSlotBytes; }
	BurstCtrl* ctrl() { return reinterpret_cast<BurstCtrl*>(ring.as<uint8_t>() + (size_t)kBurstSlots * kBurstSlotBytes); }
	bool alive()
	{
		const uint32_t ex = __atomic_load_n(&ctrl()->exited_gen, __ATOMIC_ACQUIRE);
		return launched_gen.load(std::memory_order_acquire) != ex;
	}
};

namespace {

BurstState* burst_state(Index& ix)
{
	if (BurstState* have = __atomic_load_n(&ix.burst, __ATOMIC_ACQUIRE))
		return have;
	std::lock_guard<std::mutex> l(ix.hf_mu);
	if (!ix.burst) {
		std::unique_ptr<BurstState> b(new BurstState());
		b->ring.ensure((size_t)kBurstSlots * kBurstSlotBytes + 4096);
		memset(b->ring.p, 0, (size_t)kBurstSlots * kBurstSlotBytes + 4096);
		BLMX_HIP(hipStreamCreateWithFlags(&b->s, hipStreamNonBlocking));
		BLMX_HIP(b->left.alloc(64, true, b->s));
		BLMX_HIP(hipStreamSynchronize(b->s));
		// The request board: a piece of the pinned ring, or (BLIGHT_SERVE_BOARD=device) device memory, if THIS process can store into it
		// (large BAR).  Asked without a signal: the kernel is made to store into the page -- read(2) from /dev/zero -- and an address the
		// process cannot write is EFAULT there; the board then stays in the pinned ring.
		b->board = b->ring.as<uint8_t>() + (size_t)kBurstSlots * kBurstSlotBytes + 2048;
		const char* where = getenv("BLIGHT_SERVE_BOARD");
		if (where && !strcmp(where, "device")) {
			void* d = nullptr;
			if (hipExtMallocWithFlags(&d, kBoardBytes, hipDeviceMallocUncached) == hipSuccess && d) {
				bool ok = hipMemsetAsync(d, 0, kBoardBytes, b->s) == hipSuccess && hipStreamSynchronize(b->s) == hipSuccess;
				if (ok) {
					const int fd = open("/dev/zero", O_RDONLY | O_CLOEXEC);
					ok = fd >= 0 && read(fd, d, kBoardBytes) == (ssize_t)kBoardBytes;
					if (fd >= 0)
						close(fd);
				}
				if (ok) {
					b->board_dev = d;
					b->board = reinterpret_cast<uint8_t*>(d);
				} else {
					(void)hipFree(d);
				}
			}
			(void)hipGetLastError();
		}
		if (const char* e = getenv("BLIGHT_SERVE_IDLE_US"))
			b->idle_us = (uint32_t)std::max(1, atoi(e));
		if (const char* e = getenv("BLIGHT_SERVE_MAX_MS"))
			b->max_ms = (uint32_t)std::max(1, atoi(e));
		__atomic_store_n(&ix.burst, b.release(), __ATOMIC_RELEASE);
	}
	return ix.burst;
}

template <int KW> void launch_burst_kw(const Index& ix, BurstState& b, uint32_t gen)
{
	const DeviceIndexView v = ix.view();
	// s_memrealtime counts at 100 MHz; the poll cap bounds the kernel even if that clock were to stand still (a poll takes >= 1 us)
	const uint32_t idle_ticks = b.idle_us * 100u, max_ticks = b.max_ms * 100000u, max_polls = b.max_ms * 4000u + 1000u;
#define BLMX_SV(LP)                                                                                                                                    \
	case LP:                                                                                                                                           \
		hipLaunchKernelGGL((k_serve_burst<KW, LP>), dim3(kBurstWaves), dim3(kBurstThreads), 0, b.s, v, b.ring.as<uint8_t>(), b.ctrl(), b.board, b.left.as<BurstShared>(), gen, \
		                   idle_ticks, max_ticks, max_polls);                                                                                          \
		break;
	switch (v.b) {
		BLMX_SV(0)
		BLMX_SV(1)
		BLMX_SV(2)
		BLMX_SV(3)
		BLMX_SV(4)
		BLMX_SV(5)
		BLMX_SV(6)
		BLMX_SV(7)
		BLMX_SV(8)
	default:
		throw StatusError{BLMX_EINVAL, "internal: the serving burst needs bits_to_save <= 8"};
	}
#undef BLMX_SV
	BLMX_HIP(hipGetLastError());
}

// make sure a serving kernel is (still) there; returns after the launch has been enqueued
void ensure_burst(Index& ix, BurstState& b)
{
	if (b.alive())
		return;
	std::lock_guard<std::mutex> l(b.launch_mu);
	if (b.alive())
		return;
	BLMX_HIP(hipSetDevice(ix.prm.device));
	b.board_write(kBoardStopLine, 0, 0);
	uint32_t gen = b.launched_gen.load(std::memory_order_relaxed) + 1;
	if (gen == 0)
		gen = 1;
	if (ix.KW == 1)
		launch_burst_kw<1>(ix, b, gen);
	else
		launch_burst_kw<2>(ix, b, gen);
	b.launched_gen.store(gen, std::memory_order_release);
	b.launches.fetch_add(1, std::memory_order_relaxed);
}

// A thread goes back to the slot it had last time (nobody else touches that line then); the first call of a thread, or one that finds
// its slot taken, draws the next one -- consecutive slots belong to different serving waves.
int claim_slot(BurstState& b)
{
	thread_local uint32_t mine = 0xffffffffu;
	uint32_t z = 0;
	if (mine < kBurstSlots && b.st[mine].busy.compare_exchange_strong(z, 1u, std::memory_order_acquire))
		return (int)mine;
	const uint32_t h = b.next_hint.fetch_add(1, std::memory_order_relaxed);
	for (uint32_t i = 0; i < kBurstSlots; i++) {
		const uint32_t sidx = (h + i) % kBurstSlots;
		z = 0;
		if (b.st[sidx].busy.compare_exchange_strong(z, 1u, std::memory_order_acquire)) {
			mine = sidx;
			return (int)sidx;
		}
	}
	return -1;
}

// post the request that sits in slot `si` (payload and head fields written) and wait for its answer
void post_and_wait(Index& ix, BurstState& b, int si, uint32_t what, const uint64_t kmer0[2])
{
	BurstHead* h = reinterpret_cast<BurstHead*>(b.slot((uint32_t)si));
	uint32_t seq = ++b.st[si].seq;
	if (seq == 0)
		seq = ++b.st[si].seq;
	(void)kmer0[1];  // (the high word of a 128-bit k-mer travels in the payload)
	b.board_write((uint32_t)si, kmer0[0], (uint64_t)seq | ((uint64_t)what << 32));
	ensure_burst(ix, b);
	const auto t0 = std::chrono::steady_clock::now();
	auto deadline = t0 + std::chrono::microseconds(200);
	for (uint32_t spin = 0;; spin++) {
		if (__atomic_load_n(&h->done, __ATOMIC_ACQUIRE) == seq)
			return;
		_mm_pause();
		if ((spin & 31u) == 31u && !b.alive()) {
			// the kernel had decided to leave when the request arrived (its last wave out has said so in pinned memory): the next
			// one takes the request.  Looked at every ~microsecond, not only at the deadline below: under steady load a kernel lives
			// its 20 ms, and every caller whose request it no longer saw used to wait out 200 us (r04: 8 % of the callers' time)
			b.relaunch_after_exit.fetch_add(1, std::memory_order_relaxed);
			ensure_burst(ix, b);
		}
		if ((spin & 255u) == 255u) {
			const auto now = std::chrono::steady_clock::now();
			if (now > deadline) {
				if (now - t0 > std::chrono::milliseconds(50))
					throw StatusError{BLMX_EHIP, "the serving kernel did not answer a call within 50 ms"};
				if (!b.alive()) {
					b.relaunch_after_timeout.fetch_add(1, std::memory_order_relaxed);
					ensure_burst(ix, b);
				}
				deadline = now + std::chrono::microseconds(200);
			}
		}
	}
}

struct SlotLease {
	BurstState& b;
	int si;
	~SlotLease() { b.st[si].busy.store(0, std::memory_order_release); }
};

}  // namespace

bool burst_usable(const Index& ix) { return ix.tune_serve_burst != 0 && ix.built && ix.prm.bits_saved <= 8; }

// n canonical k-mers (KW words each) -> IDs (int64) or membership bytes; false: not taken (the caller uses the launch-per-round path)
bool burst_query_kmers(Index& ix, const void* kmers, uint64_t n, void* out, bool want_bool)
{
	if (!burst_usable(ix) || n == 0 || n > kBurstMaxKmers)
		return false;
	BurstState& b = *burst_state(ix);
	const int si = claim_slot(b);
	if (si < 0)
		return false;
	SlotLease lease{b, si};
	const size_t kb = (size_t)ix.KW * 8;
	uint8_t* slot = b.slot((uint32_t)si);
	uint64_t k0[2] = {0, 0};
	memcpy(k0, kmers, kb);
	if (!(n == 1 && ix.KW == 1))
		memcpy(slot + kBurstPayloadAt, kmers, n * kb);
	post_and_wait(ix, b, si, (uint32_t)n | (want_bool ? 0x40000000u : 0u), k0);
	if (n == 1 && ix.KW == 1) {  // the answer came with `done`
		const int64_t id = reinterpret_cast<BurstHead*>(slot)->id0;
		if (want_bool)
			*reinterpret_cast<uint8_t*>(out) = (uint8_t)id;
		else
			*reinterpret_cast<int64_t*>(out) = id;
	} else {
		memcpy(out, slot + kBurstResultsAt, want_bool ? n : n * 8);
	}
	b.st[si].kmers.fetch_add(n, std::memory_order_relaxed);
	b.st[si].calls.fetch_add(1, std::memory_order_relaxed);
	return true;
}

// one sequence of nbytes characters (nk = nbytes - k + 1 > 0 k-mers) -> IDs; throws BLMX_EDNA like the other paths
bool burst_query_sequence(Index& ix, const char* chars, uint64_t nbytes, uint64_t nk, int64_t* out)
{
	if (!burst_usable(ix) || nbytes > kBurstMaxChars || nk == 0 || nk > kBurstMaxIds)
		return false;
	BurstState& b = *burst_state(ix);
	const int si = claim_slot(b);
	if (si < 0)
		return false;
	SlotLease lease{b, si};
	uint8_t* slot = b.slot((uint32_t)si);
	memcpy(slot + kBurstPayloadAt, chars, nbytes);
	memset(slot + kBurstPayloadAt + nbytes, 0, ((nbytes + 15) & ~(uint64_t)15) - nbytes);
	const uint64_t k0[2] = {0, 0};
	post_and_wait(ix, b, si, (uint32_t)nbytes | 0x80000000u, k0);
	if (reinterpret_cast<BurstHead*>(slot)->err)
		throw StatusError{BLMX_EDNA, "Invalid char in DNA"};
	memcpy(out, slot + kBurstResultsAt, nk * 8);
	b.st[si].kmers.fetch_add(nk, std::memory_order_relaxed);
	b.st[si].calls.fetch_add(1, std::memory_order_relaxed);
	return true;
}

// tell the waves to leave and wait until they have (bounded: the kernel leaves by itself after max_ms)
void burst_stop(Index& ix)
{
	BurstState* b = ix.burst;
	if (!b)
		return;
	std::lock_guard<std::mutex> l(b->launch_mu);
	if (!b->alive())
		return;
	b->board_write(kBoardStopLine, 0, 1);
	const auto t0 = std::chrono::steady_clock::now();
	while (b->alive() && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200))
		_mm_pause();
	(void)hipStreamSynchronize(b->s);  // (returns at once when the waves have left; bounded by the kernel's own limit otherwise)
}

void burst_destroy(Index& ix)
{
	if (!ix.burst)
		return;
	burst_stop(ix);
	if (getenv("BLIGHT_TRACE") && atoi(getenv("BLIGHT_TRACE")) != 0 && ix.burst->calls_total())
		fprintf(stderr, "[blmx serve] %llu calls answered by %llu serving kernels (%llu launched by a caller that saw the kernel gone, %llu after a caller's 200-us deadline)\n",
		        (unsigned long long)ix.burst->calls_total(), (unsigned long long)ix.burst->launches.load(),
		        (unsigned long long)ix.burst->relaunch_after_exit.load(), (unsigned long long)ix.burst->relaunch_after_timeout.load());
	if (ix.burst->s) {
		(void)hipStreamSynchronize(ix.burst->s);
		(void)hipStreamDestroy(ix.burst->s);
	}
	delete ix.burst;
	ix.burst = nullptr;
}

// k-mers answered by the burst: they are counted per slot, not on the index's shared counter (blmx_stats adds them)
uint64_t burst_number_query(const Index& ix)
{
	const BurstState* b = __atomic_load_n(&ix.burst, __ATOMIC_ACQUIRE);
	uint64_t n = 0;
	if (b)
		for (const auto& x : b->st)
			n += x.kmers.load(std::memory_order_relaxed);
	return n;
}

// what the tests look at: {kernels launched, calls answered, relaunches by a waiting caller (kernel gone / deadline), 1 if a kernel is there now}
void burst_stats(Index& ix, uint64_t out[4])
{
	out[0] = out[1] = out[2] = out[3] = 0;
	if (!ix.burst)
		return;
	out[0] = ix.burst->launches.load();
	out[1] = ix.burst->calls_total();
	out[2] = ix.burst->relaunch_after_timeout.load() + ix.burst->relaunch_after_exit.load();
	out[3] = ix.burst->alive() ? 1 : 0;
}

}  // namespace blmx
