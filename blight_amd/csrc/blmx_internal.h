// blmx_internal.h -- host-side structures shared by the build, query and API translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/blight_mi355x.h"
#include "device_prims.h"

namespace blmx {

// ---- HBM layout of the finished index ------------------------------------------------------------------------------
// One bucket_minimizer (reference blight.h:29-34) per minimizer value: 16 B so a lane fetches it with one dwordx4.
struct alignas(16) BucketDesc {
	uint64_t start;  // first nt of the bucket in the packed sequence
	uint32_t nt;     // nuc_minimizer
	uint32_t pad;
};

// One info_mphf (reference blight.h:39-44) + the BBHash level table of the group's MPHF (bbhash.h:780).
// 192 B = three 64-B sectors; everything a positive query needs (prefix, posstart, line_base, nbits, sizes 0..2) sits
// in the first sector.
struct alignas(64) GroupDesc {
	uint64_t prefix;     // ID base = number of keys in lower groups (mphf_size after blight.cpp:299-301)
	uint64_t posstart;   // first bit of the group's slice of `positions` (info_mphf::start)
	uint64_t line_base;  // first 64-B line of the group's MPHF in the line array
	uint32_t nbits;      // bit_to_encode
	uint32_t fb_count;   // entries of the group's fallback table
	uint64_t fb_start;   // first entry in the fallback arrays
	uint64_t size[kLevels];  // level sizes in bits (_hash_domains)
	uint64_t nkeys;
	uint64_t wbase;      // build time: first logical u64 word of the group's level bit array
	uint64_t nwords;     // logical words (sum of sizes / 64)
};
static_assert(sizeof(GroupDesc) == 192 + 0, "GroupDesc must be 192 bytes");

// What a query needs from its group up to MPHF level 2, in 32 B (two 16-B loads instead of four): built from the group
// table once the index is finished (make_query_heads).  Only used when every group fits the narrow fields; otherwise
// DeviceIndexView::qheads is null and the kernels read the GroupDesc head.
struct alignas(32) QueryHead {
	uint64_t prefix;
	uint64_t posstart_nbits;  // posstart (56 bits) | nbits << 56
	uint32_t line_base;
	uint32_t size0, size1, size2;
};
static_assert(sizeof(QueryHead) == 32, "QueryHead must be 32 bytes");

struct DeviceIndexView {
	int k, m, n, b, gshift;
	int hint;  // extra low bits of every position field (physical layout only, see read_position_field); 0 = the reference's fields
	uint32_t nbuckets, ngroups;
	const BucketDesc* buckets;
	const GroupDesc* groups;
	const QueryHead* qheads;    // compact per-group head for the query kernel, or nullptr
	const uint64_t* seq;        // 2-bit packed, MSB-first
	const uint64_t* positions;  // LSB-first bit fields
	const uint64_t* lines;      // MPHF lines: [rank | 7 data words]
	const uint64_t* fb_keys;    // fallback keys (KW words each), sorted by (group, key)
	const uint64_t* fb_ranks;
	// the fallback entries once more as per-group open-addressing tables (index_device.h: fallback_find); fb_dir[g] = first slot of the
	// group's table (low 56 bits) | log2 of its slots (high 8 bits), 0 = the group has no fallback keys
	const uint64_t* fb_table;
	const uint64_t* fb_dir;
	uint64_t total_nt;
	// absent-key filter (blocked Bloom filter over the keys of this index, one 64-bit block per probe; nullptr = off)
	const uint64_t* filter;
	uint64_t id_base;     // ID of the first key this index holds (0 unless sharded): group regions of the filter follow local IDs
	uint32_t filter_bpk;  // filter bits per key
	bool fat;             // level-0 lines are 64 B and carry the filter (index_device.h); `filter` is then only a flag (non-null = consult it)
	bool rec;             // (implies fat) the 64-B level-0 units are RECORDS of 32 level bits that also carry their keys' position fields
	// one word: 1 = consult the filter, 0 = skip it (the last large batch had almost no absent k-mers: the filter would only cost its
	// load; k_gate_sample decides after every large batch).  nullptr = always use.
	uint32_t* filter_gate;
};

// ---- device memory: blocks are recycled, within a build and across builds -----------------------------------------------------------
// On this driver hipMalloc of VRAM the process has used before costs 25-100 ms per GB (pages are cleared when they are handed out
// again), and once a process has allocated as much as the card holds a call has to wait for VRAM the driver has not finished
// reclaiming: 2.9-3.4 s, measured on every fourth or fifth C4-size build in a row (profiles/r03_exp_slow_hipmalloc.txt).
//
// ScratchCache (process-wide, per device): device blocks of 1 MiB and more that a finished build, a destroyed index or a staging buffer
// gave up are parked here -- up to a cap per device, default 1/16 of its memory, BLIGHT_SCRATCH_CACHE_GB -- and serve the next request
// they fit: a build's temporaries with a loose fit, index arrays and staging buffers (DevBuf outside a pool) with a tight one, so
// that an index reports and writes the sizes it asked for.  blmx_release_scratch, and any allocation that fails INSIDE this library,
// empty it; an allocator that shares the card (torch's) does not know about it: such a process calls blmx_release_scratch when its
// own allocation fails, or keeps the cap small.
//
// BuildPool (below): the cache of ONE build.  A human-scale build allocates ~60 temporaries whose lifetimes barely overlap; blocks
// released by one are handed to the next request they fit (all work of a build is on one stream, so reuse is ordered), and what is
// left when the build ends goes to the ScratchCache.
struct ScratchCache {
	struct Block {
		void* p;
		size_t bytes;
		int device;
	};
	static constexpr int kMaxDevices = 64;
	std::mutex mu;
	std::vector<Block> blocks;         // nothing on the device touches them any more
	size_t held[kMaxDevices] = {};     // bytes parked per device
	long long cap[kMaxDevices];        // per device; -1 = not decided yet
	ScratchCache()
	{
		for (long long& c : cap)
			c = -1;
	}
	static ScratchCache& get()
	{
		static ScratchCache* c = new ScratchCache();  // never destroyed (the HIP runtime may be gone when static destructors run)
		return *c;
	}
	// cap of one device: BLIGHT_SCRATCH_CACHE_GB, else 1/16 of the device's memory (18 GB of an MI355X's 288: what a process that
	// shares the card with another allocator -- torch's caching allocator, a host application -- can be expected to leave parked;
	// a process that builds human-scale indexes in a row raises it, bench.py does and says so)
	long long cap_of(int dev)
	{
		if (cap[dev] < 0) {
			const char* e = getenv("BLIGHT_SCRATCH_CACHE_GB");
			size_t fr = 0, tot = 0;
			if (e)
				cap[dev] = (long long)(atof(e) * 1e9);
			else
				cap[dev] = hipMemGetInfo(&fr, &tot) == hipSuccess ? (long long)(tot / 16) : 0;
		}
		return cap[dev];
	}
	void give(void* p, size_t bytes)
	{
		int dev = 0;
		if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices) {
			std::lock_guard<std::mutex> lock(mu);
			if ((long long)(held[dev] + bytes) <= cap_of(dev)) {
				blocks.push_back({p, bytes, dev});
				held[dev] += bytes;
				return;
			}
		}
		(void)hipFree(p);
	}
	// tight: index arrays and staging buffers (what they hold is what the index reports) take a block at most 1/8 larger; build
	// temporaries one at most twice as large (or 1 MiB larger: a byte-sized temporary does not pin a 64-MiB block)
	bool take(size_t n, void** p, size_t* bytes, bool tight = false)
	{
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
			return false;
		const size_t most = tight ? n + n / 8 + (1u << 20) : std::max(2 * n, n + (size_t)(1u << 20));
		std::lock_guard<std::mutex> lock(mu);
		size_t best = blocks.size();
		for (size_t i = 0; i < blocks.size(); i++)
			if (blocks[i].device == dev && blocks[i].bytes >= n && blocks[i].bytes <= most && (best == blocks.size() || blocks[i].bytes < blocks[best].bytes))
				best = i;
		if (best == blocks.size())
			return false;
		*p = blocks[best].p;
		*bytes = blocks[best].bytes;
		held[dev] -= std::min(held[dev], blocks[best].bytes);
		blocks.erase(blocks.begin() + (long)best);
		return true;
	}
	// a buffer that did not come from a build's pool (index array, staging, query scratch): kernels that use it may still be queued --
	// hipFree would wait for them, and so does this: a buffer that grows in the middle of a multi-stream pipeline is only handed to
	// somebody else once everything queued before has finished
	void give_synced(void* p, size_t bytes)
	{
		if (bytes < kMinBytes) {
			(void)hipFree(p);
			return;
		}
		(void)hipDeviceSynchronize();
		give(p, bytes);
	}
	static constexpr size_t kMinBytes = 1u << 20;
	size_t release_all()  // everything back to the driver; returns the bytes released
	{
		std::vector<Block> mine;
		{
			std::lock_guard<std::mutex> lock(mu);
			mine.swap(blocks);
			for (size_t& h : held)
				h = 0;
		}
		size_t n = 0;
		int cur = 0;
		const bool have_cur = hipGetDevice(&cur) == hipSuccess;
		for (const Block& b : mine) {
			if (have_cur && b.device != cur)
				(void)hipSetDevice(b.device);
			(void)hipFree(b.p);
			n += b.bytes;
		}
		if (have_cur)
			(void)hipSetDevice(cur);
		return n;
	}
};

struct BuildPool {
	struct Block {
		void* p;
		size_t bytes;
	};
	std::vector<Block> free_blocks;
	size_t held = 0, peak_held = 0;  // bytes this pool holds (live temporaries + parked blocks), and their high-water mark
	~BuildPool() { drain(); }
	void drain(bool to_driver = false)  // parked blocks -> the process-wide cache (or straight back to the driver)
	{
		if (!to_driver && !free_blocks.empty())
			(void)hipDeviceSynchronize();  // (what hipFree did implicitly: the next owner may run on another stream)
		for (const Block& b : free_blocks) {
			if (to_driver)
				(void)hipFree(b.p);
			else
				ScratchCache::get().give(b.p, b.bytes);
			held -= std::min(held, b.bytes);
		}
		free_blocks.clear();
	}
	hipError_t get(size_t n, void** p, size_t* bytes)
	{
		size_t best = free_blocks.size();
		for (size_t i = 0; i < free_blocks.size(); i++)
			if (free_blocks[i].bytes >= n && free_blocks[i].bytes <= 2 * n + (64u << 20) &&
			    (best == free_blocks.size() || free_blocks[i].bytes < free_blocks[best].bytes))
				best = i;
		if (best != free_blocks.size()) {
			*p = free_blocks[best].p;
			*bytes = free_blocks[best].bytes;
			free_blocks.erase(free_blocks.begin() + (long)best);
			return hipSuccess;
		}
		if (ScratchCache::get().take(n, p, bytes)) {
			held += *bytes;
			peak_held = std::max(peak_held, held);
			return hipSuccess;
		}
		const auto t0 = std::chrono::steady_clock::now();
		hipError_t e = hipMalloc(p, n);
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		if (ms > 20.0 && getenv("BLIGHT_BUILD_TRACE"))
			fprintf(stderr, "[blmx build] hipMalloc of %.2f GB took %.1f ms\n", n / 1e9, ms);
		if (e != hipSuccess) {  // out of memory with blocks parked here or in the process-wide cache: give them back and retry
			(void)hipGetLastError();
			drain(true);
			ScratchCache::get().release_all();
			e = hipMalloc(p, n);
		}
		*bytes = n;
		if (e == hipSuccess) {
			held += n;
			peak_held = std::max(peak_held, held);
		}
		return e;
	}
	void put(void* p, size_t bytes) { free_blocks.push_back(Block{p, bytes}); }
	static BuildPool*& current()
	{
		static thread_local BuildPool* pool = nullptr;
		return pool;
	}
};

struct DevBuf {
	void* p = nullptr;
	size_t bytes = 0;
	bool persistent = false;     // index array: exact hipMalloc, never pooled
	BuildPool* owner = nullptr;  // pool the current block came from (build temporaries)
	size_t cap = 0;              // outside a pool: size of the block behind p when it came from the process-wide cache (0: bytes)
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	~DevBuf() { release(); }
	void release()
	{
		if (p) {
			if (owner)
				owner->put(p, bytes);
			else
				ScratchCache::get().give_synced(p, cap ? cap : bytes);
		}
		p = nullptr;
		bytes = 0;
		cap = 0;
		owner = nullptr;
	}
	void swap(DevBuf& o)  // (persistent stays with the object: it says what the object is, not what it holds)
	{
		std::swap(p, o.p);
		std::swap(bytes, o.bytes);
		std::swap(cap, o.cap);
		std::swap(owner, o.owner);
	}
	hipError_t alloc(size_t n, bool zero, hipStream_t s)
	{
		release();
		if (n == 0)
			n = 16;
		BuildPool* pool = persistent ? nullptr : BuildPool::current();
		hipError_t e;
		if (pool) {
			e = pool->get(n, &p, &bytes);
			owner = e == hipSuccess ? pool : nullptr;
		} else {
			if (n >= ScratchCache::kMinBytes && ScratchCache::get().take(n, &p, &cap, true)) {
				e = hipSuccess;
				bytes = n;  // (what was asked for: the index reports and writes exactly that)
			} else {
				const auto t0 = std::chrono::steady_clock::now();
				e = hipMalloc(&p, n);
				const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
				if (ms > 20.0 && getenv("BLIGHT_BUILD_TRACE"))
					fprintf(stderr, "[blmx] hipMalloc of %.2f GB (index array / staging) took %.1f ms\n", n / 1e9, ms);
				if (e == hipErrorOutOfMemory && ScratchCache::get().release_all()) {
					(void)hipGetLastError();
					e = hipMalloc(&p, n);
				}
				bytes = n;
			}
		}
		if (e != hipSuccess) {
			p = nullptr;
			bytes = 0;
			cap = 0;
			owner = nullptr;
			return e;
		}
		if (zero) {
			e = hipMemsetAsync(p, 0, n, s);
		} else {
			// BLIGHT_POISON=1 (test hook): a buffer that is not asked to be zero starts as 0xA5 bytes -- recycled blocks hold arbitrary
			// data, fresh ones happen to be zero: nothing may depend on either
			static const bool poison = getenv("BLIGHT_POISON") && atoi(getenv("BLIGHT_POISON")) != 0;
			if (poison)
				e = hipMemsetAsync(p, 0xA5, cap ? cap : bytes, s);
		}
		return e;
	}
	template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct StatusError {
	int code;
	std::string msg;
};

// ---- pinned host staging buffer (grow-only; `keep` bytes survive a grow) -----------------------------------------------------
size_t release_pinned_cache();  // blmx_stream.hip: parked pinned blocks back to the driver; returns their bytes
struct Pinned {
	void* p = nullptr;
	size_t bytes = 0;
	Pinned() = default;
	Pinned(const Pinned&) = delete;
	Pinned& operator=(const Pinned&) = delete;
	~Pinned() { release(); }
	void release();                          // blmx_stream.hip: large blocks go to a process-wide cache (pinning costs ~0.7 ms per MiB)
	void ensure(size_t n, size_t keep = 0);  // blmx_stream.hip
	template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// device scratch of the sequence / read front end (blmx_reads.hip): bitmaps of one window and their popcount prefixes
struct FrontScratch {
	DevBuf marks;  // sequence-start bitmap (concatenated sequences) or newline bitmap (raw FASTA bytes)
	DevBuf mpre;   // raw: newlines before each 64-position word
	DevBuf nlpos;  // raw: position of every newline (+ the end of the unterminated last line)
	DevBuf ok;     // raw: per line, 1 = its k-mers are queried
	DevBuf valid;  // bit q = a k-mer starts at window position q
	DevBuf vpre;   // k-mers that start before each 64-position word (the last entry = all of them)
	DevBuf tmp;    // rocPRIM scan scratch
	DevBuf lpre;   // raw, construction: per line, sequences and characters of the window's sequence lines before it
	DevBuf stage;  // raw, construction: the window's sequence characters, compacted, on their way into the 2-bit text
};
struct BurstState;  // blmx_serve.hip: the bounded serving burst of small calls (request ring in pinned memory)
struct HostFront;  // blmx_stream.hip: pinned staging, window slots, small-call slots and worker threads of one index
void destroy_host_front(HostFront*);

#define BLMX_HIP(expr)                                                                                         \
	do {                                                                                                       \
		hipError_t _e = (expr);                                                                                \
		if (_e != hipSuccess)                                                                                  \
			throw ::blmx::StatusError{_e == hipErrorOutOfMemory ? BLMX_ENOMEM : BLMX_EHIP,                      \
			                          std::string(#expr) + ": " + hipGetErrorString(_e)};                      \
	} while (0)

struct ShardExchange;

struct Index {
	blmx_params prm{};
	int KW = 1;  // u64 words per k-mer
	int gshift = 0;
	uint32_t nbuckets = 0, ngroups = 0;
	bool built = false;
	uint32_t shard_rank = 0, shard_world = 1;
	std::vector<uint32_t> shard_bounds;  // world + 1 group boundaries (identical on every rank)
	hipStream_t stream = nullptr;
	std::mutex mu;  // serialises the large host-buffer / file operations (they share the window slots)
	std::atomic<uint64_t> number_query{0};  // blight.h:56; incremented atomically like the reference (blight.cpp:687-688, 717-718)
	HostFront* hf = nullptr;                // created on first use (blmx_stream.hip)
	BurstState* burst = nullptr;            // created on first use (blmx_serve.hip)
	int tune_serve_burst = 1;               // small calls are answered by a bounded serving kernel (0: one launch per round of the combiner)
	ShardExchange* xb = nullptr;            // state of a sharded build between its phases (blmx_build.hip)
	std::mutex hf_mu;

	// finished index (HBM)
	DevBuf buckets, groups, qheads, seq, positions, lines, fb_keys, fb_ranks, fb_table, fb_dir, filter, filter_gate;
	uint64_t filter_words = 0;
	uint32_t filter_bpk = 0;   // bits per key the filter was built with (0 = none)
	bool fat_lines = false;    // level-0 lines carry the filter (64-B lines; no separate filter array)
	bool rec_lines = false;    // ... as records of 32 level bits with the position fields of their keys (index_device.h); implies fat_lines
	// The filter's passes (k_filter_keys: every key; k_filter_junctions: the few non-keys the reference's scan can find) run on their
	// own stream AFTER the build has returned, and a helper thread waits for it, gives the build's temporaries back to the driver (hipFree waits for the
	// device, so the build itself must not do it) and arms the filter.  0 = pass pending, 2 = filter usable.
	mutable std::atomic<int> filter_state{2};
	DevBuf filter_pending[7];                         // what the passes read: super-k-mer starts, their minimizers, key offsets, keys (buffer of
	                                                  // one build pass), and -- several build passes -- what re-expands the keys: text, order, positions
	std::vector<BuildPool::Block> deferred_blocks;    // temporaries of the build, freed by the helper thread
	std::thread filter_thread;
	hipStream_t bg_stream = nullptr;
	hipEvent_t filter_ready = nullptr;
	bool filter_usable() const { return filter_state.load(std::memory_order_acquire) == 2; }
	void start_filter_thread()
	{
		if (filter_state.load() != 0 || filter_thread.joinable())
			return;
		filter_thread = std::thread([this] {
			(void)hipSetDevice(prm.device);
			(void)hipEventSynchronize(filter_ready);
			for (DevBuf& b : filter_pending)
				b.release();
			for (const BuildPool::Block& b : deferred_blocks)
				ScratchCache::get().give(b.p, b.bytes);
			deferred_blocks.clear();
			std::lock_guard<std::mutex> lock(late_mu);
			for (const BuildPool::Block& q : late_free)
				ScratchCache::get().give(q.p, q.bytes);
			late_free.clear();
			filter_state.store(2, std::memory_order_release);
		});
	}
	// Input buffers of a build whose owner returns while the filter's passes still run: hipFree would wait for them.
	std::mutex late_mu;
	std::vector<BuildPool::Block> late_free;
	void free_after_filter(DevBuf& b)
	{
		std::lock_guard<std::mutex> lock(late_mu);
		if (b.p && !b.owner && filter_state.load(std::memory_order_acquire) == 0) {
			late_free.push_back(BuildPool::Block{b.p, b.cap ? b.cap : b.bytes});
			b.p = nullptr;
			b.bytes = 0;
			b.cap = 0;
		} else
			b.release();
	}
	void filter_wait()  // block until the second pass is done (image writer, destroy, rebuild, blmx_set_tuning("absent_filter", 1))
	{
		if (filter_thread.joinable())
			filter_thread.join();
	}
	bool has_qheads = false;
	// host mirror of the group table (export / stats)
	std::vector<GroupDesc> h_groups;
	std::vector<uint32_t> h_fb_group;
	blmx_stats_t st{};
	uint64_t positions_words = 0, seq_words = 0, total_lines = 0, total_logical_words = 0;
	uint32_t pos_hint = 0;  // extra low bits per position field in HBM (0 = the reference's field width)
	// shards: nucleotides of the NEXT owners' first buckets kept behind this shard's sequence (what the single index holds there);
	// halo_done = false between blmx_shard_build_finish and blmx_shard_halo_set (windows at the shard's end still see padding)
	uint64_t halo_nt = 0;
	bool halo_done = true;

	// staging for host-buffer queries
	DevBuf q_in, q_aux, q_out;
	// scratch of the locality-binned batch query (grow-only; guarded by bin_mu)
	DevBuf b_bins, b_minis, b_sorted, b_inv, b_ids, b_hist, b_offs, b_tmp, b_plan, b_frac, b_pstats;
	// k-mers a partition pass could not place (k_bin_onepass: cannot happen -- the overflow region holds a whole batch -- but a k-mer
	// without a slot would come back as "absent", so it is counted): a word in pinned host memory the device adds to, looked at by
	// every later call on this index.  test_ovf_cap (test hook) shrinks the overflow region to provoke it.
	unsigned long long* h_dropped = nullptr;
	int tune_test_ovf_cap = 0;
	int tune_bin_rounds = 2;  // k_bin_onepass: 2 = two workgroups per CU staging half a tile at a time (default), 1 = one per CU staging all of it
	int tune_max_batch = 0;  // test hook: k-mers per piece of a device-resident batch (0 = 2^30)
	unsigned long long* dropped_word()
	{
		if (!h_dropped) {
			if (hipHostMalloc(reinterpret_cast<void**>(&h_dropped), 64, hipHostMallocDefault) != hipSuccess) {
				h_dropped = nullptr;
				(void)hipGetLastError();
			} else
				*h_dropped = 0;
		}
		return h_dropped;
	}
	void check_dropped()
	{
		if (h_dropped) {
			const unsigned long long n = __atomic_load_n(h_dropped, __ATOMIC_RELAXED);
			if (n) {
				__atomic_store_n(h_dropped, 0ull, __ATOMIC_RELAXED);
				throw StatusError{BLMX_EHIP, "internal: " + std::to_string(n) + " k-mers of an earlier batch found no slot in the partition (their answers were -1)"};
			}
		}
	}
	int plan_bits = -1;  // bin count (log2) the learnt shares in b_frac belong to, -1 = nothing learnt yet
	uint32_t plan_nbins_sharded = 0;  // the same for the sharded exchange (bins split at the shard boundaries): bins of b_frac_sharded
	DevBuf b_frac_sharded, b_foreign;  // b_foreign: workgroup layout of the owner's slot kernel (k_foreign_plan)
	std::mutex bin_mu;
	hipEvent_t bin_done = nullptr;       // recorded after each binned pipeline: a call on another stream waits for it
	hipStream_t bin_last_stream = nullptr;
	bool bin_used = false;
	// stage timing of the last query pipeline (blmx_set_profiling): events at the stage boundaries
	// [0] start | [1] front end done | [2] k_bin_hist | [3] scan | [4] k_bin_scatter | [5] k_query | [6] k_gather_ids
	static constexpr int kProfEvents = 7;
	bool prof_on = false;
	hipEvent_t prof_ev[kProfEvents] = {};
	bool prof_set[kProfEvents] = {};
	int bin_policy = 0;  // 0 = automatic, 1 = always bin, -1 = never (BLIGHT_QUERY_BINNING env at creation / blmx_set_query_binning)
	// per-index tuning knobs (blmx_set_tuning; the BLIGHT_* environment variables of the same name seed them at creation)
	int tune_bin_bits = 0;       // partition bins = 2^bits (0 = default)
	int tune_mini_handoff = -1;  // partition pass hands the minimizers to the query kernel: 1 / 0 / -1 = automatic
	int tune_query_heads = 1;    // 0 = query kernels read the full group descriptor instead of the compact 32-B head
	int tune_reads_mode = 0;     // sequence / file queries: 0 = automatic, 1 = always fused front end + query, 2 = always front end -> binned pipeline
	uint32_t staged_tile = 0;    // tile of the last blmx_partition_device (blmx_gather_ids_device reads the order back tile by tile)
	int tune_partition = 0;      // batch partition: 0 = exact for the first batch, then one pass into bin regions sized after the previous batch;
	                             // 1 = always one pass; 2 = always exact (histogram, scan, scatter)
	int tune_slot_slack_pct = 0; // one-pass partition: slots per bin = mean occupancy + this many percent (0 = default 12)
	int tune_narrow_ids = 1;     // host-buffer results cross PCIe as 32-bit words when every ID fits (0 = always 64-bit)
	int tune_absent_filter = 1;  // queries consult the absent-key filter (when the index was built with one)
	int tune_pos_hint = 1;          // build: position fields keep position_hint_bits(b) more low bits (0 = the reference's fields)
	int tune_filter_gate = 1;       // large batches with almost no absent k-mers switch the filter off for the next ones (0 = never)
	int tune_filter_junctions = 1;  // test hook: 0 = skip the filter's second pass (the filter then loses the reference's false positives)
	int tune_filter_bits = -1;   // bits per key of the filter built at construction (0 = build none, -1 = automatic: 8, or 16 with the
	                             // position hint of the wide windows; BLIGHT_FILTER_BITS)
	int tune_window_kb = 0;      // characters per front-end window in KiB (0 = default: 32 MiB)
	int tune_fat_lines = -1;     // build: with 8 filter bits per key the filter lives inside 64-B level-0 units: 0 = separate filter array and 16-B
	                             // lines, 1 = fat lines (96 level bits + filter), 2 = records (32 level bits + filter + position fields),
	                             // -1 = automatic = fat lines (records measured slower on a 50/50 batch, profiles/r04_exp_level0_layouts_c4.txt)
	int tune_fb_table = 1;       // the fallback entries also as per-group hash tables (0: lookups search the sorted entries, as in r03)
	int tune_build_passes = 0;   // build: passes over ranges of MPHF groups (0 = automatic: at most 2^log2_superbuckets, <= 2^30 keys per pass)
	int tune_slot_piece_min = 0; // sharded exchange: pieces of at least this many k-mers travel as bin regions (0 = default 2^19; tests lower it)

	Index()
	{
		for (DevBuf* b : {&buckets, &groups, &qheads, &seq, &positions, &lines, &fb_keys, &fb_ranks, &fb_table, &fb_dir, &filter, &filter_gate})
			b->persistent = true;
	}

	void prof_mark(int i, hipStream_t s)
	{
		if (!prof_on)
			return;
		if (!prof_ev[i])
			(void)hipEventCreate(&prof_ev[i]);
		(void)hipEventRecord(prof_ev[i], s);
		prof_set[i] = true;
	}
	void prof_reset()
	{
		for (bool& b : prof_set)
			b = false;
	}

	DeviceIndexView view() const
	{
		DeviceIndexView v;
		v.k = (int)prm.k;
		v.m = (int)prm.m;
		v.n = (int)prm.log2_mphf;
		v.b = (int)prm.bits_saved;
		v.hint = (int)pos_hint;
		v.gshift = gshift;
		v.nbuckets = nbuckets;
		v.ngroups = ngroups;
		v.buckets = buckets.as<BucketDesc>();
		v.groups = groups.as<GroupDesc>();
		v.qheads = (has_qheads && tune_query_heads) ? qheads.as<QueryHead>() : nullptr;
		v.seq = seq.as<uint64_t>();
		v.positions = positions.as<uint64_t>();
		v.lines = lines.as<uint64_t>();
		v.fb_keys = fb_keys.as<uint64_t>();
		v.fb_ranks = fb_ranks.as<uint64_t>();
		v.fb_table = fb_table.as<uint64_t>();
		v.fb_dir = fb_dir.p ? fb_dir.as<uint64_t>() : nullptr;
		v.total_nt = st.total_nt;
		v.fat = fat_lines;
		v.rec = rec_lines;
		v.filter = (filter_bpk && tune_absent_filter && filter_usable()) ? (fat_lines ? lines.as<uint64_t>() : filter.as<uint64_t>()) : nullptr;
		v.filter_bpk = filter_bpk;
		v.filter_gate = (v.filter && tune_filter_gate) ? filter_gate.as<uint32_t>() : nullptr;
		v.id_base = h_groups.empty() ? 0 : h_groups[0].prefix;
		return v;
	}
};

// Completion signal of a small launch (the combiner of blmx_front.hip): the last wave / workgroup of the launch writes `ticket` to a
// flag in pinned host memory behind a system-scope fence, and the host spins on that word instead of going through
// hipStreamSynchronize (which costs more than the kernel itself for a launch of a few waves).  counter lives in HBM and only ever
// grows: the launch is complete when it reaches `target`.  flag == nullptr: no signal.
struct DoneSignal {
	unsigned int* counter;
	volatile unsigned int* flag;
	unsigned int target, ticket;
};

// build (blmx_build.hip)
void make_query_heads(Index& ix);  // from ix.h_groups, after the group table is final (build and load)
void make_fallback_table(Index& ix, hipStream_t s);  // from fb_keys / fb_ranks / h_fb_group / h_groups (build and load)
void init_filter_gate(Index& ix, hipStream_t s);  // with the filter (build and load)
void positions_to_reference_layout(const Index& ix, DevBuf& out, hipStream_t s);  // `positions` with the reference's field width (export)
void build_from_device(Index& ix, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len);
// sharded build from slices, phase by phase (blmx_build.hip); hist_out holds ngroups + 1 words: key histogram, local super-k-mers
// packed != nullptr: the slice's text is already 2-bit packed (taken over; d_ascii unused)
void shard_build_begin(Index& ix, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len, uint64_t* hist_out, DevBuf* packed = nullptr);
// sequences [u0, u1) of a 2-bit text with device offsets -> a text of their own (offsets relative to its start)
void slice_packed_text(const DevBuf& d_text, const uint64_t* d_offs_all, uint64_t u0, uint64_t u1, DevBuf& out_text, DevBuf& out_offs, uint64_t& total_len,
                       hipStream_t s);
void shard_build_plan(Index& ix, const uint64_t* global_hist, uint32_t world, uint64_t* send_sk, uint64_t* send_words, const void** d_meta,
                      const void** d_words);
void shard_build_finish(Index& ix, const void* d_meta_in, const void* d_words_in, const uint64_t* recv_sk, const uint64_t* recv_words);
void shard_build_abort(Index& ix);
uint64_t shard_halo_words(const Index& ix);  // u64 words of one halo record
void shard_halo_get(Index& ix, uint64_t* rec);
void shard_halo_set(Index& ix, const uint64_t* recs, uint32_t world);

// query (blmx_query.hip).  mode: 0 = IDs (query_kmer_hash / query_get_hash), 1 = positions (query_get_pos_unitig),
// 2 = membership byte.  d_minis == nullptr -> minimizers are computed on the device.
enum QueryMode { kQueryIds = 0, kQueryPos = 1, kQueryBool = 2, kQueryIds32 = 3 /* IDs as 32-bit words, 0xffffffff = absent */,
                 kQueryXcdMajor = 0x100 /* flag: XCD-major block order */,
                 kQueryNoFilter = 0x200 /* flag: the minimizers come from the caller and may be wrong: the absent-key filter only knows
                                           what a query with the k-mer's OWN minimizer can find */ };
void launch_query(const Index& ix, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, void* d_out, QueryMode mode,
                  hipStream_t s, DoneSignal done = DoneSignal{nullptr, nullptr, 0, 0});
unsigned int query_signal_units(uint64_t nq);  // what a launch_query of nq k-mers adds to DoneSignal::counter (its waves)
// Large batches: partition the queries by the top bits of their minimizer first (one counting pass), run the query
// kernel on the partitioned batch, gather the IDs back into the caller's order.
void launch_query_ids_auto(Index& ix, const void* d_kmers, uint64_t nq, int64_t* d_out, hipStream_t s);
void launch_partition(Index& ix, const void* d_kmers, uint64_t nq, void* d_sorted, uint32_t* d_inv, uint64_t* d_rank_counts, hipStream_t s);
void launch_query_partitioned(Index& ix, const void* d_sorted_kmers, uint64_t nq, int64_t* d_ids, hipStream_t s);
void launch_gather_ids(Index& ix, const int64_t* d_ids_sorted, const uint32_t* d_inv, uint64_t nq, int64_t* d_out, hipStream_t s);
void launch_kmer_groups(const Index& ix, const void* d_kmers, uint64_t nq, uint32_t* d_groups, hipStream_t s);
// ---- the one-pass partition as the sender's side of the sharded exchange (blmx_query.hip).  A piece of a batch is partitioned into
// bin regions (slots) exactly like a single-GPU batch; the bins are split at the shard boundaries, so the regions of owner r's bins
// are ONE contiguous slot range [seg[r], seg[r + 1]) that travels as it is (k-mers, their minimizers, and a header word per bin:
// first slot | slots used); the owner answers the used slots of every bin (launch_query_foreign_slots) and the IDs come back into the
// same slot range, narrow (4 B) when every ID fits; k_gather_runs then brings them into the caller's order.
struct SlotPiece {
	DevBuf slots, minis, ids, lpos, plan, cursors, hdr, info;
	Pinned h_info;         // after a stream synchronisation: [0] k-mers in the overflow region, [1 + r] = seg[r] for r = 0 .. world
	uint64_t nq = 0, max_main = 0;
	uint32_t nbins = 0, nblk = 0, tile = 0, world = 1;
	bool narrow = false;
	uint32_t first_bin[17] = {};
	size_t lpos_bytes = 0, runs = 0;
	const uint32_t* seg() const { return h_info.as<uint32_t>() + 1; }
	uint32_t overflow() const { return h_info.as<uint32_t>()[0]; }
};
bool slot_piece_possible(const Index& ix, uint64_t n);
uint32_t slot_owner_bins(const Index& ix, uint32_t owner);  // bins of the sharded partition that belong to `owner` (the same on every rank)
void slot_piece_partition(Index& ix, SlotPiece& sp, const void* d_kmers, const uint32_t* d_minis, uint64_t n, hipStream_t s);
struct ForeignSlots {       // what an owner answers: the slot range one sender holds for it
	const uint64_t* slots;  // k-mers (KW words each), slot 0 = the sender's slot seg[me]
	const uint32_t* minis;
	void* ids;              // out: one word per slot (4 B narrow / 8 B)
	const uint64_t* hdr;    // nb + 1 words: low half = first slot of the bin's region (sender's numbering), high half = slots used
	uint32_t nslots, nb;
};
void launch_query_foreign_slots(Index& ix, const ForeignSlots* src, uint32_t nsrc, bool narrow, hipStream_t s);
void slot_piece_gather(Index& ix, SlotPiece& sp, int64_t* d_out, hipStream_t s);
// same with the minimizers already known (read front end): the partition pass reads them instead of hashing 23 m-mers per
// k-mer, and they travel with the k-mers to the query kernel
void launch_query_ids_with_minis(Index& ix, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, int64_t* d_out, hipStream_t s);
bool query_batch_worth_binning(const Index& ix, uint64_t nq);
// sequence / read front end (blmx_reads.hip).  A window = nbytes characters in device-visible memory; k-mers may start at its
// first npos positions.  front_valid_* fill fs.valid / fs.vpre; launch_read_front writes canonical k-mers + minimizers in
// position order, launch_read_query answers them in the same launch (ids and / or found / not-found counters).  With
// fs == nullptr the k-mer positions come from a binary search in offs / koffs (few sequences; pointers may be pinned host memory).
void front_valid_concat(const Index& ix, FrontScratch& fs, const uint64_t* d_offs, uint64_t nseq, uint64_t origin, uint64_t npos, uint64_t nbytes,
                        hipStream_t s);
void front_valid_raw(const Index& ix, FrontScratch& fs, const uint8_t* d_bytes, uint64_t nbytes, int mode, const uint64_t* d_carry_in,
                     uint64_t* d_carry_out, int* d_short_flag, hipStream_t s);
void launch_read_front(const Index& ix, const FrontScratch* fs, const uint8_t* d_bytes, uint64_t nbytes, uint64_t npos, const uint64_t* offs,
                       const uint64_t* koffs, uint64_t nseq, void* d_kmers, uint32_t* d_minis, int* d_err, hipStream_t s);
void launch_read_query(const Index& ix, const FrontScratch* fs, const uint8_t* d_bytes, uint64_t nbytes, uint64_t npos, const uint64_t* offs,
                       const uint64_t* koffs, uint64_t nseq, int64_t* d_ids, unsigned long long* d_counts, int* d_err, hipStream_t s,
                       DoneSignal done = DoneSignal{nullptr, nullptr, 0, 0});
unsigned int read_query_signal_units(uint64_t npos);  // what a launch_read_query over npos positions adds to DoneSignal::counter (its workgroups)
bool read_query_fusable(const Index& ix);
void front_seq_lines(const Index& ix, FrontScratch& fs, const uint8_t* d_bytes, uint64_t nbytes, const uint64_t* d_carry_in, uint64_t* d_carry_out,
                     int* d_short_flag, uint64_t* d_tot, uint8_t* d_text, uint64_t* d_packed, int* d_dna_flag, uint64_t* d_offs, hipStream_t s);
// construct_index(file) with the parse on the device (blmx_front.hip): raw bytes -> pinned windows -> HBM -> 2-bit text + offsets
void load_fasta_raw_to_device(Index& ix, const char* path, DevBuf& d_packed, DevBuf& d_offs, uint64_t& nseq, uint64_t& total);
// the build from a text that is already 2-bit packed in HBM (first nucleotide of a word most significant, zero behind the end;
// (total + 31) / 32 + 4 words); d_packed is taken over
void build_from_device_packed(Index& ix, DevBuf& d_packed, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len);
// the bounded serving burst (blmx_serve.hip); the queries return false when they do not take the request
bool burst_query_kmers(Index& ix, const void* kmers, uint64_t n, void* out, bool want_bool);
bool burst_query_sequence(Index& ix, const char* chars, uint64_t nbytes, uint64_t nk, int64_t* out);
void burst_stop(Index& ix);
void burst_destroy(Index& ix);
void burst_stats(Index& ix, uint64_t out[4]);
uint64_t burst_number_query(const Index& ix);  // k-mers the burst answered (kept per slot, added to number_query when it is read)
// host-buffer entry points (blmx_stream.hip): pinned staging, windows in flight, small-call slots
void query_sequences_host(Index& ix, const char* concat, const uint64_t* offsets, uint64_t nseq, int64_t* ids_out);
void query_kmers_host(Index& ix, const void* kmers, const uint32_t* minis, uint64_t nq, void* out, QueryMode mode);
void query_sequences_device(Index& ix, const uint8_t* d_chars, const uint64_t* d_offs, uint64_t nseq, uint64_t total, uint64_t nk, int64_t* d_ids,
                            int* d_err, hipStream_t s);
// found / not-found counters are kCountStripes (found, not found) pairs that the caller sums (one hot counter serialises)
constexpr int kCountStripes = 64;
void launch_count_found(const int64_t* d_ids, uint64_t n, unsigned long long* d_counts, hipStream_t s);
void launch_narrow_ids(const int64_t* d_ids, uint64_t n, uint32_t* d_out, hipStream_t s);
void launch_kmers_at(const Index& ix, const uint8_t* d_ascii, const uint64_t* d_pos, uint64_t nq, void* d_out, int* d_err, hipStream_t s);
void launch_canonicalize(const Index& ix, void* d_kmers, uint64_t nq, hipStream_t s);
void launch_route_count(const Index& ix, const void* d_kmers, uint64_t nq, const uint32_t* d_bounds, uint32_t world, uint8_t* d_dest,
                        uint64_t* d_counts, hipStream_t s);
void launch_route_scatter(const Index& ix, const void* d_kmers, const uint8_t* d_dest, uint64_t nq, const uint64_t* d_offsets, uint32_t world,
                          void* d_sorted, uint32_t* d_order, uint64_t* d_cursor, hipStream_t s);
void launch_unpermute(const int64_t* d_vals, const uint32_t* d_order, uint64_t nq, int64_t* d_out, hipStream_t s);
// native multi-GPU exchange (blmx_comm.hip)
struct Comm;
void comm_unique_id(void* id_out);
Comm* comm_create(const void* id_bytes, uint32_t rank, uint32_t world, int device);
void comm_destroy(Comm* c);
void comm_loopback_id(void* id_out);
void comm_set_tuning(Comm& c, const char* key, int64_t value);
const char* comm_transport(const Comm* c);
void build_sharded(Index& ix, Comm& c, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len,
                   const StatusError* already_failed = nullptr, DevBuf* packed = nullptr);
void build_sharded_from_fasta(Index& ix, Comm& c, const char* path);
void query_kmers_sharded_host(Index& ix, Comm& c, const void* kmers, uint64_t nq, int64_t* ids_out);
void query_sequences_sharded_device(Index& ix, Comm& c, const uint8_t* d_chars, const uint64_t* d_offs, uint64_t nseq, uint64_t total, uint64_t nk,
                                    int64_t* d_ids, int* d_err, hipStream_t s);
void query_sequences_sharded_host(Index& ix, Comm& c, const char* concat, const uint64_t* offsets, uint64_t nseq, int64_t* ids_out);
void load_fasta_to_device(Index& ix, const char* path, DevBuf& d_ascii, std::vector<uint64_t>& offs, uint64_t& total_out);
void query_kmers_sharded(Index& ix, Comm& c, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, int64_t* d_ids, hipStream_t s);
// streaming front end (blmx_stream.hip)
void query_file(Index& ix, const char* path, uint64_t& good, uint64_t& erroneous);
void build_from_fasta_streamed(Index& ix, const char* path);
// payload applications on top of the IDs (reference Colored_/Abundance_De_Bruijn_graph_snippet.cpp), blmx_stream.hip
struct Payload {
	Index* ix = nullptr;
	int kind = 0;           // 0 = abundance (uint8 per k-mer), 1 = colours (bit matrix, ncolors bits per k-mer)
	uint32_t ncolors = 1;
	uint64_t nids = 0;      // number_kmer of the index
	uint64_t bytes = 0;     // payload size in bytes (device buffer is padded to 4)
	DevBuf data;
};
void payload_init(Payload& pl);
void payload_add_ids(Payload& pl, const int64_t* d_ids, uint64_t n, uint32_t color, hipStream_t s);
void payload_add_file(Payload& pl, const char* path, uint32_t color, uint64_t& found, uint64_t& absent);
// index image on disk (blmx_stream.hip)
void save_index(Index& ix, const char* path);
void load_index(Index& ix, const char* path);
int peek_index_params(const char* path, blmx_params& out);
void launch_primitive(int op, int KW, const void* d_in, uint64_t n, int arg, int arg2, void* d_out, hipStream_t s);
void launch_sort_primitive(const uint32_t* d_keys, uint64_t n, unsigned bits, uint32_t* d_out, hipStream_t s);  // blmx_build.hip

}  // namespace blmx
