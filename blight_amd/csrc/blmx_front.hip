// blmx_front.hip -- host side of the host-buffer and file entry points: query_sequence_hash / query_sequence_bool
// (reference blight.cpp:554-591), query_kmer_hash / query_kmer_bool for batches (blight.cpp:537-550) and file_query
// (blight.cpp:746-799) + the payload applications on top of it.
//
// The host never parses or encodes: it moves bytes.
//   * One call on a few short sequences (what the reference's callers do: one query_sequence_hash per read, from OpenMP
//     threads): the characters are copied into a pinned slot that the device reads directly, ONE fused kernel (front end +
//     query body, blmx_reads.hip) writes the IDs straight back into the pinned slot, one stream synchronisation.  No
//     allocation, no hipMemcpy, no lock around the GPU work: callers run concurrently on a pool of slots.
//   * Large inputs go through two window slots (pinned staging + device scratch that persist with the index): copy-in by
//     several host threads | H2D | kernels | D2H | copy-out overlap across windows.
//   * Files are read into the pinned window as RAW bytes (pread by several threads, or gzread) and cut at a newline; lines,
//     the reference's record rule and k-mer starts are found on the device (blmx_reads.hip).  Only the two recap counters
//     come back.
#include <fcntl.h>
#include <immintrin.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>

#include "blmx_internal.h"

namespace blmx {

namespace {

// ---- persistent host threads for parallel copies and reads -------------------------------------------------------------------
class WorkerPool;
using WorkerPoolRef = WorkerPool&;
class WorkerPool {
public:
	explicit WorkerPool(unsigned nthreads)
	{
		for (unsigned i = 1; i < nthreads; i++)
			th_.emplace_back([this] { loop(); });
	}
	~WorkerPool()
	{
		{
			std::lock_guard<std::mutex> l(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		for (auto& t : th_)
			t.join();
	}
	// f(i) for i in [0, njobs); returns when all are done (the caller works too).  One run at a time.
	void run(size_t njobs, const std::function<void(size_t)>& f)
	{
		if (njobs == 0)
			return;
		if (njobs == 1 || th_.empty()) {
			for (size_t i = 0; i < njobs; i++)
				f(i);
			return;
		}
		{
			std::lock_guard<std::mutex> l(mu_);
			job_ = &f;
			njobs_ = njobs;
			next_.store(0);
			active_ = th_.size();
			gen_++;
		}
		cv_.notify_all();
		work();
		std::unique_lock<std::mutex> l(mu_);
		cv_done_.wait(l, [this] { return active_ == 0; });
		job_ = nullptr;
	}
	unsigned size() const { return (unsigned)th_.size() + 1; }

private:
	void work()
	{
		for (;;) {
			const size_t i = next_.fetch_add(1);
			if (i >= njobs_)
				break;
			(*job_)(i);
		}
	}
	void loop()
	{
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> l(mu_);
				cv_.wait(l, [&] { return stop_ || gen_ != seen; });
				if (stop_)
					return;
				seen = gen_;
			}
			work();
			std::lock_guard<std::mutex> l(mu_);
			if (--active_ == 0)
				cv_done_.notify_one();
		}
	}
	std::vector<std::thread> th_;
	std::mutex mu_;
	std::condition_variable cv_, cv_done_;
	const std::function<void(size_t)>* job_ = nullptr;
	size_t njobs_ = 0;
	std::atomic<size_t> next_{0};
	size_t active_ = 0;
	uint64_t gen_ = 0;
	bool stop_ = false;
};

constexpr size_t kCopyPiece = 4u << 20;

// BLIGHT_TRACE=1: wall time of the host phases of the window pipelines on stderr
struct Trace {
	bool on;
	std::chrono::steady_clock::time_point t;
	Trace() : on(getenv("BLIGHT_TRACE") && atoi(getenv("BLIGHT_TRACE")) != 0), t(std::chrono::steady_clock::now()) {}
	void lap(const char* what, uint64_t window)
	{
		if (!on)
			return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[blmx front] window %3llu %-18s %8.3f ms\n", (unsigned long long)window, what, std::chrono::duration<double, std::milli>(now - t).count());
		t = now;
	}
};

// Large staging copies write memory that the CPU does not read again (a pinned buffer on its way to the GPU, or the caller's
// result array): non-temporal stores skip the read-for-ownership of the destination lines.
__attribute__((target("avx2"))) void stream_copy_avx2(char* dst, const char* src, size_t n)
{
	while (n && ((uintptr_t)dst & 31)) {
		*dst++ = *src++;
		n--;
	}
	for (; n >= 128; n -= 128, dst += 128, src += 128) {
		const __m256i a = _mm256_loadu_si256((const __m256i*)src), b = _mm256_loadu_si256((const __m256i*)(src + 32)),
		              c = _mm256_loadu_si256((const __m256i*)(src + 64)), d = _mm256_loadu_si256((const __m256i*)(src + 96));
		_mm256_stream_si256((__m256i*)dst, a);
		_mm256_stream_si256((__m256i*)(dst + 32), b);
		_mm256_stream_si256((__m256i*)(dst + 64), c);
		_mm256_stream_si256((__m256i*)(dst + 96), d);
	}
	_mm_sfence();
	if (n)
		memcpy(dst, src, n);
}

// IDs cross PCIe as 32-bit words when the index holds fewer than 2^32 - 1 k-mers (0xffffffff = absent): widen on the way out
__attribute__((target("avx2"))) void widen_ids_avx2(int64_t* dst, const uint32_t* src, size_t n)
{
	const __m256i absent = _mm256_set1_epi64x(0xffffffffll);
	size_t i = 0;
	if (((uintptr_t)dst & 31) == 0)
		for (; i + 8 <= n; i += 8) {
			const __m256i a = _mm256_cvtepu32_epi64(_mm_loadu_si128((const __m128i*)(src + i)));
			const __m256i b = _mm256_cvtepu32_epi64(_mm_loadu_si128((const __m128i*)(src + i + 4)));
			_mm256_stream_si256((__m256i*)(dst + i), _mm256_or_si256(a, _mm256_cmpeq_epi64(a, absent)));
			_mm256_stream_si256((__m256i*)(dst + i + 4), _mm256_or_si256(b, _mm256_cmpeq_epi64(b, absent)));
		}
	_mm_sfence();
	for (; i < n; i++)
		dst[i] = src[i] == 0xffffffffu ? -1 : (int64_t)src[i];
}

void widen_ids(WorkerPoolRef pool, int64_t* dst, const uint32_t* src, size_t n);

void copy_piece(char* dst, const char* src, size_t n)
{
	static const bool nt = __builtin_cpu_supports("avx2") && !(getenv("BLIGHT_NT_COPY") && atoi(getenv("BLIGHT_NT_COPY")) == 0);
	if (nt && n >= (256u << 10))
		stream_copy_avx2(dst, src, n);
	else
		memcpy(dst, src, n);
}

void parallel_copy(WorkerPool& pool, void* dst, const void* src, size_t n)
{
	if (n <= kCopyPiece) {
		copy_piece((char*)dst, (const char*)src, n);
		return;
	}
	const size_t pieces = (n + kCopyPiece - 1) / kCopyPiece;
	pool.run(pieces, [&](size_t i) {
		const size_t a = i * kCopyPiece, z = std::min(n, a + kCopyPiece);
		copy_piece((char*)dst + a, (const char*)src + a, z - a);
	});
}

void widen_ids(WorkerPoolRef pool, int64_t* dst, const uint32_t* src, size_t n)
{
	static const bool avx2 = __builtin_cpu_supports("avx2");
	const size_t piece = 1u << 20;
	pool.run((n + piece - 1) / piece, [&](size_t i) {
		const size_t a = i * piece, z = std::min(n, a + piece);
		if (avx2) {
			widen_ids_avx2(dst + a, src + a, z - a);
		} else {
			for (size_t j = a; j < z; j++)
				dst[j] = src[j] == 0xffffffffu ? -1 : (int64_t)src[j];
		}
	});
}

// ---- slots -----------------------------------------------------------------------------------------------------------------------
struct WindowSlot {
	Pinned h_in, h_aux, h_out, h_flags;
	DevBuf d_in, d_aux, d_aux2, d_kmers, d_minis, d_ids, d_flags;
	FrontScratch fs;
	hipStream_t s = nullptr;
	hipEvent_t done = nullptr;
	bool in_flight = false;
	uint64_t nk = 0, out_index = 0;  // results of the window in flight: count and where they go in the caller's array
	size_t out_elem = 8;
	bool narrow = false;  // the IDs in h_out are 32-bit words
	~WindowSlot()
	{
		if (done)
			(void)hipEventDestroy(done);
		if (s)
			(void)hipStreamDestroy(s);
	}
};

constexpr size_t kSmallSlotBytes = 1u << 20;
constexpr unsigned kSmallSlots = 8;
constexpr uint64_t kSmallChars = 48u << 10;  // one fused launch answers up to this many characters ...
constexpr uint64_t kSmallSeqs = 1024;        // ... of up to this many sequences, read directly from the pinned slot
constexpr uint64_t kSmallKmers = 8192;       // k-mer batches up to this size: one launch on a pinned slot

struct SmallSlot {
	Pinned h;
	hipStream_t s = nullptr;
	DevBuf counter;            // completion counter of the slot's launches (DoneSignal): only ever grows
	unsigned int target = 0, ticket = 0;
	~SmallSlot()
	{
		if (s) {
			(void)hipStreamSynchronize(s);
			(void)hipStreamDestroy(s);
		}
	}
};

// One small call waiting to be answered.  Calls that arrive while a launch is in flight are answered TOGETHER by the next one: a
// caller either leads (packs everything that waits into one pinned slot, launches once, hands the answers out) or spins on its
// request until a leader has answered it.  With c concurrent callers a launch carries up to c calls, so the per-call cost of the
// launch and of its completion falls with the number of callers instead of being paid by each (reference: every caller runs the
// query on its own core, blight.cpp:537-550, 575-591).
struct SmallReq {
	enum Kind { kSequences, kKmers } kind;
	// sequences: characters [a0, a0 + nbytes) of `chars`, nseq sequences with offsets `offs` (absolute), koffs = k-mers before each
	const char* chars = nullptr;
	const uint64_t* offs = nullptr;
	const uint64_t* koffs = nullptr;
	uint64_t a0 = 0, nbytes = 0, nseq = 0;
	// k-mers
	const void* kmers = nullptr;
	bool want_bool = false;
	uint64_t n = 0;  // answers (k-mers)
	void* out = nullptr;
	std::atomic<int> state{0};  // 0 waiting | 1 answered | 2 failed (code, msg) | 3 "you lead now"
	int code = 0;
	std::string msg;
};

void grow(DevBuf& b, size_t bytes, hipStream_t s)
{
	if (bytes > b.bytes)
		BLMX_HIP(b.alloc(bytes + bytes / 4 + 256, false, s));
}

}  // namespace

constexpr int kWindowSlots = 3;  // copy-in | H2D + kernels + D2H | copy-out of three windows overlap

struct HostFront {
	WindowSlot win[kWindowSlots];
	std::vector<std::unique_ptr<SmallSlot>> small_all;
	std::vector<SmallSlot*> small_free;
	std::mutex small_mu;
	std::condition_variable small_cv;
	std::unique_ptr<WorkerPool> pool;
	std::mutex comb_mu;                 // the combiner of small calls: requests that wait for a launch, and whether somebody leads
	std::vector<SmallReq*> comb_wait;
	bool comb_led = false;
	std::atomic<int> comb_nwait{0};  // comb_wait.size(), readable without the lock
	std::atomic<int> comb_peak{1};   // calls the recent launches carried (decays): what a new leader may expect to gather
	// BLIGHT_TRACE: what the combiner's rounds carried and where their time went (printed when the index is destroyed)
	const bool comb_trace = getenv("BLIGHT_TRACE") && atoi(getenv("BLIGHT_TRACE")) != 0;
	std::atomic<uint64_t> cs_rounds{0}, cs_calls{0}, cs_gather_ns{0}, cs_pack_ns{0}, cs_launch_ns{0}, cs_wait_ns{0}, cs_hand_ns{0};
	DevBuf d_counts, d_carry;  // file streaming: found / not found, line carry between windows
	// scratch of blmx_query_sequences_device (calls on different streams are ordered with an event)
	FrontScratch dev_fs;
	DevBuf dev_kmers, dev_minis, dev_err;
	std::mutex dev_mu;
	hipEvent_t dev_done = nullptr;
	hipStream_t dev_last = nullptr;
	bool dev_used = false;
	~HostFront()
	{
		if (comb_trace && cs_rounds.load()) {
			const double r = (double)cs_rounds.load();
			fprintf(stderr, "[blmx front] small calls: %llu launches carried %llu calls (%.2f each); per launch: gather %.2f us, pack %.2f us, launch %.2f us, wait %.2f us, hand out %.2f us\n",
			        (unsigned long long)cs_rounds.load(), (unsigned long long)cs_calls.load(), cs_calls.load() / r, cs_gather_ns.load() / r / 1e3, cs_pack_ns.load() / r / 1e3,
			        cs_launch_ns.load() / r / 1e3, cs_wait_ns.load() / r / 1e3, cs_hand_ns.load() / r / 1e3);
		}
		if (dev_done)
			(void)hipEventDestroy(dev_done);
	}
};

void destroy_host_front(HostFront* hf) { delete hf; }

namespace {

HostFront& host_front(Index& ix)
{
	if (HostFront* have = __atomic_load_n(&ix.hf, __ATOMIC_ACQUIRE))  // (small calls come from many threads: no lock once it exists)
		return *have;
	std::lock_guard<std::mutex> l(ix.hf_mu);
	if (!ix.hf) {
		std::unique_ptr<HostFront> hf(new HostFront());
		const char* e = getenv("BLIGHT_HOST_THREADS");
		if (!e)
			e = getenv("BLIGHT_PARSE_THREADS");
		const unsigned hw = std::thread::hardware_concurrency();
		unsigned nt = e ? (unsigned)atoi(e) : (hw && hw < 16 ? hw : 16u);
		nt = nt < 1 ? 1u : (nt > 64 ? 64u : nt);
		hf->pool.reset(new WorkerPool(nt));
		__atomic_store_n(&ix.hf, hf.release(), __ATOMIC_RELEASE);
	}
	return *ix.hf;
}

void prepare_window_slot(WindowSlot& w)
{
	if (!w.s)
		BLMX_HIP(hipStreamCreateWithFlags(&w.s, hipStreamNonBlocking));
	if (!w.done)
		BLMX_HIP(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
	w.h_flags.ensure(64);
	if (!w.d_flags.p)
		BLMX_HIP(w.d_flags.alloc(64, true, w.s));
}

struct SmallLease {
	HostFront& hf;
	SmallSlot* sl;
	SmallLease(Index& ix, HostFront& h) : hf(h), sl(nullptr)
	{
		std::unique_lock<std::mutex> l(hf.small_mu);
		for (;;) {
			if (!hf.small_free.empty()) {
				sl = hf.small_free.back();
				hf.small_free.pop_back();
				return;
			}
			if (hf.small_all.size() < kSmallSlots) {
				std::unique_ptr<SmallSlot> n(new SmallSlot());
				BLMX_HIP(hipSetDevice(ix.prm.device));
				n->h.ensure(kSmallSlotBytes);
				memset(n->h.p, 0, n->h.bytes);  // (a recycled pinned block: the completion flag must start below every target)
				BLMX_HIP(hipStreamCreateWithFlags(&n->s, hipStreamNonBlocking));
				BLMX_HIP(n->counter.alloc(64, true, n->s));
				BLMX_HIP(hipStreamSynchronize(n->s));
				sl = n.get();
				hf.small_all.push_back(std::move(n));
				return;
			}
			hf.small_cv.wait(l);
		}
	}
	~SmallLease()
	{
		{
			std::lock_guard<std::mutex> l(hf.small_mu);
			hf.small_free.push_back(sl);
		}
		hf.small_cv.notify_one();
	}
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// every ID (and -1 as 0xffffffff) fits 32 bits: results cross PCIe as 4-byte words
bool ids_fit_32(const Index& ix) { return ix.tune_narrow_ids && ix.st.number_kmer < 0xffffffffull; }

uint64_t window_positions(const Index& ix)
{
	const uint64_t kb = ix.tune_window_kb > 0 ? (uint64_t)ix.tune_window_kb : 32768;
	return std::min<uint64_t>(kb << 10, 1ull << 31);
}

// front end -> IDs of the nk k-mers of a window whose VALID bitmap is in w.fs (or counters only)
void query_window(Index& ix, WindowSlot& w, uint64_t nbytes, uint64_t npos, uint64_t nk, bool want_ids, unsigned long long* d_counts)
{
	const bool binned = !read_query_fusable(ix) || ix.tune_reads_mode == 2 || (ix.tune_reads_mode == 0 && query_batch_worth_binning(ix, nk));
	int* d_err = w.d_flags.as<int>();
	if (binned) {
		grow(w.d_kmers, nk * 8 * (size_t)ix.KW, w.s);
		grow(w.d_minis, nk * 4, w.s);
		grow(w.d_ids, nk * 8, w.s);
		launch_read_front(ix, &w.fs, w.d_in.as<uint8_t>(), nbytes, npos, nullptr, nullptr, 0, w.d_kmers.p, w.d_minis.as<uint32_t>(), d_err, w.s);
		launch_query_ids_with_minis(ix, w.d_kmers.p, w.d_minis.as<uint32_t>(), nk, w.d_ids.as<int64_t>(), w.s);
		if (d_counts)
			launch_count_found(w.d_ids.as<int64_t>(), nk, d_counts, w.s);
	} else {
		if (want_ids)
			grow(w.d_ids, nk * 8, w.s);
		launch_read_query(ix, &w.fs, w.d_in.as<uint8_t>(), nbytes, npos, nullptr, nullptr, 0, want_ids ? w.d_ids.as<int64_t>() : nullptr, d_counts,
		                  d_err, w.s);
	}
}

}  // namespace

// ---- the combiner of small calls ---------------------------------------------------------------------------------------------------
namespace {

// wait until the launch with this ticket has stored all its answers: spin on the flag the kernel writes (pinned memory), then --
// should that take implausibly long -- fall back to the stream
void wait_small(SmallSlot& sl, volatile unsigned int* flag, unsigned int ticket)
{
	for (int spin = 0; spin < 400000; spin++) {
		if (__atomic_load_n(const_cast<const unsigned int*>(flag), __ATOMIC_ACQUIRE) == ticket)  // (the answers are read behind this)
			return;
		_mm_pause();
	}
	// The flag never came (a launch of this round failed after the slot's target had been raised, or the device is slow beyond reason):
	// the stream says when the round is over, and the slot's count starts afresh -- otherwise every later round on this slot would
	// spin out its 400,000 polls before falling back to the stream.
	BLMX_HIP(hipStreamSynchronize(sl.s));
	BLMX_HIP(hipMemsetAsync(sl.counter.p, 0, 64, sl.s));
	BLMX_HIP(hipStreamSynchronize(sl.s));
	sl.target = 0;
	std::atomic_thread_fence(std::memory_order_acquire);
}

// answer everything in `batch` with (at most) one fused read launch and one k-mer launch on one pinned slot
template <typename AfterLaunch>
void serve_small(Index& ix, HostFront& hf, std::vector<SmallReq*>& batch, AfterLaunch&& after_launch)
{
	SmallLease lease(ix, hf);
	SmallSlot& sl = *lease.sl;
	char* base = sl.h.as<char>();
	const size_t kb = (size_t)ix.KW * 8;
	const bool tr = hf.comb_trace;
	auto now_ns = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const uint64_t t_begin = tr ? now_ns() : 0;
	uint64_t nbytes = 0, nseq = 0, nk_seq = 0, nkm = 0;
	for (const SmallReq* r : batch) {
		if (r->kind == SmallReq::kSequences) {
			nbytes += r->nbytes;
			nseq += r->nseq;
			nk_seq += r->n;
		} else {
			nkm += r->n;
		}
	}
	// slot layout: [flag 64 B][characters][offsets nseq + 1][k-mers before each, nseq + 1][error word][IDs of the sequences][k-mers][IDs of the k-mers]
	const size_t o_chars = 64, o_offs = align_up(o_chars + nbytes + 64, 64), o_koffs = o_offs + (nseq + 1) * 8, o_err = o_koffs + (nseq + 1) * 8,
	             o_ids = align_up(o_err + 8, 64), o_kmers = align_up(o_ids + nk_seq * 8, 64), o_kids = align_up(o_kmers + nkm * kb, 64);
	volatile unsigned int* flag = reinterpret_cast<volatile unsigned int*>(base);
	uint64_t* h_offs = reinterpret_cast<uint64_t*>(base + o_offs);
	uint64_t* h_koffs = reinterpret_cast<uint64_t*>(base + o_koffs);
	int* h_err = reinterpret_cast<int*>(base + o_err);
	*h_err = 0;
	uint64_t cb = 0, cs = 0, ck = 0, cm = 0;
	for (const SmallReq* r : batch) {
		if (r->kind == SmallReq::kSequences) {
			memcpy(base + o_chars + cb, r->chars + r->a0, r->nbytes);
			for (uint64_t i = 0; i < r->nseq; i++) {
				h_offs[cs + i] = cb + (r->offs[i] - r->a0);
				h_koffs[cs + i] = ck + r->koffs[i];
			}
			cb += r->nbytes;
			cs += r->nseq;
			ck += r->n;
		} else {
			memcpy(base + o_kmers + cm * kb, r->kmers, r->n * kb);
			cm += r->n;
		}
	}
	h_offs[nseq] = nbytes;
	h_koffs[nseq] = nk_seq;
	unsigned int units = 0;
	if (nk_seq)
		units += read_query_signal_units(nbytes);
	if (nkm)
		units += query_signal_units(nkm);
	sl.ticket++;
	if (units > 1)
		sl.target += units;
	// a round of ONE wave / workgroup (a k-mer, a read) signals by itself: no device-side count
	const DoneSignal done{units > 1 ? sl.counter.as<unsigned int>() : nullptr, flag, sl.target, sl.ticket};
	const uint64_t t_packed = tr ? now_ns() : 0;
	try {
		if (nk_seq)
			launch_read_query(ix, nullptr, reinterpret_cast<const uint8_t*>(base + o_chars), nbytes, nbytes, h_offs, h_koffs, nseq,
			                  reinterpret_cast<int64_t*>(base + o_ids), nullptr, h_err, sl.s, done);
		if (nkm)
			launch_query(ix, base + o_kmers, nullptr, nkm, base + o_kids, kQueryIds, sl.s, done);
	} catch (...) {
		// one of the launches was not enqueued: the slot's device counter will never reach the target that was raised for this round
		(void)hipStreamSynchronize(sl.s);
		(void)hipMemsetAsync(sl.counter.p, 0, 64, sl.s);
		(void)hipStreamSynchronize(sl.s);
		sl.target = 0;
		throw;
	}
	after_launch();  // (the next leader may gather, pack and launch on another slot while this launch is in flight)
	const uint64_t t_launched = tr ? now_ns() : 0;
	if (units)
		wait_small(sl, flag, sl.ticket);
	const uint64_t t_done = tr ? now_ns() : 0;
	// hand the answers out.  An invalid character fails the sequence calls of this launch that hold one (the reference throws per call).
	const bool bad_char = *h_err != 0;
	ck = cm = 0;
	for (SmallReq* r : batch) {
		if (r->kind == SmallReq::kSequences) {
			bool bad = false;
			if (bad_char)
				for (uint64_t i = 0; i < r->nbytes && !bad; i++) {
					const unsigned char c = (unsigned char)r->chars[r->a0 + i];
					const unsigned d = (unsigned)c - 65u;
					bad = !(d <= 51u && ((0x0008004500080045ull >> d) & 1ull));
				}
			if (bad) {
				r->code = BLMX_EDNA;
				r->msg = "Invalid char in DNA";
			} else if (r->n) {
				memcpy(r->out, base + o_ids + ck * 8, r->n * 8);
			}
			ck += r->n;
		} else {
			const int64_t* ids = reinterpret_cast<const int64_t*>(base + o_kids) + cm;
			if (r->want_bool)
				for (uint64_t i = 0; i < r->n; i++)
					reinterpret_cast<uint8_t*>(r->out)[i] = ids[i] >= 0 ? 1 : 0;
			else
				memcpy(r->out, ids, r->n * 8);
			cm += r->n;
		}
	}
	ix.number_query.fetch_add(nk_seq + nkm, std::memory_order_relaxed);
	if (tr) {
		hf.cs_rounds.fetch_add(1, std::memory_order_relaxed);
		hf.cs_calls.fetch_add(batch.size(), std::memory_order_relaxed);
		hf.cs_pack_ns.fetch_add(t_packed - t_begin, std::memory_order_relaxed);
		hf.cs_launch_ns.fetch_add(t_launched - t_packed, std::memory_order_relaxed);
		hf.cs_wait_ns.fetch_add(t_done - t_launched, std::memory_order_relaxed);
		hf.cs_hand_ns.fetch_add(now_ns() - t_done, std::memory_order_relaxed);
	}
}

// what one launch may carry (one pinned slot of kSmallSlotBytes)
constexpr uint64_t kCombineChars = kSmallChars, kCombineSeqs = kSmallSeqs, kCombineKmers = kSmallKmers;

// the calling thread's request: answered by a leader's launch, or by its own when nobody leads
void combine_small(Index& ix, HostFront& hf, SmallReq& me)
{
	bool lead;
	{
		std::lock_guard<std::mutex> l(hf.comb_mu);
		lead = !hf.comb_led;
		if (lead) {
			hf.comb_led = true;
		} else {
			hf.comb_wait.push_back(&me);
			hf.comb_nwait.store((int)hf.comb_wait.size(), std::memory_order_relaxed);
		}
	}
	if (!lead) {
		for (int spin = 0;; spin++) {
			const int st = me.state.load(std::memory_order_acquire);
			if (st == 1 || st == 2)
				break;
			if (st == 3) {  // the previous leader has left and made this caller the next one (its request is no longer in the queue)
				lead = true;
				break;
			}
			if (spin < 4000)
				_mm_pause();
			else
				std::this_thread::yield();
		}
	}
	if (lead) {
		// Callers that were answered by the previous launch are on their way back with their next call: a leader that launched at once
		// would carry only the calls that arrived while the previous launch was in flight (half of the callers, and each call would wait
		// two launches).  So it gives the others a few microseconds -- as long as recent launches carried more calls than wait now;
		// a lone caller never waits.
		static const int gather_us = getenv("BLIGHT_COMBINE_GATHER_US") ? atoi(getenv("BLIGHT_COMBINE_GATHER_US")) : 5;
		const int expect = hf.comb_peak.load(std::memory_order_relaxed) - 1;
		if (expect > 0 && gather_us > 0) {
			const auto t0 = std::chrono::steady_clock::now();
			for (int spin = 0; hf.comb_nwait.load(std::memory_order_relaxed) < expect; spin++) {
				_mm_pause();
				if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(gather_us))
					break;
			}
			if (hf.comb_trace)
				hf.cs_gather_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
				                          std::memory_order_relaxed);
		}
		std::vector<SmallReq*> batch;
		batch.push_back(&me);
		uint64_t chars = me.kind == SmallReq::kSequences ? me.nbytes : 0, seqs = me.kind == SmallReq::kSequences ? me.nseq : 0,
		         kmers = me.kind == SmallReq::kKmers ? me.n : 0, kseq = me.kind == SmallReq::kSequences ? me.n : 0;
		{
			std::lock_guard<std::mutex> l(hf.comb_mu);
			size_t keep = 0;
			for (SmallReq* r : hf.comb_wait) {  // everything that fits one slot rides along
				const bool sq = r->kind == SmallReq::kSequences;
				const bool fits = sq ? (chars + r->nbytes <= kCombineChars && seqs + r->nseq <= kCombineSeqs && kseq + r->n <= 48000) : (kmers + r->n <= kCombineKmers);
				if (fits) {
					batch.push_back(r);
					if (sq) {
						chars += r->nbytes;
						seqs += r->nseq;
						kseq += r->n;
					} else {
						kmers += r->n;
					}
				} else {
					hf.comb_wait[keep++] = r;
				}
			}
			hf.comb_wait.resize(keep);
			hf.comb_nwait.store((int)keep, std::memory_order_relaxed);
			const int peak = hf.comb_peak.load(std::memory_order_relaxed);
			hf.comb_peak.store(std::max<int>((int)batch.size(), peak - (peak > 1 && (int)batch.size() < peak ? 1 : 0)), std::memory_order_relaxed);
		}
		int code = BLMX_OK;
		std::string msg;
		bool handed = false;
		auto hand_over = [&] {
			// leadership goes to a caller that is still waiting (it will observe state 3), or nobody leads.  This happens as soon as the
			// launch has been enqueued: rounds overlap (each on its own pinned slot and stream), a call waits for ITS round only.
			if (handed)
				return;
			handed = true;
			std::lock_guard<std::mutex> l(hf.comb_mu);
			if (hf.comb_wait.empty()) {
				hf.comb_led = false;
			} else {
				SmallReq* next = hf.comb_wait.front();
				hf.comb_wait.erase(hf.comb_wait.begin());
				hf.comb_nwait.store((int)hf.comb_wait.size(), std::memory_order_relaxed);
				next->state.store(3, std::memory_order_release);
			}
		};
		try {
			serve_small(ix, hf, batch, hand_over);
		} catch (const StatusError& e) {
			code = e.code;
			msg = e.msg;
		} catch (const std::bad_alloc&) {  // whatever ends the leader's round: the callers that ride along must not wait for ever
			code = BLMX_ENOMEM;
			msg = "out of host memory";
		} catch (const std::exception& e) {
			code = BLMX_EHIP;
			msg = e.what();
		}
		hand_over();  // (a launch that failed before it was enqueued)
		for (SmallReq* r : batch) {
			if (code != BLMX_OK && r->code == BLMX_OK) {
				r->code = code;
				r->msg = msg;
			}
			if (r != &me)
				r->state.store(r->code == BLMX_OK ? 1 : 2, std::memory_order_release);
		}
	}
	if (me.code != BLMX_OK)
		throw StatusError{me.code, me.msg};
}

}  // namespace

// ---- query_sequence_hash for nseq sequences (reference blight.cpp:575-591) -----------------------------------------------------
void query_sequences_host(Index& ix, const char* concat, const uint64_t* offsets, uint64_t nseq, int64_t* ids_out)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	const uint64_t k = ix.prm.k;
	if (nseq == 0)
		return;
	for (uint64_t i = 0; i < nseq; i++)
		if (offsets[i + 1] < offsets[i])
			throw StatusError{BLMX_EINVAL, "offsets must be non-decreasing"};
	const uint64_t a0 = offsets[0], total = offsets[nseq];
	std::vector<uint64_t> koffs(nseq + 1, 0);
	for (uint64_t i = 0; i < nseq; i++) {
		const uint64_t len = offsets[i + 1] - offsets[i];
		koffs[i + 1] = koffs[i] + (len >= k ? len - k + 1 : 0);  // blight.cpp:577-579: shorter than k -> nothing
	}
	const uint64_t nk_all = koffs[nseq];
	if (nk_all == 0)
		return;
	if (!ids_out)
		throw StatusError{BLMX_EINVAL, "ids_out is null"};
	if (nseq == 1 && burst_query_sequence(ix, concat + a0, total - a0, nk_all, ids_out))  // one read: the serving burst (blmx_serve.hip)
		return;
	BLMX_HIP(hipSetDevice(ix.prm.device));
	HostFront& hf = host_front(ix);
	// ---- one launch on a pinned slot
	if (total - a0 <= kSmallChars && nseq <= kSmallSeqs && nk_all <= 48000 && read_query_fusable(ix)) {
		SmallReq rq;  // answered by ONE launch together with whatever other callers are waiting (combine_small)
		rq.kind = SmallReq::kSequences;
		rq.chars = concat;
		rq.offs = offsets;
		rq.koffs = koffs.data();
		rq.a0 = a0;
		rq.nbytes = total - a0;
		rq.nseq = nseq;
		rq.n = nk_all;
		rq.out = ids_out;
		combine_small(ix, hf, rq);
		return;
	}
	// ---- windows in position space: k-mers may start at [A, B), the window holds the characters [A, B + k - 1)
	std::lock_guard<std::mutex> lock(ix.mu);
	const uint64_t W = window_positions(ix);
	auto kmers_before = [&](uint64_t P) -> uint64_t {  // k-mers that start before position P
		if (P <= a0)
			return 0;
		if (P >= total)
			return nk_all;
		const uint64_t s = (uint64_t)(std::upper_bound(offsets, offsets + nseq + 1, P) - offsets) - 1;  // offsets[s] <= P < offsets[s+1]
		return koffs[s] + std::min(P - offsets[s], koffs[s + 1] - koffs[s]);
	};
	auto finish = [&](WindowSlot& w) {
		if (!w.in_flight)
			return;
		BLMX_HIP(hipEventSynchronize(w.done));
		w.in_flight = false;
		if (w.h_flags.as<int>()[0])
			throw StatusError{BLMX_EDNA, "Invalid char in DNA"};
		if (w.narrow)
			widen_ids(*hf.pool, ids_out + w.out_index, w.h_out.as<uint32_t>(), w.nk);
		else
			parallel_copy(*hf.pool, ids_out + w.out_index, w.h_out.p, w.nk * 8);
	};
	struct Drain {  // an exception must not leave copies into the caller's buffers in flight
		HostFront& hf;
		~Drain()
		{
			for (auto& w : hf.win)
				if (w.in_flight) {
					(void)hipEventSynchronize(w.done);
					w.in_flight = false;
				}
		}
	} drain{hf};
	int cur = 0;
	for (uint64_t A = a0; A < total; A += W) {
		const uint64_t B = std::min(A + W, total), nbytes = std::min(B + k - 1, total) - A, npos = B - A;
		const uint64_t nk0 = kmers_before(A), nk = kmers_before(B) - nk0;
		if (nk == 0)
			continue;
		WindowSlot& w = hf.win[cur];
		cur = (cur + 1) % kWindowSlots;
		prepare_window_slot(w);
		finish(w);
		// sequences that overlap the window's characters: their starts break k-mers
		const uint64_t s_lo = (uint64_t)(std::upper_bound(offsets, offsets + nseq + 1, A) - offsets) - 1;
		const uint64_t s_hi = (uint64_t)(std::lower_bound(offsets, offsets + nseq + 1, A + nbytes) - offsets);  // first start >= end of window
		const uint64_t ns = std::min(s_hi, nseq) - s_lo;
		w.h_in.ensure(nbytes + 64);
		parallel_copy(*hf.pool, w.h_in.p, concat + A, nbytes);
		w.h_aux.ensure((ns + 1) * 8);
		memcpy(w.h_aux.p, offsets + s_lo, (ns + 1) * 8);
		grow(w.d_in, nbytes + 64, w.s);
		grow(w.d_aux, (ns + 1) * 8, w.s);
		BLMX_HIP(hipMemcpyAsync(w.d_in.p, w.h_in.p, nbytes, hipMemcpyHostToDevice, w.s));
		BLMX_HIP(hipMemcpyAsync(w.d_aux.p, w.h_aux.p, (ns + 1) * 8, hipMemcpyHostToDevice, w.s));
		BLMX_HIP(hipMemsetAsync(w.d_flags.p, 0, 16, w.s));
		front_valid_concat(ix, w.fs, w.d_aux.as<uint64_t>(), ns + 1, A, npos, nbytes, w.s);
		query_window(ix, w, nbytes, npos, nk, true, nullptr);
		w.narrow = ids_fit_32(ix);
		if (w.narrow) {
			grow(w.d_aux2, nk * 4, w.s);
			launch_narrow_ids(w.d_ids.as<int64_t>(), nk, w.d_aux2.as<uint32_t>(), w.s);
		}
		const size_t ob = w.narrow ? 4 : 8;
		w.h_out.ensure(nk * ob);
		BLMX_HIP(hipMemcpyAsync(w.h_out.p, w.narrow ? w.d_aux2.p : w.d_ids.p, nk * ob, hipMemcpyDeviceToHost, w.s));
		BLMX_HIP(hipMemcpyAsync(w.h_flags.p, w.d_flags.p, 4, hipMemcpyDeviceToHost, w.s));
		BLMX_HIP(hipEventRecord(w.done, w.s));
		w.in_flight = true;
		w.nk = nk;
		w.out_index = nk0;
	}
	for (int i = 0; i < kWindowSlots; i++) {  // oldest first
		finish(hf.win[cur]);
		cur = (cur + 1) % kWindowSlots;
	}
	ix.number_query.fetch_add(nk_all, std::memory_order_relaxed);
}

// The same for sequences that are already resident in HBM (concatenated characters + nseq + 1 offsets, offsets[0] = 0):
// asynchronous on the caller's stream; nk = sum of max(0, len - k + 1) as the caller computed it.
void query_sequences_device(Index& ix, const uint8_t* d_chars, const uint64_t* d_offs, uint64_t nseq, uint64_t total, uint64_t nk, int64_t* d_ids,
                            int* d_err, hipStream_t s)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	if (nk == 0 || nseq == 0)
		return;
	if (total >= 0xffffffc0ull)
		throw StatusError{BLMX_ELIMIT, "a device-resident batch holds fewer than 2^32 characters"};
	BLMX_HIP(hipSetDevice(ix.prm.device));
	HostFront& hf = host_front(ix);
	std::lock_guard<std::mutex> lock(hf.dev_mu);
	if (!hf.dev_done)
		BLMX_HIP(hipEventCreateWithFlags(&hf.dev_done, hipEventDisableTiming));
	if (hf.dev_used && hf.dev_last != s)
		BLMX_HIP(hipStreamWaitEvent(s, hf.dev_done, 0));
	if (!d_err) {
		if (!hf.dev_err.p)
			BLMX_HIP(hf.dev_err.alloc(64, true, s));
		d_err = hf.dev_err.as<int>();
	}
	ix.prof_reset();
	ix.prof_mark(0, s);
	front_valid_concat(ix, hf.dev_fs, d_offs, nseq + 1, 0, total, total, s);
	const bool binned = !read_query_fusable(ix) || ix.tune_reads_mode == 2 || (ix.tune_reads_mode == 0 && query_batch_worth_binning(ix, nk));
	if (binned) {
		grow(hf.dev_kmers, nk * 8 * (size_t)ix.KW, s);
		grow(hf.dev_minis, nk * 4, s);
		launch_read_front(ix, &hf.dev_fs, d_chars, total, total, nullptr, nullptr, 0, hf.dev_kmers.p, hf.dev_minis.as<uint32_t>(), d_err, s);
		ix.prof_mark(1, s);
		launch_query_ids_with_minis(ix, hf.dev_kmers.p, hf.dev_minis.as<uint32_t>(), nk, d_ids, s);
	} else {
		ix.prof_mark(1, s);
		ix.prof_mark(4, s);
		launch_read_query(ix, &hf.dev_fs, d_chars, total, total, nullptr, nullptr, 0, d_ids, nullptr, d_err, s);
		ix.prof_mark(5, s);
	}
	BLMX_HIP(hipEventRecord(hf.dev_done, s));
	hf.dev_last = s;
	hf.dev_used = true;
	ix.number_query.fetch_add(nk, std::memory_order_relaxed);
}

// Sequences resident in HBM against a SHARDED index: the read front end of this rank turns them into canonical k-mers, the
// exchange (blmx_comm.hip) sends every k-mer to the owner of its MPHF group and brings the IDs back.  Collective.
void query_sequences_sharded_device(Index& ix, Comm& c, const uint8_t* d_chars, const uint64_t* d_offs, uint64_t nseq, uint64_t total, uint64_t nk,
                                    int64_t* d_ids, int* d_err, hipStream_t s)
{
	BLMX_HIP(hipSetDevice(ix.prm.device));
	// a rank whose own batch cannot be taken still joins the collective (with no k-mers, or -- index not built -- with the failure
	// flag of query_kmers_sharded), so that the others do not wait for it; it reports its error afterwards
	if (!ix.built || total >= 0xffffffc0ull) {
		query_kmers_sharded(ix, c, nullptr, nullptr, 0, d_ids, s);
		throw StatusError{BLMX_ELIMIT, "a device-resident batch holds fewer than 2^32 characters"};
	}
	HostFront& hf = host_front(ix);
	std::lock_guard<std::mutex> lock(hf.dev_mu);
	if (!hf.dev_done)
		BLMX_HIP(hipEventCreateWithFlags(&hf.dev_done, hipEventDisableTiming));
	if (hf.dev_used && hf.dev_last != s)
		BLMX_HIP(hipStreamWaitEvent(s, hf.dev_done, 0));
	if (!d_err) {
		if (!hf.dev_err.p)
			BLMX_HIP(hf.dev_err.alloc(64, true, s));
		d_err = hf.dev_err.as<int>();
	}
	grow(hf.dev_kmers, nk * 8 * (size_t)ix.KW + 16, s);
	grow(hf.dev_minis, nk * 4 + 16, s);
	if (nk && nseq) {
		front_valid_concat(ix, hf.dev_fs, d_offs, nseq + 1, 0, total, total, s);
		launch_read_front(ix, &hf.dev_fs, d_chars, total, total, nullptr, nullptr, 0, hf.dev_kmers.p, hf.dev_minis.as<uint32_t>(), d_err, s);
	}
	query_kmers_sharded(ix, c, hf.dev_kmers.p, hf.dev_minis.as<uint32_t>(), nk, d_ids, s);  // every rank takes part, also with an empty batch
	BLMX_HIP(hipEventRecord(hf.dev_done, s));
	hf.dev_last = s;
	hf.dev_used = true;
}

// host characters in, host IDs out (staged through the index's device buffers)
void query_sequences_sharded_host(Index& ix, Comm& c, const char* concat, const uint64_t* offsets, uint64_t nseq, int64_t* ids_out)
{
	BLMX_HIP(hipSetDevice(ix.prm.device));
	const uint64_t total = nseq ? offsets[nseq] : 0;
	if (nseq && offsets[0] != 0) {  // (collective: take part with an empty batch, then report)
		query_kmers_sharded(ix, c, nullptr, nullptr, 0, nullptr, ix.stream);
		throw StatusError{BLMX_EINVAL, "offsets[0] must be 0"};
	}
	const uint64_t k = ix.prm.k;
	uint64_t nk = 0;
	for (uint64_t i = 0; i < nseq; i++) {
		const uint64_t len = offsets[i + 1] - offsets[i];
		nk += len >= k ? len - k + 1 : 0;
	}
	std::lock_guard<std::mutex> lock(ix.mu);
	hipStream_t s = ix.stream;
	grow(ix.q_in, total + 64, s);
	grow(ix.q_aux, (nseq + 1) * 8 + 16, s);
	grow(ix.q_out, nk * 8 + 16, s);
	DevBuf d_err;
	BLMX_HIP(d_err.alloc(64, true, s));
	if (total)
		BLMX_HIP(hipMemcpyAsync(ix.q_in.p, concat, total, hipMemcpyHostToDevice, s));
	if (nseq)
		BLMX_HIP(hipMemcpyAsync(ix.q_aux.p, offsets, (nseq + 1) * 8, hipMemcpyHostToDevice, s));
	query_sequences_sharded_device(ix, c, ix.q_in.as<uint8_t>(), ix.q_aux.as<uint64_t>(), nseq, total, nk, ix.q_out.as<int64_t>(), d_err.as<int>(), s);
	int err = 0;
	BLMX_HIP(hipMemcpyAsync(&err, d_err.p, sizeof err, hipMemcpyDeviceToHost, s));
	if (nk)
		BLMX_HIP(hipMemcpyAsync(ids_out, ix.q_out.p, nk * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	if (err)
		throw StatusError{BLMX_EDNA, "Invalid char in DNA"};
}

// ---- query_kmer_hash / query_kmer_bool / query_get_hash / query_get_pos_unitig for a host batch (blight.cpp:537-550, 686-742) ----
void query_kmers_host(Index& ix, const void* kmers, const uint32_t* minis, uint64_t nq, void* out, QueryMode mode)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	if (nq == 0)
		return;
	if (!minis && (mode == kQueryIds || mode == kQueryBool) && burst_query_kmers(ix, kmers, nq, out, mode == kQueryBool))  // blmx_serve.hip
		return;
	BLMX_HIP(hipSetDevice(ix.prm.device));
	HostFront& hf = host_front(ix);
	const size_t kb = (size_t)ix.KW * 8;
	const size_t ob = mode == kQueryIds ? 8 : (mode == kQueryPos ? 4 : 1);
	if (nq <= kSmallKmers && !minis && (mode == kQueryIds || mode == kQueryBool) && ix.prm.bits_saved <= 8) {
		SmallReq rq;  // scalar calls of the facade land here: answered by ONE launch together with whatever other callers are waiting
		rq.kind = SmallReq::kKmers;
		rq.kmers = kmers;
		rq.want_bool = mode == kQueryBool;
		rq.n = nq;
		rq.out = out;
		combine_small(ix, hf, rq);
		return;
	}
	if (nq <= kSmallKmers) {  // caller-supplied minimizers / positions: one launch of its own on a pinned slot
		SmallLease lease(ix, hf);
		SmallSlot& sl = *lease.sl;
		char* base = sl.h.as<char>();
		const size_t o_minis = align_up(nq * kb, 64), o_out = o_minis + align_up(nq * 4, 64);
		memcpy(base, kmers, nq * kb);
		if (minis)
			memcpy(base + o_minis, minis, nq * 4);
		launch_query(ix, base, minis ? reinterpret_cast<const uint32_t*>(base + o_minis) : nullptr, nq, base + o_out,
		             (QueryMode)(mode | (minis ? kQueryNoFilter : 0)), sl.s);
		BLMX_HIP(hipStreamSynchronize(sl.s));
		memcpy(out, base + o_out, nq * ob);
		ix.number_query.fetch_add(nq, std::memory_order_relaxed);  // blight.cpp:717-718
		return;
	}
	// chunks through the two window slots: copy-in (host threads) | H2D | kernels | D2H | copy-out overlap
	std::lock_guard<std::mutex> lock(ix.mu);
	const uint64_t C = std::max<uint64_t>(1u << 16, window_positions(ix) / 4);  // k-mers per chunk
	auto finish = [&](WindowSlot& w) {
		if (!w.in_flight)
			return;
		BLMX_HIP(hipEventSynchronize(w.done));
		w.in_flight = false;
		if (w.narrow)
			widen_ids(*hf.pool, (int64_t*)out + w.out_index, w.h_out.as<uint32_t>(), w.nk);
		else
			parallel_copy(*hf.pool, (char*)out + w.out_index * w.out_elem, w.h_out.p, w.nk * w.out_elem);
	};
	struct Drain {
		HostFront& hf;
		~Drain()
		{
			for (auto& w : hf.win)
				if (w.in_flight) {
					(void)hipEventSynchronize(w.done);
					w.in_flight = false;
				}
		}
	} drain{hf};
	int cur = 0;
	Trace tr;
	for (uint64_t a = 0; a < nq; a += C) {
		const uint64_t n = std::min(C, nq - a);
		WindowSlot& w = hf.win[cur];
		cur = (cur + 1) % kWindowSlots;
		prepare_window_slot(w);
		finish(w);
		tr.lap("wait+copy out", a / C);
		w.h_in.ensure(n * kb);
		parallel_copy(*hf.pool, w.h_in.p, (const char*)kmers + a * kb, n * kb);
		tr.lap("copy in", a / C);
		grow(w.d_kmers, n * kb, w.s);
		grow(w.d_ids, n * 8, w.s);
		BLMX_HIP(hipMemcpyAsync(w.d_kmers.p, w.h_in.p, n * kb, hipMemcpyHostToDevice, w.s));
		const uint32_t* d_minis = nullptr;
		if (minis) {
			w.h_aux.ensure(n * 4);
			memcpy(w.h_aux.p, minis + a, n * 4);
			grow(w.d_minis, n * 4, w.s);
			BLMX_HIP(hipMemcpyAsync(w.d_minis.p, w.h_aux.p, n * 4, hipMemcpyHostToDevice, w.s));
			d_minis = w.d_minis.as<uint32_t>();
		}
		if (mode == kQueryIds && !d_minis)
			launch_query_ids_auto(ix, w.d_kmers.p, n, w.d_ids.as<int64_t>(), w.s);
		else
			launch_query(ix, w.d_kmers.p, d_minis, n, w.d_ids.p, (QueryMode)(mode | (minis ? kQueryNoFilter : 0)), w.s);
		w.narrow = mode == kQueryIds && ids_fit_32(ix);
		if (w.narrow) {
			grow(w.d_aux2, n * 4, w.s);
			launch_narrow_ids(w.d_ids.as<int64_t>(), n, w.d_aux2.as<uint32_t>(), w.s);
		}
		const size_t wire = w.narrow ? 4 : ob;
		w.h_out.ensure(n * wire);
		BLMX_HIP(hipMemcpyAsync(w.h_out.p, w.narrow ? w.d_aux2.p : w.d_ids.p, n * wire, hipMemcpyDeviceToHost, w.s));
		BLMX_HIP(hipEventRecord(w.done, w.s));
		w.in_flight = true;
		w.nk = n;
		w.out_index = a;
		w.out_elem = ob;
		tr.lap("enqueue", a / C);
	}
	for (int i = 0; i < kWindowSlots; i++) {
		finish(hf.win[cur]);
		cur = (cur + 1) % kWindowSlots;
	}
	ix.number_query.fetch_add(nq, std::memory_order_relaxed);
}

// ---- file_query (reference blight.cpp:746-799) and the payload applications' file pass ------------------------------------------
namespace {

// raw bytes of a plain or gzip file, window after window, each cut behind its last newline
class RawFile {
public:
	explicit RawFile(const char* path)
	{
		fd_ = open(path, O_RDONLY);
		if (fd_ < 0)
			return;
		unsigned char magic[2] = {0, 0};
		const ssize_t got = pread(fd_, magic, 2, 0);
		struct stat st;
		const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
		if (!gz && fstat(fd_, &st) == 0 && S_ISREG(st.st_mode)) {
			size_ = (uint64_t)st.st_size;
			ok_ = true;
			return;
		}
		close(fd_);
		fd_ = -1;
		gz_ = gzopen(path, "rb");
		if (gz_) {
			gzbuffer(gz_, 1 << 20);
			ok_ = true;
		}
	}
	~RawFile() { shut(); }
	void shut()
	{
		if (fd_ >= 0)
			close(fd_);
		if (gz_)
			gzclose(gz_);
		fd_ = -1;
		gz_ = nullptr;
	}
	RawFile(const RawFile&) = delete;
	RawFile& operator=(const RawFile&) = delete;
	bool ok() const { return ok_; }
	uint64_t size() const { return fd_ >= 0 ? size_ : 0; }  // bytes of a plain file (0: gzip, unknown)

	// Fills `pin` with the next window (about `want` bytes, more when a line is longer) and returns its length; 0 = end of file.
	// Every window but the last ends behind a newline.
	uint64_t next(Pinned& pin, uint64_t want, WorkerPool& pool)
	{
		uint64_t have = 0;
		if (!tail_.empty()) {  // gzip: the bytes behind the previous window's last newline
			pin.ensure(tail_.size() + want + 64);
			memcpy(pin.p, tail_.data(), tail_.size());
			have = tail_.size();
			tail_.clear();
		}
		for (;;) {
			pin.ensure(have + want + 64, have);
			const uint64_t got = read_more(pin.as<char>() + have, want, pool);
			have += got;
			const bool eof = got < want;
			if (have == 0)
				return 0;
			if (eof)
				return have;
			const char* nl = (const char*)memrchr(pin.as<char>(), '\n', have);
			if (nl) {
				const uint64_t wlen = (uint64_t)(nl - pin.as<char>()) + 1;
				if (fd_ >= 0)
					pos_ -= have - wlen;  // plain file: the rest is read again with the next window
				else
					tail_.assign(pin.as<char>() + wlen, pin.as<char>() + have);
				return wlen;
			}
			want = std::max<uint64_t>(want, have);  // one line longer than the window: double it
		}
	}

private:
	uint64_t read_more(char* dst, uint64_t want, WorkerPool& pool)
	{
		if (fd_ >= 0) {
			constexpr uint64_t kReadPiece = 2u << 20;  // a 32-MiB window keeps 16 workers busy
			const uint64_t n = pos_ < size_ ? std::min(want, size_ - pos_) : 0;
			const size_t pieces = (size_t)((n + kReadPiece - 1) / kReadPiece);
			std::atomic<bool> bad{false};
			const uint64_t pos = pos_;
			const int fd = fd_;
			// (pread by 16 workers in 2-MiB pieces: 44 GB/s from tmpfs / the page cache with the H2D copies running; copying out of a
			// mapping is faster per byte, but tearing its page tables down costs as much again, profiles/r03_readbench.txt)
			pool.run(pieces, [&](size_t i) {
				uint64_t a = (uint64_t)i * kReadPiece;
				const uint64_t z = std::min(n, a + kReadPiece);
				while (a < z) {
					const ssize_t r = pread(fd, dst + a, z - a, (off_t)(pos + a));
					if (r <= 0) {
						bad = true;
						return;
					}
					a += (uint64_t)r;
				}
			});
			if (bad)
				throw StatusError{BLMX_EIO, "read error"};
			pos_ += n;
			return n;
		}
		uint64_t n = 0;
		while (n < want) {
			const int r = gzread(gz_, dst + n, (unsigned)std::min<uint64_t>(want - n, 1u << 30));
			if (r < 0)
				throw StatusError{BLMX_EIO, "gzip read error"};
			if (r == 0)
				break;
			n += (uint64_t)r;
		}
		return n;
	}
	int fd_ = -1;
	gzFile gz_ = nullptr;
	uint64_t size_ = 0, pos_ = 0;
	std::vector<char> tail_;
	bool ok_ = false;
};

// mode 0: 2-line FASTA records (file_query); mode 1: every line that starts with A/C/G/T (the payload snippets)
void stream_file(Index& ix, const char* path, int mode, Payload* payload, uint32_t color, uint64_t& good, uint64_t& erroneous)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	RawFile rf(path);
	if (!rf.ok())
		throw StatusError{BLMX_EIO, "Problem with files opening"};
	BLMX_HIP(hipSetDevice(ix.prm.device));
	HostFront& hf = host_front(ix);
	for (auto& w : hf.win)
		prepare_window_slot(w);
	hipStream_t s0 = hf.win[0].s;
	if (!hf.d_counts.p)
		BLMX_HIP(hf.d_counts.alloc(16 * kCountStripes, false, s0));
	if (!hf.d_carry.p)
		BLMX_HIP(hf.d_carry.alloc(32, false, s0));
	BLMX_HIP(hipMemsetAsync(hf.d_counts.p, 0, 16 * kCountStripes, s0));
	BLMX_HIP(hipMemsetAsync(hf.d_carry.p, 0, 32, s0));
	BLMX_HIP(hipStreamSynchronize(s0));
	auto finish = [&](WindowSlot& w) {
		if (!w.in_flight)
			return;
		BLMX_HIP(hipEventSynchronize(w.done));
		w.in_flight = false;
		if (w.h_flags.as<int>()[4])
			throw StatusError{BLMX_EDNA, "Invalid char in DNA"};
	};
	struct Drain {
		HostFront& hf;
		~Drain()
		{
			for (auto& w : hf.win)
				if (w.in_flight) {
					(void)hipEventSynchronize(w.done);
					w.in_flight = false;
				}
		}
	} drain{hf};
	const uint64_t W = window_positions(ix);
	uint64_t queries = 0;
	unsigned long long* d_counts = hf.d_counts.as<unsigned long long>();
	Trace tr;
	for (uint64_t i = 0;; i++) {
		WindowSlot& w = hf.win[i % kWindowSlots];
		finish(w);
		tr.lap("wait slot", i);
		const uint64_t nbytes = rf.next(w.h_in, W, *hf.pool);
		tr.lap("read", i);
		if (nbytes == 0)
			break;
		if (nbytes >= 0xffffffc0ull)
			throw StatusError{BLMX_ELIMIT, "a line of 4 GiB or more"};
		grow(w.d_in, nbytes + 64, w.s);
		BLMX_HIP(hipMemcpyAsync(w.d_in.p, w.h_in.p, nbytes, hipMemcpyHostToDevice, w.s));
		BLMX_HIP(hipMemsetAsync(w.d_flags.p, 0, 16, w.s));
		uint64_t* carry = hf.d_carry.as<uint64_t>();
		front_valid_raw(ix, w.fs, w.d_in.as<uint8_t>(), nbytes, mode, carry + 2 * (i & 1), carry + 2 * ((i + 1) & 1), nullptr, w.s);
		const uint64_t nwords = (nbytes + 63) / 64;
		BLMX_HIP(hipMemcpyAsync(w.h_flags.p, w.fs.vpre.as<uint32_t>() + nwords, 4, hipMemcpyDeviceToHost, w.s));
		tr.lap("enqueue prep", i);
		BLMX_HIP(hipStreamSynchronize(w.s));  // the number of k-mers sizes the rest of the window's work
		tr.lap("sync prep", i);
		const uint64_t nk = w.h_flags.as<uint32_t>()[0];
		if (nk) {
			query_window(ix, w, nbytes, nbytes, nk, payload != nullptr, d_counts);
			if (payload)
				payload_add_ids(*payload, w.d_ids.as<int64_t>(), nk, color, w.s);
		}
		BLMX_HIP(hipMemcpyAsync(w.h_flags.as<int>() + 4, w.d_flags.p, 4, hipMemcpyDeviceToHost, w.s));
		BLMX_HIP(hipEventRecord(w.done, w.s));
		w.in_flight = true;
		queries += nk;
		tr.lap("enqueue query", i);
	}
	for (auto& w : hf.win)
		finish(w);
	uint64_t h_counts[2 * kCountStripes];
	BLMX_HIP(hipMemcpy(h_counts, hf.d_counts.p, sizeof h_counts, hipMemcpyDeviceToHost));
	good = erroneous = 0;
	for (int i = 0; i < kCountStripes; i++) {
		good += h_counts[2 * i];
		erroneous += h_counts[2 * i + 1];
	}
	ix.number_query.fetch_add(queries, std::memory_order_relaxed);
}

}  // namespace

// construct_index(file) (reference blight.cpp:108-125, reader :212-229) with the parse on the device: the RAW bytes of the file
// stream through pinned windows into HBM like file_query's, and newlines, the record rule and the sequences themselves are found
// there (front_seq_lines): the host only moves bytes.  Out: the sequences' characters back to back, nseq + 1 offsets, both in HBM.
void load_fasta_raw_to_device(Index& ix, const char* path, DevBuf& d_text, DevBuf& d_offs, uint64_t& nseq, uint64_t& total)
{
	// d_text holds the 2-bit text (one byte per four nucleotides: an ASCII copy of a human-scale input would cost more to allocate
	// -- 4.4 GB of fresh HBM: 0.25 s -- than to parse)
	Trace tr;
	RawFile rf(path);
	if (!rf.ok())
		throw StatusError{BLMX_EIO, "Problem with files opening"};  // blight.cpp:188-189
	BLMX_HIP(hipSetDevice(ix.prm.device));
	HostFront& hf = host_front(ix);
	std::lock_guard<std::mutex> lock(ix.mu);
	for (auto& w : hf.win)
		prepare_window_slot(w);
	// every window's kernels run on ONE stream (the appends are ordered) and the H2D copies on a second one, so that a window's copy
	// overlaps the kernels of the window before it; the pinned slots rotate so that reading the next window overlaps both
	hipStream_t s = hf.win[0].s, s_copy = hf.win[1].s;
	DevBuf d_state;  // [0..3] line carry (two generations), [4] sequences, [5] characters, [6] short flag, [7] invalid-character flag
	BLMX_HIP(d_state.alloc(64, true, s));
	uint64_t* carry = d_state.as<uint64_t>();
	uint64_t* d_tot = carry + 4;
	int* d_short = reinterpret_cast<int*>(carry + 6);
	int* d_dna = reinterpret_cast<int*>(carry + 7);
	const uint64_t W = window_positions(ix), k = ix.prm.k;
	const uint64_t fsize = rf.size();  // 0 = unknown (gzip)
	uint64_t cap_text = 0, cap_offs = 0, seen = 0;  // `seen` raw bytes bound what has been appended
	auto reserve = [&](uint64_t more) {
		// bytes of packed text (four nucleotides each, + the words the build reads behind the end), offsets (a record holds k + 1 bytes or more)
		const uint64_t need_text = (seen + more) / 4 + 64, need_offs = (seen + more) / (k + 1) + 4;
		if (need_text > cap_text) {
			DevBuf bigger;
			const uint64_t nb = (std::max<uint64_t>(need_text, fsize ? fsize / 4 + 64 : 2 * cap_text) + 7) & ~7ull;
			BLMX_HIP(bigger.alloc(nb, true, s));  // zeroed: the windows OR their words in
			if (cap_text) {
				BLMX_HIP(hipMemcpyAsync(bigger.p, d_text.p, cap_text, hipMemcpyDeviceToDevice, s));
				BLMX_HIP(hipStreamSynchronize(s));
			}
			bigger.swap(d_text);
			cap_text = nb;
		}
		if (need_offs > cap_offs) {
			DevBuf bigger;
			const uint64_t no = std::max<uint64_t>(need_offs, fsize ? fsize / (k + 1) + 4 : 2 * cap_offs);
			BLMX_HIP(bigger.alloc(no * 8, false, s));
			if (cap_offs) {
				BLMX_HIP(hipMemcpyAsync(bigger.p, d_offs.p, cap_offs * 8, hipMemcpyDeviceToDevice, s));
				BLMX_HIP(hipStreamSynchronize(s));
			} else {
				BLMX_HIP(hipMemsetAsync(bigger.p, 0, 8, s));  // offs[0] = 0
			}
			bigger.swap(d_offs);
			cap_offs = no;
		}
	};
	tr.lap("open + slots", 0);
	reserve(0);
	tr.lap("allocate text+offsets", 0);
	hipEvent_t copied[2 * kWindowSlots] = {};  // [slot]: the slot's H2D copy has left the pinned buffer; [slots + slot]: its kernels are done
	hipEvent_t* parsed = copied + kWindowSlots;
	struct EvGuard {
		hipEvent_t* e;
		hipStream_t s, s_copy;
		~EvGuard()
		{
			(void)hipStreamSynchronize(s_copy);  // no copy out of a pinned slot may still be in flight when the caller unwinds
			(void)hipStreamSynchronize(s);
			for (int i = 0; i < 2 * kWindowSlots; i++)
				if (e[i])
					(void)hipEventDestroy(e[i]);
		}
	} guard{copied, s, s_copy};
	for (int i = 0; i < 2 * kWindowSlots; i++)
		BLMX_HIP(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
	bool used[kWindowSlots] = {};
	FrontScratch& fs = hf.win[0].fs;
	for (uint64_t i = 0;; i++) {
		const int slot = (int)(i % kWindowSlots);
		WindowSlot& w = hf.win[slot];
		if (used[slot])
			BLMX_HIP(hipEventSynchronize(copied[slot]));  // the pinned buffer is free again once its copy has left
		tr.lap("wait slot", i);
		const uint64_t nbytes = rf.next(w.h_in, W, *hf.pool);
		tr.lap("read", i);
		if (nbytes == 0)
			break;
		if (nbytes >= 0xffffffc0ull)
			throw StatusError{BLMX_ELIMIT, "a line of 4 GiB or more"};
		reserve(nbytes);
		seen += nbytes;
		if (nbytes + 64 > w.d_in.bytes && used[slot])
			BLMX_HIP(hipEventSynchronize(parsed[slot]));  // (the slot's device buffer is about to be replaced)
		grow(w.d_in, nbytes + 64, s_copy);  // (on the COPY's stream: a fill of the fresh block -- BLIGHT_POISON -- must not land on top of the copy)
		if (used[slot])
			BLMX_HIP(hipStreamWaitEvent(s_copy, parsed[slot], 0));  // the kernels of the slot's previous window have read its device buffer
		BLMX_HIP(hipMemcpyAsync(w.d_in.p, w.h_in.p, nbytes, hipMemcpyHostToDevice, s_copy));
		BLMX_HIP(hipEventRecord(copied[slot], s_copy));
		used[slot] = true;
		BLMX_HIP(hipStreamWaitEvent(s, copied[slot], 0));
		front_seq_lines(ix, fs, w.d_in.as<uint8_t>(), nbytes, carry + 2 * (i & 1), carry + 2 * ((i + 1) & 1), d_short, d_tot, nullptr, d_text.as<uint64_t>(), d_dna,
		                d_offs.as<uint64_t>(), s);
		BLMX_HIP(hipEventRecord(parsed[slot], s));
		tr.lap("enqueue", i);
	}
	uint64_t h_state[8] = {};
	BLMX_HIP(hipMemcpyAsync(h_state, d_state.p, 64, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	tr.lap("wait for the device", 0);
	if (*reinterpret_cast<const int*>(&h_state[7]))
		throw StatusError{BLMX_EDNA, "Invalid char in DNA"};
	if (*reinterpret_cast<const int*>(&h_state[6]))
		throw StatusError{BLMX_ESHORT, "a sequence is shorter than k"};
	nseq = h_state[4];
	total = h_state[5];
	rf.shut();
	tr.lap("close", 0);
}

// file_query: returns good / erroneous k-mer counts (the reference prints them: "Good kmer", "Erroneous kmers").
void query_file(Index& ix, const char* path, uint64_t& good, uint64_t& erroneous) { stream_file(ix, path, 0, nullptr, 0, good, erroneous); }

void payload_add_file(Payload& pl, const char* path, uint32_t color, uint64_t& found, uint64_t& absent)
{
	if (pl.kind == 1 && color >= pl.ncolors)
		throw StatusError{BLMX_EINVAL, "colour out of range"};
	stream_file(*pl.ix, path, 1, &pl, color, found, absent);
}

}  // namespace blmx
