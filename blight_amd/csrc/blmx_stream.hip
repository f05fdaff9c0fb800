// blmx_stream.hip -- streaming front end of the query path: file_query (reference blight.cpp:746-799).
//
// The reference reads 512 records at a time under an OpenMP critical section and answers them k-mer by k-mer.  Here the
// host only parses records (chunked gzread + memchr, plain or gzip), packs them into pinned batches and hands them to
// the GPU; two batches are in flight so that parsing, H2D, the kernels (rolling canonical k-mers -> query pipeline ->
// on-device count of found / not found) and the 16-byte D2H of the counters overlap.  Nothing but the two counters
// comes back over PCIe.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <exception>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>

#include "blmx_internal.h"

namespace blmx {

// ---- 2-line FASTA records, reference blight.cpp:212-229 / 760-775 ----------------------------------------------------
// Read a header line; if it is empty, swallow one more line and start over; read the sequence line; skip it if empty.
// Lines are returned as views into an internal buffer that is refilled in 8 MiB chunks (gzread reads plain files too).
class RecordReader {
public:
	// A plain file is mapped and scanned in place (no read() copy through zlib's buffers: the parse is the slowest stage of
	// ingest and file_query); a gzip file goes through gzread in 8 MiB chunks.
	explicit RecordReader(const char* path)
	{
		const int fd = open(path, O_RDONLY);
		if (fd < 0)
			return;
		unsigned char magic[2] = {0, 0};
		const ssize_t got = pread(fd, magic, 2, 0);
		struct stat st;
		const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
		if (!gz && fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
			if (st.st_size == 0) {
				opened_ = true;
				eof_ = true;
			} else {
				void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
				if (m != MAP_FAILED) {
					(void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
					map_ = static_cast<const char*>(m);
					map_bytes_ = (size_t)st.st_size;
					data_ = map_;
					end_ = map_bytes_;
					eof_ = true;  // everything is already "read"
					opened_ = true;
				}
			}
		}
		close(fd);
		if (!opened_) {
			f_ = gzopen(path, "rb");
			if (f_) {
				gzbuffer(f_, 1 << 20);
				buf_.resize(kChunk);
				data_ = buf_.data();
				opened_ = true;
			}
		}
	}
	~RecordReader()
	{
		if (f_)
			gzclose(f_);
		if (map_)
			munmap(const_cast<char*>(map_), map_bytes_);
	}
	RecordReader(const RecordReader&) = delete;
	RecordReader& operator=(const RecordReader&) = delete;
	bool ok() const { return opened_; }
	bool mapped() const { return map_ != nullptr; }

	// next sequence line (header already skipped); false at end of file
	bool next_sequence(const char*& p, size_t& len)
	{
		for (;;) {
			if (!next_line(p, len))
				return false;
			if (len == 0) {  // empty header: swallow one more line
				if (!next_line(p, len))
					return false;
				continue;
			}
			if (!next_line(p, len))
				return false;
			if (len == 0)
				continue;
			return true;
		}
	}

	// Bulk fast path for mapped plain files: parses the next window of up to `cap` bytes with several host threads and
	// appends every sequence of at least `minlen` characters to dst (concatenated) / offs (cumulative ends, continuing from
	// offs.back()) / koffs (cumulative k-mer counts len - k + 1, when given).  The reader's record rule (header line,
	// sequence line; an empty header swallows a line, an empty sequence is skipped) depends on everything before a line, so
	// the threads only take over when the window is REGULAR -- no empty line in it -- where the rule reduces to "odd lines
	// are sequences": pass A counts lines and empties per sub-range, a prefix over the sub-ranges gives every thread its line
	// parity and its output offsets, pass B copies.  Returns the number of sequences appended; 0 means "not applicable here"
	// (gzip input, an irregular window, nothing left) and the caller continues with next_sequence().
	size_t bulk(char* dst, size_t cap, std::vector<uint64_t>& offs, std::vector<uint64_t>* koffs, uint64_t k, size_t minlen)
	{
		if (!map_ || beg_ >= end_ || cap < (1u << 20) || beg_ < no_bulk_before_)
			return 0;
		static const unsigned nthreads = [] {
			const char* e = getenv("BLIGHT_PARSE_THREADS");
			const unsigned hw = std::thread::hardware_concurrency();
			const unsigned v = e ? (unsigned)atoi(e) : (hw && hw < 16 ? hw : 16u);
			return v < 1 ? 1u : (v > 64 ? 64u : v);
		}();
		size_t wend = beg_ + cap < end_ ? beg_ + cap : end_;
		if (wend < end_) {  // end the window after a newline
			const char* nl = (const char*)memrchr(data_ + beg_, '\n', wend - beg_);
			if (!nl)
				return 0;  // one line longer than the window
			wend = (size_t)(nl - data_) + 1;
		}
		struct Part {
			size_t a = 0, z = 0;  // lines STARTING in [a, z)
			uint64_t lines = 0;
			bool empty = false;
			uint64_t nseq[2] = {0, 0}, bytes[2] = {0, 0}, kmers[2] = {0, 0};  // by local line parity
		};
		std::vector<Part> part(nthreads);
		const size_t span = wend - beg_;
		for (unsigned t = 0; t < nthreads; t++) {
			size_t a = beg_ + span * t / nthreads;
			if (t > 0) {  // move to the start of the next line
				const char* nl = a < wend ? (const char*)memchr(data_ + a - 1, '\n', wend - (a - 1)) : nullptr;
				a = nl ? (size_t)(nl - data_) + 1 : wend;
			}
			part[t].a = a;
			if (t > 0)
				part[t - 1].z = a;
		}
		part[nthreads - 1].z = wend;
		const char* base = data_;
		auto scan = [base, wend, k, minlen](Part& pt) {
			size_t p = pt.a;
			while (p < pt.z) {
				const char* nl = (const char*)memchr(base + p, '\n', wend - p);
				const size_t e = nl ? (size_t)(nl - base) : wend;  // last line of the file may lack its newline
				const size_t len = e - p;
				const unsigned par = (unsigned)(pt.lines & 1);
				if (len == 0)
					pt.empty = true;
				if (len >= minlen && len >= k) {
					pt.nseq[par]++;
					pt.bytes[par] += len;
					pt.kmers[par] += len - k + 1;
				}
				pt.lines++;
				p = e + 1;
			}
		};
		{
			std::vector<std::thread> th;
			for (unsigned t = 1; t < nthreads; t++)
				th.emplace_back(scan, std::ref(part[t]));
			scan(part[0]);
			for (auto& x : th)
				x.join();
		}
		uint64_t lines = 0;
		for (const Part& pt : part) {
			if (pt.empty) {
				no_bulk_before_ = wend;  // irregular window: the sequential rule decides it; do not scan it again
				return 0;
			}
			lines += pt.lines;
		}
		if (lines < 2)
			return 0;
		// an odd number of lines: the last one is a header whose sequence lies beyond the window (or beyond the end of the
		// file) -- it is left for the next call; as a line of parity 0 it contributes nothing to the sums below
		size_t new_beg = wend;
		if (lines & 1) {
			const size_t content_end = data_[wend - 1] == '\n' ? wend - 1 : wend;
			const char* nl = content_end > beg_ ? (const char*)memrchr(data_ + beg_, '\n', content_end - beg_) : nullptr;
			if (!nl)
				return 0;
			new_beg = (size_t)(nl - data_) + 1;
		}
		// global parity of every part's first line; sequences are the lines of global parity 1
		uint64_t before = 0, nseq_total = 0, bytes_total = 0;
		std::vector<uint64_t> seq_at(nthreads), byte_at(nthreads), kmer_at(nthreads);
		std::vector<unsigned> seqpar(nthreads);
		uint64_t kmers_total = 0;
		for (unsigned t = 0; t < nthreads; t++) {
			seqpar[t] = (unsigned)((before & 1) ^ 1);  // local parity of the sequence lines in this part
			seq_at[t] = nseq_total;
			byte_at[t] = bytes_total;
			kmer_at[t] = kmers_total;
			nseq_total += part[t].nseq[seqpar[t]];
			bytes_total += part[t].bytes[seqpar[t]];
			kmers_total += part[t].kmers[seqpar[t]];
			before += part[t].lines;
		}
		if (bytes_total > cap)
			return 0;
		const uint64_t off0 = offs.back(), koff0 = koffs ? koffs->back() : 0;
		const size_t idx0 = offs.size();
		offs.resize(idx0 + nseq_total);
		if (koffs)
			koffs->resize(idx0 + nseq_total);
		uint64_t* offp = offs.data() + idx0;
		uint64_t* koffp = koffs ? koffs->data() + idx0 : nullptr;
		auto emit = [base, wend, k, minlen, dst, offp, koffp, off0, koff0](const Part& pt, unsigned sp, uint64_t si, uint64_t bi, uint64_t ki) {
			size_t p = pt.a;
			uint64_t line = 0;
			while (p < pt.z) {
				const char* nl = (const char*)memchr(base + p, '\n', wend - p);
				const size_t e = nl ? (size_t)(nl - base) : wend;
				const size_t len = e - p;
				if ((line & 1) == sp && len >= minlen && len >= k) {
					memcpy(dst + bi, base + p, len);
					bi += len;
					ki += len - k + 1;
					offp[si] = off0 + bi;
					if (koffp)
						koffp[si] = koff0 + ki;
					si++;
				}
				line++;
				p = e + 1;
			}
		};
		{
			std::vector<std::thread> th;
			for (unsigned t = 1; t < nthreads; t++)
				th.emplace_back(emit, std::cref(part[t]), seqpar[t], seq_at[t], byte_at[t], kmer_at[t]);
			emit(part[0], seqpar[0], seq_at[0], byte_at[0], kmer_at[0]);
			for (auto& x : th)
				x.join();
		}
		beg_ = new_beg;
		return (size_t)nseq_total;
	}

	// next line that starts with A/C/G/T (the payload snippets treat every such LINE as a sequence, whatever the record
	// structure: Colored_De_Bruijn_graph_snippet.cpp:139); false at end of file
	bool next_acgt_line(const char*& p, size_t& len)
	{
		while (next_line(p, len)) {
			if (len && (p[0] == 'A' || p[0] == 'C' || p[0] == 'G' || p[0] == 'T'))
				return true;
		}
		return false;
	}

private:
	static constexpr size_t kChunk = 8u << 20;
	gzFile f_ = nullptr;
	std::vector<char> buf_;
	const char* map_ = nullptr;
	size_t map_bytes_ = 0;
	const char* data_ = nullptr;  // buf_.data() or the mapping
	size_t beg_ = 0, end_ = 0;
	size_t no_bulk_before_ = 0;
	bool eof_ = false, opened_ = false;

	bool refill()
	{
		if (eof_)
			return false;
		if (beg_ > 0) {  // keep the unread tail
			memmove(buf_.data(), buf_.data() + beg_, end_ - beg_);
			end_ -= beg_;
			beg_ = 0;
		}
		if (buf_.size() - end_ < kChunk)
			buf_.resize(end_ + kChunk);
		data_ = buf_.data();
		const int n = gzread(f_, buf_.data() + end_, (unsigned)kChunk);
		if (n <= 0) {
			eof_ = true;
			return false;
		}
		end_ += (size_t)n;
		return true;
	}

	// a line without its '\n'; the view is valid until the next call
	bool next_line(const char*& p, size_t& len)
	{
		size_t scan_from = beg_;
		for (;;) {
			const char* nl = end_ > scan_from ? (const char*)memchr(data_ + scan_from, '\n', end_ - scan_from) : nullptr;
			if (nl) {
				p = data_ + beg_;
				len = (size_t)(nl - p);
				beg_ = (size_t)(nl - data_) + 1;
				return true;
			}
			const size_t had = end_ - beg_;
			if (!refill()) {
				if (end_ > beg_) {  // last line without a newline
					p = data_ + beg_;
					len = end_ - beg_;
					beg_ = end_;
					return true;
				}
				return false;
			}
			scan_from = beg_ + had;  // refill moved the tail to the front
		}
	}
};

// ---- on-device recap counters ------------------------------------------------------------------------------------------
// counts = kCountStripes pairs (found, not found): every block adds its totals to ONE pair, blocks are spread over the pairs --
// a single hot counter serialises at ~12 ns per atomic on MI355X, which made the per-wave version of this kernel the most
// expensive kernel of file_query (3.9 ms per 2.2e7 IDs); the caller sums the pairs.
__global__ __launch_bounds__(256) void k_count_found(const int64_t* __restrict__ ids, uint64_t n, unsigned long long* __restrict__ counts)
{
	__shared__ unsigned int s_found, s_absent;
	if (threadIdx.x == 0)
		s_found = s_absent = 0;
	__syncthreads();
	unsigned int f = 0, a = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const bool found = ids[i] >= 0;
		f += found ? 1u : 0u;
		a += found ? 0u : 1u;
	}
	for (int o = 32; o > 0; o >>= 1) {
		f += __shfl_xor(f, o);
		a += __shfl_xor(a, o);
	}
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&s_found, f);
		atomicAdd(&s_absent, a);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned stripe = blockIdx.x % kCountStripes;
		if (s_found)
			atomicAdd(&counts[2 * stripe], (unsigned long long)s_found);
		if (s_absent)
			atomicAdd(&counts[2 * stripe + 1], (unsigned long long)s_absent);
	}
}

void launch_count_found(const int64_t* d_ids, uint64_t n, unsigned long long* d_counts, hipStream_t s)
{
	if (n == 0)
		return;
	const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 2048);
	hipLaunchKernelGGL(k_count_found, dim3(grid), dim3(256), 0, s, d_ids, n, d_counts);
	BLMX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_narrow_ids(const int64_t* __restrict__ ids, uint64_t n, uint32_t* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		out[i] = (uint32_t)ids[i];  // -1 -> 0xffffffff; IDs are below 2^32 - 1 (checked by the caller)
}

void launch_narrow_ids(const int64_t* d_ids, uint64_t n, uint32_t* d_out, hipStream_t s)
{
	if (n == 0)
		return;
	hipLaunchKernelGGL(k_narrow_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ids, n, d_out);
	BLMX_HIP(hipGetLastError());
}

// Pinning host memory costs about 0.7 ms per MiB here (3 x 128 MiB: 285 ms, profiles/r03_readbench.txt): the window slots of an index
// that is destroyed are parked (up to kPinnedCacheBytes per process and device) and serve the next index instead of being unpinned.
namespace {
constexpr size_t kPinnedCacheBytes = 1ull << 30, kPinnedCacheMin = 1u << 20;
struct PinnedCache {
	struct Block {
		void* p;
		size_t bytes;
		int device;
	};
	std::mutex mu;
	std::vector<Block> blocks;
	size_t held = 0;
	static PinnedCache& get()
	{
		static PinnedCache* c = new PinnedCache();  // never destroyed: the HIP runtime may be gone when static destructors run
		return *c;
	}
};
}  // namespace

size_t release_pinned_cache()
{
	PinnedCache& c = PinnedCache::get();
	std::vector<PinnedCache::Block> mine;
	{
		std::lock_guard<std::mutex> lock(c.mu);
		mine.swap(c.blocks);
		c.held = 0;
	}
	size_t n = 0;
	for (const PinnedCache::Block& b : mine) {
		(void)hipHostFree(b.p);
		n += b.bytes;
	}
	return n;
}

void Pinned::release()
{
	if (!p)
		return;
	PinnedCache& c = PinnedCache::get();
	int dev = 0;
	bool parked = false;
	// (while an exception unwinds, copies may still be in flight: hipHostFree waits for them, a parked block would not)
	if (bytes >= kPinnedCacheMin && std::uncaught_exceptions() == 0 && hipGetDevice(&dev) == hipSuccess) {
		std::lock_guard<std::mutex> lock(c.mu);
		if (c.held + bytes <= kPinnedCacheBytes) {
			c.blocks.push_back({p, bytes, dev});
			c.held += bytes;
			parked = true;
		}
	}
	if (!parked)
		(void)hipHostFree(p);
	p = nullptr;
	bytes = 0;
}

void Pinned::ensure(size_t n, size_t keep)
{
	if (n <= bytes)
		return;
	const size_t nb = n + n / 4 + 4096;
	void* q = nullptr;
	size_t qb = nb;
	int dev = 0;
	if (n >= kPinnedCacheMin && hipGetDevice(&dev) == hipSuccess) {
		PinnedCache& c = PinnedCache::get();
		std::lock_guard<std::mutex> lock(c.mu);
		size_t best = c.blocks.size();
		for (size_t i = 0; i < c.blocks.size(); i++)
			if (c.blocks[i].device == dev && c.blocks[i].bytes >= n && (best == c.blocks.size() || c.blocks[i].bytes < c.blocks[best].bytes))
				best = i;
		if (best != c.blocks.size()) {
			q = c.blocks[best].p;
			qb = c.blocks[best].bytes;
			c.held -= qb;
			c.blocks.erase(c.blocks.begin() + (long)best);
		}
	}
	if (!q)
		BLMX_HIP(hipHostMalloc(&q, nb, hipHostMallocDefault));
	if (p && keep)
		memcpy(q, p, std::min(keep, bytes));
	release();
	p = q;
	bytes = qb;
}

// ---- streaming ingest for construction (reference blight.cpp:212-229 reads line by line) ---------------------
// The host only parses: sequence lines are packed into two pinned 64 MiB buffers that leave for HBM with asynchronous
// copies while the next chunk is being parsed, so the H2D transfer hides behind the (host-bound) parse and no pageable
// whole-file copy of the input is ever made.  The device buffer starts at the file size (an upper bound for a plain file;
// a guess for gzip) and doubles with a device-to-device copy when it runs out.
// the sequences of a FASTA file (plain or gzip) -> device memory: characters back to back + host offsets
void load_fasta_to_device(Index& ix, const char* path, DevBuf& d_ascii, std::vector<uint64_t>& offs, uint64_t& total_out)
{
	RecordReader rd(path);
	if (!rd.ok())
		throw StatusError{BLMX_EIO, "Problem with files opening"};  // blight.cpp:188-189
	BLMX_HIP(hipSetDevice(ix.prm.device));
	uint64_t file_bytes = 0;
	bool gz = false;
	if (FILE* f = fopen(path, "rb")) {
		unsigned char magic[2] = {0, 0};
		gz = fread(magic, 1, 2, f) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
		if (fseek(f, 0, SEEK_END) == 0)
			file_bytes = (uint64_t)ftell(f);
		fclose(f);
	}
	constexpr size_t kChunk = 64u << 20;
	hipStream_t s = nullptr;
	BLMX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	struct StreamGuard {
		hipStream_t s;
		~StreamGuard() { (void)hipStreamDestroy(s); }
	} guard{s};
	Pinned pin[2];
	hipEvent_t sent[2] = {nullptr, nullptr};
	struct EventGuard {
		hipEvent_t* e;
		~EventGuard()
		{
			for (int i = 0; i < 2; i++)
				if (e[i])
					(void)hipEventDestroy(e[i]);
		}
	} eguard{sent};
	for (int i = 0; i < 2; i++) {
		pin[i].ensure(kChunk);
		BLMX_HIP(hipEventCreateWithFlags(&sent[i], hipEventDisableTiming));
	}
	BLMX_HIP(d_ascii.alloc(std::max<uint64_t>(gz ? file_bytes * 5 : file_bytes + 64, kChunk) + 64, false, s));
	offs.assign(1, 0);
	uint64_t total = 0;  // bytes already on their way to the device
	size_t fill = 0;
	int cur = 0;
	bool busy[2] = {false, false};
	auto flush = [&] {
		if (fill == 0)
			return;
		if (total + fill + 64 > d_ascii.bytes) {  // grow: double, copy what is there (stream-ordered), drop the old block
			DevBuf bigger;
			BLMX_HIP(bigger.alloc(std::max<uint64_t>(2 * d_ascii.bytes, total + fill + 64), false, s));
			BLMX_HIP(hipMemcpyAsync(bigger.p, d_ascii.p, total, hipMemcpyDeviceToDevice, s));
			BLMX_HIP(hipStreamSynchronize(s));
			bigger.swap(d_ascii);
		}
		BLMX_HIP(hipMemcpyAsync((char*)d_ascii.p + total, pin[cur].p, fill, hipMemcpyHostToDevice, s));
		BLMX_HIP(hipEventRecord(sent[cur], s));
		busy[cur] = true;
		total += fill;
		fill = 0;
		cur ^= 1;
		if (busy[cur]) {  // the other buffer is about to be overwritten
			BLMX_HIP(hipEventSynchronize(sent[cur]));
			busy[cur] = false;
		}
	};
	const char* p;
	size_t len;
	for (;;) {
		if (rd.mapped()) {  // plain file: whole windows are parsed by several threads straight into the pinned buffer
			if (pin[cur].bytes - fill < (16u << 20))
				flush();
			if (rd.bulk(pin[cur].as<char>() + fill, pin[cur].bytes - fill, offs, nullptr, 0, 1)) {
				fill = (size_t)(offs.back() - total);
				continue;
			}
		}
		if (!rd.next_sequence(p, len))
			break;
		if (fill + len > pin[cur].bytes) {
			flush();
			pin[cur].ensure(len);  // a single sequence longer than the chunk
		}
		memcpy(pin[cur].as<char>() + fill, p, len);
		fill += len;
		offs.push_back(total + fill);
	}
	flush();
	BLMX_HIP(hipStreamSynchronize(s));
	total_out = total;
}

void build_from_fasta_streamed(Index& ix, const char* path)
{
	if (ix.built)
		throw StatusError{BLMX_ESTATE, "index already built"};
	DevBuf d_ascii, d_offs;
	uint64_t total = 0, nseq = 0;
	const char* e = getenv("BLIGHT_HOST_PARSE");
	if (e && atoi(e) != 0) {  // the round-2 reader: host threads parse the records (kept for comparison)
		std::vector<uint64_t> offs;
		load_fasta_to_device(ix, path, d_ascii, offs, total);
		nseq = offs.size() - 1;
		BLMX_HIP(d_offs.alloc((nseq + 1) * 8, false, ix.stream));
		BLMX_HIP(hipMemcpyAsync(d_offs.p, offs.data(), (nseq + 1) * 8, hipMemcpyHostToDevice, ix.stream));
		BLMX_HIP(hipStreamSynchronize(ix.stream));
	} else {
		const bool trace = getenv("BLIGHT_TRACE") && atoi(getenv("BLIGHT_TRACE")) != 0;
		const auto t0 = std::chrono::steady_clock::now();
		load_fasta_raw_to_device(ix, path, d_ascii, d_offs, nseq, total);  // (d_ascii: the 2-bit text)
		const auto t1 = std::chrono::steady_clock::now();
		build_from_device_packed(ix, d_ascii, d_offs.as<uint64_t>(), nseq, total);
		const auto t2 = std::chrono::steady_clock::now();
		ix.free_after_filter(d_ascii);
		ix.free_after_filter(d_offs);
		ix.built = true;
		if (trace)
			fprintf(stderr, "[blmx front] construct_index(file): load %.1f ms, build %.1f ms, free %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
			        std::chrono::duration<double, std::milli>(t2 - t1).count(),
			        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
		return;
	}
	build_from_device(ix, d_ascii.as<uint8_t>(), d_offs.as<uint64_t>(), nseq, total);
	ix.free_after_filter(d_ascii);
	ix.free_after_filter(d_offs);
	ix.built = true;
}

// ---- index image on disk (the fork has no dump/load; mainline Blight's dump_disk / load constructor is the model) --
// Layout: FileHeader, shard bounds (u32 x nbounds), fallback group tags (u32 x nfb), then the seven HBM arrays exactly as they
// sit in device memory (padding included), each preceded by its byte count, then the end tag.  Arrays move through a pinned
// 64 MiB bounce buffer; nothing is recomputed on load except the host mirror of the group table.
namespace {

constexpr char kImageMagic[8] = {'B', 'L', 'M', 'X', 'I', 'D', 'X', '1'};
constexpr char kImageEnd[8] = {'B', 'L', 'M', 'X', 'E', 'N', 'D', '1'};
// (r04: 3 / 4 instead of 1 / 2 -- level-0 units in cross-paired 128-B blocks, blmx_stats_t grew; images of earlier rounds are refused)
constexpr uint32_t kImageVersion = 3;
constexpr uint32_t kImageVersionFilter = 4;  // + the absent-key filter after the fallback ranks

struct FileHeader {
	char magic[8];
	uint32_t version, abi, stats_bytes, group_desc_bytes;
	uint32_t k, m, log2_mphf, log2_superbuckets, bits_saved, kmer_bytes;
	uint32_t kw, gshift, nbuckets, ngroups, shard_rank, shard_world, nbounds, filter_bpk;  // filter_bpk: 0 (version 1 images) = no absent-key filter
	uint64_t nfb, positions_words, seq_words, total_lines, total_logical_words;
	blmx_stats_t st;
};

struct File {
	FILE* f = nullptr;
	~File()
	{
		if (f)
			fclose(f);
	}
};

void put(FILE* f, const void* p, size_t n)
{
	if (n && fwrite(p, 1, n, f) != n)
		throw StatusError{BLMX_EIO, "short write to the index file"};
}
void get(FILE* f, void* p, size_t n)
{
	if (n && fread(p, 1, n, f) != n)
		throw StatusError{BLMX_EIO, "index file truncated"};
}

constexpr size_t kBounce = 64u << 20;

void put_device(FILE* f, const DevBuf& b, Pinned& pin, hipStream_t s)
{
	const uint64_t bytes = b.bytes;
	put(f, &bytes, 8);
	for (size_t o = 0; o < bytes; o += kBounce) {
		const size_t n = std::min(kBounce, (size_t)bytes - o);
		BLMX_HIP(hipMemcpyAsync(pin.p, (const char*)b.p + o, n, hipMemcpyDeviceToHost, s));
		BLMX_HIP(hipStreamSynchronize(s));
		put(f, pin.p, n);
	}
}

void get_device(FILE* f, DevBuf& b, uint64_t min_bytes, Pinned& pin, hipStream_t s)
{
	uint64_t bytes = 0;
	get(f, &bytes, 8);
	if (bytes < min_bytes || bytes > (1ull << 40))
		throw StatusError{BLMX_EIO, "index file is inconsistent (array size)"};
	BLMX_HIP(b.alloc(bytes, false, s));
	for (size_t o = 0; o < bytes; o += kBounce) {
		const size_t n = std::min(kBounce, (size_t)bytes - o);
		get(f, pin.p, n);
		BLMX_HIP(hipMemcpyAsync((char*)b.p + o, pin.p, n, hipMemcpyHostToDevice, s));
		BLMX_HIP(hipStreamSynchronize(s));
	}
}

}  // namespace

void save_index(Index& ix, const char* path)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	BLMX_HIP(hipSetDevice(ix.prm.device));
	ix.filter_wait();  // the image holds the finished filter
	File fh;
	fh.f = fopen(path, "wb");
	if (!fh.f)
		throw StatusError{BLMX_EIO, "Problem with files opening"};
	FileHeader h{};
	memcpy(h.magic, kImageMagic, 8);
	h.version = (ix.filter_bpk || ix.pos_hint) ? kImageVersionFilter : kImageVersion;
	// low byte: filter bits per key; next byte: hint bits of the position fields; bit 16: the filter lives in 64-B level-0 units; bit 17: those
	// are records (32 level bits + the position fields of their keys)
	h.filter_bpk = ix.filter_bpk | (ix.pos_hint << 8) | ((uint32_t)ix.fat_lines << 16) | ((uint32_t)ix.rec_lines << 17);
	h.abi = BLMX_ABI_VERSION;
	h.stats_bytes = sizeof(blmx_stats_t);
	h.group_desc_bytes = sizeof(GroupDesc);
	h.k = ix.prm.k;
	h.m = ix.prm.m;
	h.log2_mphf = ix.prm.log2_mphf;
	h.log2_superbuckets = ix.prm.log2_superbuckets;
	h.bits_saved = ix.prm.bits_saved;
	h.kmer_bytes = ix.prm.kmer_bytes;
	h.kw = (uint32_t)ix.KW;
	h.gshift = (uint32_t)ix.gshift;
	h.nbuckets = ix.nbuckets;
	h.ngroups = ix.ngroups;
	h.shard_rank = ix.shard_rank;
	h.shard_world = ix.shard_world;
	h.nbounds = (uint32_t)ix.shard_bounds.size();
	h.nfb = ix.h_fb_group.size();
	h.positions_words = ix.positions_words;
	h.seq_words = ix.seq_words;
	h.total_lines = ix.total_lines;
	h.total_logical_words = ix.total_logical_words;
	h.st = ix.st;
	put(fh.f, &h, sizeof h);
	put(fh.f, ix.shard_bounds.data(), ix.shard_bounds.size() * 4);
	put(fh.f, ix.h_fb_group.data(), ix.h_fb_group.size() * 4);
	Pinned pin;
	pin.ensure(kBounce);
	for (const DevBuf* b : {&ix.buckets, &ix.groups, &ix.seq, &ix.positions, &ix.lines, &ix.fb_keys, &ix.fb_ranks})
		put_device(fh.f, *b, pin, ix.stream);
	if (ix.filter_bpk && !ix.fat_lines)
		put_device(fh.f, ix.filter, pin, ix.stream);
	put(fh.f, kImageEnd, 8);
	if (fflush(fh.f) != 0)
		throw StatusError{BLMX_EIO, "short write to the index file"};
}

int peek_index_params(const char* path, blmx_params& out)
{
	File fh;
	fh.f = fopen(path, "rb");
	if (!fh.f)
		return BLMX_EIO;
	FileHeader h{};
	if (fread(&h, 1, sizeof h, fh.f) != sizeof h || memcmp(h.magic, kImageMagic, 8) != 0 || (h.version != kImageVersion && h.version != kImageVersionFilter))
		return BLMX_EIO;
	out = blmx_params{};
	out.k = h.k;
	out.m = h.m;
	out.log2_mphf = h.log2_mphf;
	out.log2_superbuckets = h.log2_superbuckets;
	out.threads = 1;
	out.bits_saved = h.bits_saved;
	out.kmer_bytes = h.kmer_bytes;
	return BLMX_OK;
}

void load_index(Index& ix, const char* path)
{
	if (ix.built)
		throw StatusError{BLMX_ESTATE, "index already built"};
	BLMX_HIP(hipSetDevice(ix.prm.device));
	File fh;
	fh.f = fopen(path, "rb");
	if (!fh.f)
		throw StatusError{BLMX_EIO, "Problem with files opening"};
	FileHeader h{};
	get(fh.f, &h, sizeof h);
	if (memcmp(h.magic, kImageMagic, 8) != 0)
		throw StatusError{BLMX_EIO, "not a Blight MI355X index file"};
	const uint32_t img_hint = (h.filter_bpk >> 8) & 0xffu;
	const bool img_fat = (h.filter_bpk >> 16) & 1u, img_rec = (h.filter_bpk >> 17) & 1u;
	h.filter_bpk &= 0xffu;
	if ((h.version != kImageVersion && h.version != kImageVersionFilter) || (h.version == kImageVersion && (h.filter_bpk || img_hint)) || h.filter_bpk > 32 ||
	    (img_fat && h.filter_bpk != 8) || (img_rec && !img_fat) ||
	    img_hint > 3 || img_hint > h.bits_saved ||
	    h.stats_bytes != sizeof(blmx_stats_t) || h.group_desc_bytes != sizeof(GroupDesc))
		throw StatusError{BLMX_EIO, "index file written by an incompatible library version"};
	if (h.k != ix.prm.k || h.m != ix.prm.m || h.log2_mphf != ix.prm.log2_mphf || h.bits_saved != ix.prm.bits_saved ||
	    h.kmer_bytes != ix.prm.kmer_bytes)
		throw StatusError{BLMX_EINVAL, "index file parameters (k, m, log2_mphf, bits_to_save, k-mer width) differ from this object's"};
	if (h.kw != (uint32_t)ix.KW || h.gshift != (uint32_t)ix.gshift || h.nbuckets != ix.nbuckets || h.ngroups != ix.ngroups ||
	    h.nbounds > (1u << 20) || h.nfb > (1ull << 36))
		throw StatusError{BLMX_EIO, "index file is inconsistent (header)"};
	hipStream_t s = ix.stream;
	ix.shard_bounds.resize(h.nbounds);
	get(fh.f, ix.shard_bounds.data(), (size_t)h.nbounds * 4);
	ix.h_fb_group.resize(h.nfb);
	get(fh.f, ix.h_fb_group.data(), (size_t)h.nfb * 4);
	Pinned pin;
	pin.ensure(kBounce);
	get_device(fh.f, ix.buckets, (uint64_t)h.nbuckets * sizeof(BucketDesc), pin, s);
	get_device(fh.f, ix.groups, (uint64_t)h.ngroups * sizeof(GroupDesc), pin, s);
	get_device(fh.f, ix.seq, h.seq_words * 8, pin, s);
	get_device(fh.f, ix.positions, h.positions_words * 8, pin, s);
	get_device(fh.f, ix.lines, (h.total_lines + 1) * kLineBytes, pin, s);
	get_device(fh.f, ix.fb_keys, (h.nfb + 1) * 8 * h.kw, pin, s);
	get_device(fh.f, ix.fb_ranks, (h.nfb + 1) * 8, pin, s);
	ix.filter_bpk = 0;
	ix.filter_words = 0;
	ix.fat_lines = img_fat;
	ix.rec_lines = img_rec;
	if (h.filter_bpk) {
		if (h.st.local_number_kmer > (1ull << 40))
			throw StatusError{BLMX_EIO, "index file is inconsistent (header)"};
		if (!img_fat) {
			ix.filter_words = ((h.st.local_number_kmer * h.filter_bpk) >> 6) + 64;
			get_device(fh.f, ix.filter, ix.filter_words * 8, pin, s);
		}
		ix.filter_bpk = h.filter_bpk;
		init_filter_gate(ix, s);
	}
	ix.filter_state.store(2);
	char end[8];
	get(fh.f, end, 8);
	if (memcmp(end, kImageEnd, 8) != 0)
		throw StatusError{BLMX_EIO, "index file is inconsistent (end tag)"};
	ix.h_groups.resize(h.ngroups);
	BLMX_HIP(hipMemcpyAsync(ix.h_groups.data(), ix.groups.p, (size_t)h.ngroups * sizeof(GroupDesc), hipMemcpyDeviceToHost, s));
	std::vector<BucketDesc> hb(h.nbuckets);
	BLMX_HIP(hipMemcpyAsync(hb.data(), ix.buckets.p, (size_t)h.nbuckets * sizeof(BucketDesc), hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	// Every offset the kernels follow must stay inside its array: a corrupted or foreign image with plausible sizes ends
	// in BLMX_EIO here instead of a GPU memory fault later.
	{
		auto bad = [&](const char* what) {
			for (DevBuf* b : {&ix.buckets, &ix.groups, &ix.qheads, &ix.seq, &ix.positions, &ix.lines, &ix.fb_keys, &ix.fb_ranks, &ix.fb_table, &ix.fb_dir, &ix.filter, &ix.filter_gate})
				b->release();
			ix.filter_bpk = 0;
			ix.fat_lines = false;
			ix.rec_lines = false;
			ix.h_groups.clear();
			ix.h_fb_group.clear();
			throw StatusError{BLMX_EIO, std::string("index file is inconsistent (") + what + ")"};
		};
		const uint64_t total_nt = h.st.total_nt;
		if (total_nt / 32 + 1 > h.seq_words)
			bad("sequence length");
		for (uint32_t bkt = 0; bkt < h.nbuckets; bkt++)
			if (hb[bkt].start > total_nt || (uint64_t)hb[bkt].nt > total_nt - hb[bkt].start)
				bad("bucket table");
		uint64_t keys = 0;
		for (uint32_t g = 0; g < h.ngroups; g++) {
			const GroupDesc& gd = ix.h_groups[g];
			uint64_t bits = 0;
			for (int l = 0; l < kLevels; l++) {
				if (gd.size[l] == 0 || gd.size[l] > (1ull << 40))
					bad("level sizes");
				bits += gd.size[l];
			}
			uint64_t glines = gd.nkeys ? (bits / 32 + kLineData32 - 1) / kLineData32 : 0;  // 16-B units of the group's lines
			if (img_fat && gd.nkeys) {
				const uint64_t w0 = gd.size[0] / 32, F = img_rec ? w0 : (w0 + kLineData32 - 1) / kLineData32;
				glines = 8 * ((F + 1) / 2) + (bits / 32 - w0 + kLineData32 - 1) / kLineData32;
				if (gd.line_base % 8)  // (128-B blocks)
					bad("group table");
			}
			if (gd.nbits <= img_hint || gd.nbits > 32 || gd.nkeys > (1ull << 40) || gd.line_base > h.total_lines || glines > h.total_lines - gd.line_base ||
			    gd.posstart > h.positions_words * 64 || gd.nkeys * gd.nbits > h.positions_words * 64 - gd.posstart || gd.fb_start > h.nfb ||
			    gd.fb_count > h.nfb - gd.fb_start)
				bad("group table");
			if (ix.filter_bpk && !img_fat) {  // the filter region a query of this group can touch (index_device.h: filter_word)
				const uint64_t local = gd.prefix - ix.h_groups[0].prefix;
				if (gd.prefix < ix.h_groups[0].prefix || local > h.st.local_number_kmer ||
				    ((local * ix.filter_bpk) >> 6) + (((gd.size[0] >> 1) * ix.filter_bpk) >> 6) + 1 > ix.filter_words)
					bad("filter regions");
			}
			keys += gd.nkeys;
		}
		if (keys != h.st.local_number_kmer)
			bad("key counts");
	}
	make_query_heads(ix);
	make_fallback_table(ix, s);  // (derived from the sorted entries: not part of the image)
	ix.pos_hint = img_hint;
	ix.shard_rank = h.shard_rank;
	ix.shard_world = h.shard_world;
	ix.positions_words = h.positions_words;
	ix.seq_words = h.seq_words;
	ix.total_lines = h.total_lines;
	ix.total_logical_words = h.total_logical_words;
	ix.st = h.st;
	ix.st.number_query = 0;  // the live counter is ix.number_query
	ix.built = true;
}

// ---- payload kernels --------------------------------------------------------------------------------------------------
// abundance[id]++ on uint8 counters (Abundance_De_Bruijn_graph_snippet.cpp:147-150: wraps at 256 like the reference's
// uint8_t): four counters share a 32-bit word, the increment is a CAS on the word that touches only its own byte.
__global__ __launch_bounds__(256) void k_payload_abundance(const int64_t* __restrict__ ids, uint64_t n, uint64_t nids,
                                                           uint32_t* __restrict__ words)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const int64_t id = ids[i];
	if (id < 0 || (uint64_t)id >= nids)  // absent k-mer, or an ID that is not of this index
		return;
	uint32_t* w = &words[(uint64_t)id >> 2];
	const uint32_t sh = 8u * ((uint32_t)id & 3u);
	uint32_t old = *w, assumed;
	do {
		assumed = old;
		const uint32_t nb = ((assumed >> sh) + 1u) & 0xffu;
		old = atomicCAS(w, assumed, (assumed & ~(0xffu << sh)) | (nb << sh));
	} while (old != assumed);
}

// color_me_amaze[id * color_number + i_file] = true (Colored_De_Bruijn_graph_snippet.cpp:146-150); bit index order of the
// reference's vector<bool>, 32 bits per word, least significant first.
__global__ __launch_bounds__(256) void k_payload_color(const int64_t* __restrict__ ids, uint64_t n, uint64_t nids, uint32_t color,
                                                       uint32_t ncolors, uint32_t* __restrict__ bits)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const int64_t id = ids[i];
	if (id < 0 || (uint64_t)id >= nids)
		return;
	const uint64_t bit = (uint64_t)id * ncolors + color;
	atomicOr(&bits[bit >> 5], 1u << (bit & 31u));
}

void payload_init(Payload& pl)
{
	Index& ix = *pl.ix;
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	if (pl.kind != 0 && pl.kind != 1)
		throw StatusError{BLMX_EINVAL, "payload kind must be 0 (abundance) or 1 (colours)"};
	if (pl.kind == 1 && pl.ncolors == 0)
		throw StatusError{BLMX_EINVAL, "a colour payload needs at least one colour"};
	BLMX_HIP(hipSetDevice(ix.prm.device));
	pl.nids = ix.st.number_kmer;
	pl.bytes = pl.kind == 0 ? pl.nids : (pl.nids * pl.ncolors + 7) / 8;
	BLMX_HIP(pl.data.alloc((pl.bytes + 3) / 4 * 4 + 4, true, ix.stream));
	BLMX_HIP(hipStreamSynchronize(ix.stream));
}

void payload_add_ids(Payload& pl, const int64_t* d_ids, uint64_t n, uint32_t color, hipStream_t s)
{
	if (n == 0)
		return;
	if (pl.kind == 1 && color >= pl.ncolors)
		throw StatusError{BLMX_EINVAL, "colour out of range"};
	const unsigned grid = (unsigned)((n + 255) / 256);
	if (pl.kind == 0)
		hipLaunchKernelGGL(k_payload_abundance, dim3(grid), dim3(256), 0, s, d_ids, n, pl.nids, pl.data.as<uint32_t>());
	else
		hipLaunchKernelGGL(k_payload_color, dim3(grid), dim3(256), 0, s, d_ids, n, pl.nids, color, pl.ncolors, pl.data.as<uint32_t>());
	BLMX_HIP(hipGetLastError());
}

}  // namespace blmx
