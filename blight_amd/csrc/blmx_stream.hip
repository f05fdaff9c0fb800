// blmx_stream.hip -- streaming front end of the query path: file_query (reference blight.cpp:746-799).
//
// The reference reads 512 records at a time under an OpenMP critical section and answers them k-mer by k-mer.  Here the
// host only parses records (chunked gzread + memchr, plain or gzip), packs them into pinned batches and hands them to
// the GPU; two batches are in flight so that parsing, H2D, the kernels (rolling canonical k-mers -> query pipeline ->
// on-device count of found / not found) and the 16-byte D2H of the counters overlap.  Nothing but the two counters
// comes back over PCIe.
#include <zlib.h>

#include <cstring>

#include "blmx_internal.h"

namespace blmx {

// ---- 2-line FASTA records, reference blight.cpp:212-229 / 760-775 ----------------------------------------------------
// Read a header line; if it is empty, swallow one more line and start over; read the sequence line; skip it if empty.
// Lines are returned as views into an internal buffer that is refilled in 8 MiB chunks (gzread reads plain files too).
class RecordReader {
public:
	explicit RecordReader(const char* path) : f_(gzopen(path, "rb"))
	{
		if (f_)
			gzbuffer(f_, 1 << 20);
		buf_.resize(kChunk);
	}
	~RecordReader()
	{
		if (f_)
			gzclose(f_);
	}
	bool ok() const { return f_ != nullptr; }

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

private:
	static constexpr size_t kChunk = 8u << 20;
	gzFile f_;
	std::vector<char> buf_;
	size_t beg_ = 0, end_ = 0;
	bool eof_ = false;

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
			const char* nl = (const char*)memchr(buf_.data() + scan_from, '\n', end_ - scan_from);
			if (nl) {
				p = buf_.data() + beg_;
				len = (size_t)(nl - p);
				beg_ = (size_t)(nl - buf_.data()) + 1;
				return true;
			}
			const size_t had = end_ - beg_;
			if (!refill()) {
				if (end_ > beg_) {  // last line without a newline
					p = buf_.data() + beg_;
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

int read_fasta_records(const char* path, std::string& concat, std::vector<uint64_t>& offs)
{
	RecordReader rd(path);
	if (!rd.ok())
		return BLMX_EIO;
	offs.assign(1, 0);
	const char* p;
	size_t len;
	while (rd.next_sequence(p, len)) {
		concat.append(p, len);
		offs.push_back(concat.size());
	}
	return BLMX_OK;
}

// ---- on-device recap counters ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count_found(const int64_t* __restrict__ ids, uint64_t n, unsigned long long* __restrict__ counts)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = i < n;
	const bool found = in && ids[i] >= 0;
	const unsigned long long bf = __ballot(found), bi = __ballot(in);
	if ((threadIdx.x & 63) == 0) {
		if (bf)
			atomicAdd(&counts[0], (unsigned long long)__popcll(bf));
		if (bi != bf)
			atomicAdd(&counts[1], (unsigned long long)(__popcll(bi) - __popcll(bf)));
	}
}

namespace {

struct Pinned {
	void* p = nullptr;
	size_t bytes = 0;
	~Pinned()
	{
		if (p)
			(void)hipHostFree(p);
	}
	void ensure(size_t n)
	{
		if (n <= bytes)
			return;
		if (p)
			(void)hipHostFree(p);
		p = nullptr;
		bytes = 0;
		BLMX_HIP(hipHostMalloc(&p, n + n / 4 + 4096, hipHostMallocDefault));
		bytes = n + n / 4 + 4096;
	}
	template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Batch {
	Pinned h_seq, h_offs, h_koffs, h_counts;
	DevBuf d_seq, d_offs, d_koffs, d_kmers, d_ids, d_counts, d_err;
	std::vector<uint64_t> offs, koffs;
	uint64_t total = 0, nk = 0;
	hipStream_t s = nullptr;
	hipEvent_t done = nullptr;
	bool in_flight = false;
	~Batch()
	{
		if (done)
			(void)hipEventDestroy(done);
		if (s)
			(void)hipStreamDestroy(s);
	}
};

void grow(DevBuf& b, size_t bytes, hipStream_t s)
{
	if (bytes > b.bytes)
		BLMX_HIP(b.alloc(bytes + bytes / 4 + 256, false, s));
}

}  // namespace

// file_query: returns good / erroneous k-mer counts (the reference prints them: "Good kmer", "Erroneous kmers").
void query_file(Index& ix, const char* path, uint64_t& good, uint64_t& erroneous)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	RecordReader rd(path);
	if (!rd.ok())
		throw StatusError{BLMX_EIO, "Problem with files opening"};
	BLMX_HIP(hipSetDevice(ix.prm.device));
	const uint64_t k = ix.prm.k;
	constexpr size_t kBatchBytes = 48u << 20;  // sequence characters per batch
	Batch batch[2];
	for (auto& b : batch) {
		BLMX_HIP(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking));
		BLMX_HIP(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
		b.h_counts.ensure(16);
		BLMX_HIP(b.d_counts.alloc(16, true, b.s));
		BLMX_HIP(b.d_err.alloc(4, true, b.s));
	}
	good = erroneous = 0;
	uint64_t queries = 0;
	auto retire = [&](Batch& b) {
		if (!b.in_flight)
			return;
		BLMX_HIP(hipEventSynchronize(b.done));
		good += b.h_counts.as<uint64_t>()[0];
		erroneous += b.h_counts.as<uint64_t>()[1];
		if (reinterpret_cast<int*>(b.h_counts.as<uint64_t>() + 2)[0])
			throw StatusError{BLMX_EDNA, "Invalid char in DNA"};
		queries += b.nk;
		b.in_flight = false;
	};
	const char* p;
	size_t len;
	bool more = true;
	int cur = 0;
	while (more) {
		Batch& b = batch[cur];
		retire(b);  // its pinned buffers are about to be overwritten
		b.offs.assign(1, 0);
		b.koffs.assign(1, 0);
		b.total = 0;
		b.h_seq.ensure(kBatchBytes + (1u << 20));
		while (b.total < kBatchBytes) {
			more = rd.next_sequence(p, len);
			if (!more)
				break;
			if (len < k)  // blight.cpp:782: shorter than k -> skipped
				continue;
			b.h_seq.ensure(b.total + len + 64);
			memcpy(b.h_seq.as<char>() + b.total, p, len);
			b.total += len;
			b.offs.push_back(b.total);
			b.koffs.push_back(b.koffs.back() + (len - k + 1));
		}
		const uint64_t nseq = b.offs.size() - 1;
		b.nk = b.koffs.back();
		if (nseq == 0)
			continue;
		b.h_offs.ensure((nseq + 1) * 8);
		b.h_koffs.ensure((nseq + 1) * 8);
		memcpy(b.h_offs.p, b.offs.data(), (nseq + 1) * 8);
		memcpy(b.h_koffs.p, b.koffs.data(), (nseq + 1) * 8);
		b.h_counts.ensure(32);
		grow(b.d_seq, b.total + 64, b.s);
		grow(b.d_offs, (nseq + 1) * 8, b.s);
		grow(b.d_koffs, (nseq + 1) * 8, b.s);
		grow(b.d_kmers, b.nk * 8 * (size_t)ix.KW, b.s);
		grow(b.d_ids, b.nk * 8, b.s);
		BLMX_HIP(hipMemcpyAsync(b.d_seq.p, b.h_seq.p, b.total, hipMemcpyHostToDevice, b.s));
		BLMX_HIP(hipMemcpyAsync(b.d_offs.p, b.h_offs.p, (nseq + 1) * 8, hipMemcpyHostToDevice, b.s));
		BLMX_HIP(hipMemcpyAsync(b.d_koffs.p, b.h_koffs.p, (nseq + 1) * 8, hipMemcpyHostToDevice, b.s));
		BLMX_HIP(hipMemsetAsync(b.d_counts.p, 0, 16, b.s));
		BLMX_HIP(hipMemsetAsync(b.d_err.p, 0, 4, b.s));
		launch_sequence_kmers(ix, b.d_seq.as<uint8_t>(), b.d_offs.as<uint64_t>(), b.d_koffs.as<uint64_t>(), nseq, b.nk, b.d_kmers.p,
		                      b.d_err.as<int>(), b.s);
		launch_query_ids_auto(ix, b.d_kmers.p, b.nk, b.d_ids.as<int64_t>(), b.s);
		hipLaunchKernelGGL(k_count_found, dim3((unsigned)((b.nk + 255) / 256)), dim3(256), 0, b.s, b.d_ids.as<int64_t>(), b.nk,
		                   b.d_counts.as<unsigned long long>());
		BLMX_HIP(hipGetLastError());
		BLMX_HIP(hipMemcpyAsync(b.h_counts.p, b.d_counts.p, 16, hipMemcpyDeviceToHost, b.s));
		BLMX_HIP(hipMemcpyAsync(b.h_counts.as<uint64_t>() + 2, b.d_err.p, 4, hipMemcpyDeviceToHost, b.s));
		BLMX_HIP(hipEventRecord(b.done, b.s));
		b.in_flight = true;
		cur ^= 1;
	}
	retire(batch[0]);
	retire(batch[1]);
	ix.st.number_query += queries;
}

}  // namespace blmx
