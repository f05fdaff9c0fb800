// bench_calls.cpp -> tools/libbench_calls.so -- the reference's per-read calling pattern, timed from inside one process.
//
// bench.py builds its index through the C ABI; this helper attaches the drop-in C++ facade (include/blight.h) to that
// handle and calls kmer_Set_Light::query_sequence_hash(const std::string&) once per read from `threads` OpenMP threads,
// exactly like the reference's applications (Colored_De_Bruijn_graph_snippet.cpp:122-157).  Returns 0 and fills the
// timing / result counters; the IDs themselves are summed into an order-independent checksum.
#include <omp.h>

#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "blight.h"

extern "C" int blight_bench_read_calls(blmx_index* h, const char* concat, const uint64_t* offsets, uint64_t nreads, int threads, int repeats,
                                       double* seconds, uint64_t* kmers, uint64_t* found, uint64_t* absent, uint64_t* checksum)
{
	try {
		std::unique_ptr<kmer_Set_Light> index(kmer_Set_Light::attach(h));
		std::vector<std::string> reads(nreads);
		for (uint64_t i = 0; i < nreads; i++)
			reads[i].assign(concat + offsets[i], concat + offsets[i + 1]);
		if (nreads)
			index->query_sequence_hash(reads[0]);  // the first call creates the calling slots
		uint64_t nk = 0, nf = 0, na = 0, cs = 0;
		const auto t0 = std::chrono::steady_clock::now();
		for (int rep = 0; rep < repeats; rep++) {
			nk = nf = na = cs = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16) reduction(+ : nk, nf, na, cs)
			for (long i = 0; i < (long)nreads; i++) {
				const std::vector<int64_t> ids = index->query_sequence_hash(reads[(size_t)i]);
				nk += ids.size();
				for (size_t j = 0; j < ids.size(); j++) {
					(ids[j] >= 0 ? nf : na)++;
					cs += ((uint64_t)i * 1000003ull + j + 1) * (uint64_t)(ids[j] + 2);
				}
			}
		}
		*seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (repeats > 0 ? repeats : 1);
		*kmers = nk;
		*found = nf;
		*absent = na;
		*checksum = cs;
		return 0;
	} catch (const std::exception&) {
		return -1;
	}
}

// The scalar calling pattern: kmer_Set_Light::query_kmer_hash(kmer_t) once per k-mer (reference blight.cpp:545-550) from `threads`
// OpenMP threads.  kmers = n canonical k-mers (one or two 64-bit words each, low word first); ids_out (may be null) gets the answers.
extern "C" int blight_bench_kmer_calls(blmx_index* h, const uint64_t* kmers, uint64_t n, int threads, double* seconds, uint64_t* found, uint64_t* checksum,
                                       int64_t* ids_out)
{
	try {
		std::unique_ptr<kmer_Set_Light> index(kmer_Set_Light::attach(h));
		constexpr int kWords = sizeof(kmer_t) / 8;
		auto kmer_at = [&](uint64_t i) -> kmer_t {
			kmer_t x = 0;
			for (int w = kWords - 1; w >= 0; w--)
				x = (kWords > 1 ? (x << 32) << 32 : (kmer_t)0) | (kmer_t)kmers[i * kWords + (uint64_t)w];
			return x;
		};
		if (n)
			index->query_kmer_hash(kmer_at(0));
		uint64_t nf = 0, cs = 0;
		const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64) reduction(+ : nf, cs)
		for (long i = 0; i < (long)n; i++) {
			const int64_t id = index->query_kmer_hash(kmer_at((uint64_t)i));
			if (ids_out)
				ids_out[i] = id;
			nf += id >= 0;
			cs += ((uint64_t)i + 1) * (uint64_t)(id + 2);
		}
		*seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		*found = nf;
		*checksum = cs;
		return 0;
	} catch (const std::exception&) {
		return -1;
	}
}
