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
