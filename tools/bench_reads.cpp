// bench_reads.cpp -- the reference's per-read calling pattern on the MI355X path, timed and checksummed.
//
// The reference's applications call query_sequence_hash once per read from OpenMP threads
// (Colored_De_Bruijn_graph_snippet.cpp:122-157, Abundance_De_Bruijn_graph_snippet.cpp:122-157).  This tool does exactly that
// through the drop-in facade (include/blight.h): every line of -q that starts with A/C/G/T is one call, -t threads call
// concurrently.  It prints ONE JSON line: calls/s, k-mers/s, found / absent counts and an order-independent checksum of
// (read index, k-mer index, ID) that tests compare with the CPU oracle.
//
//   bench_reads -g unitigs.fa -q reads.txt -k 31 [-m 9 -n 14 -b 6] [-t 8] [-r repeats] [-l index image instead of -g]
#include <omp.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "blight.h"

int main(int argc, char** argv)
{
	std::map<char, std::string> text;
	std::map<char, unsigned> number{{'k', 31}, {'m', 9}, {'n', 14}, {'s', 6}, {'b', 6}, {'t', 8}, {'r', 1}};
	for (int c; (c = getopt(argc, argv, "g:q:l:k:m:n:s:b:t:r:")) != -1;) {
		if (number.count((char)c))
			number[(char)c] = (unsigned)std::stoul(optarg);
		else if (c != '?')
			text[(char)c] = optarg;
	}
	if ((text['g'].empty() && text['l'].empty()) || text['q'].empty()) {
		std::fprintf(stderr, "usage: bench_reads -g unitigs.fa | -l index.blmx  -q reads  -k K [-m -n -b] [-t threads] [-r repeats]\n");
		return 2;
	}
	try {
		kmer_Set_Light index(number['k'], number['m'], number['n'], number['s'], 1, number['b']);
		std::streambuf* keep = std::cout.rdbuf(std::cerr.rdbuf());  // the facade's recap goes to stderr: stdout is the JSON line
		if (!text['l'].empty())
			index.load(text['l']);
		else
			index.construct_index(text['g']);
		std::cout.rdbuf(keep);
		std::vector<std::string> reads;
		{
			std::ifstream in(text['q']);
			for (std::string line; std::getline(in, line);)
				if (!line.empty() && (line[0] == 'A' || line[0] == 'C' || line[0] == 'G' || line[0] == 'T'))
					reads.push_back(line);
		}
		const int threads = (int)number['t'];
		uint64_t found = 0, absent = 0, checksum = 0, kmers = 0;
		index.query_sequence_hash(reads.empty() ? std::string("A") : reads[0]);  // first call creates the calling slots
		const auto t0 = std::chrono::steady_clock::now();
		for (unsigned rep = 0; rep < number['r']; rep++) {
			found = absent = checksum = kmers = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16) reduction(+ : found, absent, checksum, kmers)
			for (long i = 0; i < (long)reads.size(); i++) {
				const std::vector<int64_t> ids = index.query_sequence_hash(reads[(size_t)i]);
				kmers += ids.size();
				for (size_t j = 0; j < ids.size(); j++) {
					(ids[j] >= 0 ? found : absent)++;
					checksum += ((uint64_t)i * 1000003ull + j + 1) * (uint64_t)(ids[j] + 2);
				}
			}
		}
		const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / number['r'];
		std::printf("{\"reads\": %zu, \"threads\": %d, \"seconds\": %.6f, \"calls_per_s\": %.1f, \"kmers\": %llu, \"kmers_per_s\": %.1f, "
		            "\"found\": %llu, \"absent\": %llu, \"checksum\": %llu, \"number_query\": %llu}\n",
		            reads.size(), threads, secs, reads.size() / secs, (unsigned long long)kmers, kmers / secs, (unsigned long long)found,
		            (unsigned long long)absent, (unsigned long long)checksum, (unsigned long long)(size_t)index.number_query);
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}
	return 0;
}
