// bench_blight.cpp -- command-line harness with the interface of the reference's tool of the same name (reference
// bench_blight.cpp:37-107): flags -g (graph FASTA) -q (query FASTA) -k -m -n -s -t -b with the reference's defaults
// (k 31, m 9, n 17, s 6, t 1, b 6); when only one of -g / -q is given it serves as both; the index is constructed and the
// query file answered through the facade include/blight.h, which prints the reference's recaps.  Extensions: -d writes the
// index image after construction, -l loads one instead of constructing.  Host C++ only; all compute is in
// libblight_mi355x.so.
//
//   g++ -O2 -std=c++17 -Iinclude tools/bench_blight.cpp -o tools/bench_blight -Lblight_amd -lblight_mi355x -lz
//       -Wl,-rpath,'$ORIGIN/../blight_amd'
#include <unistd.h>

#include <cstdio>
#include <iostream>
#include <map>
#include <string>

#include "blight.h"

namespace {

struct Options {
	std::map<char, std::string> text;      // -g -q -d -l
	std::map<char, unsigned> number{{'k', 31}, {'m', 9}, {'n', 17}, {'s', 6}, {'t', 1}, {'b', 6}};
};

void usage(const Options& o)
{
	std::printf("bench_blight (MI355X)  --  build a Blight index from unitigs and query it\n\n"
	            "  required\n"
	            "    -g FILE   unitig FASTA to index          (defaults to -q)\n"
	            "    -q FILE   FASTA whose k-mers are queried (defaults to -g)\n"
	            "    -k INT    k-mer length [%u]\n\n"
	            "  tuning\n"
	            "    -m INT    minimizer length, odd [%u]\n"
	            "    -n INT    log2 of the number of MPHFs, at most 2m-1 [%u]\n"
	            "    -b INT    position bits dropped per k-mer: a query scans 2^b positions [%u]\n"
	            "    -s INT    log2 of the reference's temp-file count; kept for compatibility, at most n [%u]\n"
	            "    -t INT    host threads of the reference; kept for compatibility [%u]\n\n"
	            "  extensions\n"
	            "    -d FILE   write the index image after construction\n"
	            "    -l FILE   load an index image (same -k -m -n -b) instead of reading -g\n",
	            o.number.at('k'), o.number.at('m'), o.number.at('n'), o.number.at('b'), o.number.at('s'), o.number.at('t'));
}

}  // namespace

int main(int argc, char** argv)
{
	Options opt;
	for (int c; (c = getopt(argc, argv, "g:q:d:l:k:m:n:s:t:b:")) != -1;) {
		if (opt.number.count((char)c))
			opt.number[(char)c] = (unsigned)std::stoul(optarg);
		else if (c != '?')
			opt.text[(char)c] = optarg;
	}
	std::string graph = opt.text['g'], reads = opt.text['q'];
	if (graph.empty())
		graph = reads;
	if (reads.empty())
		reads = graph;
	const unsigned k = opt.number['k'], m = opt.number['m'], n = opt.number['n'], s = opt.number['s'], t = opt.number['t'], b = opt.number['b'];
	if (graph.empty() || k == 0) {
		usage(opt);
		return 0;
	}
	try {
		std::cout << "I use -g " << graph << " -q " << reads << " -k " << k << " -m  " << m << " -n  " << n << " -s  " << s << " -t " << t
		          << " -b " << b << std::endl;
		kmer_Set_Light index(k, m, n, s, t, b);
		if (!opt.text['l'].empty())
			index.load(opt.text['l']);
		else
			index.construct_index(graph);
		if (!opt.text['d'].empty())
			index.dump_disk(opt.text['d']);
		index.file_query(reads);
		std::cout << "I am glad you are here with me. Here at the end of all things." << std::endl;
	} catch (const std::exception& e) {
		std::cerr << "error: " << e.what() << std::endl;
		return 1;
	}
	return 0;
}
