// payload_dbg.cpp -- coloured / abundance de Bruijn graph annotation on the MI355X path.
//
// Same job and same files as the reference's two demo programs (Colored_De_Bruijn_graph_snippet.cpp,
// Abundance_De_Bruijn_graph_snippet.cpp): index the unitigs of -g, stream every file listed in -o through the index while a
// per-k-mer payload is updated (one colour bit per listed file, or a wrapping 8-bit occurrence count with -a), then annotate
// the k-mers of -q into "outcolor" / "outabundance" (one output line per k-mer of every query line that starts with A/C/G/T:
// the colours / the count followed by a tab when the k-mer is indexed, an empty line otherwise; the first empty query line
// ends the file).  The payload lives in HBM and is updated by the GPU while the files stream (kmer_Set_Light::payload);
// host C++ only, all compute is in libblight_mi355x.so.
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "blight.h"

namespace {

struct Options {
	std::map<char, std::string> text;  // -g -o -q
	std::map<char, unsigned> number{{'k', 0}, {'m', 9}, {'n', 17}, {'s', 6}, {'t', 1}, {'b', 6}};
	bool abundance = false;
};

void usage()
{
	std::printf("payload_dbg (MI355X)  --  colour or count the k-mers of a de Bruijn graph\n\n"
	            "  required\n"
	            "    -g FILE   unitig FASTA of the union graph\n"
	            "    -o FILE   list of the original sequence files, one path per line (one colour each)\n"
	            "    -q FILE   sequences to annotate\n"
	            "    -k INT    k-mer length of the graph\n\n"
	            "  optional\n"
	            "    -a        abundances (8-bit, wrapping) instead of colours\n"
	            "    -m -n -s -t -b   as bench_blight [9 17 6 1 6]\n");
}

std::vector<std::string> non_empty_lines(const std::string& path)
{
	std::vector<std::string> out;
	std::ifstream in(path);
	for (std::string line; std::getline(in, line);)
		if (!line.empty())
			out.push_back(line);
	return out;
}

bool starts_with_nucleotide(const std::string& s) { return !s.empty() && (s[0] == 'A' || s[0] == 'C' || s[0] == 'G' || s[0] == 'T'); }

double seconds_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }

}  // namespace

int main(int argc, char** argv)
{
	Options opt;
	for (int c; (c = getopt(argc, argv, "g:o:q:k:m:n:s:t:b:a")) != -1;) {
		if (c == 'a')
			opt.abundance = true;
		else if (opt.number.count((char)c))
			opt.number[(char)c] = (unsigned)std::stoul(optarg);
		else if (c != '?')
			opt.text[(char)c] = optarg;
	}
	if (opt.text['g'].empty() || opt.text['o'].empty() || opt.text['q'].empty() || opt.number['k'] == 0) {
		usage();
		return 0;
	}
	try {
		kmer_Set_Light index(opt.number['k'], opt.number['m'], opt.number['n'], opt.number['s'], opt.number['t'], opt.number['b']);
		index.construct_index(opt.text['g']);

		auto t0 = std::chrono::steady_clock::now();
		const std::vector<std::string> sources = non_empty_lines(opt.text['o']);
		const uint64_t ncolors = sources.size();
		using payload = kmer_Set_Light::payload;
		payload annotation(index, opt.abundance ? payload::abundance : payload::colors, (uint32_t)ncolors);
		for (uint32_t color = 0; color < ncolors; color++)
			annotation.add_file(sources[color], color);
		const std::vector<uint8_t> table = annotation.to_host();
		std::cout << (opt.abundance ? "Abundancy computed: " : "Coloration done: ") << seconds_since(t0) << " seconds." << std::endl;

		t0 = std::chrono::steady_clock::now();
		std::ofstream out(opt.abundance ? "outabundance" : "outcolor");
		std::ifstream queries(opt.text['q']);
		for (std::string line; std::getline(queries, line) && !line.empty();) {
			if (!starts_with_nucleotide(line))
				continue;
			std::string block;
			for (const int64_t id : index.query_sequence_hash(line)) {
				if (id >= 0 && opt.abundance) {
					block += std::to_string(table[(size_t)id]) + "\t";
				} else if (id >= 0) {
					const uint64_t first = (uint64_t)id * ncolors;
					for (uint64_t c = 0; c < ncolors; c++)
						if (table[(first + c) >> 3] >> ((first + c) & 7) & 1)
							block += std::to_string(c) + "\t";
				}
				block += "\n";
			}
			out << block;
		}
		std::cout << "Query done: " << seconds_since(t0) << " seconds." << std::endl;
	} catch (const std::exception& e) {
		std::cerr << "error: " << e.what() << std::endl;
		return 1;
	}
	return 0;
}
