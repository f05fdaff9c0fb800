// payload_dbg.cpp -- the reference's two demo applications (Colored_De_Bruijn_graph_snippet.cpp, Abundance_De_Bruijn_graph_
// snippet.cpp) on the MI355X path: same getopt flags (-g graph, -o file of files, -q query, -k, -m, -n, -s, -t, -b), same
// output files ("outcolor" / "outabundance": one line per k-mer of every query line that starts with A/C/G/T; the colours /
// the abundance of a k-mer that is in the index followed by a tab, nothing for an alien k-mer).  -a selects the abundance
// application, the default is the coloured graph.  Host C++ only; all compute is in libblight_mi355x.so.
#include <unistd.h>

#include <chrono>
#include <fstream>
#include <iostream>
#include <string>

#include "blight.h"

using namespace std;
using namespace chrono;

int main(int argc, char** argv)
{
	int ch;
	string input, query, fof;
	uint k(0), m1(9), m2(17), m3(6), c(1), bit(6);
	bool abundance = false;
	while ((ch = getopt(argc, argv, "g:q:k:m:n:s:t:b:o:a")) != -1) {
		switch (ch) {
		case 'q': query = optarg; break;
		case 'g': input = optarg; break;
		case 'o': fof = optarg; break;
		case 'k': k = stoi(optarg); break;
		case 'm': m1 = stoi(optarg); break;
		case 'n': m2 = stoi(optarg); break;
		case 's': m3 = stoi(optarg); break;
		case 't': c = stoi(optarg); break;
		case 'b': bit = stoi(optarg); break;
		case 'a': abundance = true; break;
		}
	}
	if (query == "" or input == "" or fof == "" or k == 0) {
		cout << "Mandatory arguments" << endl
		     << "\t-g graph file constructed fom all your file" << endl
		     << "\t-o your original files in a file of file" << endl
		     << "\t-q query file" << endl
		     << "\t-k k value used for graph " << endl
		     << "\t-a abundance instead of colours" << endl;
		return 0;
	}
	try {
		kmer_Set_Light ksl(k, m1, m2, m3, c, bit);
		ksl.construct_index(input);
		const auto t1 = high_resolution_clock::now();
		ifstream fofin(fof);
		vector<string> file_names;
		string file_name;
		while (getline(fofin, file_name))
			if (not file_name.empty())
				file_names.push_back(file_name);
		const uint64_t color_number = file_names.size();
		kmer_Set_Light::payload pl(ksl, abundance ? kmer_Set_Light::payload::abundance : kmer_Set_Light::payload::colors, (uint32_t)color_number);
		for (uint i_file = 0; i_file < file_names.size(); ++i_file)
			pl.add_file(file_names[i_file], i_file);
		const vector<uint8_t> host = pl.to_host();
		const auto t12 = high_resolution_clock::now();
		cout << (abundance ? "Abundancy computed: " : "Coloration done: ") << duration_cast<duration<double>>(t12 - t1).count() << " seconds." << endl;
		ofstream out(abundance ? "outabundance" : "outcolor");
		ifstream query_file(query);
		string qline;
		while (getline(query_file, qline)) {
			if (qline.empty())
				break;  // like the reference: the first empty line ends the query file
			if (not(qline[0] == 'A' or qline[0] == 'C' or qline[0] == 'G' or qline[0] == 'T'))
				continue;
			string toWrite;
			const vector<int64_t> kmer_ids = ksl.query_sequence_hash(qline);
			for (uint64_t i = 0; i < kmer_ids.size(); ++i) {
				if (kmer_ids[i] >= 0) {
					if (abundance) {
						toWrite += to_string(host[(size_t)kmer_ids[i]]) + "\t";
					} else {
						for (uint64_t i_color = 0; i_color < color_number; ++i_color) {
							const uint64_t bitidx = (uint64_t)kmer_ids[i] * color_number + i_color;
							if (host[bitidx >> 3] >> (bitidx & 7) & 1)
								toWrite += to_string(i_color) + "\t";
						}
					}
				}
				toWrite += "\n";
			}
			out << toWrite;
		}
		cout << "Query done: " << duration_cast<duration<double>>(high_resolution_clock::now() - t12).count() << " seconds." << endl;
	} catch (const std::exception& e) {
		cerr << "error: " << e.what() << endl;
		return 1;
	}
	return 0;
}
