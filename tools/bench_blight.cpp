// bench_blight.cpp -- the reference's CLI harness (reference bench_blight.cpp:37-107) on top of the facade include/blight.h:
// same getopt flags (-g graph, -q query, -k, -m, -n, -s, -t, -b) and defaults (k=31 m=9 n=17 s=6 t=1 b=6), query file
// defaults to the graph file, builds the index and runs file_query.  Host C++ only; all compute is in libblight_mi355x.so.
//
//   g++ -O2 -std=c++17 -Iinclude tools/bench_blight.cpp -o tools/bench_blight -Lblight_amd -lblight_mi355x -lz
//       -Wl,-rpath,'$ORIGIN/../blight_amd'
#include <unistd.h>

#include <iostream>
#include <string>

#include "blight.h"

using namespace std;

int main(int argc, char** argv)
{
	int ch;
	string input, query, dump, load;
	uint k(31), m1(9), m2(17), m3(6), c(1), bit(6);
	while ((ch = getopt(argc, argv, "g:q:k:m:n:s:t:b:d:l:")) != -1) {
		switch (ch) {
		case 'q': query = optarg; break;
		case 'g': input = optarg; break;
		case 'k': k = stoi(optarg); break;
		case 'm': m1 = stoi(optarg); break;
		case 'n': m2 = stoi(optarg); break;
		case 's': m3 = stoi(optarg); break;
		case 't': c = stoi(optarg); break;
		case 'b': bit = stoi(optarg); break;
		case 'd': dump = optarg; break;  // extension: write the index image after construction
		case 'l': load = optarg; break;  // extension: load an index image instead of constructing
		}
	}
	if (query == "" and input != "")
		query = input;
	if (input == "" and query != "")
		input = query;
	if (query == "" or input == "" or k == 0) {
		cout << "Mandatory arguments" << endl
		     << "\t-g graph file" << endl
		     << "\t-q query file" << endl
		     << "\t-k k value used for graph (" << k << ")" << endl
		     << endl
		     << "Performances arguments" << endl
		     << "\t-m minimizer size (" << m1 << ")" << endl
		     << "\t-n to create 2^n mphf (2^" << m2 << "=" << (1u << m2) << "). More mean slower construction but better index, must be <=2*m-1" << endl
		     << "\t-s to use 2^s files (2^" << m3 << "=" << (1u << m3) << "). Accepted for compatibility (no temp files are used), must be <=n" << endl
		     << "\t-t core used (" << c << "). Accepted for compatibility (the GPU path has no host threads)" << endl
		     << "\t-d file: also write the index image (extension)" << endl
		     << "\t-l file: load an index image written with the same -k -m -n -b instead of building from -g (extension)" << endl
		     << "\t-b bit saved to encode positions (" << bit << "). Will reduce the memory usage of b bit per kmer but query have to check 2^b kmers" << endl;
		return 0;
	}
	try {
		cout << "I use -g " + input + " -q " + query + " -k " + to_string(k) + " -m  " + to_string(m1) + " -n  " + to_string(m2) + " -s  " + to_string(m3) +
		            " -t " + to_string(c) + " -b " + to_string(bit)
		     << endl;
		kmer_Set_Light ksl(k, m1, m2, m3, c, bit);
		if (load != "")
			ksl.load(load);
		else
			ksl.construct_index(input);
		if (dump != "")
			ksl.dump_disk(dump);
		ksl.file_query(query);
		cout << "I am glad you are here with me. Here at the end of all things." << endl;
	} catch (const std::exception& e) {
		cerr << "error: " << e.what() << endl;
		return 1;
	}
	return 0;
}
