// shard_query -- several GPUs from C++, no Python: one process per GPU, each builds its shard of the index from its part of the
// unitig file and answers its part of the reads through the native exchange (include/blight.h: kmer_Set_Light::communicator).
//
//   shard_query -g unitigs.fa -q reads.fa -k 31 -m 9 -n 7 -b 6 -r RANK -w WORLD -i ID_FILE [-d DEVICE]
//   shard_query -g unitigs.fa -q reads.fa -k 31 ... --transport loopback --world N [-d DEVICE]
//
// Rank 0 writes the communicator id to ID_FILE, the other ranks wait for it (any other way of handing 128 bytes over works as
// well).  Every rank prints one JSON line: its reads, k-mers, found / absent counts and an order-independent checksum over
// (read, position, ID) that a single-GPU run of the same reads reproduces.
// With --transport loopback the N ranks are N threads of THIS process on one GPU (the library's loopback transport: the same
// choreography, device copies instead of xGMI), for rehearsing the multi-rank path on a one-GPU box; N lines are printed.
#include <getopt.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <thread>

#include "blight.h"

namespace {

[[noreturn]] void usage(const char* why)
{
	std::fprintf(stderr, "%s\nusage: shard_query -g unitigs.fa -q reads -k K [-m M -n N -b B] -r rank -w world -i id_file [-d device]\n", why);
	std::exit(2);
}

std::vector<std::string> read_sequences(const std::string& path)
{
	std::ifstream in(path);
	if (!in)
		usage(("cannot open " + path).c_str());
	std::vector<std::string> out;
	for (std::string line; std::getline(in, line);)
		if (!line.empty() && line[0] != '>')
			out.push_back(line);
	return out;
}

kmer_Set_Light::communicator::id_t exchange_id(const std::string& file, uint32_t rank)
{
	kmer_Set_Light::communicator::id_t id;
	if (rank == 0) {
		id = kmer_Set_Light::communicator::unique_id();
		const std::string tmp = file + ".tmp";
		std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<const char*>(id.data()), id.size());
		if (std::rename(tmp.c_str(), file.c_str()) != 0)
			usage("cannot write the id file");
		return id;
	}
	for (int tries = 0; tries < 1200; tries++) {  // two minutes
		std::ifstream in(file, std::ios::binary);
		if (in && in.read(reinterpret_cast<char*>(id.data()), id.size()))
			return id;
		std::this_thread::sleep_for(std::chrono::milliseconds(100));
	}
	usage("no id file after two minutes");
}

// everything one rank does; returns its JSON line
std::string run_rank(const kmer_Set_Light::communicator::id_t& id, uint32_t rank, uint32_t world, std::map<char, std::string> opt)
{
	const uint k = (uint)std::stoul(opt['k']);
	kmer_Set_Light::communicator comm(id, rank, world, std::stoi(opt['d']));
	kmer_Set_Light ksl(k, (ksize_t)std::stoul(opt['m']), (bitsize_t)std::stoul(opt['n']), (bitsize_t)std::min(3ul, std::stoul(opt['n'])), 1,
	                   (uint)std::stoul(opt['b']), std::stoi(opt['d']));
	ksl.construct_index(opt['g'], comm);
	// reads: rank r answers the r-th consecutive part, in batches; every rank runs the same number of batches (collective calls)
	const std::vector<std::string> reads = read_sequences(opt['q']);
	const size_t lo = reads.size() * rank / world, hi = reads.size() * (rank + 1) / world;
	const size_t per_batch = 4096, batches = ((reads.size() + world - 1) / world + per_batch - 1) / per_batch + 1;
	uint64_t kmers = 0, found = 0, absent = 0, checksum = 0;
	for (size_t bt = 0; bt < batches; bt++) {
		const size_t a = std::min(hi, lo + bt * per_batch), z = std::min(hi, a + per_batch);
		const std::vector<std::string> batch(reads.begin() + (long)a, reads.begin() + (long)z);
		const std::vector<int64_t> ids = ksl.query_sequences_hash(batch, comm);
		size_t o = 0;
		for (size_t i = 0; i < batch.size(); i++) {
			const size_t nk = batch[i].size() >= k ? batch[i].size() - k + 1 : 0;
			for (size_t j = 0; j < nk; j++, o++) {
				(ids[o] >= 0 ? found : absent)++;
				checksum += (uint64_t)((a + i) * 1000003 + j + 1) * (uint64_t)(ids[o] + 2);
			}
		}
		kmers += o;
	}
	char line[512];
	std::snprintf(line, sizeof line,
	              "{\"rank\": %u, \"world\": %u, \"transport\": \"%s\", \"number_kmer\": %zu, \"reads\": %zu, \"kmers\": %llu, \"found\": %llu, "
	              "\"absent\": %llu, \"checksum\": %llu}",
	              rank, world, comm.transport().c_str(), (size_t)ksl.number_kmer, hi - lo, (unsigned long long)kmers, (unsigned long long)found,
	              (unsigned long long)absent, (unsigned long long)checksum);
	return line;
}

}  // namespace

int main(int argc, char** argv)
{
	std::map<char, std::string> opt = {{'m', "9"}, {'n', "7"}, {'b', "6"}, {'r', "0"}, {'w', "1"}, {'d', "0"}, {'t', "rccl"}};
	static const option long_opts[] = {{"transport", required_argument, nullptr, 't'}, {"world", required_argument, nullptr, 'w'}, {nullptr, 0, nullptr, 0}};
	for (int c; (c = getopt_long(argc, argv, "g:q:k:m:n:b:r:w:i:d:t:", long_opts, nullptr)) != -1;) {
		if (c == '?')
			usage("unknown option");
		opt[(char)c] = optarg;
	}
	const bool loopback = opt['t'] == "loopback";
	if (!loopback && opt['t'] != "rccl")
		usage("--transport is rccl or loopback");
	for (char need : {'g', 'q', 'k'})
		if (!opt.count(need))
			usage("missing option");
	if (!loopback && !opt.count('i'))
		usage("missing option -i");
	const uint32_t rank = (uint32_t)std::stoul(opt['r']), world = (uint32_t)std::stoul(opt['w']);
	try {
		if (!loopback) {
			std::puts(run_rank(exchange_id(opt['i'], rank), rank, world, opt).c_str());
			return 0;
		}
		const kmer_Set_Light::communicator::id_t id = kmer_Set_Light::communicator::loopback_id();
		std::vector<std::string> lines(world), errors(world);
		std::vector<std::thread> th;
		for (uint32_t r = 0; r < world; r++)
			th.emplace_back([&, r] {
				try {
					lines[r] = run_rank(id, r, world, opt);
				} catch (const std::exception& e) {
					errors[r] = e.what();
				}
			});
		for (auto& t : th)
			t.join();
		int rc = 0;
		for (uint32_t r = 0; r < world; r++) {
			if (!errors[r].empty()) {
				std::fprintf(stderr, "shard_query: rank %u: %s\n", r, errors[r].c_str());
				rc = 1;
			} else {
				std::puts(lines[r].c_str());
			}
		}
		return rc;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "shard_query: %s\n", e.what());
		return 1;
	}
}
