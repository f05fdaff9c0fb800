// blight.h -- source-compatible facade of the reference's `class kmer_Set_Light` (reference blight.h:15-136) over the
// MI355X C ABI (blight_mi355x.h / libblight_mi355x.so).  A program written against the reference -- bench_blight.cpp,
// the coloured / abundance de Bruijn graph snippets -- compiles against this header unchanged and links with
// -lblight_mi355x instead of blight.o.
//
// What is kept: the constructor signature and its std::invalid_argument messages (blight.h:62-92), construct_index,
// query_kmer_hash / query_kmer_bool / query_get_hash / single_query / query_sequence_hash / query_sequence_bool /
// file_query, the public counters callers read (number_kmer, number_super_kmer, number_query, largest_MPHF,
// positions_total_size, bit_per_kmer), and the typedefs kmer_t / minimizer_t / ksize_t / bitsize_t (kmer.h:15-19).
// What is not: the reference's internal containers (bucketSeq, positions, all_buckets, all_mphf: the index lives in
// HBM) and its declared-but-never-defined members.  Every query runs on the GPU; batch overloads (std::vector<kmer_t>)
// are added because a scalar call pays one kernel launch per k-mer.
//
// Define BLIGHT_KMER_128 before including this header for k > 32 (the reference's `kmer_t = __uint128_t` switch).
#ifndef BLIGHT_MI355X_FACADE_H
#define BLIGHT_MI355X_FACADE_H

#include <array>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "blight_mi355x.h"

using nuc_t = uint_fast8_t;      // kmer.h:15
using bitsize_t = uint_fast8_t;  // kmer.h:16
using ksize_t = bitsize_t;       // kmer.h:17
#ifdef BLIGHT_KMER_128
using kmer_t = __uint128_t;  // kmer.h:18 (k <= 63)
#else
using kmer_t = uint64_t;  // kmer.h:18 (k <= 32)
#endif
using minimizer_t = uint32_t;  // kmer.h:19
#ifndef _GNU_SOURCE
typedef unsigned int uint;
#endif

class kmer_Set_Light {
public:
	const ksize_t _k, _minimizer_length;
	const ksize_t _core_number;
	const bitsize_t bit_saved_sub;

	size_t number_kmer = 0;
	size_t number_super_kmer = 0;
	size_t largest_MPHF = 0;
	size_t positions_total_size = 0;
	std::atomic<size_t> number_query{0};  // incremented atomically like the reference (blight.cpp:687-688): callers query from OpenMP threads
	double bit_per_kmer = 0;
	uint largest_bucket_nuc_all = 0;
	const uint gammaFactor = 2;

	// blight.h:62 -- same argument order and meaning; std::invalid_argument with the reference's messages
	kmer_Set_Light(uint k, ksize_t minimizer_length, bitsize_t log2_mphfs_number, bitsize_t log2_superbuckets_number, uint cores_number,
	               uint bits_to_save, int device = 0)
	    : _k((ksize_t)k), _minimizer_length(minimizer_length), _core_number((ksize_t)cores_number), bit_saved_sub((bitsize_t)bits_to_save)
	{
		if (blmx_abi_version() != BLMX_ABI_VERSION)
			throw std::runtime_error("libblight_mi355x.so was built from a different blight_mi355x.h (ABI version mismatch): rebuild the caller");
		blmx_params p;
		p.k = k;
		p.m = minimizer_length;
		p.log2_mphf = log2_mphfs_number;
		p.log2_superbuckets = log2_superbuckets_number;
		p.threads = cores_number;
		p.bits_saved = bits_to_save;
		p.kmer_bytes = (uint32_t)sizeof(kmer_t);
		p.device = device;
		const int rc = blmx_create(&p, &_h);
		if (rc == BLMX_EINVAL)
			throw std::invalid_argument(blmx_create_error());
		if (rc != BLMX_OK)
			throw std::runtime_error(blmx_create_error());
	}
	kmer_Set_Light(const kmer_Set_Light&) = delete;
	kmer_Set_Light& operator=(const kmer_Set_Light&) = delete;
	~kmer_Set_Light()
	{
		if (_owned)
			blmx_destroy(_h);
	}
	// Facade over an index that was created (and stays owned) by a caller of the C ABI, e.g. another language binding.
	static kmer_Set_Light* attach(blmx_index* h)
	{
		blmx_stats_t st;
		if (blmx_abi_version() != BLMX_ABI_VERSION || blmx_stats(h, &st) != BLMX_OK)
			throw std::runtime_error("cannot attach to this index handle");
		if (st.kmer_bytes != sizeof(kmer_t))
			throw std::invalid_argument("index was created with another kmer_t width (BLIGHT_KMER_128)");
		kmer_Set_Light* f = new kmer_Set_Light(h, st);
		f->refresh();
		return f;
	}

	// blight.cpp:108-125 (prints the reference's timing lines and index recap, blight.cpp:361-378)
	void construct_index(const std::string& input_file)
	{
		const auto t1 = std::chrono::high_resolution_clock::now();
		check(blmx_build_from_fasta(_h, input_file.c_str()));
		refresh();
		const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t1).count();
		std::cout << "Indexes created: " << secs << " seconds." << std::endl;
		std::cout << "The whole indexing took me " << secs << " seconds." << std::endl;
		std::cout << "----------------------INDEX RECAP----------------------------" << std::endl;
		std::cout << "Kmer in graph: " << intToString(number_kmer) << std::endl;
		std::cout << "Super Kmer in graph: " << intToString(number_super_kmer) << std::endl;
		if (number_super_kmer)
			std::cout << "Average size of Super Kmer: " << intToString(number_kmer / number_super_kmer) << std::endl;
		std::cout << "Total size of the partitionned graph: " << intToString(_st.total_nt) << std::endl;
		std::cout << "Largest MPHF: " << intToString(largest_MPHF) << std::endl;
		std::cout << "Largest Bucket: " << intToString(largest_bucket_nuc_all) << std::endl;
		if (number_kmer) {
			std::cout << "Size of the partitionned graph (bit per kmer): " << (double)(2 * _st.total_nt) / number_kmer << std::endl;
			std::cout << "Total Positions size (bit per kmer): " << (double)positions_total_size / number_kmer << std::endl;
			std::cout << "TOTAL Bits per kmer (without bbhash): " << bit_per_kmer << std::endl;
			std::cout << "MPHF size in HBM incl. ranks (bit per kmer): " << (double)(8 * _st.mphf_bytes) / number_kmer << std::endl;
		}
	}

	// Index image on disk.  Absent from this fork of the reference; named after mainline Blight's pair.
	// load() needs an object constructed with the parameters the image was written with (std::invalid_argument otherwise).
	void dump_disk(const std::string& output_file) { check(blmx_save(_h, output_file.c_str())); }
	void load(const std::string& index_file)
	{
		check(blmx_load(_h, index_file.c_str()));
		refresh();
	}

	// blight.cpp:545-550
	int64_t query_kmer_hash(kmer_t canon)
	{
		int64_t id = -1;
		check(blmx_query_kmers(_h, &canon, 1, &id));
		number_query += 1;
		return id;
	}
	std::vector<int64_t> query_kmer_hash(const std::vector<kmer_t>& canon)
	{
		std::vector<int64_t> ids(canon.size());
		check(blmx_query_kmers(_h, canon.data(), canon.size(), ids.data()));
		number_query += canon.size();
		return ids;
	}
	// blight.cpp:537-541
	bool query_kmer_bool(kmer_t canon) { return query_kmer_hash(canon) >= 0; }
	// blight.cpp:716-742
	int64_t query_get_hash(const kmer_t canon, uint32_t minimizer)
	{
		int64_t id = -1;
		check(blmx_query_kmers_with_minimizer(_h, &canon, &minimizer, 1, &id));
		number_query += 1;
		return id;
	}
	// blight.cpp:686-712
	int32_t query_get_pos_unitig(const kmer_t canon, minimizer_t minimizer)
	{
		int32_t pos = -1;
		check(blmx_query_positions(_h, &canon, &minimizer, 1, &pos));
		number_query += 1;
		return pos;
	}
	// blight.cpp:658-660
	bool single_query(minimizer_t minimizer, kmer_t kastor) { return query_get_pos_unitig(kastor, minimizer) >= 0; }

	// blight.cpp:575-591
	std::vector<int64_t> query_sequence_hash(const std::string& query)
	{
		const uint64_t offs[2] = {0, query.size()};
		std::vector<int64_t> res(blmx_sequence_kmer_count(_h, offs, 1));
		check(blmx_query_sequences(_h, query.data(), offs, 1, res.data()));
		number_query += res.size();
		return res;
	}
	// many sequences in one launch (ids of sequence i follow those of sequence i-1)
	std::vector<int64_t> query_sequences_hash(const std::vector<std::string>& queries)
	{
		std::string concat;
		std::vector<uint64_t> offs(1, 0);
		for (const auto& q : queries) {
			concat += q;
			offs.push_back(concat.size());
		}
		std::vector<int64_t> res(blmx_sequence_kmer_count(_h, offs.data(), queries.size()));
		check(blmx_query_sequences(_h, concat.data(), offs.data(), queries.size(), res.data()));
		number_query += res.size();
		return res;
	}
	// blight.cpp:554-571 -> (found, not found)
	std::pair<uint32_t, uint32_t> query_sequence_bool(const std::string& query)
	{
		uint32_t res = 0, fail = 0;
		for (int64_t id : query_sequence_hash(query))
			(id >= 0 ? res : fail)++;
		return std::make_pair(res, fail);
	}

	// blight.cpp:746-799: streamed through the GPU by blmx_query_file (sequences shorter than k are skipped, :782);
	// prints "QUERY RECAP 2"
	void file_query(const std::string& query_file)
	{
		const auto t1 = std::chrono::high_resolution_clock::now();
		uint64_t TP = 0, FP = 0;
		check(blmx_query_file(_h, query_file.c_str(), &TP, &FP));
		number_query += TP + FP;
		std::cout << "-----------------------QUERY RECAP 2----------------------------" << std::endl;
		std::cout << "Good kmer: " << intToString(TP) << std::endl;
		std::cout << "Erroneous kmers: " << intToString(FP) << std::endl;
		std::cout << "Query performed: " << intToString(number_query.load()) << std::endl;
		const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t1).count();
		std::cout << "The whole QUERY took me " << secs << " seconds." << std::endl;
	}

	// ---- several GPUs, one process per GPU (no counterpart in the reference, which is one shared-memory process): the MPHF
	// groups are split over the ranks by key count, IDs stay those of the single index.  Rank 0 makes an id and hands it to the
	// others by any means (a file, MPI_Bcast, a socket); every rank then creates its communicator (collective).
	class communicator {
	public:
		typedef std::array<unsigned char, BLMX_COMM_ID_BYTES> id_t;
		static id_t unique_id()
		{
			id_t id;
			if (blmx_comm_unique_id(id.data()) != BLMX_OK)
				throw std::runtime_error(blmx_last_error(nullptr));
			return id;
		}
		// an id for `world` communicators inside ONE process (one thread per rank, all on one GPU): the loopback transport, for
		// rehearsing the multi-rank path on a one-GPU box
		static id_t loopback_id()
		{
			id_t id;
			if (blmx_comm_loopback_id(id.data()) != BLMX_OK)
				throw std::runtime_error(blmx_last_error(nullptr));
			return id;
		}
		communicator(const id_t& id, uint32_t rank_, uint32_t world_, int device = 0) : rank(rank_), world(world_)
		{
			if (blmx_comm_create(id.data(), rank, world, device, &_c) != BLMX_OK)
				throw std::runtime_error(blmx_last_error(nullptr));
		}
		communicator(const communicator&) = delete;
		communicator& operator=(const communicator&) = delete;
		~communicator() { blmx_comm_destroy(_c); }
		blmx_comm* handle() const { return _c; }
		std::string transport() const { return blmx_comm_transport(_c); }
		const uint32_t rank, world;

	private:
		blmx_comm* _c = nullptr;
	};
	// construct_index over several GPUs (collective).  File: every rank names the SAME file and indexes the rank-th consecutive part
	// of its sequences.  Sequences: every rank passes ITS OWN consecutive part (rank order = input order).
	void construct_index(const std::string& input_file, communicator& c)
	{
		check(blmx_set_shard(_h, c.rank, c.world));
		check(blmx_build_sharded_from_fasta(_h, c.handle(), input_file.c_str()));
		refresh();
	}
	void construct_index(const std::vector<std::string>& my_sequences, communicator& c)
	{
		std::string concat;
		std::vector<uint64_t> offs(1, 0);
		for (const auto& q : my_sequences) {
			concat += q;
			offs.push_back(concat.size());
		}
		check(blmx_set_shard(_h, c.rank, c.world));
		check(blmx_build_sharded_from_sequences(_h, c.handle(), concat.data(), offs.data(), my_sequences.size()));
		refresh();
	}
	// queries against the sharded index (collective: every rank calls with its own batch, possibly empty); every k-mer is answered
	// by the rank that owns its MPHF group and the IDs come back in the caller's order
	std::vector<int64_t> query_kmer_hash(const std::vector<kmer_t>& canon, communicator& c)
	{
		std::vector<int64_t> ids(canon.size());
		check(blmx_query_kmers_sharded(_h, c.handle(), canon.data(), canon.size(), ids.data()));
		number_query += canon.size();
		return ids;
	}
	std::vector<int64_t> query_sequences_hash(const std::vector<std::string>& queries, communicator& c)
	{
		std::string concat;
		std::vector<uint64_t> offs(1, 0);
		for (const auto& q : queries) {
			concat += q;
			offs.push_back(concat.size());
		}
		std::vector<int64_t> res(blmx_sequence_kmer_count(_h, offs.data(), queries.size()));
		check(blmx_query_sequences_sharded(_h, c.handle(), concat.data(), offs.data(), queries.size(), res.data()));
		number_query += res.size();
		return res;
	}

	blmx_index* handle() const { return _h; }

	// Payload applications (reference Colored_/Abundance_De_Bruijn_graph_snippet.cpp): a per-k-mer array in HBM keyed by the
	// IDs of this index, updated on the device while a file streams through the query pipeline.
	class payload {
	public:
		enum kind_t { abundance = 0, colors = 1 };
		payload(kmer_Set_Light& ksl, kind_t kind, uint32_t ncolors = 1) : _ksl(ksl), _ncolors(kind == abundance ? 1 : ncolors)
		{
			_ksl.check(blmx_payload_create(_ksl._h, (int)kind, _ncolors, &_p));
		}
		payload(const payload&) = delete;
		payload& operator=(const payload&) = delete;
		~payload() { blmx_payload_destroy(_p); }
		// every line of the file that starts with A/C/G/T is a sequence (snippet :139); returns k-mers found in the index
		uint64_t add_file(const std::string& path, uint32_t color = 0)
		{
			uint64_t found = 0, absent = 0;
			_ksl.check(blmx_payload_add_file(_p, path.c_str(), color, &found, &absent));
			_ksl.number_query += found + absent;
			return found;
		}
		// abundance: one byte per k-mer ID; colours: bit (id * ncolors + color), least significant bit first
		std::vector<uint8_t> to_host()
		{
			const int64_t n = blmx_payload_export(_p, nullptr, 0);
			if (n < 0)
				_ksl.check((int)n);
			std::vector<uint8_t> v((size_t)n);
			blmx_payload_export(_p, v.data(), (uint64_t)n);
			return v;
		}
		uint32_t ncolors() const { return _ncolors; }

	private:
		kmer_Set_Light& _ksl;
		blmx_payload* _p = nullptr;
		uint32_t _ncolors;
	};

private:
	blmx_index* _h = nullptr;
	blmx_stats_t _st{};
	bool _owned = true;

	kmer_Set_Light(blmx_index* h, const blmx_stats_t& st)
	    : _k((ksize_t)st.k), _minimizer_length((ksize_t)st.m), _core_number(1), bit_saved_sub((bitsize_t)st.bits_saved), _h(h), _owned(false)
	{
	}

	void check(int rc)
	{
		if (rc == BLMX_OK)
			return;
		const std::string msg = blmx_last_error(_h);
		if (rc == BLMX_EINVAL)
			throw std::invalid_argument(msg);
		if (rc == BLMX_EDNA)
			throw std::domain_error("Invalid char in DNA");  // kmer.h:68
		throw std::runtime_error(msg.empty() ? "blight_mi355x error" : msg);
	}
	void refresh()
	{
		blmx_stats(_h, &_st);
		number_kmer = _st.number_kmer;
		number_super_kmer = _st.number_super_kmer;
		largest_MPHF = _st.largest_mphf;
		positions_total_size = _st.positions_total_size;
		largest_bucket_nuc_all = (uint)_st.largest_bucket;
		bit_per_kmer = number_kmer ? ((double)(2 * _st.total_nt) + (double)positions_total_size) / number_kmer : 0;  // blight.cpp:373-375
	}
	static std::string intToString(uint64_t n)  // blight.cpp:35-47: thousands separators
	{
		if (n < 1000)
			return std::to_string(n);
		std::string end(std::to_string(n % 1000));
		if (end.size() == 3)
			return intToString(n / 1000) + "," + end;
		if (end.size() == 2)
			return intToString(n / 1000) + ",0" + end;
		return intToString(n / 1000) + ",00" + end;
	}
};

#endif  // BLIGHT_MI355X_FACADE_H
