/*
 * blight_mi355x.h -- C ABI of the MI355X-native Blight hot path (libblight_mi355x.so).
 *
 * This is the drop-in boundary: plain C types, caller-owned host buffers (or raw device pointers for the *_device
 * entry points), no exceptions, no C++ or torch types.  The reference has no FFI layer -- its boundary is the public
 * surface of `class kmer_Set_Light` (reference blight.h:15-136); each entry point below names the member it replaces.
 * The source-compatible C++ facade that calls this ABI is include/blight.h.
 *
 * Semantics are those of the reference with the fixes P1/P2 of SURVEY.md App. B-1 (the shipped reference crashes in
 * its query path and mis-encodes unitig starts), at -t 1; results (IDs, membership bits, positions) are bit-identical.
 * All compute runs in hand-written HIP kernels on gfx950; there is no CPU fallback: every entry point that needs the
 * GPU fails with BLMX_EHIP when no device is usable.
 */
#ifndef BLIGHT_MI355X_H
#define BLIGHT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header changes layout; blmx_abi_version() returns the value the library was built with,
 * so a stale caller binary is detected instead of corrupting memory. */
#define BLMX_ABI_VERSION 11

#define BLMX_OK 0
#define BLMX_EINVAL (-1)  /* bad parameter: std::invalid_argument in the reference ctor, blight.h:75-92            */
#define BLMX_ESHORT (-2)  /* a sequence is shorter than k: UB/abort in the reference (blight.cpp:235, kmer.h:705)  */
#define BLMX_EDNA (-3)    /* character outside ACGTacgt: std::domain_error("Invalid char in DNA"), kmer.h:68       */
#define BLMX_EIO (-4)     /* file cannot be opened/read: std::runtime_error, blight.cpp:188-189                    */
#define BLMX_EHIP (-5)    /* HIP runtime error / no usable GPU                                                     */
#define BLMX_ENOMEM (-6)  /* host or device allocation failed                                                      */
#define BLMX_ESTATE (-7)  /* call order: query before build, build twice                                           */
#define BLMX_ERCCL (-9)   /* RCCL cannot be loaded or a collective failed (multi-GPU entry points only)                 */
#define BLMX_ELIMIT (-8)  /* input exceeds an implementation limit (e.g. a bucket of >= 2^32 nt, as in the reference) */

typedef struct blmx_index blmx_index;

/* kmer_Set_Light(k, minimizer_length, log2_mphfs_number, log2_superbuckets_number, cores_number, bits_to_save),
 * reference blight.h:62.  log2_superbuckets / threads are validated like the reference and otherwise unused (no temp
 * files, no host threads).  kmer_bytes selects the k-mer word: 8 = uint64_t (k <= 32, reference default kmer.h:18),
 * 16 = __uint128_t (k <= 63, the reference's typedef switch); 0 = choose from k.  device = HIP device ordinal. */
typedef struct blmx_params {
	uint32_t k;
	uint32_t m;
	uint32_t log2_mphf;
	uint32_t log2_superbuckets;
	uint32_t threads;
	uint32_t bits_saved;
	uint32_t kmer_bytes;
	int32_t device;
} blmx_params;

/* Public counters of kmer_Set_Light (reference blight.h:52-59) + array sizes of the recap (blight.cpp:361-378). */
typedef struct blmx_stats_t {
	uint64_t number_kmer;          /* blight.h:52 */
	uint64_t number_super_kmer;    /* blight.h:53 */
	uint64_t largest_mphf;         /* blight.h:54 largest_MPHF */
	uint64_t positions_total_size; /* positions.size() in bits, blight.cpp:305 */
	uint64_t number_query;         /* blight.h:56: incremented once per k-mer query */
	uint64_t total_nt;             /* bucketSeq.size() / 2, blight.cpp:280 */
	uint64_t largest_bucket;       /* largest_bucket_nuc_all, blight.h:59 */
	uint64_t mphf_bytes;           /* MPHF lines in HBM (bit arrays + interleaved ranks) */
	uint64_t fallback_keys;        /* entries of the per-group fallback tables (_final_hash, bbhash.h:781) */
	uint64_t device_bytes;         /* total HBM held by the finished index */
	uint32_t nbuckets;             /* 2^(2m-1), blight.h:70 */
	uint32_t ngroups;              /* 2^log2_mphf, blight.h:68 */
	uint32_t k, m, log2_mphf, bits_saved, kmer_bytes;
	uint32_t max_levels_used;
	double build_seconds;          /* wall time of the last build (device work + host orchestration) */
	/* sharding (blmx_set_shard): this index holds the MPHF groups [shard_group_lo, shard_group_hi) only; number_kmer stays
	 * the GLOBAL count (IDs are global), local_number_kmer is what this shard holds */
	uint64_t local_number_kmer;
	uint32_t shard_rank, shard_world, shard_group_lo, shard_group_hi;
	uint32_t mphf_groups_lds_built; /* MPHF groups built by the LDS-staged kernel (the rest: global level kernels) */
	uint32_t reserved0;
	uint64_t build_scratch_peak_bytes; /* HBM the last build held for its temporaries at its peak (next to the finished arrays and the
	                                    * caller's input); given back when the build -- and the filter's background passes -- are done */
} blmx_stats_t;

/* ctor / dtor.  On failure *out is NULL and blmx_create_error() returns the reference's exception text
 * ("kmer size too large", "minimizer_length must be odd", ... blight.h:77-92). */
int blmx_create(const blmx_params* p, blmx_index** out);
void blmx_destroy(blmx_index* ix);
const char* blmx_create_error(void);
const char* blmx_last_error(const blmx_index* ix);

/* construct_index(file), reference blight.cpp:108-125: 2-line FASTA records (header line skipped, next line =
 * sequence; blight.cpp:212-229), plain or gzip.  Every sequence must hold >= k nucleotides of ACGTacgt. */
int blmx_build_from_fasta(blmx_index* ix, const char* path);
/* Same, from memory: nseq sequences concatenated in `concat`, sequence i = concat[offsets[i] .. offsets[i+1]). */
int blmx_build_from_sequences(blmx_index* ix, const char* concat, const uint64_t* offsets, uint64_t nseq);
/* Same, input already resident in HBM (device pointers: ASCII bytes and nseq+1 offsets).  Used for large synthetic
 * inputs generated on the device.  Synchronous; the device is synchronised first, so inputs produced on any stream are seen. */
int blmx_build_from_device_sequences(blmx_index* ix, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq,
                                     uint64_t total_len);

/* query_kmer_hash(kmer_t canon), reference blight.cpp:545-550, for a batch: canon_kmers = nq little-endian words of
 * kmer_bytes each (already canonical, as the reference requires); ids_out[i] = ID in [0, number_kmer) or -1. */
int blmx_query_kmers(blmx_index* ix, const void* canon_kmers, uint64_t nq, int64_t* ids_out);
/* query_get_hash(canon, minimizer), reference blight.cpp:716-742 (caller-supplied minimizers; minimizers outside
 * [0, 2^(2m-1)) give -1 where the reference reads out of bounds). */
int blmx_query_kmers_with_minimizer(blmx_index* ix, const void* canon_kmers, const uint32_t* minimizers, uint64_t nq,
                                    int64_t* ids_out);
/* query_get_pos_unitig(canon, minimizer), reference blight.cpp:686-712: bucket-relative position of the FIRST match
 * in the 2^b window, or -1. */
int blmx_query_positions(blmx_index* ix, const void* canon_kmers, const uint32_t* minimizers, uint64_t nq,
                         int32_t* pos_out);
/* query_kmer_bool, reference blight.cpp:537-541: present_out[i] = 1/0. */
int blmx_query_membership(blmx_index* ix, const void* canon_kmers, uint64_t nq, uint8_t* present_out);
/* query_sequence_hash for nseq sequences (reference blight.cpp:575-591): canonical k-mers and their minimizers are made on
 * the GPU by a wave-cooperative front end (one m-mer hash per position, window minimum through LDS).  ids_out holds
 * sum_i max(0, len_i - k + 1) entries, sequence after sequence; sequences shorter than k contribute nothing.  Returns
 * BLMX_EDNA on an invalid character.  Thread-safe: calls on a few short sequences (one read) run concurrently, each as ONE
 * kernel launch on a pinned slot the device reads and writes directly; large inputs stream through pinned windows. */
int blmx_query_sequences(blmx_index* ix, const char* concat, const uint64_t* offsets, uint64_t nseq, int64_t* ids_out);
/* file_query(file), reference blight.cpp:746-799: the RAW bytes of `path` (plain or gzip) stream through the GPU in pinned
 * windows; lines, the reference's record rule (lines are consumed in pairs; the second line of a pair is a sequence iff both
 * are non-empty, blight.cpp:760-775) and k-mer starts are found on the device; sequences shorter than k are skipped
 * (blight.cpp:782).  Returns the recap counters "Good kmer" / "Erroneous kmers"; only the two counters cross PCIe back. */
int blmx_query_file(blmx_index* ix, const char* path, uint64_t* good_out, uint64_t* erroneous_out);
/* Index image on disk.  This fork of the reference cannot persist an index (README: "no dump/load"); mainline
 * Blight's dump_disk(file) / kmer_Set_Light(file) pair is the model.  blmx_save writes the finished HBM arrays verbatim
 * (buckets, group table, 2-bit sequences, positions, MPHF lines, fallback table) behind a header carrying the parameters
 * and public counters; blmx_load fills an index created with the SAME (k, m, log2_mphf, bits_saved, kmer_bytes) -- any
 * difference is BLMX_EINVAL -- and leaves it in the built state; a truncated or foreign file is BLMX_EIO.
 * blmx_peek_params reads just the parameters, so that a caller can blmx_create the right object first. */
int blmx_save(blmx_index* ix, const char* path);
int blmx_load(blmx_index* ix, const char* path);
int blmx_peek_params(const char* path, blmx_params* out);
/* Payload applications (reference Colored_De_Bruijn_graph_snippet.cpp:113-157 and
 * Abundance_De_Bruijn_graph_snippet.cpp:117-157): a per-k-mer payload in HBM keyed by the IDs of this index.
 *   kind 0 = abundance: one uint8 counter per k-mer, ++ per occurrence, wrapping at 256 like the reference's uint8_t;
 *   kind 1 = colours:   ncolors bits per k-mer; bit (id * ncolors + color) in the reference's vector<bool> order.
 * blmx_payload_add_file streams `path` (plain or gzip) through the GPU like blmx_query_file, but -- like the snippets --
 * takes every LINE that starts with A/C/G/T as a sequence; lines shorter than k contribute nothing; the update is done
 * on the device (no ID crosses PCIe).  found_out / absent_out (may be NULL) = k-mers with ID >= 0 / < 0.
 * blmx_payload_add_ids_device applies IDs that are already in HBM (asynchronous on `stream`).
 * blmx_payload_export copies the payload to the host (abundance: number_kmer bytes; colours: ceil(number_kmer*ncolors/8)
 * bytes, least significant bit first) and returns its size in bytes. */
typedef struct blmx_payload blmx_payload;
int blmx_payload_create(blmx_index* ix, int kind, uint32_t ncolors, blmx_payload** out);
void blmx_payload_destroy(blmx_payload* p);
int blmx_payload_add_file(blmx_payload* p, const char* path, uint32_t color, uint64_t* found_out, uint64_t* absent_out);
int blmx_payload_add_ids_device(blmx_payload* p, const int64_t* d_ids, uint64_t n, uint32_t color, void* stream);
int64_t blmx_payload_export(blmx_payload* p, void* out, uint64_t cap_bytes);
/* number of IDs blmx_query_sequences writes for these offsets */
uint64_t blmx_sequence_kmer_count(const blmx_index* ix, const uint64_t* offsets, uint64_t nseq);

/* Device-resident entry points (all *_device functions): pointers are DEVICE pointers and `stream` is the hipStream_t the
 * work is enqueued on -- NULL is HIP's null stream, treated like any other stream.  The calls are asynchronous: they
 * return after enqueueing and are ordered with the caller's other work on that stream; the caller synchronises.
 * Calls on one index from several streams are serialised internally where they share scratch memory. */
int blmx_query_kmers_device(blmx_index* ix, const void* d_canon_kmers, uint64_t nq, int64_t* d_ids_out, void* stream);
/* query_sequence_hash for sequences that are already in HBM: d_concat = concatenated characters, d_offsets = nseq + 1 offsets
 * (offsets[0] = 0), total_len = offsets[nseq] < 2^32, nk = sum_i max(0, len_i - k + 1) as the caller computed it; d_ids gets nk
 * IDs, sequence after sequence.  *d_err_flag (device int, may be NULL) is set to 1 on a character outside ACGTacgt. */
int blmx_query_sequences_device(blmx_index* ix, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq, uint64_t total_len, uint64_t nk,
                                int64_t* d_ids, int* d_err_flag, void* stream);
/* Stage timing (HIP events on the caller's stream) of the last blmx_query_kmers_device / blmx_query_sequences_device call:
 * after blmx_set_profiling(ix, 1) every such call records events at its stage boundaries; blmx_last_profile waits for the last
 * one and writes 6 durations in milliseconds: [0] read front end (VALID bitmap, scan, k_read_front) | [1] k_bin_hist |
 * [2] scan | [3] k_bin_scatter | [4] k_query (or the fused k_read_query) | [5] k_gather_ids; stages that did not run are 0.
 * Returns the number of values written (6), 0 when nothing was recorded. */
int blmx_set_profiling(blmx_index* ix, int on);
int blmx_last_profile(blmx_index* ix, float* ms_out, int cap);
/* Minimizer -> owning MPHF group for a batch of canonical k-mers on the device (group = minimizer >> (2m-1-n),
 * reference blight.cpp:692); used by the multi-GPU router to pick the destination rank. */
int blmx_kmer_groups_device(blmx_index* ix, const void* d_canon_kmers, uint64_t nq, uint32_t* d_groups_out, void* stream);

/* Device utilities for callers that keep their data in HBM (bench, pipelines); all pointers are DEVICE pointers.
 * blmx_kmers_at_device: canonical k-mer (str2num + min with rcb, reference kmer.h:90-98, blight.cpp:86-91) starting at
 * each given byte position of an ASCII buffer; the caller guarantees position + k <= end of its sequence.  *d_err_flag
 * (int, may be NULL) is OR-ed with 1 on a character outside ACGTacgt.
 * blmx_canonicalize_device: in place, x -> min(x mod 4^k, rcb(x mod 4^k)) for arbitrary words (random negatives). */
int blmx_kmers_at_device(blmx_index* ix, const void* d_ascii, const uint64_t* d_positions, uint64_t nq, void* d_kmers_out,
                         int* d_err_flag, void* stream);
int blmx_canonicalize_device(blmx_index* ix, void* d_kmers, uint64_t nq, void* stream);

/* ---- multi-GPU sharding: ownership unit = MPHF group (a contiguous range of minimizer buckets, reference
 * blight.cpp:692).  blmx_set_shard (before the build) makes this index keep only the groups of rank `rank` of `world`;
 * every rank is given the SAME input, chops it, and derives the same key-balanced group boundaries, so no build-time
 * exchange is needed and IDs stay global.  world <= 16 (the partitioned exchange splits its bins at up to 15 boundaries).
 * blmx_shard_bounds returns the world+1 group boundaries.
 * Answers are those of the single index for EVERY k-mer, including the reference's own false positives: its 2^b scan runs past the
 * end of a bucket into the next one (blight.cpp:729-739), so every shard keeps a halo behind its sequence -- the first 2^b + k
 * nucleotides the single index holds behind the shard's last bucket (the head of the next owners' first buckets).
 * Routing helpers (device pointers): count -> destination rank of every k-mer + per-destination counts; scatter -> k-mers
 * grouped by destination (+ the permutation); unpermute -> out[order[i]] = vals[i]. */
int blmx_set_shard(blmx_index* ix, uint32_t rank, uint32_t world);
int blmx_shard_bounds(const blmx_index* ix, uint32_t* bounds_out);
int blmx_route_count_device(blmx_index* ix, const void* d_canon_kmers, uint64_t nq, const uint32_t* d_bounds, uint32_t world,
                            uint8_t* d_dest_out, uint64_t* d_counts_out, void* stream);
int blmx_route_scatter_device(blmx_index* ix, const void* d_canon_kmers, const uint8_t* d_dest, uint64_t nq, const uint64_t* d_offsets,
                              uint32_t world, void* d_sorted_kmers_out, uint32_t* d_order_out, uint64_t* d_cursor_scratch, void* stream);
int blmx_unpermute_device(blmx_index* ix, const int64_t* d_vals, const uint32_t* d_order, uint64_t nq, int64_t* d_out, void* stream);

/* The query pipeline as three separate stages, for the sharded exchange (one all-to-all-v out, one back):
 *   blmx_partition_device         on the rank that holds the queries: orders the batch by minimizer prefix; on a sharded
 *                                 index (blmx_set_shard) every partition bin belongs to exactly one owner, so the result
 *                                 is a concatenation of per-owner segments: rank_counts_out[r] (device, world entries,
 *                                 may be NULL) = length of owner r's segment.  inv_out[i] = position of query i in the
 *                                 ordered batch.  nq < 2^32 - 1.
 *   blmx_query_partitioned_device on the owner: answers a batch that is already ordered (a concatenation of ordered
 *                                 segments received from the peers) without another partition pass; ids in batch order.
 *   blmx_gather_ids_device        back on the first rank: ids_out[i] = ids_ordered[inv[i]].
 * With world = 1 the three calls in sequence are exactly blmx_query_kmers_device. */
int blmx_partition_device(blmx_index* ix, const void* d_kmers, uint64_t nq, void* d_ordered_kmers_out, uint32_t* d_inv_out,
                          uint64_t* d_rank_counts_out, void* stream);
int blmx_query_partitioned_device(blmx_index* ix, const void* d_ordered_kmers, uint64_t nq, int64_t* d_ids_out, void* stream);
int blmx_gather_ids_device(blmx_index* ix, const int64_t* d_ids_ordered, const uint32_t* d_inv, uint64_t nq, int64_t* d_ids_out,
                           void* stream);
/* ---- native multi-GPU queries: RCCL over xGMI behind this ABI, no Python / torch involved.  One process per GPU.  Rank 0 makes
 * a unique id (blmx_comm_unique_id, BLMX_COMM_ID_BYTES bytes) and hands it to the other ranks by any means (a file, MPI, a
 * socket); every rank then calls blmx_comm_create(id, rank, world, device) -- collective, like ncclCommInitRank.
 * blmx_query_kmers_sharded_device is the whole exchange step for one batch per rank: partition by owner (bins split at the shard
 * boundaries), all-to-all-v of the k-mers (grouped ncclSend / ncclRecv), the owner's query kernel, all-to-all-v of the IDs back,
 * gather into the caller's order; pieces of the batch are pipelined so that the exchange of one piece hides behind the queries
 * of another.  It is a collective: every rank of the communicator calls it, in the same order, each with its own batch (nq may
 * differ and may be 0).  The index must have been built with blmx_set_shard(rank, world) of the same rank / world.
 * librccl.so.1 is loaded at run time (BLMX_ERCCL when absent); single-GPU callers never need it. */
typedef struct blmx_comm blmx_comm;
#define BLMX_COMM_ID_BYTES 128
int blmx_comm_unique_id(void* id_out);
/* An id for the LOOPBACK transport: blmx_comm_create with such an id makes `world` communicators that live in ONE process -- one
 * caller thread per rank, all on the same GPU -- and move every flow with device copies under the same grouping and stream
 * ordering as RCCL.  For rehearsing and testing the multi-rank choreography on a one-GPU box (tests, tools/shard_query
 * --transport loopback); not a performance path.  Collectives block until every rank of the id has arrived (BLMX_ERCCL after
 * BLIGHT_LOOPBACK_TIMEOUT seconds, default 120). */
int blmx_comm_loopback_id(void* id_out);
/* "rccl" or "loopback" */
const char* blmx_comm_transport(const blmx_comm* c);
/* Communicator knobs; set the same value on every rank.  "flow_words": pairwise flows of the all-to-all-v are cut into rounds of at
 * most this many 8-byte words (default 2^25 = 256 MiB).  "pieces": pieces of a query batch that run as a software pipeline
 * (1..4; 0 = the environment's BLIGHT_DIST_CHUNKS, or 2).  "ship_minimizers": 1 = the minimizers of a slot range travel with its
 * k-mers (12 bytes per slot on the wire), 0 = the owner hashes the k-mers it receives again (8 bytes; the default, also -1;
 * BLIGHT_DIST_SHIP_MINIMIZERS sets it where the communicator does not). */
int blmx_comm_set_tuning(blmx_comm* c, const char* key, int64_t value);
int blmx_comm_create(const void* id, uint32_t rank, uint32_t world, int device, blmx_comm** out);
void blmx_comm_destroy(blmx_comm* c);
int blmx_query_kmers_sharded_device(blmx_index* ix, blmx_comm* c, const void* d_canon_kmers, uint64_t nq, int64_t* d_ids_out, void* stream);
/* the same collective with host buffers (staged through the index's device buffers; returns when ids_out is filled) */
int blmx_query_kmers_sharded(blmx_index* ix, blmx_comm* c, const void* canon_kmers, uint64_t nq, int64_t* ids_out);
/* query_sequence_hash against a sharded index: this rank's read front end makes the canonical k-mers, the exchange answers them.
 * Arguments as blmx_query_sequences / blmx_query_sequences_device; collective (a rank without reads passes nseq = 0). */
int blmx_query_sequences_sharded(blmx_index* ix, blmx_comm* c, const char* concat, const uint64_t* offsets, uint64_t nseq, int64_t* ids_out);
int blmx_query_sequences_sharded_device(blmx_index* ix, blmx_comm* c, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq,
                                        uint64_t total_len, uint64_t nk, int64_t* d_ids_out, int* d_err_flag, void* stream);

/* ---- sharded CONSTRUCTION from slices: every rank passes only ITS OWN consecutive piece of the input (rank order = input
 * order) and the super-k-mers travel to the rank that owns their MPHF group; IDs, membership and every array are those of the
 * whole-input build (blmx_set_shard + blmx_build_from_*), which every rank would have had to feed the complete input.
 * blmx_build_sharded_from_sequences / _device_sequences: the native collective (all-reduce of the per-group key histograms,
 * all-gather of the counts, all-to-all-v of the super-k-mers over RCCL); every rank of the communicator calls it; a rank whose
 * input is rejected makes all ranks fail together.  Arguments as blmx_build_from_sequences / blmx_build_from_device_sequences.
 * The three phases are also exported for callers with their own transport (and for the single-GPU virtual-rank tests):
 *   blmx_shard_build_begin[_device]  chop the slice; hist_out[2^log2_mphf + 1] = this slice's keys per MPHF group, then its
 *                                    super-k-mer count.  The caller sums hist_out over all ranks (element-wise).
 *   blmx_shard_build_plan            global_hist = that sum; world = the one given to blmx_set_shard (checked).  Derives the shard boundaries, orders the slice's super-k-mers by
 *                                    owner and packs them: send_sk[world] super-k-mers / send_words[world] u64 words go to each
 *                                    rank, taken consecutively from *d_meta (one u64 per super-k-mer) and *d_words (2-bit
 *                                    nucleotides; every destination's part starts on a word).  The buffers stay valid until finish.
 *   blmx_shard_build_finish          d_meta_in / d_words_in = what this rank received, concatenated in SOURCE-RANK order;
 *                                    recv_sk[world] / recv_words[world] = the counts per source.  Builds the shard.
 *   blmx_shard_halo_get / _set       after EVERY rank has finished: each rank publishes the head of its packed sequence
 *                                    (blmx_shard_halo_words() u64 words: valid nucleotides, then 2-bit packed words), the caller
 *                                    gathers the records of all ranks in rank order, and blmx_shard_halo_set(records, world) copies
 *                                    what follows this shard in the single index behind its own sequence (and adds the k-mers a
 *                                    window can now read there to the absent-key filter; waits for the filter's background passes).
 *                                    Until then windows that run past the shard's last bucket see padding: answers are those of the
 *                                    single index except for its false positives across the cut.  No-op on the last rank and on
 *                                    whole-input builds.  The native collectives do all of this themselves. */
int blmx_build_sharded_from_sequences(blmx_index* ix, blmx_comm* c, const char* concat, const uint64_t* offsets, uint64_t nseq);
/* every rank names the SAME file and keeps the rank-th consecutive part of its sequences ([nseq*rank/world, nseq*(rank+1)/world)) */
int blmx_build_sharded_from_fasta(blmx_index* ix, blmx_comm* c, const char* path);
int blmx_build_sharded_from_device_sequences(blmx_index* ix, blmx_comm* c, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq,
                                             uint64_t total_len);
int blmx_shard_build_begin(blmx_index* ix, const char* concat, const uint64_t* offsets, uint64_t nseq, uint64_t* hist_out);
int blmx_shard_build_begin_device(blmx_index* ix, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq, uint64_t total_len,
                                  uint64_t* hist_out);
int blmx_shard_build_plan(blmx_index* ix, const uint64_t* global_hist, uint32_t world, uint64_t* send_sk, uint64_t* send_words,
                          const void** d_meta, const void** d_words);
int blmx_shard_build_finish(blmx_index* ix, const void* d_meta_in, const void* d_words_in, const uint64_t* recv_sk, const uint64_t* recv_words);
uint64_t blmx_shard_halo_words(const blmx_index* ix);
int blmx_shard_halo_get(blmx_index* ix, uint64_t* record_out);
int blmx_shard_halo_set(blmx_index* ix, const uint64_t* records, uint32_t world);

/* Batch-query strategy of blmx_query_kmers / blmx_query_kmers_device: 0 = automatic (large batches on large indexes are first
 * partitioned by the top bits of their minimizer so that concurrently running blocks share a slice of the index in
 * L2 / Infinity Cache; results are identical and in the caller's order), 1 = always partition, -1 = never.
 * Environment override when 0: BLIGHT_QUERY_BINNING=1|-1. */
int blmx_set_query_binning(blmx_index* ix, int policy);
/* Per-index tuning knobs (experiments, tests, workload hints; results never depend on them).  The environment variables
 * BLIGHT_BIN_BITS, BLIGHT_MINI_HANDOFF, BLIGHT_QUERY_HEADS, BLIGHT_READS_MODE, BLIGHT_WINDOW_KB, BLIGHT_QUERY_BINNING,
 * BLIGHT_FILTER_BITS, BLIGHT_ABSENT_FILTER seed them when the index is created.  Keys:
 *   "filter_bits"    bits per key of the absent-key filter built at construction (0 = build none; default / negative = automatic: 8, and
 *                    16 at bits_to_save >= 7).  Set before building.
 *                    The filter is a blocked Bloom filter over the keys plus the few non-keys the reference's scan can find; a k-mer
 *                    it rejects is answered -1 after one 8-byte load.  It costs filter_bits / 8 bytes of HBM per indexed k-mer.
 *   "absent_filter"  0: queries ignore the filter (workloads in which nearly every k-mer is indexed run ~5 % faster without it);
 *                    1: use it, and wait for its background passes if they are still running (they start when the build returns and
 *                    take about as long as the build; until they are done queries run unfiltered).
 *   "pos_hint"       before building: 0 = position fields exactly as wide as the reference's.  Default 1: at bits_to_save >= 7 a
 *                    field keeps three more low bits of the position, so that a query finds an indexed k-mer in a 2^(b-3)
 *                    sub-window instead of scanning the reference's 2^b window (which remains the fallback).
 *   "filter_gate"    1 (default): a large batch in which fewer than 1 in 12 answers were "absent" switches the filter off for the
 *                    batches after it (and one with more switches it on again); 0: the filter is always consulted.
 *   "partition"      large k-mer batches are ordered by minimizer prefix in front of the query kernel: 0 = exact counting sort for
 *                    the first batch, afterwards one pass into bin regions sized after the previous batch (default); 1 = always
 *                    one pass; 2 = always the exact sort.  "slot_slack_pct": slack of the regions (0 = default).
 *   "bin_bits"       partition bins = 2^value (0 = default: 10 with the one-pass partition, 9 with the exact one)
 *   "bin_rounds"     the one-pass partition kernel stages a tile of k-mers in LDS in 2 pieces, two workgroups per compute unit (default), or
 *                    in 1, one workgroup per compute unit that loads the next tile's k-mers ahead
 *   "mini_handoff"   1/0/-1: the partition pass hands the minimizers to the query kernel / it recomputes them / automatic
 *   "query_heads"    0: query kernels read the full group descriptor instead of the compact head
 *   "reads_mode"     sequence / file queries: 0 automatic, 1 front end fused with the query body, 2 front end -> partitioned pipeline
 *   "window_kb"      KiB of characters per front-end window
 *   "fat_lines"      before building: 0 = the absent-key filter in an array of its own and 16-byte MPHF lines throughout (default 1: with 8
 *                    filter bits per key the filter lives inside 64-byte level-0 lines, one request for filter test and level-0 probe)
 *   "build_passes"   before building: the construction expands keys, builds the MPHFs and fills the positions for one range of MPHF
 *                    groups at a time (the reference: one of 2^log2_superbuckets super-buckets at a time, blight.cpp:335-356), which
 *                    bounds its scratch memory.  0 = automatic (as few passes as keep a pass below 2^30 keys, at most
 *                    2^min(log2_superbuckets, 6)); n > 0 = that many (rounded up to a power of two).  The index does not depend on it.
 *   "slot_piece_min" sharded exchange: pieces of a batch with at least this many k-mers travel as the one-pass partition's bin regions
 *                    (k-mers + minimizers out, 4-byte IDs back); smaller ones, and the first batch of an index, as dense per-owner
 *                    segments (0 = default 2^19).  The same on every rank is not required: the format travels with the counts.
 *   "narrow_ids"     0: host-buffer results cross PCIe as 64-bit words even when every ID fits 32 bits
 *   "query_binning"  as blmx_set_query_binning
 *   "filter_junctions"  test hook: 0 skips the filter's second pass (the filter then loses the reference's false positives) */
int blmx_set_tuning(blmx_index* ix, const char* key, int value);

int blmx_stats(const blmx_index* ix, blmx_stats_t* out);
/* One-pass partition of large k-mer batches (bin regions sized after the PREVIOUS batch): out4 = [batches that took it, their k-mers,
 * k-mers that did not fit their bin's region and went through the overflow region (they are answered all the same, only without
 * locality), the largest such count of one batch]; reset != 0 clears the counters.  Synchronises the device. */
int blmx_partition_stats(blmx_index* ix, uint64_t* out4, int reset);
/* The bounded serving burst that answers small calls (one k-mer, one read: the reference's calling pattern, blight.cpp:537-550,
 * 575-591): out4 = [serving kernels launched so far, calls they answered, kernels launched because a caller's 200-us deadline found
 * the previous one gone, 1 if a serving kernel is alive right now].  A kernel leaves by itself after BLIGHT_SERVE_IDLE_US (60)
 * microseconds without a request or BLIGHT_SERVE_MAX_MS (20) milliseconds, whichever comes first; blmx_set_tuning("serve_burst", 0)
 * stops it and sends small calls through one launch per round instead. */
int blmx_serve_stats(blmx_index* ix, uint64_t* out4);
/* Memory the library keeps between calls, per process: device blocks of 1 MiB and more that a finished build or a destroyed index
 * gave up (scratch, index arrays, staging; up to 1/6 of the device's memory, or BLIGHT_SCRATCH_CACHE_GB; they serve the next build
 * or index of the process -- on this driver hipMalloc of VRAM the process has used before costs 25-100 ms per GB, and ~3 s when the
 * driver has not finished reclaiming it) and the pinned window buffers of destroyed indexes (up to 1 GiB; pinning costs ~0.7 ms per MiB).  This call returns all
 * of it to the driver; *device_bytes / *pinned_bytes (either may be NULL) receive what was released.  Safe at any time: the caches
 * also empty themselves when an allocation fails. */
int blmx_release_scratch(uint64_t* device_bytes, uint64_t* pinned_bytes);

/* Copy index arrays back to the host for array-level parity checks (tests only; not needed by callers).
 * which: 0 bucket_nt u32[nbuckets] | 1 bucket_start u64[nbuckets] | 2 packed sequence u64[ceil(total_nt/32)]
 *        3 positions u64[ceil(bits/64)] | 4 group nkeys u64 | 5 group prefix u64 | 6 group posstart u64
 *        7 group nbits u32 | 8 level sizes u64[ngroups*16] | 9 group logical word base u64[ngroups+1]
 *        10 MPHF logical bit array u64[total words] (levels of all groups back to back, ranks stripped)
 *        11 fallback keys (kmer_bytes each) | 12 fallback ranks u64 | 13 fallback group u32
 * Returns the number of ELEMENTS available; copies min(cap, available) elements into out when out != NULL. */
int64_t blmx_export(blmx_index* ix, int which, void* out, uint64_t cap);

/* Primitive known-answer entry points (device kernels; tests only): op 0 revhash32(u32)->i32, 1 rc32 (arg m),
 * 2 parity_canon (arg m), 3 kmer rc (arg k), 4 kmer minimizer (arg k, arg2 m) -> u32, 5 level hashes: 16 u64 per key,
 * 6 nuc code of bytes -> u8 (4 = invalid), 7 MurmurHash3_x64_128 (reference MurmurHash3.cpp:255; dead code there, a
 * device utility here) of the first arg2 bytes (0 = all) of every kmer_bytes-byte record, seed arg -> 2 u64.
 * 8 the build's stable sort: n u32 keys with arg significant bits (1..32) -> n sorted keys followed by the n input indices in that order.
 * in/out are host buffers. */
int blmx_primitive(int op, int kmer_bytes, const void* in, uint64_t n, int arg, int arg2, void* out);

const char* blmx_version(void);
int blmx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BLIGHT_MI355X_H */
