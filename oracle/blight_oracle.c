/*
 * blight_oracle.c -- CPU ORACLE for the Blight k-mer index hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg may load this library; the shipped library (blight_amd/csrc -> libblight_mi355x.so) never links, loads
 * or calls it and has no CPU fallback.
 *
 * What it is: a clean-room plain-C restatement of the reference algorithm (Piezoid/Blight @ /root/reference,
 * with the three one-token fixes P1/P2 of SURVEY.md App. B-1 that make the reference self-consistent), written
 * from the validated specification in SURVEY.md App. A.  Every function cites the reference file:line it restates.
 *
 * Parity pinning: this restatement is checked bit-for-bit (IDs, membership, positions, and every internal array:
 * bucket table, group table, packed sequence, position fields, MPHF level sizes / bit arrays / fallback tables)
 * against the REAL reference compiled from /root/reference by oracle/build_ref.sh (oracle/_ref/libblight_ref*.so),
 * both live (tests/test_oracle_vs_ref.py, runs where /root/reference or the prebuilt _ref exists) and through the
 * committed golden vectors in tests/golden/ that oracle/make_golden.py produced from that reference build.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_NLEVELS 16          /* mphf<..., _nb_levels = 16>, bbhash.h:516 */
#define ORC_WORDS_PER_RANK 16   /* bitVector::_words_per_rankingblock, bbhash.h:505 */
#define ORC_NOT_FOUND 0xffffffffffffffffull /* ULLONG_MAX, bbhash.h:574 */
#define ORC_EDNA (-3)   /* std::domain_error("Invalid char in DNA"), kmer.h:68 */
#define ORC_ESHORT (-2) /* sequence shorter than k: UB / abort in the reference (SURVEY C-13) */
#define ORC_EINVAL (-1)

/* ---- width-independent primitives -------------------------------------------------------------- */

/* nuc2int, kmer.h:56-69: A0 C1 T2 G3 (= (c>>1)&3), either case; anything else is an error (-1 here). */
static inline int orc_nuc2int(char c)
{
	uint8_t d = (uint8_t)(c - 65);
	if (d <= 51 && (0x0008004500080045ull & (1ull << d)))
		return (c >> 1) & 3;
	return -1;
}

/* rcb(uint64_t, n), kmer.h:218-232 */
static inline uint64_t orc_rc64(uint64_t in, int n)
{
	uint64_t res = __builtin_bswap64(in ^ 0xaaaaaaaaaaaaaaaaull);
	const uint64_t c1 = 0x0f0f0f0f0f0f0f0full, c2 = 0x3333333333333333ull;
	res = ((res & c1) << 4) | ((res & (c1 << 4)) >> 4);
	res = ((res & c2) << 2) | ((res & (c2 << 2)) >> 2);
	return n ? res >> (64 - 2 * n) : 0;
}

/* rcb(uint32_t, n), kmer.h:236-251 */
static inline uint32_t orc_rc32(uint32_t in, int n)
{
	uint32_t res = __builtin_bswap32(in ^ 0xaaaaaaaau);
	const uint32_t c1 = 0x0f0f0f0fu, c2 = 0x33333333u;
	res = ((res & c1) << 4) | ((res & (c1 << 4)) >> 4);
	res = ((res & c2) << 2) | ((res & (c2 << 2)) >> 2);
	return res >> (32 - 2 * n);
}

/* ParityCanonical::canonize, kmer.h:480-482 (odd m: exactly one strand has odd popcount) */
static inline uint32_t orc_parity_canon(uint32_t x, int m)
{
	return ((__builtin_popcount(x) & 1) ? x : orc_rc32(x, m)) >> 1;
}

/* revhash(uint32_t) -> int32_t, kmer.h:102-108 */
static inline int32_t orc_revhash32(uint32_t x)
{
	x = ((x >> 16) ^ x) * 0x2c1b3c6du;
	x = ((x >> 16) ^ x) * 0x297a2d39u;
	x = ((x >> 16) ^ x);
	return (int32_t)x;
}

/* SingleHashFunctor::hash_bis, bbhash.h:172-185 */
static inline uint64_t orc_hash_bis(uint64_t key, uint64_t seed)
{
	uint64_t hash = seed;
	hash ^= (hash << 7) ^ key * (hash >> 3) ^ (~((hash << 11) + (key ^ (hash >> 5))));
	hash = (~hash) + (hash << 21);
	hash = hash ^ (hash >> 24);
	hash = (hash + (hash << 3)) + (hash << 8);
	hash = hash ^ (hash >> 14);
	hash = (hash + (hash << 2)) + (hash << 4);
	hash = hash ^ (hash >> 28);
	hash = hash + (hash << 31);
	return hash;
}

/* XorshiftHashFunctors::iter / h0 / h1 / next, bbhash.h:219-250 */
static inline uint64_t orc_level_hash_step(uint64_t key, uint64_t *s0, uint64_t *s1, unsigned level)
{
	if (level == 0) {
		*s0 = orc_hash_bis(key, 0xAAAAAAAA55555555ull);
		return *s0;
	}
	if (level == 1) {
		*s1 = orc_hash_bis(key, 0x33333333CCCCCCCCull);
		return *s1;
	}
	uint64_t a = *s0;
	const uint64_t b = *s1;
	*s0 = b;
	a ^= a << 23;
	*s1 = a ^ b ^ (a >> 17) ^ (b >> 26);
	return *s1 + b;
}

/* fastmod64, bbhash.h:660-662 */
static inline uint64_t orc_mulhi64(uint64_t a, uint64_t b)
{
	return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
}

/* mphf::configureLevels, bbhash.h:591-614 (IEEE double + libm pow, gamma = 2.0).  Returns the total bit count. */
static uint64_t orc_configure_levels(uint64_t nelem, uint64_t sizes[ORC_NLEVELS])
{
	const double gamma = 2.0;
	double proba_collision = 1.0 - pow(((gamma * (double)nelem - 1) / (gamma * (double)nelem)), (double)(nelem - 1));
	uint64_t total = 0;
	size_t hash_domain = (size_t)(ceil((double)nelem * gamma));
	for (unsigned ii = 0; ii < ORC_NLEVELS; ii++) {
		sizes[ii] = (((uint64_t)((double)hash_domain * pow(proba_collision, (double)ii)) + 63) / 64) * 64;
		if (sizes[ii] == 0)
			sizes[ii] = 64;
		total += sizes[ii];
	}
	return total;
}

/* ---- the two width instantiations -------------------------------------------------------------- */

#define ORC_CAT2(a, b) a##b
#define ORC_CAT(a, b) ORC_CAT2(a, b)

#define KMER_BITS 64
#define KMER_T uint64_t
#define SUF(x) ORC_CAT(x, _64)
#include "oracle_impl.h"
#undef KMER_BITS
#undef KMER_T
#undef SUF

#define KMER_BITS 128
#define KMER_T unsigned __int128
#define SUF(x) ORC_CAT(x, _128)
#include "oracle_impl.h"
#undef KMER_BITS
#undef KMER_T
#undef SUF

/* ---- exported C API (ctypes) -------------------------------------------------------------------- */

typedef struct orc_handle {
	int kmer_bytes; /* 8 or 16 */
	int borrowed;   /* orc_import: the big arrays belong to the caller */
	orc_index_64 *i64;
	orc_index_128 *i128;
} orc_handle;

/* kmer_Set_Light ctor validation, blight.h:75-92.  Returns 0 or a negative code; *msg gets the reference's text. */
int orc_check_params(int k, int m, int n, int s, int kmer_bytes, const char **msg)
{
	static const char *M[] = {"", "kmer size too large", "minimizer_length must be odd", "minimizer_length size too large",
	                          "log2_mphfs_number must not be larger than 2*minimizer_length - 1",
	                          "log2_superbuckets_number must not be larger than log2_mphfs_number"};
	int code = 0;
	if (k > kmer_bytes * 4)
		code = 1;
	else if ((m & 1) == 0)
		code = 2;
	else if (m > 16)
		code = 3;
	else if (n > 2 * m - 1)
		code = 4;
	else if (s > n)
		code = 5;
	if (msg)
		*msg = M[code];
	return -code;
}

/* Build from nseq sequences given as one concatenated buffer + nseq+1 offsets.  kmer_bytes = 8 (k<=32) or 16. */
int orc_build(orc_handle **out, int k, int m, int n, int b, int kmer_bytes, const char *concat, const uint64_t *offs,
              uint64_t nseq)
{
	if (!out || (kmer_bytes != 8 && kmer_bytes != 16) || k < m || k > kmer_bytes * 4 || (m & 1) == 0 || m > 16 ||
	    n > 2 * m - 1 || b < 0 || b > 24 || (kmer_bytes == 16 && k > 63) || (kmer_bytes == 8 && k > 32))
		return ORC_EINVAL;
	orc_handle *h = (orc_handle *)calloc(1, sizeof(*h));
	h->kmer_bytes = kmer_bytes;
	int rc = (kmer_bytes == 8) ? orc_build_64(&h->i64, k, m, n, b, concat, offs, nseq)
	                           : orc_build_128(&h->i128, k, m, n, b, concat, offs, nseq);
	if (rc) {
		free(h);
		return rc;
	}
	*out = h;
	return 0;
}

void orc_free(orc_handle *h)
{
	if (!h)
		return;
	if (h->borrowed) {
		orc_free_imported_64(h->i64);
		orc_free_imported_128(h->i128);
		free(h);
		return;
	}
	orc_free_index_64(h->i64);
	orc_free_index_128(h->i128);
	free(h);
}

/* query_kmer_hash (blight.cpp:545-550): ids[i] = ID of canonical k-mer i or -1.  kmers: nq * kmer_bytes, little endian. */
void orc_query_kmers(const orc_handle *h, const void *kmers, uint64_t nq, int64_t *ids, int threads)
{
	(void)threads;
	if (h->kmer_bytes == 8) {
		const uint64_t *q = (const uint64_t *)kmers;
		const orc_index_64 *ix = h->i64;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
		for (int64_t i = 0; i < (int64_t)nq; i++)
			ids[i] = orc_query_one_64(ix, q[i], orc_minimizer_64(q[i], ix->k, ix->m), 0);
	} else {
		const unsigned __int128 *q = (const unsigned __int128 *)kmers;
		const orc_index_128 *ix = h->i128;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
		for (int64_t i = 0; i < (int64_t)nq; i++) {
			unsigned __int128 x;
			memcpy(&x, &q[i], 16);
			ids[i] = orc_query_one_128(ix, x, orc_minimizer_128(x, ix->k, ix->m), 0);
		}
	}
}

/* query_get_hash(canon, minimizer) (blight.cpp:716-742) / query_get_pos_unitig (:686-712) with caller minimizers */
void orc_query_kmers_mini(const orc_handle *h, const void *kmers, const uint32_t *minis, uint64_t nq, int64_t *out,
                          int want_pos)
{
	if (h->kmer_bytes == 8) {
		const uint64_t *q = (const uint64_t *)kmers;
		for (uint64_t i = 0; i < nq; i++)
			out[i] = orc_query_one_64(h->i64, q[i], minis[i], want_pos);
	} else {
		const unsigned __int128 *q = (const unsigned __int128 *)kmers;
		for (uint64_t i = 0; i < nq; i++) {
			unsigned __int128 x;
			memcpy(&x, &q[i], 16);
			out[i] = orc_query_one_128(h->i128, x, minis[i], want_pos);
		}
	}
}

/*
 * query_sequence_hash (blight.cpp:575-591): rolling canonical k-mers (updateK / updateRCK, :78-98) over an ASCII
 * sequence; out must hold max(0, len-k+1) entries.  Returns the number of k-mers, or ORC_EDNA.
 * If kmers_out != NULL the canonical k-mers are stored there too (kmer_bytes each).
 */
int64_t orc_query_sequence(const orc_handle *h, const char *s, uint64_t len, int64_t *out, void *kmers_out)
{
	const int k = h->kmer_bytes == 8 ? h->i64->k : h->i128->k;
	if (len < (uint64_t)k)
		return 0;
	if (h->kmer_bytes == 8) {
		const orc_index_64 *ix = h->i64;
		uint64_t seq = 0, rc;
		const uint64_t mask = orc_kmask_64(k);
		for (int i = 0; i < k; i++) { /* str2num, kmer.h:90-98 */
			int c = orc_nuc2int(s[i]);
			if (c < 0)
				return ORC_EDNA;
			seq = (seq << 2) | (uint64_t)c;
		}
		rc = orc_rc_64(seq, k);
		uint64_t *ko = (uint64_t *)kmers_out;
		for (uint64_t i = 0;; i++) {
			uint64_t canon = seq <= rc ? seq : rc;
			if (ko)
				ko[i] = canon;
			if (out)
				out[i] = orc_query_one_64(ix, canon, orc_minimizer_64(canon, k, ix->m), 0);
			if (i + (uint64_t)k >= len)
				return (int64_t)(i + 1);
			int c = orc_nuc2int(s[i + (uint64_t)k]);
			if (c < 0)
				return ORC_EDNA;
			seq = ((seq << 2) | (uint64_t)c) & mask;                      /* updateK  */
			rc = (rc >> 2) | ((uint64_t)(c ^ 2) << (2 * k - 2));          /* updateRCK */
		}
	} else {
		const orc_index_128 *ix = h->i128;
		unsigned __int128 seq = 0, rc;
		const unsigned __int128 mask = orc_kmask_128(k);
		for (int i = 0; i < k; i++) {
			int c = orc_nuc2int(s[i]);
			if (c < 0)
				return ORC_EDNA;
			seq = (seq << 2) | (unsigned)c;
		}
		rc = orc_rc_128(seq, k);
		unsigned char *ko = (unsigned char *)kmers_out;
		for (uint64_t i = 0;; i++) {
			unsigned __int128 canon = seq <= rc ? seq : rc;
			if (ko)
				memcpy(ko + 16 * i, &canon, 16);
			if (out)
				out[i] = orc_query_one_128(ix, canon, orc_minimizer_128(canon, k, ix->m), 0);
			if (i + (uint64_t)k >= len)
				return (int64_t)(i + 1);
			int c = orc_nuc2int(s[i + (uint64_t)k]);
			if (c < 0)
				return ORC_EDNA;
			seq = ((seq << 2) | (unsigned)c) & mask;
			rc = (rc >> 2) | ((unsigned __int128)(unsigned)(c ^ 2) << (2 * k - 2));
		}
	}
}

/*
 * Import an index image produced elsewhere (bench.py's cpu_baseline leg: the arrays exported from the HIP build through
 * blmx_export) so that the restated CPU query path (orc_query_kmers) can be timed and cross-checked on the SAME index
 * at sizes the single-threaded CPU build cannot reach in bench time.  All arrays are BORROWED (the caller keeps them
 * alive); only the rank samples (bitVector::build_ranks, bbhash.h:447-465) and the per-group fallback tables are built
 * here.  seq must be padded with zero words (>= (2^b + k)/32 + 4) past total_nt.  64-bit k-mers only.
 */
int orc_import(orc_handle **out, int k, int m, int n, int b, int kmer_bytes, uint64_t total_nt, uint64_t number_kmer, const uint32_t *bucket_nt,
               const uint64_t *bucket_start, const uint64_t *seq, const uint64_t *positions, uint64_t positions_bits,
               const uint64_t *g_nkeys, const uint64_t *g_prefix, const uint64_t *g_posstart, const uint32_t *g_nbits,
               const uint64_t *level_sizes, const uint64_t *wbase, const uint64_t *bits, const uint64_t *fb_keys,
               const uint64_t *fb_ranks, const uint32_t *fb_group, uint64_t nfb)
{
	if (kmer_bytes != 8 && kmer_bytes != 16)
		return ORC_EINVAL;
	orc_handle *h = (orc_handle *)calloc(1, sizeof(*h));
	h->kmer_bytes = kmer_bytes;
	h->borrowed = 1;
	if (kmer_bytes == 8)
		h->i64 = orc_import_64(k, m, n, b, total_nt, number_kmer, bucket_nt, bucket_start, seq, positions, positions_bits, g_nkeys, g_prefix,
		                       g_posstart, g_nbits, level_sizes, wbase, bits, fb_keys, fb_ranks, fb_group, nfb);
	else
		h->i128 = orc_import_128(k, m, n, b, total_nt, number_kmer, bucket_nt, bucket_start, seq, positions, positions_bits, g_nkeys, g_prefix,
		                         g_posstart, g_nbits, level_sizes, wbase, bits, fb_keys, fb_ranks, fb_group, nfb);
	*out = h;
	return 0;
}

/* ---- introspection (for array-level parity tests against oracle/_ref and the HIP build) ----------- */

typedef struct {
	uint64_t number_kmer, number_super_kmer, total_nt, largest_mphf, largest_bucket, positions_bits, seq_words;
	uint32_t nbuckets, ngroups;
	int k, m, n, b, kmer_bytes;
} orc_stats_t;

void orc_stats(const orc_handle *h, orc_stats_t *st)
{
#define FILL(ix)                                      \
	st->number_kmer = ix->number_kmer;                \
	st->number_super_kmer = ix->number_super_kmer;    \
	st->total_nt = ix->total_nt;                      \
	st->largest_mphf = ix->largest_mphf;              \
	st->largest_bucket = ix->largest_bucket;          \
	st->positions_bits = ix->positions_bits;          \
	st->seq_words = ix->seq_words;                    \
	st->nbuckets = ix->nbuckets;                      \
	st->ngroups = ix->ngroups;                        \
	st->k = ix->k;                                    \
	st->m = ix->m;                                    \
	st->n = ix->n;                                    \
	st->b = ix->b;
	if (h->kmer_bytes == 8) {
		FILL(h->i64)
	} else {
		FILL(h->i128)
	}
	st->kmer_bytes = h->kmer_bytes;
#undef FILL
}

#define ORC_PTR(field) (h->kmer_bytes == 8 ? (const void *)h->i64->field : (const void *)h->i128->field)
const uint32_t *orc_bucket_nt(const orc_handle *h) { return (const uint32_t *)ORC_PTR(bucket_nt); }
const uint64_t *orc_bucket_start(const orc_handle *h) { return (const uint64_t *)ORC_PTR(bucket_start); }
const uint64_t *orc_seq(const orc_handle *h) { return (const uint64_t *)ORC_PTR(seq); }
const uint64_t *orc_positions(const orc_handle *h) { return (const uint64_t *)ORC_PTR(positions); }
const uint64_t *orc_group_nkeys(const orc_handle *h) { return (const uint64_t *)ORC_PTR(g_nkeys); }
const uint64_t *orc_group_prefix(const orc_handle *h) { return (const uint64_t *)ORC_PTR(g_prefix); }
const uint64_t *orc_group_posstart(const orc_handle *h) { return (const uint64_t *)ORC_PTR(g_posstart); }
const uint8_t *orc_group_nbits(const orc_handle *h) { return (const uint8_t *)ORC_PTR(g_nbits); }

/* per-group MPHF: level sizes (16), bit array, fallback table.  Returns nwords; pointers may be NULL for empty groups */
uint64_t orc_mphf_info(const orc_handle *h, uint32_t g, const uint64_t **sizes, const uint64_t **bits, uint64_t *nfallback,
                       int *levels_used)
{
	if (h->kmer_bytes == 8) {
		const orc_mphf_64 *mp = &h->i64->mphf[g];
		*sizes = mp->sizes;
		*bits = mp->bits;
		*nfallback = mp->nfallback;
		*levels_used = mp->levels_used;
		return mp->nwords;
	}
	const orc_mphf_128 *mp = &h->i128->mphf[g];
	*sizes = mp->sizes;
	*bits = mp->bits;
	*nfallback = mp->nfallback;
	*levels_used = mp->levels_used;
	return mp->nwords;
}

/* fallback entry i of group g: key (kmer_bytes, LE) and rank */
void orc_mphf_fallback(const orc_handle *h, uint32_t g, uint64_t i, void *key_out, uint64_t *rank)
{
	if (h->kmer_bytes == 8) {
		memcpy(key_out, &h->i64->mphf[g].fallback[i].key, 8);
		*rank = h->i64->mphf[g].fallback[i].rank;
	} else {
		memcpy(key_out, &h->i128->mphf[g].fallback[i].key, 16);
		*rank = h->i128->mphf[g].fallback[i].rank;
	}
}

/* raw MPHF rank of a key within group g (mphf::lookup), ORC_NOT_FOUND if none */
uint64_t orc_mphf_lookup(const orc_handle *h, uint32_t g, const void *key)
{
	if (h->kmer_bytes == 8) {
		uint64_t x;
		memcpy(&x, key, 8);
		return h->i64->mphf[g].nelem ? orc_mphf_lookup_64(&h->i64->mphf[g], x) : ORC_NOT_FOUND;
	}
	unsigned __int128 x;
	memcpy(&x, key, 16);
	return h->i128->mphf[g].nelem ? orc_mphf_lookup_128(&h->i128->mphf[g], x) : ORC_NOT_FOUND;
}

/* ---- primitive KAT entry points ------------------------------------------------------------------ */
int orc_p_nuc2int(char c) { return orc_nuc2int(c); }
int32_t orc_p_revhash32(uint32_t x) { return orc_revhash32(x); }
uint32_t orc_p_rc32(uint32_t x, int m) { return orc_rc32(x, m); }
uint32_t orc_p_parity_canon(uint32_t x, int m) { return orc_parity_canon(x, m); }
uint64_t orc_p_rc64(uint64_t x, int k) { return orc_rc64(x, k); }
void orc_p_rc128(const void *in, int k, void *out)
{
	unsigned __int128 x;
	memcpy(&x, in, 16);
	x = orc_rc_128(x, k);
	memcpy(out, &x, 16);
}
uint32_t orc_p_minimizer64(uint64_t x, int k, int m) { return orc_minimizer_64(x, k, m); }
uint32_t orc_p_minimizer128(const void *in, int k, int m)
{
	unsigned __int128 x;
	memcpy(&x, in, 16);
	return orc_minimizer_128(x, k, m);
}
uint64_t orc_p_hash_bis(uint64_t key, uint64_t seed) { return orc_hash_bis(key, seed); }
void orc_p_level_hashes(uint64_t key, unsigned nlevels, uint64_t *out)
{
	uint64_t s0 = 0, s1 = 0;
	for (unsigned l = 0; l < nlevels; l++)
		out[l] = orc_level_hash_step(key, &s0, &s1, l);
}
uint64_t orc_p_mulhi64(uint64_t a, uint64_t b) { return orc_mulhi64(a, b); }
/*
 * MurmurHash3_x64_128 (reference MurmurHash3.cpp:255-334; Appleby, public domain).  DEAD CODE in the reference;
 * restated because the product ships a device version of it.  Pinned by the known answers of SURVEY App. B-4 and against
 * the reference's own MurmurHash3.cpp compiled into oracle/_ref/libmurmur3_ref.so.
 */
static uint64_t orc_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t orc_fmix64(uint64_t k) /* MurmurHash3.cpp:81-90 */
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdull;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ull;
	k ^= k >> 33;
	return k;
}
static uint64_t orc_le(const uint8_t *p, int nbytes)
{
	uint64_t v = 0;
	for (int i = 0; i < nbytes; i++)
		v |= (uint64_t)p[i] << (8 * i);
	return v;
}
void orc_p_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2])
{
	const uint8_t *p = (const uint8_t *)key;
	const uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
	uint64_t h1 = seed, h2 = seed;
	const int nblocks = len / 16;
	for (int i = 0; i < nblocks; i++) { /* :272-288 */
		uint64_t k1 = orc_le(p + 16 * i, 8), k2 = orc_le(p + 16 * i + 8, 8);
		k1 *= c1; k1 = orc_rotl64(k1, 31); k1 *= c2; h1 ^= k1;
		h1 = orc_rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
		k2 *= c2; k2 = orc_rotl64(k2, 33); k2 *= c1; h2 ^= k2;
		h2 = orc_rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
	}
	const uint8_t *tail = p + 16 * nblocks; /* :290-317 */
	const int rem = len & 15;
	if (rem > 8) {
		uint64_t k2 = orc_le(tail + 8, rem - 8);
		k2 *= c2; k2 = orc_rotl64(k2, 33); k2 *= c1; h2 ^= k2;
	}
	if (rem > 0) {
		uint64_t k1 = orc_le(tail, rem > 8 ? 8 : rem);
		k1 *= c1; k1 = orc_rotl64(k1, 31); k1 *= c2; h1 ^= k1;
	}
	h1 ^= (uint64_t)len; h2 ^= (uint64_t)len; /* :319-331 */
	h1 += h2; h2 += h1;
	h1 = orc_fmix64(h1); h2 = orc_fmix64(h2);
	h1 += h2; h2 += h1;
	out[0] = h1;
	out[1] = h2;
}

uint64_t orc_p_configure_levels(uint64_t n, uint64_t *sizes) { return orc_configure_levels(n, sizes); }
/* str2num (kmer.h:90-98) for k <= 32; returns 0 and sets *err on an invalid character */
uint64_t orc_p_str2num64(const char *s, int k, int *err)
{
	uint64_t r = 0;
	*err = 0;
	for (int i = 0; i < k; i++) {
		int c = orc_nuc2int(s[i]);
		if (c < 0) {
			*err = 1;
			return 0;
		}
		r = (r << 2) | (uint64_t)c;
	}
	return r;
}

/*
 * Super-k-mer list of one sequence (SuperKChopper, kmer.h:629-788 + P2): writes up to cap records of
 * (offset, length, minimizer) and returns the count (or a negative error).
 */
int64_t orc_p_chop(const char *s, uint64_t len, int k, int m, uint32_t *offs, uint32_t *lens, uint32_t *minis, uint64_t cap)
{
	orc_superk_vec_64 sv = {0, 0, 0};
	int rc = orc_chop_64(s, len, k, m, 0, &sv);
	if (rc) {
		free(sv.v);
		return rc;
	}
	for (uint64_t i = 0; i < sv.n && i < cap; i++) {
		offs[i] = sv.v[i].offset;
		lens[i] = sv.v[i].length;
		minis[i] = sv.v[i].mini;
	}
	int64_t n = (int64_t)sv.n;
	free(sv.v);
	return n;
}

/*
 * FASTA reader restating blight.cpp:212-229 / :760-775: strictly alternating (header, sequence) lines; an empty
 * header line swallows one more line; an empty sequence line is skipped.  Plain text only (tests write plain files).
 * Output: malloc'ed concat buffer + offsets (caller frees with orc_free_buf).
 */
int orc_read_fasta(const char *path, char **concat_out, uint64_t **offs_out, uint64_t *nseq_out)
{
	FILE *f = fopen(path, "rb");
	if (!f)
		return ORC_EINVAL;
	char *line = NULL;
	size_t lcap = 0;
	char *buf = NULL;
	uint64_t blen = 0, bcap = 0;
	uint64_t *offs = (uint64_t *)malloc(sizeof(uint64_t) * 1024);
	uint64_t nseq = 0, ocap = 1024;
	offs[0] = 0;
	for (;;) {
		ssize_t r = getline(&line, &lcap, f);
		if (r < 0)
			break;
		while (r > 0 && line[r - 1] == '\n')
			r--;
		if (r == 0) { /* empty header: skip one more line */
			if (getline(&line, &lcap, f) < 0)
				break;
			continue;
		}
		r = getline(&line, &lcap, f);
		if (r < 0)
			break;
		while (r > 0 && line[r - 1] == '\n')
			r--;
		if (r == 0)
			continue;
		if (blen + (uint64_t)r > bcap) {
			bcap = (blen + (uint64_t)r) * 2 + 4096;
			buf = (char *)realloc(buf, bcap);
		}
		memcpy(buf + blen, line, (size_t)r);
		blen += (uint64_t)r;
		if (nseq + 2 > ocap) {
			ocap *= 2;
			offs = (uint64_t *)realloc(offs, sizeof(uint64_t) * ocap);
		}
		offs[++nseq] = blen;
	}
	free(line);
	fclose(f);
	if (!buf)
		buf = (char *)malloc(1);
	*concat_out = buf;
	*offs_out = offs;
	*nseq_out = nseq;
	return 0;
}

void orc_free_buf(void *p) { free(p); }
