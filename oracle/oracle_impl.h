/*
 * oracle_impl.h -- width-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by blight_oracle.c, once with KMER_T = uint64_t (k <= 32, the reference's
 * shipped kmer_t, kmer.h:18) and once with KMER_T = unsigned __int128 (k <= 63, the reference's
 * documented "T128" typedef switch, kmer.h:18 comment).  SUF(x) appends _64 / _128.
 *
 * This is a clean-room restatement of the reference algorithm; every function cites the reference
 * file:line it restates (paths relative to /root/reference).  Nothing here is shipped in the product
 * library: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 */

/* ---- k-mer primitives ------------------------------------------------------------------------ */

/* rcb(kmer_t, n): reverse the n 2-bit digits and complement each (XOR 0b10).  kmer.h:191-232 */
static inline KMER_T SUF(orc_rc)(KMER_T x, int k)
{
#if KMER_BITS == 64
	return orc_rc64(x, k);
#else
	uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
	/* reverse all 64 digits of the 128-bit value, then realign right */
	KMER_T r = ((KMER_T)orc_rc64(lo, 32) << 64) | (KMER_T)orc_rc64(hi, 32);
	return r >> (128 - 2 * k);
#endif
}

/* min_k: canonical = numeric min(fwd, rc).  blight.cpp:86-91 */
static inline KMER_T SUF(orc_canon)(KMER_T x, int k)
{
	KMER_T r = SUF(orc_rc)(x, k);
	return x <= r ? x : r;
}

static inline KMER_T SUF(orc_kmask)(int k)
{
	return (2 * k >= KMER_BITS) ? ~(KMER_T)0 : ((((KMER_T)1) << (2 * k)) - 1);
}

/* minimizer_naive with patch P1 (canonize(mmer, m)).  kmer.h:791-810 */
static inline uint32_t SUF(orc_minimizer)(KMER_T seq, int k, int m)
{
	const uint64_t mask = (m >= 16) ? 0xffffffffull : ((1ull << (2 * m)) - 1);
	uint32_t mmer = (uint32_t)((uint64_t)seq & mask);
	uint32_t mini = orc_parity_canon(mmer, m);
	int32_t hash_mini = orc_revhash32(mini);
	for (int i = 1; i <= k - m; i++) {
		seq >>= 2;
		mmer = (uint32_t)((uint64_t)seq & mask);
		mmer = orc_parity_canon(mmer, m);
		int32_t h = orc_revhash32(mmer);
		if (hash_mini > h) {
			mini = mmer;
			hash_mini = h;
		}
	}
	return mini;
}

/* ---- the index ------------------------------------------------------------------------------- */

typedef struct {
	KMER_T key;
	uint64_t rank;
} SUF(orc_fb_entry);

typedef struct {
	uint64_t nelem;
	uint64_t sizes[ORC_NLEVELS]; /* _hash_domains, bbhash.h:780 */
	uint64_t nwords;             /* words of the concatenated level bit array */
	uint64_t *bits;              /* bitVector::_bitArray, bbhash.h:502 */
	uint64_t nranks;
	uint64_t *ranks;             /* one sample per 16 words, bbhash.h:505 */
	uint64_t nfallback;          /* _final_hash (bbhash.h:781) as a sorted table */
	SUF(orc_fb_entry) *fallback;
	int levels_used;             /* number of levels processed (diagnostic) */
} SUF(orc_mphf);

typedef struct {
	int k, m, n, b;
	int gshift;
	uint32_t nbuckets, ngroups;
	uint64_t number_kmer, number_super_kmer, total_nt;
	uint64_t largest_mphf;
	uint32_t largest_bucket;
	uint32_t *bucket_nt;    /* bucket_minimizer::nuc_minimizer, blight.h:33 */
	uint64_t *bucket_start; /* bucket_minimizer::start,        blight.h:31 */
	uint64_t seq_words;
	uint64_t *seq; /* bucketSeq (blight.h:46) packed 2 bit/nt, nt p at bits 63-2(p%32)..62-2(p%32) of word p/32 */
	uint64_t *valid; /* Valid_kmer (blight.h:48) as one global bitmap over nt positions (build-time only) */
	uint64_t *g_nkeys;    /* per-group key count                                */
	uint64_t *g_prefix;   /* info_mphf::mphf_size after blight.cpp:299-301      */
	uint64_t *g_posstart; /* info_mphf::start (bits), blight.cpp:297             */
	uint8_t *g_nbits;     /* info_mphf::bit_to_encode, blight.cpp:293-296        */
	uint64_t positions_bits;
	uint64_t *positions; /* positions (blight.h:47), bit i = word i/64 bit i%64 like vector<bool> */
	SUF(orc_mphf) *mphf;
} SUF(orc_index);

/* get_kmer: k nt starting at GLOBAL nt position p.  blight.cpp:387-396 */
static inline KMER_T SUF(orc_get_kmer)(const SUF(orc_index) *ix, uint64_t p, int k)
{
	const uint64_t *s = ix->seq;
	uint64_t w = p >> 5;
	int sh = 2 * (int)(p & 31);
#if KMER_BITS == 64
	uint64_t v = sh ? ((s[w] << sh) | (s[w + 1] >> (64 - sh))) : s[w];
	return v >> (64 - 2 * k);
#else
	uint64_t a = s[w], b = s[w + 1], c = s[w + 2];
	uint64_t v0 = sh ? ((a << sh) | (b >> (64 - sh))) : a;
	uint64_t v1 = sh ? ((b << sh) | (c >> (64 - sh))) : b;
	KMER_T v = ((KMER_T)v0 << 64) | v1;
	return v >> (128 - 2 * k);
#endif
}

static inline unsigned SUF(orc_get_nt)(const SUF(orc_index) *ix, uint64_t p)
{
	return (unsigned)(ix->seq[p >> 5] >> (62 - 2 * (p & 31))) & 3u;
}

/* ---- BBHash ---------------------------------------------------------------------------------- */

/* level_finder_tmp ctor, bbhash.h:619-639.  Returns 1 if found (bit set in a level < max_level);
 * *bit / *level_bit are those of the level where the probe stopped. */
static inline int SUF(orc_level_find)(const SUF(orc_mphf) *mp, KMER_T key, unsigned max_level, uint64_t *bit,
                                       uint64_t *level_bit)
{
	uint64_t s0 = 0, s1 = 0;
	uint64_t bit_begin = 0;
	uint64_t k64 = (uint64_t)key; /* bbhash.h:160-170: 128-bit keys are hashed by their low 64 bits (SURVEY C-7) */
	for (unsigned level = 0; level < ORC_NLEVELS; level++) {
		uint64_t h = orc_level_hash_step(k64, &s0, &s1, level);
		uint64_t dom = mp->sizes[level];
		*level_bit = orc_mulhi64(h, dom);
		*bit = *level_bit + bit_begin;
		if (level >= max_level)
			return 0;
		if ((mp->bits[*bit >> 6] >> (*bit & 63)) & 1)
			return 1;
		bit_begin += dom;
	}
	return 0;
}

/* bitVector::rank, bbhash.h:467-480 */
static inline uint64_t SUF(orc_rank)(const SUF(orc_mphf) *mp, uint64_t pos)
{
	uint64_t word_idx = pos / 64, word_off = pos % 64;
	uint64_t block = word_idx / ORC_WORDS_PER_RANK;
	uint64_t r = mp->ranks[block];
	for (uint64_t w = block * ORC_WORDS_PER_RANK; w < word_idx; ++w)
		r += (uint64_t)__builtin_popcountll(mp->bits[w]);
	uint64_t mask = (1ull << word_off) - 1;
	r += (uint64_t)__builtin_popcountll(mp->bits[word_idx] & mask);
	return r;
}

/* bitVector::build_ranks, bbhash.h:447-465 */
static uint64_t SUF(orc_build_ranks)(SUF(orc_mphf) *mp, uint64_t upto)
{
	uint64_t max_idx = (upto + 63) / 64;
	mp->nranks = (max_idx + (ORC_WORDS_PER_RANK - 1)) / ORC_WORDS_PER_RANK;
	mp->ranks = (uint64_t *)calloc(mp->nranks ? mp->nranks : 1, sizeof(uint64_t));
	uint64_t cur = 0, ridx = 0;
	for (uint64_t ii = 0; ii < max_idx; ii++) {
		if (ii % ORC_WORDS_PER_RANK == 0)
			mp->ranks[ridx++] = cur;
		cur += (uint64_t)__builtin_popcountll(mp->bits[ii]);
	}
	return cur;
}

static int SUF(orc_fb_cmp)(const void *a, const void *b)
{
	const SUF(orc_fb_entry) *x = (const SUF(orc_fb_entry) *)a, *y = (const SUF(orc_fb_entry) *)b;
	return (x->key > y->key) - (x->key < y->key);
}

/* mphf::mphf(n, range, gamma=2), bbhash.h:531-559, processLevel_ :677-709, processFallbackLevel :711-728 */
static void SUF(orc_mphf_build)(SUF(orc_mphf) *mp, const KMER_T *keys, uint64_t n)
{
	memset(mp, 0, sizeof(*mp));
	mp->nelem = n;
	if (n == 0)
		return;
	uint64_t total = orc_configure_levels(n, mp->sizes);
	mp->nwords = (total + 63) / 64;
	mp->bits = (uint64_t *)calloc(mp->nwords, sizeof(uint64_t));
	uint64_t *accommodated = (uint64_t *)calloc((n + 63) / 64, sizeof(uint64_t));
	uint64_t *collisions = (uint64_t *)calloc((mp->sizes[0] + 63) / 64, sizeof(uint64_t));
	uint64_t level_offset = 0;
	for (unsigned level = 0;; level++) {
		mp->levels_used = (int)level + 1;
		if (level < ORC_NLEVELS) {
			uint64_t dom = mp->sizes[level];
			if (level > 0)
				memset(collisions, 0, (dom / 64) * sizeof(uint64_t));
			int did_collide = 0;
			for (uint64_t ki = 0; ki < n; ki++) {
				if (level >= 2 && ((accommodated[ki >> 6] >> (ki & 63)) & 1))
					continue;
				uint64_t bit, lbit;
				if (SUF(orc_level_find)(mp, keys[ki], level, &bit, &lbit)) {
					accommodated[ki >> 6] |= 1ull << (ki & 63);
				} else {
					/* level_finder::set -> bitVector::test_and_set, bbhash.h:643-650, 411-418 */
					uint64_t old = mp->bits[bit >> 6];
					mp->bits[bit >> 6] = old | (1ull << (bit & 63));
					if ((old >> (bit & 63)) & 1) {
						collisions[lbit >> 6] |= 1ull << (lbit & 63);
						did_collide = 1;
					}
				}
			}
			if (did_collide) {
				/* clearCollisions, bbhash.h:358-368 */
				uint64_t ids = level_offset / 64;
				for (uint64_t ii = 0; ii < dom / 64; ii++)
					mp->bits[ids + ii] &= ~collisions[ii];
			}
			level_offset += dom;
			if (did_collide)
				continue;
			SUF(orc_build_ranks)(mp, level_offset);
			break;
		} else {
			mp->levels_used = ORC_NLEVELS;
			uint64_t next_rank = SUF(orc_build_ranks)(mp, mp->nwords * 64);
			/* count first, then fill */
			uint64_t cnt = 0;
			for (uint64_t ki = 0; ki < n; ki++) {
				if ((accommodated[ki >> 6] >> (ki & 63)) & 1)
					continue;
				uint64_t bit, lbit;
				if (!SUF(orc_level_find)(mp, keys[ki], ORC_NLEVELS, &bit, &lbit))
					cnt++;
			}
			SUF(orc_fb_entry) *tmp = (SUF(orc_fb_entry) *)malloc((cnt ? cnt : 1) * sizeof(*tmp));
			uint64_t t = 0;
			for (uint64_t ki = 0; ki < n; ki++) {
				if ((accommodated[ki >> 6] >> (ki & 63)) & 1)
					continue;
				uint64_t bit, lbit;
				if (!SUF(orc_level_find)(mp, keys[ki], ORC_NLEVELS, &bit, &lbit)) {
					tmp[t].key = keys[ki];
					tmp[t].rank = next_rank++; /* _final_hash[val] = next_rank++  (bbhash.h:724) */
					t++;
				}
			}
			/* unordered_map semantics: a later assignment to an equal key overwrites.  Stable order by key,
			 * then keep the LAST entry of each run (ranks were assigned in input order, so last = largest). */
			qsort(tmp, cnt, sizeof(*tmp), SUF(orc_fb_cmp));
			uint64_t o = 0;
			for (uint64_t i = 0; i < cnt; i++) {
				if (o > 0 && tmp[o - 1].key == tmp[i].key) {
					if (tmp[i].rank > tmp[o - 1].rank)
						tmp[o - 1].rank = tmp[i].rank;
				} else {
					tmp[o++] = tmp[i];
				}
			}
			mp->nfallback = o;
			mp->fallback = tmp;
			break;
		}
	}
	free(accommodated);
	free(collisions);
}

/* mphf::lookup, bbhash.h:561-577 */
static inline uint64_t SUF(orc_mphf_lookup)(const SUF(orc_mphf) *mp, KMER_T key)
{
	uint64_t bit, lbit;
	if (SUF(orc_level_find)(mp, key, ORC_NLEVELS, &bit, &lbit))
		return SUF(orc_rank)(mp, bit);
	uint64_t lo = 0, hi = mp->nfallback;
	while (lo < hi) {
		uint64_t mid = (lo + hi) / 2;
		if (mp->fallback[mid].key < key)
			lo = mid + 1;
		else
			hi = mid;
	}
	if (lo < mp->nfallback && mp->fallback[lo].key == key)
		return mp->fallback[lo].rank;
	return ORC_NOT_FOUND;
}

/* ---- positions ------------------------------------------------------------------------------- */

/* int_to_bool, blight.cpp:464-469 (assignment, not OR: a later write to the same field replaces it) */
static inline void SUF(orc_pos_write)(SUF(orc_index) *ix, int nbits, uint64_t x, uint64_t h, uint64_t start)
{
	for (int i = 0; i < nbits; i++) {
		uint64_t bit = (uint64_t)i + h * (uint64_t)nbits + start;
		uint64_t m = 1ull << (bit & 63);
		if (x & 1)
			ix->positions[bit >> 6] |= m;
		else
			ix->positions[bit >> 6] &= ~m;
		x >>= 1;
	}
}

/* bool_to_int, blight.cpp:473-482 (result * 2^b in uint32_t arithmetic) */
static inline uint32_t SUF(orc_pos_read)(const SUF(orc_index) *ix, int nbits, uint64_t h, uint64_t start)
{
	uint32_t res = 0, acc = 1;
	uint64_t bit = h * (uint64_t)nbits + start;
	for (int i = 0; i < nbits; i++, acc <<= 1, bit++) {
		if ((ix->positions[bit >> 6] >> (bit & 63)) & 1)
			res |= acc;
	}
	return res << ix->b;
}

/* ---- construction ---------------------------------------------------------------------------- */

typedef struct {
	uint32_t unitig;
	uint32_t offset;
	uint32_t length;
	uint32_t mini;
} SUF(orc_superk);

typedef struct {
	SUF(orc_superk) *v;
	uint64_t n, cap;
} SUF(orc_superk_vec);

static void SUF(orc_sk_push)(SUF(orc_superk_vec) *sv, uint32_t unitig, uint32_t off, uint32_t len, uint32_t mini)
{
	if (sv->n == sv->cap) {
		sv->cap = sv->cap ? sv->cap * 2 : 1024;
		sv->v = (SUF(orc_superk) *)realloc(sv->v, sv->cap * sizeof(*sv->v));
	}
	sv->v[sv->n].unitig = unitig;
	sv->v[sv->n].offset = off;
	sv->v[sv->n].length = len;
	sv->v[sv->n].mini = mini;
	sv->n++;
}

/*
 * SuperKChopper (kmer.h:629-788) with patch P2: sliding-window minimum (window w = k-m+1 m-mers) of the signed
 * revhash of the parity-canonical m-mers, kept in a monotone deque; a super-k-mer is emitted whenever the VALUE of
 * the window minimum changes (kmer.h:666).  Deque rules restated from first_window (:704-732) and update_mini
 * (:734-781): pop from the back while back.hash >= new.hash, push, then expire the front when front.pos + w <= pos.
 * (KissRing's fixed capacity of 32, kmer.h:581, is not reproduced: SURVEY C-8.)
 */
static int SUF(orc_chop)(const char *s, uint64_t len, int k, int m, uint32_t unitig, SUF(orc_superk_vec) *out)
{
	const int w = k - m + 1;
	const uint32_t mmask = (m >= 16) ? 0xffffffffu : ((1u << (2 * m)) - 1);
	const int left = 2 * (m - 1);
	uint32_t fwd = 0, rev = 0;
	/* deque of (pos, canon, hash); at most w live entries */
	uint32_t *dq_pos = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(w + 1));
	uint32_t *dq_can = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(w + 1));
	int32_t *dq_h = (int32_t *)malloc(sizeof(int32_t) * (size_t)(w + 1));
	uint64_t head = 0, tail = 0; /* logical indices; storage index = idx % (w+1) */
	const uint64_t cap = (uint64_t)w + 1;
	int rc = 0;
	uint64_t n_mmers = len - (uint64_t)m + 1;
	uint64_t sk_first_kmer = 0; /* k-mer index where the current super-k-mer starts */
	uint32_t prev_mini = 0;
	for (uint64_t i = 0; i < len; i++) {
		int c = orc_nuc2int(s[i]);
		if (c < 0) {
			rc = ORC_EDNA;
			goto done;
		}
		/* SlidingKMer::_push_back, kmer.h:551-554 */
		fwd = ((fwd << 2) & mmask) | (uint32_t)c;
		rev = (rev >> 2) | ((uint32_t)(2 ^ c) << left);
		if (i + 1 < (uint64_t)m)
			continue;
		uint64_t pos = i + 1 - (uint64_t)m; /* m-mer index */
		uint32_t canon = ((__builtin_popcount(fwd) & 1) ? fwd : rev) >> 1; /* ParityCanonical::canonize_bidir, kmer.h:477-478 */
		int32_t h = orc_revhash32(canon);
		while (tail > head && dq_h[(tail - 1) % cap] >= h)
			tail--;
		dq_pos[tail % cap] = (uint32_t)pos;
		dq_can[tail % cap] = canon;
		dq_h[tail % cap] = h;
		tail++;
		if (pos + 1 < (uint64_t)w)
			continue; /* first window not complete yet */
		if ((uint64_t)dq_pos[head % cap] + (uint64_t)w <= pos)
			head++;
		uint64_t kmer_idx = pos + 1 - (uint64_t)w;
		uint32_t cur = dq_can[head % cap];
		if (kmer_idx == 0) {
			prev_mini = cur;
			sk_first_kmer = 0;
		} else if (cur != prev_mini) {
			/* SuperKChopper::next, kmer.h:666-679 */
			SUF(orc_sk_push)(out, unitig, (uint32_t)sk_first_kmer, (uint32_t)(kmer_idx - sk_first_kmer + (uint64_t)k - 1),
			                 prev_mini);
			prev_mini = cur;
			sk_first_kmer = kmer_idx;
		}
	}
	{
		uint64_t n_kmers = n_mmers - (uint64_t)w + 1;
		/* last super-k-mer, kmer.h:687-692 */
		SUF(orc_sk_push)(out, unitig, (uint32_t)sk_first_kmer, (uint32_t)(n_kmers - sk_first_kmer + (uint64_t)k - 1), prev_mini);
	}
done:
	free(dq_pos);
	free(dq_can);
	free(dq_h);
	return rc;
}

static void SUF(orc_free_index)(SUF(orc_index) *ix)
{
	if (!ix)
		return;
	if (ix->mphf) {
		for (uint32_t g = 0; g < ix->ngroups; g++) {
			free(ix->mphf[g].bits);
			free(ix->mphf[g].ranks);
			free(ix->mphf[g].fallback);
		}
	}
	free(ix->mphf);
	free(ix->bucket_nt);
	free(ix->bucket_start);
	free(ix->seq);
	free(ix->valid);
	free(ix->g_nkeys);
	free(ix->g_prefix);
	free(ix->g_posstart);
	free(ix->g_nbits);
	free(ix->positions);
	free(ix);
}

/* An index image exported from elsewhere (bench.py: the HIP build's arrays copied back from HBM): the big arrays stay the
 * caller's; ranks are rebuilt here (bitVector::build_ranks, bbhash.h:447-465), the fallback table keeps the LAST entry of every
 * run of equal keys (the map's last assignment wins, bbhash.h:724).  fb_keys: KMER_BITS / 64 little-endian words per key. */
static SUF(orc_index) *SUF(orc_import)(int k, int m, int n, int b, uint64_t total_nt, uint64_t number_kmer, const uint32_t *bucket_nt,
                                       const uint64_t *bucket_start, const uint64_t *seq, const uint64_t *positions, uint64_t positions_bits,
                                       const uint64_t *g_nkeys, const uint64_t *g_prefix, const uint64_t *g_posstart, const uint32_t *g_nbits,
                                       const uint64_t *level_sizes, const uint64_t *wbase, const uint64_t *bits, const uint64_t *fb_keys,
                                       const uint64_t *fb_ranks, const uint32_t *fb_group, uint64_t nfb)
{
	SUF(orc_index) *ix = (SUF(orc_index) *)calloc(1, sizeof(*ix));
	ix->k = k;
	ix->m = m;
	ix->n = n;
	ix->b = b;
	ix->gshift = 2 * m - 1 - n;
	ix->nbuckets = 1u << (2 * m - 1);
	ix->ngroups = 1u << n;
	ix->total_nt = total_nt;
	ix->number_kmer = number_kmer;
	ix->bucket_nt = (uint32_t *)bucket_nt;
	ix->bucket_start = (uint64_t *)bucket_start;
	ix->seq = (uint64_t *)seq;
	ix->positions = (uint64_t *)positions;
	ix->positions_bits = positions_bits;
	ix->g_nkeys = (uint64_t *)g_nkeys;
	ix->g_prefix = (uint64_t *)g_prefix;
	ix->g_posstart = (uint64_t *)g_posstart;
	ix->g_nbits = (uint8_t *)malloc(ix->ngroups);
	ix->mphf = (SUF(orc_mphf) *)calloc(ix->ngroups, sizeof(SUF(orc_mphf)));
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t g = 0; g < (int64_t)ix->ngroups; g++) {
		SUF(orc_mphf) *mp = &ix->mphf[g];
		ix->g_nbits[g] = (uint8_t)g_nbits[g];
		mp->nelem = g_nkeys[g];
		if (!mp->nelem)
			continue;
		memcpy(mp->sizes, level_sizes + (uint64_t)g * ORC_NLEVELS, sizeof(mp->sizes));
		mp->nwords = wbase[g + 1] - wbase[g];
		mp->bits = (uint64_t *)bits + wbase[g];
		SUF(orc_build_ranks)(mp, mp->nwords * 64);
	}
	const unsigned kw = KMER_BITS / 64;
	uint64_t f = 0;
	while (f < nfb) {
		const uint32_t g = fb_group[f];
		uint64_t e = f;
		while (e < nfb && fb_group[e] == g)
			e++;
		SUF(orc_mphf) *mp = &ix->mphf[g];
		mp->fallback = (SUF(orc_fb_entry) *)malloc((e - f) * sizeof(SUF(orc_fb_entry)));
		uint64_t o = 0;
		for (uint64_t i = f; i < e; i++) {
			if (i + 1 < e && memcmp(fb_keys + (i + 1) * kw, fb_keys + i * kw, 8 * kw) == 0)
				continue;
			KMER_T key = (KMER_T)fb_keys[i * kw];
#if KMER_BITS == 128
			key |= (KMER_T)fb_keys[i * kw + 1] << 64;
#endif
			mp->fallback[o].key = key;
			mp->fallback[o].rank = fb_ranks[i];
			o++;
		}
		mp->nfallback = o;
		f = e;
	}
	return ix;
}

static void SUF(orc_free_imported)(SUF(orc_index) *ix)
{
	if (!ix)
		return;
	for (uint32_t g = 0; g < ix->ngroups; g++) {
		free(ix->mphf[g].ranks);
		free(ix->mphf[g].fallback);
	}
	free(ix->mphf);
	free(ix->g_nbits);
	free(ix);
}

/*
 * construct_index (blight.cpp:108-125) = create_super_buckets_regular (:185-307) + read_super_buckets (:328-383),
 * at t = 1, without the gzip temp files (their only effect is the order in which super-k-mers reach str2bool:
 * bucket ascending, then unitig file order, then offset -- reproduced by the per-bucket cursors below).
 */
static int SUF(orc_build)(SUF(orc_index) **out, int k, int m, int n, int b, const char *concat, const uint64_t *offs,
                          uint64_t nseq)
{
	SUF(orc_index) *ix = (SUF(orc_index) *)calloc(1, sizeof(*ix));
	ix->k = k;
	ix->m = m;
	ix->n = n;
	ix->b = b;
	ix->gshift = 2 * m - 1 - n;
	ix->nbuckets = 1u << (2 * m - 1);
	ix->ngroups = 1u << n;
	ix->bucket_nt = (uint32_t *)calloc(ix->nbuckets, sizeof(uint32_t));
	ix->bucket_start = (uint64_t *)calloc(ix->nbuckets, sizeof(uint64_t));
	ix->g_nkeys = (uint64_t *)calloc(ix->ngroups, sizeof(uint64_t));
	ix->g_prefix = (uint64_t *)calloc(ix->ngroups, sizeof(uint64_t));
	ix->g_posstart = (uint64_t *)calloc(ix->ngroups, sizeof(uint64_t));
	ix->g_nbits = (uint8_t *)calloc(ix->ngroups, 1);
	ix->mphf = (SUF(orc_mphf) *)calloc(ix->ngroups, sizeof(SUF(orc_mphf)));

	/* phase 1: chop + count (blight.cpp:234-249, 259-277) */
	SUF(orc_superk_vec) sv = {0, 0, 0};
	int rc = 0;
	for (uint64_t u = 0; u < nseq && !rc; u++) {
		uint64_t len = offs[u + 1] - offs[u];
		if (len < (uint64_t)k) {
			rc = ORC_ESHORT;
			break;
		}
		rc = SUF(orc_chop)(concat + offs[u], len, k, m, (uint32_t)u, &sv);
	}
	if (rc) {
		free(sv.v);
		SUF(orc_free_index)(ix);
		return rc;
	}
	uint64_t total_nt = 0;
	for (uint64_t i = 0; i < sv.n; i++) {
		ix->bucket_nt[sv.v[i].mini] += sv.v[i].length;
		ix->g_nkeys[sv.v[i].mini >> ix->gshift] += sv.v[i].length - (uint32_t)k + 1;
		total_nt += sv.v[i].length;
	}
	ix->total_nt = total_nt;
	/* layout (blight.cpp:280-306) */
	{
		uint64_t i = 0, total_pos = 0, hash_base = 0;
		uint32_t max_bucket = 0;
		const uint32_t per = 1u << ix->gshift;
		for (uint32_t bc = 0; bc < ix->nbuckets; bc++) {
			ix->bucket_start[bc] = i;
			i += ix->bucket_nt[bc];
			if (ix->bucket_nt[bc] > max_bucket)
				max_bucket = ix->bucket_nt[bc];
			if (ix->bucket_nt[bc] > ix->largest_bucket)
				ix->largest_bucket = ix->bucket_nt[bc];
			if ((bc + 1) % per == 0) {
				uint32_t g = bc / per;
				int nb = (int)(ceil(log2((double)(uint32_t)(max_bucket + 1))) - (double)b);
				if (nb < 1)
					nb = 1;
				ix->g_nbits[g] = (uint8_t)nb;
				ix->g_posstart[g] = total_pos;
				total_pos += (uint64_t)nb * ix->g_nkeys[g] + 8; /* round_eight, blight.cpp:102-104 */
				ix->g_prefix[g] = hash_base;
				hash_base += ix->g_nkeys[g];
				if (ix->g_nkeys[g] > ix->largest_mphf)
					ix->largest_mphf = ix->g_nkeys[g];
				max_bucket = 0;
			}
		}
		ix->positions_bits = total_pos;
		ix->positions = (uint64_t *)calloc(total_pos / 64 + 2, sizeof(uint64_t));
	}
	/* phase 2a: str2bool (blight.cpp:311-324) in bucket / unitig / offset order */
	{
		uint64_t pad_nt = (1ull << b) + (uint64_t)k + 64;
		ix->seq_words = (total_nt + pad_nt) / 32 + 4;
		ix->seq = (uint64_t *)calloc(ix->seq_words, sizeof(uint64_t));
		ix->valid = (uint64_t *)calloc(total_nt / 64 + 2, sizeof(uint64_t));
		uint64_t *cursor = (uint64_t *)malloc(sizeof(uint64_t) * ix->nbuckets);
		memcpy(cursor, ix->bucket_start, sizeof(uint64_t) * ix->nbuckets);
		for (uint64_t i = 0; i < sv.n; i++) {
			const SUF(orc_superk) *sk = &sv.v[i];
			const char *s = concat + offs[sk->unitig] + sk->offset;
			uint64_t p0 = cursor[sk->mini];
			for (uint32_t j = 0; j < sk->length; j++) {
				uint64_t p = p0 + j;
				uint64_t c = (uint64_t)orc_nuc2int(s[j]);
				ix->seq[p >> 5] |= c << (62 - 2 * (p & 31));
				if (j + (uint32_t)k <= sk->length)
					ix->valid[p >> 6] |= 1ull << (p & 63);
			}
			cursor[sk->mini] = p0 + sk->length;
			ix->number_kmer += sk->length - (uint32_t)k + 1; /* blight.cpp:347 */
			ix->number_super_kmer++;                       /* blight.cpp:349 */
		}
		free(cursor);
	}
	free(sv.v);
	/* phase 2b: create_mphf (blight.cpp:421-460) + fill_positions (:486-519) per group */
	{
		const uint32_t per = 1u << ix->gshift;
		KMER_T *anchors = (KMER_T *)malloc(sizeof(KMER_T) * (ix->largest_mphf ? ix->largest_mphf : 1));
		for (uint32_t g = 0; g < ix->ngroups; g++) {
			uint64_t na = 0;
			for (uint32_t bc = g * per; bc < (g + 1) * per; bc++) {
				uint64_t st = ix->bucket_start[bc];
				uint32_t nt = ix->bucket_nt[bc];
				for (uint64_t j = 0; j + (uint64_t)k <= nt; j++) {
					uint64_t p = st + j;
					if ((ix->valid[p >> 6] >> (p & 63)) & 1)
						anchors[na++] = SUF(orc_canon)(SUF(orc_get_kmer)(ix, p, k), k);
				}
			}
			if (na == 0)
				continue; /* blight.cpp:452: no MPHF for an empty group */
			SUF(orc_mphf_build)(&ix->mphf[g], anchors, na);
			const int nbits = ix->g_nbits[g];
			for (uint32_t bc = g * per; bc < (g + 1) * per; bc++) {
				uint64_t st = ix->bucket_start[bc];
				uint32_t nt = ix->bucket_nt[bc];
				/* position 0 is never written (blight.cpp:493-494 starts at j+1); the array is zero-initialised */
				for (uint64_t j = 1; j + (uint64_t)k <= nt; j++) {
					uint64_t p = st + j;
					if ((ix->valid[p >> 6] >> (p & 63)) & 1) {
						KMER_T canon = SUF(orc_canon)(SUF(orc_get_kmer)(ix, p, k), k);
						SUF(orc_pos_write)(ix, nbits, j >> b, SUF(orc_mphf_lookup)(&ix->mphf[g], canon),
						                   ix->g_posstart[g]);
					}
				}
			}
		}
		free(anchors);
	}
	*out = ix;
	return 0;
}

/* ---- query ----------------------------------------------------------------------------------- */

/* query_get_hash (blight.cpp:716-742) when want_pos == 0, query_get_pos_unitig (:686-712) when want_pos == 1 */
static inline int64_t SUF(orc_query_one)(const SUF(orc_index) *ix, KMER_T canon, uint32_t mini, int want_pos)
{
	if (mini >= ix->nbuckets)
		return -1; /* the reference would read out of bounds; defined here as "absent" */
	if (ix->bucket_nt[mini] == 0)
		return -1;
	const uint32_t g = mini >> ix->gshift;
	const SUF(orc_mphf) *mp = &ix->mphf[g];
	uint64_t h = SUF(orc_mphf_lookup)(mp, canon);
	if (h == ORC_NOT_FOUND)
		return -1;
	uint64_t pos = SUF(orc_pos_read)(ix, ix->g_nbits[g], h, ix->g_posstart[g]);
	const int k = ix->k;
	if (pos + (uint64_t)k - 1 < ix->bucket_nt[mini]) {
		const uint64_t base = ix->bucket_start[mini];
		KMER_T seq = SUF(orc_get_kmer)(ix, base + pos, k);
		const KMER_T mask = SUF(orc_kmask)(k);
		const uint64_t P = 1ull << ix->b;
		for (uint64_t j = pos; j < pos + P; j++) {
			KMER_T r = SUF(orc_rc)(seq, k);
			KMER_T c = seq <= r ? seq : r;
			if (c == canon)
				return want_pos ? (int64_t)(int32_t)j : (int64_t)(h + ix->g_prefix[g]);
			/* update_kmer, blight.cpp:406-417 (reads past the bucket / array end are zero padding here) */
			seq = ((seq << 2) | (KMER_T)SUF(orc_get_nt)(ix, base + j + (uint64_t)k)) & mask;
		}
	}
	return -1;
}
