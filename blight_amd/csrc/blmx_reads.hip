// blmx_reads.hip -- gfx950 sequence / read front end of the query path.
//
// Reference: query_sequence_hash / query_sequence_bool (blight.cpp:554-591) roll a canonical k-mer along a string with
// updateK / updateRCK (blight.cpp:78-98) and hand every k-mer to query_kmer_hash, which recomputes the minimizer from
// scratch (minimizer_naive, kmer.h:791-810); file_query (blight.cpp:746-799) reads 2-line FASTA records on the host.
//
// Here a workgroup owns 512 consecutive character positions of a window that is resident in HBM (or in pinned host
// memory for a single short read) and works on them cooperatively:
//   * a lane encodes FOUR characters (one dword load) into one byte of the 2-bit packed words that the block keeps in LDS
//     (nuc2int / str2num, kmer.h:56-98); invalid characters set bits of an LDS bitmap;
//   * every lane hashes ONE m-mer (ParityCanonical kmer.h:475-487, signed revhash kmer.h:102-108) and the minimum over the
//     w = k-m+1 m-mers of each k-mer is assembled from power-of-two prefix minima in ceil(log2 w)+1 LDS rounds -- the data
//     parallel form of the chopper's monotone deque (SuperKChopper::update_mini, kmer.h:734-781); consecutive k-mers
//     share w-1 of their w m-mer hashes instead of each k-mer hashing its own window;
//   * the k-mer of every position is one funnel shift of the LDS words, canonical = min(x, rc(x)) (min_k, blight.cpp:86-91).
// Which positions start a k-mer (and where its ID goes) comes either from a VALID bitmap + its popcount prefix (windows of
// many sequences / raw FASTA bytes) or from a binary search in the sequence offsets (one call on a few sequences).
// The result feeds the locality-binned batch pipeline (k-mers + minimizers, blmx_query.hip) or, fused in the same kernel,
// the query body itself (query_device.h).
//
// Raw FASTA windows are parsed on the device too: newline bitmap (ballot), line table, the reference's record rule
// (blight.cpp:212-229 / 760-775 consumes lines in PAIRS and keeps the second line of a pair iff both are non-empty), VALID.
#include <algorithm>
#include <cstring>

#include <hip/hip_runtime.h>

#include "blmx_internal.h"
#include "scan_kernels.h"
#include "query_device.h"

namespace blmx {

constexpr int kFrontBlock = 512;  // positions per workgroup (8 waves)
constexpr int kFrontHalo = 64;    // k - 1 <= 62 trailing characters / w - 1 <= 62 trailing m-mers
constexpr int kFrontWords = (kFrontBlock + 2 * kFrontHalo) / 32 + 2;

template <int KW> struct FrontLane {
	Kmer<KW> x, xr;  // canonical k-mer and its reverse complement (x <= xr)
	uint32_t mini;
	bool bad;        // a character outside ACGTacgt inside the k-mer
};

// The cooperative part: every thread of the 512-thread block calls this; lane t gets the k-mer that starts at `base + t`
// (meaningful only where a k-mer really starts: the caller knows).  Characters at or beyond `total` read as 'A'.
template <int KW>
__device__ __forceinline__ void front_block(const uint8_t* __restrict__ bytes, uint64_t total, uint64_t base, int k, int m, FrontLane<KW>& out)
{
	__shared__ uint64_t s_src[kFrontWords];
	__shared__ uint64_t s_bad[kFrontWords / 2 + 2];
	__shared__ int32_t s_pow[kFrontBlock / 64][2][kFrontHalo + 64];  // per wave: two generations of prefix minima
	const int tid = (int)threadIdx.x;
	const int w = k - m + 1;
	// ---- characters -> 2-bit words + invalid-character bitmap: a lane encodes FOUR characters (one dword load) into one byte
	// of the packed words in LDS (nucleotide j of a word in bits 63-2j .. 62-2j: byte 7 - (j / 4) of the little-endian word).
	// (Assembling the words from wave ballots on the scalar unit was measured first: ~60 scalar instructions per 64 characters
	// made the kernel scalar-issue bound.)
	const int need = kFrontBlock + k - 1;  // characters that the block's k-mers cover
	const int nquads = (need + 3) / 4;
	uint8_t* s_src8 = reinterpret_cast<uint8_t*>(s_src);
	// word j of the invalid-character bitmap belongs to the loader lanes 16 j .. 16 j + 15 -- one wave --, so its first lane zeroes
	// it right before the others may OR into it (LDS operations of a wave execute in order: no barrier)
	if ((tid & 15) == 0 && (tid >> 4) < (int)(sizeof(s_bad) / 8))
		s_bad[tid >> 4] = 0;
	for (int q = nquads + tid; q < kFrontWords * 8; q += kFrontBlock)  // bytes behind the last character (funnel shifts read them)
		s_src8[(q & ~7) | (7 - (q & 7))] = 0;
	__builtin_amdgcn_wave_barrier();
	if (tid < nquads) {
		const uint64_t p = base + 4 * (uint64_t)tid;
		uint32_t v = 0x41414141u;  // 'A': positions at or beyond `total`
		if (p + 4 <= total) {
			v = *reinterpret_cast<const uint32_t*>(bytes + p);
		} else {
			for (int j = 0; j < 4; j++)
				if (p + j < total)
					v = (v & ~(0xffu << (8 * j))) | ((uint32_t)bytes[p + j] << (8 * j));
		}
		const uint32_t c0 = nuc_code((uint8_t)v), c1 = nuc_code((uint8_t)(v >> 8)), c2 = nuc_code((uint8_t)(v >> 16)), c3 = nuc_code((uint8_t)(v >> 24));
		s_src8[((tid >> 3) << 3) | (7 - (tid & 7))] = (uint8_t)(((c0 & 3u) << 6) | ((c1 & 3u) << 4) | ((c2 & 3u) << 2) | (c3 & 3u));
		const uint32_t bad4 = (c0 > 3 ? 1u : 0u) | (c1 > 3 ? 2u : 0u) | (c2 > 3 ? 4u : 0u) | (c3 > 3 ? 8u : 0u);
		if (bad4)  // rare: the bitmap starts zeroed
			atomicOr(reinterpret_cast<unsigned long long*>(&s_bad[tid >> 4]), (unsigned long long)bad4 << (4 * (tid & 15)));
	}
	__syncthreads();
	// ---- m-mer hashes and window minima, WAVE by wave: a wave owns 64 positions and hashes the 64 + w - 1 m-mers they need
	// itself (a third more hashes than sharing them across the block), so that the ceil(log2 w) + 1 rounds of prefix minima
	// need no workgroup barrier at all -- LDS operations of one wave execute in order.  (The block-wide version spent its
	// time in seven barriers per 512 positions.)
	// M_{p+1}[e] = min(M_p[e], M_p[e + 2^p]); the window of w m-mers is the disjoint union of the blocks named by the binary
	// digits of w.
	const int lane = tid & 63;
	int32_t(*pw)[kFrontHalo + 64] = s_pow[tid >> 6];
	const int nent = 64 + w - 1;
	for (int el = lane; el < nent; el += 64) {
		const int e = (tid & ~63) + el;
		const int wd = e >> 5, sh = 2 * (e & 31);
		const uint32_t f = (uint32_t)(funnel_left(s_src[wd], s_src[wd + 1], sh) >> (64 - 2 * m));
		const uint32_t r = (uint32_t)(rc_digits64((uint64_t)f) >> (64 - 2 * m));
		pw[0][el] = revhash32(((__popc(f) & 1) ? f : r) >> 1);
	}
	__builtin_amdgcn_wave_barrier();
	int32_t mn = 0x7fffffff;
	int cur = 0, off = 0;
	for (int p2 = 0; (1 << p2) <= w; p2++) {
		const int step = 1 << p2;
		if (w & step) {
			mn = min(mn, pw[cur][lane + off]);
			off += step;
		}
		if ((2 << p2) <= w) {  // another level is needed
			for (int el = lane; el + step < nent; el += 64)
				pw[cur ^ 1][el] = min(pw[cur][el], pw[cur][el + step]);
			__builtin_amdgcn_wave_barrier();
			cur ^= 1;
		}
	}
	out.mini = unrevhash32((uint32_t)mn);  // revhash is a bijection: the smallest hash names its m-mer
	// ---- the k-mer at this position: one funnel shift of the LDS words
	Kmer<KW> x;
	{
		const int wd = tid >> 5, sh = 2 * (tid & 31);
		if constexpr (KW == 1) {
			x.lo = funnel_left(s_src[wd], s_src[wd + 1], sh) >> (64 - 2 * k);
		} else {
			const uint64_t a = s_src[wd], b = s_src[wd + 1], c = s_src[wd + 2];
			const uint64_t v0 = funnel_left(a, b, sh), v1 = funnel_left(b, c, sh);
			const int s2 = 128 - 2 * k;
			if (s2 >= 64) {
				x.lo = v0 >> (s2 - 64);
				x.hi = 0;
			} else {
				x.lo = (v1 >> s2) | (v0 << (64 - s2));
				x.hi = v0 >> s2;
			}
		}
	}
	const Kmer<KW> r = kmer_rc<KW>(x, k);
	const bool fwd = x <= r;
	out.x = fwd ? x : r;
	out.xr = fwd ? r : x;
	{  // any invalid character among the k covered ones?
		const int wd = tid >> 6, sh = tid & 63;
		uint64_t v = s_bad[wd] >> sh;
		if (sh)
			v |= s_bad[wd + 1] << (64 - sh);
		out.bad = (v & ((k >= 64) ? ~0ull : ((1ull << k) - 1ull))) != 0;
	}
}

// Which positions start a k-mer and where their result goes.
//   bitmap mode : VALID bitmap (LSB first) + exclusive popcount prefix per 64-position word (windows, raw FASTA)
//   offsets mode: sequence s = [offs[s], offs[s+1]), its k-mers go to koffs[s] .. (few sequences: binary search per lane)
struct FrontWhere {
	const unsigned long long* valid;
	const uint32_t* vpre;
	const uint64_t* offs;
	const uint64_t* koffs;
	uint64_t nseq;
};

__device__ __forceinline__ bool front_where(const FrontWhere& fw, uint64_t q, uint64_t total, int k, uint64_t& o)
{
	if (fw.valid) {
		const uint64_t vw = fw.valid[q >> 6];
		const uint32_t lane = (uint32_t)q & 63u;
		o = (uint64_t)fw.vpre[q >> 6] + (uint64_t)__popcll(vw & ((1ull << lane) - 1ull));
		return (vw >> lane) & 1ull;
	}
	if (q >= total)
		return false;
	uint64_t lo = 0, hi = fw.nseq;  // last s with offs[s] <= q
	while (hi - lo > 1) {
		const uint64_t mid = (lo + hi) >> 1;
		if (fw.offs[mid] <= q)
			lo = mid;
		else
			hi = mid;
	}
	o = fw.koffs[lo] + (q - fw.offs[lo]);
	return q >= fw.offs[lo] && q + (uint64_t)k <= fw.offs[lo + 1];
}

// block-uniform: does any k-mer start in this block?  (bitmap mode only; offsets mode always says yes)
__device__ __forceinline__ bool front_block_live(const FrontWhere& fw, uint64_t base)
{
	if (!fw.valid)
		return true;
	return __syncthreads_or(fw.valid[(base >> 6) + (threadIdx.x >> 6)] != 0ull) != 0;
}

// front end alone: canonical k-mers + their minimizers, compacted in position order
template <int KW>
__global__ __launch_bounds__(kFrontBlock) void k_read_front(const uint8_t* __restrict__ bytes, uint64_t total, int k, int m, FrontWhere fw,
                                                            uint64_t* __restrict__ kmers, uint32_t* __restrict__ minis, int* __restrict__ err)
{
	const uint64_t base = (uint64_t)blockIdx.x * kFrontBlock;
	if (!front_block_live(fw, base))
		return;
	FrontLane<KW> f;
	front_block<KW>(bytes, total, base, k, m, f);
	uint64_t o;
	if (!front_where(fw, base + threadIdx.x, total, k, o))
		return;
	if (f.bad)
		*err = 1;  // plain store: the flag may live in pinned host memory and every writer stores the same value
	if constexpr (KW == 1) {
		kmers[o] = f.x.lo;
	} else {
		kmers[2 * o] = f.x.lo;
		kmers[2 * o + 1] = f.x.hi;
	}
	minis[o] = f.mini;
}

// front end + query body in one launch, IDs in position order (query_sequence_hash, blight.cpp:575-591).
// counts (may be null): kCountStripes pairs (found, not found); a block adds to ONE pair (query_sequence_bool, blight.cpp:554-571).
template <int KW, int LOGP>
__global__ __launch_bounds__(kFrontBlock) void k_read_query(DeviceIndexView v, const uint8_t* __restrict__ bytes, uint64_t total, FrontWhere fw,
                                                            int64_t* __restrict__ ids, unsigned long long* __restrict__ counts,
                                                            int* __restrict__ err, DoneSignal done)
{
	const uint64_t base = (uint64_t)blockIdx.x * kFrontBlock;
	if (!front_block_live(fw, base))  // (bitmap mode only; launches with a completion signal use the offsets mode: every workgroup stays)
		return;
	FrontLane<KW> f;
	front_block<KW>(bytes, total, base, v.k, v.m, f);
	uint64_t o;
	const bool valid = front_where(fw, base + threadIdx.x, total, v.k, o);
	int64_t id = -1;
	if (valid) {
		if (f.bad)
			*err = 1;
		int32_t posj;
		query_one<KW, LOGP, false>(v, f.x, f.xr, f.mini, id, posj);
		if (ids)
			ids[o] = id;
	}
	if (counts) {
		__shared__ unsigned int s_found, s_absent;
		if (threadIdx.x == 0)
			s_found = s_absent = 0;
		__syncthreads();
		const unsigned long long bv = __ballot(valid), bf = __ballot(valid && id >= 0);
		if ((threadIdx.x & 63) == 0) {
			if (bf)
				atomicAdd(&s_found, (unsigned int)__popcll(bf));
			if (bv != bf)
				atomicAdd(&s_absent, (unsigned int)(__popcll(bv) - __popcll(bf)));
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			const unsigned stripe = blockIdx.x % kCountStripes;
			if (s_found)
				atomicAdd(&counts[2 * stripe], (unsigned long long)s_found);
			if (s_absent)
				atomicAdd(&counts[2 * stripe + 1], (unsigned long long)s_absent);
		}
	}
	if (done.flag) {  // small launches: the last workgroup that has stored its answers tells the host
		__threadfence_system();
		__syncthreads();
		if (threadIdx.x == 0 && (!done.counter || atomicAdd(done.counter, 1u) + 1u == done.target))  // (nullptr: the only workgroup of the round)
			*done.flag = done.ticket;
	}
}

unsigned int read_query_signal_units(uint64_t npos) { return (unsigned int)((npos + kFrontBlock - 1) / kFrontBlock); }

// ---- VALID bitmap of a window of concatenated sequences --------------------------------------------------------------------
// starts: bit p set iff a sequence starts at window position p.  A k-mer starts at q iff q + k <= total and no sequence
// starts inside (q, q + k - 1].
__global__ void k_mark_seq_starts(const uint64_t* __restrict__ offs, uint64_t nseq, uint64_t origin, uint64_t span,
                                  unsigned long long* __restrict__ starts)
{
	const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseq)
		return;
	const uint64_t a = offs[s];
	if (a >= origin && a - origin < span)
		atomicOr(&starts[(a - origin) >> 6], 1ull << ((a - origin) & 63));
}

__global__ __launch_bounds__(256) void k_valid_from_starts(const unsigned long long* __restrict__ starts, uint64_t npos, uint64_t total, int k,
                                                           unsigned long long* __restrict__ valid)
{
	const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	bool ok = false;
	if (q < npos && q + (uint64_t)k <= total) {
		const uint64_t p = q + 1, wd = p >> 6;
		const uint32_t sh = (uint32_t)p & 63u;
		const uint64_t lo = starts[wd], hi = starts[wd + 1];
		const uint64_t win = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
		ok = (win & ((k - 1 >= 64) ? ~0ull : ((1ull << (k - 1)) - 1ull))) == 0;
	}
	const unsigned long long bal = __ballot(ok);
	if ((threadIdx.x & 63) == 0)
		valid[q >> 6] = bal;
}

// ---- raw FASTA bytes: newline bitmap, line table, record rule, VALID ----------------------------------------------------------
// exact per-byte test "byte == c" of the four bytes of x -> bits 0..3
__device__ __forceinline__ uint32_t bytes_equal_mask(uint32_t x, uint32_t c4)
{
	const uint32_t t = x ^ c4;
	const uint32_t z = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);  // 0x80 in every byte of t that is zero
	return ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
}

// one lane per 16 bytes (one dwordx4 load), four lanes assemble one 64-position word of the bitmap
__global__ __launch_bounds__(256) void k_raw_newlines(const uint8_t* __restrict__ bytes, uint64_t total, uint64_t nwords,
                                                      unsigned long long* __restrict__ nl)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t p = i * 16;
	uint32_t m = 0;
	if (p + 16 <= total && (reinterpret_cast<uintptr_t>(bytes) & 15) == 0) {
		const uint4 v = *reinterpret_cast<const uint4*>(bytes + p);
		const uint32_t c4 = 0x0a0a0a0au;
		m = bytes_equal_mask(v.x, c4) | (bytes_equal_mask(v.y, c4) << 4) | (bytes_equal_mask(v.z, c4) << 8) | (bytes_equal_mask(v.w, c4) << 12);
	} else {
		for (int j = 0; j < 16; j++)
			if (p + j < total && bytes[p + j] == (uint8_t)'\n')
				m |= 1u << j;
	}
	const uint32_t m1 = __shfl_down(m, 1), m2 = __shfl_down(m, 2), m3 = __shfl_down(m, 3);
	if ((threadIdx.x & 3) == 0 && (i >> 2) < nwords)
		nl[i >> 2] = (unsigned long long)m | ((unsigned long long)m1 << 16) | ((unsigned long long)m2 << 32) | ((unsigned long long)m3 << 48);
}

// nlpos[j] = position of the j-th newline of the window; nlpos[number of newlines] = total (end of the unterminated last line)
__global__ void k_line_table(const unsigned long long* __restrict__ nl, const uint32_t* __restrict__ nlpre, uint64_t nwords, uint64_t total,
                             uint32_t* __restrict__ nlpos)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w > nwords)
		return;
	if (w == nwords) {
		nlpos[nlpre[nwords]] = (uint32_t)total;
		return;
	}
	unsigned long long b = nl[w];
	uint32_t o = nlpre[w];
	while (b) {
		nlpos[o++] = (uint32_t)(w * 64 + (uint64_t)(__ffsll(b) - 1));
		b &= b - 1;
	}
}

// carry[0] = lines of the file before this window, carry[1] = length of the last of them (a window ends behind a newline)
// mode 0: 2-line records (blight.cpp:212-229 / 760-775): lines are consumed in pairs; the second line of a pair is a sequence
//         iff both lines are non-empty; sequences shorter than k are skipped (blight.cpp:782), or flagged (construction)
// mode 1: every line that starts with A/C/G/T (Colored_De_Bruijn_graph_snippet.cpp:139) and holds at least k characters
__global__ void k_line_ok(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ nlpos, const uint32_t* __restrict__ nlpre, uint64_t nwords,
                          const uint64_t* __restrict__ carry_in, uint64_t* __restrict__ carry_out, int k, int mode, uint8_t* __restrict__ ok,
                          int* __restrict__ short_flag)
{
	const uint64_t nlines = (uint64_t)nlpre[nwords] + 1;  // the last one is unterminated (possibly empty)
	for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nlines; j += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t start = j ? (uint64_t)nlpos[j - 1] + 1 : 0, len = (uint64_t)nlpos[j] - start;
		bool good;
		if (mode == 0) {
			const uint64_t prev_len = j == 0 ? carry_in[1] : (start - 1) - (j >= 2 ? (uint64_t)nlpos[j - 2] + 1 : 0);
			const bool second = ((carry_in[0] + j) & 1ull) != 0;
			good = second && prev_len > 0 && len >= (uint64_t)k;
			if (second && prev_len > 0 && len > 0 && len < (uint64_t)k && short_flag)
				*short_flag = 1;
		} else {
			const uint8_t c = len ? bytes[start] : 0;
			good = len >= (uint64_t)k && (c == 'A' || c == 'C' || c == 'G' || c == 'T');
		}
		ok[j] = good ? 1 : 0;
		if (j + 1 == nlines) {  // complete lines of this window: all but the unterminated one
			carry_out[0] = carry_in[0] + (nlines - 1);
			carry_out[1] = nlines >= 2 ? (uint64_t)nlpos[j - 1] - (j >= 2 ? (uint64_t)nlpos[j - 2] + 1 : 0) : carry_in[1];
		}
	}
}

__global__ void k_valid_lines(const unsigned long long* __restrict__ nl, const uint32_t* __restrict__ nlpre, const uint32_t* __restrict__ nlpos,
                              const uint8_t* __restrict__ ok, uint64_t nwords, int k, unsigned long long* __restrict__ valid)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords)
		return;
	unsigned long long bits = nl[w];
	uint32_t j = nlpre[w];
	int a = 0;
	unsigned long long out = 0;
	for (;;) {
		const int e = bits ? __ffsll(bits) - 1 : 64;  // local position of the newline that ends line j, if it is in this word
		if (ok[j]) {
			// k-mer starts q with q + k <= end of line j
			const int64_t lim = (int64_t)nlpos[j] - (int64_t)k - (int64_t)(w * 64) + 1;
			const int hi = (int)min((int64_t)e, lim);
			if (hi > a)
				out |= ((hi >= 64) ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << a) - 1ull);
		}
		if (e >= 64)
			break;
		bits &= bits - 1;
		a = e + 1;
		j++;
		if (a >= 64)
			break;
	}
	valid[w] = out;
}

// ---- raw FASTA bytes -> the sequences themselves (construction: construct_index(file), reference blight.cpp:212-229) ---------------
// Behind the line table and the record rule above (k_line_ok, mode 0: the second line of a pair is a sequence iff both lines are
// non-empty; a sequence shorter than k raises the short flag), the sequence lines of a window are appended to the text the build
// takes -- characters back to back + one offset per sequence -- entirely on the device: the host only moves the file's bytes.
// Two bitmaps per window, both one lane per 64-byte word like k_valid_lines: mask (bit i = byte i belongs to a sequence line) and
// ends (bit e = the newline at e ends a sequence line); their popcount prefixes give, for every byte, the characters and the sequences
// appended before it -- no per-line scan, and no size the host would have to read back.
__global__ void k_seq_masks(const unsigned long long* __restrict__ nl, const uint32_t* __restrict__ nlpre, const uint8_t* __restrict__ ok, uint64_t nwords,
                            uint64_t total, unsigned long long* __restrict__ mask, unsigned long long* __restrict__ ends)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords)
		return;
	unsigned long long bits = nl[w];
	uint32_t j = nlpre[w];
	int a = 0;
	unsigned long long out = 0, en = 0;
	for (;;) {
		const int e = bits ? __ffsll(bits) - 1 : 64;  // local position of the newline that ends line j, if it is in this word
		if (ok[j]) {
			if (e > a)
				out |= ((e >= 64) ? ~0ull : ((1ull << e) - 1ull)) & ~((1ull << a) - 1ull);
			if (e < 64)
				en |= 1ull << e;
		}
		if (e >= 64)
			break;
		bits &= bits - 1;
		a = e + 1;
		j++;
		if (a >= 64)
			break;
	}
	if (total - w * 64 < 64)  // the last word: nothing behind the end of the window (an unterminated last line runs up to it)
		out &= (1ull << (total - w * 64)) - 1ull;
	mask[w] = out;
	ends[w] = en;
}
// tot[0] = sequences, tot[1] = characters appended before this window (device-resident: no host round trip per window).
// One lane per word writes the end offset of every sequence line that ends in it.
__global__ void k_seq_offsets(const unsigned long long* __restrict__ mask, const uint32_t* __restrict__ mpre, const unsigned long long* __restrict__ ends,
                              const uint32_t* __restrict__ epre, uint64_t nwords, const uint64_t* __restrict__ tot, uint64_t* __restrict__ offs_out)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords)
		return;
	unsigned long long en = ends[w];
	const unsigned long long m = mask[w];
	uint64_t idx = tot[0] + epre[w];
	while (en) {
		const int e = __ffsll(en) - 1;
		offs_out[++idx] = tot[1] + (uint64_t)mpre[w] + (uint64_t)__popcll(m & ((1ull << e) - 1ull));
		en &= en - 1;
	}
}
// the unterminated last line of the file, when it is a sequence, and the new totals
__global__ void k_seq_advance(const uint8_t* __restrict__ ok, const uint32_t* __restrict__ nlpre, const uint32_t* __restrict__ mpre,
                              const uint32_t* __restrict__ epre, uint64_t nwords, uint64_t* __restrict__ tot, uint64_t* __restrict__ offs_out)
{
	const uint64_t nseq = tot[0] + epre[nwords], nchar = tot[1] + mpre[nwords];
	const uint32_t last = nlpre[nwords];  // the line behind the window's last newline (empty when the window ends behind a newline)
	uint64_t add = 0;
	if (ok[last]) {
		offs_out[nseq + 1] = nchar;
		add = 1;
	}
	tot[0] = nseq + add;
	tot[1] = nchar;
}
// one lane per byte: consecutive lanes read consecutive bytes and -- inside a sequence line -- write consecutive bytes
__global__ __launch_bounds__(256) void k_seq_copy(const uint8_t* __restrict__ bytes, uint64_t total, const unsigned long long* __restrict__ mask,
                                                  const uint32_t* __restrict__ mpre, const uint64_t* __restrict__ tot, uint8_t* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= total)
		return;
	const unsigned long long m = mask[i >> 6];
	const uint32_t lane = (uint32_t)i & 63u;
	if ((m >> lane) & 1ull)  // tot == nullptr: the window's own sequence characters, from 0
		out[(tot ? tot[1] : 0ull) + (uint64_t)mpre[i >> 6] + (uint64_t)__popcll(m & ((1ull << lane) - 1ull))] = bytes[i];
}
// The window's sequence characters (`chars`, *count of them) -> 2-bit codes appended to the packed text at nucleotide tot[1]: one lane
// per 64-bit word of the packed text that the window touches (first nucleotide most significant, like k_encode_pack of the build);
// the first and last word are shared with the neighbouring windows, so every word is ORed into the zeroed array.  A character
// outside ACGTacgt raises *err (nuc2int throws, kmer.h:56-69).
__global__ __launch_bounds__(256) void k_seq_pack(const uint8_t* __restrict__ chars, const uint32_t* __restrict__ count, const uint64_t* __restrict__ tot,
                                                  unsigned long long* __restrict__ packed, int* __restrict__ err)
{
	const uint64_t base = tot[1], n = *count;
	const uint64_t gw = (base >> 5) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (n == 0 || gw > (base + n - 1) >> 5)
		return;
	const uint64_t a = max(base, gw * 32), z = min(base + n, gw * 32 + 32);
	uint64_t word = 0;
	bool bad = false;
	for (uint64_t p = a; p < z; p++) {
		const uint32_t c = nuc_code(chars[p - base]);
		bad |= c > 3;
		word |= (uint64_t)(c & 3u) << (62 - 2 * (p & 31));
	}
	if (bad)
		*err = 1;
	atomicOr(&packed[gw], (unsigned long long)word);
}


// ---- launchers ---------------------------------------------------------------------------------------------------------------
namespace {
inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)std::max<uint64_t>(1, (n + block - 1) / block); }
// a 512-position workgroup reads 8 words of VALID / vpre whatever the window length: the bitmaps are padded with zero words
constexpr uint64_t kBitmapPad = kFrontBlock / 64 + 2;

void ensure_scratch(DevBuf& b, size_t bytes, hipStream_t s)
{
	if (bytes > b.bytes)
		BLMX_HIP(b.alloc(bytes + bytes / 8 + 256, false, s));
}

// exclusive prefix of the popcounts of `words` bitmap words (n + 1 outputs: the last one is the total)
void popcount_prefix(FrontScratch& fs, const unsigned long long* d_bits, uint64_t n, uint32_t* d_pre, hipStream_t s)
{
	ensure_scratch(fs.tmp, scan_sums_words(n + 1) * 4, s);
	exclusive_scan_of<uint32_t>(ScanPopc64<uint32_t>{d_bits}, d_pre, n + 1, fs.tmp.as<uint32_t>(), s);
}
}  // namespace

void front_valid_concat(const Index& ix, FrontScratch& fs, const uint64_t* d_offs, uint64_t nseq, uint64_t origin, uint64_t npos, uint64_t nbytes,
                        hipStream_t s)
{
	const uint64_t nwords = (npos + 63) / 64;
	const uint64_t swords = (nbytes + 64) / 64 + 2;
	ensure_scratch(fs.marks, swords * 8, s);
	ensure_scratch(fs.valid, (nwords + kBitmapPad) * 8, s);
	ensure_scratch(fs.vpre, (nwords + kBitmapPad) * 4, s);
	BLMX_HIP(hipMemsetAsync(fs.marks.p, 0, swords * 8, s));
	BLMX_HIP(hipMemsetAsync((char*)fs.valid.p + nwords * 8, 0, kBitmapPad * 8, s));
	if (nseq)
		hipLaunchKernelGGL(k_mark_seq_starts, dim3(grid_for(nseq, 256)), dim3(256), 0, s, d_offs, nseq, origin, nbytes + 1,
		                   fs.marks.as<unsigned long long>());
	hipLaunchKernelGGL(k_valid_from_starts, dim3(grid_for(nwords * 64, 256)), dim3(256), 0, s, fs.marks.as<unsigned long long>(), npos, nbytes,
	                   (int)ix.prm.k, fs.valid.as<unsigned long long>());
	BLMX_HIP(hipGetLastError());
	popcount_prefix(fs, fs.valid.as<unsigned long long>(), nwords, fs.vpre.as<uint32_t>(), s);
}

void front_valid_raw(const Index& ix, FrontScratch& fs, const uint8_t* d_bytes, uint64_t nbytes, int mode, const uint64_t* d_carry_in,
                     uint64_t* d_carry_out, int* d_short_flag, hipStream_t s)
{
	const uint64_t nwords = (nbytes + 63) / 64;
	ensure_scratch(fs.marks, (nwords + 2) * 8, s);
	ensure_scratch(fs.mpre, (nwords + 2) * 4, s);
	ensure_scratch(fs.valid, (nwords + kBitmapPad) * 8, s);
	ensure_scratch(fs.vpre, (nwords + kBitmapPad) * 4, s);
	ensure_scratch(fs.nlpos, (nbytes + 2) * 4, s);  // at most one line per byte
	ensure_scratch(fs.ok, nbytes + 2, s);
	BLMX_HIP(hipMemsetAsync((char*)fs.marks.p + nwords * 8, 0, 16, s));
	BLMX_HIP(hipMemsetAsync((char*)fs.valid.p + nwords * 8, 0, kBitmapPad * 8, s));
	hipLaunchKernelGGL(k_raw_newlines, dim3(grid_for(nwords * 4, 256)), dim3(256), 0, s, d_bytes, nbytes, nwords, fs.marks.as<unsigned long long>());
	BLMX_HIP(hipGetLastError());
	popcount_prefix(fs, fs.marks.as<unsigned long long>(), nwords, fs.mpre.as<uint32_t>(), s);
	hipLaunchKernelGGL(k_line_table, dim3(grid_for(nwords + 1, 256)), dim3(256), 0, s, fs.marks.as<unsigned long long>(), fs.mpre.as<uint32_t>(), nwords,
	                   nbytes, fs.nlpos.as<uint32_t>());
	hipLaunchKernelGGL(k_line_ok, dim3(1024), dim3(256), 0, s, d_bytes, fs.nlpos.as<uint32_t>(), fs.mpre.as<uint32_t>(), nwords, d_carry_in, d_carry_out,
	                   (int)ix.prm.k, mode, fs.ok.as<uint8_t>(), d_short_flag);
	hipLaunchKernelGGL(k_valid_lines, dim3(grid_for(nwords, 256)), dim3(256), 0, s, fs.marks.as<unsigned long long>(), fs.mpre.as<uint32_t>(),
	                   fs.nlpos.as<uint32_t>(), fs.ok.as<uint8_t>(), nwords, (int)ix.prm.k, fs.valid.as<unsigned long long>());
	BLMX_HIP(hipGetLastError());
	popcount_prefix(fs, fs.valid.as<unsigned long long>(), nwords, fs.vpre.as<uint32_t>(), s);
}

// The sequences of a raw FASTA window (cut behind a newline unless it is the last one) appended to d_text / d_offs at the running
// totals d_tot = [sequences, characters] (device-resident, advanced here).  d_carry as in front_valid_raw.  *d_short_flag is raised by
// a sequence shorter than k.  The caller guarantees room for nbytes more characters and nbytes / (k + 1) + 1 more offsets.
// The text is kept as characters (d_text) or -- d_packed != nullptr -- as the 2-bit words the build works on (a quarter of the bytes,
// zeroed by the caller; *d_dna_flag is raised by a character outside ACGTacgt).
void front_seq_lines(const Index& ix, FrontScratch& fs, const uint8_t* d_bytes, uint64_t nbytes, const uint64_t* d_carry_in, uint64_t* d_carry_out,
                     int* d_short_flag, uint64_t* d_tot, uint8_t* d_text, uint64_t* d_packed, int* d_dna_flag, uint64_t* d_offs, hipStream_t s)
{
	const uint64_t nwords = (nbytes + 63) / 64;
	ensure_scratch(fs.marks, (nwords + 2) * 8, s);
	ensure_scratch(fs.mpre, (nwords + 2) * 4, s);
	ensure_scratch(fs.valid, (nwords + kBitmapPad) * 8, s);
	ensure_scratch(fs.vpre, (nwords + kBitmapPad) * 4, s);
	ensure_scratch(fs.nlpos, (nbytes + 2) * 4, s);  // at most one line per byte
	ensure_scratch(fs.ok, nbytes + 2, s);
	BLMX_HIP(hipMemsetAsync((char*)fs.marks.p + nwords * 8, 0, 16, s));
	hipLaunchKernelGGL(k_raw_newlines, dim3(grid_for(nwords * 4, 256)), dim3(256), 0, s, d_bytes, nbytes, nwords, fs.marks.as<unsigned long long>());
	BLMX_HIP(hipGetLastError());
	popcount_prefix(fs, fs.marks.as<unsigned long long>(), nwords, fs.mpre.as<uint32_t>(), s);
	hipLaunchKernelGGL(k_line_table, dim3(grid_for(nwords + 1, 256)), dim3(256), 0, s, fs.marks.as<unsigned long long>(), fs.mpre.as<uint32_t>(), nwords,
	                   nbytes, fs.nlpos.as<uint32_t>());
	hipLaunchKernelGGL(k_line_ok, dim3(1024), dim3(256), 0, s, d_bytes, fs.nlpos.as<uint32_t>(), fs.mpre.as<uint32_t>(), nwords, d_carry_in, d_carry_out,
	                   (int)ix.prm.k, 0, fs.ok.as<uint8_t>(), d_short_flag);
	BLMX_HIP(hipGetLastError());
	// (the number of lines is only known on the device: everything below runs over the window's 64-byte words)
	ensure_scratch(fs.lpre, (nwords + 2) * 8 + (nwords + 2) * 4, s);
	unsigned long long* d_ends = fs.lpre.as<unsigned long long>();
	uint32_t* d_epre = reinterpret_cast<uint32_t*>(d_ends + nwords + 2);
	hipLaunchKernelGGL(k_seq_masks, dim3(grid_for(nwords, 256)), dim3(256), 0, s, fs.marks.as<unsigned long long>(), fs.mpre.as<uint32_t>(), fs.ok.as<uint8_t>(),
	                   nwords, nbytes, fs.valid.as<unsigned long long>(), d_ends);
	BLMX_HIP(hipGetLastError());
	popcount_prefix(fs, fs.valid.as<unsigned long long>(), nwords, fs.vpre.as<uint32_t>(), s);
	popcount_prefix(fs, d_ends, nwords, d_epre, s);
	hipLaunchKernelGGL(k_seq_offsets, dim3(grid_for(nwords, 256)), dim3(256), 0, s, fs.valid.as<unsigned long long>(), fs.vpre.as<uint32_t>(), d_ends, d_epre, nwords,
	                   d_tot, d_offs);
	if (d_packed) {  // 2-bit text: the window's characters are compacted into a window-sized stage first, then packed at the running total
		ensure_scratch(fs.stage, nbytes + 64, s);
		hipLaunchKernelGGL(k_seq_copy, dim3(grid_for(nbytes, 256)), dim3(256), 0, s, d_bytes, nbytes, fs.valid.as<unsigned long long>(), fs.vpre.as<uint32_t>(),
		                   (const uint64_t*)nullptr, fs.stage.as<uint8_t>());
		hipLaunchKernelGGL(k_seq_pack, dim3(grid_for(nbytes / 32 + 2, 256)), dim3(256), 0, s, fs.stage.as<uint8_t>(), fs.vpre.as<uint32_t>() + nwords, d_tot,
		                   reinterpret_cast<unsigned long long*>(d_packed), d_dna_flag);
	} else {
		hipLaunchKernelGGL(k_seq_copy, dim3(grid_for(nbytes, 256)), dim3(256), 0, s, d_bytes, nbytes, fs.valid.as<unsigned long long>(), fs.vpre.as<uint32_t>(), d_tot,
		                   d_text);
	}
	hipLaunchKernelGGL(k_seq_advance, dim3(1), dim3(1), 0, s, fs.ok.as<uint8_t>(), fs.mpre.as<uint32_t>(), fs.vpre.as<uint32_t>(), d_epre, nwords, d_tot, d_offs);
	BLMX_HIP(hipGetLastError());
}

static FrontWhere where_of(const FrontScratch* fs, const uint64_t* offs, const uint64_t* koffs, uint64_t nseq)
{
	FrontWhere fw;
	fw.valid = fs ? fs->valid.as<unsigned long long>() : nullptr;
	fw.vpre = fs ? fs->vpre.as<uint32_t>() : nullptr;
	fw.offs = offs;
	fw.koffs = koffs;
	fw.nseq = nseq;
	return fw;
}

void launch_read_front(const Index& ix, const FrontScratch* fs, const uint8_t* d_bytes, uint64_t nbytes, uint64_t npos, const uint64_t* offs,
                       const uint64_t* koffs, uint64_t nseq, void* d_kmers, uint32_t* d_minis, int* d_err, hipStream_t s)
{
	if (npos == 0)
		return;
	const FrontWhere fw = where_of(fs, offs, koffs, nseq);
	const unsigned grid = grid_for(npos, kFrontBlock);
	if (ix.KW == 1)
		hipLaunchKernelGGL((k_read_front<1>), dim3(grid), dim3(kFrontBlock), 0, s, d_bytes, nbytes, (int)ix.prm.k, (int)ix.prm.m, fw,
		                   reinterpret_cast<uint64_t*>(d_kmers), d_minis, d_err);
	else
		hipLaunchKernelGGL((k_read_front<2>), dim3(grid), dim3(kFrontBlock), 0, s, d_bytes, nbytes, (int)ix.prm.k, (int)ix.prm.m, fw,
		                   reinterpret_cast<uint64_t*>(d_kmers), d_minis, d_err);
	BLMX_HIP(hipGetLastError());
}

template <int KW>
static void launch_read_query_kw(const Index& ix, const FrontWhere& fw, const uint8_t* d_bytes, uint64_t nbytes, uint64_t npos, int64_t* d_ids,
                                 unsigned long long* d_counts, int* d_err, hipStream_t s, DoneSignal done)
{
	const DeviceIndexView v = ix.view();
	const unsigned grid = grid_for(npos, kFrontBlock);
#define BLMX_RQ(LP)                                                                                                                      \
	case LP:                                                                                                                             \
		hipLaunchKernelGGL((k_read_query<KW, LP>), dim3(grid), dim3(kFrontBlock), 0, s, v, d_bytes, nbytes, fw, d_ids, d_counts, d_err, done); \
		break;
	switch (v.b) {
		BLMX_RQ(0)
		BLMX_RQ(1)
		BLMX_RQ(2)
		BLMX_RQ(3)
		BLMX_RQ(4)
		BLMX_RQ(5)
		BLMX_RQ(6)
		BLMX_RQ(7)
		BLMX_RQ(8)
	default:
		throw StatusError{BLMX_EINVAL, "internal: fused read query needs bits_to_save <= 8"};
	}
#undef BLMX_RQ
}

bool read_query_fusable(const Index& ix) { return ix.prm.bits_saved <= 8; }

void launch_read_query(const Index& ix, const FrontScratch* fs, const uint8_t* d_bytes, uint64_t nbytes, uint64_t npos, const uint64_t* offs,
                       const uint64_t* koffs, uint64_t nseq, int64_t* d_ids, unsigned long long* d_counts, int* d_err, hipStream_t s, DoneSignal done)
{
	if (npos == 0)
		return;
	const FrontWhere fw = where_of(fs, offs, koffs, nseq);
	if (ix.KW == 1)
		launch_read_query_kw<1>(ix, fw, d_bytes, nbytes, npos, d_ids, d_counts, d_err, s, done);
	else
		launch_read_query_kw<2>(ix, fw, d_bytes, nbytes, npos, d_ids, d_counts, d_err, s, done);
	BLMX_HIP(hipGetLastError());
}

}  // namespace blmx
