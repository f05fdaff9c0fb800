// query_device.h -- the per-k-mer query body shared by the batch query kernel (blmx_query.hip) and the fused read front end
// (blmx_reads.hip): 2^b window scan + bucket / MPHF / position chain of query_get_hash (reference blight.cpp:716-742).
#pragma once
#include "index_device.h"

namespace blmx {

// The 2^b scan of query_get_hash / query_get_pos_unitig (blight.cpp:729-739 / 699-709).
// canon(stored) == x  <=>  stored in {x, rc(x)} when x is canonical (x <= rc(x)); a non-canonical x never matches, so
// no reverse complement is computed inside the loop.  Like the reference the scan ignores super-k-mer and bucket
// boundaries.  Branch-free formulation for wave64: the window (NW words) is preloaded into registers, the 32*KW
// nucleotides at the current position are kept TOP-ALIGNED in 32-bit words c[0..2KW) and advanced by one nucleotide
// with a v_alignbit chain; the stream of following nucleotides is pre-aligned per lane (S[r] = the 32 nucleotides that
// enter during segment r), so that every lane runs exactly 2^b steps with wave-uniform trip counts and refills.
// A step = mask + xor/or compare against x and rc(x) + one v_min3; no exec-mask juggling, no branches.
// ID / membership mode: returns 0 on any match, -1 otherwise (the ID does not depend on WHICH position matched,
// blight.cpp:736).  WANT_POS: returns the index j of the FIRST match in [0, 2^LOGP) (blight.cpp:706) or -1.
template <int KW, int LOGP, bool WANT_POS, int NW>
__device__ __forceinline__ int scan_window_exact(const uint64_t (&W)[NW], int s0, int k, Kmer<KW> x, Kmer<KW> xr)
{
	constexpr int P = 1 << LOGP;
	constexpr int NSEG = (P + 31) / 32;  // 32-step segments, one pre-aligned stream word each
	constexpr int NWORD = 2 * KW;
	// top-aligned window and query words, most significant first
	uint32_t c[NWORD], X[NWORD], XR[NWORD], M[NWORD];
	const int sh = 64 * KW - 2 * k;  // insignificant low bits of the top-aligned window
	if constexpr (KW == 1) {
		const uint64_t cur = funnel_left(W[0], W[1], 2 * s0);
		const uint64_t xt = x.lo << sh, xrt = xr.lo << sh;
		c[0] = (uint32_t)(cur >> 32);
		c[1] = (uint32_t)cur;
		X[0] = (uint32_t)(xt >> 32);
		X[1] = (uint32_t)xt;
		XR[0] = (uint32_t)(xrt >> 32);
		XR[1] = (uint32_t)xrt;
	} else {
		const uint64_t ch = funnel_left(W[0], W[1], 2 * s0), cl = funnel_left(W[1], W[2], 2 * s0);
		uint64_t xh, xl, xrh, xrl;
		if (sh >= 64) {
			xh = x.lo << (sh - 64);
			xl = 0;
			xrh = xr.lo << (sh - 64);
			xrl = 0;
		} else {
			xh = (x.hi << sh) | (sh ? (x.lo >> (64 - sh)) : 0ull);
			xl = x.lo << sh;
			xrh = (xr.hi << sh) | (sh ? (xr.lo >> (64 - sh)) : 0ull);
			xrl = xr.lo << sh;
		}
		c[0] = (uint32_t)(ch >> 32);
		c[1] = (uint32_t)ch;
		c[2] = (uint32_t)(cl >> 32);
		c[3] = (uint32_t)cl;
		X[0] = (uint32_t)(xh >> 32);
		X[1] = (uint32_t)xh;
		X[2] = (uint32_t)(xl >> 32);
		X[3] = (uint32_t)xl;
		XR[0] = (uint32_t)(xrh >> 32);
		XR[1] = (uint32_t)xrh;
		XR[2] = (uint32_t)(xrl >> 32);
		XR[3] = (uint32_t)xrl;
	}
#pragma unroll
	for (int i = 0; i < NWORD; i++) {  // word i covers bits [32i, 32i+32) from the top; 2k bits are significant
		const int nb = 2 * k - 32 * i;
		M[i] = nb >= 32 ? 0xffffffffu : (nb <= 0 ? 0u : (0xffffffffu << (32 - nb)));
	}
	uint32_t acc = 0xffffffffu;  // ID mode: min over steps of the xor distance (0 = match)
	int first = 0x7fffffff;      // POS mode: first matching step
#pragma unroll
	for (int r = 0; r < NSEG; r++) {
		const uint64_t sw = funnel_left(W[KW + r], W[KW + r + 1], 2 * s0);  // the 32 nucleotides entering in this segment
		uint32_t n_hi = (uint32_t)(sw >> 32), n_lo = (uint32_t)sw;
		constexpr int kSteps = P < 32 ? P : 32;
#pragma unroll 8
		for (int t = 0; t < kSteps; t++) {
			uint32_t d1 = 0, d2 = 0;
#pragma unroll
			for (int i = 0; i < NWORD; i++) {
				const uint32_t a = c[i] & M[i];
				d1 |= a ^ X[i];
				d2 |= a ^ XR[i];
			}
			const uint32_t d = min(d1, d2);
			if constexpr (WANT_POS)
				first = min(first, d == 0 ? (32 * r + t) : 0x7fffffff);
			else
				acc = min(acc, d);
#pragma unroll
			for (int i = 0; i + 1 < NWORD; i++)
				c[i] = __builtin_amdgcn_alignbit(c[i], c[i + 1], 30);
			c[NWORD - 1] = __builtin_amdgcn_alignbit(c[NWORD - 1], n_hi, 30);
			n_hi = __builtin_amdgcn_alignbit(n_hi, n_lo, 30);
			n_lo <<= 2;
		}
		// all lanes of the wave that are scanning have their answer: stop (self-queries end after the first segments)
		if constexpr (NSEG > 1) {
			if (WANT_POS ? __all(first != 0x7fffffff) : __all(acc == 0))
				break;
		}
	}
	if constexpr (WANT_POS)
		return first == 0x7fffffff ? -1 : first;
	else
		return acc == 0 ? 0 : -1;
}

// The k-mer at nucleotide offset T of the register window W (T < 32 * (NW - KW)): the two or three words are picked with
// compares against the (lane-varying) word index, then shifted like packed_kmer.
template <int KW, int NW> __device__ __forceinline__ Kmer<KW> window_kmer(const uint64_t (&W)[NW], int T, int k)
{
	const int wi = T >> 5, sb = 2 * (T & 31);
	uint64_t a = 0, b = 0, c = 0;
#pragma unroll
	for (int i = 0; i + KW < NW; i++) {
		if (wi == i) {
			a = W[i];
			b = W[i + 1];
			if constexpr (KW == 2)
				c = W[i + 2];
		}
	}
	Kmer<KW> r;
	if constexpr (KW == 1) {
		r.lo = funnel_left(a, b, sb) >> (64 - 2 * k);
	} else {
		const uint64_t v0 = funnel_left(a, b, sb), v1 = funnel_left(b, c, sb);
		const int sh = 128 - 2 * k;
		if (sh >= 64) {
			r.lo = v0 >> (sh - 64);
			r.hi = 0;
		} else {
			r.lo = (v1 >> sh) | (v0 << (64 - sh));
			r.hi = v0 >> sh;
		}
	}
	return r;
}

// The scan as the query kernel runs it: a FILTER on the 16 leading nucleotides, then one exact comparison.
// A stored k-mer can equal x or rc(x) only if its top 32 bits equal theirs, so every step is: one v_alignbit with a
// compile-time shift (the lane's stream is pre-aligned once per 32-step segment), two v_cmp_eq, one v_cndmask -- 4 VALU
// instead of the 12 of the exact step, and no serial dependency between steps.  Steps run downwards inside a segment so
// that "candidate = this step" needs no min.  The first candidate (smallest step) is then compared exactly in registers;
// it is the first full match unless the filter was hit by a different k-mer with the same 16-nucleotide prefix
// (~2^-32 per step on random data, common in repeats): only then the exact scan above runs.  k < 16 (top word not fully
// significant) always takes the exact scan.
template <int KW, int LOGP, bool WANT_POS>
__device__ __forceinline__ int scan_window(const uint64_t* __restrict__ seq, uint64_t g0, int k, Kmer<KW> x, Kmer<KW> xr)
{
	constexpr int P = 1 << LOGP;
	constexpr int NSEG = (P + 31) / 32;
	constexpr int NW = KW + NSEG + 1;
	constexpr int kNone = 0x7fffffff;
	const uint64_t w0 = g0 >> 5;
	const int s0 = (int)(g0 & 31);
	uint64_t W[NW];
#pragma unroll
	for (int i = 0; i < NW; i++)
		W[i] = exp_load<2>(&seq[w0 + i]);
	if (k >= 16) {
	uint32_t X0, XR0;  // top 32 bits of the top-aligned queries
	if constexpr (KW == 1) {
		const int sh = 64 - 2 * k;
		X0 = (uint32_t)((x.lo << sh) >> 32);
		XR0 = (uint32_t)((xr.lo << sh) >> 32);
	} else {
		const int sh = 128 - 2 * k;
		uint64_t xh, xrh;
		if (sh >= 64) {
			xh = x.lo << (sh - 64);
			xrh = xr.lo << (sh - 64);
		} else {
			xh = (x.hi << sh) | (sh ? (x.lo >> (64 - sh)) : 0ull);
			xrh = (xr.hi << sh) | (sh ? (xr.lo >> (64 - sh)) : 0ull);
		}
		X0 = (uint32_t)(xh >> 32);
		XR0 = (uint32_t)(xrh >> 32);
	}
	int first = kNone;
	uint64_t sw = funnel_left(W[0], W[1], 2 * s0);
#pragma unroll
	for (int r = 0; r < NSEG; r++) {
		const uint64_t swn = funnel_left(W[r + 1], W[r + 2], 2 * s0);
		const uint32_t A0 = (uint32_t)(sw >> 32), A1 = (uint32_t)sw, A2 = (uint32_t)(swn >> 32);
		constexpr int kSteps = P < 32 ? P : 32;
		int f = kNone;
#pragma unroll
		for (int t = kSteps - 1; t >= 0; t--) {
			const uint32_t hi = t < 16 ? A0 : A1, lo = t < 16 ? A1 : A2;
			const int sft = 2 * (t & 15);
			const uint32_t c0 = sft == 0 ? hi : __builtin_amdgcn_alignbit(hi, lo, 32 - sft);
			f = ((c0 == X0) | (c0 == XR0)) ? 32 * r + t : f;
		}
		first = min(first, f);
		sw = swn;
		// all lanes of the wave that are scanning have a candidate: stop (self-queries end after the first segments)
		if constexpr (NSEG > 1) {
			if (__all(first != kNone))
				break;
		}
	}
	if (first == kNone)
		return -1;
	const Kmer<KW> cand = window_kmer<KW, NW>(W, s0 + first, k);
	if (cand == x || cand == xr)
		return first;
	}
	return scan_window_exact<KW, LOGP, WANT_POS, NW>(W, s0, k, x, xr);
}

// query_get_hash / query_get_pos_unitig for ONE canonical k-mer whose minimizer is known and inside [0, nbuckets)
// (reference blight.cpp:716-742 / 686-712).  id = h + group prefix (:736) or -1; posj = bucket-relative position of the
// first match (:706) or -1.
// The filter gate (word 0 of filter_gate: 0 = the last large batch had almost no absent k-mers) is read ONCE at the start of a
// kernel and folded into the view, away from the dependent chain of the query body.
__device__ __forceinline__ DeviceIndexView gated_view(DeviceIndexView v)
{
	if (v.filter && v.filter_gate && *reinterpret_cast<const volatile uint32_t*>(v.filter_gate) == 0)
		v.filter = nullptr;
	return v;
}

// the query body behind the bucket descriptor and the group head (query_one loads them from HBM, k_query_slots from LDS)
template <int KW, int LOGP, bool WANT_POS>
__device__ __forceinline__ void query_one_with(const DeviceIndexView& v, Kmer<KW> x, Kmer<KW> xr, const GroupDesc* gd, const GroupHead& gh, uint64_t bstart,
                                               uint32_t bnt, int64_t& id, int32_t& posj);

template <int KW, int LOGP, bool WANT_POS>
__device__ __forceinline__ void query_one(const DeviceIndexView& v, Kmer<KW> x, Kmer<KW> xr, uint32_t mini, int64_t& id, int32_t& posj)
{
	// bucket descriptor + group head: one round trip (both addresses depend only on the minimizer)
	uint64_t bstart;
	uint32_t bnt;
	const GroupDesc* gd = &v.groups[mini >> v.gshift];
	const GroupHead gh = v.qheads ? load_bucket_and_qhead(&v.buckets[mini], &v.qheads[mini >> v.gshift], bstart, bnt)
	                              : load_bucket_and_head(&v.buckets[mini], gd, bstart, bnt);
	query_one_with<KW, LOGP, WANT_POS>(v, x, xr, gd, gh, bstart, bnt, id, posj);
}

template <int KW, int LOGP, bool WANT_POS>
__device__ __forceinline__ void query_one_with(const DeviceIndexView& v, Kmer<KW> x, Kmer<KW> xr, const GroupDesc* gd, const GroupHead& gh, uint64_t bstart,
                                               uint32_t bnt, int64_t& id, int32_t& posj)
{
	id = -1;
	posj = -1;
	if (bnt == 0)  // blight.cpp:719-720
		return;
	if (v.filter && !v.fat) {  // absent-key filter (index_device.h): one 8-B load instead of the levels, the position and the window
		const uint64_t fh = filter_hash<KW>(x);
		const uint64_t fm = filter_mask(fh);
		if ((v.filter[filter_word(gh.prefix - v.id_base, gh.size0, v.filter_bpk, fh)] & fm) != fm)
			return;
	}
	// (fat level-0 lines: the filter word comes with the level-0 line, inside the lookup)
	uint32_t field;  // level-0 records: the position field comes with the rank (index_device.h)
	const uint64_t h = mphf_lookup<KW>(v, gd, gh, x, v.fat && v.filter != nullptr, field);
	if (h == kNotFound)  // blight.cpp:725-726
		return;
	if (field == kNoField)
		field = read_position_field(v.positions, gh.posstart, gh.nbits, h);
	const uint64_t pos = (uint64_t)(field >> v.hint) << v.b;  // the reference's value: first position of the 2^b window
	if (!(pos + (uint64_t)v.k - 1 < (uint64_t)bnt))  // blight.cpp:729
		return;
	if (xr < x)  // non-canonical query: canonR == canon can never hold (blight.cpp:733-735)
		return;
	// The field's extra low bits name the sub-window that holds the key this rank belongs to: an indexed k-mer is found there in
	// 2^(b - hint) steps.  Only a k-mer that is NOT there (a false positive of the filter, or one of the reference's own false
	// positives, which sit anywhere in the window) goes through the reference's whole window; the first match of the whole
	// window is what the positional query reports, so that one always scans all of it.
	constexpr int kHint = position_hint_bits(LOGP);
	if constexpr (kHint > 0 && !WANT_POS) {
		if (v.hint == kHint) {
			const uint64_t sub = pos + ((uint64_t)(field & ((1u << kHint) - 1u)) << (LOGP - kHint));
			if (scan_window<KW, LOGP - kHint, false>(v.seq, bstart + sub, v.k, x, xr) >= 0) {
				id = (int64_t)(h + gh.prefix);
				return;
			}
		}
	}
	const int j = scan_window<KW, LOGP, WANT_POS>(v.seq, bstart + pos, v.k, x, xr);
	if (j >= 0) {
		id = (int64_t)(h + gh.prefix);        // blight.cpp:736
		posj = (int32_t)(pos + (uint64_t)j);  // blight.cpp:706
	}
}

}  // namespace blmx
