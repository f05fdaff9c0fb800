// blmx_build.hip -- gfx950 index construction (reference construct_index, blight.cpp:108-125).
//
// The reference streams unitigs through a per-thread chopper, shuffles super-k-mers through gzip temp files and builds
// one MPHF after the other.  Here the whole input is resident in HBM and every phase is a data-parallel kernel:
//
//   K1  k_mark_starts / k_encode_pack   ASCII -> 2-bit packed source (nuc2int, kmer.h:56-69) + unitig-start bitmap
//   K2  k_superkmer_flags               per k-mer minimizer (kmer.h:791-810) ; super-k-mer head / end bitmaps via wave
//                                       ballot (a super-k-mer = maximal run of equal minimizer VALUE, kmer.h:654-693)
//   K3  scan + k_collect / k_sk_minis   compact super-k-mers: (source position, #k-mers, minimizer)
//   K4  radix sort by minimizer (stable: bucket order = unitig file order then offset, as the reference at t=1) ;
//       scans -> global nt / key offsets ; bucket table ; group table (blight.cpp:259-306)
//   K5  k_scatter_pack                  super-k-mers -> bucketSeq (str2bool, blight.cpp:311-324)
//   K6  k_expand_keys                   canonical k-mer of every valid position, in (bucket, position) order (:421-451)
//   K7  k_mphf_mark / k_mphf_survive    BBHash levels for ALL groups at once (bbhash.h:677-709): atomicOr test-and-set,
//                                       collision bitmap, survivors compacted with wave-aggregated appends
//   K8  k_line_popc / scan / k_line_write  clear collisions (bbhash.h:358-368) + ranks (bbhash.h:447-465) -> 16-B lines [u32 rank | 96 bits]
//   K9  fallback table (bbhash.h:711-728) ; k_fill_positions (blight.cpp:486-519)
//
// All build-time atomics are ORs into zeroed arrays or order-independent appends that are sorted afterwards, so the
// finished index is bit-for-bit reproducible.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <cmath>

#include "blmx_internal.h"
#include "scan_kernels.h"
#include "index_device.h"

namespace blmx {

namespace {

constexpr unsigned kBlock = 256;
inline unsigned grid_for(uint64_t n, unsigned block = kBlock) { return (unsigned)std::max<uint64_t>(1, (n + block - 1) / block); }

enum BuildErr : int { kErrShort = 1, kErrDna = 2, kErrLimit = 4 };

// ---- K1 ----------------------------------------------------------------------------------------------------------------
__global__ void k_mark_starts(const uint64_t* __restrict__ offs, uint64_t nseq, int k, unsigned long long* __restrict__ ustart,
                              int* __restrict__ err)
{
	const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (u >= nseq)
		return;
	const uint64_t a = offs[u], b = offs[u + 1];
	if (b < a + (uint64_t)k)
		atomicOr(err, kErrShort);
	atomicOr(&ustart[a >> 6], 1ull << (a & 63));
}

// 32 nt -> one u64, first nt most significant.  Full words are read as two 16-B loads when the buffer is 16-B aligned.
template <bool ALIGNED>
__global__ void k_encode_pack(const uint8_t* __restrict__ ascii, uint64_t total_len, uint64_t nwords, uint64_t* __restrict__ src,
                              int* __restrict__ err)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords)
		return;
	const uint64_t p0 = w * 32;
	uint64_t word = 0;
	bool bad = false;
	if (ALIGNED && p0 + 32 <= total_len) {
		const uint4* q = reinterpret_cast<const uint4*>(ascii + p0);
		const uint4 a = q[0], b = q[1];
		const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
		for (int i = 0; i < 8; i++) {
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const uint32_t c = nuc_code((uint8_t)(v[i] >> (8 * t)));
				bad |= c > 3;
				word = (word << 2) | (c & 3u);
			}
		}
	} else {
		for (int t = 0; t < 32; t++) {
			const uint64_t p = p0 + t;
			uint32_t c = 0;
			if (p < total_len) {
				c = nuc_code(ascii[p]);
				bad |= c > 3;
			}
			word = (word << 2) | (c & 3u);
		}
	}
	if (bad)
		atomicOr(err, kErrDna);
	src[w] = word;
}

// ---- K2 ----------------------------------------------------------------------------------------------------------------
// bits of the LSB-first bitmap U at positions q .. q+63
__device__ __forceinline__ uint64_t bitmap_window(const unsigned long long* __restrict__ U, uint64_t q)
{
	const uint64_t w = q >> 6;
	const uint32_t s = (uint32_t)q & 63u;
	const uint64_t lo = U[w], hi = U[w + 1];
	return s ? ((lo >> s) | (hi << (64 - s))) : lo;
}

// One lane per source position q in [0, total_len].  valid(q): a k-mer starts at q and lies inside one unitig.
// cont(q): k-mers q-1 and q belong to the same super-k-mer (both valid, same unitig, equal minimizer).
//   head bit q = valid(q) && !cont(q)          -> a super-k-mer starts at q
//   end  bit q = valid(q-1) && !cont(q)        -> a super-k-mer's last k-mer is q-1
// Each wave covers one aligned 64-position word of both bitmaps and writes it with one ballot.
//
// Rolling minimizer (SuperKChopper::update_mini, kmer.h:734-781, is a monotone deque over the m-mer hashes): here every
// lane hashes ONE m-mer (the one starting at its position: parity-canonical form, signed revhash) into LDS and the
// window minimum over the w = k-m+1 m-mers of each k-mer is assembled from power-of-two prefix minima
// (M_{p+1}[i] = min(M_p[i], M_p[i+2^p]); the window is the disjoint union of the blocks given by the binary digits of w):
// ceil(log2 w)+1 LDS rounds instead of w hashes per k-mer.  revhash is a bijection, so two k-mers have the same
// minimizer VALUE iff their window-minimum hashes are equal: the comparison is done on the hashes.
template <int KW>
__global__ __launch_bounds__(kBlock) void k_superkmer_flags(const uint64_t* __restrict__ src, const unsigned long long* __restrict__ ustart,
                                                            uint64_t total_len, int k, int m, unsigned long long* __restrict__ Hb,
                                                            unsigned long long* __restrict__ Eb)
{
	constexpr int kMaxW = 64;
	__shared__ int32_t s_pow[2][kBlock + kMaxW + 2];  // prefix minima over blocks of 2^p m-mers, double buffered
	__shared__ int32_t s_min[kBlock + 1];             // window minimum of the k-mer at position base-1+e
	__shared__ uint8_t s_valid[kBlock + 1];
	const int w = k - m + 1;
	const int L = (int)kBlock + w;  // m-mer entries e = 0 .. L-1  <->  positions base-1 .. base+kBlock-1+w-1
	// (an AQL dispatch counts work-items in 32 bits per dimension: more than 2^32 positions need the second grid dimension)
	const uint64_t base = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * kBlock;
	if (base > total_len)
		return;
	const int tid = (int)threadIdx.x;
	auto mmer_hash = [&](int e) -> int32_t {
		const int64_t p = (int64_t)base - 1 + e;
		if (p < 0 || (uint64_t)p + (uint64_t)m > total_len)
			return 0x7fffffff;
		const uint64_t wd = (uint64_t)p >> 5;
		const int sh = 2 * (int)(p & 31);
		const uint32_t f = (uint32_t)(funnel_left(src[wd], src[wd + 1], sh) >> (64 - 2 * m));
		const uint32_t r = (uint32_t)(rc_digits64((uint64_t)f) >> (64 - 2 * m));
		return revhash32(((__popc(f) & 1) ? f : r) >> 1);  // ParityCanonical, kmer.h:475-487; revhash kmer.h:102-108
	};
	auto kmer_valid = [&](int64_t p) -> bool {
		if (p < 0 || (uint64_t)p + (uint64_t)k > total_len)
			return false;
		// unitig starts inside (p, p+k-1] make the k-mer straddle two unitigs
		const uint64_t win = bitmap_window(ustart, (uint64_t)p + 1);
		const uint64_t msk = (k - 1 >= 64) ? ~0ull : ((1ull << (k - 1)) - 1ull);
		return (win & msk) == 0;
	};
	for (int e = tid; e < L; e += (int)kBlock)
		s_pow[0][e] = mmer_hash(e);
	for (int e = tid; e <= (int)kBlock; e += (int)kBlock) {
		s_min[e] = 0x7fffffff;
		s_valid[e] = kmer_valid((int64_t)base - 1 + e) ? 1 : 0;
	}
	__syncthreads();
	int cur = 0, off = 0;
	for (int p2 = 0; (1 << p2) <= w; p2++) {
		const int step = 1 << p2;
		if (w & step) {
			for (int e = tid; e <= (int)kBlock; e += (int)kBlock)
				s_min[e] = min(s_min[e], s_pow[cur][e + off]);
			off += step;
		}
		for (int e = tid; e + step < L; e += (int)kBlock)
			s_pow[cur ^ 1][e] = min(s_pow[cur][e], s_pow[cur][e + step]);
		__syncthreads();
		cur ^= 1;
	}
	const uint64_t q = base + threadIdx.x;
	const bool valid = s_valid[tid + 1] != 0, pvalid = s_valid[tid] != 0;
	const bool is_start = q <= total_len && ((ustart[q >> 6] >> (q & 63)) & 1ull);
	const bool cont = pvalid && valid && s_min[tid] == s_min[tid + 1] && !is_start;
	const unsigned long long hb = __ballot(valid && !cont);
	const unsigned long long eb = __ballot(pvalid && !cont);
	if ((threadIdx.x & 63) == 0) {
		Hb[q >> 6] = hb;
		Eb[q >> 6] = eb;
	}
}

// ---- K3 ----------------------------------------------------------------------------------------------------------------
// One lane per bitmap word: scatter the positions of its set bits (heads -> sk_pos, ends -> sk_last = q-1).
__global__ void k_collect(const unsigned long long* __restrict__ Hb, const unsigned long long* __restrict__ Eb,
                          const uint64_t* __restrict__ Hpre, const uint64_t* __restrict__ Epre, uint64_t nwords,
                          uint64_t* __restrict__ sk_pos, uint64_t* __restrict__ sk_last)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords)
		return;
	unsigned long long h = Hb[w];
	uint64_t o = Hpre[w];
	while (h) {
		const int b = __ffsll(h) - 1;
		sk_pos[o++] = w * 64 + (uint64_t)b;
		h &= h - 1;
	}
	unsigned long long e = Eb[w];
	o = Epre[w];
	while (e) {
		const int b = __ffsll(e) - 1;
		sk_last[o++] = w * 64 + (uint64_t)b - 1;
		e &= e - 1;
	}
}

template <int KW>
__global__ void k_sk_minis(const uint64_t* __restrict__ src, const uint64_t* __restrict__ sk_pos, uint64_t nsk, int k, int m,
                           uint32_t* __restrict__ sk_mini)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsk)
		return;
	const Kmer<KW> x = packed_kmer<KW>(src, sk_pos[i], k);
	sk_mini[i] = kmer_minimizer<KW>(x, kmer_rc<KW>(x, k), k, m);
}

// ---- sharding: per-group key histogram over ALL super-k-mers, then keep the owned ones -------------------------------
__global__ void k_group_hist(const uint32_t* __restrict__ sk_mini, const uint64_t* __restrict__ sk_pos, const uint64_t* __restrict__ sk_last,
                             uint64_t nsk, int gshift, unsigned long long* __restrict__ ghist)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsk)
		return;
	atomicAdd(&ghist[sk_mini[i] >> gshift], (unsigned long long)(sk_last[i] - sk_pos[i] + 1));
}

__global__ void k_shard_flags(const uint32_t* __restrict__ sk_mini, uint64_t nsk, int gshift, uint32_t glo, uint32_t ghi, uint8_t* __restrict__ flags)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsk)
		return;
	const uint32_t g = sk_mini[i] >> gshift;
	flags[i] = (g >= glo && g < ghi) ? 1 : 0;
}

// ---- sharded build from slices: super-k-mers travel to the rank that owns their MPHF group ------------------------------
struct OwnerTable {
	uint32_t world;
	uint32_t bound[17];    // group boundaries (shard_bounds)
	uint64_t seg_first[17];  // first entry of every segment in the owner-sorted (or source-ordered) list
	uint64_t seg_word[17];   // first u64 word of every segment in the packed stream (each segment starts on a word)
};

__global__ void k_sk_owner(const uint32_t* __restrict__ sk_mini, uint64_t nsk, int gshift, OwnerTable t, uint32_t* __restrict__ owner)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsk)
		return;
	const uint32_t g = sk_mini[i] >> gshift;
	uint32_t r = 0;
	while (r + 1 < t.world && g >= t.bound[r + 1])
		r++;
	owner[i] = r;
}

// first index of every owner in the sorted owner list (world + 1 binary searches)
__global__ void k_owner_segments(const uint32_t* __restrict__ sorted_owner, uint64_t nsk, uint32_t world, uint64_t* __restrict__ seg_first)
{
	const uint32_t d = threadIdx.x;
	if (d > world)
		return;
	uint64_t lo = 0, hi = nsk;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (sorted_owner[mid] < d)
			lo = mid + 1;
		else
			hi = mid;
	}
	seg_first[d] = lo;
}

// out[t] = first index i in [0, n) with arr[i] >= target[t] (n if none); arr non-decreasing
__global__ void k_lower_bounds(const uint64_t* __restrict__ arr, uint64_t n, const uint64_t* __restrict__ target, uint32_t m, uint64_t* __restrict__ out)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m)
		return;
	uint64_t lo = 0, hi = n;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (arr[mid] < target[t])
			lo = mid + 1;
		else
			hi = mid;
	}
	out[t] = lo;
}

__device__ __forceinline__ uint32_t segment_of(const OwnerTable& t, uint64_t i)
{
	uint32_t r = 0;
	while (r + 1 < t.world && i >= t.seg_first[r + 1])
		r++;
	return r;
}

// nt offsets of a running scan -> offsets inside a stream whose segments each start on a u64 word
__global__ void k_segment_offsets(const uint64_t* __restrict__ nt_scan, uint64_t nsk, OwnerTable t, uint64_t* __restrict__ nt_out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsk)
		return;
	const uint32_t r = segment_of(t, i);
	nt_out[i] = nt_scan[i] - nt_scan[t.seg_first[r]] + 32 * t.seg_word[r];
}

// what travels next to the packed nucleotides: (minimizer << 32) | length in nt, in owner order
__global__ void k_sk_meta(const uint32_t* __restrict__ order, const uint32_t* __restrict__ sk_mini, const uint64_t* __restrict__ len_nt, uint64_t nsk,
                          uint64_t* __restrict__ meta)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nsk)
		meta[i] = ((uint64_t)sk_mini[order[i]] << 32) | (uint64_t)(uint32_t)len_nt[i];
}

// kept super-k-mers (flag 1) -> slot[i] of the three output arrays; the last thread publishes the number kept
__global__ void k_select3(const uint8_t* __restrict__ flag, const uint32_t* __restrict__ slot, uint64_t n, const uint64_t* __restrict__ pos,
                          const uint64_t* __restrict__ last, const uint32_t* __restrict__ mini, uint64_t* __restrict__ pos2, uint64_t* __restrict__ last2,
                          uint32_t* __restrict__ mini2, uint64_t* __restrict__ kept)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	if (flag[i]) {
		const uint32_t o = slot[i];
		pos2[o] = pos[i];
		last2[o] = last[i];
		mini2[o] = mini[i];
	}
	if (i == n - 1)
		*kept = slot[n];
}

struct MetaLen {  // scan input: the length field of a received super-k-mer's meta word
	const uint64_t* meta;
	__device__ uint64_t operator()(uint64_t i) const { return meta[i] & 0xffffffffull; }
};

// received (meta, running nt scan) -> the super-k-mer list the rest of the build works on
__global__ void k_sk_from_meta(const uint64_t* __restrict__ meta, const uint64_t* __restrict__ nt_scan, uint64_t nsk, int k, OwnerTable t,
                               uint64_t* __restrict__ sk_pos, uint64_t* __restrict__ sk_last, uint32_t* __restrict__ sk_mini)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsk)
		return;
	const uint32_t r = segment_of(t, i);
	const uint64_t pos = nt_scan[i] - nt_scan[t.seg_first[r]] + 32 * t.seg_word[r];
	const uint64_t len = meta[i] & 0xffffffffull;
	sk_pos[i] = pos;
	sk_last[i] = pos + len - (uint64_t)k;
	sk_mini[i] = (uint32_t)(meta[i] >> 32);
}

// ---- K4 ----------------------------------------------------------------------------------------------------------------
__global__ void k_sorted_lengths(const uint32_t* __restrict__ order, const uint64_t* __restrict__ sk_pos,
                                 const uint64_t* __restrict__ sk_last, uint64_t nsk, int k, uint64_t* __restrict__ len_nt,
                                 uint64_t* __restrict__ len_k)
{
	const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j > nsk)
		return;
	if (j == nsk) {  // sentinel so that the exclusive scans also deliver the totals
		len_nt[j] = 0;
		len_k[j] = 0;
		return;
	}
	const uint32_t i = order[j];
	const uint64_t nk = sk_last[i] - sk_pos[i] + 1;
	len_k[j] = nk;
	len_nt[j] = nk + (uint64_t)k - 1;
}

// run boundaries of equal minimizers in the sorted super-k-mer list -> per-bucket [begin, end) in nt and in keys
__global__ void k_bucket_runs(const uint32_t* __restrict__ smini, const uint64_t* __restrict__ nt_start,
                              const uint64_t* __restrict__ key_start, uint64_t nsk, uint64_t* __restrict__ b_nt_begin,
                              uint64_t* __restrict__ b_nt_end, uint64_t* __restrict__ b_key_begin, uint64_t* __restrict__ b_key_end)
{
	const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nsk)
		return;
	const uint32_t mi = smini[j];
	if (j == 0 || smini[j - 1] != mi) {
		b_nt_begin[mi] = nt_start[j];
		b_key_begin[mi] = key_start[j];
	}
	if (j + 1 == nsk || smini[j + 1] != mi) {
		b_nt_end[mi] = nt_start[j + 1];
		b_key_end[mi] = key_start[j + 1];
	}
}

__global__ void k_bucket_finish(const uint64_t* __restrict__ b_nt_begin, const uint64_t* __restrict__ b_nt_end,
                                const uint64_t* __restrict__ b_key_begin, const uint64_t* __restrict__ b_key_end, uint32_t nbuckets,
                                uint64_t* __restrict__ b_nt, uint64_t* __restrict__ b_keys, int* __restrict__ err)
{
	const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b > nbuckets)
		return;
	if (b == nbuckets) {
		b_nt[b] = 0;
		b_keys[b] = 0;
		return;
	}
	const uint64_t nt = b_nt_end[b] - b_nt_begin[b];
	if (nt > 0xffffffffull)  // bucket_minimizer::nuc_minimizer is uint32_t (blight.h:33)
		atomicOr(err, kErrLimit);
	b_nt[b] = nt;
	b_keys[b] = b_key_end[b] - b_key_begin[b];
}

__global__ void k_bucket_table(const uint64_t* __restrict__ b_nt, const uint64_t* __restrict__ b_start, uint32_t nbuckets,
                               BucketDesc* __restrict__ out)
{
	const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nbuckets)
		return;
	BucketDesc d;
	d.start = b_start[b];
	d.nt = (uint32_t)b_nt[b];
	d.pad = 0;
	out[b] = d;
}

// one wave per group: key count and largest bucket (blight.cpp:285-303)
__global__ void k_group_tables(const uint64_t* __restrict__ b_nt, const uint64_t* __restrict__ b_key_prefix, uint32_t ngroups, int gshift,
                               uint64_t* __restrict__ g_nkeys, uint64_t* __restrict__ g_prefix, uint32_t* __restrict__ g_max)
{
	const uint64_t gw = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t lane = threadIdx.x & 63;
	if (gw >= ngroups)
		return;
	const uint64_t b0 = gw << gshift, b1 = (gw + 1) << gshift;
	uint32_t mx = 0;
	for (uint64_t b = b0 + lane; b < b1; b += 64)
		mx = max(mx, (uint32_t)b_nt[b]);
	for (int o = 32; o > 0; o >>= 1)
		mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
	if (lane == 0) {
		g_max[gw] = mx;
		g_prefix[gw] = b_key_prefix[b0];
		g_nkeys[gw] = b_key_prefix[b1] - b_key_prefix[b0];
	}
}

// ---- K5 ----------------------------------------------------------------------------------------------------------------
// str2bool (blight.cpp:311-324): copy the nucleotides of sorted super-k-mer j to its slot of the packed sequence.
// `clip`: nothing is written at or behind this destination position (the halo of a shard keeps only the head of what follows it).
__global__ void k_scatter_pack(const uint64_t* __restrict__ src, const uint32_t* __restrict__ order, const uint64_t* __restrict__ sk_pos,
                               const uint64_t* __restrict__ nt_start, const uint64_t* __restrict__ len_nt, uint64_t nsk,
                               unsigned long long* __restrict__ seq, uint64_t clip)
{
	const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nsk)
		return;
	const uint64_t p = sk_pos[order[j]];
	// destinations are back to back (len_nt == nullptr: the next start is this end) or carry their own lengths (segmented streams)
	const uint64_t d = nt_start[j];
	if (d >= clip)
		return;
	const uint64_t e = min(len_nt ? d + len_nt[j] : nt_start[j + 1], clip);
	for (uint64_t dw = d >> 5; dw <= (e - 1) >> 5; dw++) {
		const uint64_t a = max(d, dw * 32), z = min(e, dw * 32 + 32);
		const int cnt = (int)(z - a);
		const uint64_t sp = p + (a - d);
		uint64_t v = funnel_left(src[sp >> 5], src[(sp >> 5) + 1], 2 * (int)(sp & 31));
		if (cnt < 32)
			v &= ~0ull << (64 - 2 * cnt);
		v >>= 2 * (a & 31);
		atomicOr(&seq[dw], v);
	}
}

// The halo of a shard: `n` nucleotides (2-bit packed, first nt most significant, starting at bit 63 of halo[0]; one readable word
// behind them) ORed into the zeroed tail of the packed sequence at nucleotide `at`.  One thread per destination word.
__global__ void k_put_halo(const uint64_t* __restrict__ halo, uint64_t n, uint64_t at, unsigned long long* __restrict__ seq)
{
	const uint64_t dw = (at >> 5) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t e = at + n;
	if (n == 0 || dw > (e - 1) >> 5)
		return;
	const uint64_t a = max(at, dw * 32), z = min(e, dw * 32 + 32);
	const int cnt = (int)(z - a);
	const uint64_t sp = a - at;
	uint64_t v = funnel_left(halo[sp >> 5], halo[(sp >> 5) + 1], 2 * (int)(sp & 31));
	if (cnt < 32)
		v &= ~0ull << (64 - 2 * cnt);
	v >>= 2 * (a & 31);
	atomicOr(&seq[dw], v);
}

// ---- K6 ----------------------------------------------------------------------------------------------------------------
// Load-balanced expansion shared by K6 and K9b: a block owns kSkPerBlock consecutive sorted super-k-mers, stages their
// key_start[] in LDS and hands out the KEYS round-robin to its lanes (key idx -> super-k-mer by a binary search in LDS),
// so that lanes are equally loaded whatever the super-k-mer lengths and the key array is accessed fully coalesced
// (one lane per super-k-mer wrote 84-B-strided runs and idled behind the longest super-k-mer of its wave).
constexpr int kSkPerBlock = 256;
__device__ __forceinline__ uint32_t sk_of_key(const uint64_t* s_ks, uint32_t nj, uint64_t idx)
{
	uint32_t lo = 0, hi = nj;  // last j in [0, nj) with s_ks[j] <= idx
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (s_ks[mid] <= idx)
			lo = mid;
		else
			hi = mid;
	}
	return lo;
}

template <int KW>
__global__ __launch_bounds__(kBlock) void k_expand_keys(const uint64_t* __restrict__ src, const uint32_t* __restrict__ order,
                                                        const uint64_t* __restrict__ sk_pos, const uint64_t* __restrict__ key_start, uint64_t j_begin,
                                                        uint64_t j_end, int k, uint64_t* __restrict__ keys)
{
	__shared__ uint64_t s_ks[kSkPerBlock + 1];
	__shared__ uint64_t s_p[kSkPerBlock];
	const uint64_t j0 = j_begin + (uint64_t)blockIdx.x * kSkPerBlock;  // the sorted super-k-mers [j_begin, j_end) (a pass of the build)
	const uint32_t nj = (uint32_t)min((uint64_t)kSkPerBlock, j_end - j0);
	for (uint32_t t = threadIdx.x; t <= nj; t += kBlock)
		s_ks[t] = key_start[j0 + t];
	for (uint32_t t = threadIdx.x; t < nj; t += kBlock)
		s_p[t] = sk_pos[order[j0 + t]];
	__syncthreads();
	const uint64_t k1 = s_ks[nj];
	for (uint64_t idx = s_ks[0] + threadIdx.x; idx < k1; idx += kBlock) {
		const uint32_t j = sk_of_key(s_ks, nj, idx);
		const Kmer<KW> c = kmer_canon<KW>(packed_kmer<KW>(src, s_p[j] + (idx - s_ks[j]), k), k);
		if constexpr (KW == 1) {
			keys[idx] = c.lo;
		} else {
			keys[2 * idx] = c.lo;
			keys[2 * idx + 1] = c.hi;
		}
	}
}

// ---- K7 ----------------------------------------------------------------------------------------------------------------
struct BuildGroups {
	const uint64_t* g_prefix;  // [ngroups + 1], exclusive key prefix
	const GroupDesc* groups;   // wbase / size[]
	const uint64_t* loff;      // [ngroups * 16] level offsets in bits
	uint32_t ngroups;
	const uint8_t* small;      // per group: 1 = built by the LDS-staged kernel (the global kernels skip its keys), may be null
};

// group of key index idx: prefix[g] <= idx < prefix[g+1] (empty groups are skipped by the upper bound)
__device__ __forceinline__ uint32_t group_of_key(const uint64_t* __restrict__ prefix, uint32_t ngroups, uint64_t idx)
{
	uint32_t lo = 0, hi = ngroups;  // answer in [lo, hi)
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (prefix[mid] <= idx)
			lo = mid;
		else
			hi = mid;
	}
	return lo;
}

template <int KW>
__device__ __forceinline__ uint64_t level_bit_address(const BuildGroups& bg, const uint64_t* __restrict__ keys, uint64_t idx, int level,
                                                      uint64_t& mask, uint32_t& g)
{
	g = group_of_key(bg.g_prefix, bg.ngroups, idx);
	const uint64_t key = KW == 1 ? keys[idx] : keys[2 * idx];  // 128-bit keys: low 64 bits only (SURVEY C-7)
	LevelHasher hs(key);
	uint64_t h = 0;
	for (int l = 0; l <= level; l++)
		h = hs.next(l);
	const GroupDesc* gd = &bg.groups[g];
	const uint64_t bit = bg.loff[(uint64_t)g * kLevels + level] + reduce_level(h, gd->size[level]);
	mask = 1ull << (bit & 63);
	return gd->wbase + (bit >> 6);
}

// processLevel_ (bbhash.h:677-709): every surviving key test-and-sets its level bit; a second hit marks the collision map.
template <int KW>
__global__ void k_mphf_mark(BuildGroups bg, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ list, uint64_t count, int level,
                            unsigned long long* __restrict__ bits, unsigned long long* __restrict__ coll)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= count)
		return;
	const uint64_t idx = list ? (uint64_t)list[t] : t;
	uint64_t mask;
	uint32_t g;
	const uint64_t w = level_bit_address<KW>(bg, keys, idx, level, mask, g);
	if (bg.small && bg.small[g])
		return;
	const unsigned long long old = atomicOr(&bits[w], mask);
	if (old & mask)
		atomicOr(&coll[w], mask);
}

// Block-aggregated append: every thread of the block calls this; ONE atomic per block reserves the slots (a single
// hot counter serialises at ~12 ns per atomic on MI355X, so per-wave appends dominated the level time).
__device__ __forceinline__ uint64_t block_append_slot(bool pred, unsigned long long* counter)
{
	__shared__ uint32_t s_wave_cnt[16];
	__shared__ unsigned long long s_base;
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = (blockDim.x + 63) >> 6;
	const uint64_t ballot = __ballot(pred);
	if (lane == 0)
		s_wave_cnt[wave] = (uint32_t)__popcll(ballot);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (uint32_t w = 0; w < nwaves; w++) {
			const uint32_t c = s_wave_cnt[w];
			s_wave_cnt[w] = total;
			total += c;
		}
		s_base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
	}
	__syncthreads();
	return s_base + s_wave_cnt[wave] + (uint64_t)__popcll(ballot & ((1ull << lane) - 1ull));
}

constexpr unsigned kSurviveBlock = 1024;

// keys whose level bit collided go on to the next level (their bit is cleared when the lines are written)
template <int KW>
__global__ __launch_bounds__(kSurviveBlock) void k_mphf_survive(BuildGroups bg, const uint64_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ list, uint64_t count, int level,
                                                                const unsigned long long* __restrict__ coll, uint32_t* __restrict__ next,
                                                                unsigned long long* __restrict__ next_count)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	bool survives = false;
	uint64_t idx = 0;
	if (t < count) {
		idx = list ? (uint64_t)list[t] : t;
		uint64_t mask;
		uint32_t g;
		const uint64_t w = level_bit_address<KW>(bg, keys, idx, level, mask, g);
		survives = !(bg.small && bg.small[g]) && (coll[w] & mask) != 0;
	}
	const uint64_t slot = block_append_slot(survives, next_count);
	if (survives)
		next[slot] = (uint32_t)idx;
}

// ---- K7 (LDS-staged): one workgroup builds ALL levels of one MPHF group ------------------------------------------------
// Global atomicOr executes at the memory side (~20 G scattered atomics/s chip-wide), which made the level kernels above
// the most expensive part of the build.  A group's level-i bit array (2n bits at level 0 for gamma = 2) fits the 160 KB
// LDS for groups of up to ~295 K keys, so one 1024-thread workgroup per group keeps the current level's bit and
// collision arrays in LDS (ds atomics), compacts the survivors into the group's own slice of two global lists, and
// writes the finished level (bits & ~collisions) with plain coalesced stores.  Levels larger than the LDS arrays are
// built in up to kLdsMaxSlices slices (one walk over the level's keys per slice).
constexpr uint32_t kLdsLevelWords = 18432;                 // 32-bit words per LDS array: 72 KB each, two arrays
constexpr uint64_t kLdsLevelBits = (uint64_t)kLdsLevelWords * 32;  // bits per LDS slice (589,824)
constexpr uint32_t kLdsMaxSlices = 4;  // larger groups (> ~1.18 M keys) use the global level kernels
constexpr unsigned kGroupBlock = 1024;

template <int KW>
__global__ __launch_bounds__(kGroupBlock) void k_mphf_group_lds(BuildGroups bg, const uint64_t* __restrict__ keys, uint32_t* __restrict__ listA,
                                                                uint32_t* __restrict__ listB, uint32_t* __restrict__ gbits32,
                                                                uint32_t* __restrict__ fb_count, int* __restrict__ levels_max,
                                                                const uint32_t* __restrict__ order)
{
	__shared__ uint32_t s_bits[kLdsLevelWords];
	__shared__ uint32_t s_coll[kLdsLevelWords];
	__shared__ uint32_t s_cnt;
	const uint32_t g = order[blockIdx.x];  // largest groups first: the multi-slice groups must not be the tail of the grid
	const GroupDesc* gd = &bg.groups[g];
	const uint64_t kb = bg.g_prefix[g];
	const uint32_t n = (uint32_t)gd->nkeys;
	uint32_t* cur = listA + kb;
	uint32_t* nxt = listB + kb;
	uint32_t count = n;
	uint64_t loff = 0;
	int level = 0;
	for (; level < kLevels && count > 0; level++) {
		const uint64_t size = gd->size[level];
		const uint32_t words = (uint32_t)(size >> 5);
		// Each key's level bit is cached between the mark and the survivor pass, so that a key is hashed once per level.
		// The cache borrows list space: at level 0 the current list is implicit, its slice is free; deeper, the tail of the
		// next list's slice is free as long as no more than half of the group's keys are still alive (the appended survivors
		// cannot reach it).  Otherwise (mass duplicates) the survivor pass hashes again.
		uint32_t* bitcache = level == 0 ? cur : (count <= n / 2 ? nxt + (n - count) : nullptr);
		if (threadIdx.x == 0)
			s_cnt = 0;
		// a level larger than the LDS arrays is built in slices of kLdsLevelWords words: every pass walks the level's keys
		// and handles those whose bit falls into the slice (key reads are coalesced and cheap next to global atomics)
		for (uint32_t w0 = 0; w0 < words; w0 += kLdsLevelWords) {
			const uint32_t wn = min(kLdsLevelWords, words - w0);
			for (uint32_t w = threadIdx.x; w < wn; w += kGroupBlock) {
				s_bits[w] = 0;
				s_coll[w] = 0;
			}
			__syncthreads();
			// mark: test-and-set, a second hit marks the collision map (processLevel_, bbhash.h:677-709)
			for (uint32_t i = threadIdx.x; i < count; i += kGroupBlock) {
				const uint32_t j = level == 0 ? i : cur[i];
				LevelHasher hs(KW == 1 ? keys[kb + j] : keys[2 * (kb + j)]);
				uint64_t h = 0;
				for (int l = 0; l <= level; l++)
					h = hs.next(l);
				const uint32_t b = (uint32_t)reduce_level(h, size);
				if (w0 == 0 && bitcache)
					bitcache[i] = b;
				const uint32_t lw = (b >> 5) - w0;
				if (lw < wn) {
					const uint32_t m = 1u << (b & 31);
					if (atomicOr(&s_bits[lw], m) & m)
						atomicOr(&s_coll[lw], m);
				}
			}
			__syncthreads();
			// survivors of this slice (wave-aggregated append into the group's slice of the next list)
			const uint32_t rounds = (count + kGroupBlock - 1) / kGroupBlock;
			for (uint32_t r = 0; r < rounds; r++) {
				const uint32_t i = r * kGroupBlock + threadIdx.x;
				bool surv = false;
				uint32_t j = 0;
				if (i < count) {
					j = level == 0 ? i : cur[i];
					uint32_t b;
					if (bitcache) {
						b = bitcache[i];
					} else {
						LevelHasher hs(KW == 1 ? keys[kb + j] : keys[2 * (kb + j)]);
						uint64_t h = 0;
						for (int l = 0; l <= level; l++)
							h = hs.next(l);
						b = (uint32_t)reduce_level(h, size);
					}
					const uint32_t lw = (b >> 5) - w0;
					surv = lw < wn && ((s_coll[lw] >> (b & 31)) & 1u);
				}
				const unsigned long long bal = __ballot(surv);
				if (bal) {
					const uint32_t lane = threadIdx.x & 63;
					uint32_t base = 0;
					const int leader = __ffsll(bal) - 1;
					if ((int)lane == leader)
						base = atomicAdd(&s_cnt, (uint32_t)__popcll(bal));
					base = __shfl(base, leader);
					if (surv)
						nxt[base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = j;
				}
			}
			__syncthreads();
			// the finished slice: bits that exactly one key claimed (clearCollisions, bbhash.h:358-368)
			uint32_t* out = gbits32 + 2 * gd->wbase + (loff >> 5) + w0;
			for (uint32_t w = threadIdx.x; w < wn; w += kGroupBlock)
				out[w] = s_bits[w] & ~s_coll[w];
			__syncthreads();
		}
		const uint32_t c = s_cnt;
		__syncthreads();
		count = c;
		loff += size;
		uint32_t* t = cur;
		cur = nxt;
		nxt = t;
	}
	if (threadIdx.x == 0) {
		fb_count[g] = count;  // keys left after 16 levels stay in `cur` (their positions in the group)
		atomicMax(levels_max, level);
	}
	// leave the survivors where the host-side merge expects them: listA slice
	if (count > 0 && cur != listA + kb) {
		for (uint32_t i = threadIdx.x; i < count; i += kGroupBlock)
			listA[kb + i] = cur[i];
	}
}

// ---- K7 (LDS-staged, one level of one SLICE per workgroup): groups above the single-workgroup budget ----------------------------
// A group of more than ~1.18 M keys (low-complexity sequence puts millions of k-mers under a few minimizers) would leave LDS for
// memory-side atomics (~2e10 scattered atomics/s chip-wide: 340 ms for the 2.5e8 keys of the skewed human-scale input).  Here a
// level of such a group is cut into slices of kLdsLevelBits bits and every (group, slice) pair of the level gets its own
// workgroup: it walks the group's current key list, marks the keys that hash into its slice in LDS, reserves room for its
// survivors in the group's next list with ONE global atomic (their number = keys in the slice - bits finally set, no extra
// walk), walks the list again to write them, and stores the finished slice.  Keys are hashed once per walk and slice, which
// is cheap next to atomics; one launch per level, workgroups of finished groups exit at once.
// lvcnt[level * ngroups + g] = keys of group g alive at `level` (level 0: all keys, implicit list).
// r04: the level-`level` bit index of every key a sliced group still holds, ONCE per level (blockIdx.y = the group's number in glist,
// blockIdx.x = a chunk of its current key list).  The slice workgroups below then walk 4-byte bit indices: before, each of the up to
// 45 slices of a 13-M-key group hashed every key of the group itself, twice (mark and survive walks) -- 6e8 hash_bis per group
// and level 0 -- and the deep levels of groups full of duplicated keys re-hashed every survivor from level 0 in every walk.
template <int KW>
__global__ __launch_bounds__(kBlock) void k_sliced_level_bits(BuildGroups bg, const uint64_t* __restrict__ keys, int level, const uint32_t* __restrict__ glist,
                                                              const uint32_t* __restrict__ listA, const uint32_t* __restrict__ listB,
                                                              const uint32_t* __restrict__ lvcnt, uint32_t* __restrict__ lbits)
{
	const uint32_t g = glist[blockIdx.y];
	const GroupDesc* gd = &bg.groups[g];
	const uint64_t kb = bg.g_prefix[g];
	const uint32_t count = level == 0 ? (uint32_t)gd->nkeys : lvcnt[(uint64_t)level * bg.ngroups + g];
	const uint64_t size = gd->size[level];
	const uint32_t* cur = ((level & 1) ? listB : listA) + kb;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (uint64_t)gridDim.x * kBlock) {
		const uint32_t j = level == 0 ? (uint32_t)i : cur[i];
		const uint64_t key = KW == 1 ? keys[kb + j] : keys[2 * (kb + j)];
		LevelHasher hs(key);
		uint64_t h = 0;
		for (int l = 0; l <= level; l++)
			h = hs.next(l);
		lbits[kb + i] = (uint32_t)reduce_level(h, size);
	}
}

template <int KW>
__global__ __launch_bounds__(kGroupBlock) void k_mphf_level_sliced(BuildGroups bg, const uint64_t* __restrict__ keys, int level,
                                                                   const uint2* __restrict__ items, uint32_t* __restrict__ listA,
                                                                   uint32_t* __restrict__ listB, uint32_t* __restrict__ lvcnt,
                                                                   uint32_t* __restrict__ gbits32, const uint32_t* __restrict__ lbits)
{
	__shared__ uint32_t s_bits[kLdsLevelWords];
	__shared__ uint32_t s_coll[kLdsLevelWords];
	__shared__ uint32_t s_in, s_set, s_base, s_cnt;
	const uint32_t g = items[blockIdx.x].x, sl = items[blockIdx.x].y;
	const GroupDesc* gd = &bg.groups[g];
	const uint64_t kb = bg.g_prefix[g];
	const uint32_t count = level == 0 ? (uint32_t)gd->nkeys : lvcnt[(uint64_t)level * bg.ngroups + g];
	if (count == 0)
		return;
	const uint64_t size = gd->size[level];
	const uint32_t words = (uint32_t)(size >> 5), w0 = sl * kLdsLevelWords;
	if (w0 >= words)
		return;
	const uint32_t wn = min(kLdsLevelWords, words - w0);
	const uint32_t* cur = ((level & 1) ? listB : listA) + kb;  // survivors of level - 1 (level 0: implicit)
	uint32_t* nxt = ((level & 1) ? listA : listB) + kb;
	for (uint32_t w = threadIdx.x; w < wn; w += kGroupBlock) {
		s_bits[w] = 0;
		s_coll[w] = 0;
	}
	if (threadIdx.x == 0)
		s_in = s_set = s_cnt = 0;
	__syncthreads();
	// mark (processLevel_, bbhash.h:677-709) + count the keys that fall into this slice.  The walk is a stream of dependent-free loads
	// and every slice of the group walks ALL of the group's bit indices: 16-byte loads, two per thread in flight (r04 before: 4-byte
	// loads, four in flight -- 16 KB in flight per CU is 4 GB/s per workgroup, and the giant groups of the skewed inputs spent 25 ms in
	// level 0 alone).  [0, head) and [tail0, count): the few indices in front of / behind the 16-byte-aligned quads.
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	constexpr uint32_t kIlp = 2;
	const uint32_t* __restrict__ mybits = lbits + kb;  // bit index of the i-th key of the group's current list (k_sliced_level_bits)
	(void)keys;
	const uint32_t head = min(count, (4u - (uint32_t)((reinterpret_cast<uintptr_t>(mybits) >> 2) & 3u)) & 3u);
	const uint32_t nquads = (count - head) >> 2, tail0 = head + (nquads << 2);
	const u32x4* __restrict__ quads = reinterpret_cast<const u32x4*>(mybits + head);
	const uint32_t rounds = (nquads + kGroupBlock * kIlp - 1) / (kGroupBlock * kIlp);
	uint32_t mine = 0;
	auto mark = [&](uint32_t b) {
		const uint32_t lw = (b >> 5) - w0;
		if (lw < wn) {
			const uint32_t m = 1u << (b & 31);
			if (atomicOr(&s_bits[lw], m) & m)
				atomicOr(&s_coll[lw], m);
			mine++;
		}
	};
	if (threadIdx.x < head)
		mark(mybits[threadIdx.x]);
	if (tail0 + threadIdx.x < count)
		mark(mybits[tail0 + threadIdx.x]);
	for (uint32_t r = 0; r < rounds; r++) {
		u32x4 qv[kIlp];
		bool in[kIlp];
#pragma unroll
		for (uint32_t u = 0; u < kIlp; u++) {
			const uint32_t q = (r * kIlp + u) * kGroupBlock + threadIdx.x;
			in[u] = q < nquads;
			qv[u] = in[u] ? quads[q] : u32x4{0u, 0u, 0u, 0u};
		}
#pragma unroll
		for (uint32_t u = 0; u < kIlp; u++) {
			if (!in[u])
				continue;
			mark(qv[u].x);
			mark(qv[u].y);
			mark(qv[u].z);
			mark(qv[u].w);
		}
	}
	for (int o = 32; o > 0; o >>= 1)
		mine += __shfl_xor(mine, o);
	if ((threadIdx.x & 63) == 0 && mine)
		atomicAdd(&s_in, mine);
	__syncthreads();
	// the finished slice (clearCollisions, bbhash.h:358-368) and the number of bits it keeps
	uint32_t* out = gbits32 + 2 * gd->wbase + (uint32_t)(bg.loff[(uint64_t)g * kLevels + level] >> 5) + w0;
	uint32_t kept = 0;
	for (uint32_t w = threadIdx.x; w < wn; w += kGroupBlock) {
		const uint32_t v = s_bits[w] & ~s_coll[w];
		out[w] = v;
		kept += __popc(v);
	}
	for (int o = 32; o > 0; o >>= 1)
		kept += __shfl_xor(kept, o);
	if ((threadIdx.x & 63) == 0 && kept)
		atomicAdd(&s_set, kept);
	__syncthreads();
	const uint32_t nsurv = s_in - s_set;  // every key of the slice either keeps its bit for itself or goes on
	if (nsurv == 0)
		return;
	if (threadIdx.x == 0)
		s_base = atomicAdd(&lvcnt[(uint64_t)(level + 1) * bg.ngroups + g], nsurv);
	__syncthreads();
	// the survivors (keys whose bit collided) go to the group's next list: one LDS atomic per wave and element, called by whole waves
	auto emit = [&](bool surv, uint32_t idx) {
		const unsigned long long bal = __ballot(surv);
		if (bal) {
			const uint32_t lane = threadIdx.x & 63;
			uint32_t base = 0;
			const int leader = __ffsll(bal) - 1;
			if ((int)lane == leader)
				base = atomicAdd(&s_cnt, (uint32_t)__popcll(bal));
			base = __shfl(base, leader);
			if (surv)  // (the key's index: only survivors read the list)
				nxt[s_base + base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = level == 0 ? idx : cur[idx];
		}
	};
	auto collided = [&](uint32_t b) {
		const uint32_t lw = (b >> 5) - w0;
		return lw < wn && ((s_coll[lw] >> (b & 31)) & 1u);
	};
	{
		const bool h = threadIdx.x < head, t = tail0 + threadIdx.x < count;
		emit(h && collided(mybits[h ? threadIdx.x : 0]), threadIdx.x);
		emit(t && collided(mybits[t ? tail0 + threadIdx.x : 0]), tail0 + threadIdx.x);
	}
	for (uint32_t r = 0; r < rounds; r++) {
		u32x4 qv[kIlp];
		uint32_t qi[kIlp];
		bool in[kIlp];
#pragma unroll
		for (uint32_t u = 0; u < kIlp; u++) {
			const uint32_t q = (r * kIlp + u) * kGroupBlock + threadIdx.x;
			in[u] = q < nquads;
			qi[u] = head + (q << 2);
			qv[u] = in[u] ? quads[q] : u32x4{0u, 0u, 0u, 0u};
		}
#pragma unroll
		for (uint32_t u = 0; u < kIlp; u++) {
			emit(in[u] && collided(qv[u].x), qi[u]);
			emit(in[u] && collided(qv[u].y), qi[u] + 1);
			emit(in[u] && collided(qv[u].z), qi[u] + 2);
			emit(in[u] && collided(qv[u].w), qi[u] + 3);
		}
	}
}

// after the last level: survivors of every sliced group -> fb_count (they sit in listA like those of k_mphf_group_lds), levels used
__global__ void k_sliced_finish(const uint32_t* __restrict__ glist, uint32_t nlarge, uint32_t ngroups, const uint32_t* __restrict__ lvcnt,
                                uint32_t* __restrict__ fb_count, int* __restrict__ levels_max)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nlarge)
		return;
	const uint32_t g = glist[t];
	int used = 1;
	for (int l = 1; l < kLevels; l++)
		if (lvcnt[(uint64_t)l * ngroups + g])
			used = l + 1;
	fb_count[g] = lvcnt[(uint64_t)kLevels * ngroups + g];
	atomicMax(levels_max, used);
}

// first work list of the global level kernels: the key indices of the groups that do NOT fit the LDS path
__global__ void k_seed_large_list(const uint64_t* __restrict__ g_prefix, const uint8_t* __restrict__ small, const uint64_t* __restrict__ seed_off,
                                  uint32_t ngroups, uint32_t* __restrict__ list, uint32_t g_base)
{
	const uint32_t g = g_base + blockIdx.x;
	if (g >= ngroups || small[g])
		return;
	const uint64_t kb = g_prefix[g], n = g_prefix[g + 1] - kb, o = seed_off[g];
	for (uint64_t i = threadIdx.x; i < n; i += blockDim.x)
		list[o + i] = (uint32_t)(kb + i);
}

__global__ void k_widen_u32(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		out[i] = in[i];
}

// fallback keys of the LDS-built groups -> global key indices appended to the fallback list
__global__ void k_fb_collect(const uint64_t* __restrict__ g_prefix, const uint32_t* __restrict__ fb_count, const uint64_t* __restrict__ fb_off,
                             const uint32_t* __restrict__ listA, uint32_t ngroups, uint64_t* __restrict__ fb_list, uint32_t g_base)
{
	const uint32_t g = g_base + blockIdx.x;
	if (g >= ngroups)
		return;
	const uint32_t c = fb_count[g];
	const uint64_t kb = g_prefix[g];
	for (uint32_t i = threadIdx.x; i < c; i += blockDim.x)
		fb_list[fb_off[g] + i] = kb + listA[kb + i];  // absolute key index: 64 bits (an index may hold 2^32 k-mers or more)
}

// ---- K8 ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t group_of_line(const uint64_t* __restrict__ line_base, uint32_t ngroups, uint64_t line)
{
	uint32_t lo = 0, hi = ngroups;
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (line_base[mid] <= line)
			lo = mid;
		else
			hi = mid;
	}
	return lo;
}

// Level bits of line L of its group: logical 32-bit words 3*(L - line_base) .. +2 of the group's level array (the u64
// build arrays viewed as little-endian u32 pairs), collisions cleared (bbhash.h:358-368).
__device__ __forceinline__ uint32_t line_data_word(const GroupDesc* __restrict__ gd, const unsigned long long* __restrict__ bits,
                                                   const unsigned long long* __restrict__ coll, uint64_t w32)
{
	if (w32 >= 2 * gd->nwords)
		return 0u;
	const uint64_t w = gd->wbase + (w32 >> 1);
	const unsigned long long v = bits[w] & ~coll[w];
	return (uint32_t)((w32 & 1) ? (v >> 32) : v);
}

// A group's lines in rank order ("logical lines"): without fat lines, line l covers the 32-bit words 3l .. 3l+2 of all levels back to
// back; with fat lines (index_device.h) the first F = level0_units64(size0) lines cover level 0 alone -- 96 bits each, the last of
// them padded with zeros, or 32 bits each when they are records -- and the lines behind them the deeper levels.  line_base[] counts
// LOGICAL lines (rank bookkeeping); GroupDesc::line_base counts 16-B units of the line array (a fat line or record takes four).
// layout: 0 = 16-B lines only, 1 = fat level-0 lines, 2 = level-0 records
__device__ __forceinline__ uint32_t logical_line_word(const GroupDesc* __restrict__ gd, const unsigned long long* __restrict__ bits,
                                                      const unsigned long long* __restrict__ coll, uint64_t lrel, int s, int layout)
{
	if (layout == 0)
		return line_data_word(gd, bits, coll, lrel * kLineData32 + s);
	const uint64_t w0 = gd->size[0] >> 5, F = level0_units64(gd->size[0], layout == 2);
	if (lrel < F) {
		if (layout == 2)
			return s == 0 ? line_data_word(gd, bits, coll, lrel) : 0u;
		const uint64_t w32 = lrel * kLineData32 + s;
		return w32 < w0 ? line_data_word(gd, bits, coll, w32) : 0u;
	}
	return line_data_word(gd, bits, coll, w0 + (lrel - F) * kLineData32 + s);
}
__device__ __forceinline__ uint64_t logical_line_unit(const GroupDesc* __restrict__ gd, uint64_t lrel, int layout)
{
	if (layout == 0)
		return gd->line_base + lrel;
	const uint64_t F = level0_units64(gd->size[0], layout == 2);  // (cross-paired 128-B blocks of two units: index_device.h)
	return lrel < F ? gd->line_base + (lrel >> 1) * kBlockUnits + (lrel & 1) * 4 : gd->line_base + level0_block_units(gd->size[0], layout == 2) + (lrel - F);
}

__global__ void k_line_popc(const GroupDesc* __restrict__ groups, const uint64_t* __restrict__ line_base, uint32_t ngroups, uint64_t L0, uint64_t nlines,
                            const unsigned long long* __restrict__ bits, const unsigned long long* __restrict__ coll,
                            uint64_t* __restrict__ linepc, int layout)
{
	const uint64_t L = L0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // the logical lines [L0, nlines) (a pass of the build)
	linepc -= L0;
	if (L > nlines)
		return;
	if (L == nlines) {
		linepc[L] = 0;
		return;
	}
	const uint32_t g = group_of_line(line_base, ngroups, L);
	const GroupDesc* gd = &groups[g];
	uint64_t pc = 0;
	for (int s = 0; s < kLineData32; s++)
		pc += (uint64_t)__popc(logical_line_word(gd, bits, coll, L - line_base[g], s, layout));
	linepc[L] = pc;
}

__global__ void k_line_write(const GroupDesc* __restrict__ groups, const uint64_t* __restrict__ line_base, uint32_t ngroups, uint64_t L0, uint64_t nlines,
                             const unsigned long long* __restrict__ bits, const unsigned long long* __restrict__ coll,
                             const uint64_t* __restrict__ linepre, uint4* __restrict__ lines, int layout)
{
	const uint64_t L = L0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	linepre -= L0;
	if (L >= nlines)
		return;
	const uint32_t g = group_of_line(line_base, ngroups, L);
	const GroupDesc* gd = &groups[g];
	const uint64_t lrel = L - line_base[g];
	uint4 o;
	o.x = (uint32_t)(linepre[L] - linepre[line_base[g]]);  // rank restarts at 0 in every group (bbhash.h:447-465)
	o.y = logical_line_word(gd, bits, coll, lrel, 0, layout);
	o.z = logical_line_word(gd, bits, coll, lrel, 1, layout);
	o.w = logical_line_word(gd, bits, coll, lrel, 2, layout);
	// (a fat line's filter words stay zero here: the filter passes set them; a record's position fields -- o.z, o.w are the first two
	// words of them, zero here -- are written by k_record_positions once the pass's positions are packed)
	lines[logical_line_unit(gd, lrel, layout)] = o;
}

// Level-0 records: copy the position fields of the keys a record ranks from `positions` into the record (index_device.h).  One lane
// per logical line of the pass (the deeper-level lines leave at once); runs behind k_pack_positions of the same groups.
__global__ void k_record_positions(const GroupDesc* __restrict__ groups, const uint64_t* __restrict__ line_base, uint32_t ngroups, uint64_t L0, uint64_t nlines,
                                   const uint64_t* __restrict__ positions, uint4* __restrict__ lines)
{
	const uint64_t L = L0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (L >= nlines)
		return;
	const uint32_t g = group_of_line(line_base, ngroups, L);
	const GroupDesc* gd = &groups[g];
	const uint64_t lrel = L - line_base[g];
	if (lrel >= (gd->size[0] >> 5))
		return;
	uint32_t* recp = reinterpret_cast<uint32_t*>(lines + gd->line_base + (lrel >> 1) * kBlockUnits + (lrel & 1) * 4);
	const uint32_t rank = recp[0], nset = (uint32_t)__popc(recp[1]), nbits = gd->nbits;
	const uint32_t nf = min(nset, rec_fields(nbits));
	uint32_t w[kRecPosBits / 32];
	for (int i = 0; i < kRecPosBits / 32; i++)
		w[i] = 0;
	for (uint32_t j = 0; j < nf; j++) {
		const uint64_t f = read_position_field(positions, gd->posstart, nbits, (uint64_t)rank + j);
		const uint32_t o = j * nbits, wi = o >> 5, sh = o & 31u;
		const uint64_t v = f << sh;
		for (int i = 0; i < kRecPosBits / 32; i++) {  // (constant indices: w stays in registers)
			if ((uint32_t)i == wi)
				w[i] |= (uint32_t)v;
			if ((uint32_t)i == wi + 1)
				w[i] |= (uint32_t)(v >> 32);
		}
	}
	for (int i = 0; i < kRecPosBits / 32; i++)
		recp[kRecPosWord + i] = w[i];
}

// total set bits per group (= first fallback rank, bbhash.h:554)
__global__ void k_group_popc(const uint64_t* __restrict__ line_base, const uint64_t* __restrict__ linepre, uint32_t g_begin, uint32_t g_end, uint64_t L0,
                             uint64_t* __restrict__ g_popc)
{
	const uint64_t g = g_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= g_end)
		return;
	g_popc[g] = linepre[line_base[g + 1] - L0] - linepre[line_base[g] - L0];
}

// ---- K9: fallback -----------------------------------------------------------------------------------------------------
// processFallbackLevel (bbhash.h:711-728): keys left after 16 levels, in input (key index) order, get ranks R, R+1, ...
template <int KW>
__global__ void k_fb_prepare(BuildGroups bg, const uint64_t* __restrict__ keys, const uint64_t* __restrict__ fb_sorted, uint64_t F,
                             const uint64_t* __restrict__ g_popc, uint64_t* __restrict__ fk, uint64_t* __restrict__ frank,
                             uint32_t* __restrict__ fgrp)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= F)
		return;
	const uint64_t idx = fb_sorted[t];
	const uint32_t g = group_of_key(bg.g_prefix, bg.ngroups, idx);
	// first fallback entry of this group: lower bound of prefix[g] in the sorted index list
	uint64_t lo = 0, hi = F;
	const uint64_t first_key = bg.g_prefix[g];
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (fb_sorted[mid] < first_key)
			lo = mid + 1;
		else
			hi = mid;
	}
	frank[t] = g_popc[g] + (t - lo);
	fgrp[t] = g;
	if constexpr (KW == 1) {
		fk[t] = keys[idx];
	} else {
		fk[2 * t] = keys[2 * idx];
		fk[2 * t + 1] = keys[2 * idx + 1];
	}
}

// sort key of fallback entry perm[e] (perm == nullptr: identity, and the identity is written to iota_out):
// which = 0 low key word, 1 high key word (128-bit k-mers), 2 group
__global__ void k_fb_sortkey(const uint64_t* __restrict__ fk, const uint32_t* __restrict__ fgrp, const uint32_t* __restrict__ perm, uint64_t F, int KW,
                             int which, uint64_t* __restrict__ key_out, uint32_t* __restrict__ iota_out)
{
	const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= F)
		return;
	const uint32_t t = perm ? perm[e] : (uint32_t)e;
	if (iota_out)
		iota_out[e] = (uint32_t)e;
	key_out[e] = which == 2 ? (uint64_t)fgrp[t] : fk[(uint64_t)KW * t + (which == 1 ? 1 : 0)];
}

template <int KW>
__global__ void k_fb_table(const uint32_t* __restrict__ perm, uint64_t F, const uint64_t* __restrict__ fk, const uint64_t* __restrict__ frank,
                           const uint32_t* __restrict__ fgrp, uint64_t* __restrict__ tk, uint64_t* __restrict__ trank,
                           uint32_t* __restrict__ tgrp, uint64_t* __restrict__ g_fb_start, uint32_t* __restrict__ g_fb_count)
{
	const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= F)
		return;
	const uint32_t t = perm[e];
	const uint32_t g = fgrp[t];
	if constexpr (KW == 1) {
		tk[e] = fk[t];
	} else {
		tk[2 * e] = fk[2 * t];
		tk[2 * e + 1] = fk[2 * t + 1];
	}
	trank[e] = frank[t];
	tgrp[e] = g;
	if (e == 0 || fgrp[perm[e - 1]] != g)
		g_fb_start[g] = e;
	if (e + 1 == F || fgrp[perm[e + 1]] != g)
		g_fb_count[g] = (uint32_t)(e + 1);  // END of the group's run (the host subtracts the start): no per-entry atomics on a few hot counters
}

// ---- positions -----------------------------------------------------------------------------------------------------------
// fill_positions (blight.cpp:486-519): one lane per key (load-balanced over the block's super-k-mers) looks its rank up and
// stores the position field in a 32-bit array indexed by rank.  Keys that live in the fallback table are written by
// k_fb_positions (a duplicated key keeps the position of its LAST occurrence: assignment order).
template <int KW>
__global__ __launch_bounds__(kBlock) void k_fill_positions(DeviceIndexView v, const uint32_t* __restrict__ smini, const uint64_t* __restrict__ nt_start,
                                                           const uint64_t* __restrict__ key_start, uint64_t j_begin, uint64_t nsk, const uint64_t* __restrict__ keys,
                                                           uint64_t id_base, uint32_t* __restrict__ field_of_rank)
{
	__shared__ uint64_t s_ks[kSkPerBlock + 1];
	__shared__ uint64_t s_pos0[kSkPerBlock];
	__shared__ uint32_t s_mini[kSkPerBlock];
	// (the sorted super-k-mers [j_begin, nsk): a pass of the build)
	// XCD-major block order (workgroup b runs on XCD b % 8; the grid is a multiple of 8): XCD x takes the x-th contiguous
	// eighth of the sorted super-k-mers, so that the ~600 KB rank range a group's scattered 4-B stores fall into is owned by
	// ONE L2, which merges them into full lines, instead of being written in pieces from all eight (90 -> 65 ms at C4)
	const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
	const uint64_t j0 = j_begin + (uint64_t)blk * kSkPerBlock;
	if (j0 >= nsk)
		return;
	const uint32_t nj = (uint32_t)min((uint64_t)kSkPerBlock, nsk - j0);
	for (uint32_t t = threadIdx.x; t <= nj; t += kBlock)
		s_ks[t] = key_start[j0 + t];
	for (uint32_t t = threadIdx.x; t < nj; t += kBlock) {
		const uint32_t mi = smini[j0 + t];
		s_mini[t] = mi;
		s_pos0[t] = nt_start[j0 + t] - v.buckets[mi].start;
	}
	__syncthreads();
	const uint64_t k1 = s_ks[nj];
	for (uint64_t idx = s_ks[0] + threadIdx.x; idx < k1; idx += kBlock) {
		const uint32_t j = sk_of_key(s_ks, nj, idx);
		const GroupDesc* gd = &v.groups[s_mini[j] >> v.gshift];
		const GroupHead gh = load_group_head(gd);
		const Kmer<KW> x = load_kmer<KW>(keys, idx);
		const uint64_t h = mphf_lookup_levels<KW>(v, gd, gh, x);
		if (h != kNotFound)  // ranks are distinct: a plain 4-B store, no atomics (k_pack_positions makes the bit fields)
			field_of_rank[gh.prefix - id_base + h] = (uint32_t)((s_pos0[j] + (idx - s_ks[j])) >> (v.b - v.hint));
	}
}

template <int KW>
__global__ void k_fb_positions(DeviceIndexView v, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ fb_sorted, uint64_t F,
                               const uint64_t* __restrict__ tk, const uint64_t* __restrict__ trank, const uint32_t* __restrict__ tgrp,
                               const uint32_t* __restrict__ smini, const uint64_t* __restrict__ nt_start,
                               const uint64_t* __restrict__ key_start, uint64_t nsk, uint64_t id_base, uint32_t* __restrict__ field_of_rank)
{
	const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= F)
		return;
	// winner = last entry of its (group, key) run
	if (e + 1 < F && tgrp[e + 1] == tgrp[e] && load_kmer<KW>(tk, e + 1) == load_kmer<KW>(tk, e))
		return;
	const uint64_t idx = fb_sorted[perm[e]];
	// super-k-mer of key idx: key_start[j] <= idx < key_start[j+1]
	uint64_t lo = 0, hi = nsk;
	while (hi - lo > 1) {
		const uint64_t mid = (lo + hi) >> 1;
		if (key_start[mid] <= idx)
			lo = mid;
		else
			hi = mid;
	}
	const uint32_t mi = smini[lo];
	const GroupDesc* gd = &v.groups[mi >> v.gshift];
	const uint64_t pos = nt_start[lo] - v.buckets[mi].start + (idx - key_start[lo]);
	field_of_rank[gd->prefix - id_base + trank[e]] = (uint32_t)(pos >> (v.b - v.hint));
}

// Absent-key filter, first pass: every key (also the duplicates and the ones the fallback table ranks) sets its five bits in the
// word its group's region assigns it.  Same walk over the keys as k_fill_positions; the atomics of a workgroup fall into the few
// hundred KB of one group's region (XCD-major block order: one L2 owns it).
template <int KW>
__global__ __launch_bounds__(kBlock) void k_filter_keys(DeviceIndexView v, const uint32_t* __restrict__ smini, const uint64_t* __restrict__ key_start,
                                                        uint64_t j_begin, uint64_t nsk, const uint64_t* __restrict__ keys, unsigned long long* __restrict__ filter)
{
	__shared__ uint64_t s_ks[kSkPerBlock + 1];
	__shared__ uint32_t s_mini[kSkPerBlock];
	const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
	const uint64_t j0 = j_begin + (uint64_t)blk * kSkPerBlock;
	if (j0 >= nsk)
		return;
	const uint32_t nj = (uint32_t)min((uint64_t)kSkPerBlock, nsk - j0);
	for (uint32_t t = threadIdx.x; t <= nj; t += kBlock)
		s_ks[t] = key_start[j0 + t];
	for (uint32_t t = threadIdx.x; t < nj; t += kBlock)
		s_mini[t] = smini[j0 + t];
	__syncthreads();
	const uint64_t k1 = s_ks[nj];
	for (uint64_t idx = s_ks[0] + threadIdx.x; idx < k1; idx += kBlock) {
		const uint32_t j = sk_of_key(s_ks, nj, idx);
		const GroupDesc* gd = &v.groups[s_mini[j] >> v.gshift];
		const Kmer<KW> x = load_kmer<KW>(keys, idx);
		const uint64_t fh = filter_hash<KW>(x);
		if (v.fat)
			atomicOr(fat_filter_slot<KW>(v, gd->line_base, gd->size[0], x, fh), (unsigned long long)filter_mask(fh));
		else
			atomicOr(&filter[filter_word(gd->prefix - v.id_base, gd->size[0], v.filter_bpk, fh)], (unsigned long long)filter_mask(fh));
	}
}

// The absent-key filter must never reject a k-mer the reference answers.  Besides the keys, the reference's 2^b scan
// (blight.cpp:729-739) can find k-mers that are NOT keys: it compares whatever it reads from the bucket's concatenated sequence,
// and a k-mer read across the junction of two super-k-mers (or past the end of the bucket, into the next one) is not a key of
// anybody.  A query for such a k-mer x is answered with an ID iff the MPHF of ITS minimizer's group happens to send x to a rank
// whose position window covers the place where x can be read.  One lane per k-mer that starts in the last k-1 nucleotides of a
// super-k-mer: canonical form, its own minimizer -> the bucket a query would look in, and exactly the query's test; the few that
// pass (a 2^b window among a bucket's positions) are added to the filter.
// Candidates [0, junction_cand) are the junctions between super-k-mers (junction_cand = (nsk - 1) (k - 1)), the ones behind them the
// end of the sequence; a launch with junction_cand = 0 and nt_start = nullptr runs the end junction alone (a shard whose halo -- the
// head of the sequence that follows it in the single index -- arrives after its own build: shard_halo_set).
template <int KW>
__global__ __launch_bounds__(kBlock) void k_filter_junctions(DeviceIndexView v, const uint64_t* __restrict__ nt_start, uint64_t junction_cand,
                                                             unsigned long long* __restrict__ filter)
{
	const uint64_t idx = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
	const uint32_t per = (uint32_t)v.k - 1u;
	uint64_t a;  // absolute position in the packed sequence
	if (idx < junction_cand) {
		const uint64_t j = idx / per + 1;
		a = nt_start[j] - per + (uint32_t)(idx - (j - 1) * per);  // the k-mer ends inside super-k-mer j
	} else {
		// ... and the end of the whole sequence is a junction too: the windows of the last buckets run past it into the zero padding
		// (the reference reads its vector<bool> out of range there; here and in the oracle those reads are defined as 'A') -- or, on a
		// shard, into its halo: the nucleotides the single index has there.  The k-mers that start in the last k-1 nucleotides or
		// behind the end where a window can still reach are candidates like the others
		const uint64_t e = idx - junction_cand;
		if (e >= (uint64_t)per + (1ull << v.b) || v.total_nt < per)
			return;
		a = v.total_nt - per + e;
	}
	const Kmer<KW> y = packed_kmer<KW>(v.seq, a, v.k);
	const Kmer<KW> yr = kmer_rc<KW>(y, v.k);
	const bool fwd = !(yr < y);
	const Kmer<KW> x = fwd ? y : yr, xr = fwd ? yr : y;
	const uint32_t mini = kmer_minimizer<KW>(x, xr, v.k, v.m);
	const BucketDesc bd = v.buckets[mini];
	if (bd.nt == 0 || a < bd.start)
		return;
	const uint64_t rel = a - bd.start;
	if (rel >= (uint64_t)bd.nt + (1ull << v.b))  // beyond every window of that bucket
		return;
	const GroupDesc* gd = &v.groups[mini >> v.gshift];
	const GroupHead gh = load_group_head(gd);
	const uint64_t h = mphf_lookup<KW>(v, gd, gh, x);
	if (h == kNotFound)
		return;
	const uint64_t pos = read_position(v.positions, gh.posstart, gh.nbits, h, v.hint, v.b);
	if (!(pos + (uint64_t)v.k - 1 < (uint64_t)bd.nt) || rel < pos || rel >= pos + (1ull << v.b))
		return;
	const uint64_t fh = filter_hash<KW>(x);
	if (v.fat)
		atomicOr(fat_filter_slot<KW>(v, gh.line_base, gh.size0, x, fh), (unsigned long long)filter_mask(fh));
	else
		atomicOr(&filter[filter_word(gh.prefix - v.id_base, gh.size0, v.filter_bpk, fh)], (unsigned long long)filter_mask(fh));
}

// int_to_bool for a whole group (blight.cpp:464-469): one workgroup packs the group's 32-bit fields (indexed by rank)
// into its nbits-wide little-endian slice of `positions`.  Per chunk of up to 512 output words the fields that touch the chunk
// are staged in LDS with coalesced loads, then one lane assembles one 64-bit word (all offsets relative to the chunk:
// 32-bit arithmetic).  Only the first and last word of a slice can be shared with the neighbouring groups: those are
// ORed atomically (the array is zero-initialised), every other word is a plain store.
constexpr int kPackThreads = 512;       // one output word per thread and chunk
constexpr uint32_t kPackFields = 16384;  // fields staged per chunk (64 KB of LDS)
__global__ __launch_bounds__(kPackThreads) void k_pack_positions(const GroupDesc* __restrict__ groups, uint32_t g_base, uint64_t id_base,
                                                                 const uint32_t* __restrict__ field_of_rank, unsigned long long* __restrict__ positions)
{
	__shared__ uint32_t s_f[kPackFields];
	const GroupDesc* gd = &groups[g_base + blockIdx.x];
	const uint64_t n = gd->nkeys;
	const uint32_t nbits = gd->nbits;
	if (n == 0 || nbits == 0)
		return;
	const uint64_t ps = gd->posstart, pe = ps + n * nbits;
	const uint32_t* __restrict__ f = field_of_rank + (gd->prefix - id_base);
	const uint64_t w_first = ps >> 6, w_last = (pe - 1) >> 6;
	const uint32_t fmask = nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u);
	// words per chunk: as many as there are threads, fewer when the fields that touch them would not fit the LDS stage
	// (a chunk of W words is touched by at most W * 64 / nbits + 2 fields)
	const uint32_t chunk = min((uint32_t)kPackThreads, max(1u, (uint32_t)(((uint64_t)(kPackFields - 2) * nbits) >> 6)));
	for (uint64_t wc = w_first; wc <= w_last; wc += chunk) {
		const uint64_t wend = min(wc + (uint64_t)chunk - 1, w_last);
		const uint64_t cb0 = max(wc << 6, ps), cb1 = min((wend << 6) + 64, pe);  // bits of the slice inside this chunk
		const uint64_t hA = (cb0 - ps) / nbits, hB = (cb1 - 1 - ps) / nbits;     // fields touching the chunk (uniform)
		const uint32_t nf = (uint32_t)(hB - hA + 1);
		__syncthreads();  // the previous chunk has been consumed
		for (uint32_t i = threadIdx.x; i < nf; i += kPackThreads)
			s_f[i] = f[hA + i] & fmask;
		__syncthreads();
		const uint64_t w = wc + threadIdx.x;
		if (threadIdx.x < chunk && w <= wend) {
			const int32_t r0 = (int32_t)((int64_t)(w << 6) - (int64_t)(ps + hA * nbits));  // word start relative to field hA
			const int32_t r1 = (int32_t)((int64_t)min((w << 6) + 64, pe) - (int64_t)(ps + hA * nbits));
			const uint32_t lo = r0 < 0 ? 0u : (uint32_t)r0;
			const uint32_t i0 = lo / nbits, i1 = ((uint32_t)r1 - 1u) / nbits;
			uint64_t acc = 0;
			for (uint32_t i = i0; i <= i1; i++) {
				const uint64_t val = s_f[i];
				const int32_t o = (int32_t)(i * nbits) - r0;
				acc |= o >= 0 ? (val << o) : (val >> (-o));
			}
			if ((w == w_first && (ps & 63)) || (w == w_last && (pe & 63)))
				atomicOr(&positions[w], (unsigned long long)acc);
			else
				positions[w] = acc;
		}
	}
}

// ---- host orchestration ------------------------------------------------------------------------------------------------
// mphf::configureLevels, bbhash.h:591-614: IEEE double + libm pow on the HOST (never recomputed on the device: a 1-ulp
// difference in pow could move a level boundary by 64 bits).  gamma = 2 (blight.h:60).
uint64_t configure_levels(uint64_t nelem, uint64_t sizes[kLevels])
{
	const double gamma = 2.0;
	const double proba_collision = 1.0 - pow(((gamma * (double)nelem - 1) / (gamma * (double)nelem)), (double)(nelem - 1));
	uint64_t total = 0;
	const size_t hash_domain = (size_t)(ceil((double)nelem * gamma));
	for (int ii = 0; ii < kLevels; ii++) {
		sizes[ii] = (((uint64_t)((double)hash_domain * pow(proba_collision, (double)ii)) + 63) / 64) * 64;
		if (sizes[ii] == 0)
			sizes[ii] = 64;
		total += sizes[ii];
	}
	return total;
}

struct Scratch {
	DevBuf buf;
	hipStream_t s;
	void* get(size_t bytes)
	{
		if (bytes > buf.bytes)
			BLMX_HIP(buf.alloc(bytes, false, s));
		return buf.p;
	}
};

// 64-bit exclusive prefix sums (scan_kernels.h); `in` is a functor or a pointer
template <typename In> void exclusive_scan_u64(Scratch& sc, In in, uint64_t* out, uint64_t n, hipStream_t s)
{
	uint64_t* d_sums = static_cast<uint64_t*>(sc.get(scan_sums_words(n) * 8));
	exclusive_scan_of<uint64_t>(in, out, n, d_sums, s);
}
inline void exclusive_scan_u64(Scratch& sc, const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t s)
{
	exclusive_scan_u64(sc, ScanPlain<uint64_t>{in}, out, n, s);
}
inline void exclusive_scan_u64(Scratch& sc, uint64_t* in, uint64_t* out, uint64_t n, hipStream_t s)
{
	exclusive_scan_u64(sc, ScanPlain<uint64_t>{in}, out, n, s);
}

// ---- the build's shuffle: a stable LSD radix sort of 32-bit keys that also returns the order (r04; was rocprim::radix_sort_pairs) ----
// Super-k-mers go to their buckets in INPUT order (blight.cpp:234-277 appends them to the bucket files as they come), so the sort by
// minimizer has to be stable.  The key width is known -- 2m - 1 bits, 17 at m = 9 -- so it is ceil(bits / 9) passes of at most 9 bits:
// per pass a histogram per tile of 8192 keys (k_rs_hist), an exclusive scan of the (digit, tile) matrix (k_scan_*: three small kernels)
// and a stable scatter (k_rs_scatter).  Stability inside a tile without atomics: every wave owns a contiguous segment of the tile and
// walks it 64 keys at a time; the lanes of a round that hold the same digit find each other with one ballot per digit bit, the lowest
// of them adds the group's size to the wave's counter of that digit and hands the old value to the others (rank = old + peers below
// me); waves are then ordered by a prefix over the four counters.  The tile is staged in LDS in digit order and copied out in runs.
constexpr int kRsThreads = 256;
constexpr int kRsMaxBits = 9;
template <typename K> struct RsShape {  // keys per thread and tile: 8192 32-bit keys or 4096 64-bit ones (the tile is staged in LDS)
	static constexpr int per = sizeof(K) == 4 ? 32 : 16;
	static constexpr int tile = kRsThreads * per;
};

template <int BITS, typename K>
__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const K* __restrict__ keys, uint64_t n, int shift, uint32_t nblk, uint32_t* __restrict__ hist)
{
	constexpr int NB = 1 << BITS, PER = RsShape<K>::per, TILE = RsShape<K>::tile;
	__shared__ uint32_t s_h[NB];
	for (int d = threadIdx.x; d < NB; d += kRsThreads)
		s_h[d] = 0;
	__syncthreads();
	const uint64_t base = (uint64_t)blockIdx.x * TILE;
#pragma unroll 4
	for (int e = 0; e < PER; e++) {
		const uint64_t i = base + (uint64_t)e * kRsThreads + threadIdx.x;
		if (i < n)
			atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & (NB - 1)], 1u);
	}
	__syncthreads();
	for (int d = threadIdx.x; d < NB; d += kRsThreads)
		hist[(size_t)d * nblk + blockIdx.x] = s_h[d];  // digit-major: the scan of the matrix gives every (digit, tile) its first output slot
}

// One tile, one digit: the threads hold the tile's keys and values (wave w the segment [w 64 PER, (w + 1) 64 PER), round e its keys
// e 64 + lane) and leave them in LDS in digit order, stable.  s_start (may be null) receives the tile-local start of every digit.
template <int BITS, typename K, int PER>
__device__ __forceinline__ void rs_tile_sort(const K (&key)[PER], const uint32_t (&val)[PER], uint32_t nloc, int shift, uint32_t dmask, K* __restrict__ s_key,
                                             uint32_t* __restrict__ s_val, uint32_t (*__restrict__ s_wcnt)[1 << BITS], uint32_t* __restrict__ s_w,
                                             uint32_t* __restrict__ s_start)
{
	constexpr int NB = 1 << BITS, WAVES = kRsThreads / 64, DPT = (NB + kRsThreads - 1) / kRsThreads;  // digits per thread
	const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
	for (int d = t; d < NB; d += kRsThreads)
#pragma unroll
		for (int w = 0; w < WAVES; w++)
			s_wcnt[w][d] = 0;
	__syncthreads();
	uint16_t rank[PER];
	const uint64_t below = (1ull << lane) - 1ull;
	const uint32_t seg = wv * (64 * PER);
#pragma unroll
	for (int e = 0; e < PER; e++) {
		const bool valid = seg + (uint32_t)e * 64 + lane < nloc;
		const uint32_t d = (uint32_t)(key[e] >> shift) & dmask;
		uint64_t peers = __ballot(valid);
#pragma unroll
		for (int b = 0; b < BITS; b++) {
			const bool bit = (d >> b) & 1u;
			const uint64_t bal = __ballot(bit);
			peers &= bit ? bal : ~bal;
		}
		uint32_t old = 0;
		const int leader = valid ? __ffsll((unsigned long long)peers) - 1 : (int)lane;
		if (valid && (int)lane == leader) {
			old = s_wcnt[wv][d];
			s_wcnt[wv][d] = old + (uint32_t)__popcll(peers);
		}
		old = (uint32_t)__shfl((int)old, leader);
		rank[e] = (uint16_t)(old + (uint32_t)__popcll(peers & below));
	}
	__syncthreads();
	// per digit: tile-local start (exclusive scan over the digits of the tile's counts), the waves' segments behind one another
	{
		uint32_t c[DPT][WAVES], tot[DPT], mine = 0;
#pragma unroll
		for (int j = 0; j < DPT; j++) {
			const int d = (int)t * DPT + j;
			tot[j] = 0;
#pragma unroll
			for (int w = 0; w < WAVES; w++) {
				c[j][w] = d < NB ? s_wcnt[w][d] : 0u;
				tot[j] += c[j][w];
			}
			mine += tot[j];
		}
		uint32_t v = mine;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t u = __shfl_up(v, o);
			if ((int)lane >= o)
				v += u;
		}
		if (lane == 63)
			s_w[wv] = v;
		__syncthreads();
		uint32_t at = v - mine;
		for (uint32_t w = 0; w < wv; w++)
			at += s_w[w];
#pragma unroll
		for (int j = 0; j < DPT; j++) {
			const int d = (int)t * DPT + j;
			if (d < NB) {
				if (s_start)
					s_start[d] = at;
				uint32_t a = at;
#pragma unroll
				for (int w = 0; w < WAVES; w++) {
					s_wcnt[w][d] = a;
					a += c[j][w];
				}
			}
			at += tot[j];
		}
	}
	__syncthreads();
#pragma unroll
	for (int e = 0; e < PER; e++) {
		if (seg + (uint32_t)e * 64 + lane < nloc) {
			const uint32_t lp = s_wcnt[wv][(uint32_t)(key[e] >> shift) & dmask] + rank[e];
			s_key[lp] = key[e];
			s_val[lp] = val[e];
		}
	}
	__syncthreads();
}

template <int BITS, typename K>
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in /* null: the index itself */,
                                                         uint64_t n, int shift, const uint32_t* __restrict__ offs, uint32_t nblk, K* __restrict__ keys_out,
                                                         uint32_t* __restrict__ vals_out /* null: keys only */)
{
	constexpr int NB = 1 << BITS, WAVES = kRsThreads / 64;
	constexpr int PER = RsShape<K>::per, TILE = RsShape<K>::tile;
	__shared__ K s_key[TILE];
	__shared__ uint32_t s_val[TILE];
	__shared__ uint32_t s_wcnt[WAVES][NB];  // keys of a digit in a wave's segment, then the tile-local slot of the segment's first one
	__shared__ uint32_t s_gb[NB];          // tile-local start of the digit, then: first output slot of (digit, this tile) minus that start (mod 2^32)
	__shared__ uint32_t s_w[WAVES];
	const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
	const uint64_t base = (uint64_t)blockIdx.x * TILE;
	const uint32_t count = (uint32_t)min((uint64_t)TILE, n - base);
	K key[PER];
	uint32_t val[PER];
#pragma unroll
	for (int e = 0; e < PER; e++) {
		const uint32_t l = wv * (64 * PER) + (uint32_t)e * 64 + lane;
		key[e] = l < count ? keys_in[base + l] : (K)0;
		val[e] = l < count ? (vals_in ? vals_in[base + l] : (uint32_t)(base + l)) : 0u;
	}
	rs_tile_sort<BITS, K, PER>(key, val, count, shift, (uint32_t)NB - 1u, s_key, s_val, s_wcnt, s_w, s_gb);
	for (int d = t; d < NB; d += kRsThreads)
		s_gb[d] = offs[(size_t)d * nblk + blockIdx.x] - s_gb[d];
	__syncthreads();
	for (uint32_t j = t; j < count; j += kRsThreads) {
		const K kx = s_key[j];
		const uint32_t pos = s_gb[(uint32_t)(kx >> shift) & (NB - 1)] + j;
		keys_out[pos] = kx;
		if (vals_out)
			vals_out[pos] = s_val[j];
	}
}

// up to one tile of keys: every pass inside ONE workgroup, the tile going back and forth between registers and LDS (a fallback table of a
// few hundred entries took 8 passes x 5 launches through the kernels above)
template <typename K>
__global__ __launch_bounds__(kRsThreads) void k_rs_small(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t n, unsigned bits,
                                                       K* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
	constexpr int BITS = kRsMaxBits, NB = 1 << BITS, WAVES = kRsThreads / 64;
	constexpr int PER = RsShape<K>::per, TILE = RsShape<K>::tile;
	__shared__ K s_key[TILE];
	__shared__ uint32_t s_val[TILE];
	__shared__ uint32_t s_wcnt[WAVES][NB];
	__shared__ uint32_t s_w[WAVES];
	const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
	K key[PER];
	uint32_t val[PER];
#pragma unroll
	for (int e = 0; e < PER; e++) {
		const uint32_t l = wv * (64 * PER) + (uint32_t)e * 64 + lane;
		key[e] = l < n ? keys_in[l] : (K)0;
		val[e] = l < n ? (vals_in ? vals_in[l] : l) : 0u;
	}
	for (unsigned shift = 0; shift < bits; shift += BITS) {
		const unsigned pb = min((unsigned)BITS, bits - shift);
		rs_tile_sort<BITS, K, PER>(key, val, n, (int)shift, (1u << pb) - 1u, s_key, s_val, s_wcnt, s_w, nullptr);
#pragma unroll
		for (int e = 0; e < PER; e++) {
			const uint32_t l = wv * (64 * PER) + (uint32_t)e * 64 + lane;
			if (l < n) {
				key[e] = s_key[l];
				val[e] = s_val[l];
			}
		}
		__syncthreads();
	}
#pragma unroll
	for (int e = 0; e < PER; e++) {
		const uint32_t l = wv * (64 * PER) + (uint32_t)e * 64 + lane;
		if (l < n) {
			keys_out[l] = key[e];
			if (vals_out)
				vals_out[l] = val[e];
		}
	}
}

// keys[0, n) (n < 2^32, `bits` significant bits) -> keys_out sorted by those bits, STABLE; vals_out[i] = vals_in[j] (vals_in null: j
// itself) for the input index j of the i-th key; vals_out null: keys only
template <typename K>
void stable_sort_pairs(Scratch& sc, const K* keys, K* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint64_t n, unsigned bits, hipStream_t s)
{
	if (!n)
		return;
	if (bits == 0)
		bits = 1;
	constexpr int TILE = RsShape<K>::tile;
	if (n <= (uint64_t)TILE) {
		hipLaunchKernelGGL((k_rs_small<K>), dim3(1), dim3(kRsThreads), 0, s, keys, vals_in, (uint32_t)n, bits, keys_out, vals_out);
		BLMX_HIP(hipGetLastError());
		return;
	}
	const int npass = (int)((bits + kRsMaxBits - 1) / kRsMaxBits);
	const uint32_t nblk = (uint32_t)((n + TILE - 1) / TILE);
	const size_t hist_words = ((size_t)nblk << kRsMaxBits);
	const size_t sums_words = scan_sums_words(hist_words);
	// scratch: [histogram | scan sums | (more than one pass:) a second array of keys and one of values]
	const size_t kw = sizeof(K) / 4;
	const size_t pair_words = npass > 1 ? (kw + 1) * ((size_t)n + 2) : 0;
	uint32_t* d_hist = static_cast<uint32_t*>(sc.get((hist_words + sums_words + pair_words + 2) * 4));
	uint32_t* d_sums = d_hist + hist_words;
	K* tk = reinterpret_cast<K*>(d_sums + ((sums_words + 1) & ~(size_t)1));
	uint32_t* tv = reinterpret_cast<uint32_t*>(tk + (n + 1));
	// the passes alternate between the caller's arrays and the temporaries so that the LAST one writes the caller's
	const K* kin = keys;
	const uint32_t* vin = vals_in;
	for (int p = 0; p < npass; p++) {
		const int shift = p * kRsMaxBits;
		const int pb = (int)std::min<unsigned>(kRsMaxBits, bits - (unsigned)shift);
		const bool to_caller = ((npass - 1 - p) % 2) == 0;
		K* kout = to_caller ? keys_out : tk;
		uint32_t* vout = vals_out ? (to_caller ? vals_out : tv) : nullptr;
#define BLMX_RS(B)                                                                                                                          \
	case B:                                                                                                                                 \
		hipLaunchKernelGGL((k_rs_hist<B, K>), dim3(nblk), dim3(kRsThreads), 0, s, kin, n, shift, nblk, d_hist);                            \
		exclusive_scan_u32(d_hist, d_hist, (uint64_t)nblk << B, d_sums, s);                                                                 \
		hipLaunchKernelGGL((k_rs_scatter<B, K>), dim3(nblk), dim3(kRsThreads), 0, s, kin, vin, n, shift, d_hist, nblk, kout, vout);        \
		break;
		switch (pb) {
			BLMX_RS(1)
			BLMX_RS(2)
			BLMX_RS(3)
			BLMX_RS(4)
			BLMX_RS(5)
			BLMX_RS(6)
			BLMX_RS(7)
			BLMX_RS(8)
			BLMX_RS(9)
		}
#undef BLMX_RS
		BLMX_HIP(hipGetLastError());
		kin = kout;
		vin = vout;
	}
}
// 32-bit keys -> sorted keys and the order (the build's shuffle)
void stable_sort_by_key_u32(Scratch& sc, const uint32_t* keys, uint32_t* keys_out, uint32_t* order_out, uint64_t n, unsigned bits, hipStream_t s)
{
	stable_sort_pairs<uint32_t>(sc, keys, keys_out, nullptr, order_out, n, bits, s);
}

template <typename T> T read_scalar(const T* d, hipStream_t s)
{
	T h;
	BLMX_HIP(hipMemcpyAsync(&h, d, sizeof(T), hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	return h;
}

void check_build_err(const int* d_err, hipStream_t s)
{
	const int e = read_scalar(d_err, s);
	if (e & kErrDna)
		throw StatusError{BLMX_EDNA, "Invalid char in DNA"};
	if (e & kErrShort)
		throw StatusError{BLMX_ESHORT, "a sequence is shorter than k"};
	if (e & kErrLimit)
		throw StatusError{BLMX_ELIMIT, "a minimizer bucket holds 2^32 or more nucleotides"};
}

// temporaries of a build recycle each other's device blocks
struct BuildPoolScope {
	BuildPool own;
	Index* defer_to;  // while the filter's background pass runs, the cached blocks are parked in the index instead of being freed here
	explicit BuildPoolScope(BuildPool* shared = nullptr, Index* defer = nullptr) : defer_to(defer)
	{
		const char* e = getenv("BLIGHT_BUILD_POOL");
		BuildPool::current() = (e && atoi(e) == 0) ? nullptr : (shared ? shared : &own);
	}
	~BuildPoolScope()
	{
		BuildPool* used = BuildPool::current();
		BuildPool::current() = nullptr;
		if (defer_to && used)
			defer_to->st.build_scratch_peak_bytes = std::max<uint64_t>(defer_to->st.build_scratch_peak_bytes, used->peak_held);
		if (defer_to && defer_to->filter_state.load() == 0) {
			defer_to->deferred_blocks.insert(defer_to->deferred_blocks.end(), own.free_blocks.begin(), own.free_blocks.end());
			own.free_blocks.clear();
		}
	}
};

// BLIGHT_BUILD_TRACE=1: wall time of every phase on stderr (each mark synchronises the stream); =2: host time between the marks
// without synchronising (where does the HOST stall?)
struct PhaseMarks {
	hipStream_t s;
	bool on, host_only;
	std::chrono::steady_clock::time_point t_last;
	explicit PhaseMarks(hipStream_t st)
	    : s(st), on(getenv("BLIGHT_BUILD_TRACE") && atoi(getenv("BLIGHT_BUILD_TRACE")) != 0),
	      host_only(getenv("BLIGHT_BUILD_TRACE") && atoi(getenv("BLIGHT_BUILD_TRACE")) == 2), t_last(std::chrono::steady_clock::now())
	{
	}
	void operator()(const char* what)
	{
		if (!on)
			return;
		if (!host_only)
			(void)hipStreamSynchronize(s);
		const auto now = std::chrono::steady_clock::now();
		const BuildPool* pool = BuildPool::current();
		size_t parked = 0;
		if (pool)
			for (const BuildPool::Block& b : pool->free_blocks)
				parked += b.bytes;
		fprintf(stderr, "[blmx build] %-28s %8.2f ms   scratch held %6.2f GB (parked %6.2f GB)\n", what,
		        std::chrono::duration<double, std::milli>(now - t_last).count(), pool ? pool->held / 1e9 : 0.0, parked / 1e9);
		t_last = now;
	}
};

// K1-K3 of an input: its 2-bit packed text and its super-k-mers in input order
struct Chopped {
	DevBuf src, skpos, sklast, skmini;  // packed text; first nt of every super-k-mer; start of its last k-mer; its minimizer
	uint64_t nsk = 0;
};

// what a shard knows about the whole index
struct ShardCtx {
	uint64_t id_base = 0, global_N = 0, global_nsk = 0;
	uint32_t glo = 0, ghi = 0;
	// halo: the first halo_nt nucleotides the SINGLE index holds behind this shard's last bucket (the head of the next owners' buckets),
	// 2-bit packed in device memory; written behind the shard's own sequence so that windows that run past its last bucket read what
	// the reference reads there (blight.cpp:729-739 scans on into the next bucket).  nullptr = none (last rank, or shard_halo_set later)
	const uint64_t* d_halo = nullptr;
	uint64_t halo_nt = 0;
	bool halo_later = false;  // construction from slices: the next owners' heads only exist once every rank has finished (shard_halo_set)
};

// nucleotides of halo a shard keeps: a window starts at most k nucleotides before the end of a bucket and covers 2^b positions + k - 1
inline uint64_t halo_length(const Index& ix) { return (1ull << ix.prm.bits_saved) + (uint64_t)ix.prm.k; }

// packed != nullptr: the text is already 2-bit packed (the file loader's output); it is taken over and K1's encoding is skipped
template <int KW>
void chop_input(Index& ix, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len, Chopped& out, PhaseMarks& mark,
                DevBuf* packed = nullptr)
{
	hipStream_t s = ix.stream;
	const int k = (int)ix.prm.k, m = (int)ix.prm.m;
	Scratch sc;
	sc.s = s;
	DevBuf& d_src = out.src;
	DevBuf& d_skpos = out.skpos;
	DevBuf& d_sklast = out.sklast;
	DevBuf& d_skmini = out.skmini;
	DevBuf d_err;
	BLMX_HIP(d_err.alloc(sizeof(int), true, s));

	// ---- K1: packed source + unitig-start bitmap
	const uint64_t npos = total_len + 1;                      // positions 0 .. total_len
	const uint64_t nbw = ((npos + kBlock - 1) / kBlock) * (kBlock / 64);  // bitmap words written by K2
	const uint64_t src_words = (total_len + 31) / 32;
	DevBuf d_ustart, d_Hb, d_Eb;
	if (packed) {
		if (packed->bytes < (src_words + 4) * 8)
			throw StatusError{BLMX_EINVAL, "internal: packed text too small"};
		d_src.release();
		d_src.swap(*packed);
	} else {
		BLMX_HIP(d_src.alloc((src_words + 4) * 8, true, s));
	}
	BLMX_HIP(d_ustart.alloc((nbw + 4) * 8, true, s));
	BLMX_HIP(d_Hb.alloc((nbw + 1) * 8, true, s));
	BLMX_HIP(d_Eb.alloc((nbw + 1) * 8, true, s));
	if (nseq) {
		hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(nseq)), dim3(kBlock), 0, s, d_offs, nseq, k, d_ustart.as<unsigned long long>(),
		                   d_err.as<int>());
		if (packed)
			;  // already encoded
		else if ((reinterpret_cast<uintptr_t>(d_ascii) & 15) == 0)
			hipLaunchKernelGGL((k_encode_pack<true>), dim3(grid_for(src_words)), dim3(kBlock), 0, s, d_ascii, total_len, src_words,
			                   d_src.as<uint64_t>(), d_err.as<int>());
		else
			hipLaunchKernelGGL((k_encode_pack<false>), dim3(grid_for(src_words)), dim3(kBlock), 0, s, d_ascii, total_len, src_words,
			                   d_src.as<uint64_t>(), d_err.as<int>());
		BLMX_HIP(hipGetLastError());
		check_build_err(d_err.as<int>(), s);
	}

	mark("K1 encode+starts");
	// ---- K2: super-k-mer head / end bitmaps
	uint64_t nsk = 0;
	DevBuf d_Hpre, d_Epre;
	BLMX_HIP(d_Hpre.alloc((nbw + 1) * 8, false, s));
	BLMX_HIP(d_Epre.alloc((nbw + 1) * 8, false, s));
	if (nseq) {
		const uint64_t k2_blocks = (npos + kBlock - 1) / kBlock;
		const unsigned k2_gx = (unsigned)std::min<uint64_t>(k2_blocks, 1u << 22), k2_gy = (unsigned)((k2_blocks + k2_gx - 1) / k2_gx);
		hipLaunchKernelGGL((k_superkmer_flags<KW>), dim3(k2_gx, k2_gy), dim3(kBlock), 0, s, d_src.as<uint64_t>(),
		                   d_ustart.as<unsigned long long>(), total_len, k, m, d_Hb.as<unsigned long long>(),
		                   d_Eb.as<unsigned long long>());
		BLMX_HIP(hipGetLastError());
		const ScanPopc64<uint64_t> hin{d_Hb.as<unsigned long long>()}, ein{d_Eb.as<unsigned long long>()};
		exclusive_scan_u64(sc, hin, d_Hpre.as<uint64_t>(), nbw + 1, s);
		exclusive_scan_u64(sc, ein, d_Epre.as<uint64_t>(), nbw + 1, s);
		nsk = read_scalar(d_Hpre.as<uint64_t>() + nbw, s);
		const uint64_t nend = read_scalar(d_Epre.as<uint64_t>() + nbw, s);
		if (nend != nsk)
			throw StatusError{BLMX_EHIP, "internal: super-k-mer head/end count mismatch"};
		if (nsk >= 0xffffffffull)
			throw StatusError{BLMX_ELIMIT, "more than 2^32 super-k-mers"};
	}
	d_ustart.release();

	mark("K2 flags+scans");
	// ---- K3: compact super-k-mers
	BLMX_HIP(d_skpos.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(d_sklast.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(d_skmini.alloc((nsk + 1) * 4, false, s));
	if (nsk) {
		hipLaunchKernelGGL(k_collect, dim3(grid_for(nbw)), dim3(kBlock), 0, s, d_Hb.as<unsigned long long>(), d_Eb.as<unsigned long long>(),
		                   d_Hpre.as<uint64_t>(), d_Epre.as<uint64_t>(), nbw, d_skpos.as<uint64_t>(), d_sklast.as<uint64_t>());
		hipLaunchKernelGGL((k_sk_minis<KW>), dim3(grid_for(nsk)), dim3(kBlock), 0, s, d_src.as<uint64_t>(), d_skpos.as<uint64_t>(), nsk, k, m,
		                   d_skmini.as<uint32_t>());
		BLMX_HIP(hipGetLastError());
	}
	d_Hb.release();
	d_Eb.release();
	d_Hpre.release();
	d_Epre.release();

	mark("K3 collect");
	out.nsk = nsk;
}

template <int KW>
void build_tail(Index& ix, Chopped& chopped, const ShardCtx& shard, PhaseMarks& mark, std::chrono::steady_clock::time_point t_begin);

// the per-group key histogram of a list of super-k-mers -> host
std::vector<uint64_t> group_histogram(Index& ix, const Chopped& c)
{
	hipStream_t s = ix.stream;
	DevBuf d_ghist;
	BLMX_HIP(d_ghist.alloc((size_t)ix.ngroups * 8, true, s));
	if (c.nsk)
		hipLaunchKernelGGL(k_group_hist, dim3(grid_for(c.nsk)), dim3(kBlock), 0, s, c.skmini.as<uint32_t>(), c.skpos.as<uint64_t>(),
		                   c.sklast.as<uint64_t>(), c.nsk, ix.gshift, d_ghist.as<unsigned long long>());
	std::vector<uint64_t> h((size_t)ix.ngroups);
	BLMX_HIP(hipMemcpyAsync(h.data(), d_ghist.p, (size_t)ix.ngroups * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	return h;
}

// key-balanced shard boundaries from the GLOBAL histogram (every rank derives the same ones); returns the ID prefix of each group
std::vector<uint64_t> derive_shard_bounds(Index& ix, const std::vector<uint64_t>& hist)
{
	const uint32_t world = ix.shard_world, ngroups = ix.ngroups;
	std::vector<uint64_t> pre((size_t)ngroups + 1, 0);
	for (uint32_t g = 0; g < ngroups; g++)
		pre[g + 1] = pre[g] + hist[g];
	ix.shard_bounds.assign((size_t)world + 1, 0);
	ix.shard_bounds[world] = ngroups;
	for (uint32_t r = 1; r < world; r++) {
		const uint64_t target = (uint64_t)(((unsigned __int128)pre[ngroups] * r) / world);
		ix.shard_bounds[r] = (uint32_t)(std::lower_bound(pre.begin(), pre.begin() + ngroups, target) - pre.begin());
	}
	return pre;
}

template <int KW>
void build_impl(Index& ix, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len, DevBuf* packed = nullptr)
{
	const auto t_begin = std::chrono::steady_clock::now();
	ix.st.build_scratch_peak_bytes = 0;
	struct StartThread {  // declared before the pool scope: runs after it has parked its blocks in the index
		Index& ix;
		~StartThread() { ix.start_filter_thread(); }
	} start_thread{ix};
	BuildPoolScope pool_scope(nullptr, &ix);  // destroyed after every temporary
	hipStream_t s = ix.stream;
	PhaseMarks mark(s);
	Scratch sc;
	sc.s = s;
	const uint32_t ngroups = ix.ngroups;
	const int gshift = ix.gshift;
	Chopped chopped;
	chop_input<KW>(ix, d_ascii, d_offs, nseq, total_len, chopped, mark, packed);
	DevBuf& d_skpos = chopped.skpos;
	DevBuf& d_sklast = chopped.sklast;
	DevBuf& d_skmini = chopped.skmini;
	uint64_t nsk = chopped.nsk;
	// ---- sharding, whole-input mode: every rank sees the whole input, derives the same key-balanced group boundaries
	// from the global per-group key histogram, and keeps only the super-k-mers of its own groups.  IDs stay global (id_base).
	// (Ranks that only hold a slice of the input exchange super-k-mers instead: ShardExchange below.)
	ShardCtx shard;
	shard.global_nsk = nsk;
	shard.ghi = ngroups;
	const uint32_t world = ix.shard_world, rank = ix.shard_rank;
	ix.shard_bounds.assign((size_t)world + 1, 0);
	ix.shard_bounds[world] = ngroups;
	DevBuf d_halo;  // lives until build_tail has copied it behind the shard's sequence
	if (world > 1) {
		const std::vector<uint64_t> pre = derive_shard_bounds(ix, group_histogram(ix, chopped));
		shard.global_N = pre[ngroups];
		const uint32_t glo = shard.glo = ix.shard_bounds[rank];
		const uint32_t ghi = shard.ghi = ix.shard_bounds[rank + 1];
		shard.id_base = pre[glo];
		// the super-k-mers of the groups [g0, g1), in input order
		DevBuf d_flags, d_cnt;
		BLMX_HIP(d_flags.alloc(nsk + 1, true, s));  // (flag[nsk] stays 0: the scan's last output is the number kept)
		BLMX_HIP(d_cnt.alloc(8, true, s));
		auto select_groups = [&](uint32_t g0, uint32_t g1, DevBuf& d_pos2, DevBuf& d_last2, DevBuf& d_mini2) -> uint64_t {
			BLMX_HIP(d_pos2.alloc((nsk + 1) * 8, false, s));
			BLMX_HIP(d_last2.alloc((nsk + 1) * 8, false, s));
			BLMX_HIP(d_mini2.alloc((nsk + 1) * 4, false, s));
			if (!nsk)
				return 0;
			hipLaunchKernelGGL(k_shard_flags, dim3(grid_for(nsk)), dim3(kBlock), 0, s, d_skmini.as<uint32_t>(), nsk, gshift, g0, g1, d_flags.as<uint8_t>());
			BLMX_HIP(hipGetLastError());
			// stream compaction in input order: exclusive scan of the flags, then every kept super-k-mer moves to its slot
			uint32_t* d_slot = static_cast<uint32_t*>(sc.get((nsk + 1 + scan_sums_words(nsk + 1)) * 4));
			exclusive_scan_of<uint32_t>(ScanPlain<uint8_t>{d_flags.as<uint8_t>()}, d_slot, nsk + 1, d_slot + nsk + 1, s);  // (flag[nsk] is 0)
			hipLaunchKernelGGL(k_select3, dim3(grid_for(nsk)), dim3(kBlock), 0, s, d_flags.as<uint8_t>(), d_slot, nsk, d_skpos.as<uint64_t>(),
			                   d_sklast.as<uint64_t>(), d_skmini.as<uint32_t>(), d_pos2.as<uint64_t>(), d_last2.as<uint64_t>(), d_mini2.as<uint32_t>(),
			                   d_cnt.as<uint64_t>());
			BLMX_HIP(hipGetLastError());
			return read_scalar(d_cnt.as<uint64_t>(), s);
		};
		// ---- halo: the head of what follows this shard in the single index = the first buckets of the groups behind ghi, laid out like
		// build_tail lays out its own (stable minimizer sort, back to back).  The groups [ghi, gh2) with gh2 = the first boundary
		// behind which at least halo_length keys (hence nucleotides) have been seen hold all of it.
		const uint64_t H = halo_length(ix);
		uint32_t gh2 = ghi;
		while (gh2 < ngroups && pre[gh2] - pre[ghi] < H)
			gh2++;
		if (gh2 > ghi && pre[gh2] > pre[ghi]) {
			DevBuf d_hpos, d_hlast, d_hmini, d_hsmini, d_horder, d_hlennt, d_hlenk, d_hstart;
			const uint64_t hn = select_groups(ghi, gh2, d_hpos, d_hlast, d_hmini);
			BLMX_HIP(d_halo.alloc(((H + 31) / 32 + 2) * 8, true, s));
			BLMX_HIP(d_hsmini.alloc((hn + 1) * 4, false, s));
			BLMX_HIP(d_horder.alloc((hn + 1) * 4, false, s));
			BLMX_HIP(d_hlennt.alloc((hn + 1) * 8, false, s));
			BLMX_HIP(d_hlenk.alloc((hn + 1) * 8, false, s));
			BLMX_HIP(d_hstart.alloc((hn + 1) * 8, false, s));
			if (hn) {
				stable_sort_by_key_u32(sc, d_hmini.as<uint32_t>(), d_hsmini.as<uint32_t>(), d_horder.as<uint32_t>(), hn, (unsigned)(2 * ix.prm.m - 1), s);
				hipLaunchKernelGGL(k_sorted_lengths, dim3(grid_for(hn + 1)), dim3(kBlock), 0, s, d_horder.as<uint32_t>(), d_hpos.as<uint64_t>(),
				                   d_hlast.as<uint64_t>(), hn, (int)ix.prm.k, d_hlennt.as<uint64_t>(), d_hlenk.as<uint64_t>());
				BLMX_HIP(hipGetLastError());
				exclusive_scan_u64(sc, d_hlennt.as<uint64_t>(), d_hstart.as<uint64_t>(), hn + 1, s);
				shard.halo_nt = std::min<uint64_t>(H, read_scalar(d_hstart.as<uint64_t>() + hn, s));
				hipLaunchKernelGGL(k_scatter_pack, dim3(grid_for(hn)), dim3(kBlock), 0, s, chopped.src.as<uint64_t>(), d_horder.as<uint32_t>(), d_hpos.as<uint64_t>(),
				                   d_hstart.as<uint64_t>(), (const uint64_t*)nullptr, hn, d_halo.as<unsigned long long>(), H);
				BLMX_HIP(hipGetLastError());
				BLMX_HIP(hipStreamSynchronize(s));  // the temporaries of this block go back to the pool
				shard.d_halo = d_halo.as<uint64_t>();
			}
		}
		DevBuf d_pos2, d_last2, d_mini2;
		const uint64_t kept = select_groups(glo, ghi, d_pos2, d_last2, d_mini2);
		d_skpos.swap(d_pos2);
		d_sklast.swap(d_last2);
		d_skmini.swap(d_mini2);
		nsk = kept;
	}

	mark("shard filter");
	chopped.nsk = nsk;
	build_tail<KW>(ix, chopped, shard, mark, t_begin);
}

// K4 .. K9 and the public counters: from the super-k-mers this index keeps (in input order) to the finished arrays
template <int KW>
void build_tail(Index& ix, Chopped& chopped, const ShardCtx& shard, PhaseMarks& mark, std::chrono::steady_clock::time_point t_begin)
{
	hipStream_t s = ix.stream;
	const int k = (int)ix.prm.k, m = (int)ix.prm.m, b = (int)ix.prm.bits_saved;
	const uint32_t nbuckets = ix.nbuckets, ngroups = ix.ngroups;
	const int gshift = ix.gshift;
	const uint32_t world = ix.shard_world, rank = ix.shard_rank;
	const uint64_t id_base = shard.id_base, global_N = shard.global_N, global_nsk = shard.global_nsk;
	const uint32_t glo = shard.glo, ghi = shard.ghi;
	const bool trace = mark.on;
	(void)trace;
	Scratch sc;
	sc.s = s;
	DevBuf& d_src = chopped.src;
	DevBuf& d_skpos = chopped.skpos;
	DevBuf& d_sklast = chopped.sklast;
	DevBuf& d_skmini = chopped.skmini;
	const uint64_t nsk = chopped.nsk;
	DevBuf d_err;
	BLMX_HIP(d_err.alloc(sizeof(int), true, s));
	// ---- K4: stable sort by minimizer, offsets, bucket + group tables
	DevBuf d_smini, d_order, d_lennt, d_lenk, d_ntstart, d_keystart;
	BLMX_HIP(d_smini.alloc((nsk + 1) * 4, false, s));
	BLMX_HIP(d_order.alloc((nsk + 1) * 4, false, s));
	stable_sort_by_key_u32(sc, d_skmini.as<uint32_t>(), d_smini.as<uint32_t>(), d_order.as<uint32_t>(), nsk, (unsigned)(2 * m - 1), s);
	sc.buf.release();  // (the sort's second pair of arrays: 8 bytes per super-k-mer; the scans below ask for a few KB)
	d_skmini.release();
	BLMX_HIP(d_lennt.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(d_lenk.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(d_ntstart.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(d_keystart.alloc((nsk + 1) * 8, false, s));
	hipLaunchKernelGGL(k_sorted_lengths, dim3(grid_for(nsk + 1)), dim3(kBlock), 0, s, d_order.as<uint32_t>(), d_skpos.as<uint64_t>(),
	                   d_sklast.as<uint64_t>(), nsk, k, d_lennt.as<uint64_t>(), d_lenk.as<uint64_t>());
	BLMX_HIP(hipGetLastError());
	exclusive_scan_u64(sc, d_lennt.as<uint64_t>(), d_ntstart.as<uint64_t>(), nsk + 1, s);
	exclusive_scan_u64(sc, d_lenk.as<uint64_t>(), d_keystart.as<uint64_t>(), nsk + 1, s);
	const uint64_t total_nt = read_scalar(d_ntstart.as<uint64_t>() + nsk, s);
	const uint64_t N = read_scalar(d_keystart.as<uint64_t>() + nsk, s);
	d_lennt.release();
	d_lenk.release();
	d_sklast.release();
	// k-mer counts are 64-bit throughout; 32-bit quantities are per MPHF group (key lists, ranks) and per batch

	DevBuf d_bnb, d_bne, d_bkb, d_bke, d_bnt, d_bkeys, d_bstart, d_bkpre;
	const size_t bb = ((size_t)nbuckets + 1) * 8;
	BLMX_HIP(d_bnb.alloc(bb, true, s));
	BLMX_HIP(d_bne.alloc(bb, true, s));
	BLMX_HIP(d_bkb.alloc(bb, true, s));
	BLMX_HIP(d_bke.alloc(bb, true, s));
	BLMX_HIP(d_bnt.alloc(bb, false, s));
	BLMX_HIP(d_bkeys.alloc(bb, false, s));
	BLMX_HIP(d_bstart.alloc(bb, false, s));
	BLMX_HIP(d_bkpre.alloc(bb, false, s));
	if (nsk)
		hipLaunchKernelGGL(k_bucket_runs, dim3(grid_for(nsk)), dim3(kBlock), 0, s, d_smini.as<uint32_t>(), d_ntstart.as<uint64_t>(),
		                   d_keystart.as<uint64_t>(), nsk, d_bnb.as<uint64_t>(), d_bne.as<uint64_t>(), d_bkb.as<uint64_t>(),
		                   d_bke.as<uint64_t>());
	hipLaunchKernelGGL(k_bucket_finish, dim3(grid_for((uint64_t)nbuckets + 1)), dim3(kBlock), 0, s, d_bnb.as<uint64_t>(), d_bne.as<uint64_t>(),
	                   d_bkb.as<uint64_t>(), d_bke.as<uint64_t>(), nbuckets, d_bnt.as<uint64_t>(), d_bkeys.as<uint64_t>(), d_err.as<int>());
	BLMX_HIP(hipGetLastError());
	check_build_err(d_err.as<int>(), s);
	exclusive_scan_u64(sc, d_bnt.as<uint64_t>(), d_bstart.as<uint64_t>(), (uint64_t)nbuckets + 1, s);
	exclusive_scan_u64(sc, d_bkeys.as<uint64_t>(), d_bkpre.as<uint64_t>(), (uint64_t)nbuckets + 1, s);
	d_bnb.release();
	d_bne.release();
	d_bkb.release();
	d_bke.release();
	BLMX_HIP(ix.buckets.alloc((size_t)nbuckets * sizeof(BucketDesc), false, s));
	hipLaunchKernelGGL(k_bucket_table, dim3(grid_for(nbuckets)), dim3(kBlock), 0, s, d_bnt.as<uint64_t>(), d_bstart.as<uint64_t>(), nbuckets,
	                   ix.buckets.as<BucketDesc>());
	DevBuf d_gnk, d_gpre, d_gmax;
	BLMX_HIP(d_gnk.alloc((size_t)ngroups * 8, false, s));
	BLMX_HIP(d_gpre.alloc(((size_t)ngroups + 1) * 8, false, s));
	BLMX_HIP(d_gmax.alloc((size_t)ngroups * 4, false, s));
	hipLaunchKernelGGL(k_group_tables, dim3(grid_for((uint64_t)ngroups * 64)), dim3(kBlock), 0, s, d_bnt.as<uint64_t>(), d_bkpre.as<uint64_t>(),
	                   ngroups, gshift, d_gnk.as<uint64_t>(), d_gpre.as<uint64_t>(), d_gmax.as<uint32_t>());
	BLMX_HIP(hipGetLastError());
	BLMX_HIP(hipMemcpyAsync(d_gpre.as<uint64_t>() + ngroups, &N, 8, hipMemcpyHostToDevice, s));
	std::vector<uint64_t> h_gnk(ngroups), h_gpre(ngroups);
	std::vector<uint32_t> h_gmax(ngroups);
	BLMX_HIP(hipMemcpyAsync(h_gnk.data(), d_gnk.p, (size_t)ngroups * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipMemcpyAsync(h_gpre.data(), d_gpre.p, (size_t)ngroups * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipMemcpyAsync(h_gmax.data(), d_gmax.p, (size_t)ngroups * 4, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	d_bnt.release();
	d_bkeys.release();
	d_bstart.release();
	d_bkpre.release();
	d_gnk.release();
	d_gmax.release();

	mark("K4 sort+tables");
	// ---- host: group descriptors (blight.cpp:292-303) and BBHash level tables (bbhash.h:591-614)
	ix.h_groups.assign(ngroups, GroupDesc{});
	std::vector<uint64_t> h_loff((size_t)ngroups * kLevels, 0), h_line_base((size_t)ngroups + 1, 0);
	uint64_t total_pos = 0, phys_pos = 0, wbase = 0, lbase = 0, largest_mphf = 0;
	uint32_t largest_bucket = 0;
	// physical position fields: the reference's nbits plus `hint` low bits (the sub-window of the key inside its 2^b window), when
	// every group's field still fits 32 bits; the reference's own offsets (total_pos) are kept for the counters and the export
	uint32_t hint = (ix.tune_pos_hint && b <= 8 && (b >= 7 || ix.tune_pos_hint == 2)) ? (uint32_t)position_hint_bits(b) : 0u;
	for (uint32_t g = 0; g < ngroups && hint; g++)
		if ((int)(ceil(log2((double)(uint32_t)(h_gmax[g] + 1))) - (double)b) + (int)hint > 32)
			hint = 0;
	ix.pos_hint = hint;
	// absent-key filter (index_device.h): tune_filter_bits per key.  Automatic: 8 bits per key (2-3 % false positives; more bits were
	// no faster at b = 6) -- but 16 where the position hint shortens the scan: every false positive of the filter sends its whole wave
	// through the reference's 2^b window (C5: 3.83 -> 2.75 ms).  With 8 bits per key the filter lives INSIDE the level-0 lines (64-B
	// fat lines: filter test and level-0 probe in one request) unless tuned off.
	const uint32_t bpk = ix.tune_filter_bits < 0 ? (hint ? 16u : 8u) : (uint32_t)std::min(ix.tune_filter_bits, 32);
	const bool fat = bpk == 8 && ix.tune_fat_lines != 0;
	ix.fat_lines = fat;
	// ... as level-0 RECORDS (32 level bits + filter + the position fields of their keys: 64 B per 16 keys instead of per 48) only when
	// asked for (fat_lines = 2): a level-0 key's position then comes with its rank, but the records of a human-scale index are 10 GB
	// instead of 3.3, a batch touches every line of them, and the slot kernel pays per line: all-indexed batches 4 % faster, 50/50
	// batches 8 % slower, all-absent ones 8 % slower (profiles/r04_exp_level0_layouts_c4.txt)
	const bool rec = fat && ix.tune_fat_lines == 2;
	ix.rec_lines = rec;
	const int layout = rec ? 2 : (fat ? 1 : 0);
	uint64_t ubase = 0;  // 16-B units of the line array (lbase below counts logical lines)
	for (uint32_t g = 0; g < ngroups; g++) {
		GroupDesc& gd = ix.h_groups[g];
		int nb = (int)(ceil(log2((double)(uint32_t)(h_gmax[g] + 1))) - (double)b);
		if (nb < 1)
			nb = 1;
		gd.nbits = (uint32_t)nb + hint;
		gd.posstart = phys_pos;
		phys_pos += (uint64_t)(nb + hint) * h_gnk[g] + 8;
		total_pos += (uint64_t)nb * h_gnk[g] + 8;  // round_eight, blight.cpp:102-104
		gd.prefix = h_gpre[g];
		gd.nkeys = h_gnk[g];
		gd.wbase = wbase;
		gd.line_base = ubase;
		h_line_base[g] = lbase;
		if (h_gnk[g]) {
			const uint64_t total_bits = configure_levels(h_gnk[g], gd.size);
			uint64_t off = 0;
			for (int l = 0; l < kLevels; l++) {
				h_loff[(size_t)g * kLevels + l] = off;
				off += gd.size[l];
			}
			gd.nwords = total_bits / 64;
			if (gd.nwords >= (1ull << 31))  // the query kernel indexes a group's level array with 32-bit word numbers
				throw StatusError{BLMX_ELIMIT, "an MPHF group holds more than 2^37 level bits"};
			wbase += gd.nwords;
			if (fat) {
				// (the level-0 units in cross-paired 128-B blocks: the group's lines start on a multiple of eight 16-B units)
				const uint64_t F = level0_units64(gd.size[0], rec), T = (2 * gd.nwords - (gd.size[0] >> 5) + kLineData32 - 1) / kLineData32;
				ubase = (ubase + kBlockUnits - 1) / kBlockUnits * kBlockUnits;
				gd.line_base = ubase;
				lbase += F + T;
				ubase += level0_block_units(gd.size[0], rec) + T;
			} else {
				const uint64_t L = (2 * gd.nwords + kLineData32 - 1) / kLineData32;
				lbase += L;
				ubase += L;
			}
			if (ubase >= (1ull << 32))
				throw StatusError{BLMX_ELIMIT, "the MPHF lines of one index (shard) hold less than 64 GB"};
		} else {
			// no MPHF for an empty group (blight.cpp:452); size 64 keeps the probe arithmetic defined, never reached
			for (int l = 0; l < kLevels; l++)
				gd.size[l] = 64;
			gd.nwords = 0;
		}
		largest_mphf = std::max(largest_mphf, h_gnk[g]);
		largest_bucket = std::max(largest_bucket, h_gmax[g]);
	}
	h_line_base[ngroups] = lbase;
	const uint64_t total_words = wbase, total_units = ubase;

	DevBuf d_loff, d_line_base, d_bgroups;
	BLMX_HIP(d_loff.alloc(h_loff.size() * 8, false, s));
	BLMX_HIP(d_line_base.alloc(h_line_base.size() * 8, false, s));
	BLMX_HIP(d_bgroups.alloc((size_t)ngroups * sizeof(GroupDesc), false, s));
	BLMX_HIP(hipMemcpyAsync(d_loff.p, h_loff.data(), h_loff.size() * 8, hipMemcpyHostToDevice, s));
	BLMX_HIP(hipMemcpyAsync(d_line_base.p, h_line_base.data(), h_line_base.size() * 8, hipMemcpyHostToDevice, s));
	BLMX_HIP(hipMemcpyAsync(d_bgroups.p, ix.h_groups.data(), (size_t)ngroups * sizeof(GroupDesc), hipMemcpyHostToDevice, s));

	mark("host descriptors");
	// ---- K5: packed bucket sequence.  Padding: the 2^b scan may read NW words past the last k-mer (blight.cpp:738)
	const uint64_t pad_words = ((1ull << std::min(b, 20)) + 64 * (uint64_t)KW + 64) / 32 + 4 + 16;  // (+ 128 B: whole-line reads, index_device.h)
	ix.seq_words = total_nt / 32 + pad_words;
	BLMX_HIP(ix.seq.alloc(ix.seq_words * 8, true, s));
	if (nsk)
		hipLaunchKernelGGL(k_scatter_pack, dim3(grid_for(nsk)), dim3(kBlock), 0, s, d_src.as<uint64_t>(), d_order.as<uint32_t>(),
		                   d_skpos.as<uint64_t>(), d_ntstart.as<uint64_t>(), (const uint64_t*)nullptr, nsk, ix.seq.as<unsigned long long>(), ~0ull);

	// a shard's halo behind its own sequence (inside the padding: halo_length <= 2^b + 63)
	ix.halo_nt = 0;
	ix.halo_done = !shard.halo_later;
	if (shard.d_halo && shard.halo_nt) {
		hipLaunchKernelGGL(k_put_halo, dim3(grid_for(shard.halo_nt / 32 + 2)), dim3(kBlock), 0, s, shard.d_halo, shard.halo_nt, total_nt,
		                   ix.seq.as<unsigned long long>());
		ix.halo_nt = shard.halo_nt;
	}

	mark("K5 scatter_pack");
	// ---- K6 .. K9b in PASSES over ranges of MPHF groups (the reference builds one of 2^s super-buckets at a time to bound its memory,
	// blight.cpp:335-356; log2_superbuckets = s bounds the number of passes here).  Keys, survivor lists and position fields only exist
	// for the groups of the current pass; everything a pass produces (level bits -> lines, fallback entries, positions) goes straight
	// into the finished arrays.  Groups are independent from K6 on, so the arrays do not depend on the number of passes.
	uint32_t npass = 1;
	{
		// automatic: as few passes as keep a pass at or below 2^30 keys (keys + two survivor lists + fields = 20 B per key of the pass)
		const uint32_t cap = 1u << std::min<uint32_t>(ix.prm.log2_superbuckets, 6u);
		const int forced = ix.tune_build_passes;
		uint64_t want = forced > 0 ? (uint64_t)forced : (N + (1ull << 30) - 1) >> 30;
		while (npass < want && npass < (forced > 0 ? 64u : cap) && npass < ngroups)
			npass <<= 1;
	}
	std::vector<uint32_t> pg(npass + 1, 0);  // group boundaries of the passes, balanced by keys
	pg[npass] = ngroups;
	for (uint32_t p = 1; p < npass; p++) {
		const uint64_t target = (uint64_t)(((unsigned __int128)N * p) / npass);
		pg[p] = (uint32_t)(std::lower_bound(h_gpre.begin(), h_gpre.end(), target) - h_gpre.begin());
		pg[p] = std::max(pg[p], pg[p - 1]);
	}
	auto key_at = [&](uint32_t g) -> uint64_t { return g < ngroups ? h_gpre[g] : N; };  // first key of group g (local numbering)
	// first sorted super-k-mer of every pass: its keys start at the pass's first key
	std::vector<uint64_t> pj(npass + 1, 0);
	pj[npass] = nsk;
	if (npass > 1 && nsk) {
		std::vector<uint64_t> tgt(npass + 1);
		for (uint32_t p = 0; p <= npass; p++)
			tgt[p] = key_at(pg[p]);
		DevBuf d_tgt, d_lb;
		BLMX_HIP(d_tgt.alloc(tgt.size() * 8, false, s));
		BLMX_HIP(d_lb.alloc(tgt.size() * 8, false, s));
		BLMX_HIP(hipMemcpyAsync(d_tgt.p, tgt.data(), tgt.size() * 8, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(k_lower_bounds, dim3(1), dim3(128), 0, s, d_keystart.as<uint64_t>(), nsk, d_tgt.as<uint64_t>(), npass + 1, d_lb.as<uint64_t>());
		BLMX_HIP(hipGetLastError());
		BLMX_HIP(hipMemcpyAsync(pj.data(), d_lb.p, pj.size() * 8, hipMemcpyDeviceToHost, s));
		BLMX_HIP(hipStreamSynchronize(s));
		pj[0] = 0;
		pj[npass] = nsk;
	}
	uint64_t max_keys = 0, max_lines = 0;
	for (uint32_t p = 0; p < npass; p++) {
		max_keys = std::max(max_keys, key_at(pg[p + 1]) - key_at(pg[p]));
		max_lines = std::max(max_lines, h_line_base[pg[p + 1]] - h_line_base[pg[p]]);
	}

	// what every pass writes into
	DevBuf d_bits, d_coll, d_cnt, d_small, d_fbcnt, d_lvmax, d_lvcnt, d_gpopc, d_gfbstart, d_gfbcount;
	BLMX_HIP(d_bits.alloc((total_words + 1) * 8, true, s));
	BLMX_HIP(d_coll.alloc((total_words + 1) * 8, true, s));
	BLMX_HIP(d_cnt.alloc(8, true, s));
	BLMX_HIP(d_fbcnt.alloc((size_t)ngroups * 4, true, s));
	BLMX_HIP(d_lvmax.alloc(4, true, s));
	BLMX_HIP(d_gpopc.alloc(((size_t)ngroups + 1) * 8, true, s));
	BLMX_HIP(d_gfbstart.alloc((size_t)ngroups * 8, true, s));
	BLMX_HIP(d_gfbcount.alloc((size_t)ngroups * 4, true, s));
	BLMX_HIP(ix.lines.alloc((total_units + 1 + kBlockUnits) * kLineBytes, true, s));
	std::vector<uint8_t> h_small(ngroups, 0);  // 1 = one workgroup builds all levels in LDS, 2 = one workgroup per (level, slice), 0 = global kernels
	uint64_t n_small = 0, n_sliced = 0;
	const char* lds_env = getenv("BLIGHT_BUILD_NO_LDS");  // test hook: 1 = global level kernels for every group, 2 = for the groups above the
	const int lds_off = lds_env ? atoi(lds_env) : 0;      //            single-workgroup budget only (no sliced path)
	for (uint32_t g = 0; g < ngroups; g++) {
		const GroupDesc& gd = ix.h_groups[g];
		if (gd.nkeys && lds_off != 1 && gd.size[0] <= kLdsLevelBits * kLdsMaxSlices) {
			h_small[g] = 1;
			n_small++;
		} else if (gd.nkeys && lds_off == 0 && gd.nkeys < 0xffffffffull) {
			h_small[g] = 2;
			n_sliced++;
		}
	}
	BLMX_HIP(d_small.alloc((size_t)ngroups, false, s));
	BLMX_HIP(hipMemcpyAsync(d_small.p, h_small.data(), ngroups, hipMemcpyHostToDevice, s));
	if (n_sliced)
		BLMX_HIP(d_lvcnt.alloc((size_t)(kLevels + 1) * ngroups * 4, true, s));
	BuildGroups bg{d_gpre.as<uint64_t>(), d_bgroups.as<GroupDesc>(), d_loff.as<uint64_t>(), ngroups, d_small.as<uint8_t>()};
	// final group table minus the fallback ranges (they are only known after the passes; the kernels of the passes do not read them);
	// a shard's IDs are global: add the key count of all lower groups
	if (id_base)
		for (uint32_t g = 0; g < ngroups; g++)
			ix.h_groups[g].prefix += id_base;
	BLMX_HIP(ix.groups.alloc((size_t)ngroups * sizeof(GroupDesc), false, s));
	BLMX_HIP(hipMemcpyAsync(ix.groups.p, ix.h_groups.data(), (size_t)ngroups * sizeof(GroupDesc), hipMemcpyHostToDevice, s));
	ix.positions_words = phys_pos / 64 + 3 + 16;  // (+ 128 B: whole-line reads, index_device.h)
	BLMX_HIP(ix.positions.alloc(ix.positions_words * 8, true, s));
	ix.st.total_nt = total_nt;
	// absent-key filter: a separate array (filter_bits per key), or -- fat lines -- the six words behind every level-0 line
	ix.filter_bpk = 0;
	ix.filter_words = 0;
	ix.filter.release();
	if (bpk && N) {
		if (!fat) {
			ix.filter_words = (uint64_t)(((unsigned __int128)N * bpk) >> 6) + 64;
			BLMX_HIP(ix.filter.alloc(ix.filter_words * 8, true, s));
		}
		ix.filter_bpk = bpk;
		init_filter_gate(ix, s);
	}
	const bool have_filter = ix.filter_bpk != 0;
	unsigned long long* d_filter = (have_filter && !fat) ? ix.filter.as<unsigned long long>() : nullptr;
	DeviceIndexView v = ix.view();
	v.filter = have_filter ? (fat ? ix.lines.as<uint64_t>() : ix.filter.as<uint64_t>()) : nullptr;  // whatever the query-time switch says
	v.id_base = id_base;
	v.fb_keys = nullptr;  // (not built yet; k_fill_positions does not look there)

	// buffers of ONE pass: keys, two survivor lists, position fields, line popcounts
	DevBuf d_keys, d_lA, d_lB, d_field, d_linepc, d_linepre, d_lbits;
	if (npass > 1 && BuildPool::current() && !(getenv("BLIGHT_POOL_TRIM") && atoi(getenv("BLIGHT_POOL_TRIM")) == 0))
		BuildPool::current()->drain(true);  // bounded memory was asked for: the blocks K1-K5 have parked go back before the pass buffers are taken
	BLMX_HIP(d_keys.alloc((max_keys + 1) * 8 * KW, false, s));
	if (n_small || n_sliced) {
		BLMX_HIP(d_lA.alloc((max_keys + 1) * 4, false, s));
		BLMX_HIP(d_lB.alloc((max_keys + 1) * 4, false, s));
	}
	BLMX_HIP(d_field.alloc((max_keys + 1) * 4, false, s));
	if (n_sliced)
		BLMX_HIP(d_lbits.alloc((max_keys + 1) * 4, false, s));  // per-level bit indices of the sliced groups' keys (k_sliced_level_bits)
	BLMX_HIP(d_linepc.alloc((max_lines + 2) * 8, false, s));
	BLMX_HIP(d_linepre.alloc((max_lines + 2) * 8, false, s));
	// the fallback table grows pass by pass (groups in increasing order = table order)
	std::vector<std::unique_ptr<DevBuf>> fb_tk, fb_trank;
	std::vector<uint64_t> fb_count_of_pass;
	ix.h_fb_group.clear();
	uint64_t F = 0;
	int levels_used = 0;
	mark("passes: setup");

	// several passes: the filter's key pass of a pass runs on the background stream next to the pass's level and position kernels (it
	// only reads the keys); the next pass's keys wait for it
	// Several passes: the keys of a pass are gone when the next pass starts, so the filter's key pass either runs inside the passes
	// (BLIGHT_FILTER_IN_PASSES=1: next to the pass's level and position kernels, on the background stream) or -- default -- AFTER the
	// build has returned, like the one-pass build's: the background stream expands the keys of every pass once more (4 ms per 6e8 keys)
	// into the pass's key buffer, which leaves the build's pool together with what the expansion reads.
	const bool filter_in_passes = have_filter && npass > 1 && getenv("BLIGHT_FILTER_IN_PASSES") && atoi(getenv("BLIGHT_FILTER_IN_PASSES")) != 0;
	const bool filter_keys_later = have_filter && npass > 1 && !filter_in_passes;
	hipEvent_t ev_keys = nullptr, ev_filtered = nullptr;
	struct PassEvents {
		hipEvent_t &a, &b;
		~PassEvents()
		{
			if (a)
				(void)hipEventDestroy(a);
			if (b)
				(void)hipEventDestroy(b);
		}
	} pass_events{ev_keys, ev_filtered};
	if (filter_in_passes) {
		ix.filter_wait();
		if (!ix.bg_stream)
			BLMX_HIP(hipStreamCreateWithFlags(&ix.bg_stream, hipStreamNonBlocking));
		BLMX_HIP(hipEventCreateWithFlags(&ev_keys, hipEventDisableTiming));
		BLMX_HIP(hipEventCreateWithFlags(&ev_filtered, hipEventDisableTiming));
	}
	bool filter_pending_on_bg = false;
	for (uint32_t p = 0; p < npass; p++) {
		const uint32_t g0 = pg[p], g1 = pg[p + 1];
		const uint64_t kb0 = key_at(g0), kb1 = key_at(g1), Np = kb1 - kb0, j0 = pj[p], j1 = pj[p + 1];
		if (Np == 0)
			continue;
		if (filter_pending_on_bg)
			BLMX_HIP(hipStreamWaitEvent(s, ev_filtered, 0));  // the previous pass's keys are still being read
		// absolute key numbers index the pass buffers: pointers shifted by the pass's first key
		uint64_t* keys_p = reinterpret_cast<uint64_t*>(reinterpret_cast<uintptr_t>(d_keys.p) - (uintptr_t)kb0 * 8 * KW);
		uint32_t* lA_p = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(d_lA.p) - (uintptr_t)kb0 * 4);
		uint32_t* lB_p = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(d_lB.p) - (uintptr_t)kb0 * 4);
		uint32_t* field_p = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(d_field.p) - (uintptr_t)kb0 * 4);
		// ---- K6: keys in (bucket, position) order
		hipLaunchKernelGGL((k_expand_keys<KW>), dim3(grid_for(j1 - j0, kSkPerBlock)), dim3(kBlock), 0, s, d_src.as<uint64_t>(), d_order.as<uint32_t>(),
		                   d_skpos.as<uint64_t>(), d_keystart.as<uint64_t>(), j0, j1, k, keys_p);
		BLMX_HIP(hipGetLastError());
		if (filter_in_passes) {
			BLMX_HIP(hipEventRecord(ev_keys, s));
			BLMX_HIP(hipStreamWaitEvent(ix.bg_stream, ev_keys, 0));
			hipLaunchKernelGGL((k_filter_keys<KW>), dim3((grid_for(j1 - j0, kSkPerBlock) + 7u) & ~7u), dim3(kBlock), 0, ix.bg_stream, v, d_smini.as<uint32_t>(),
			                   d_keystart.as<uint64_t>(), j0, j1, keys_p, d_filter);
			BLMX_HIP(hipGetLastError());
			BLMX_HIP(hipEventRecord(ev_filtered, ix.bg_stream));
			filter_pending_on_bg = true;
		}
		if (npass == 1)
			mark("K6 expand_keys");
		// ---- K7: BBHash levels.  Groups whose level-0 array fits the LDS budget are built by one workgroup each
		// (k_mphf_group_lds); the others go through the global level kernels, which skip the keys of the LDS-built groups.
		std::vector<uint32_t> h_order, h_sliced;
		uint64_t keys_large = 0;
		for (uint32_t g = g0; g < g1; g++) {
			if (h_small[g] == 1)
				h_order.push_back(g);
			else if (h_small[g] == 2)
				h_sliced.push_back(g);
			else
				keys_large += ix.h_groups[g].nkeys;
		}
		auto by_size = [&](uint32_t a, uint32_t b2) { return ix.h_groups[a].nkeys != ix.h_groups[b2].nkeys ? ix.h_groups[a].nkeys > ix.h_groups[b2].nkeys : a < b2; };
		uint64_t F_lds = 0;
		DevBuf d_fball;
		if (!h_order.empty() || !h_sliced.empty()) {
			if (!h_sliced.empty()) {
				// work items (group, slice) of every level, largest groups first inside a level
				std::sort(h_sliced.begin(), h_sliced.end(), by_size);
				std::vector<uint2> h_items;
				size_t level_off[kLevels + 1];
				for (int l = 0; l < kLevels; l++) {
					level_off[l] = h_items.size();
					for (uint32_t g : h_sliced) {
						const uint32_t words = (uint32_t)(ix.h_groups[g].size[l] >> 5);
						for (uint32_t sl = 0; sl * kLdsLevelWords < words; sl++)
							h_items.push_back(make_uint2(g, sl));
					}
				}
				level_off[kLevels] = h_items.size();
				DevBuf d_items, d_glist;
				BLMX_HIP(d_items.alloc(h_items.size() * sizeof(uint2), false, s));
				BLMX_HIP(d_glist.alloc(h_sliced.size() * 4, false, s));
				BLMX_HIP(hipMemcpyAsync(d_items.p, h_items.data(), h_items.size() * sizeof(uint2), hipMemcpyHostToDevice, s));
				BLMX_HIP(hipMemcpyAsync(d_glist.p, h_sliced.data(), h_sliced.size() * 4, hipMemcpyHostToDevice, s));
				uint32_t* lbits_p = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(d_lbits.p) - (uintptr_t)kb0 * 4);
				uint64_t largest = 1;
				for (uint32_t g : h_sliced)
					largest = std::max<uint64_t>(largest, ix.h_groups[g].nkeys);
				for (int l = 0; l < kLevels; l++) {
					const size_t ni = level_off[l + 1] - level_off[l];
					if (ni) {
						// (the grid covers the largest group's level-0 list; deeper lists are shorter: workgroups beyond a list leave at once)
						const unsigned gx = (unsigned)std::min<uint64_t>((largest + kBlock - 1) / kBlock, 1u << 16);
						hipLaunchKernelGGL((k_sliced_level_bits<KW>), dim3(l == 0 ? gx : std::max(1u, gx >> std::min(l, 6)), (unsigned)h_sliced.size()), dim3(kBlock), 0, s, bg, keys_p, l,
						                   d_glist.as<uint32_t>(), lA_p, lB_p, d_lvcnt.as<uint32_t>(), lbits_p);
						hipLaunchKernelGGL((k_mphf_level_sliced<KW>), dim3((unsigned)ni), dim3(kGroupBlock), 0, s, bg, keys_p, l, d_items.as<uint2>() + level_off[l], lA_p,
						                   lB_p, d_lvcnt.as<uint32_t>(), d_bits.as<uint32_t>(), lbits_p);
					}
					if (trace) {
						char what[64];
						snprintf(what, sizeof what, "  K7 sliced level %d (%zu wg)", l, ni);
						mark(what);
					}
				}
				hipLaunchKernelGGL(k_sliced_finish, dim3(grid_for(h_sliced.size())), dim3(kBlock), 0, s, d_glist.as<uint32_t>(), (uint32_t)h_sliced.size(),
				                   ngroups, d_lvcnt.as<uint32_t>(), d_fbcnt.as<uint32_t>(), d_lvmax.as<int>());
				BLMX_HIP(hipGetLastError());
				BLMX_HIP(hipStreamSynchronize(s));  // the host vectors and the item buffers go out of scope
			}
			DevBuf d_gorder;
			if (!h_order.empty()) {
				std::sort(h_order.begin(), h_order.end(), by_size);  // largest groups first: the multi-slice groups must not be the tail of the grid
				BLMX_HIP(d_gorder.alloc(h_order.size() * 4, false, s));
				BLMX_HIP(hipMemcpyAsync(d_gorder.p, h_order.data(), h_order.size() * 4, hipMemcpyHostToDevice, s));
				hipLaunchKernelGGL((k_mphf_group_lds<KW>), dim3((unsigned)h_order.size()), dim3(kGroupBlock), 0, s, bg, keys_p, lA_p, lB_p, d_bits.as<uint32_t>(),
				                   d_fbcnt.as<uint32_t>(), d_lvmax.as<int>(), d_gorder.as<uint32_t>());
				BLMX_HIP(hipGetLastError());
			}
			std::vector<uint32_t> h_fbcnt(g1 - g0);
			BLMX_HIP(hipMemcpyAsync(h_fbcnt.data(), d_fbcnt.as<uint32_t>() + g0, (size_t)(g1 - g0) * 4, hipMemcpyDeviceToHost, s));
			BLMX_HIP(hipStreamSynchronize(s));
			std::vector<uint64_t> h_fboff(ngroups, 0);
			for (uint32_t g = g0; g < g1; g++) {
				h_fboff[g] = F_lds;
				F_lds += h_fbcnt[g - g0];
			}
			if (F_lds) {
				DevBuf d_fboff;
				BLMX_HIP(d_fboff.alloc((size_t)ngroups * 8, false, s));
				BLMX_HIP(hipMemcpyAsync(d_fboff.p, h_fboff.data(), (size_t)ngroups * 8, hipMemcpyHostToDevice, s));
				BLMX_HIP(d_fball.alloc((F_lds + 1) * 8, false, s));
				hipLaunchKernelGGL(k_fb_collect, dim3(g1 - g0), dim3(256), 0, s, d_gpre.as<uint64_t>(), d_fbcnt.as<uint32_t>(), d_fboff.as<uint64_t>(), lA_p, ngroups,
				                   d_fball.as<uint64_t>(), g0);
				BLMX_HIP(hipGetLastError());
				BLMX_HIP(hipStreamSynchronize(s));  // h_fboff / d_fboff go out of scope
			}
		}
		// global path: seeded with the keys of the large groups only
		if (keys_large && N >= 0xffffffffull)
			throw StatusError{BLMX_ELIMIT, "the global MPHF level kernels index keys with 32 bits (an MPHF group of 2^32 keys or more, or BLIGHT_BUILD_NO_LDS)"};
		uint64_t count = keys_large, Fg = 0;
		const uint32_t* cur = nullptr;
		uint32_t* nxt = nullptr;
		DevBuf d_seed, d_listA, d_listB;
		if (keys_large) {
			std::vector<uint64_t> h_seedoff(ngroups, 0);
			uint64_t o = 0;
			for (uint32_t g = g0; g < g1; g++) {
				h_seedoff[g] = o;
				if (!h_small[g])
					o += ix.h_groups[g].nkeys;
			}
			DevBuf d_seedoff;
			BLMX_HIP(d_seedoff.alloc((size_t)ngroups * 8, false, s));
			BLMX_HIP(hipMemcpyAsync(d_seedoff.p, h_seedoff.data(), (size_t)ngroups * 8, hipMemcpyHostToDevice, s));
			BLMX_HIP(d_seed.alloc((keys_large + 1) * 4, false, s));
			hipLaunchKernelGGL(k_seed_large_list, dim3(g1 - g0), dim3(256), 0, s, d_gpre.as<uint64_t>(), d_small.as<uint8_t>(), d_seedoff.as<uint64_t>(), ngroups,
			                   d_seed.as<uint32_t>(), g0);
			BLMX_HIP(hipGetLastError());
			BLMX_HIP(hipStreamSynchronize(s));  // h_seedoff goes out of scope
			cur = d_seed.as<uint32_t>();
		}
		for (int level = 0; level < kLevels && count > 0; level++) {
			levels_used = std::max(levels_used, level + 1);
			hipLaunchKernelGGL((k_mphf_mark<KW>), dim3(grid_for(count)), dim3(kBlock), 0, s, bg, keys_p, cur, count, level, d_bits.as<unsigned long long>(),
			                   d_coll.as<unsigned long long>());
			DevBuf& dst = (level & 1) ? d_listB : d_listA;
			BLMX_HIP(dst.alloc((std::min(count, keys_large) + 1) * 4, false, s));
			nxt = dst.as<uint32_t>();
			BLMX_HIP(hipMemsetAsync(d_cnt.p, 0, 8, s));
			hipLaunchKernelGGL((k_mphf_survive<KW>), dim3(grid_for(count, kSurviveBlock)), dim3(kSurviveBlock), 0, s, bg, keys_p, cur, count, level,
			                   d_coll.as<unsigned long long>(), nxt, d_cnt.as<unsigned long long>());
			BLMX_HIP(hipGetLastError());
			count = read_scalar(d_cnt.as<unsigned long long>(), s);
			cur = nxt;
			if (level == kLevels - 1)
				Fg = count;
		}
		// one fallback list of absolute key indices: the global path's survivors followed by those of the LDS-built groups
		// (sorted by key index below)
		const uint64_t Fp = Fg + F_lds;
		if (F + Fp >= 0xffffffffull)
			throw StatusError{BLMX_ELIMIT, "2^32 - 1 or more fallback keys"};
		DevBuf d_fbmerged;
		if (Fp) {
			BLMX_HIP(d_fbmerged.alloc((Fp + 1) * 8, false, s));
			if (Fg)
				hipLaunchKernelGGL(k_widen_u32, dim3(grid_for(Fg)), dim3(kBlock), 0, s, cur, Fg, d_fbmerged.as<uint64_t>());
			if (F_lds)
				BLMX_HIP(hipMemcpyAsync(d_fbmerged.as<uint64_t>() + Fg, d_fball.p, F_lds * 8, hipMemcpyDeviceToDevice, s));
			BLMX_HIP(hipGetLastError());
		}
		if (npass == 1)
			mark("K7 mphf levels");
		// ---- K8: collisions cleared + ranks -> lines [u32 rank | 96 bits] of this pass's groups
		const uint64_t L0 = h_line_base[g0], L1 = h_line_base[g1];
		hipLaunchKernelGGL(k_line_popc, dim3(grid_for(L1 - L0 + 1)), dim3(kBlock), 0, s, d_bgroups.as<GroupDesc>(), d_line_base.as<uint64_t>(), ngroups, L0, L1,
		                   d_bits.as<unsigned long long>(), d_coll.as<unsigned long long>(), d_linepc.as<uint64_t>(), layout);
		BLMX_HIP(hipGetLastError());
		exclusive_scan_u64(sc, d_linepc.as<uint64_t>(), d_linepre.as<uint64_t>(), L1 - L0 + 1, s);
		if (L1 > L0)
			hipLaunchKernelGGL(k_line_write, dim3(grid_for(L1 - L0)), dim3(kBlock), 0, s, d_bgroups.as<GroupDesc>(), d_line_base.as<uint64_t>(), ngroups, L0, L1,
			                   d_bits.as<unsigned long long>(), d_coll.as<unsigned long long>(), d_linepre.as<uint64_t>(), ix.lines.as<uint4>(), layout);
		hipLaunchKernelGGL(k_group_popc, dim3(grid_for(g1 - g0)), dim3(kBlock), 0, s, d_line_base.as<uint64_t>(), d_linepre.as<uint64_t>(), g0, g1, L0,
		                   d_gpopc.as<uint64_t>());
		BLMX_HIP(hipGetLastError());
		if (npass == 1)
			mark("K8 lines+ranks");
		// ---- K9a: fallback table of this pass's groups
		DevBuf d_fbsorted, d_perm, d_tgrp;
		DevBuf* tk = nullptr;
		DevBuf* trank = nullptr;
		if (Fp) {
			DevBuf d_fk, d_frank, d_fgrp;
			BLMX_HIP(d_fbsorted.alloc(Fp * 8, false, s));
			unsigned idx_bits = 1;
			while (idx_bits < 64 && (N >> idx_bits))
				idx_bits++;
			stable_sort_pairs<uint64_t>(sc, d_fbmerged.as<uint64_t>(), d_fbsorted.as<uint64_t>(), nullptr, nullptr, Fp, idx_bits, s);
			BLMX_HIP(d_fk.alloc(Fp * 8 * KW, false, s));
			BLMX_HIP(d_frank.alloc(Fp * 8, false, s));
			BLMX_HIP(d_fgrp.alloc(Fp * 4, false, s));
			hipLaunchKernelGGL((k_fb_prepare<KW>), dim3(grid_for(Fp)), dim3(kBlock), 0, s, bg, keys_p, d_fbsorted.as<uint64_t>(), Fp, d_gpopc.as<uint64_t>(),
			                   d_fk.as<uint64_t>(), d_frank.as<uint64_t>(), d_fgrp.as<uint32_t>());
			BLMX_HIP(hipGetLastError());
			// table order = (group, key, input order).  The entries arrive in input order (= group, position), so two or three
			// STABLE radix sorts do it: by the key (low word, then high word), then by the group -- a comparison sort through an
			// indirect comparator took 100 ms for the 3.7e7 fallback keys of the skewed human-scale input.
			BLMX_HIP(d_perm.alloc(Fp * 4, false, s));
			{
				DevBuf d_k0, d_k1, d_p0, d_p1;
				BLMX_HIP(d_k0.alloc(Fp * 8, false, s));
				BLMX_HIP(d_k1.alloc(Fp * 8, false, s));
				BLMX_HIP(d_p0.alloc(Fp * 4, false, s));
				BLMX_HIP(d_p1.alloc(Fp * 4, false, s));
				auto sort_by = [&](const uint64_t* kin, const uint32_t* pin, uint32_t* pout, unsigned bits) {
					stable_sort_pairs<uint64_t>(sc, kin, d_k1.as<uint64_t>(), pin, pout, Fp, bits, s);
				};
				hipLaunchKernelGGL(k_fb_sortkey, dim3(grid_for(Fp)), dim3(kBlock), 0, s, d_fk.as<uint64_t>(), d_fgrp.as<uint32_t>(), (const uint32_t*)nullptr, Fp, KW, 0,
				                   d_k0.as<uint64_t>(), d_p0.as<uint32_t>());
				sort_by(d_k0.as<uint64_t>(), d_p0.as<uint32_t>(), d_p1.as<uint32_t>(), 64u);
				uint32_t* cur_p = d_p1.as<uint32_t>();
				uint32_t* other_p = d_p0.as<uint32_t>();
				if (KW == 2) {
					hipLaunchKernelGGL(k_fb_sortkey, dim3(grid_for(Fp)), dim3(kBlock), 0, s, d_fk.as<uint64_t>(), d_fgrp.as<uint32_t>(), cur_p, Fp, KW, 1,
					                   d_k0.as<uint64_t>(), (uint32_t*)nullptr);
					sort_by(d_k0.as<uint64_t>(), cur_p, other_p, 64u);
					std::swap(cur_p, other_p);
				}
				hipLaunchKernelGGL(k_fb_sortkey, dim3(grid_for(Fp)), dim3(kBlock), 0, s, d_fk.as<uint64_t>(), d_fgrp.as<uint32_t>(), cur_p, Fp, KW, 2, d_k0.as<uint64_t>(),
				                   (uint32_t*)nullptr);
				sort_by(d_k0.as<uint64_t>(), cur_p, d_perm.as<uint32_t>(), 32u);
				BLMX_HIP(hipGetLastError());
				BLMX_HIP(hipStreamSynchronize(s));
			}
			fb_tk.emplace_back(new DevBuf());
			fb_trank.emplace_back(new DevBuf());
			tk = fb_tk.back().get();
			trank = fb_trank.back().get();
			BLMX_HIP(tk->alloc((Fp + 1) * 8 * KW, false, s));
			BLMX_HIP(trank->alloc((Fp + 1) * 8, false, s));
			fb_count_of_pass.push_back(Fp);
			BLMX_HIP(d_tgrp.alloc(Fp * 4, false, s));
			hipLaunchKernelGGL((k_fb_table<KW>), dim3(grid_for(Fp)), dim3(kBlock), 0, s, d_perm.as<uint32_t>(), Fp, d_fk.as<uint64_t>(), d_frank.as<uint64_t>(),
			                   d_fgrp.as<uint32_t>(), tk->as<uint64_t>(), trank->as<uint64_t>(), d_tgrp.as<uint32_t>(), d_gfbstart.as<uint64_t>(),
			                   d_gfbcount.as<uint32_t>());
			BLMX_HIP(hipGetLastError());
			std::vector<uint64_t> h_fs(g1 - g0);
			std::vector<uint32_t> h_fc(g1 - g0);
			const size_t at = ix.h_fb_group.size();
			ix.h_fb_group.resize(at + Fp);
			BLMX_HIP(hipMemcpyAsync(h_fs.data(), d_gfbstart.as<uint64_t>() + g0, (size_t)(g1 - g0) * 8, hipMemcpyDeviceToHost, s));
			BLMX_HIP(hipMemcpyAsync(h_fc.data(), d_gfbcount.as<uint32_t>() + g0, (size_t)(g1 - g0) * 4, hipMemcpyDeviceToHost, s));
			BLMX_HIP(hipMemcpyAsync(ix.h_fb_group.data() + at, d_tgrp.p, Fp * 4, hipMemcpyDeviceToHost, s));
			BLMX_HIP(hipStreamSynchronize(s));
			for (uint32_t g = g0; g < g1; g++) {  // (k_fb_table numbers the entries of THIS pass: the table of the passes before lies in front)
				ix.h_groups[g].fb_start = h_fc[g - g0] ? F + h_fs[g - g0] : 0;
				ix.h_groups[g].fb_count = h_fc[g - g0] ? h_fc[g - g0] - (uint32_t)h_fs[g - g0] : 0;
			}
		}
		if (npass == 1)
			mark("K9a fallback");
		// ---- K9b: positions of this pass's keys
		BLMX_HIP(hipMemsetAsync(d_field.p, 0, (Np + 1) * 4, s));  // ranks nobody owns -- overwritten duplicates -- stay 0 like in the reference
		hipLaunchKernelGGL((k_fill_positions<KW>), dim3((grid_for(j1 - j0, kSkPerBlock) + 7u) & ~7u), dim3(kBlock), 0, s, v, d_smini.as<uint32_t>(),
		                   d_ntstart.as<uint64_t>(), d_keystart.as<uint64_t>(), j0, j1, keys_p, id_base, field_p);
		if (Fp)
			hipLaunchKernelGGL((k_fb_positions<KW>), dim3(grid_for(Fp)), dim3(kBlock), 0, s, v, d_perm.as<uint32_t>(), d_fbsorted.as<uint64_t>(), Fp, tk->as<uint64_t>(),
			                   trank->as<uint64_t>(), d_tgrp.as<uint32_t>(), d_smini.as<uint32_t>(), d_ntstart.as<uint64_t>(), d_keystart.as<uint64_t>(), nsk, id_base,
			                   field_p);
		hipLaunchKernelGGL(k_pack_positions, dim3(g1 - g0), dim3(kPackThreads), 0, s, ix.groups.as<GroupDesc>(), g0, id_base, field_p,
		                   ix.positions.as<unsigned long long>());
		if (rec && L1 > L0)  // level-0 records: the fields of the keys they rank, next to the rank
			hipLaunchKernelGGL(k_record_positions, dim3(grid_for(L1 - L0)), dim3(kBlock), 0, s, d_bgroups.as<GroupDesc>(), d_line_base.as<uint64_t>(), ngroups, L0, L1,
			                   ix.positions.as<uint64_t>(), ix.lines.as<uint4>());
		BLMX_HIP(hipGetLastError());
		BLMX_HIP(hipStreamSynchronize(s));  // the pass's temporaries go out of scope
		F += Fp;
		if (npass > 1 && trace) {
			char what[64];
			snprintf(what, sizeof what, "pass %u of %u (%llu keys)", p + 1, npass, (unsigned long long)Np);
			mark(what);
		}
	}
	if (filter_pending_on_bg)
		BLMX_HIP(hipEventSynchronize(ev_filtered));  // the last pass's keys
	levels_used = std::max(levels_used, read_scalar(d_lvmax.as<int>(), s));
	if (!filter_keys_later) {
		d_src.release();
		d_skpos.release();
		d_order.release();
	}
	d_bits.release();
	d_coll.release();
	d_lA.release();
	d_lB.release();
	d_field.release();
	d_linepc.release();
	d_linepre.release();
	// the fallback table: the passes' segments back to back
	BLMX_HIP(ix.fb_keys.alloc((F + 1) * 8 * KW, true, s));
	BLMX_HIP(ix.fb_ranks.alloc((F + 1) * 8, true, s));
	{
		uint64_t at = 0;
		for (size_t i = 0; i < fb_tk.size(); i++) {
			const uint64_t c = fb_count_of_pass[i];
			BLMX_HIP(hipMemcpyAsync(ix.fb_keys.as<uint64_t>() + at * KW, fb_tk[i]->p, c * 8 * KW, hipMemcpyDeviceToDevice, s));
			BLMX_HIP(hipMemcpyAsync(ix.fb_ranks.as<uint64_t>() + at, fb_trank[i]->p, c * 8, hipMemcpyDeviceToDevice, s));
			at += c;
		}
		BLMX_HIP(hipStreamSynchronize(s));
		fb_tk.clear();
		fb_trank.clear();
	}
	// final group table (with the fallback ranges)
	BLMX_HIP(hipMemcpyAsync(ix.groups.p, ix.h_groups.data(), (size_t)ngroups * sizeof(GroupDesc), hipMemcpyHostToDevice, s));
	make_query_heads(ix);
	make_fallback_table(ix, s);
	v = ix.view();
	v.filter = have_filter ? (fat ? ix.lines.as<uint64_t>() : ix.filter.as<uint64_t>()) : nullptr;
	v.id_base = id_base;
	mark(npass == 1 ? "K9b positions" : "passes: fallback table + group table");
	{
		// the filter's passes on their own stream: the build does not wait for them, queries use the filter once they are done.  With one
		// pass: every key (k_filter_keys) and the junction k-mers; with several: the junction k-mers (the keys went in pass by pass).
		ix.filter_wait();
		ix.filter_state.store(2);
		if (have_filter && nsk) {
			if (!ix.bg_stream)
				BLMX_HIP(hipStreamCreateWithFlags(&ix.bg_stream, hipStreamNonBlocking));
			if (!ix.filter_ready)
				BLMX_HIP(hipEventCreateWithFlags(&ix.filter_ready, hipEventDisableTiming));
			// what the passes read leaves the build's pool with them (the helper thread frees it)
			DevBuf* hand_over[7] = {&d_ntstart, &d_smini, &d_keystart, &d_keys, &d_src, &d_order, &d_skpos};
			for (int i = 0; i < 7; i++) {
				ix.filter_pending[i].release();
				if (i > 0 && filter_in_passes)
					continue;  // the keys went in pass by pass: only the super-k-mer starts are still needed (junctions)
				if (i > 3 && !filter_keys_later)
					continue;  // (already released: the keys of a one-pass build are all there)
				ix.filter_pending[i].p = hand_over[i]->p;
				ix.filter_pending[i].bytes = hand_over[i]->bytes;
				ix.filter_pending[i].cap = hand_over[i]->cap;
				ix.filter_pending[i].owner = nullptr;
				if (hand_over[i]->owner)
					hand_over[i]->owner->held -= std::min(hand_over[i]->owner->held, hand_over[i]->bytes);
				hand_over[i]->p = nullptr;
				hand_over[i]->bytes = 0;
				hand_over[i]->cap = 0;
				hand_over[i]->owner = nullptr;
			}
			ix.filter_state.store(0);
			if (npass == 1)
				hipLaunchKernelGGL((k_filter_keys<KW>), dim3((grid_for(nsk, kSkPerBlock) + 7u) & ~7u), dim3(kBlock), 0, ix.bg_stream, v,
				                   ix.filter_pending[1].as<uint32_t>(), ix.filter_pending[2].as<uint64_t>(), (uint64_t)0, nsk, ix.filter_pending[3].as<uint64_t>(), d_filter);
			if (filter_keys_later)
				for (uint32_t p = 0; p < npass; p++) {
					const uint64_t kb0 = key_at(pg[p]), j0 = pj[p], j1 = pj[p + 1];
					if (key_at(pg[p + 1]) == kb0)
						continue;
					uint64_t* keys_p = reinterpret_cast<uint64_t*>(reinterpret_cast<uintptr_t>(ix.filter_pending[3].p) - (uintptr_t)kb0 * 8 * KW);
					hipLaunchKernelGGL((k_expand_keys<KW>), dim3(grid_for(j1 - j0, kSkPerBlock)), dim3(kBlock), 0, ix.bg_stream, ix.filter_pending[4].as<uint64_t>(),
					                   ix.filter_pending[5].as<uint32_t>(), ix.filter_pending[6].as<uint64_t>(), ix.filter_pending[2].as<uint64_t>(), j0, j1, k, keys_p);
					hipLaunchKernelGGL((k_filter_keys<KW>), dim3((grid_for(j1 - j0, kSkPerBlock) + 7u) & ~7u), dim3(kBlock), 0, ix.bg_stream, v,
					                   ix.filter_pending[1].as<uint32_t>(), ix.filter_pending[2].as<uint64_t>(), j0, j1, keys_p, d_filter);
				}
			if (ix.tune_filter_junctions) {
				const uint64_t cand = (nsk - 1) * (uint64_t)(k - 1) + (uint64_t)(k - 1) + (1ull << b), blocks = (cand + kBlock - 1) / kBlock;
				const unsigned gx = (unsigned)std::min<uint64_t>(blocks, 1u << 22), gy = (unsigned)((blocks + gx - 1) / gx);
				hipLaunchKernelGGL((k_filter_junctions<KW>), dim3(gx, gy), dim3(kBlock), 0, ix.bg_stream, v, ix.filter_pending[0].as<uint64_t>(),
				                   (nsk - 1) * (uint64_t)(k - 1), d_filter);
			}
			BLMX_HIP(hipGetLastError());
			BLMX_HIP(hipEventRecord(ix.filter_ready, ix.bg_stream));
			if (mark.on && !mark.host_only) {
				BLMX_HIP(hipEventSynchronize(ix.filter_ready));
				mark("K9c filter: keys + junctions (in the background when not traced)");
			}
		}
	}

	// ---- public counters (blight.h:52-59, recap blight.cpp:361-378)
	ix.total_lines = total_units;  // 16-B units of the line array
	ix.total_logical_words = total_words;
	blmx_stats_t& st = ix.st;
	st.number_kmer = world > 1 ? global_N : N;
	st.local_number_kmer = N;
	st.number_super_kmer = world > 1 ? global_nsk : nsk;
	st.shard_rank = rank;
	st.shard_world = world;
	st.shard_group_lo = glo;
	st.shard_group_hi = ghi;
	st.mphf_groups_lds_built = (uint32_t)(n_small + n_sliced);
	st.largest_mphf = largest_mphf;
	st.positions_total_size = total_pos;
	st.total_nt = total_nt;
	st.largest_bucket = largest_bucket;
	st.mphf_bytes = total_units * kLineBytes;
	st.fallback_keys = F;
	st.nbuckets = nbuckets;
	st.ngroups = ngroups;
	st.k = ix.prm.k;
	st.m = ix.prm.m;
	st.log2_mphf = ix.prm.log2_mphf;
	st.bits_saved = ix.prm.bits_saved;
	st.kmer_bytes = (uint32_t)(8 * KW);
	st.max_levels_used = (uint32_t)levels_used;
	st.device_bytes = ix.buckets.bytes + ix.groups.bytes + ix.qheads.bytes + ix.seq.bytes + ix.positions.bytes + ix.lines.bytes + ix.filter.bytes + ix.fb_keys.bytes +
	                  ix.fb_ranks.bytes + ix.fb_table.bytes + ix.fb_dir.bytes;
	st.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
}

// ---- sharded build from slices: every rank chops ITS OWN sequences; super-k-mers travel to the owner of their
// MPHF group.  Slices are consecutive pieces of the input in rank order, so "source rank, then position in the slice" is the
// input order and the stable minimizer sort of the tail reproduces the single-GPU bucket content, hence the same IDs.
}  // namespace

// blmx_primitive op 8 (tests): the build's stable sort on its own -- n keys with `bits` significant bits -> n sorted keys, then the n
// input indices in that order
void launch_sort_primitive(const uint32_t* d_keys, uint64_t n, unsigned bits, uint32_t* d_out, hipStream_t s)
{
	Scratch sc;
	sc.s = s;
	stable_sort_by_key_u32(sc, d_keys, d_out, d_out + n, n, bits, s);
	BLMX_HIP(hipStreamSynchronize(s));
}

struct ShardExchange {
	BuildPool pool;  // first: outlives the buffers below
	std::chrono::steady_clock::time_point t_begin;
	Chopped local;
	int phase = 0;  // 1 = chopped, 2 = outgoing streams packed
	ShardCtx shard;
	DevBuf out_words, out_meta;
};

void shard_build_abort(Index& ix)
{
	if (!ix.xb)
		return;
	PhaseMarks mark(ix.stream);
	if (ix.filter_state.load() == 0) {  // the filter's background pass is running: the helper thread frees the blocks
		for (DevBuf* b : {&ix.xb->local.src, &ix.xb->local.skpos, &ix.xb->local.sklast, &ix.xb->local.skmini, &ix.xb->out_words, &ix.xb->out_meta})
			b->release();
		ix.deferred_blocks.insert(ix.deferred_blocks.end(), ix.xb->pool.free_blocks.begin(), ix.xb->pool.free_blocks.end());
		ix.xb->pool.free_blocks.clear();
	}
	delete ix.xb;
	ix.xb = nullptr;
	ix.start_filter_thread();
	mark("exchange: free scratch");
}

namespace {

template <int KW>
void shard_begin_impl(Index& ix, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len, uint64_t* hist_out, DevBuf* packed)
{
	shard_build_abort(ix);
	std::unique_ptr<ShardExchange> x(new ShardExchange());
	x->t_begin = std::chrono::steady_clock::now();
	BuildPoolScope scope(&x->pool);
	PhaseMarks mark(ix.stream);
	chop_input<KW>(ix, d_ascii, d_offs, nseq, total_len, x->local, mark, packed);
	const std::vector<uint64_t> h = group_histogram(ix, x->local);
	memcpy(hist_out, h.data(), (size_t)ix.ngroups * 8);
	hist_out[ix.ngroups] = x->local.nsk;
	x->phase = 1;
	ix.xb = x.release();
}

}  // namespace

void shard_build_begin(Index& ix, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len, uint64_t* hist_out, DevBuf* packed)
{
	if (ix.built)
		throw StatusError{BLMX_ESTATE, "index already built"};
	if (ix.KW == 1)
		shard_begin_impl<1>(ix, d_ascii, d_offs, nseq, total_len, hist_out, packed);
	else
		shard_begin_impl<2>(ix, d_ascii, d_offs, nseq, total_len, hist_out, packed);
}

// ---- a consecutive part of a 2-bit text (the file loader's output) as a text of its own: sequences [u0, u1) --------------------------
namespace {
__global__ void k_slice_text(const uint64_t* __restrict__ in, uint64_t start_nt, uint64_t len_nt, uint64_t nwords, uint64_t* __restrict__ out)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords)
		return;
	const uint64_t p0 = w * 32;
	uint64_t v = 0;
	if (p0 < len_nt) {
		const uint64_t q = start_nt + p0;
		v = funnel_left(in[q >> 5], in[(q >> 5) + 1], 2 * (int)(q & 31));
		const uint64_t rem = len_nt - p0;
		if (rem < 32)
			v &= ~0ull << (64 - 2 * rem);  // what follows the part belongs to the next rank
	}
	out[w] = v;
}
__global__ void k_slice_offsets(const uint64_t* __restrict__ offs, uint64_t u0, uint64_t n, uint64_t* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		out[i] = offs[u0 + i] - offs[u0];
}
}  // namespace

void slice_packed_text(const DevBuf& d_text, const uint64_t* d_offs_all, uint64_t u0, uint64_t u1, DevBuf& out_text, DevBuf& out_offs, uint64_t& total_len,
                       hipStream_t s)
{
	uint64_t h[2] = {0, 0};
	BLMX_HIP(hipMemcpyAsync(&h[0], d_offs_all + u0, 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipMemcpyAsync(&h[1], d_offs_all + u1, 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	total_len = h[1] - h[0];
	const uint64_t nwords = (total_len + 31) / 32 + 4;
	BLMX_HIP(out_text.alloc(nwords * 8, false, s));
	BLMX_HIP(out_offs.alloc((u1 - u0 + 1) * 8, false, s));
	hipLaunchKernelGGL(k_slice_text, dim3(grid_for(nwords)), dim3(kBlock), 0, s, d_text.as<uint64_t>(), h[0], total_len, nwords, out_text.as<uint64_t>());
	hipLaunchKernelGGL(k_slice_offsets, dim3(grid_for(u1 - u0 + 1)), dim3(kBlock), 0, s, d_offs_all, u0, u1 - u0 + 1, out_offs.as<uint64_t>());
	BLMX_HIP(hipGetLastError());
}

void shard_build_plan(Index& ix, const uint64_t* global_hist, uint32_t world_of_caller, uint64_t* send_sk, uint64_t* send_words, const void** d_meta,
                      const void** d_words)
{
	if (world_of_caller != ix.shard_world)  // send_sk / send_words hold one entry per rank of the index's world
		throw StatusError{BLMX_EINVAL, "blmx_shard_build_plan: world differs from the one given to blmx_set_shard"};
	ShardExchange* x = ix.xb;
	if (!x || x->phase != 1)
		throw StatusError{BLMX_ESTATE, "blmx_shard_build_plan needs a blmx_shard_build_begin first"};
	hipStream_t s = ix.stream;
	BuildPoolScope scope(&x->pool);
	Scratch sc;
	sc.s = s;
	const uint32_t world = ix.shard_world, rank = ix.shard_rank, ngroups = ix.ngroups;
	const int k = (int)ix.prm.k;
	const std::vector<uint64_t> hist(global_hist, global_hist + ngroups);
	const std::vector<uint64_t> pre = derive_shard_bounds(ix, hist);
	x->shard.global_N = pre[ngroups];
	x->shard.global_nsk = global_hist[ngroups];
	x->shard.glo = ix.shard_bounds[rank];
	x->shard.ghi = ix.shard_bounds[rank + 1];
	x->shard.id_base = pre[x->shard.glo];
	Chopped& c = x->local;
	const uint64_t nsk = c.nsk;
	OwnerTable t{};
	t.world = world;
	for (uint32_t r = 0; r <= world; r++)
		t.bound[r] = ix.shard_bounds[r];
	DevBuf d_owner, d_sowner, d_order, d_lennt, d_lenk, d_ntscan, d_seg, d_nt2;
	BLMX_HIP(d_owner.alloc((nsk + 1) * 4, false, s));
	BLMX_HIP(d_sowner.alloc((nsk + 1) * 4, false, s));
	BLMX_HIP(d_order.alloc((nsk + 1) * 4, false, s));
	BLMX_HIP(d_seg.alloc(17 * 8, true, s));
	if (nsk) {
		hipLaunchKernelGGL(k_sk_owner, dim3(grid_for(nsk)), dim3(kBlock), 0, s, c.skmini.as<uint32_t>(), nsk, ix.gshift, t, d_owner.as<uint32_t>());
		stable_sort_by_key_u32(sc, d_owner.as<uint32_t>(), d_sowner.as<uint32_t>(), d_order.as<uint32_t>(), nsk, 5u, s);
	}
	d_owner.release();
	hipLaunchKernelGGL(k_owner_segments, dim3(1), dim3(64), 0, s, d_sowner.as<uint32_t>(), nsk, world, d_seg.as<uint64_t>());
	BLMX_HIP(d_lennt.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(d_lenk.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(d_ntscan.alloc((nsk + 1) * 8, false, s));
	hipLaunchKernelGGL(k_sorted_lengths, dim3(grid_for(nsk + 1)), dim3(kBlock), 0, s, d_order.as<uint32_t>(), c.skpos.as<uint64_t>(),
	                   c.sklast.as<uint64_t>(), nsk, k, d_lennt.as<uint64_t>(), d_lenk.as<uint64_t>());
	BLMX_HIP(hipGetLastError());
	d_lenk.release();
	exclusive_scan_u64(sc, d_lennt.as<uint64_t>(), d_ntscan.as<uint64_t>(), nsk + 1, s);
	uint64_t seg[17] = {0};
	BLMX_HIP(hipMemcpyAsync(seg, d_seg.p, ((size_t)world + 1) * 8, hipMemcpyDeviceToHost, s));
	BLMX_HIP(hipStreamSynchronize(s));
	uint64_t words = 0;
	for (uint32_t r = 0; r <= world; r++)
		t.seg_first[r] = seg[r];
	for (uint32_t r = 0; r < world; r++) {
		const uint64_t nt = read_scalar(d_ntscan.as<uint64_t>() + seg[r + 1], s) - read_scalar(d_ntscan.as<uint64_t>() + seg[r], s);
		t.seg_word[r] = words;
		send_sk[r] = seg[r + 1] - seg[r];
		send_words[r] = (nt + 31) / 32;
		words += send_words[r];
	}
	t.seg_word[world] = words;
	BLMX_HIP(d_nt2.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(x->out_words.alloc((words + 4) * 8, true, s));
	BLMX_HIP(x->out_meta.alloc((nsk + 1) * 8, false, s));
	if (nsk) {
		hipLaunchKernelGGL(k_segment_offsets, dim3(grid_for(nsk)), dim3(kBlock), 0, s, d_ntscan.as<uint64_t>(), nsk, t, d_nt2.as<uint64_t>());
		hipLaunchKernelGGL(k_scatter_pack, dim3(grid_for(nsk)), dim3(kBlock), 0, s, c.src.as<uint64_t>(), d_order.as<uint32_t>(), c.skpos.as<uint64_t>(),
		                   d_nt2.as<uint64_t>(), d_lennt.as<uint64_t>(), nsk, x->out_words.as<unsigned long long>(), ~0ull);
		hipLaunchKernelGGL(k_sk_meta, dim3(grid_for(nsk)), dim3(kBlock), 0, s, d_order.as<uint32_t>(), c.skmini.as<uint32_t>(), d_lennt.as<uint64_t>(), nsk,
		                   x->out_meta.as<uint64_t>());
		BLMX_HIP(hipGetLastError());
	}
	BLMX_HIP(hipStreamSynchronize(s));
	// the slice itself is no longer needed
	c.src.release();
	c.skpos.release();
	c.sklast.release();
	c.skmini.release();
	c.nsk = 0;
	*d_meta = x->out_meta.p;
	*d_words = x->out_words.p;
	x->phase = 2;
}

void shard_build_finish(Index& ix, const void* d_meta_in, const void* d_words_in, const uint64_t* recv_sk, const uint64_t* recv_words)
{
	ShardExchange* x = ix.xb;
	if (!x || x->phase != 2)
		throw StatusError{BLMX_ESTATE, "blmx_shard_build_finish needs a blmx_shard_build_plan first"};
	hipStream_t s = ix.stream;
	struct Cleanup {
		Index& ix;
		~Cleanup() { shard_build_abort(ix); }
	} cleanup{ix};
	BuildPoolScope scope(&x->pool, &ix);  // (records the exchange pool's high-water mark in the stats)
	PhaseMarks mark(s);
	Scratch sc;
	sc.s = s;
	const uint32_t world = ix.shard_world;
	const int k = (int)ix.prm.k;
	OwnerTable t{};
	t.world = world;
	uint64_t nsk = 0, words = 0;
	for (uint32_t r = 0; r < world; r++) {
		t.seg_first[r] = nsk;
		t.seg_word[r] = words;
		nsk += recv_sk[r];
		words += recv_words[r];
	}
	t.seg_first[world] = nsk;
	t.seg_word[world] = words;
	if (nsk >= 0xffffffffull)
		throw StatusError{BLMX_ELIMIT, "more than 2^32 super-k-mers on one shard"};
	Chopped c;
	BLMX_HIP(c.src.alloc((words + 4) * 8, true, s));
	if (words)
		BLMX_HIP(hipMemcpyAsync(c.src.p, d_words_in, words * 8, hipMemcpyDeviceToDevice, s));
	BLMX_HIP(c.skpos.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(c.sklast.alloc((nsk + 1) * 8, false, s));
	BLMX_HIP(c.skmini.alloc((nsk + 1) * 4, false, s));
	c.nsk = nsk;
	if (nsk) {
		DevBuf d_ntscan;
		BLMX_HIP(d_ntscan.alloc((nsk + 1) * 8, false, s));
		const MetaLen lens{reinterpret_cast<const uint64_t*>(d_meta_in)};
		exclusive_scan_u64(sc, lens, d_ntscan.as<uint64_t>(), nsk, s);
		hipLaunchKernelGGL(k_sk_from_meta, dim3(grid_for(nsk)), dim3(kBlock), 0, s, reinterpret_cast<const uint64_t*>(d_meta_in), d_ntscan.as<uint64_t>(), nsk,
		                   k, t, c.skpos.as<uint64_t>(), c.sklast.as<uint64_t>(), c.skmini.as<uint32_t>());
		BLMX_HIP(hipGetLastError());
	}
	BLMX_HIP(hipStreamSynchronize(s));
	// only now: with a world of one rank a caller may hand the outgoing buffers straight back as the incoming ones
	x->out_words.release();
	x->out_meta.release();
	mark("exchange: assemble");
	x->shard.halo_later = world > 1 && ix.shard_rank + 1 < world;
	if (ix.KW == 1)
		build_tail<1>(ix, c, x->shard, mark, x->t_begin);
	else
		build_tail<2>(ix, c, x->shard, mark, x->t_begin);
	mark("counters + query heads");
}

// ---- the halo of a shard built from slices ---------------------------------------------------------------------------------------
// The reference's 2^b scan does not stop at the end of a bucket (blight.cpp:729-739): a window that starts in the last bucket of a
// shard reads on into what the SINGLE index holds next -- the first buckets of the next owner.  After every rank has finished its
// shard, each publishes the head of its packed sequence (shard_halo_get: [valid nucleotides, <= halo_length | packed words]), the
// records of all ranks are gathered, and shard_halo_set copies the heads of the following ranks behind the shard's own sequence
// (zero padding -- 'A', like the single index behind ITS end -- where nothing follows) and gives the absent-key filter the k-mers a
// window can now read across that junction.  The whole-input build (blmx_set_shard + blmx_build_from_*) computes its halo itself.
uint64_t shard_halo_words(const Index& ix) { return 1 + (halo_length(ix) + 31) / 32; }

void shard_halo_get(Index& ix, uint64_t* rec)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	const uint64_t H = halo_length(ix), nw = (H + 31) / 32;
	const uint64_t n = std::min<uint64_t>(ix.st.total_nt, H);
	std::fill(rec, rec + 1 + nw, 0ull);
	rec[0] = n;
	const uint64_t have = (n + 31) / 32;
	if (have) {
		BLMX_HIP(hipMemcpyAsync(rec + 1, ix.seq.p, have * 8, hipMemcpyDeviceToHost, ix.stream));
		BLMX_HIP(hipStreamSynchronize(ix.stream));
		if (n & 31)  // behind the shard's own nucleotides sits its halo (or padding): not part of the head
			rec[have] &= ~0ull << (64 - 2 * (n & 31));
	}
}

void shard_halo_set(Index& ix, const uint64_t* recs, uint32_t world)
{
	if (!ix.built)
		throw StatusError{BLMX_ESTATE, "index not built"};
	if (world != ix.shard_world)
		throw StatusError{BLMX_EINVAL, "halo records of another world size"};
	if (ix.halo_done)  // the last rank, a whole-input build, or a second call
		return;
	const uint64_t H = halo_length(ix), rw = shard_halo_words(ix);
	std::vector<uint64_t> halo((H + 31) / 32 + 2, 0);
	uint64_t pos = 0;
	for (uint32_t r = ix.shard_rank + 1; r < world && pos < H; r++) {
		const uint64_t* rec = recs + (size_t)r * rw;
		if (rec[0] > H)
			throw StatusError{BLMX_EINVAL, "halo record longer than the halo"};
		const uint64_t n = std::min<uint64_t>(rec[0], H - pos);
		for (uint64_t i = 0; i < n; i++, pos++) {
			const uint64_t c = (rec[1 + (i >> 5)] >> (62 - 2 * (i & 31))) & 3ull;
			halo[pos >> 5] |= c << (62 - 2 * (pos & 31));
		}
	}
	hipStream_t s = ix.stream;
	ix.filter_wait();  // the filter's own passes are done: what follows only adds to it
	if (pos) {
		DevBuf d_halo;
		BLMX_HIP(d_halo.alloc(halo.size() * 8, false, s));
		BLMX_HIP(hipMemcpyAsync(d_halo.p, halo.data(), halo.size() * 8, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(k_put_halo, dim3(grid_for(pos / 32 + 2)), dim3(kBlock), 0, s, d_halo.as<uint64_t>(), pos, ix.st.total_nt,
		                   ix.seq.as<unsigned long long>());
		BLMX_HIP(hipGetLastError());
		if (ix.filter_bpk && ix.st.local_number_kmer) {  // the end junction again, now with the halo behind it
			DeviceIndexView v = ix.view();
			v.filter = ix.fat_lines ? ix.lines.as<uint64_t>() : ix.filter.as<uint64_t>();
			const uint64_t cand = (uint64_t)(ix.prm.k - 1) + (1ull << ix.prm.bits_saved);
			if (ix.KW == 1)
				hipLaunchKernelGGL((k_filter_junctions<1>), dim3(grid_for(cand)), dim3(kBlock), 0, s, v, (const uint64_t*)nullptr, (uint64_t)0,
				                   ix.filter.as<unsigned long long>());
			else
				hipLaunchKernelGGL((k_filter_junctions<2>), dim3(grid_for(cand)), dim3(kBlock), 0, s, v, (const uint64_t*)nullptr, (uint64_t)0,
				                   ix.filter.as<unsigned long long>());
			BLMX_HIP(hipGetLastError());
		}
		BLMX_HIP(hipStreamSynchronize(s));
	}
	ix.halo_nt = pos;
	ix.halo_done = true;
}

// `positions` as the reference lays it out (fields of nbits - hint bits at its own offsets) from the wider fields in HBM
__global__ __launch_bounds__(256) void k_positions_reference_layout(const GroupDesc* __restrict__ groups, const uint64_t* __restrict__ ref_posstart,
                                                                  uint32_t hint, const uint64_t* __restrict__ phys, unsigned long long* __restrict__ out)
{
	const GroupDesc* gd = &groups[blockIdx.x];
	const uint32_t nb = gd->nbits, nl = nb - hint;
	const uint64_t n = gd->nkeys, ps = gd->posstart, ls = ref_posstart[blockIdx.x];
	for (uint64_t r = threadIdx.x; r < n; r += blockDim.x) {
		const uint64_t field = (uint64_t)(read_position_field(phys, ps, nb, r) >> hint);
		const uint64_t o = ls + r * nl;
		const uint32_t sh = (uint32_t)o & 63u;
		if (field) {
			atomicOr(&out[o >> 6], (unsigned long long)(field << sh));
			if (sh + nl > 64)
				atomicOr(&out[(o >> 6) + 1], (unsigned long long)(field >> (64 - sh)));
		}
	}
}

void positions_to_reference_layout(const Index& ix, DevBuf& out, hipStream_t s)
{
	const uint32_t ng = ix.ngroups;
	std::vector<uint64_t> ref((size_t)ng + 1, 0);
	for (uint32_t g = 0; g < ng; g++)
		ref[g + 1] = ref[g] + (uint64_t)(ix.h_groups[g].nbits - ix.pos_hint) * ix.h_groups[g].nkeys + 8;
	DevBuf d_ref;
	BLMX_HIP(d_ref.alloc(ref.size() * 8, false, s));
	BLMX_HIP(hipMemcpyAsync(d_ref.p, ref.data(), ref.size() * 8, hipMemcpyHostToDevice, s));
	BLMX_HIP(out.alloc((ref[ng] / 64 + 3) * 8, true, s));
	hipLaunchKernelGGL(k_positions_reference_layout, dim3(ng), dim3(256), 0, s, ix.groups.as<GroupDesc>(), d_ref.as<uint64_t>(), ix.pos_hint,
	                   ix.positions.as<uint64_t>(), out.as<unsigned long long>());
	BLMX_HIP(hipGetLastError());
	BLMX_HIP(hipStreamSynchronize(s));
}

void init_filter_gate(Index& ix, hipStream_t s)
{
	const uint32_t one = 1;  // consult the filter until a large batch says otherwise
	BLMX_HIP(ix.filter_gate.alloc(64, true, s));
	BLMX_HIP(hipMemcpyAsync(ix.filter_gate.p, &one, 4, hipMemcpyHostToDevice, s));
	BLMX_HIP(hipStreamSynchronize(s));
}

// One lane per fallback entry (sorted by group, key, input order): the LAST entry of a run of equal keys claims a slot of its group's
// table by its rank word (all keys that are inserted differ, so a claimed slot is somebody else's: next slot) and writes its key.
template <int KW>
__global__ void k_fb_table_insert(const uint64_t* __restrict__ fk, const uint64_t* __restrict__ frank, const uint32_t* __restrict__ fgrp, uint64_t F,
                                  const uint64_t* __restrict__ dir, unsigned long long* __restrict__ table)
{
	const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= F)
		return;
	const uint32_t g = fgrp[e];
	const Kmer<KW> x = load_kmer<KW>(fk, e);
	if (e + 1 < F && fgrp[e + 1] == g && load_kmer<KW>(fk, e + 1) == x)
		return;  // an earlier assignment of a duplicated key: overwritten in the reference's map (bbhash.h:724)
	const uint64_t d = dir[g], first = d & 0x00ffffffffffffffull, mask = (1ull << (d >> 56)) - 1ull;
	constexpr int W = FbSlot<KW>::words;
	uint64_t at = fb_mix(x.lo, kmer_hi(x)) & mask;
	for (;;) {
		unsigned long long* sl = table + (first + at) * W;
		if (atomicCAS(&sl[W == 2 ? 1 : 2], (unsigned long long)kNotFound, (unsigned long long)frank[e]) == (unsigned long long)kNotFound) {
			sl[0] = x.lo;
			if constexpr (KW == 2)
				sl[1] = x.hi;
			return;
		}
		at = (at + 1) & mask;
	}
}

void make_fallback_table(Index& ix, hipStream_t s)
{
	ix.fb_table.release();
	ix.fb_dir.release();
	const uint64_t F = ix.h_fb_group.size();
	if (F == 0 || ix.tune_fb_table == 0)
		return;
	std::vector<uint64_t> dir(ix.ngroups, 0);
	uint64_t slots = 0;
	for (uint32_t g = 0; g < ix.ngroups; g++) {
		const uint64_t c = ix.h_groups[g].fb_count;
		if (!c)
			continue;
		uint32_t lg = 1;
		while ((1ull << lg) * 2 < c * 3)  // at least 1.5 slots per entry (duplicates included: an upper bound of what is inserted)
			lg++;
		dir[g] = slots | ((uint64_t)lg << 56);
		slots += 1ull << lg;
	}
	const size_t W = ix.KW == 1 ? 2 : 4;
	BLMX_HIP(ix.fb_table.alloc(slots * W * 8 + 64, false, s));
	BLMX_HIP(hipMemsetAsync(ix.fb_table.p, 0xff, slots * W * 8 + 64, s));  // every rank word = kNotFound: empty
	BLMX_HIP(ix.fb_dir.alloc((size_t)ix.ngroups * 8, false, s));
	BLMX_HIP(hipMemcpyAsync(ix.fb_dir.p, dir.data(), (size_t)ix.ngroups * 8, hipMemcpyHostToDevice, s));
	DevBuf d_grp;
	BLMX_HIP(d_grp.alloc(F * 4, false, s));
	BLMX_HIP(hipMemcpyAsync(d_grp.p, ix.h_fb_group.data(), F * 4, hipMemcpyHostToDevice, s));
	if (ix.KW == 1)
		hipLaunchKernelGGL((k_fb_table_insert<1>), dim3(grid_for(F)), dim3(kBlock), 0, s, ix.fb_keys.as<uint64_t>(), ix.fb_ranks.as<uint64_t>(), d_grp.as<uint32_t>(), F,
		                   ix.fb_dir.as<uint64_t>(), ix.fb_table.as<unsigned long long>());
	else
		hipLaunchKernelGGL((k_fb_table_insert<2>), dim3(grid_for(F)), dim3(kBlock), 0, s, ix.fb_keys.as<uint64_t>(), ix.fb_ranks.as<uint64_t>(), d_grp.as<uint32_t>(), F,
		                   ix.fb_dir.as<uint64_t>(), ix.fb_table.as<unsigned long long>());
	BLMX_HIP(hipGetLastError());
	BLMX_HIP(hipStreamSynchronize(s));  // (dir and d_grp go out of scope)
}

void make_query_heads(Index& ix)
{
	ix.has_qheads = false;
	std::vector<QueryHead> h(ix.ngroups);
	for (uint32_t g = 0; g < ix.ngroups; g++) {
		const GroupDesc& gd = ix.h_groups[g];
		if ((gd.line_base >> 32) || (gd.size[0] >> 32) || (gd.size[1] >> 32) || (gd.size[2] >> 32) || (gd.posstart >> 56) || gd.nbits > 255)
			return;  // a group too large for the narrow fields: the kernels keep reading the GroupDesc head
		h[g].prefix = gd.prefix;
		h[g].posstart_nbits = gd.posstart | ((uint64_t)gd.nbits << 56);
		h[g].line_base = (uint32_t)gd.line_base;
		h[g].size0 = (uint32_t)gd.size[0];
		h[g].size1 = (uint32_t)gd.size[1];
		h[g].size2 = (uint32_t)gd.size[2];
	}
	BLMX_HIP(ix.qheads.alloc((size_t)ix.ngroups * sizeof(QueryHead), false, ix.stream));
	BLMX_HIP(hipMemcpyAsync(ix.qheads.p, h.data(), h.size() * sizeof(QueryHead), hipMemcpyHostToDevice, ix.stream));
	BLMX_HIP(hipStreamSynchronize(ix.stream));
	ix.has_qheads = true;
}

void build_from_device_packed(Index& ix, DevBuf& d_packed, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len)
{
	if (ix.KW == 1)
		build_impl<1>(ix, nullptr, d_offs, nseq, total_len, &d_packed);
	else
		build_impl<2>(ix, nullptr, d_offs, nseq, total_len, &d_packed);
}

void build_from_device(Index& ix, const uint8_t* d_ascii, const uint64_t* d_offs, uint64_t nseq, uint64_t total_len)
{
	if (ix.KW == 1)
		build_impl<1>(ix, d_ascii, d_offs, nseq, total_len);
	else
		build_impl<2>(ix, d_ascii, d_offs, nseq, total_len);
}

}  // namespace blmx
