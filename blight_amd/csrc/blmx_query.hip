// blmx_query.hip -- gfx950 batch query kernels of the Blight hot path.
//
// One lane answers one k-mer (reference query_kmer_hash -> query_get_hash, blight.cpp:545-550, 716-742):
//   canonical k-mer -> minimizer (kmer.h:791-810) -> bucket descriptor (16 B) -> group descriptor -> BBHash probe
//   (bbhash.h:561-577, 617-663; one 16-B line [u32 rank | 96 level bits] per probed level) -> position field
//   (blight.cpp:473-482) -> 2^b window of the packed sequence (blight.cpp:729-739) compared in registers.
// The path is a chain of ~5 dependent random accesses per k-mer (query_device.h): latency / line-rate bound, no MFMA.
#include <cstdio>
#include <cstring>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "blmx_internal.h"
#include "scan_kernels.h"
#include "index_device.h"
#include "query_device.h"

namespace blmx {

// The filter gate of the NEXT batch from a sample of THIS batch's answers (16384 of them, evenly spaced in the caller's order, read
// back after the gather): consult the filter iff at least 1 in 12 was "absent".  Counting inside the query kernel was tried first:
// device-scope atomics from eight XCDs on a few addresses are served at memory at about one per microsecond -- a counter per wave
// made the kernel 8x slower, one workgroup in 1024 still cost 2 %.  One workgroup, after the pipeline, costs nothing there.
constexpr uint32_t kGateSamples = 1u << 14;  // 16 workgroups, one sample per lane (one workgroup reading 2^17 took 0.3 ms)
__global__ __launch_bounds__(1024) void k_gate_sample(const int64_t* __restrict__ ids, uint64_t nq, uint32_t* __restrict__ gate)
{
	__shared__ uint32_t s_cnt[16];
	const uint64_t stride = nq < kGateSamples ? 1 : nq / kGateSamples, n = nq < kGateSamples ? nq : kGateSamples;
	const uint64_t t = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
	uint32_t c = (t < n && ids[t * stride] < 0) ? 1u : 0u;
	for (int o = 32; o > 0; o >>= 1)
		c += __shfl_down(c, o);
	if ((threadIdx.x & 63u) == 0)
		s_cnt[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (int w = 0; w < 16; w++)
			tot += s_cnt[w];
		if (tot)
			atomicAdd(&gate[1], tot);
	}
}
__global__ void k_gate_set(uint32_t* __restrict__ gate, uint64_t nq)
{
	const uint64_t n = nq < kGateSamples ? nq : kGateSamples;
	gate[0] = (uint64_t)gate[1] * 12 >= n ? 1u : 0u;
	gate[1] = 0;
}

template <int KW, int LOGP, bool WANT_POS>
__global__ __launch_bounds__(256) void k_query(DeviceIndexView v, const uint64_t* __restrict__ kmers,
                                               const uint32_t* __restrict__ minis, uint64_t nq, void* __restrict__ out,
                                               int mode, DoneSignal done)
{
	// kQueryXcdMajor (batch partitioned by minimizer): workgroup b runs on XCD b % 8, each XCD has its own L2.  Give XCD x
	// the x-th contiguous eighth of the batch, so that one bin's slice of the index is cached by ONE L2 instead of being
	// fetched into all eight (the grid is a multiple of 8).
	const DeviceIndexView vq = gated_view(v);
	uint32_t blk = blockIdx.x;
	const bool partitioned = (mode & kQueryXcdMajor) != 0;
	if (partitioned)
		blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
	mode &= 0xff;
	const uint64_t i = (uint64_t)blk * blockDim.x + threadIdx.x;
	if (i >= nq)
		return;
	const Kmer<KW> x = load_kmer<KW>(kmers, i);
	const Kmer<KW> xr = kmer_rc<KW>(x, v.k);
	int64_t id = -1;
	int32_t posj = -1;
	uint32_t mini;
	bool in_range = true;
	if (minis) {
		mini = minis[i];
		in_range = mini < v.nbuckets;  // the reference reads all_buckets[] out of bounds here: defined as absent
	} else {
		mini = kmer_minimizer<KW>(x, xr, v.k, v.m);
	}
	if (in_range)
		query_one<KW, LOGP, WANT_POS>(vq, x, xr, mini, id, posj);
	const uint64_t o = i;
	if (mode == kQueryIds)
		reinterpret_cast<int64_t*>(out)[o] = id;
	else if (mode == kQueryIds32)
		reinterpret_cast<uint32_t*>(out)[o] = (uint32_t)id;  // -1 -> 0xffffffff; the caller knows that every ID fits
	else if (mode == kQueryPos)
		reinterpret_cast<int32_t*>(out)[o] = posj;
	else
		reinterpret_cast<uint8_t*>(out)[o] = id >= 0 ? 1 : 0;
	if (done.flag) {  // small launches: the last wave that has stored its answers tells the host (waves without a k-mer left above)
		__threadfence_system();
		// (counter == nullptr: this wave is the whole round -- one k-mer, a few k-mers -- and needs nobody's count)
		if (lane_id() == (uint32_t)(__ffsll((unsigned long long)__ballot(1)) - 1) && (!done.counter || atomicAdd(done.counter, 1u) + 1u == done.target))
			*done.flag = done.ticket;
	}
}

unsigned int query_signal_units(uint64_t nq) { return (unsigned int)((nq + 63) / 64); }

// (An absent-key filter makes the absent half of a wave's lanes leave early; compacting the survivors through an LDS ring so that
// the query body runs with all 64 lanes busy -- per workgroup, per wave, or persistent waves striding over the batch -- was built and
// measured: 30 % fewer VALU instructions, but the kernel is bound by outstanding L1 misses, not by issue slots, and the dense body
// raises the average L2 round trip from 713 to 979 cycles: 5.2 - 6.5 ms instead of 4.0 ms at C4.  Dropped;
// profiles/r02_pmc_exp_compact_rejected_c4.json next to r02_pmc_exp_filter_perlane_c4.json.)

// Generic window for b > 8 (rare): same semantics, words fetched on demand.
template <int KW>
__global__ __launch_bounds__(256) void k_query_anyb(DeviceIndexView v, const uint64_t* __restrict__ kmers,
                                                    const uint32_t* __restrict__ minis, uint64_t nq, void* __restrict__ out,
                                                    int mode)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq)
		return;
	const Kmer<KW> x = load_kmer<KW>(kmers, i);
	const Kmer<KW> xr = kmer_rc<KW>(x, v.k);
	int64_t id = -1;
	int32_t posj = -1;
	do {
		uint32_t mini;
		if (minis) {
			mini = minis[i];
			if (mini >= v.nbuckets)
				break;
		} else {
			mini = kmer_minimizer<KW>(x, xr, v.k, v.m);
		}
		const BucketDesc bd = v.buckets[mini];
		if (bd.nt == 0)
			break;
		const GroupDesc* gd = &v.groups[mini >> v.gshift];
		const GroupHead gh = load_group_head(gd);
		const uint64_t h = mphf_lookup<KW>(v, gd, gh, x);
		if (h == kNotFound)
			break;
		const uint64_t pos = read_position(v.positions, gh.posstart, gh.nbits, h, v.hint, v.b);
		if (!(pos + (uint64_t)v.k - 1 < (uint64_t)bd.nt) || xr < x)
			break;
		const uint64_t P = 1ull << v.b;
		for (uint64_t j = 0; j < P; j++) {
			const Kmer<KW> s = packed_kmer<KW>(v.seq, bd.start + pos + j, v.k);
			if (s == x || s == xr) {
				id = (int64_t)(h + gh.prefix);
				posj = (int32_t)(pos + j);
				break;
			}
		}
	} while (0);
	const uint64_t o = i;
	if (mode == kQueryIds)
		reinterpret_cast<int64_t*>(out)[o] = id;
	else if (mode == kQueryIds32)
		reinterpret_cast<uint32_t*>(out)[o] = (uint32_t)id;
	else if (mode == kQueryPos)
		reinterpret_cast<int32_t*>(out)[o] = posj;
	else
		reinterpret_cast<uint8_t*>(out)[o] = id >= 0 ? 1 : 0;
}

template <int KW>
__global__ __launch_bounds__(256) void k_kmer_groups(int k, int m, int gshift, const uint64_t* __restrict__ kmers, uint64_t nq,
                                                     uint32_t* __restrict__ groups)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq)
		return;
	const Kmer<KW> x = load_kmer<KW>(kmers, i);
	groups[i] = kmer_minimizer<KW>(x, kmer_rc<KW>(x, k), k, m) >> gshift;
}

// canonical k-mer starting at byte position pos[i] of an ASCII buffer (str2num, kmer.h:90-98 + min_k, blight.cpp:86-91)
template <int KW>
__global__ __launch_bounds__(256) void k_kmers_at(int k, const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ pos, uint64_t nq,
                                                  uint64_t* __restrict__ out, int* __restrict__ err)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq)
		return;
	const uint64_t p = pos[i];
	Kmer<KW> x;
	x.lo = 0;
	if constexpr (KW == 2)
		x.hi = 0;
	bool bad = false;
	for (int t = 0; t < k; t++) {
		const uint32_t c = nuc_code(ascii[p + t]);
		bad |= c > 3;
		if constexpr (KW == 2)
			x.hi = (x.hi << 2) | (x.lo >> 62);
		x.lo = (x.lo << 2) | (c & 3u);
	}
	if (bad && err)
		atomicOr(err, 1);
	const Kmer<KW> c = kmer_canon<KW>(x, k);
	if constexpr (KW == 1) {
		out[i] = c.lo;
	} else {
		out[2 * i] = c.lo;
		out[2 * i + 1] = c.hi;
	}
}

template <int KW> __global__ __launch_bounds__(256) void k_canonicalize(int k, uint64_t* __restrict__ kmers, uint64_t nq)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq)
		return;
	Kmer<KW> x = load_kmer<KW>(kmers, i);
	if constexpr (KW == 1) {
		if (k < 32)
			x.lo &= (1ull << (2 * k)) - 1ull;
	} else {
		if (k <= 32) {
			x.hi = 0;
			if (k < 32)
				x.lo &= (1ull << (2 * k)) - 1ull;
		} else {
			x.hi &= (1ull << (2 * k - 64)) - 1ull;
		}
	}
	const Kmer<KW> c = kmer_canon<KW>(x, k);
	if constexpr (KW == 1) {
		kmers[i] = c.lo;
	} else {
		kmers[2 * i] = c.lo;
		kmers[2 * i + 1] = c.hi;
	}
}

// ---- multi-GPU routing --------------------------------------------------------------------------------------
// destination rank of a k-mer = the rank whose group range [bounds[r], bounds[r+1]) holds minimizer >> gshift
template <int KW>
__global__ __launch_bounds__(256) void k_route_count(int k, int m, int gshift, const uint64_t* __restrict__ kmers, uint64_t nq,
                                                     const uint32_t* __restrict__ bounds, uint32_t world, uint8_t* __restrict__ dest,
                                                     unsigned long long* __restrict__ counts)
{
	__shared__ uint32_t s_cnt[256];
	s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nq) {
		const Kmer<KW> x = load_kmer<KW>(kmers, i);
		const uint32_t g = kmer_minimizer<KW>(x, kmer_rc<KW>(x, k), k, m) >> gshift;
		uint32_t d = 0;
		for (uint32_t r = 1; r < world; r++)
			d += (g >= bounds[r]) ? 1u : 0u;
		dest[i] = (uint8_t)d;
		atomicAdd(&s_cnt[d], 1u);
	}
	__syncthreads();
	if (threadIdx.x < world && s_cnt[threadIdx.x])
		atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

template <int KW>
__global__ __launch_bounds__(256) void k_route_scatter(const uint64_t* __restrict__ kmers, const uint8_t* __restrict__ dest, uint64_t nq,
                                                       const uint64_t* __restrict__ offsets, uint32_t world, uint64_t* __restrict__ sorted,
                                                       uint32_t* __restrict__ order, unsigned long long* __restrict__ cursor)
{
	__shared__ uint32_t s_cnt[256];
	__shared__ unsigned long long s_base[256];
	s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t d = 0, local = 0;
	if (i < nq) {
		d = dest[i];
		local = atomicAdd(&s_cnt[d], 1u);
	}
	__syncthreads();
	if (threadIdx.x < world)
		s_base[threadIdx.x] = offsets[threadIdx.x] + (s_cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]) : 0ull);
	__syncthreads();
	if (i < nq) {
		const uint64_t p = s_base[d] + local;
		if constexpr (KW == 1) {
			sorted[p] = kmers[i];
		} else {
			sorted[2 * p] = kmers[2 * i];
			sorted[2 * p + 1] = kmers[2 * i + 1];
		}
		order[p] = (uint32_t)i;
	}
}

__global__ __launch_bounds__(256) void k_unpermute(const int64_t* __restrict__ vals, const uint32_t* __restrict__ order, uint64_t nq,
                                                   int64_t* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nq)
		out[order[i]] = vals[i];
}

// ---- primitive KAT kernel (tests) ------------------------------------------------------------------------------------
template <int KW>
__global__ void k_primitive(int op, const void* __restrict__ in, uint64_t n, int arg, int arg2, void* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	switch (op) {
	case 0:
		reinterpret_cast<int32_t*>(out)[i] = revhash32(reinterpret_cast<const uint32_t*>(in)[i]);
		break;
	case 1: {  // rcb(uint32_t, m), kmer.h:236-251
		const uint32_t x = reinterpret_cast<const uint32_t*>(in)[i];
		reinterpret_cast<uint32_t*>(out)[i] = (uint32_t)(rc_digits64((uint64_t)x) >> (64 - 2 * arg));
		break;
	}
	case 2: {  // ParityCanonical::canonize, kmer.h:480-482
		const uint32_t x = reinterpret_cast<const uint32_t*>(in)[i];
		const uint32_t r = (uint32_t)(rc_digits64((uint64_t)x) >> (64 - 2 * arg));
		reinterpret_cast<uint32_t*>(out)[i] = ((__popc(x) & 1) ? x : r) >> 1;
		break;
	}
	case 3: {
		const Kmer<KW> x = load_kmer<KW>(reinterpret_cast<const uint64_t*>(in), i);
		const Kmer<KW> r = kmer_rc<KW>(x, arg);
		uint64_t* o = reinterpret_cast<uint64_t*>(out);
		if constexpr (KW == 1) {
			o[i] = r.lo;
		} else {
			o[2 * i] = r.lo;
			o[2 * i + 1] = r.hi;
		}
		break;
	}
	case 4: {
		const Kmer<KW> x = load_kmer<KW>(reinterpret_cast<const uint64_t*>(in), i);
		reinterpret_cast<uint32_t*>(out)[i] = kmer_minimizer<KW>(x, kmer_rc<KW>(x, arg), arg, arg2);
		break;
	}
	case 7: {  // MurmurHash3_x64_128 of the first arg2 (0: all) bytes of record i (8 * KW bytes each), seed = arg
		uint64_t h[2];
		murmur3_x64_128(reinterpret_cast<const uint8_t*>(in) + i * 8 * KW, arg2 > 0 ? arg2 : 8 * KW, (uint32_t)arg, h);
		reinterpret_cast<uint64_t*>(out)[2 * i] = h[0];
		reinterpret_cast<uint64_t*>(out)[2 * i + 1] = h[1];
		break;
	}
	case 5: {
		LevelHasher hs(reinterpret_cast<const uint64_t*>(in)[i]);
		for (int l = 0; l < kLevels; l++)
			reinterpret_cast<uint64_t*>(out)[i * kLevels + l] = hs.next(l);
		break;
	}
	case 6:
		reinterpret_cast<uint8_t*>(out)[i] = (uint8_t)nuc_code(reinterpret_cast<const uint8_t*>(in)[i]);
		break;
	default:
		break;
	}
}

// ---- launchers -----------------------------------------------------------------------------------------------------
static inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

template <int KW>
static void launch_query_kw(const Index& ix, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, void* d_out, int mode,
                            hipStream_t s, DoneSignal done)
{
	DeviceIndexView v = ix.view();
	if (mode & kQueryNoFilter)
		v.filter = nullptr;
	mode &= ~(int)kQueryNoFilter;
	const uint64_t* km = reinterpret_cast<const uint64_t*>(d_kmers);
	unsigned grid = grid_for(nq, 256);
	if ((mode & kQueryXcdMajor) && v.b <= 8)
		grid = (grid + 7u) & ~7u;
	else
		mode &= 0xff;
#define BLMX_Q(LP)                                                                                  \
	case LP:                                                                                        \
		if ((mode & 0xff) == (int)kQueryPos)                                                        \
			hipLaunchKernelGGL((k_query<KW, LP, true>), dim3(grid), dim3(256), 0, s, v, km, d_minis, nq, d_out, mode, done);  \
		else                                                                                        \
			hipLaunchKernelGGL((k_query<KW, LP, false>), dim3(grid), dim3(256), 0, s, v, km, d_minis, nq, d_out, mode, done); \
		break;
	switch (v.b) {
		BLMX_Q(0)
		BLMX_Q(1)
		BLMX_Q(2)
		BLMX_Q(3)
		BLMX_Q(4)
		BLMX_Q(5)
		BLMX_Q(6)
		BLMX_Q(7)
		BLMX_Q(8)
	default:
		hipLaunchKernelGGL((k_query_anyb<KW>), dim3(grid), dim3(256), 0, s, v, km, d_minis, nq, d_out, mode);
		break;
	}
#undef BLMX_Q
}

void launch_query(const Index& ix, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, void* d_out, QueryMode mode,
                  hipStream_t s, DoneSignal done)
{
	if (nq == 0)
		return;
	if (done.flag && ix.prm.bits_saved > 8)
		throw StatusError{BLMX_EINVAL, "internal: completion signals need bits_to_save <= 8"};
	if (ix.KW == 1)
		launch_query_kw<1>(ix, d_kmers, d_minis, nq, d_out, (int)mode, s, done);
	else
		launch_query_kw<2>(ix, d_kmers, d_minis, nq, d_out, (int)mode, s, done);
	BLMX_HIP(hipGetLastError());
}

// ---- locality-binned batch query ---------------------------------------------------------------------------------------
// Random k-mers touch ~6 random 64-B sectors each (bucket/group descriptors, MPHF lines, position field, sequence window)
// and the kernel runs at the chip's random-sector rate.  Partitioning the batch by the top bits of the minimizer (= by
// contiguous ranges of buckets, hence of MPHF groups, positions and packed sequence) makes concurrently running blocks
// work on the same ~1/1024 slice of the index, which L2 / Infinity Cache then serve: measured 13.4 -> 7.2 ms per 1e8
// queries on the human-scale index.  One LSD counting pass (per-block LDS histograms, bin-major scan, scatter); the
// IDs are gathered back through the inverse permutation, whose 1024 streams are each (nearly) sequential.
constexpr int kBinThreads = 1024;  // 16 waves per CU: the scatter holds 108 KB of LDS, so one block per CU
constexpr int kBinBitsMax = 10;
// 512 bins: (tile, bin) runs of 16 k-mers = 128 B keep the scatter's stores and the gather's reads line-sized; with the
// XCD-major query order the finer 1024-bin partition no longer pays for its 64-B runs (C4: 7.33 ms vs 7.60 ms per step).
constexpr int kBinBitsDefault = 9;
// queries per block in the partition passes: the scatter stages a whole tile (k-mers + bin ids) in LDS
template <int KW> struct BinTile { static constexpr int value = 8192 / KW; };
// (Tiles of 16384 k-mers -- possible when the minimizers are not staged in LDS -- with 1024 bins keep the (tile, bin) runs 16
// k-mers long and make the query kernel 0.3 ms faster at C4 (4.66 -> 4.36 ms), but the gather then reads 1024 lines per tile
// instead of 512, which no longer fit the CU's 32 KB L1: k_gather_ids 0.74 -> 1.05 ms, scatter +0.1 ms -- measured and dropped,
// profiles/r02_exp_big_tile_c4.jsonl.)

// minimizer -> partition bin.  Bins are minimizer prefixes; on a sharded index every bin that contains a shard boundary
// is split there (at most 15 extra bins), so that bins stay monotone in the minimizer AND every bin belongs to exactly
// one owner rank: a batch partitioned with this map can be cut into per-owner segments without another pass.
struct BinMap {
	int shift;
	uint32_t nbound;
	uint32_t bound[15];  // bound[r] = first minimizer owned by rank r + 1 (non-decreasing)
	__host__ __device__ uint32_t bin_of(uint32_t mini) const
	{
		uint32_t b = mini >> shift;
		for (uint32_t r = 0; r < nbound; r++)
			b += mini >= bound[r] ? 1u : 0u;
		return b;
	}
};

template <int KW, int kTile>
__global__ __launch_bounds__(kBinThreads) void k_bin_hist(int k, int m, BinMap bm, uint32_t nbins, const uint64_t* __restrict__ kmers,
                                                          uint64_t nq, uint32_t* __restrict__ minibuf, uint32_t* __restrict__ hist, uint32_t nblk)
{
	__shared__ uint32_t s_hist[1 << kBinBitsMax];
	for (uint32_t b = threadIdx.x; b < nbins; b += kBinThreads)
		s_hist[b] = 0;
	__syncthreads();
	const uint64_t base = (uint64_t)blockIdx.x * kTile;
	for (int it = 0; it < kTile / kBinThreads; it++) {
		const uint64_t i = base + (uint64_t)it * kBinThreads + threadIdx.x;
		if (i < nq) {
			const Kmer<KW> x = load_kmer<KW>(kmers, i);
			const uint32_t mini = kmer_minimizer<KW>(x, kmer_rc<KW>(x, k), k, m);
			minibuf[i] = mini;  // travels with the k-mer to the query kernel: the 23 m-mer hashes are paid once
			atomicAdd(&s_hist[bm.bin_of(mini)], 1u);
		}
	}
	__syncthreads();
	for (uint32_t b = threadIdx.x; b < nbins; b += kBinThreads)
		hist[(uint64_t)b * nblk + blockIdx.x] = s_hist[b];
}

// the same histogram when the minimizers are already known (read front end): 4 B per k-mer instead of 23 m-mer hashes
template <int KW, int kTile>
__global__ __launch_bounds__(kBinThreads) void k_bin_hist_minis(BinMap bm, uint32_t nbins, const uint32_t* __restrict__ minis, uint64_t nq,
                                                                uint32_t* __restrict__ hist, uint32_t nblk)
{
	__shared__ uint32_t s_hist[1 << kBinBitsMax];
	for (uint32_t b = threadIdx.x; b < nbins; b += kBinThreads)
		s_hist[b] = 0;
	__syncthreads();
	const uint64_t base = (uint64_t)blockIdx.x * kTile;
	for (int it = 0; it < kTile / kBinThreads; it++) {
		const uint64_t i = base + (uint64_t)it * kBinThreads + threadIdx.x;
		if (i < nq)
			atomicAdd(&s_hist[bm.bin_of(minis[i])], 1u);
	}
	__syncthreads();
	for (uint32_t b = threadIdx.x; b < nbins; b += kBinThreads)
		hist[(uint64_t)b * nblk + blockIdx.x] = s_hist[b];
}

// Scatter with a tile-local sort in LDS: the tile's k-mers are first placed in bin order in LDS, then copied out so that
// each (tile, bin) run is written as consecutive 8-B elements (64-128 B per run) instead of isolated 8-B stores
// (measured: isolated stores wrote 2.8x the payload to HBM).  inv[i] = final position of query i (coalesced by i).
template <int KW, int kTile>
__global__ __launch_bounds__(kBinThreads) void k_bin_scatter(const uint64_t* __restrict__ kmers, uint64_t nq, const uint32_t* __restrict__ minibuf,
                                                             BinMap bm, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                                             uint32_t nblk, uint32_t nbins, uint64_t* __restrict__ sorted,
                                                             uint32_t* __restrict__ sorted_minis, uint32_t* __restrict__ inv)
{
	constexpr int kNB = 1 << kBinBitsMax;
	__shared__ uint64_t s_keys[kTile * KW];
	__shared__ uint32_t s_mini[kTile];
	__shared__ uint32_t s_lstart[kNB];  // first slot of each bin inside the tile
	__shared__ uint32_t s_cur[kNB];
	__shared__ uint32_t s_gstart[kNB];  // first global slot of this tile's run of each bin
	__shared__ uint32_t s_wsum[kBinThreads / 64];
	// tile-local exclusive scan of this block's bin counts (one bin per thread: wave scan, wave totals through LDS)
	static_assert(kBinThreads == (1 << kBinBitsMax), "one bin per thread");
	const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const uint32_t c0 = t < nbins ? hist[(uint64_t)t * nblk + blockIdx.x] : 0u;
	uint32_t v = c0;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t u = __shfl_up(v, o);
		if ((int)lane >= o)
			v += u;
	}
	if (lane == 63)
		s_wsum[wave] = v;
	__syncthreads();
	uint32_t wbase = 0;
	for (uint32_t w = 0; w < wave; w++)
		wbase += s_wsum[w];
	const uint32_t excl = wbase + v - c0;
	if (t < nbins) {
		s_lstart[t] = excl;
		s_cur[t] = excl;
		s_gstart[t] = offs[(uint64_t)t * nblk + blockIdx.x];
	}
	__syncthreads();
	const uint64_t base = (uint64_t)blockIdx.x * kTile;
	for (int it = 0; it < kTile / kBinThreads; it++) {
		const uint64_t i = base + (uint64_t)it * kBinThreads + t;
		if (i < nq) {
			const uint32_t mini = minibuf[i];
			const uint32_t b = bm.bin_of(mini);
			const uint32_t r = atomicAdd(&s_cur[b], 1u);
			if constexpr (KW == 1) {
				s_keys[r] = kmers[i];
			} else {
				s_keys[2 * r] = kmers[2 * i];
				s_keys[2 * r + 1] = kmers[2 * i + 1];
			}
			s_mini[r] = mini;
			inv[i] = s_gstart[b] + (r - s_lstart[b]);
		}
	}
	__syncthreads();
	const uint32_t count = (uint32_t)min((uint64_t)kTile, nq - base);
	for (uint32_t sidx = t; sidx < count; sidx += kBinThreads) {
		const uint32_t mini = s_mini[sidx];
		const uint32_t b = bm.bin_of(mini);
		const uint64_t dst = (uint64_t)s_gstart[b] + (sidx - s_lstart[b]);
		if (sorted_minis)
			sorted_minis[dst] = mini;
		if constexpr (KW == 1) {
			sorted[dst] = s_keys[sidx];
		} else {
			sorted[2 * dst] = s_keys[2 * sidx];
			sorted[2 * dst + 1] = s_keys[2 * sidx + 1];
		}
	}
}

// ---- one-pass partition into bin regions -----------------------------------------------------------------------------------
// The exact partition above costs two passes over the batch (histogram, scan, scatter) because every (tile, bin) run needs its
// final offset before anything is written.  A batch that is only going to be QUERIED does not need a dense order: bin b owns a
// region of slots (a multiple of the query block), a tile reserves the slot range of each of its runs with ONE global atomic per
// (tile, bin) on the bin's cursor, and whatever does not fit a region goes to an overflow region behind them (one more atomic per
// overflowing run).  One kernel computes the minimizers, sorts the tile in LDS and writes the runs; the query kernel walks the
// slots and skips the unused tail of every region.  Results never depend on the layout.
// Minimizer-prefix bins are far from equally loaded (99 k .. 328 k k-mers around a mean of 195 k at C4; 54 k .. 399 k for
// 63-mers): with ONE capacity for every bin 4-8 % of a batch overflows and nothing is gained (7.28 vs 7.31 ms per step).  The
// regions are therefore sized after the shares the bins had of the PREVIOUS batch (k_plan_learn* -> k_plan_make, all on the
// device); the first batch of an index goes through the exact partition and leaves its shares behind.  A batch distributed unlike
// its predecessor (tests: a few k-mers repeated a thousand times after a uniform batch) only loses locality for what overflows.
// C4: 1.63 ms for this pass instead of 0.92 + 0.04 + 0.86, and 1024 bins become affordable.
// The one-pass kernels hold up to 2^kBinBitsSlots bins: BPT bins per thread of their 1024-thread workgroups (BPT = 1: up to 1024 bins,
// the sharded exchange and 128-bit k-mers; BPT = 2: 2048 bins for 64-bit k-mers, r04 -- a bin's slice of a human-scale index is then
// 5 MB instead of 10 against the 4 MB of the XCD's L2 that serves it, and the slot kernel is bound by the requests its L2 misses send
// to the fabric).
constexpr int kBinBitsSlots = 11;
struct SlotPlan {            // lives in HBM (written by k_plan_make, read by the slot kernels): no host round trip per batch
	uint32_t nbins;
	uint32_t main_slots;     // slots of all bin regions, rounded up to a multiple of 2048 (8 XCDs x 256-slot blocks)
	uint32_t ovf_cap;        // slots of the overflow region behind them
	uint32_t ovf_blocks;     // looping blocks that answer the overflow region
	uint32_t base[(1 << kBinBitsSlots) + 1];  // first slot of every bin region (multiples of 256); base[nbins] = end of the last one
	__device__ uint64_t ovf_base() const { return main_slots; }
	__device__ uint64_t total_slots() const { return (uint64_t)main_slots + ovf_cap; }
};

// Capacities of the bin regions for a batch of nq k-mers: the share every bin had of the PREVIOUS batch (frac, learnt by k_plan_learn*)
// plus slack_pct percent plus one block, or equal shares when nothing has been learnt.  One workgroup, one thread per bin.
template <int BPT>
__global__ __launch_bounds__(1 << kBinBitsMax) void k_plan_make(const float* __restrict__ frac, uint64_t nq, uint32_t nbins, uint32_t slack_pct,
                                                              uint32_t max_main_slots, uint32_t ovf_cap, uint32_t ovf_blocks, SlotPlan* __restrict__ sp,
                                                              uint32_t* __restrict__ cursors)
{
	__shared__ uint32_t s_wsum[(1 << kBinBitsMax) / 64];
	const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	uint32_t cap[BPT], mine = 0;
#pragma unroll
	for (int j = 0; j < BPT; j++) {
		const uint32_t b = t * BPT + j;
		cap[j] = 0;
		if (b < nbins) {
			double share = frac ? (double)frac[b] * (double)nq : (double)nq / (double)nbins;
			share = share >= 0.0 ? (share <= (double)nq ? share : (double)nq) : 0.0;  // (whatever the share table holds: regions stay regions)
			const double want = share * (100.0 + (double)slack_pct) / 100.0 + 256.0;
			cap[j] = (uint32_t)(((uint64_t)want + 255u) & ~255ull);
		}
		mine += cap[j];
	}
	uint32_t v = mine;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t u = __shfl_up(v, o);
		if ((int)lane >= o)
			v += u;
	}
	if (lane == 63)
		s_wsum[wave] = v;
	__syncthreads();
	uint32_t wbase = 0;
	for (uint32_t w = 0; w < wave; w++)
		wbase += s_wsum[w];
	uint32_t at = wbase + v - mine;
#pragma unroll
	for (int j = 0; j < BPT; j++) {
		const uint32_t b = t * BPT + j;
		const uint32_t start = min(at, max_main_slots);  // what does not fit the region goes to the overflow slots
		const uint32_t end = min(start + cap[j], max_main_slots);
		if (b < nbins) {
			sp->base[b] = start;
			cursors[b] = 0;
		}
		if (b + 1 == nbins) {
			sp->base[nbins] = end;
			sp->nbins = nbins;
			sp->main_slots = (end + 2047u) & ~2047u;  // <= max_main_slots, itself a multiple of 2048
			sp->ovf_cap = ovf_cap;
			sp->ovf_blocks = ovf_blocks;
		}
		at += cap[j];
	}
	if (t == 0) {
		cursors[nbins] = 0;      // the overflow cursor
		cursors[nbins + 1] = 0;  // k-mers that found no slot (must stay 0)
	}
}
// what the capacities of the next batch are made from: this batch's k-mers per bin
// stats (may be null): [0] one-pass batches, [1] their k-mers, [2] k-mers that went through the overflow region, [3] the largest
// overflow of a batch -- what blmx_partition_stats reports (a learnt plan applied to a batch it was not learnt from overflows a little)
// [4] k-mers that found no slot at all (the overflow region holds a whole batch: always 0 unless the partition is broken); those also
// go to `dropped_host`, a word in pinned host memory that every later call on the index looks at (Index::check_dropped): a k-mer
// without a slot would otherwise come back as "absent"
__global__ void k_plan_learn(const uint32_t* __restrict__ cursors, uint64_t nq, uint32_t nbins, float* __restrict__ frac, unsigned long long* __restrict__ stats,
                             unsigned long long* __restrict__ dropped_host)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nbins && nq)
		frac[t] = (float)((double)cursors[t] / (double)nq);
	if (t == 0 && stats) {
		const unsigned long long ovf = cursors[nbins], dropped = cursors[nbins + 1];
		stats[0] += 1;
		stats[1] += nq;
		stats[2] += ovf;
		stats[3] = max(stats[3], ovf);
		stats[4] += dropped;
		if (dropped && dropped_host)
			atomicAdd_system(dropped_host, dropped);
	}
}
__global__ void k_plan_learn_offs(const uint32_t* __restrict__ offs, uint32_t nblk, uint64_t nq, uint32_t nbins, float* __restrict__ frac)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nbins) {
		const uint64_t a = offs[(uint64_t)t * nblk], z = t + 1 < nbins ? offs[(uint64_t)(t + 1) * nblk] : nq;
		frac[t] = (float)((double)(z - a) / (double)nq);
	}
}

// What the gather needs to find a query's answer again (k_gather_runs): the tile-local position of the query in bin order (lpos,
// 2 B per query) and, per (tile, bin), the run's first slot inside the bin, its first overflow slot and its length (RunTable,
// 12 B per run = 1.5 B per query) -- instead of a 4-B slot number per query that the gather then chases one lane at a time.
struct RunTable {
	uint32_t* gbase;  // [tile * kNB + bin]
	uint32_t* obase;
	uint32_t* count;
};

#ifndef BLMX_EXP_STAMP
#define BLMX_EXP_STAMP 0
#endif
#if BLMX_EXP_STAMP
// variant build only (python -m blight_amd.build --variant stamp -DBLMX_EXP_STAMP=1): shader-clock time of wave 0 of every workgroup between
// the phase boundaries of k_bin_onepass, summed over tiles; printed by the launching function every 16th batch
__device__ unsigned long long g_bin_stamps[16];
#define BLMX_STAMP(I)                                                                \
	do {                                                                             \
		if (t == 0) {                                                                \
			const unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
			atomicAdd(&g_bin_stamps[I], now_ - stamp_);                              \
			stamp_ = now_;                                                           \
		}                                                                            \
	} while (0)
#else
#define BLMX_STAMP(I) \
	do {              \
	} while (0)
#endif
constexpr int kWaitVmcnt0 = 0x0F70;  // s_waitcnt vmcnt(0) alone on gfx9: expcnt = 7, lgkmcnt = 15 left open
// MINIS: the caller hands the minimizers over (read front end), they are loaded with the k-mers.
// THREADS = 1024: one workgroup per CU (123 KB of LDS), the whole tile staged in bin order and copied out at once, the next tile's k-mers
// loaded one tile ahead.  THREADS = 512 (r04, the default for up to 1024 bins): TWO workgroups per CU, each staging its tile in two
// pieces of half a tile (72 KB of LDS each), 16 k-mers and two bins per thread: while one workgroup reserves slots (a global atomic
// round trip), scatters into LDS and issues its stores -- a third of a tile's time with nothing for the vector ALUs to do, see the
// phase stamps in profiles/r04_exp_partition_kernel_c4.txt -- the other one computes minimizers.
template <int KW, int BPT, bool MINIS, int THREADS, int ROUNDS>
__global__ __launch_bounds__(THREADS, ROUNDS * THREADS / 256) void k_bin_onepass(int k, int m, BinMap bm, const SlotPlan* __restrict__ sp, const uint64_t* __restrict__ kmers,
                                                     const uint32_t* __restrict__ minis_in, uint64_t nq, uint32_t* __restrict__ cursors,
                                                     uint64_t* __restrict__ slots, uint32_t* __restrict__ slot_minis, uint16_t* __restrict__ lpos,
                                                     RunTable rt)
{
	constexpr int kTile = BinTile<KW>::value;
	constexpr int kPer = kTile / THREADS;
	constexpr int kNB = BPT * THREADS;  // BPT bins per thread
	constexpr int kStage = kTile / ROUNDS;  // the tile is staged and copied out in ROUNDS pieces of kStage k-mers (in bin order)
	constexpr bool AHEAD = ROUNDS == 1;
	__shared__ uint64_t s_keys[kStage * KW];
	__shared__ uint32_t s_mini[kStage];
	__shared__ uint32_t s_cnt2[2][kNB];  // k-mers of this tile per bin, then the tile-local start of the bin; tiles alternate between the two
	__shared__ uint32_t s_gbase[kNB];   // first slot (inside the bin) of this tile's run
	__shared__ uint32_t s_obase[kNB];   // first overflow slot of the part of the run that does not fit the bin
	__shared__ uint32_t s_rbase[kNB];   // the bin's region: first slot ...
	__shared__ uint32_t s_rcap[kNB];    // ... and slots
	__shared__ uint32_t s_wsum[THREADS / 64];
	static_assert(kTile <= 65536 && kTile % THREADS == 0 && kTile % ROUNDS == 0, "16-bit tile-local positions");
	const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
	{
		const uint32_t nb = sp->nbins;
#pragma unroll
		for (int j = 0; j < BPT; j++) {
			const uint32_t b = t * BPT + j;
			const uint32_t rb = b < nb ? sp->base[b] : 0u;
			s_rbase[b] = rb;
			s_rcap[b] = b < nb ? sp->base[b + 1] - rb : 0u;
			s_cnt2[0][b] = 0;
		}
	}
	const uint32_t ovf_at = sp->main_slots, all_slots = sp->main_slots + sp->ovf_cap, nbins_p = sp->nbins;
	// A workgroup walks over tiles (grid = a few workgroups per CU, or one per tile): the stores of one tile drain while the minimizers
	// of the next are computed, the region table is loaded once, and there is no dispatch gap between tiles.  r04: the counters are
	// double-buffered, so that no barrier separates the copy-out of tile t from the minimizer phase of tile t + 1 (a wave that has
	// issued its stores goes on hashing): the barrier behind the minimizer phase of t + 1 is the one that says "tile t has been read".
	__syncthreads();
	// One workgroup per CU: the k-mers of a tile are loaded one tile AHEAD, all of a thread's loads at once, right behind the barrier that
	// ends the minimizer phase, and they are waited for BEFORE the copy-out issues its stores: vmcnt counts loads and stores in one queue
	// on this ISA, so a load issued behind the copy-out's stores can only be waited for together with them -- which put the whole drain
	// of a tile's 114 KB (and one load latency per k-mer) in front of the minimizer phase of the next tile.  Two workgroups per CU: all
	// of a thread's loads at once at the head of the tile (the other workgroup computes meanwhile; no registers for a second tile).
	Kmer<KW> xn[kPer];
	uint32_t mn[MINIS ? kPer : 1];
	auto load_tile = [&](uint64_t tile_) {
#pragma unroll
		for (int e = 0; e < kPer; e++) {
			const uint64_t i = tile_ * kTile + (uint64_t)e * THREADS + t;
			if (i < nq) {
				xn[e] = load_kmer<KW>(kmers, i);
				if constexpr (MINIS)
					mn[e] = minis_in[i];
			}
		}
	};
	if constexpr (AHEAD) {
		load_tile(blockIdx.x);
		__builtin_amdgcn_s_waitcnt(kWaitVmcnt0);
	}
#if BLMX_EXP_STAMP
	unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
	uint32_t par = 0;
	for (uint64_t tile = blockIdx.x; tile * kTile < nq; tile += gridDim.x, par ^= 1u) {
	uint32_t* const s_cnt = s_cnt2[par];
	const uint64_t base = tile * kTile;
	Kmer<KW> x[kPer];
	uint32_t mini[kPer], rank[kPer];
	if constexpr (!AHEAD)
		load_tile(tile);
#pragma unroll
	for (int e = 0; e < kPer; e++) {
		const uint64_t i = base + (uint64_t)e * THREADS + t;
		mini[e] = 0;
		rank[e] = 0;
		x[e] = xn[e];
		if (i < nq) {
			if constexpr (MINIS)
				mini[e] = mn[e];
			else
				mini[e] = kmer_minimizer<KW>(x[e], kmer_rc<KW>(x[e], k), k, m);
			rank[e] = atomicAdd(&s_cnt[bm.bin_of(mini[e])], 1u);
		}
	}
	BLMX_STAMP(0);  // minimizer phase
	__syncthreads();
	BLMX_STAMP(1);  // its barrier
	if constexpr (AHEAD)
		load_tile(tile + gridDim.x);
	// BPT bins per thread: tile-local exclusive scan of the counts, slot reservation in the bin and in the overflow region
	{
		uint32_t c[BPT], mine = 0;
#pragma unroll
		for (int j = 0; j < BPT; j++) {
			c[j] = s_cnt[t * BPT + j];
			mine += c[j];
		}
		uint32_t v = mine;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t u = __shfl_up(v, o);
			if ((int)lane >= o)
				v += u;
		}
		if (lane == 63)
			s_wsum[wave] = v;
#pragma unroll
		for (int j = 0; j < BPT; j++) {
			const uint32_t b = t * BPT + j;
			uint32_t g = 0, ob = 0;
			if (c[j]) {
				const uint32_t capb = s_rcap[b];
				g = atomicAdd(&cursors[b], c[j]);
				const uint32_t over = g + c[j] > capb ? g + c[j] - max(g, capb) : 0u;
				if (over)
					ob = atomicAdd(&cursors[nbins_p], over);
			}
			s_gbase[b] = g;
			s_obase[b] = ob;
			s_cnt2[par ^ 1u][b] = 0;  // the next tile's counters (last read by the copy-out of the tile before this one)
			const size_t ri = (size_t)tile * kNB + b;
			rt.gbase[ri] = g;
			rt.obase[ri] = ob;
			rt.count[ri] = c[j];
		}
		BLMX_STAMP(2);  // scan + slot reservation (global atomics)
		__syncthreads();
		BLMX_STAMP(3);
		uint32_t wbase = 0;
		for (uint32_t w = 0; w < wave; w++)
			wbase += s_wsum[w];
		uint32_t at = wbase + v - mine;  // tile-local start of this thread's first bin
#pragma unroll
		for (int j = 0; j < BPT; j++) {
			s_cnt[t * BPT + j] = at;
			at += c[j];
		}
	}
	__syncthreads();
	BLMX_STAMP(4);  // bin starts + barrier
	auto slot_of = [&](uint32_t b, uint32_t r) -> uint64_t {  // r-th k-mer of this tile's run of bin b
		const uint32_t g = s_gbase[b], gs = g + r, capb = s_rcap[b];
		return gs < capb ? (uint64_t)s_rbase[b] + gs : (uint64_t)ovf_at + s_obase[b] + (gs - max(g, capb));
	};
	// rank[] becomes the tile-local position in bin order
#pragma unroll
	for (int e = 0; e < kPer; e++) {
		const uint64_t i = base + (uint64_t)e * THREADS + t;
		if (i < nq) {
			rank[e] += s_cnt[bm.bin_of(mini[e])];
			lpos[i] = (uint16_t)rank[e];
		}
	}
	const uint32_t count = (uint32_t)min((uint64_t)kTile, nq - base);
	uint32_t dropped = 0;
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const uint32_t r0 = (uint32_t)r * kStage;
		if (r)
			__syncthreads();  // the copy-out of the piece before has read the staging area
#pragma unroll
		for (int e = 0; e < kPer; e++) {
			const uint64_t i = base + (uint64_t)e * THREADS + t;
			const uint32_t ls = rank[e] - r0;  // wraps for a k-mer of an earlier piece
			if (i < nq && ls < (uint32_t)kStage) {
				if constexpr (KW == 1) {
					s_keys[ls] = x[e].lo;
				} else {
					s_keys[2 * ls] = x[e].lo;
					s_keys[2 * ls + 1] = x[e].hi;
				}
				s_mini[ls] = mini[e];
			}
		}
		BLMX_STAMP(5);  // scatter into LDS
		__syncthreads();
		BLMX_STAMP(6);
		if constexpr (AHEAD)
			__builtin_amdgcn_s_waitcnt(kWaitVmcnt0);  // the next tile's k-mers are in their registers before the first store below is issued
		const uint32_t r1 = min(count, r0 + (uint32_t)kStage);
		for (uint32_t sidx = r0 + t; sidx < r1; sidx += THREADS) {
			const uint32_t mi = s_mini[sidx - r0];
			const uint32_t b = bm.bin_of(mi);
			const uint64_t dst = slot_of(b, sidx - s_cnt[b]);
			if (dst >= all_slots) {  // cannot happen (the overflow region holds a whole batch) -- but if it ever does it must not pass for an
				dropped++;           // absent k-mer: counted, and the call fails (k_plan_learn -> Index::check_dropped)
				continue;
			}
			if (slot_minis)
				slot_minis[dst] = mi;
			if constexpr (KW == 1) {
				slots[dst] = s_keys[sidx - r0];
			} else {
				slots[2 * dst] = s_keys[2 * (sidx - r0)];
				slots[2 * dst + 1] = s_keys[2 * (sidx - r0) + 1];
			}
		}
		BLMX_STAMP(7);  // (wait for the next tile's k-mers +) copy-out issued
	}
	if (dropped)
		atomicAdd(&cursors[nbins_p + 1], dropped);
	}
}

// The query kernel over slots: blocks [0, main_blocks) answer 256 slots of a bin each (XCD-major like k_query), the blocks
// behind them loop over the overflow region.  IDs go to ids[slot].
template <int KW, int LOGP>
__device__ __forceinline__ void query_slot(const DeviceIndexView& v, const uint64_t* __restrict__ slots, const uint32_t* __restrict__ slot_minis,
                                           void* __restrict__ ids, bool narrow, uint64_t slot)
{
	Kmer<KW> x;
	if constexpr (KW == 1 && (BLMX_EXP_NT & 1))
		x.lo = exp_load<0>(&slots[slot]);
	else
		x = load_kmer<KW>(slots, slot);
	const Kmer<KW> xr = kmer_rc<KW>(x, v.k);
	const uint32_t mini = slot_minis ? exp_load<0>(&slot_minis[slot]) : kmer_minimizer<KW>(x, xr, v.k, v.m);
	int64_t id;
	int32_t posj;
	query_one<KW, LOGP, false>(v, x, xr, mini, id, posj);
	if (narrow)
		exp_store<3>(&reinterpret_cast<uint32_t*>(ids)[slot], (uint32_t)id);
	else
		exp_store<3>(&reinterpret_cast<int64_t*>(ids)[slot], id);
}
// the grid covers the largest region the plan may describe; workgroups behind the actual one leave at once
template <int KW, int LOGP>
__global__ __launch_bounds__(256) void k_query_slots(DeviceIndexView v, const SlotPlan* __restrict__ sp, const uint32_t* __restrict__ cursors,
                                                     const uint64_t* __restrict__ slots, const uint32_t* __restrict__ slot_minis, void* __restrict__ ids,
                                                     int narrow)
{
	const uint32_t mb = sp->main_slots >> 8;  // a multiple of 8
	if (blockIdx.x >= mb)
		return;
	const DeviceIndexView vq = gated_view(v);
	const uint32_t blk = (blockIdx.x & 7u) * (mb >> 3) + (blockIdx.x >> 3);
	const uint32_t first = blk << 8;
	uint32_t lo = 0, hi = sp->nbins;  // the bin whose region holds this block: base[lo] <= first < base[lo + 1] (uniform: scalar loads)
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (sp->base[mid] <= first)
			lo = mid;
		else
			hi = mid;
	}
	const uint32_t rb = sp->base[lo], cap = sp->base[lo + 1] - rb;
	const uint32_t used = min(cursors[lo], cap);
	const uint32_t slot = first + threadIdx.x;
	if (slot < rb + used)
		query_slot<KW, LOGP>(vq, slots, slot_minis, ids, narrow != 0, slot);
}

// The overflow region (usually empty) in its own launch: a fixed number of blocks loop over it.  (Looping blocks inside
// k_query_slots made the register allocator keep 256 VGPRs for 128-bit k-mers, a call 70-98: the common path stays loop-free.)
template <int KW, int LOGP>
__global__ __launch_bounds__(256) void k_query_slots_overflow(DeviceIndexView v, const SlotPlan* __restrict__ sp, const uint32_t* __restrict__ cursors,
                                                              const uint64_t* __restrict__ slots, const uint32_t* __restrict__ slot_minis,
                                                              void* __restrict__ ids, int narrow)
{
	const uint64_t end = sp->ovf_base() + min(cursors[sp->nbins], sp->ovf_cap);
	const DeviceIndexView vq = gated_view(v);
#pragma nounroll
	for (uint64_t slot = sp->ovf_base() + (uint64_t)blockIdx.x * 256 + threadIdx.x; slot < end; slot += (uint64_t)gridDim.x * 256)
		query_slot<KW, LOGP>(vq, slots, slot_minis, ids, narrow != 0, slot);
}

// out[i] = ids_sorted[inv[i]].  One block per partition TILE: the (tile, bin) runs written by k_bin_scatter are consumed by
// the block that covers the same tile, so each 64-B line of ids_sorted is fetched once by one CU instead of by many.
// NARROW: the ordered IDs are 32-bit words (0xffffffff = absent), written so by the query kernel when every ID fits
template <int TILE, bool NARROW>
__global__ __launch_bounds__(kBinThreads) void k_gather_ids(const void* __restrict__ ids_sorted, const uint32_t* __restrict__ inv, uint64_t nq,
                                                            int64_t* __restrict__ out)
{
	constexpr int kPer = TILE / kBinThreads;
	const uint64_t base = (uint64_t)blockIdx.x * TILE;
	// all positions first, then all IDs: kPer independent random reads in flight per lane instead of one dependent pair at a time
	uint32_t j[kPer];
#pragma unroll
	for (int it = 0; it < kPer; it++) {
		const uint64_t i = base + (uint64_t)it * kBinThreads + threadIdx.x;
		j[it] = i < nq ? inv[i] : 0xffffffffu;
	}
	int64_t v[kPer];
#pragma unroll
	for (int it = 0; it < kPer; it++) {
		v[it] = -1;
		if (j[it] != 0xffffffffu) {  // (an unplaced k-mer cannot happen: the overflow region of the one-pass partition holds a whole batch)
			if constexpr (NARROW) {
				const uint32_t w = reinterpret_cast<const uint32_t*>(ids_sorted)[j[it]];
				v[it] = w == 0xffffffffu ? -1 : (int64_t)w;
			} else {
				v[it] = reinterpret_cast<const int64_t*>(ids_sorted)[j[it]];
			}
		}
	}
#pragma unroll
	for (int it = 0; it < kPer; it++) {
		const uint64_t i = base + (uint64_t)it * kBinThreads + threadIdx.x;
		if (i < nq)
			out[i] = v[it];
	}
}

// The gather behind the one-pass partition.  One workgroup per partition tile: one thread per bin expands the tile's run table into
// the slot of every tile-local sorted position (LDS), the lanes then fetch the IDs of CONSECUTIVE sorted positions -- consecutive
// slots inside a run: 64 lanes touch ~8 lines instead of 64 -- into LDS, and out[i] = ids_tile[lpos[i]] goes out fully coalesced.
// (k_gather_ids chased a 4-B slot number per query with one random 4-B load per lane: 0.76 ms per 1e8 at C4, request-bound.)
template <int TILE, bool NARROW, int BPT>
__global__ __launch_bounds__(kBinThreads) void k_gather_runs(const SlotPlan* __restrict__ sp, RunTable rt, const uint16_t* __restrict__ lpos,
                                                             const void* __restrict__ ids, uint64_t nq, int64_t* __restrict__ out)
{
	constexpr int kNB = BPT << kBinBitsMax;
	constexpr int kPer = TILE / kBinThreads;
	__shared__ uint32_t s_slot[TILE];  // slot of every sorted position, then (NARROW) its ID
	__shared__ uint32_t s_hi[NARROW ? 1 : TILE];
	__shared__ uint32_t s_wsum[kBinThreads / 64];
	const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const uint64_t base = (uint64_t)blockIdx.x * TILE;
	const uint32_t count = (uint32_t)min((uint64_t)TILE, nq - base);
	const uint32_t nb = sp->nbins, ovf_at = sp->main_slots, all_slots = sp->main_slots + sp->ovf_cap;
	// tile-local start of every bin's run (BPT bins per thread)
	uint32_t c[BPT], mine = 0;
#pragma unroll
	for (int j = 0; j < BPT; j++) {
		const uint32_t b = t * BPT + j;
		c[j] = b < nb ? rt.count[(size_t)blockIdx.x * kNB + b] : 0u;
		mine += c[j];
	}
	uint32_t v = mine;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t u = __shfl_up(v, o);
		if ((int)lane >= o)
			v += u;
	}
	if (lane == 63)
		s_wsum[wave] = v;
	__syncthreads();
	uint32_t wbase = 0;
	for (uint32_t w = 0; w < wave; w++)
		wbase += s_wsum[w];
	uint32_t lstart = wbase + v - mine;
#pragma unroll
	for (int j = 0; j < BPT; j++) {
		const uint32_t b = t * BPT + j;
		if (c[j]) {
			const size_t ri = (size_t)blockIdx.x * kNB + b;
			const uint32_t g = rt.gbase[ri], ob = rt.obase[ri], rb = sp->base[b], capb = sp->base[b + 1] - rb;
			for (uint32_t r = 0; r < c[j]; r++) {
				const uint32_t gs = g + r;
				const uint64_t sl = gs < capb ? (uint64_t)rb + gs : (uint64_t)ovf_at + ob + (gs - max(g, capb));
				s_slot[lstart + r] = sl < all_slots ? (uint32_t)sl : 0xffffffffu;
			}
		}
		lstart += c[j];
	}
	__syncthreads();
	for (uint32_t sidx = t; sidx < count; sidx += kBinThreads) {
		const uint32_t sl = s_slot[sidx];
		if constexpr (NARROW) {
			s_slot[sidx] = sl == 0xffffffffu ? 0xffffffffu : reinterpret_cast<const uint32_t*>(ids)[sl];
		} else {
			const int64_t w = sl == 0xffffffffu ? -1 : reinterpret_cast<const int64_t*>(ids)[sl];
			s_slot[sidx] = (uint32_t)w;
			s_hi[sidx] = (uint32_t)((uint64_t)w >> 32);
		}
	}
	__syncthreads();
#pragma unroll
	for (int it = 0; it < kPer; it++) {
		const uint64_t i = base + (uint64_t)it * kBinThreads + t;
		if (i < nq) {
			const uint32_t p = lpos[i];
			if constexpr (NARROW) {
				const uint32_t w = s_slot[p];
				out[i] = w == 0xffffffffu ? -1 : (int64_t)w;
			} else {
				out[i] = (int64_t)((uint64_t)s_slot[p] | ((uint64_t)s_hi[p] << 32));
			}
		}
	}
}

static void ensure_buf(DevBuf& b, size_t bytes, hipStream_t s)
{
	if (bytes > b.bytes)
		BLMX_HIP(b.alloc(bytes + bytes / 8 + 256, false, s));
}

// per-owner k-mer counts of a partitioned batch: owner r holds bins [first_bin[r], first_bin[r + 1])
struct OwnerBins {
	uint32_t fb[18];
};
__global__ void k_rank_counts(const uint32_t* __restrict__ offs, uint32_t nblk, uint32_t nbins, uint64_t nq, uint32_t world, OwnerBins first_bin,
                              uint64_t* __restrict__ counts)
{
	const uint32_t r = threadIdx.x;
	if (r >= world)
		return;
	const uint32_t b0 = first_bin.fb[r], b1 = first_bin.fb[r + 1];
	const uint64_t a = b0 >= nbins ? nq : offs[(uint64_t)b0 * nblk], z = b1 >= nbins ? nq : offs[(uint64_t)b1 * nblk];
	counts[r] = z - a;
}

namespace {

struct PartitionPlan {
	BinMap bm;
	uint32_t nbins, nblk, tile;
	bool handoff;
};

// scratch of the partition passes is shared by all calls on this index: order this pipeline after the previous one when
// it was enqueued on a different stream (caller holds ix.bin_mu)
void partition_begin(Index& ix, hipStream_t s)
{
	if (!ix.bin_done)
		BLMX_HIP(hipEventCreateWithFlags(&ix.bin_done, hipEventDisableTiming));
	if (ix.bin_used && ix.bin_last_stream != s)
		BLMX_HIP(hipStreamWaitEvent(s, ix.bin_done, 0));
}
void partition_end(Index& ix, hipStream_t s)
{
	BLMX_HIP(hipEventRecord(ix.bin_done, s));
	ix.bin_last_stream = s;
	ix.bin_used = true;
}

// k_bin_hist -> scan -> k_bin_scatter.  d_sorted / d_inv are the outputs; d_sorted_minis may be null.
template <int KW, int TILE>
void partition_launch(Index& ix, const PartitionPlan& pl, const uint64_t* km, const uint32_t* d_minis_in, uint64_t nq, void* d_sorted, uint32_t* d_sorted_minis,
                      uint32_t* d_inv, hipStream_t s)
{
	const uint32_t nblk = pl.nblk, nbins = pl.nbins;
	const size_t nh = (size_t)nbins * nblk;
	const uint32_t* minibuf = d_minis_in ? d_minis_in : ix.b_bins.as<uint32_t>();
	if (d_minis_in)
		hipLaunchKernelGGL((k_bin_hist_minis<KW, TILE>), dim3(nblk), dim3(kBinThreads), 0, s, pl.bm, nbins, d_minis_in, nq, ix.b_hist.as<uint32_t>(), nblk);
	else
		hipLaunchKernelGGL((k_bin_hist<KW, TILE>), dim3(nblk), dim3(kBinThreads), 0, s, (int)ix.prm.k, (int)ix.prm.m, pl.bm, nbins, km, nq,
		                   ix.b_bins.as<uint32_t>(), ix.b_hist.as<uint32_t>(), nblk);
	BLMX_HIP(hipGetLastError());
	ix.prof_mark(2, s);
	ensure_buf(ix.b_tmp, scan_sums_words(nh) * 4, s);
	exclusive_scan_u32(ix.b_hist.as<uint32_t>(), ix.b_offs.as<uint32_t>(), nh, ix.b_tmp.as<uint32_t>(), s);
	ix.prof_mark(3, s);
	hipLaunchKernelGGL((k_bin_scatter<KW, TILE>), dim3(nblk), dim3(kBinThreads), 0, s, km, nq, minibuf, pl.bm, ix.b_hist.as<uint32_t>(),
	                   ix.b_offs.as<uint32_t>(), nblk, nbins, reinterpret_cast<uint64_t*>(d_sorted), d_sorted_minis, d_inv);
	BLMX_HIP(hipGetLastError());
	ix.prof_mark(4, s);
}

PartitionPlan partition_batch(Index& ix, const void* d_kmers, const uint32_t* d_minis_in, uint64_t nq, bool owner_bins, void* d_sorted,
                              uint32_t* d_sorted_minis, uint32_t* d_inv, hipStream_t s, int default_bits = kBinBitsDefault)
{
	const int mbits = 2 * (int)ix.prm.m - 1;
	PartitionPlan pl;
	pl.handoff = d_sorted_minis != nullptr;
	const int want_bits = ix.tune_bin_bits > 0 ? std::min(ix.tune_bin_bits, kBinBitsMax) : default_bits;
	int tbits = mbits < want_bits ? mbits : want_bits;
	const uint32_t nbound = (owner_bins && ix.shard_world > 1) ? ix.shard_world - 1 : 0;
	while (tbits > 1 && (1u << tbits) + nbound > (1u << kBinBitsMax))  // the kernels hold at most 2^kBinBitsMax bins
		tbits--;
	pl.bm.shift = mbits - tbits;
	pl.bm.nbound = 0;
	if (owner_bins && ix.shard_world > 1) {  // split the bins at the shard boundaries (first minimizer of each owner)
		if (ix.shard_world > 16)
			throw StatusError{BLMX_ELIMIT, "at most 16 shards in the partitioned exchange"};
		for (uint32_t r = 1; r < ix.shard_world; r++)
			pl.bm.bound[pl.bm.nbound++] = ix.shard_bounds[r] << ix.gshift;
	}
	pl.nbins = (1u << tbits) + pl.bm.nbound;
	pl.tile = (uint32_t)(BinTile<1>::value / ix.KW);
	pl.nblk = (uint32_t)((nq + pl.tile - 1) / pl.tile);
	const size_t nh = (size_t)pl.nbins * pl.nblk;
	if (!d_minis_in)
		ensure_buf(ix.b_bins, nq * 4, s);
	ensure_buf(ix.b_hist, nh * 4, s);
	ensure_buf(ix.b_offs, nh * 4, s);
	const uint64_t* km = reinterpret_cast<const uint64_t*>(d_kmers);
	if (ix.KW == 1)
		partition_launch<1, BinTile<1>::value>(ix, pl, km, d_minis_in, nq, d_sorted, d_sorted_minis, d_inv, s);
	else
		partition_launch<2, BinTile<2>::value>(ix, pl, km, d_minis_in, nq, d_sorted, d_sorted_minis, d_inv, s);
	return pl;
}

// out[i] = ids_ordered[inv[i]]; `tile` = the partition tile the order was made with (a gather block covers one tile, so the runs
// a scatter block wrote are read back by one CU); narrow = the ordered IDs are 32-bit words
void gather_ids(const Index& ix, const void* d_ids_sorted, const uint32_t* d_inv, uint64_t nq, int64_t* d_out, uint32_t tile, bool narrow, hipStream_t s)
{
	const uint32_t nblk = (uint32_t)((nq + tile - 1) / tile);
#define BLMX_G(T)                                                                                                                                  \
	if (tile == (uint32_t)(T)) {                                                                                                                   \
		if (narrow)                                                                                                                                \
			hipLaunchKernelGGL((k_gather_ids<T, true>), dim3(nblk), dim3(kBinThreads), 0, s, d_ids_sorted, d_inv, nq, d_out);                    \
		else                                                                                                                                       \
			hipLaunchKernelGGL((k_gather_ids<T, false>), dim3(nblk), dim3(kBinThreads), 0, s, d_ids_sorted, d_inv, nq, d_out);                   \
	}
	BLMX_G(BinTile<1>::value)
	else BLMX_G(BinTile<2>::value)
	else throw StatusError{BLMX_EINVAL, "internal: unknown partition tile"};
#undef BLMX_G
	BLMX_HIP(hipGetLastError());
}

}  // namespace

bool query_batch_worth_binning(const Index& ix, uint64_t nq)
{
	int policy = ix.bin_policy;
	const int mbits = 2 * (int)ix.prm.m - 1;
	if (nq >= 0xffffffffull || policy < 0)
		return false;
	if (policy > 0)
		return true;
	return nq >= (1ull << 20) && ix.st.device_bytes >= (64ull << 20) && mbits >= 4;
}

static void update_filter_gate(const Index& ix, const int64_t* d_ids, uint64_t nq, hipStream_t s)
{
	const DeviceIndexView v = ix.view();
	if (v.filter_gate && nq) {
		hipLaunchKernelGGL(k_gate_sample, dim3(kGateSamples / 1024), dim3(1024), 0, s, d_ids, nq, v.filter_gate);
		hipLaunchKernelGGL(k_gate_set, dim3(1), dim3(1), 0, s, v.filter_gate, nq);
	}
}

template <int KW>
static void launch_query_slots_kw(const Index& ix, const SlotPlan* d_plan, uint32_t max_main_blocks, uint32_t ovf_blocks, const uint32_t* d_cursors,
                                  const void* d_slots, const uint32_t* d_slot_minis, void* d_ids, bool narrow, hipStream_t s)
{
	const DeviceIndexView v = ix.view();
	const uint64_t* sl = reinterpret_cast<const uint64_t*>(d_slots);
#define BLMX_QS(LP)                                                                                                                                    \
	case LP:                                                                                                                                           \
		hipLaunchKernelGGL((k_query_slots<KW, LP>), dim3(max_main_blocks), dim3(256), 0, s, v, d_plan, d_cursors, sl, d_slot_minis, d_ids, (int)narrow); \
		hipLaunchKernelGGL((k_query_slots_overflow<KW, LP>), dim3(ovf_blocks), dim3(256), 0, s, v, d_plan, d_cursors, sl, d_slot_minis, d_ids, (int)narrow); \
		break;
	switch (v.b) {
		BLMX_QS(0)
		BLMX_QS(1)
		BLMX_QS(2)
		BLMX_QS(3)
		BLMX_QS(4)
		BLMX_QS(5)
		BLMX_QS(6)
		BLMX_QS(7)
		BLMX_QS(8)
	default:
		throw StatusError{BLMX_EINVAL, "internal: the slot pipeline needs bits_to_save <= 8"};
	}
#undef BLMX_QS
}

// workgroups of k_bin_onepass: three per compute unit that walk over the tiles (one per tile: 1.55 ms per 1e8 at C4; 512 .. 2048
// walking workgroups: 1.45 - 1.48 ms, profiles/r03_exp_onepass_grid_c4.txt); BLIGHT_ONEPASS_GRID=n overrides (0 = one per tile)
static uint32_t onepass_grid(uint32_t nblk, int resident = 1)
{
	static const int cus = [] {
		int dev = 0;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
			return (int)prop.multiProcessorCount;
		return 256;
	}();
	static const int env = [] {
		const char* e = getenv("BLIGHT_ONEPASS_GRID");
		return e ? atoi(e) : -1;
	}();
	const int n = env >= 0 ? env : 3 * cus * resident;  // `resident` workgroups fit a CU
	return n > 0 ? std::min<uint32_t>(nblk, (uint32_t)n) : nblk;
}

// 1024 bins for the one-pass partition (and for the exact pass in front of it, which it learns from): its runs are written region by
// region, and with the histogram pass gone the query kernel's gain from finer bins (3.9 -> 3.5 ms at C4) is no longer eaten by the
// passes around it (C4: 6.14 -> 5.73 ms per step)
constexpr int kBinBitsOnepass = 10;
// 2048 bins (bin_bits = 11, 64-bit k-mers, two bins per thread in the one-pass kernels; r04) are there to be asked for: measured at C4
// the slot kernel gains 2.5 % (3.35 -> 3.26 ms) and the passes around it lose more (partition 1.41 -> 1.58, gather 0.47 -> 0.56 ms: runs
// of four k-mers), profiles/r04_exp_2048_bins_c4.txt.
static int onepass_bits(const Index& ix)
{
	const int mbits = 2 * (int)ix.prm.m - 1;
	const int most = ix.KW == 1 ? kBinBitsSlots : kBinBitsMax;
	const int want_bits = ix.tune_bin_bits > 0 ? std::min(ix.tune_bin_bits, most) : kBinBitsOnepass;
	return std::max(3, std::min(mbits, want_bits));
}

// partition into bin regions whose capacities follow the previous batch (one pass, no histogram) -> query the slots -> gather the
// IDs into the caller's order -> learn this batch's shares.  Everything stays on the device: no host round trip.
static void query_ids_onepass(Index& ix, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, int64_t* d_out, bool handoff, hipStream_t s)
{
	const int mbits = 2 * (int)ix.prm.m - 1;
	const int tbits = onepass_bits(ix);
	BinMap bm;
	bm.shift = mbits - tbits;
	bm.nbound = 0;
	const uint32_t nbins = 1u << tbits;
	const bool learnt = ix.plan_bits == tbits;
	// slack on top of the learnt shares (a few percent: consecutive batches of one workload differ by sampling noise) or, with nothing
	// learnt, on top of equal shares (minimizer bins are uneven: whatever does not fit goes through the overflow region)
	const uint32_t pct = ix.tune_slot_slack_pct > 0 ? (uint32_t)ix.tune_slot_slack_pct : (learnt ? 6u : 12u);
	const uint64_t max_main = ((nq + nq * pct / 100 + (uint64_t)nbins * 512 + 2047) / 2048) * 2048;
	const uint64_t slots = max_main + nq;
	if (slots >= 0xffffffffull)
		throw StatusError{BLMX_ELIMIT, "a partitioned batch holds fewer than 2^31 k-mers"};
	const uint32_t tile = (uint32_t)(ix.KW == 1 ? BinTile<1>::value : BinTile<2>::value);
	const uint32_t nblk = (uint32_t)((nq + tile - 1) / tile);
	const bool narrow = ix.tune_narrow_ids && ix.st.number_kmer < 0xffffffffull;
	ensure_buf(ix.b_sorted, slots * 8 * (size_t)ix.KW, s);
	// lpos (2 B per query) followed by the run table (3 x 4 B per (tile, bin))
	const int bpt = tbits > kBinBitsMax ? 2 : 1;  // bins per thread of the one-pass kernels
	const size_t lpos_bytes = (nq * 2 + 255) & ~(size_t)255, runs = ((size_t)nblk << kBinBitsMax) * bpt;
	ensure_buf(ix.b_inv, lpos_bytes + runs * 12, s);
	uint16_t* d_lpos = ix.b_inv.as<uint16_t>();
	RunTable rt;
	rt.gbase = reinterpret_cast<uint32_t*>(ix.b_inv.as<char>() + lpos_bytes);
	rt.obase = rt.gbase + runs;
	rt.count = rt.obase + runs;
	ensure_buf(ix.b_ids, slots * 8, s);
	ensure_buf(ix.b_offs, ((size_t)nbins + 2) * 4, s);
	ensure_buf(ix.b_plan, sizeof(SlotPlan), s);
	ensure_buf(ix.b_frac, sizeof(float) << kBinBitsSlots, s);
	if (handoff)
		ensure_buf(ix.b_minis, slots * 4, s);
	uint32_t* d_slot_minis = handoff ? ix.b_minis.as<uint32_t>() : nullptr;
	uint32_t* d_cursors = ix.b_offs.as<uint32_t>();
	const SlotPlan* d_plan = ix.b_plan.as<SlotPlan>();
	const uint32_t ovf_blocks = 1024;
	const uint32_t ovf_cap = ix.tune_test_ovf_cap > 0 ? std::min<uint32_t>((uint32_t)ix.tune_test_ovf_cap, (uint32_t)nq) : (uint32_t)nq;
	if (bpt == 2)
		hipLaunchKernelGGL(k_plan_make<2>, dim3(1), dim3(1 << kBinBitsMax), 0, s, learnt ? ix.b_frac.as<float>() : nullptr, nq, nbins, pct, (uint32_t)max_main,
		                   ovf_cap, ovf_blocks, ix.b_plan.as<SlotPlan>(), d_cursors);
	else
		hipLaunchKernelGGL(k_plan_make<1>, dim3(1), dim3(1 << kBinBitsMax), 0, s, learnt ? ix.b_frac.as<float>() : nullptr, nq, nbins, pct, (uint32_t)max_main,
		                   ovf_cap, ovf_blocks, ix.b_plan.as<SlotPlan>(), d_cursors);
	const uint64_t* km = reinterpret_cast<const uint64_t*>(d_kmers);
	ix.prof_mark(2, s);
	ix.prof_mark(3, s);
#define BLMX_OP1(KWV, BPTV, MV, TV, RV)                                                                                                                        \
	hipLaunchKernelGGL((k_bin_onepass<KWV, BPTV, MV, TV, RV>), dim3(onepass_grid(nblk, RV)), dim3(TV), 0, s, (int)ix.prm.k, (int)ix.prm.m, bm, d_plan, km, d_minis, nq, \
	                   d_cursors, ix.b_sorted.as<uint64_t>(), d_slot_minis, d_lpos, rt)
#define BLMX_OP(KWV, BPTV, TV, RV)              \
	do {                                        \
		if (d_minis)                            \
			BLMX_OP1(KWV, BPTV, true, TV, RV);  \
		else                                    \
			BLMX_OP1(KWV, BPTV, false, TV, RV); \
	} while (0)
	const bool two_wg = ix.tune_bin_rounds != 1;  // two workgroups per CU staging half a tile at a time (default) or one staging all of it
	if (ix.KW == 1) {
		if (bpt == 2)
			BLMX_OP(1, 2, 1024, 1);
		else if (two_wg)
			BLMX_OP(1, 1, 1024, 2);
		else
			BLMX_OP(1, 1, 1024, 1);
	} else if (two_wg) {
		BLMX_OP(2, 1, 1024, 2);
	} else {
		BLMX_OP(2, 1, 1024, 1);
	}
#undef BLMX_OP
#undef BLMX_OP1
#if BLMX_EXP_STAMP
	{
		static int calls = 0;
		if (++calls % 16 == 0) {
			unsigned long long h[16];
			(void)hipStreamSynchronize(s);
			(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bin_stamps), sizeof h);
			fprintf(stderr, "[bin stamps, 16 batches, cycles of wave 0 summed over workgroups and tiles, 100 MHz ticks]");
			for (int i = 0; i < 8; i++)
				fprintf(stderr, " %d:%llu", i, h[i]);
			fprintf(stderr, "\n");
			memset(h, 0, sizeof h);
			(void)hipMemcpyToSymbol(HIP_SYMBOL(g_bin_stamps), h, sizeof h);
		}
	}
#endif
	BLMX_HIP(hipGetLastError());
	ix.prof_mark(4, s);
	if (ix.KW == 1)
		launch_query_slots_kw<1>(ix, d_plan, (uint32_t)(max_main >> 8), ovf_blocks, d_cursors, ix.b_sorted.p, d_slot_minis, ix.b_ids.p, narrow, s);
	else
		launch_query_slots_kw<2>(ix, d_plan, (uint32_t)(max_main >> 8), ovf_blocks, d_cursors, ix.b_sorted.p, d_slot_minis, ix.b_ids.p, narrow, s);
	BLMX_HIP(hipGetLastError());
	ix.prof_mark(5, s);
#define BLMX_GR(KWV, NARROWV, BPTV)                                                                                                                    \
	hipLaunchKernelGGL((k_gather_runs<BinTile<KWV>::value, NARROWV, BPTV>), dim3(nblk), dim3(kBinThreads), 0, s, d_plan, rt, d_lpos, ix.b_ids.p, nq, d_out)
	if (ix.KW == 1) {
		if (bpt == 2) {
			if (narrow)
				BLMX_GR(1, true, 2);
			else
				BLMX_GR(1, false, 2);
		} else {
			if (narrow)
				BLMX_GR(1, true, 1);
			else
				BLMX_GR(1, false, 1);
		}
	} else {
		if (narrow)
			BLMX_GR(2, true, 1);
		else
			BLMX_GR(2, false, 1);
	}
#undef BLMX_GR
	BLMX_HIP(hipGetLastError());
	ix.prof_mark(6, s);
	if (!ix.b_pstats.p)
		BLMX_HIP(ix.b_pstats.alloc(64, true, s));
	hipLaunchKernelGGL(k_plan_learn, dim3((nbins + 255) / 256), dim3(256), 0, s, d_cursors, nq, nbins, ix.b_frac.as<float>(), ix.b_pstats.as<unsigned long long>(),
	                   ix.dropped_word());
	ix.plan_bits = tbits;
	update_filter_gate(ix, d_out, nq, s);
	if (getenv("BLIGHT_TRACE") && atoi(getenv("BLIGHT_TRACE")) == 2) {  // bin occupancy of this batch (synchronises)
		std::vector<uint32_t> h(nbins + 1);
		SlotPlan hp;
		BLMX_HIP(hipMemcpyAsync(h.data(), d_cursors, h.size() * 4, hipMemcpyDeviceToHost, s));
		BLMX_HIP(hipMemcpyAsync(&hp, d_plan, sizeof hp, hipMemcpyDeviceToHost, s));
		BLMX_HIP(hipStreamSynchronize(s));
		uint32_t mx = 0, mn = ~0u;
		for (uint32_t b2 = 0; b2 < nbins; b2++) {
			mx = std::max(mx, h[b2]);
			mn = std::min(mn, h[b2]);
		}
		fprintf(stderr, "[blmx partition] nq %llu bins %u learnt %d main slots %u (bound %llu) min %u max %u overflow %u\n", (unsigned long long)nq, nbins,
		        (int)learnt, hp.main_slots, (unsigned long long)max_main, mn, mx, h[nbins]);
	}
}

// ---- the one-pass partition as the sender's side of the sharded exchange (SlotPiece, blmx_internal.h) -------------------------------
// pieces below this size go through the exact partition and travel densely: the regions' slack (256 slots per bin and more) would
// outweigh them
constexpr uint64_t kMinSlotPiece = 1ull << 19;

bool slot_piece_possible(const Index& ix, uint64_t n)
{
	const int mbits = 2 * (int)ix.prm.m - 1;
	const uint64_t least = ix.tune_slot_piece_min > 0 ? (uint64_t)ix.tune_slot_piece_min : kMinSlotPiece;
	return ix.tune_partition != 2 && ix.prm.bits_saved <= 8 && mbits >= 4 && n >= least && n < 1900000000ull && ix.plan_nbins_sharded != 0 &&
	       ix.shard_world <= 16;
}

struct FirstBins {
	uint32_t world;
	uint32_t fb[17];
};
// header words of every bin + what the host needs before the flows can be sized (one workgroup, one thread per bin)
__global__ __launch_bounds__(1 << kBinBitsMax) void k_slot_info(const SlotPlan* __restrict__ sp, const uint32_t* __restrict__ cursors, FirstBins f,
                                                              uint64_t* __restrict__ hdr, uint32_t* __restrict__ info)
{
	const uint32_t t = threadIdx.x, nb = sp->nbins;
	if (t < nb) {
		const uint32_t rb = sp->base[t], cap = sp->base[t + 1] - rb;
		hdr[t] = (uint64_t)rb | ((uint64_t)min(cursors[t], cap) << 32);
	}
	if (t == 0) {
		hdr[nb] = sp->base[nb];
		info[0] = cursors[nb];
	}
	if (t <= f.world)
		info[1 + t] = sp->base[min(f.fb[t], nb)];
}

static BinMap sharded_bin_map(const Index& ix, int& tbits)
{
	const int mbits = 2 * (int)ix.prm.m - 1;
	const int want_bits = ix.tune_bin_bits > 0 ? std::min(ix.tune_bin_bits, kBinBitsMax) : kBinBitsOnepass;
	tbits = std::max(3, std::min(mbits, want_bits));
	const uint32_t nbound = ix.shard_world > 1 ? ix.shard_world - 1 : 0;
	while (tbits > 1 && (1u << tbits) + nbound > (1u << kBinBitsMax))  // the kernels hold at most 2^kBinBitsMax bins
		tbits--;
	BinMap bm;
	bm.shift = mbits - tbits;
	bm.nbound = 0;
	for (uint32_t r = 1; r < ix.shard_world; r++)
		bm.bound[bm.nbound++] = ix.shard_bounds[r] << ix.gshift;
	return bm;
}

uint32_t slot_owner_bins(const Index& ix, uint32_t owner)
{
	int tbits;
	const BinMap bm = sharded_bin_map(ix, tbits);
	const uint32_t nbins = (1u << tbits) + bm.nbound, world = ix.shard_world;
	const uint32_t lo = owner == 0 ? 0u : bm.bin_of(bm.bound[owner - 1]);
	const uint32_t hi = owner + 1 >= world ? nbins : bm.bin_of(bm.bound[owner]);
	return hi - lo;
}

void slot_piece_partition(Index& ix, SlotPiece& sp, const void* d_kmers, const uint32_t* d_minis, uint64_t n, hipStream_t s)
{
	int tbits;
	const BinMap bm = sharded_bin_map(ix, tbits);
	const uint32_t nbins = (1u << tbits) + bm.nbound, world = ix.shard_world;
	const bool learnt = ix.plan_nbins_sharded == nbins;
	const uint32_t pct = ix.tune_slot_slack_pct > 0 ? (uint32_t)ix.tune_slot_slack_pct : (learnt ? 6u : 12u);
	sp.nq = n;
	sp.nbins = nbins;
	sp.world = world;
	sp.max_main = ((n + n * pct / 100 + (uint64_t)nbins * 512 + 2047) / 2048) * 2048;
	const uint64_t slots = sp.max_main + n;
	sp.tile = (uint32_t)(ix.KW == 1 ? BinTile<1>::value : BinTile<2>::value);
	sp.nblk = (uint32_t)((n + sp.tile - 1) / sp.tile);
	sp.narrow = ix.tune_narrow_ids && ix.st.number_kmer < 0xffffffffull;
	sp.lpos_bytes = (n * 2 + 255) & ~(size_t)255;
	sp.runs = (size_t)sp.nblk << kBinBitsMax;
	sp.first_bin[0] = 0;
	for (uint32_t r = 1; r < world; r++)
		sp.first_bin[r] = bm.bin_of(bm.bound[r - 1]);
	sp.first_bin[world] = nbins;
	ensure_buf(sp.slots, slots * 8 * (size_t)ix.KW, s);
	ensure_buf(sp.minis, slots * 4, s);
	ensure_buf(sp.ids, slots * 8, s);
	ensure_buf(sp.lpos, sp.lpos_bytes + sp.runs * 12, s);
	ensure_buf(sp.plan, sizeof(SlotPlan), s);
	ensure_buf(sp.cursors, ((size_t)nbins + 2) * 4, s);
	ensure_buf(sp.hdr, ((size_t)nbins + 2) * 8, s);
	ensure_buf(sp.info, 128, s);
	sp.h_info.ensure(128);
	RunTable rt;
	rt.gbase = reinterpret_cast<uint32_t*>(sp.lpos.as<char>() + sp.lpos_bytes);
	rt.obase = rt.gbase + sp.runs;
	rt.count = rt.obase + sp.runs;
	hipLaunchKernelGGL(k_plan_make<1>, dim3(1), dim3(1 << kBinBitsMax), 0, s, learnt ? ix.b_frac_sharded.as<float>() : nullptr, n, nbins, pct, (uint32_t)sp.max_main,
	                   (uint32_t)n, 1024u, sp.plan.as<SlotPlan>(), sp.cursors.as<uint32_t>());
	const uint64_t* km = reinterpret_cast<const uint64_t*>(d_kmers);
	const uint32_t pgrid = onepass_grid(sp.nblk);
#define BLMX_OPP(KWV, MV)                                                                                                                                  \
	do {                                                                                                                                                   \
		if (ix.tune_bin_rounds != 1)                                                                                                                       \
			hipLaunchKernelGGL((k_bin_onepass<KWV, 1, MV, 1024, 2>), dim3(onepass_grid(sp.nblk, 2)), dim3(kBinThreads), 0, s, (int)ix.prm.k, (int)ix.prm.m,   \
			                   bm, sp.plan.as<SlotPlan>(), km, d_minis, n, sp.cursors.as<uint32_t>(), sp.slots.as<uint64_t>(), sp.minis.as<uint32_t>(),       \
			                   sp.lpos.as<uint16_t>(), rt);                                                                                                \
		else                                                                                                                                               \
			hipLaunchKernelGGL((k_bin_onepass<KWV, 1, MV, 1024, 1>), dim3(pgrid), dim3(kBinThreads), 0, s, (int)ix.prm.k, (int)ix.prm.m, bm,                  \
			                   sp.plan.as<SlotPlan>(), km, d_minis, n, sp.cursors.as<uint32_t>(), sp.slots.as<uint64_t>(), sp.minis.as<uint32_t>(),           \
			                   sp.lpos.as<uint16_t>(), rt);                                                                                                \
	} while (0)
	if (ix.KW == 1) {
		if (d_minis)
			BLMX_OPP(1, true);
		else
			BLMX_OPP(1, false);
	} else {
		if (d_minis)
			BLMX_OPP(2, true);
		else
			BLMX_OPP(2, false);
	}
#undef BLMX_OPP
	FirstBins f;
	f.world = world;
	for (uint32_t r = 0; r <= world; r++)
		f.fb[r] = sp.first_bin[r];
	hipLaunchKernelGGL(k_slot_info, dim3(1), dim3(1 << kBinBitsMax), 0, s, sp.plan.as<SlotPlan>(), sp.cursors.as<uint32_t>(), f, sp.hdr.as<uint64_t>(),
	                   sp.info.as<uint32_t>());
	BLMX_HIP(hipGetLastError());
	BLMX_HIP(hipMemcpyAsync(sp.h_info.p, sp.info.p, 4 * (size_t)(world + 2), hipMemcpyDeviceToHost, s));
}

// The owner's kernels.  Every source is a slot range with gaps (the unused tails of the bin regions).  The workgroups walk the
// regions BIN-MAJOR -- bin 0 of every source, then bin 1 of every source, ... -- so that one bin's slice of the index is worked on by
// neighbouring workgroups whatever the number of senders (source-major order would fetch every slice once per sender).
// k_foreign_plan (one workgroup, one thread per bin) lays out the workgroups: first_block[bin * nsrc + source]; the query kernel
// finds its (bin, source) by a binary search over that table (uniform: scalar loads) and answers the slots that are in use.
struct ForeignTable {
	uint32_t nsrc, nb;
	ForeignSlots src[16];
};
__global__ __launch_bounds__(1 << kBinBitsMax) void k_foreign_plan(ForeignTable ft, uint32_t* __restrict__ first_block)
{
	__shared__ uint32_t s_wsum[(1 << kBinBitsMax) / 64];
	const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	uint32_t mine = 0;
	if (t < ft.nb)
		for (uint32_t i = 0; i < ft.nsrc; i++)
			mine += ((uint32_t)ft.src[i].hdr[t + 1] - (uint32_t)ft.src[i].hdr[t]) >> 8;
	uint32_t v = mine;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t u = __shfl_up(v, o);
		if ((int)lane >= o)
			v += u;
	}
	if (lane == 63)
		s_wsum[wave] = v;
	__syncthreads();
	uint32_t wbase = 0;
	for (uint32_t w = 0; w < wave; w++)
		wbase += s_wsum[w];
	uint32_t at = wbase + v - mine;
	if (t < ft.nb) {
		for (uint32_t i = 0; i < ft.nsrc; i++) {
			first_block[t * ft.nsrc + i] = at;
			at += ((uint32_t)ft.src[i].hdr[t + 1] - (uint32_t)ft.src[i].hdr[t]) >> 8;
		}
		if (t + 1 == ft.nb)
			first_block[ft.nb * ft.nsrc] = at;
	}
}

template <int KW, int LOGP>
__global__ __launch_bounds__(256) void k_query_foreign_slots(DeviceIndexView v, ForeignTable ft, const uint32_t* __restrict__ first_block, int narrow)
{
	const uint32_t n = ft.nb * ft.nsrc, nblocks = first_block[n];
	const uint32_t per = gridDim.x >> 3;
	const uint32_t blk = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);  // XCD x takes the x-th contiguous eighth (the grid is a multiple of 8)
	if (blk >= nblocks)
		return;
	uint32_t lo = 0, hi = n;  // the last (bin, source) whose first block is <= blk (empty regions share their successor's first block)
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (first_block[mid] <= blk)
			lo = mid;
		else
			hi = mid;
	}
	const uint32_t bin = lo / ft.nsrc, si = lo - bin * ft.nsrc;
	const ForeignSlots& fs = ft.src[si];
	const uint64_t h = fs.hdr[bin];
	const uint32_t rb = (uint32_t)h - (uint32_t)fs.hdr[0], used = (uint32_t)(h >> 32);  // region start inside the range, slots in use
	const uint32_t slot = rb + ((blk - first_block[lo]) << 8) + threadIdx.x;
	if (slot < rb + used && slot < fs.nslots)
		query_slot<KW, LOGP>(gated_view(v), fs.slots, fs.minis, fs.ids, narrow != 0, slot);
}

void launch_query_foreign_slots(Index& ix, const ForeignSlots* src, uint32_t nsrc, bool narrow, hipStream_t s)
{
	ForeignTable ft{};
	uint32_t nblocks = 0;
	for (uint32_t i = 0; i < nsrc && ft.nsrc < 16; i++) {
		if (src[i].nslots == 0 || src[i].nb == 0)
			continue;
		ft.nb = src[i].nb;  // the same for every source: the bins this rank owns
		ft.src[ft.nsrc++] = src[i];
		nblocks += (src[i].nslots + 255) >> 8;  // regions are multiples of 256 slots: this is the exact number of workgroups
	}
	if (nblocks == 0)
		return;
	std::lock_guard<std::mutex> lock(ix.bin_mu);
	partition_begin(ix, s);
	ensure_buf(ix.b_foreign, ((size_t)ft.nb * ft.nsrc + 1) * 4, s);
	hipLaunchKernelGGL(k_foreign_plan, dim3(1), dim3(1 << kBinBitsMax), 0, s, ft, ix.b_foreign.as<uint32_t>());
	const DeviceIndexView v = ix.view();
	const unsigned grid = (nblocks + 7u) & ~7u;
	const uint32_t* fb = ix.b_foreign.as<uint32_t>();
#define BLMX_QF(LP)                                                                                                            \
	case LP:                                                                                                                   \
		if (ix.KW == 1)                                                                                                        \
			hipLaunchKernelGGL((k_query_foreign_slots<1, LP>), dim3(grid), dim3(256), 0, s, v, ft, fb, (int)narrow);           \
		else                                                                                                                   \
			hipLaunchKernelGGL((k_query_foreign_slots<2, LP>), dim3(grid), dim3(256), 0, s, v, ft, fb, (int)narrow);           \
		break;
	switch (v.b) {
		BLMX_QF(0)
		BLMX_QF(1)
		BLMX_QF(2)
		BLMX_QF(3)
		BLMX_QF(4)
		BLMX_QF(5)
		BLMX_QF(6)
		BLMX_QF(7)
		BLMX_QF(8)
	default:
		throw StatusError{BLMX_EINVAL, "internal: the slot pipeline needs bits_to_save <= 8"};
	}
#undef BLMX_QF
	BLMX_HIP(hipGetLastError());
	partition_end(ix, s);
}

// IDs of the piece (they sit in sp.ids, slot by slot, wherever they were answered) -> the caller's order; this batch's shares become
// the next plan; the filter gate follows this batch's answers
void slot_piece_gather(Index& ix, SlotPiece& sp, int64_t* d_out, hipStream_t s)
{
	RunTable rt;
	rt.gbase = reinterpret_cast<uint32_t*>(sp.lpos.as<char>() + sp.lpos_bytes);
	rt.obase = rt.gbase + sp.runs;
	rt.count = rt.obase + sp.runs;
	const SlotPlan* d_plan = sp.plan.as<SlotPlan>();
	const uint16_t* d_lpos = sp.lpos.as<uint16_t>();
	if (ix.KW == 1) {
		if (sp.narrow)
			hipLaunchKernelGGL((k_gather_runs<BinTile<1>::value, true, 1>), dim3(sp.nblk), dim3(kBinThreads), 0, s, d_plan, rt, d_lpos, sp.ids.p, sp.nq, d_out);
		else
			hipLaunchKernelGGL((k_gather_runs<BinTile<1>::value, false, 1>), dim3(sp.nblk), dim3(kBinThreads), 0, s, d_plan, rt, d_lpos, sp.ids.p, sp.nq, d_out);
	} else {
		if (sp.narrow)
			hipLaunchKernelGGL((k_gather_runs<BinTile<2>::value, true, 1>), dim3(sp.nblk), dim3(kBinThreads), 0, s, d_plan, rt, d_lpos, sp.ids.p, sp.nq, d_out);
		else
			hipLaunchKernelGGL((k_gather_runs<BinTile<2>::value, false, 1>), dim3(sp.nblk), dim3(kBinThreads), 0, s, d_plan, rt, d_lpos, sp.ids.p, sp.nq, d_out);
	}
	BLMX_HIP(hipGetLastError());
	ensure_buf(ix.b_frac_sharded, sizeof(float) << kBinBitsSlots, s);
	if (!ix.b_pstats.p)
		BLMX_HIP(ix.b_pstats.alloc(64, true, s));
	hipLaunchKernelGGL(k_plan_learn, dim3((sp.nbins + 255) / 256), dim3(256), 0, s, sp.cursors.as<uint32_t>(), sp.nq, sp.nbins, ix.b_frac_sharded.as<float>(),
	                   ix.b_pstats.as<unsigned long long>(), ix.dropped_word());
	ix.plan_nbins_sharded = sp.nbins;
	update_filter_gate(ix, d_out, sp.nq, s);
}

// d_minis: minimizers of the k-mers when the caller already has them (read front end), else nullptr
static void query_ids_pipeline(Index& ix, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, int64_t* d_out, hipStream_t s)
{
	if (nq == 0)
		return;
	if (!d_minis) {  // (the read front end has already marked 0 and 1)
		ix.prof_reset();
		ix.prof_mark(0, s);
		ix.prof_mark(1, s);
	}
	if (!query_batch_worth_binning(ix, nq)) {
		ix.prof_mark(4, s);
		launch_query(ix, d_kmers, d_minis, nq, d_out, kQueryIds, s);
		ix.prof_mark(5, s);
		return;
	}
	std::lock_guard<std::mutex> lock(ix.bin_mu);
	partition_begin(ix, s);
	// hand the minimizers over to the query kernel?  Known minimizers always travel; computed ones only for 128-bit k-mers
	// (55 m-mers hashed once instead of twice; for 64-bit k-mers the extra 8 B/query of traffic ate the gain) unless tuned.
	const bool handoff = ix.tune_mini_handoff >= 0 ? ix.tune_mini_handoff != 0 : (d_minis != nullptr || ix.KW == 2);
	// One pass into bin regions sized after the previous batch when there is one to learn from (automatic), always (1) or never (2)
	// (slot numbers are 32-bit: bin regions + an overflow region that can hold the whole batch = about 2.1 slots per k-mer)
	const bool onepass_ok = ix.prm.bits_saved <= 8 && nq < 1900000000ull && 2 * (int)ix.prm.m - 1 >= 3;
	ix.check_dropped();
	// (more bins than the exact partition holds: the first batch goes through the one-pass kernels too, with equal shares -- what does
	// not fit a region goes through the overflow region -- and the next one knows its shares)
	if (onepass_ok && (ix.tune_partition == 1 || (ix.tune_partition == 0 && (ix.plan_bits == onepass_bits(ix) || onepass_bits(ix) > kBinBitsMax)))) {
		// behind the filter and with 1024 bins the slot kernel is issue-bound (90 % VALU, profiles/r02_pmc_c4.json): now the 4 B per
		// k-mer of a handed-over minimizer are cheaper than computing it again (C4: 5.71 -> 5.62 ms per step)
		const bool handoff1 = ix.tune_mini_handoff >= 0 ? ix.tune_mini_handoff != 0 : true;
		query_ids_onepass(ix, d_kmers, d_minis, nq, d_out, handoff1, s);
		partition_end(ix, s);
		return;
	}
	if (handoff)
		ensure_buf(ix.b_minis, nq * 4, s);
	uint32_t* d_sorted_minis = handoff ? ix.b_minis.as<uint32_t>() : nullptr;
	ensure_buf(ix.b_sorted, nq * 8 * (size_t)ix.KW, s);
	ensure_buf(ix.b_inv, nq * 4, s);
	ensure_buf(ix.b_ids, nq * 8, s);
	const PartitionPlan pl = partition_batch(ix, d_kmers, d_minis, nq, false, ix.b_sorted.p, d_sorted_minis, ix.b_inv.as<uint32_t>(), s,
	                                         (onepass_ok && ix.tune_partition == 0) ? kBinBitsOnepass : kBinBitsDefault);
	// (storing each ID straight into the caller's slot from k_query was measured 2.8 ms SLOWER per 1e8 queries than this
	// separate gather: 1e8 scattered 8-B stores cost more than reading the IDs back through the inverse permutation)
	// The ordered IDs are 32-bit words when every ID fits: half the bytes written by the query kernel and gathered at random.
	const bool narrow = ix.tune_narrow_ids && ix.st.number_kmer < 0xffffffffull && ix.prm.bits_saved <= 8;
	launch_query(ix, ix.b_sorted.p, d_sorted_minis, nq, ix.b_ids.p, (QueryMode)((narrow ? kQueryIds32 : kQueryIds) | kQueryXcdMajor), s);
	ix.prof_mark(5, s);
	gather_ids(ix, ix.b_ids.p, ix.b_inv.as<uint32_t>(), nq, d_out, pl.tile, narrow, s);
	ix.prof_mark(6, s);
	update_filter_gate(ix, d_out, nq, s);
	if (onepass_ok && ix.tune_partition == 0 && pl.bm.nbound == 0 && (int)pl.nbins == (1 << onepass_bits(ix))) {  // the next batch can skip the histogram
		ensure_buf(ix.b_frac, sizeof(float) << kBinBitsSlots, s);
		hipLaunchKernelGGL(k_plan_learn_offs, dim3((pl.nbins + 255) / 256), dim3(256), 0, s, ix.b_offs.as<uint32_t>(), pl.nblk, nq, pl.nbins, ix.b_frac.as<float>());
		ix.plan_bits = onepass_bits(ix);
	}
	partition_end(ix, s);
}

// Device-resident batches of any size: the pipeline numbers slots and k-mers with 32 bits, so a batch above 2^30 k-mers is answered
// piece by piece, like the host-buffer calls do with their windows (max_batch: test hook that lowers the piece size).
static void query_ids_chunked(Index& ix, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, int64_t* d_out, hipStream_t s)
{
	const uint64_t piece = ix.tune_max_batch > 0 ? (uint64_t)ix.tune_max_batch : (1ull << 30);
	for (uint64_t at = 0; at < nq || at == 0; at += piece) {
		const uint64_t n = std::min(piece, nq - at);
		query_ids_pipeline(ix, reinterpret_cast<const uint64_t*>(d_kmers) + at * (uint64_t)ix.KW, d_minis ? d_minis + at : nullptr, n, d_out + at, s);
		if (nq == 0)
			break;
	}
}

void launch_query_ids_auto(Index& ix, const void* d_kmers, uint64_t nq, int64_t* d_out, hipStream_t s) { query_ids_chunked(ix, d_kmers, nullptr, nq, d_out, s); }

void launch_query_ids_with_minis(Index& ix, const void* d_kmers, const uint32_t* d_minis, uint64_t nq, int64_t* d_out, hipStream_t s)
{
	query_ids_chunked(ix, d_kmers, d_minis, nq, d_out, s);
}

// ---- the three stages as separate entry points: the sharded exchange (blight_amd/dist.py) partitions on the rank that
// holds the queries, ships every owner its contiguous segment, queries the received (already bin-ordered) segments
// without another partition pass, ships the IDs back and gathers them through the inverse permutation.
void launch_partition(Index& ix, const void* d_kmers, uint64_t nq, void* d_sorted, uint32_t* d_inv, uint64_t* d_rank_counts, hipStream_t s)
{
	if (nq >= 0xffffffffull)
		throw StatusError{BLMX_ELIMIT, "a partitioned batch holds fewer than 2^32 - 1 k-mers"};
	const uint32_t world = ix.shard_world;
	if (nq == 0) {
		if (d_rank_counts)
			BLMX_HIP(hipMemsetAsync(d_rank_counts, 0, 8 * (size_t)world, s));
		return;
	}
	std::lock_guard<std::mutex> lock(ix.bin_mu);
	partition_begin(ix, s);
	// the same bins as the one-pass partition of the sharded exchange, which learns its region sizes from this pass
	const PartitionPlan pl = partition_batch(ix, d_kmers, nullptr, nq, true, d_sorted, nullptr, d_inv, s, ix.tune_partition == 2 ? kBinBitsDefault : kBinBitsOnepass);
	ix.staged_tile = pl.tile;
	if (d_rank_counts) {
		OwnerBins ob{};  // the bins of owner r start at bin_of(its first minimizer); by value: nothing of this frame is read later
		ob.fb[0] = 0;
		for (uint32_t r = 1; r < world; r++)
			ob.fb[r] = pl.bm.bin_of(pl.bm.bound[r - 1]);
		ob.fb[world] = pl.nbins;
		hipLaunchKernelGGL(k_rank_counts, dim3(1), dim3(64), 0, s, ix.b_offs.as<uint32_t>(), pl.nblk, pl.nbins, nq, world, ob, d_rank_counts);
		BLMX_HIP(hipGetLastError());
	}
	if (ix.tune_partition != 2 && ix.prm.bits_saved <= 8) {  // what the one-pass partition of the next sharded batch sizes its regions after
		ensure_buf(ix.b_frac_sharded, sizeof(float) << kBinBitsSlots, s);
		hipLaunchKernelGGL(k_plan_learn_offs, dim3((pl.nbins + 255) / 256), dim3(256), 0, s, ix.b_offs.as<uint32_t>(), pl.nblk, nq, pl.nbins,
		                   ix.b_frac_sharded.as<float>());
		ix.plan_nbins_sharded = pl.nbins;
	}
	partition_end(ix, s);
}

void launch_query_partitioned(Index& ix, const void* d_sorted_kmers, uint64_t nq, int64_t* d_ids, hipStream_t s)
{
	if (nq == 0)
		return;
	launch_query(ix, d_sorted_kmers, nullptr, nq, d_ids, (QueryMode)(kQueryIds | kQueryXcdMajor), s);
	update_filter_gate(ix, d_ids, nq, s);  // (an owner's segments of a sharded batch: 64-bit IDs, dense)
}

void launch_gather_ids(Index& ix, const int64_t* d_ids_sorted, const uint32_t* d_inv, uint64_t nq, int64_t* d_out, hipStream_t s)
{
	if (nq == 0)
		return;
	gather_ids(ix, d_ids_sorted, d_inv, nq, d_out, ix.staged_tile ? ix.staged_tile : (uint32_t)(BinTile<1>::value / ix.KW), false, s);
}

void launch_kmer_groups(const Index& ix, const void* d_kmers, uint64_t nq, uint32_t* d_groups, hipStream_t s)
{
	if (nq == 0)
		return;
	const uint64_t* km = reinterpret_cast<const uint64_t*>(d_kmers);
	if (ix.KW == 1)
		hipLaunchKernelGGL((k_kmer_groups<1>), dim3(grid_for(nq, 256)), dim3(256), 0, s, (int)ix.prm.k, (int)ix.prm.m, ix.gshift, km,
		                   nq, d_groups);
	else
		hipLaunchKernelGGL((k_kmer_groups<2>), dim3(grid_for(nq, 256)), dim3(256), 0, s, (int)ix.prm.k, (int)ix.prm.m, ix.gshift, km,
		                   nq, d_groups);
	BLMX_HIP(hipGetLastError());
}

void launch_kmers_at(const Index& ix, const uint8_t* d_ascii, const uint64_t* d_pos, uint64_t nq, void* d_out, int* d_err, hipStream_t s)
{
	if (nq == 0)
		return;
	uint64_t* out = reinterpret_cast<uint64_t*>(d_out);
	if (ix.KW == 1)
		hipLaunchKernelGGL((k_kmers_at<1>), dim3(grid_for(nq, 256)), dim3(256), 0, s, (int)ix.prm.k, d_ascii, d_pos, nq, out, d_err);
	else
		hipLaunchKernelGGL((k_kmers_at<2>), dim3(grid_for(nq, 256)), dim3(256), 0, s, (int)ix.prm.k, d_ascii, d_pos, nq, out, d_err);
	BLMX_HIP(hipGetLastError());
}

void launch_canonicalize(const Index& ix, void* d_kmers, uint64_t nq, hipStream_t s)
{
	if (nq == 0)
		return;
	uint64_t* km = reinterpret_cast<uint64_t*>(d_kmers);
	if (ix.KW == 1)
		hipLaunchKernelGGL((k_canonicalize<1>), dim3(grid_for(nq, 256)), dim3(256), 0, s, (int)ix.prm.k, km, nq);
	else
		hipLaunchKernelGGL((k_canonicalize<2>), dim3(grid_for(nq, 256)), dim3(256), 0, s, (int)ix.prm.k, km, nq);
	BLMX_HIP(hipGetLastError());
}

void launch_route_count(const Index& ix, const void* d_kmers, uint64_t nq, const uint32_t* d_bounds, uint32_t world, uint8_t* d_dest,
                        uint64_t* d_counts, hipStream_t s)
{
	BLMX_HIP(hipMemsetAsync(d_counts, 0, (size_t)world * 8, s));
	if (nq == 0)
		return;
	const uint64_t* km = reinterpret_cast<const uint64_t*>(d_kmers);
	unsigned long long* cnt = reinterpret_cast<unsigned long long*>(d_counts);
	if (ix.KW == 1)
		hipLaunchKernelGGL((k_route_count<1>), dim3(grid_for(nq, 256)), dim3(256), 0, s, (int)ix.prm.k, (int)ix.prm.m, ix.gshift, km, nq, d_bounds,
		                   world, d_dest, cnt);
	else
		hipLaunchKernelGGL((k_route_count<2>), dim3(grid_for(nq, 256)), dim3(256), 0, s, (int)ix.prm.k, (int)ix.prm.m, ix.gshift, km, nq, d_bounds,
		                   world, d_dest, cnt);
	BLMX_HIP(hipGetLastError());
}

void launch_route_scatter(const Index& ix, const void* d_kmers, const uint8_t* d_dest, uint64_t nq, const uint64_t* d_offsets, uint32_t world,
                          void* d_sorted, uint32_t* d_order, uint64_t* d_cursor, hipStream_t s)
{
	BLMX_HIP(hipMemsetAsync(d_cursor, 0, (size_t)world * 8, s));
	if (nq == 0)
		return;
	const uint64_t* km = reinterpret_cast<const uint64_t*>(d_kmers);
	unsigned long long* cur = reinterpret_cast<unsigned long long*>(d_cursor);
	if (ix.KW == 1)
		hipLaunchKernelGGL((k_route_scatter<1>), dim3(grid_for(nq, 256)), dim3(256), 0, s, km, d_dest, nq, d_offsets, world,
		                   reinterpret_cast<uint64_t*>(d_sorted), d_order, cur);
	else
		hipLaunchKernelGGL((k_route_scatter<2>), dim3(grid_for(nq, 256)), dim3(256), 0, s, km, d_dest, nq, d_offsets, world,
		                   reinterpret_cast<uint64_t*>(d_sorted), d_order, cur);
	BLMX_HIP(hipGetLastError());
}

void launch_unpermute(const int64_t* d_vals, const uint32_t* d_order, uint64_t nq, int64_t* d_out, hipStream_t s)
{
	if (nq == 0)
		return;
	hipLaunchKernelGGL(k_unpermute, dim3(grid_for(nq, 256)), dim3(256), 0, s, d_vals, d_order, nq, d_out);
	BLMX_HIP(hipGetLastError());
}

void launch_primitive(int op, int KW, const void* d_in, uint64_t n, int arg, int arg2, void* d_out, hipStream_t s)
{
	if (n == 0)
		return;
	if (KW == 1)
		hipLaunchKernelGGL((k_primitive<1>), dim3(grid_for(n, 256)), dim3(256), 0, s, op, d_in, n, arg, arg2, d_out);
	else
		hipLaunchKernelGGL((k_primitive<2>), dim3(grid_for(n, 256)), dim3(256), 0, s, op, d_in, n, arg, arg2, d_out);
	BLMX_HIP(hipGetLastError());
}

}  // namespace blmx
