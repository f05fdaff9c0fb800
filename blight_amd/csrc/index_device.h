// index_device.h -- device-side accessors of the finished index shared by the build (fill_positions) and query kernels.
#pragma once
#include "blmx_internal.h"

namespace blmx {

template <int KW> __device__ __forceinline__ Kmer<KW> load_kmer(const uint64_t* __restrict__ p, uint64_t i);
template <> __device__ __forceinline__ Kmer<1> load_kmer<1>(const uint64_t* __restrict__ p, uint64_t i)
{
	Kmer<1> x;
	x.lo = p[i];
	return x;
}
template <> __device__ __forceinline__ Kmer<2> load_kmer<2>(const uint64_t* __restrict__ p, uint64_t i)
{
	const ulonglong2 v = reinterpret_cast<const ulonglong2*>(p)[i];
	Kmer<2> x;
	x.lo = v.x;
	x.hi = v.y;
	return x;
}

// level_finder_tmp + bitVector::rank (bbhash.h:617-663, 467-480): probe levels until a set bit is found and rank it.
// One probe = ONE 16-B load: a line holds [u32 rank | 96 level bits], so the rank costs no second memory access.
// Logical bit b of the group's level array lives in line (b>>5)/3, data word (b>>5)%3, bit b&31.
template <int KW>
__device__ __forceinline__ uint64_t mphf_lookup_levels(const DeviceIndexView& v, const GroupDesc* __restrict__ gd, Kmer<KW> x)
{
	LevelHasher hs(x.lo);
	const uint4* __restrict__ lines = reinterpret_cast<const uint4*>(v.lines) + gd->line_base;
	uint64_t off = 0;
	for (int lv = 0; lv < kLevels; lv++) {
		const uint64_t size = gd->size[lv];
		const uint64_t bit = off + reduce_level(hs.next(lv), size);
		const uint64_t w32 = bit >> 5;
		const uint64_t line = w32 / kLineData32;
		const uint32_t slot = (uint32_t)(w32 - line * kLineData32);
		const uint4 L = lines[line];
		const uint32_t word = slot == 0 ? L.y : (slot == 1 ? L.z : L.w);
		const uint32_t bi = (uint32_t)bit & 31u;
		if ((word >> bi) & 1u) {
			const uint32_t before = (slot > 0 ? __popc(L.y) : 0) + (slot > 1 ? __popc(L.z) : 0);
			return (uint64_t)(L.x + before + __popc(word & ((1u << bi) - 1u)));
		}
		off += size;
	}
	return kNotFound;
}

// mphf::lookup (bbhash.h:561-577): levels first, then the group's fallback table.
template <int KW>
__device__ __forceinline__ uint64_t mphf_lookup(const DeviceIndexView& v, const GroupDesc* __restrict__ gd, Kmer<KW> x)
{
	const uint64_t r = mphf_lookup_levels<KW>(v, gd, x);
	if (r != kNotFound || gd->fb_count == 0)
		return r;
	// _final_hash.find(elem), bbhash.h:567-575.  The table is sorted by (group, key, input order) and keeps every
	// occurrence of a duplicated key; the map's "last assignment wins" (bbhash.h:724) = the LAST entry of an equal run.
	uint64_t lo = gd->fb_start, hi = lo + gd->fb_count;
	const uint64_t begin = lo;
	while (lo < hi) {  // upper bound
		const uint64_t mid = (lo + hi) >> 1;
		const Kmer<KW> km = load_kmer<KW>(v.fb_keys, mid);
		if (x < km)
			hi = mid;
		else
			lo = mid + 1;
	}
	if (lo > begin && load_kmer<KW>(v.fb_keys, lo - 1) == x)
		return v.fb_ranks[lo - 1];
	return kNotFound;
}

// bool_to_int, blight.cpp:473-482: nbits-wide little-endian field at bit posstart + h*nbits, times 2^b in uint32_t.
__device__ __forceinline__ uint32_t read_position(const uint64_t* __restrict__ positions, uint64_t posstart, uint32_t nbits,
                                                  uint64_t h, int b)
{
	const uint64_t o = posstart + h * (uint64_t)nbits;
	const uint64_t w = o >> 6;
	const uint32_t s = (uint32_t)o & 63u;
	const uint64_t lo = positions[w], hi = positions[w + 1];
	const uint64_t v = s ? ((lo >> s) | (hi << (64 - s))) : lo;
	const uint32_t field = (uint32_t)(v & ((nbits >= 32) ? 0xffffffffull : ((1ull << nbits) - 1ull)));
	return field << b;
}

}  // namespace blmx
