// index_device.h -- device-side accessors of the finished index shared by the build (fill_positions) and query kernels.
#pragma once
#include <cstddef>

#include "blmx_internal.h"

namespace blmx {

// Experiment hook (tools: `python -m blight_amd.build --variant nt -DBLMX_EXP_NT=mask`): non-temporal hints on the loads and stores whose
// lines nobody touches twice, so that they do not push the level-0 lines (3.8 touches per 128-B block and batch) out of the XCD's L2.
// bit 0 the batch's slots and minimizers, 1 position fields, 2 sequence windows, 3 the IDs written per slot, 4 deeper MPHF levels.
#ifndef BLMX_EXP_NT
#define BLMX_EXP_NT 0
#endif
template <int BIT, typename T> __device__ __forceinline__ T exp_load(const T* p)
{
	if constexpr ((BLMX_EXP_NT >> BIT) & 1)
		return __builtin_nontemporal_load(p);
	else
		return *p;
}
template <int BIT, typename T> __device__ __forceinline__ void exp_store(T* p, T v)
{
	if constexpr ((BLMX_EXP_NT >> BIT) & 1)
		__builtin_nontemporal_store(v, p);
	else
		*p = v;
}
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

// One BBHash probe = ONE 16-B load: a line holds [u32 rank | 96 level bits], so the rank costs no second memory access.
// Logical bit b of the group's level array lives in line (b>>5)/3, data word (b>>5)%3, bit b&31.
// Returns true and sets `rank` when the bit is set (level_finder_tmp + bitVector::rank, bbhash.h:617-663, 467-480).
__device__ __forceinline__ bool mphf_probe_line(const uint4 L, uint32_t slot, uint32_t bi, uint64_t& rank)
{
	// select without control flow so that the load stays one dwordx4
	const uint32_t word = (slot == 0 ? L.y : 0u) | (slot == 1 ? L.z : 0u) | (slot == 2 ? L.w : 0u);
	const uint32_t before = (slot > 0 ? __popc(L.y) : 0) + (slot > 1 ? __popc(L.z) : 0);
	rank = (uint64_t)(L.x + before + __popc(word & ((1u << bi) - 1u)));
	return (word >> bi) & 1u;
}
__device__ __forceinline__ bool mphf_probe(const uint4* __restrict__ lines, uint64_t bit, uint64_t& rank)
{
	const uint32_t w32 = (uint32_t)(bit >> 5);  // a group's level array is < 2^37 bits (checked at build time)
	const uint32_t line = w32 / kLineData32;
	if constexpr ((BLMX_EXP_NT >> 4) & 1) {
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&lines[line]));
		return mphf_probe_line(make_uint4(q.x, q.y, q.z, q.w), w32 - line * kLineData32, (uint32_t)bit & 31u, rank);
	}
	return mphf_probe_line(lines[line], w32 - line * kLineData32, (uint32_t)bit & 31u, rank);
}

// The first 64-B sector of a GroupDesc, fetched with four back-to-back 16-B loads and ONE wait: everything a query
// needs besides the level sizes 3..15 (which only 6 % of present and a third of absent k-mers reach).
struct GroupHead {
	uint64_t prefix, posstart, line_base, fb_start;
	uint32_t nbits, fb_count;
	uint64_t size0, size1, size2;
};
__device__ __forceinline__ GroupHead unpack_group_head(const uint4 a, const uint4 b, const uint4 c, const uint4 d)
{
	GroupHead h;
	h.prefix = (uint64_t)a.x | ((uint64_t)a.y << 32);
	h.posstart = (uint64_t)a.z | ((uint64_t)a.w << 32);
	h.line_base = (uint64_t)b.x | ((uint64_t)b.y << 32);
	h.nbits = b.z;
	h.fb_count = b.w;
	h.fb_start = (uint64_t)c.x | ((uint64_t)c.y << 32);
	h.size0 = (uint64_t)c.z | ((uint64_t)c.w << 32);
	h.size1 = (uint64_t)d.x | ((uint64_t)d.y << 32);
	h.size2 = (uint64_t)d.z | ((uint64_t)d.w << 32);
	return h;
}
__device__ __forceinline__ GroupHead load_group_head(const GroupDesc* __restrict__ gd)
{
	const uint4* p = reinterpret_cast<const uint4*>(gd);
	return unpack_group_head(p[0], p[1], p[2], p[3]);
}

// Bucket descriptor (16 B) + group head (64 B) in ONE memory round trip: both addresses depend only on the minimizer.
// hipcc splits and sinks plain loads behind the empty-bucket test (turning one round trip into three dependent ones),
// so the five 16-B loads are issued from one asm statement and drained by its own s_waitcnt (the statement counts and
// waits for its own loads; no compiler-tracked load is outstanding across it that the kernel relies on).
__device__ __forceinline__ GroupHead load_bucket_and_head(const BucketDesc* __restrict__ bp, const GroupDesc* __restrict__ gp, uint64_t& bstart,
                                                          uint32_t& bnt)
{
	uint4 bd, a, b, c, d;
	asm volatile(
	    "global_load_dwordx4 %0, %5, off\n\t"
	    "global_load_dwordx4 %1, %6, off\n\t"
	    "global_load_dwordx4 %2, %6, off offset:16\n\t"
	    "global_load_dwordx4 %3, %6, off offset:32\n\t"
	    "global_load_dwordx4 %4, %6, off offset:48\n\t"
	    "s_waitcnt vmcnt(0)"
	    : "=&v"(bd), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
	    : "v"(bp), "v"(gp)
	    : "memory");
	bstart = (uint64_t)bd.x | ((uint64_t)bd.y << 32);
	bnt = bd.z;
	return unpack_group_head(a, b, c, d);
}

__device__ __forceinline__ GroupHead unpack_query_head(const uint4& a, const uint4& b)
{
	GroupHead h;
	h.prefix = (uint64_t)a.x | ((uint64_t)a.y << 32);
	h.posstart = (uint64_t)a.z | ((uint64_t)(a.w & 0x00ffffffu) << 32);
	h.nbits = a.w >> 24;
	h.line_base = b.x;
	h.size0 = b.y;
	h.size1 = b.z;
	h.size2 = b.w;
	h.fb_start = 0;
	h.fb_count = 0;
	return h;
}

// The same round trip with the compact QueryHead (32 B): three 16-B loads instead of five.  fb_* are not part of it
// (the lookup reads them from the GroupDesc when all levels missed).
__device__ __forceinline__ GroupHead load_bucket_and_qhead(const BucketDesc* __restrict__ bp, const QueryHead* __restrict__ qp, uint64_t& bstart,
                                                           uint32_t& bnt)
{
	uint4 bd, a, b;
	asm volatile(
	    "global_load_dwordx4 %0, %3, off\n\t"
	    "global_load_dwordx4 %1, %4, off\n\t"
	    "global_load_dwordx4 %2, %4, off offset:16\n\t"
	    "s_waitcnt vmcnt(0)"
	    : "=&v"(bd), "=&v"(a), "=&v"(b)
	    : "v"(bp), "v"(qp)
	    : "memory");
	bstart = (uint64_t)bd.x | ((uint64_t)bd.y << 32);
	bnt = bd.z;
	return unpack_query_head(a, b);
}

// ---- absent-key filter ---------------------------------------------------------------------------------------------------------
// A blocked Bloom filter over the keys (plus the few non-keys the reference's scan can still find, see k_filter_junctions): ONE
// 64-bit word per probe, five bits of it set per key.  The word of a key lies in its MPHF group's own region -- words
// [local_prefix * bpk / 64, + (size0 / 2) * bpk / 64) -- so that the filter slice of a partition bin is as cache-resident as the
// bin's MPHF lines.  No counterpart in the reference: a k-mer the filter rejects is one the reference's lookup would have carried
// through 4 levels on average, a position read and a 2^b window before answering -1 as well.
__device__ __forceinline__ uint64_t filter_hash(uint64_t lo, uint64_t hi)
{
	uint64_t h = lo ^ (hi * 0x9e3779b97f4a7c15ull);
	h ^= h >> 33;
	h *= 0xff51afd7ed558ccdull;
	h ^= h >> 29;
	h *= 0xc4ceb9fe1a85ec53ull;
	return h ^ (h >> 32);
}
template <int KW> __device__ __forceinline__ uint64_t filter_hash(Kmer<KW> x);
template <> __device__ __forceinline__ uint64_t filter_hash<1>(Kmer<1> x) { return filter_hash(x.lo, 0); }
template <> __device__ __forceinline__ uint64_t filter_hash<2>(Kmer<2> x) { return filter_hash(x.lo, x.hi); }
__device__ __forceinline__ uint64_t filter_mask(uint64_t h)
{
	const uint32_t u = (uint32_t)(h >> 32);
	return (1ull << (u & 63u)) | (1ull << ((u >> 6) & 63u)) | (1ull << ((u >> 12) & 63u)) | (1ull << ((u >> 18) & 63u)) | (1ull << ((u >> 24) & 63u));
}
__device__ __forceinline__ uint64_t filter_word(uint64_t local_prefix, uint64_t size0, uint32_t bpk, uint64_t h)
{
	const uint64_t first = (local_prefix * bpk) >> 6;
	uint64_t words = ((size0 >> 1) * bpk) >> 6;  // size0 / 2 = the group's key count rounded up to 32: a few words shared with the next group
	words = words ? words : 1;
	return first + (((h & 0xffffffffull) * words) >> 32);  // a group holds fewer than 2^32 keys
}

// ---- fat level-0 lines -------------------------------------------------------------------------------------------------------------
// With DeviceIndexView::fat the lines of LEVEL 0 are 64 bytes: [u32 rank | 96 level-0 bits | 6 x u64 of the absent-key filter], and
// the filter bits of a k-mer sit in the fat line its level-0 hash selects.  About 48 keys hash into the 96 bits of a line (gamma = 2),
// so the six words give each of them the 8 bits of the separate filter (one word per key, five bits of it).  A query then gets the
// filter test AND the level-0 probe from ONE 64-byte line -- one L1 -> L2 request instead of two for every k-mer, indexed or not
// (the separate filter word cost the absent half of a 50/50 batch its only request and the indexed half a wasted one).  The deeper
// levels keep 16-byte lines behind the group's fat lines.  One unit of line_base = 16 bytes in both layouts.
__device__ __forceinline__ uint32_t fat_lines_of(uint64_t size0) { return ((uint32_t)(size0 >> 5) + kLineData32 - 1) / kLineData32; }
constexpr int kFatUnits = 4;       // 16-B units per fat line
constexpr int kFatFilterWords = 6;  // u64 filter words per fat line
__device__ __forceinline__ uint32_t fat_filter_word(uint64_t fh) { return (uint32_t)(((fh & 0xffffffffull) * (uint64_t)kFatFilterWords) >> 32); }
__device__ __forceinline__ uint32_t rec_filter_word(uint64_t fh) { return (uint32_t)(fh >> 31) & 1u; }  // (bit 31: the mask uses bits 32..61)

// ---- level-0 records (r04) ------------------------------------------------------------------------------------------------------------
// With DeviceIndexView::rec (implies fat) a level-0 unit of 64 bytes covers 32 level-0 bits instead of 96 and carries, besides the
// filter, the POSITION FIELDS of the keys it ranks:
//   u32[0] rank before the record | u32[1] 32 level-0 bits | u32[2..11] position fields, nbits each, field j = the record's j-th set
//   bit (a copy of positions[rank + j]; the first rec_fields(nbits) of them) | u64[6..7] two words of the absent-key filter.
// About 16 keys hash into the 32 bits (gamma = 2) and ~10 of them are ranked there; 320 bits hold 22 fields of 14 bits.  A level-0
// key -- 61 % of all keys -- then costs ONE 64-byte sector for filter test, level-0 probe, rank AND position: one request to the
// fabric and one dependent round trip fewer than with the field in the positions array (whose sector is somewhere else; the
// `positions` array stays complete: deeper levels, fields beyond a record's capacity, export and the builders read it).  The slot
// kernel runs at the chip's rate of random fabric requests (~5e10/s, profiles/r04_*): what it saves is requests.  Costs 64 B per
// 16 keys (32 bits per key) instead of 64 B per 48: chosen at build time while the index stays a small share of the card's HBM.
constexpr int kRecUnits = 4;       // 16-B units per record
constexpr int kRecPosWord = 2;     // first u32 of the position area
constexpr int kRecPosBits = 320;   // ten u32
constexpr int kRecFilterWord = 6;  // first u64 of the two filter words
__host__ __device__ __forceinline__ uint32_t rec_fields(uint32_t nbits)
{
	const uint32_t f = nbits ? (uint32_t)kRecPosBits / nbits : 0u;
	return f > 32u ? 32u : f;
}
// 64-B level-0 units of a group: fat lines of 96 bits, or records of 32
__host__ __device__ __forceinline__ uint64_t level0_units64(uint64_t size0, bool rec)
{
	const uint64_t w0 = size0 >> 5;
	return rec ? w0 : (w0 + kLineData32 - 1) / kLineData32;
}
// ---- cross-paired level-0 blocks (r04) --------------------------------------------------------------------------------------------
// What the slot kernel is bound by is the NUMBER of requests its L2 misses send to the fabric: 1.66e8 per 1e8 queries at 4.9e10/s --
// the rate at which the chip serves random requests of any size up to a 128-byte line (tools/sector_probe: one 16-B load per random
// line 49.4e9/s; one load that straddles the line's two 64-B sectors 49.5e9/s; two back-to-back loads to the two sectors 41-56e9/s
// and ONE request at L1 and at L2; a second sector asked for after the first has arrived: a second request).  A batch touches
// nearly every sector of the index, so asking for whole 128-B lines halves the requests.  The 64-B level-0 units (fat lines and
// records alike) are therefore stored in 128-B blocks of two, CROSSED: the head of unit u (and a record's fields) in half u & 1 of
// block u >> 1, its filter words in the OTHER half, where the partner's head is.  A query's two loads -- head and filter word, issued
// back to back as before -- then always name both sectors of one line: one request brings the 128 bytes, and the partner unit's
// queries find their halves in the L2.  Same bytes, same instructions; blocks are 128-B aligned (GroupDesc::line_base is a
// multiple of 8 units).
constexpr int kBlockUnits = 8;  // 16-B units per 128-B block
__host__ __device__ __forceinline__ uint64_t level0_block_units(uint64_t size0, bool rec) { return ((level0_units64(size0, rec) + 1) / 2) * kBlockUnits; }
__device__ __forceinline__ const uint4* level0_head(const uint4* __restrict__ group_lines, uint32_t u)
{
	return group_lines + (size_t)(u >> 1) * kBlockUnits + (u & 1u) * 4u;
}
// first filter word of unit u: behind the PARTNER's head (16 B of a fat line, 48 B of a record)
__device__ __forceinline__ const uint64_t* level0_filter(const uint4* __restrict__ group_lines, uint32_t u, bool rec)
{
	return reinterpret_cast<const uint64_t*>(group_lines + (size_t)(u >> 1) * kBlockUnits + ((u & 1u) ^ 1u) * 4u) + (rec ? kRecFilterWord : 2);
}

// head (8 B: rank, bits) and one filter word (8 B) of a record in ONE round trip
__device__ __forceinline__ void load_rec_head_and_filter(const uint4* __restrict__ recp, const uint64_t* __restrict__ fp, uint2& head, uint64_t& fw)
{
	uint2 f;
	asm volatile(
	    "global_load_dwordx2 %0, %2, off\n\t"
	    "global_load_dwordx2 %1, %3, off\n\t"
	    "s_waitcnt vmcnt(0)"
	    : "=&v"(head), "=&v"(f)
	    : "v"(recp), "v"(fp)
	    : "memory");
	fw = (uint64_t)f.x | ((uint64_t)f.y << 32);
}
// field j of a record (j < rec_fields(nbits)): the 8-B window starts at most 31 bits before the field and ends inside the record
// (u32 11 is followed by the filter words)
__device__ __forceinline__ uint32_t read_rec_field(const uint4* __restrict__ recp, uint32_t j, uint32_t nbits)
{
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
	const uint32_t o = j * nbits;
	const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint32_t*>(recp) + kRecPosWord + (o >> 5));
	const uint64_t v = ((uint64_t)w.x | ((uint64_t)w.y << 32)) >> (o & 31u);
	return (uint32_t)(v & ((nbits >= 32) ? 0xffffffffull : ((1ull << nbits) - 1ull)));
}

// head (16 B) and one filter word (8 B) of a fat line in ONE round trip: the compiler would sink the second load behind the test of
// the first
__device__ __forceinline__ void load_fat_head_and_filter(const uint4* __restrict__ line, const uint64_t* __restrict__ fp, uint4& head, uint64_t& fw)
{
	// (default cache policy: `nt` makes the slot kernel 40 % slower -- the bin's lines no longer stay in the Infinity Cache -- and `sc1`
	// 11 %: the two loads of one line then cost two L2 requests; profiles/r03_exp_fat_line_cache_policy_c4.txt)
	uint2 f;
	asm volatile(
	    "global_load_dwordx4 %0, %2, off\n\t"
	    "global_load_dwordx2 %1, %3, off\n\t"
	    "s_waitcnt vmcnt(0)"
	    : "=&v"(head), "=&v"(f)
	    : "v"(line), "v"(fp)
	    : "memory");
	fw = (uint64_t)f.x | ((uint64_t)f.y << 32);
}

// mphf::lookup restricted to the level bit arrays (bbhash.h:561-566): probe levels until a set bit is found and rank it.
// Levels 0..2 use the sizes that came with the group head; deeper levels read their size on demand.
// use_filter (fat layout only): the absent-key filter is consulted with the level-0 line; `rejected` = the filter says "not indexed"
// (kNotFound is returned as well: nothing else is looked at).
// field (records only): the position field of the returned rank when the record carries it, else kNoField.
constexpr uint32_t kNoField = 0xffffffffu;  // (a field has at most 32 bits, but 2^32 - 1 is no position of a bucket: nt < 2^32)
template <int KW>
__device__ __forceinline__ uint64_t mphf_lookup_levels(const DeviceIndexView& v, const GroupDesc* __restrict__ gd, const GroupHead& gh, Kmer<KW> x,
                                                        bool use_filter, bool& rejected, uint32_t& field)
{
	LevelHasher hs(x.lo);
	const uint4* __restrict__ lines = reinterpret_cast<const uint4*>(v.lines) + gh.line_base;
	uint64_t rank;
	uint64_t off = 0;
	rejected = false;
	field = kNoField;
	if (v.rec) {
		const uint64_t b0 = reduce_level(hs.next(0), gh.size0);
		const uint32_t w32 = (uint32_t)(b0 >> 5), bi = (uint32_t)b0 & 31u;
		const uint4* recp = level0_head(lines, w32);
		uint2 H;
		if (use_filter) {
			const uint64_t fh = filter_hash<KW>(x), fm = filter_mask(fh);
			uint64_t fw;
			load_rec_head_and_filter(recp, level0_filter(lines, w32, true) + rec_filter_word(fh), H, fw);
			if ((fw & fm) != fm) {
				rejected = true;
				return kNotFound;
			}
		} else {
			H = *reinterpret_cast<const uint2*>(recp);
		}
		if ((H.y >> bi) & 1u) {
			const uint32_t j = (uint32_t)__popc(H.y & ((1u << bi) - 1u));
			if (j < rec_fields(gh.nbits))
				field = read_rec_field(recp, j, gh.nbits);  // the sector is in flight or in the L1 / L2 it just came through
			return (uint64_t)(H.x + j);
		}
		lines += (size_t)level0_block_units(gh.size0, true);  // the deeper levels: 16-B lines over the bits behind level 0
	} else if (v.fat) {
		const uint64_t b0 = reduce_level(hs.next(0), gh.size0);
		const uint32_t w32 = (uint32_t)(b0 >> 5), fl = w32 / kLineData32;
		const uint4* fat = level0_head(lines, fl);
		uint4 L;
		if (use_filter) {
			const uint64_t fh = filter_hash<KW>(x), fm = filter_mask(fh);
			uint64_t fw;
			load_fat_head_and_filter(fat, level0_filter(lines, fl, false) + fat_filter_word(fh), L, fw);
			if ((fw & fm) != fm) {
				rejected = true;
				return kNotFound;
			}
		} else {
			L = *fat;
		}
		if (mphf_probe_line(L, w32 - fl * kLineData32, (uint32_t)b0 & 31u, rank))
			return rank;
		lines += (size_t)level0_block_units(gh.size0, false);  // the deeper levels: 16-B lines over the bits behind level 0
	} else {
		if (mphf_probe(lines, reduce_level(hs.next(0), gh.size0), rank))
			return rank;
		off = gh.size0;
	}
	if (mphf_probe(lines, off + reduce_level(hs.next(1), gh.size1), rank))
		return rank;
	off += gh.size1;
	if (mphf_probe(lines, off + reduce_level(hs.next(2), gh.size2), rank))
		return rank;
	off += gh.size2;
	// Deeper levels two at a time: both sizes come with one 16-B load (size[lv], lv odd, is 16-B aligned inside GroupDesc)
	// and both lines are requested before either is inspected -- two dependent round trips per pair instead of four.
	static_assert((offsetof(GroupDesc, size) + 8 * 3) % 16 == 0 && kLevels % 2 == 0, "size[3] must be 16-B aligned");
	for (int lv = 3; lv < kLevels; lv += 2) {
		const uint4 sz = *reinterpret_cast<const uint4*>(&gd->size[lv]);  // size[lv], size[lv + 1] (unused for lv = 15)
		const uint64_t size_a = (uint64_t)sz.x | ((uint64_t)sz.y << 32), size_b = (uint64_t)sz.z | ((uint64_t)sz.w << 32);
		const uint64_t bit_a = off + reduce_level(hs.next(lv), size_a);
		const bool two = lv + 1 < kLevels;
		const uint64_t bit_b = two ? off + size_a + reduce_level(hs.next(lv + 1), size_b) : bit_a;
		const uint32_t wa = (uint32_t)(bit_a >> 5), wb = (uint32_t)(bit_b >> 5);
		const uint32_t la = wa / kLineData32, lb = wb / kLineData32;
		const uint4 A = lines[la], B = lines[lb];
		if (mphf_probe_line(A, wa - la * kLineData32, (uint32_t)bit_a & 31u, rank))
			return rank;
		if (two && mphf_probe_line(B, wb - lb * kLineData32, (uint32_t)bit_b & 31u, rank))
			return rank;
		off += size_a + size_b;
	}
	return kNotFound;
}
template <int KW>
__device__ __forceinline__ uint64_t mphf_lookup_levels(const DeviceIndexView& v, const GroupDesc* __restrict__ gd, const GroupHead& gh, Kmer<KW> x,
                                                        bool use_filter, bool& rejected)
{
	uint32_t field;
	return mphf_lookup_levels<KW>(v, gd, gh, x, use_filter, rejected, field);
}
template <int KW>
__device__ __forceinline__ uint64_t mphf_lookup_levels(const DeviceIndexView& v, const GroupDesc* __restrict__ gd, const GroupHead& gh, Kmer<KW> x)
{
	bool rejected;
	return mphf_lookup_levels<KW>(v, gd, gh, x, false, rejected);
}

// the filter bits of k-mer x in the fat / record layout: the word of the 64-B level-0 unit its level-0 hash selects (build:
// k_filter_keys / _junctions)
template <int KW>
__device__ __forceinline__ unsigned long long* fat_filter_slot(const DeviceIndexView& v, uint64_t line_base, uint64_t size0, Kmer<KW> x, uint64_t fh)
{
	const uint64_t b0 = reduce_level(hash_bis(x.lo, kSeedH0), size0);
	const uint4* group_lines = reinterpret_cast<const uint4*>(v.lines) + line_base;
	const uint32_t u = v.rec ? (uint32_t)(b0 >> 5) : (uint32_t)(b0 >> 5) / kLineData32;
	const uint64_t* w = level0_filter(group_lines, u, v.rec) + (v.rec ? rec_filter_word(fh) : fat_filter_word(fh));
	return reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(w));
}

// ---- the fallback table as a hash table (r04) -----------------------------------------------------------------------------------------
// _final_hash of the reference is an unordered_map per MPHF (bbhash.h:781).  The sorted (group, key, input order) arrays keep every
// entry -- the export and the image are made of them -- but a lookup in them is a binary search: ~20 dependent loads where low
// complexity puts millions of keys behind a few minimizers.  Each group's entries therefore sit once more in an open-addressing table
// of its own: power-of-two slots, at least 1.5 per entry, linear probing from mix(key); a slot = [key (KW words) | rank], 16 bytes
// for 64-bit k-mers, 32 (one word unused) for 128-bit ones; rank = kNotFound marks an empty slot.  Only the entry that the map's
// "last assignment wins" (bbhash.h:724) leaves behind -- the LAST of a run of equal keys -- is in the table.  One or two loads.
template <int KW> struct FbSlot { static constexpr int words = KW == 1 ? 2 : 4; };
__device__ __forceinline__ uint64_t fb_mix(uint64_t lo, uint64_t hi)
{
	uint64_t h = lo ^ (hi * 0x9e3779b97f4a7c15ull);
	h ^= h >> 31;
	h *= 0x7fb5d329728ea185ull;
	h ^= h >> 27;
	h *= 0x81dadef4bc2dd44dull;
	return h ^ (h >> 33);
}
template <int KW>
__device__ __forceinline__ uint64_t fallback_find(const DeviceIndexView& v, uint32_t group, Kmer<KW> x)
{
	const uint64_t dir = v.fb_dir[group];
	if (dir == 0)
		return kNotFound;
	const uint64_t first = dir & 0x00ffffffffffffffull, mask = (1ull << (dir >> 56)) - 1ull;
	uint64_t at = fb_mix(x.lo, kmer_hi(x)) & mask;
	for (;;) {  // (ends: the table is never full)
		const uint64_t* sl = v.fb_table + (first + at) * FbSlot<KW>::words;
		if constexpr (KW == 1) {
			const ulonglong2 e = *reinterpret_cast<const ulonglong2*>(sl);
			if (e.y == kNotFound)
				return kNotFound;
			if (e.x == x.lo)
				return e.y;
		} else {
			const ulonglong2 e = reinterpret_cast<const ulonglong2*>(sl)[0], f = reinterpret_cast<const ulonglong2*>(sl)[1];
			if (f.x == kNotFound)
				return kNotFound;
			if (e.x == x.lo && e.y == x.hi)
				return f.x;
		}
		at = (at + 1) & mask;
	}
}

// mphf::lookup (bbhash.h:561-577): levels first, then the group's fallback table.
template <int KW>
__device__ __forceinline__ uint64_t mphf_lookup(const DeviceIndexView& v, const GroupDesc* __restrict__ gd, const GroupHead& gh, Kmer<KW> x,
                                                 bool use_filter, uint32_t& field)
{
	bool rejected;
	const uint64_t r = mphf_lookup_levels<KW>(v, gd, gh, x, use_filter, rejected, field);
	if (r != kNotFound || rejected)
		return r;
	const uint32_t fb_count = gd->fb_count;  // from the descriptor: the compact query head does not carry the fallback fields
	if (fb_count == 0)
		return r;
	// _final_hash.find(elem), bbhash.h:567-575.
	if (v.fb_dir)
		return fallback_find<KW>(v, (uint32_t)(gd - v.groups), x);
	// (no table: the sorted entries themselves.)  The table is sorted by (group, key, input order) and keeps every
	// occurrence of a duplicated key; the map's "last assignment wins" (bbhash.h:724) = the LAST entry of an equal run.
	const uint64_t fb_start = gd->fb_start;
	uint64_t lo = fb_start, hi = lo + fb_count;
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
template <int KW>
__device__ __forceinline__ uint64_t mphf_lookup(const DeviceIndexView& v, const GroupDesc* __restrict__ gd, const GroupHead& gh, Kmer<KW> x,
                                                 bool use_filter = false)
{
	uint32_t field;
	return mphf_lookup<KW>(v, gd, gh, x, use_filter, field);
}

// bool_to_int, blight.cpp:473-482: nbits-wide little-endian field at bit posstart + h*nbits.
// Physical layout: a field holds the key's position in its bucket >> (b - hint) -- `hint` more low bits than the reference keeps
// (DeviceIndexView::hint, 0 = the reference's field): field >> hint is the reference's value (the 2^b window), the low `hint` bits name
// the 2^(b-hint) sub-window inside it that holds the key itself.
__device__ __forceinline__ uint32_t read_position_field(const uint64_t* __restrict__ positions, uint64_t posstart, uint32_t nbits, uint64_t h)
{
	// nbits <= 32 and a dword-aligned 8-B window starts at most 31 bits before the field: ONE global_load_dwordx2
	// (multi-dword loads need dword alignment only) instead of two aligned 8-B words
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
	const uint64_t o = posstart + h * (uint64_t)nbits;
	const u32x2* wp = reinterpret_cast<const u32x2*>(reinterpret_cast<const uint32_t*>(positions) + (o >> 5));
	u32x2 w;
	if constexpr ((BLMX_EXP_NT >> 1) & 1)
		w = __builtin_nontemporal_load(wp);
	else
		w = *wp;
	const uint32_t s = (uint32_t)o & 31u;
	const uint64_t v = ((uint64_t)w.x | ((uint64_t)w.y << 32)) >> s;
	return (uint32_t)(v & ((nbits >= 32) ? 0xffffffffull : ((1ull << nbits) - 1ull)));
}
// the reference's value: first position of the 2^b window
__device__ __forceinline__ uint32_t read_position(const uint64_t* __restrict__ positions, uint64_t posstart, uint32_t nbits,
                                                  uint64_t h, int hint, int b)
{
	return (read_position_field(positions, posstart, nbits, h) >> hint) << b;
}
// Extra low bits a field keeps for windows of 2^logp positions.  Only for the wide windows: at b = 8 the 256 steps of the scan are most
// of the query kernel's instructions and the kernel is issue-bound (C5: 4.39 -> 3.82 ms); at b <= 6 the kernel waits for memory, the
// shorter scan buys nothing (C4: 3.40 -> 3.41 ms) and the wider fields cost 3 bits per key.
__host__ __device__ constexpr int position_hint_bits(int logp) { return logp >= 6 ? 3 : 0; }  // (b = 6: only when asked for, pos_hint = 2)

}  // namespace blmx
