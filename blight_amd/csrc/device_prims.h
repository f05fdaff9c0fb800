// device_prims.h -- gfx950 device primitives of the Blight hot path (k-mer encoding, reverse complement, canonical
// forms, minimizer order, BBHash level hashes).  Integer-only; every function cites the reference routine whose
// results it must reproduce bit for bit (paths relative to the reference repository).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blmx {

constexpr int kLevels = 16;                      // mphf<..., _nb_levels = 16>, bbhash.h:516
constexpr uint64_t kNotFound = ~0ull;            // ULLONG_MAX, bbhash.h:574
constexpr uint64_t kSeedH0 = 0xAAAAAAAA55555555ull;  // bbhash.h:221
constexpr uint64_t kSeedH1 = 0x33333333CCCCCCCCull;  // bbhash.h:227
// One MPHF "line" = 16 bytes = [u32 rank | 3 x u32 level bits]: a probe (bit test + rank) is ONE dwordx4 load.
// (Measured on MI355X: random single 16-B loads run at ~2.4x the line rate of 4 x 16-B loads of a 64-B line once the
// footprint is several GB; the 33 % rank overhead costs 1.1 bit/key.)
constexpr int kLineWords32 = 4;
constexpr int kLineData32 = 3;
constexpr int kLineBytes = 16;

// ---- k-mer value types -------------------------------------------------------------------------------------------
// KW = 1: k <= 32 in one u64 (the reference's shipped kmer_t, kmer.h:18); KW = 2: k <= 63 in (hi, lo) = __uint128_t.
template <int KW> struct Kmer;
template <> struct Kmer<1> {
	uint64_t lo;
	__host__ __device__ __forceinline__ bool operator==(const Kmer& o) const { return lo == o.lo; }
	__host__ __device__ __forceinline__ bool operator<=(const Kmer& o) const { return lo <= o.lo; }
	__host__ __device__ __forceinline__ bool operator<(const Kmer& o) const { return lo < o.lo; }
};
template <> struct Kmer<2> {
	uint64_t lo, hi;
	__host__ __device__ __forceinline__ bool operator==(const Kmer& o) const { return lo == o.lo && hi == o.hi; }
	__host__ __device__ __forceinline__ bool operator<=(const Kmer& o) const { return hi < o.hi || (hi == o.hi && lo <= o.lo); }
	__host__ __device__ __forceinline__ bool operator<(const Kmer& o) const { return hi < o.hi || (hi == o.hi && lo < o.lo); }
};

template <int KW> __host__ __device__ __forceinline__ Kmer<KW> make_kmer(uint64_t lo, uint64_t hi);
template <> __host__ __device__ __forceinline__ Kmer<1> make_kmer<1>(uint64_t lo, uint64_t)
{
	Kmer<1> x;
	x.lo = lo;
	return x;
}
template <> __host__ __device__ __forceinline__ Kmer<2> make_kmer<2>(uint64_t lo, uint64_t hi)
{
	Kmer<2> x;
	x.lo = lo;
	x.hi = hi;
	return x;
}
__host__ __device__ __forceinline__ uint64_t kmer_hi(const Kmer<1>&) { return 0; }
__host__ __device__ __forceinline__ uint64_t kmer_hi(const Kmer<2>& x) { return x.hi; }

// nuc2int, kmer.h:56-69: A0 C1 T2 G3 = (c >> 1) & 3, either case; returns 4 for anything else.
__device__ __forceinline__ uint32_t nuc_code(uint8_t c)
{
	const uint32_t d = (uint32_t)c - 65u;
	const bool ok = d <= 51u && ((0x0008004500080045ull >> d) & 1ull);
	return ok ? ((c >> 1) & 3u) : 4u;
}

// revhash(uint32_t), kmer.h:102-108.  The minimizer order compares the result as SIGNED int32.
__device__ __forceinline__ int32_t revhash32(uint32_t x)
{
	x = ((x >> 16) ^ x) * 0x2c1b3c6du;
	x = ((x >> 16) ^ x) * 0x297a2d39u;
	x = ((x >> 16) ^ x);
	return (int32_t)x;
}

// Inverse permutation of revhash32: x ^= x >> 16 is an involution, the multipliers are inverted modulo 2^32
// (0x297a2d39 * 0x0cf0b109 = 0x2c1b3c6d * 0x64ea2d65 = 1 mod 2^32).
__device__ __forceinline__ uint32_t unrevhash32(uint32_t x)
{
	x = ((x >> 16) ^ x) * 0x0cf0b109u;
	x = ((x >> 16) ^ x) * 0x64ea2d65u;
	x = ((x >> 16) ^ x);
	return x;
}

// Reverse all 32 2-bit digits of a u64 and complement each (XOR 0b10): v_bfrev + pair swap instead of the byte-swap
// ladder of rcb(), kmer.h:218-232 -- same function.
__device__ __forceinline__ uint64_t rc_digits64(uint64_t x)
{
	uint64_t r = __brevll(x);
	r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
	return r ^ 0xAAAAAAAAAAAAAAAAull;
}

template <int KW> __device__ __forceinline__ Kmer<KW> kmer_rc(Kmer<KW> x, int k);
template <> __device__ __forceinline__ Kmer<1> kmer_rc<1>(Kmer<1> x, int k)  // rcb(uint64_t, n), kmer.h:218-232
{
	Kmer<1> r;
	r.lo = rc_digits64(x.lo) >> (64 - 2 * k);
	return r;
}
template <> __device__ __forceinline__ Kmer<2> kmer_rc<2>(Kmer<2> x, int k)  // rcb(__uint128_t, n), kmer.h:191-214
{
	const uint64_t nh = rc_digits64(x.lo), nl = rc_digits64(x.hi);  // full 64-digit reverse: halves swap
	const int s = 128 - 2 * k;                                       // 2..126 for k in 1..63
	Kmer<2> r;
	if (s >= 64) {
		r.lo = nh >> (s - 64);
		r.hi = 0;
	} else {
		r.lo = (nl >> s) | (nh << (64 - s));
		r.hi = nh >> s;
	}
	return r;
}

// low 32 bits of (x >> s)
__device__ __forceinline__ uint32_t kmer_shr32(Kmer<1> x, int s) { return (uint32_t)(x.lo >> s); }
__device__ __forceinline__ uint32_t kmer_shr32(Kmer<2> x, int s)
{
	if (s >= 64)
		return (uint32_t)(x.hi >> (s - 64));
	return (uint32_t)((x.lo >> s) | (s ? (x.hi << (64 - s)) : 0ull));
}

// Minimizer of a k-mer = the parity-canonical m-mer with the smallest SIGNED revhash (minimizer_naive with fix P1,
// kmer.h:791-810; ParityCanonical kmer.h:475-487).  The reverse-complement m-mers are read from the reverse-complement
// k-mer (m-mer i of x pairs with m-mer k-m-i of rc(x)), so no per-window rcb is needed.  Ties imply equal values
// (revhash is a bijection), so the scan order is irrelevant.
//
// Only the smallest hash is tracked (one v_min_i32 per m-mer); revhash is a bijection on 32 bits, so the m-mer it belongs to
// is recovered once at the end with the inverse permutation instead of being carried through the loop.  The loop is fully
// unrolled over the window index with a wave-uniform early exit: the k-mers live in 32-bit words, the reverse complement is
// shifted ONCE so that its window for index i starts at the compile-time bit 2 (WMAX - 1 - i), and every window is then one
// v_alignbit with immediate operands (the runtime-offset version paid 64-bit shifts -- and branches for 128-bit k-mers --
// per window: 19 instructions per m-mer instead of 14; the minimizer is what bounds the partition pass).
template <int KW> __device__ __forceinline__ uint32_t kmer_minimizer(Kmer<KW> x, Kmer<KW> xr, int k, int m)
{
	constexpr int NW = 2 * KW;     // 32-bit words of a k-mer
	constexpr int WMAX = 32 * KW;  // k - m + 1 <= 32 KW windows
	const uint32_t mask = (m >= 16) ? 0xffffffffu : ((1u << (2 * m)) - 1u);
	const int w = k - m;  // last window index
	uint32_t X[NW + 1], R[NW + 1];
	X[0] = (uint32_t)x.lo;
	X[1] = (uint32_t)(x.lo >> 32);
	// xs = xr << 2 (WMAX - 1 - w): at most 2 (WMAX - 1 + m) <= 64 KW + 30 bits
	const int s = 2 * (WMAX - 1 - w);
	if constexpr (KW == 1) {
		X[2] = 0;
		const uint64_t lo = xr.lo << s, hi = s ? (xr.lo >> (64 - s)) : 0ull;
		R[0] = (uint32_t)lo;
		R[1] = (uint32_t)(lo >> 32);
		R[2] = (uint32_t)hi;
	} else {
		X[2] = (uint32_t)x.hi;
		X[3] = (uint32_t)(x.hi >> 32);
		X[4] = 0;
		// 128-bit value shifted left by s in [0, 126] into 160 bits (the top word only ever receives m-mer bits)
		uint64_t a0, a1, a2;  // bits 0-63, 64-127, 128-191
		if (s >= 64) {
			const int t = s - 64;
			a0 = 0;
			a1 = xr.lo << t;
			a2 = (xr.hi << t) | (t ? (xr.lo >> (64 - t)) : 0ull);
		} else {
			a0 = xr.lo << s;
			a1 = (xr.hi << s) | (s ? (xr.lo >> (64 - s)) : 0ull);
			a2 = s ? (xr.hi >> (64 - s)) : 0ull;
		}
		R[0] = (uint32_t)a0;
		R[1] = (uint32_t)(a0 >> 32);
		R[2] = (uint32_t)a1;
		R[3] = (uint32_t)(a1 >> 32);
		R[4] = (uint32_t)a2;
	}
	// The parity of every window from ONE prefix-parity word (r04): p = prefix XOR of the k-mer's bits (bit j = parity of bits 0..j),
	// q = p ^ (p << 2m) holds at bit j the parity of bits (j - 2m, j], and shifted down by 2m - 1 the parity of window i sits at the
	// compile-time bit 2i: one v_bfe_i32 per window instead of v_and + v_bcnt + v_bfe_i32 (a fifth of a window's 12.5 instructions),
	// for ~25 (64-bit) / ~50 (128-bit) instructions once per k-mer.
	uint32_t PQ[NW];
	{
		const int sp = 2 * m;  // 2 .. 32
		uint64_t p0 = x.lo;
		p0 ^= p0 << 1;
		p0 ^= p0 << 2;
		p0 ^= p0 << 4;
		p0 ^= p0 << 8;
		p0 ^= p0 << 16;
		p0 ^= p0 << 32;
		if constexpr (KW == 1) {
			const uint64_t q = (p0 ^ (p0 << sp)) >> (sp - 1);
			PQ[0] = (uint32_t)q;
			PQ[1] = (uint32_t)(q >> 32);
		} else {
			uint64_t p1 = x.hi;
			p1 ^= p1 << 1;
			p1 ^= p1 << 2;
			p1 ^= p1 << 4;
			p1 ^= p1 << 8;
			p1 ^= p1 << 16;
			p1 ^= p1 << 32;
			p1 ^= 0ull - (p0 >> 63);  // the parity of the whole low word carries into every bit of the high one
			const uint64_t q0 = p0 ^ (p0 << sp), q1 = p1 ^ ((p1 << sp) | (p0 >> (64 - sp)));
			const uint64_t r0 = (q0 >> (sp - 1)) | (q1 << (65 - sp)), r1 = q1 >> (sp - 1);
			PQ[0] = (uint32_t)r0;
			PQ[1] = (uint32_t)(r0 >> 32);
			PQ[2] = (uint32_t)r1;
			PQ[3] = (uint32_t)(r1 >> 32);
		}
	}
	int32_t best_h = 0x7fffffff;
	// pick = all ones when the forward strand has odd parity, then one v_bfi_b32 selects the strand and one v_bfe_u32 masks and drops
	// the low bit: no v_cmp / v_cndmask pair, no VCC hazard
	auto window = [&](uint32_t f, uint32_t r, uint32_t pick) {
		uint32_t sel;
		asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(pick), "v"(f), "v"(r));
		best_h = min(best_h, revhash32(__builtin_amdgcn_ubfe(sel, 1u, (uint32_t)(2 * m - 1))));
	};
	// Groups of four windows whose word indices and shifts are immediates, behind ONE wave-uniform branch per group (a branch
	// per window cost more than it saved: taken branches stall the wave's instruction stream) ...
#pragma unroll
	for (int g = 0; g < WMAX / 4; g++) {
		if (4 * g + 3 <= w) {
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int of = 2 * (4 * g + j), orr = 2 * (WMAX - 1 - (4 * g + j));
				window((of & 31) ? __builtin_amdgcn_alignbit(X[(of >> 5) + 1], X[of >> 5], of & 31) : X[of >> 5],
				       (orr & 31) ? __builtin_amdgcn_alignbit(R[(orr >> 5) + 1], R[orr >> 5], orr & 31) : R[orr >> 5],
				       (uint32_t)__builtin_amdgcn_sbfe(PQ[of >> 5], (uint32_t)(of & 31), 1u));
			}
		}
	}
	// ... and the last (w + 1) mod 4 windows with runtime offsets (their parity the plain way: v_and + v_bcnt + v_bfe_i32)
	for (int i = (w + 1) & ~3; i <= w; i++) {
		const uint32_t f = kmer_shr32(x, 2 * i);
		window(f, kmer_shr32(xr, 2 * (w - i)), (uint32_t)__builtin_amdgcn_sbfe(__popc(f & mask), 0u, 1u));
	}
	return unrevhash32((uint32_t)best_h);
}

// ---- 2-bit packed sequences: nt p lives in bits 63-2(p%32) .. 62-2(p%32) of word p/32 (first nt most significant) ----
__device__ __forceinline__ uint64_t funnel_left(uint64_t hi, uint64_t lo, int s)  // top 64 bits of (hi:lo) << s, s in 0..63
{
	return s ? ((hi << s) | (lo >> (64 - s))) : hi;
}

template <int KW> __device__ __forceinline__ Kmer<KW> packed_kmer(const uint64_t* __restrict__ seq, uint64_t p, int k);
template <> __device__ __forceinline__ Kmer<1> packed_kmer<1>(const uint64_t* __restrict__ seq, uint64_t p, int k)
{
	const uint64_t w = p >> 5;
	const int s = 2 * (int)(p & 31);
	Kmer<1> r;
	r.lo = funnel_left(seq[w], seq[w + 1], s) >> (64 - 2 * k);
	return r;
}
template <> __device__ __forceinline__ Kmer<2> packed_kmer<2>(const uint64_t* __restrict__ seq, uint64_t p, int k)
{
	const uint64_t w = p >> 5;
	const int s = 2 * (int)(p & 31);
	const uint64_t a = seq[w], b = seq[w + 1], c = seq[w + 2];
	const uint64_t v0 = funnel_left(a, b, s), v1 = funnel_left(b, c, s);
	const int sh = 128 - 2 * k;  // 2..62 for k in 33..63; up to 126 for small k
	Kmer<2> r;
	if (sh >= 64) {
		r.lo = v0 >> (sh - 64);
		r.hi = 0;
	} else {
		r.lo = (v1 >> sh) | (v0 << (64 - sh));
		r.hi = v0 >> sh;
	}
	return r;
}

template <int KW> __device__ __forceinline__ Kmer<KW> kmer_canon(Kmer<KW> x, int k)  // min_k, blight.cpp:86-91
{
	const Kmer<KW> r = kmer_rc<KW>(x, k);
	return (x <= r) ? x : r;
}

// ---- BBHash level hashes ---------------------------------------------------------------------------------------------
// SingleHashFunctor::hash_bis, bbhash.h:172-185
__device__ __forceinline__ uint64_t hash_bis(uint64_t key, uint64_t seed)
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

// XorshiftHashFunctors::iter (bbhash.h:241-250) as a stateful stream: level 0 -> h0, level 1 -> h1, level >= 2 ->
// xorshift128+ (bbhash.h:233-239) seeded with (h0, h1).  128-bit keys are hashed by their LOW 64 bits only
// (bbhash.h:160-170: the __int128 overload is never selected -- SURVEY C-7).
struct LevelHasher {
	uint64_t s0, s1;
	uint64_t key;
	__device__ __forceinline__ explicit LevelHasher(uint64_t k) : s0(0), s1(0), key(k) {}
	__device__ __forceinline__ uint64_t next(int level)
	{
		if (level == 0) {
			s0 = hash_bis(key, kSeedH0);
			return s0;
		}
		if (level == 1) {
			s1 = hash_bis(key, kSeedH1);
			return s1;
		}
		uint64_t a = s0;
		const uint64_t b = s1;
		s0 = b;
		a ^= a << 23;
		s1 = a ^ b ^ (a >> 17) ^ (b >> 26);
		return s1 + b;
	}
};

// fastmod64, bbhash.h:660-662
__device__ __forceinline__ uint64_t reduce_level(uint64_t hash, uint64_t size) { return __umul64hi(hash, size); }

// ---- MurmurHash3_x64_128 (Appleby; reference MurmurHash3.cpp:255-334) ---------------------------------------------------
// Dead code in the reference (linked by nothing, the MPHF hashes are hash_bis/xorshift128+), shipped as a
// device utility because the task names it; pinned by the known answers of SURVEY App. B-4 and, in the tests, against the
// reference's own MurmurHash3.cpp.  `p` may be unaligned: bytes are assembled little-endian.
__device__ __forceinline__ uint64_t murmur_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t murmur_fmix64(uint64_t k)  // MurmurHash3.cpp:81-90
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdull;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ull;
	k ^= k >> 33;
	return k;
}
__device__ __forceinline__ uint64_t murmur_le64(const uint8_t* p, int nbytes)  // nbytes <= 8
{
	uint64_t v = 0;
	for (int i = 0; i < nbytes; i++)
		v |= (uint64_t)p[i] << (8 * i);
	return v;
}
__device__ inline void murmur3_x64_128(const uint8_t* p, int len, uint32_t seed, uint64_t out[2])
{
	const uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
	uint64_t h1 = seed, h2 = seed;
	const int nblocks = len / 16;
	for (int i = 0; i < nblocks; i++) {  // body, MurmurHash3.cpp:272-288
		uint64_t k1 = murmur_le64(p + 16 * i, 8), k2 = murmur_le64(p + 16 * i + 8, 8);
		k1 *= c1;
		k1 = murmur_rotl64(k1, 31);
		k1 *= c2;
		h1 ^= k1;
		h1 = murmur_rotl64(h1, 27);
		h1 += h2;
		h1 = h1 * 5 + 0x52dce729;
		k2 *= c2;
		k2 = murmur_rotl64(k2, 33);
		k2 *= c1;
		h2 ^= k2;
		h2 = murmur_rotl64(h2, 31);
		h2 += h1;
		h2 = h2 * 5 + 0x38495ab5;
	}
	const uint8_t* tail = p + 16 * nblocks;  // tail, MurmurHash3.cpp:290-317
	const int rem = len & 15;
	if (rem > 8) {
		uint64_t k2 = murmur_le64(tail + 8, rem - 8);
		k2 *= c2;
		k2 = murmur_rotl64(k2, 33);
		k2 *= c1;
		h2 ^= k2;
	}
	if (rem > 0) {
		uint64_t k1 = murmur_le64(tail, rem > 8 ? 8 : rem);
		k1 *= c1;
		k1 = murmur_rotl64(k1, 31);
		k1 *= c2;
		h1 ^= k1;
	}
	h1 ^= (uint64_t)len;  // finalization, MurmurHash3.cpp:319-331
	h2 ^= (uint64_t)len;
	h1 += h2;
	h2 += h1;
	h1 = murmur_fmix64(h1);
	h2 = murmur_fmix64(h2);
	h1 += h2;
	h2 += h1;
	out[0] = h1;
	out[1] = h2;
}

// ---- wave helpers (wave64) -----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Wave-aggregated append: every lane with pred reserves one slot of *counter; one atomic per wave.
__device__ __forceinline__ uint64_t wave_append_slot(bool pred, unsigned long long* counter)
{
	const uint64_t ballot = __ballot(pred);
	if (ballot == 0)
		return 0;
	const uint32_t lane = lane_id();
	const int leader = __ffsll((unsigned long long)ballot) - 1;
	unsigned long long base = 0;
	if ((int)lane == leader)
		base = atomicAdd(counter, (unsigned long long)__popcll(ballot));
	base = __shfl(base, leader);
	return base + (uint64_t)__popcll(ballot & ((1ull << lane) - 1ull));
}

}  // namespace blmx
