// sector_probe.hip -- micro-benchmark behind DESIGN section 4.1 (r04): what does a RANDOM access cost the MI355X's fabric when a lane
// wants 64 bytes, when it wants both 64-B halves of one 128-B line, and how does the hardware count them?  The Blight slot kernel
// moves its index once per batch in ~1.6e8 random 64-B sector fetches at ~4.9e10 fetches/s -- the same number of fetches per second
// that a streaming copy issues as 128-B requests at 6.3 TB/s.  If a lane can get a whole 128-B line for the price of one request,
// records that pair what a query needs next (MPHF line + position fields) halve the fetch count; if not, they only add bytes.
//
// Every lane reads `per_lane` random 128-B blocks of a table (default 8 GiB, far beyond the 256-MiB Infinity Cache):
//   one16      one 16-B load at block + 0                         (one 64-B sector)
//   straddle   one 16-B load at block + 56                        (ONE instruction touching both sectors)
//   two_insts  16 B at block + 0 and 16 B at block + 64           (two instructions, issued back to back)
//   dependent  16 B at block + 0, then 8 B at block + 64 + f(data) (the second sector only after the first has arrived)
//   pair_lanes lanes 2i / 2i+1 read + 0 / + 64 of the SAME block  (one instruction, two lanes per block; half as many blocks)
//   far_two    16 B at block + 0 and 16 B at ANOTHER random block (two independent sectors: the price of two requests)
// Reported: blocks per second and ms; run under `rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum` for the request
// counts (profiles/r04_sector_probe*.txt).
//
//   hipcc --offload-arch=gfx950 -O3 -o sector_probe tools/sector_probe.hip ;  ./sector_probe [table_GiB=8] [per_lane=8]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                              \
	do {                                                                                   \
		hipError_t e = (x);                                                                \
		if (e != hipSuccess) {                                                             \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                         \
			exit(1);                                                                       \
		}                                                                                  \
	} while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
	x ^= x >> 33;
	x *= 0xff51afd7ed558ccdull;
	x ^= x >> 33;
	x *= 0xc4ceb9fe1a85ec53ull;
	x ^= x >> 33;
	return x;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));

enum { kOne16, kStraddle, kTwoInsts, kDependent, kPairLanes, kFarTwo };

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const uint8_t* __restrict__ table, uint64_t nblocks, int per_lane, uint64_t* __restrict__ sink)
{
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0;
	uint64_t h = mix((MODE == kPairLanes ? (tid >> 1) : tid) + 1);
	for (int i = 0; i < per_lane; i++) {
		h = mix(h + 0x9e3779b97f4a7c15ull);
		const uint8_t* B = table + __umul64hi(h, nblocks) * 128;
		if constexpr (MODE == kOne16) {
			const u32x4 a = *reinterpret_cast<const u32x4*>(B);
			acc += a.x ^ a.w;
		} else if constexpr (MODE == kStraddle) {
			const u32x4 a = *reinterpret_cast<const u32x4*>(B + 56);
			acc += a.x ^ a.w;
		} else if constexpr (MODE == kTwoInsts) {
			u32x4 a, b;
			asm volatile(
			    "global_load_dwordx4 %0, %2, off\n\t"
			    "global_load_dwordx4 %1, %2, off offset:64\n\t"
			    "s_waitcnt vmcnt(0)"
			    : "=&v"(a), "=&v"(b)
			    : "v"(B)
			    : "memory");
			acc += a.x ^ b.w;
		} else if constexpr (MODE == kDependent) {
			const u32x4 a = *reinterpret_cast<const u32x4*>(B);
			const u32x2 b = *reinterpret_cast<const u32x2*>(B + 64 + (a.x & 0x38u));
			acc += a.w ^ b.x;
		} else if constexpr (MODE == kPairLanes) {
			const u32x4 a = *reinterpret_cast<const u32x4*>(B + 64 * (tid & 1));
			acc += a.x ^ a.w;
		} else {
			const uint8_t* B2 = table + __umul64hi(mix(h ^ 0x5555555555555555ull), nblocks) * 128;
			u32x4 a, b;
			asm volatile(
			    "global_load_dwordx4 %0, %2, off\n\t"
			    "global_load_dwordx4 %1, %3, off\n\t"
			    "s_waitcnt vmcnt(0)"
			    : "=&v"(a), "=&v"(b)
			    : "v"(B), "v"(B2)
			    : "memory");
			acc += a.x ^ b.w;
		}
	}
	if (acc == 0x1234567)
		sink[0] = acc;
}

__global__ void k_fill(uint64_t* p, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		p[i] = mix(i);
}

int main(int argc, char** argv)
{
	const double gib = argc > 1 ? atof(argv[1]) : 8.0;
	const int per_lane = argc > 2 ? atoi(argv[2]) : 8;
	const uint64_t bytes = (uint64_t)(gib * (1ull << 30));
	const uint64_t nblocks = bytes / 128;
	uint64_t *table, *sink;
	CK(hipMalloc(&table, bytes + 256));
	CK(hipMalloc(&sink, 64));
	hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, table, bytes / 8);
	CK(hipDeviceSynchronize());
	const uint64_t lanes = 256ull * 1024 * 32;  // 8.4 M lanes, 8 waves per SIMD
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const dim3 grid((unsigned)(lanes / 256)), block(256);
	auto run = [&](const char* name, auto launch, double blocks_per_lane) {
		launch();
		CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0));
		const int reps = 5;
		for (int r = 0; r < reps; r++)
			launch();
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		const double blocks = (double)lanes * per_lane * blocks_per_lane;
		printf("{\"test\": \"%s\", \"table_GiB\": %.2f, \"blocks\": %.0f, \"ms\": %.3f, \"Gblocks_per_s\": %.2f}\n", name, gib, blocks, ms / reps,
		       blocks * reps / (ms * 1e-3) / 1e9);
		fflush(stdout);
	};
#define RUN(NAME, MODE, BPL) \
	run(NAME, [&] { hipLaunchKernelGGL((k_probe<MODE>), grid, block, 0, 0, (const uint8_t*)table, nblocks, per_lane, sink); }, BPL)
	RUN("one16", kOne16, 1.0);
	RUN("straddle", kStraddle, 1.0);
	RUN("two_insts", kTwoInsts, 1.0);
	RUN("dependent", kDependent, 1.0);
	RUN("pair_lanes", kPairLanes, 0.5);
	RUN("far_two", kFarTwo, 2.0);
	RUN("one16_again", kOne16, 1.0);
	return 0;
}
