// gather_ceiling.hip -- micro-benchmark: how fast can an MI355X serve RANDOM 64-byte lines from a table much larger than
// the 256 MiB Infinity Cache?  This is the honest ceiling of the Blight query path (a chain of dependent random sector
// reads), next to the 8 TB/s streaming peak.  Also used to calibrate rocprofv3's FETCH_SIZE on this access pattern
// (known byte count = lines * 64 B).
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_ceiling tools/gather_ceiling.hip
//   ./gather_ceiling [table_GiB=8] [lines_per_lane=16] [chain=1|4]
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

// every lane reads `per_lane` random 64-B lines (4 x 16 B); CHAIN > 1 makes groups of CHAIN reads dependent
template <int CHAIN>
__global__ __launch_bounds__(256) void k_gather(const ulonglong2* __restrict__ table, uint64_t nlines, int per_lane, uint64_t* __restrict__ sink)
{
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0;
	uint64_t h = mix(tid + 1);
	for (int i = 0; i < per_lane; i += CHAIN) {
		uint64_t dep = 0;
#pragma unroll
		for (int c = 0; c < CHAIN; c++) {
			h = mix(h + dep + 0x9e3779b97f4a7c15ull);
			const uint64_t line = __umul64hi(h, nlines);
			const ulonglong2* L = table + line * 4;
			const ulonglong2 a = L[0], b = L[1], cc = L[2], d = L[3];
			dep = a.x ^ a.y ^ b.x ^ b.y ^ cc.x ^ cc.y ^ d.x ^ d.y;
			acc += dep;
		}
	}
	if (acc == 0x1234567)
		sink[0] = acc;
}

// one 8-byte word per random line (sector fetch without using the whole line)
__global__ __launch_bounds__(256) void k_gather8(const uint64_t* __restrict__ table, uint64_t nlines, int per_lane, uint64_t* __restrict__ sink)
{
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0, h = mix(tid + 1);
	for (int i = 0; i < per_lane; i++) {
		h = mix(h + 0x9e3779b97f4a7c15ull);
		acc += table[__umul64hi(h, nlines) * 8];
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
	const int per_lane = argc > 2 ? atoi(argv[2]) : 16;
	const uint64_t bytes = (uint64_t)(gib * (1ull << 30));
	const uint64_t nlines = bytes / 64;
	uint64_t *table, *sink;
	CK(hipMalloc(&table, bytes));
	CK(hipMalloc(&sink, 64));
	hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, table, bytes / 8);
	CK(hipDeviceSynchronize());
	const uint64_t lanes = 256ull * 1024 * 32;  // 8.4 M lanes
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	auto run = [&](const char* name, auto launch, double bytes_per_line) {
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
		const double lines = (double)lanes * per_lane * reps;
		printf("{\"test\": \"%s\", \"table_GiB\": %.1f, \"lines\": %.0f, \"ms\": %.3f, \"Glines_per_s\": %.3f, \"GB_per_s\": %.1f}\n", name, gib,
		       lines / reps, ms / reps, lines / (ms * 1e-3) / 1e9, lines * bytes_per_line / (ms * 1e-3) / 1e9);
	};
	const dim3 grid((unsigned)(lanes / 256)), block(256);
	run("random_64B_lines_independent", [&] { hipLaunchKernelGGL((k_gather<1>), grid, block, 0, 0, (const ulonglong2*)table, nlines, per_lane, sink); }, 64);
	run("random_64B_lines_chain4", [&] { hipLaunchKernelGGL((k_gather<4>), grid, block, 0, 0, (const ulonglong2*)table, nlines, per_lane, sink); }, 64);
	run("random_8B_words_independent", [&] { hipLaunchKernelGGL(k_gather8, grid, block, 0, 0, (const uint64_t*)table, nlines, per_lane, sink); }, 64);
	return 0;
}
