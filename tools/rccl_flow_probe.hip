// rccl_flow_probe -- one pairwise flow of the all-to-all-v on its own: a self ncclSend / ncclRecv pair (world of one rank, as in the
// round-2 rehearsal of the construction from slices where a 1.9-GB and a 2.4-GB flow were followed by a shard that differed from the
// whole-input build) at a ladder of sizes around 2^31 bytes; the source holds a position-dependent pattern, the destination is
// compared on the device: mismatching words, first and last mismatching index.  One JSON line per size.
//   rccl_flow_probe [max_GiB]
//   rccl_flow_probe --peer [max_GiB]   the same ladder as a flow from GPU 0 to GPU 1 (one process, two communicators: the question one
//                                      GPU cannot answer -- does a PEER flow above 2^30 bytes arrive whole?); needs two GPUs
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                      \
	do {                                                                           \
		hipError_t e_ = (x);                                                       \
		if (e_ != hipSuccess) {                                                    \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
			return 1;                                                              \
		}                                                                          \
	} while (0)
#define NK(x)                                                                      \
	do {                                                                           \
		ncclResult_t r_ = (x);                                                     \
		if (r_ != ncclSuccess) {                                                   \
			fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));              \
			return 1;                                                              \
		}                                                                          \
	} while (0)

__device__ __forceinline__ unsigned long long pattern(unsigned long long i) { return (i * 0x9E3779B97F4A7C15ull) ^ (i >> 7) ^ 0x5bd1e995ull; }

__global__ void k_fill(unsigned long long* p, unsigned long long n, unsigned long long salt)
{
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
		p[i] = salt ? salt : pattern(i);
}

__global__ void k_compare(const unsigned long long* p, unsigned long long n, unsigned long long* out /* bad, first, last */)
{
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
		if (p[i] != pattern(i)) {
			atomicAdd(&out[0], 1ull);
			atomicMin(&out[1], i);
			atomicMax(&out[2], i);
		}
}

static int peer_main(double max_gib)
{
	int ndev = 0;
	CK(hipGetDeviceCount(&ndev));
	if (ndev < 2) {
		printf("{\"peer\": true, \"skipped\": \"needs two GPUs, this box has %d\"}\n", ndev);
		return 0;
	}
	int devs[2] = {0, 1};
	ncclComm_t comms[2];
	NK(ncclCommInitAll(comms, 2, devs));
	int ver = 0;
	NK(ncclGetVersion(&ver));
	hipStream_t st[2];
	const unsigned long long sizes[] = {1ull << 28, 1ull << 30, 1500000000ull, 1900000000ull, (1ull << 31) - 8, 1ull << 31, (1ull << 31) + 8, 2400000000ull, 3ull << 30, (1ull << 32) + 64};
	unsigned long long maxb = 0;
	for (unsigned long long b : sizes)
		if ((double)b <= max_gib * (double)(1ull << 30) + 1024 && b > maxb)
			maxb = b;
	unsigned long long *src = nullptr, *dst = nullptr, *res = nullptr;
	CK(hipSetDevice(0));
	CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
	CK(hipMalloc(&src, maxb));
	CK(hipSetDevice(1));
	CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
	CK(hipMalloc(&dst, maxb));
	CK(hipMalloc(&res, 64));
	for (unsigned long long bytes : sizes) {
		if (bytes > maxb)
			continue;
		const unsigned long long n = bytes / 8;
		for (int mode = 0; mode < 2; mode++) {  // 0: one grouped send/recv pair of n words; 1: the same flow in 256-MiB rounds
			CK(hipSetDevice(0));
			hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st[0], src, n, 0ull);
			CK(hipStreamSynchronize(st[0]));
			CK(hipSetDevice(1));
			hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st[1], dst, n, 0xDEADBEEFDEADBEEFull);
			const unsigned long long init[3] = {0, ~0ull, 0};
			CK(hipMemcpyAsync(res, init, 24, hipMemcpyHostToDevice, st[1]));
			CK(hipStreamSynchronize(st[1]));
			const unsigned long long fw = 1ull << 25;
			for (unsigned long long off = 0; off < n; off += mode ? fw : n) {
				const unsigned long long c = mode ? (n - off < fw ? n - off : fw) : n;
				NK(ncclGroupStart());
				NK(ncclSend(src + off, c, ncclUint64, 1, comms[0], st[0]));
				NK(ncclRecv(dst + off, c, ncclUint64, 0, comms[1], st[1]));
				NK(ncclGroupEnd());
			}
			CK(hipSetDevice(0));
			CK(hipStreamSynchronize(st[0]));
			CK(hipSetDevice(1));
			hipLaunchKernelGGL(k_compare, dim3(4096), dim3(256), 0, st[1], dst, n, res);
			unsigned long long h[3];
			CK(hipMemcpyAsync(h, res, 24, hipMemcpyDeviceToHost, st[1]));
			CK(hipStreamSynchronize(st[1]));
			printf("{\"peer\": true, \"rccl_version\": %d, \"bytes\": %llu, \"words\": %llu, \"rounds\": \"%s\", \"mismatching_words\": %llu, \"first\": %lld, \"last\": %lld}\n", ver,
			       bytes, n, mode ? "256MiB" : "one", h[0], h[0] ? (long long)h[1] : -1ll, h[0] ? (long long)h[2] : -1ll);
			fflush(stdout);
		}
	}
	ncclCommDestroy(comms[0]);
	ncclCommDestroy(comms[1]);
	return 0;
}

int main(int argc, char** argv)
{
	if (argc > 1 && std::string(argv[1]) == "--peer")
		return peer_main(argc > 2 ? atof(argv[2]) : 3.0);
	const double max_gib = argc > 1 ? atof(argv[1]) : 3.0;
	CK(hipSetDevice(0));
	ncclUniqueId id;
	NK(ncclGetUniqueId(&id));
	ncclComm_t comm;
	NK(ncclCommInitRank(&comm, 1, id, 0));
	int ver = 0;
	NK(ncclGetVersion(&ver));
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	const unsigned long long sizes[] = {1ull << 30, 1500000000ull, 1900000000ull, (1ull << 31) - 8, 1ull << 31, (1ull << 31) + 8, 2400000000ull, 3ull << 30, (1ull << 32) + 64};
	unsigned long long *src = nullptr, *dst = nullptr, *res = nullptr;
	unsigned long long maxb = 0;
	for (unsigned long long b : sizes)
		if ((double)b <= max_gib * (double)(1ull << 30) + 1024 && b > maxb)
			maxb = b;
	CK(hipMalloc(&src, maxb));
	CK(hipMalloc(&dst, maxb));
	CK(hipMalloc(&res, 64));
	for (unsigned long long bytes : sizes) {
		if (bytes > maxb)
			continue;
		const unsigned long long n = bytes / 8;
		for (int mode = 0; mode < 2; mode++) {  // 0: one grouped send/recv pair of n words; 1: the same flow in 256-MiB rounds
			hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, src, n, 0ull);
			hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, dst, n, 0xDEADBEEFDEADBEEFull);
			const unsigned long long init[3] = {0, ~0ull, 0};
			CK(hipMemcpyAsync(res, init, 24, hipMemcpyHostToDevice, s));
			const unsigned long long fw = 1ull << 25;
			for (unsigned long long off = 0; off < n; off += mode ? fw : n) {
				const unsigned long long c = mode ? (n - off < fw ? n - off : fw) : n;
				NK(ncclGroupStart());
				NK(ncclSend(src + off, c, ncclUint64, 0, comm, s));
				NK(ncclRecv(dst + off, c, ncclUint64, 0, comm, s));
				NK(ncclGroupEnd());
			}
			hipLaunchKernelGGL(k_compare, dim3(4096), dim3(256), 0, s, dst, n, res);
			unsigned long long h[3];
			CK(hipMemcpyAsync(h, res, 24, hipMemcpyDeviceToHost, s));
			CK(hipStreamSynchronize(s));
			printf("{\"rccl_version\": %d, \"bytes\": %llu, \"words\": %llu, \"rounds\": \"%s\", \"mismatching_words\": %llu, \"first\": %lld, \"last\": %lld}\n", ver, bytes, n,
			       mode ? "256MiB" : "one", h[0], h[0] ? (long long)h[1] : -1ll, h[0] ? (long long)h[2] : -1ll);
			fflush(stdout);
		}
	}
	ncclCommDestroy(comm);
	return 0;
}
