// pingpong_probe.hip -- micro-benchmark behind DESIGN section 4.2 (r04): what does ONE blocking request/answer round trip between a host
// thread and a resident wave cost, depending on WHERE the request word lives?  The serving burst (blmx_serve.hip) answers the
// reference's per-call pattern (blight.cpp:537-550) from a kernel that polls request slots; a blocking call is then two crossings of
// PCIe plus the kernel's own dependent loads.  With the request word in pinned HOST memory the wave's poll is a read over PCIe (a full
// round trip per look); with the request word in DEVICE memory that the host writes through the BAR the poll is local and the host's
// store is a posted write.  The answer always goes to pinned host memory (a posted write of the device, polled locally by the host).
//
//   request in:  host      hipHostMalloc (coherent)                    -- what blmx_serve.hip did before
//                dev       hipMalloc, written by the host through the BAR
//                dev_fine  hipExtMallocWithFlags(hipDeviceMallocFinegrained)
//                dev_unc   hipExtMallocWithFlags(hipDeviceMallocUncached)
//
// A variant whose memory the host cannot write (no large BAR) is reported as "host cannot write" (SIGSEGV caught), not a crash.  The
// kernel is bounded: it leaves after `rounds` answers or 3 seconds, whichever first.
//
//   hipcc --offload-arch=gfx950 -O2 -o pingpong_probe tools/pingpong_probe.hip ;  ./pingpong_probe [rounds=20000] [chain=0]
//   chain = dependent 16-B loads from a 1-GiB table between request and answer (4 = what a scalar query does)
#include <hip/hip_runtime.h>

#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <immintrin.h>
#include <unistd.h>

#define CK(x)                                                                              \
	do {                                                                                   \
		hipError_t e = (x);                                                                \
		if (e != hipSuccess) {                                                             \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                         \
			exit(1);                                                                       \
		}                                                                                  \
	} while (0)

__global__ __launch_bounds__(64) void k_pong(uint32_t* req, uint32_t* ans, uint32_t rounds, const uint64_t* __restrict__ table, uint64_t nwords,
                                             int chain, unsigned long long max_ticks)
{
	if (threadIdx.x != 0)
		return;
	const unsigned long long t0 = wall_clock64();
	uint32_t next = 1;
	uint64_t h = 88172645463325252ull;
	while (next <= rounds) {
		const uint32_t r = __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		if (r == next) {
			for (int c = 0; c < chain; c++) {  // a scalar query's dependent round trips to HBM
				h ^= h << 13;
				h ^= h >> 7;
				h ^= h << 17;
				h += table[(h % nwords)];
			}
			__hip_atomic_store(ans, next + (h == 42u ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // (the answer depends on the chain)
			next++;
		} else if (wall_clock64() - t0 > max_ticks) {
			break;
		}
	}
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static bool host_can_write(volatile uint32_t* p)
{
	struct sigaction sa{}, old_segv{}, old_bus{};
	sa.sa_handler = on_segv;
	sigemptyset(&sa.sa_mask);
	sigaction(SIGSEGV, &sa, &old_segv);
	sigaction(SIGBUS, &sa, &old_bus);
	bool ok = false;
	if (sigsetjmp(g_jmp, 1) == 0) {
		*p = 0;
		_mm_sfence();
		ok = true;
	}
	sigaction(SIGSEGV, &old_segv, nullptr);
	sigaction(SIGBUS, &old_bus, nullptr);
	return ok;
}

// the same question without a signal: let the KERNEL store into the page (read(2) from /dev/zero); an address the process cannot write is
// EFAULT, not SIGSEGV -- what a library can afford to ask
static int kernel_can_write(void* p)
{
	const int fd = open("/dev/zero", O_RDONLY);
	if (fd < 0)
		return -1;
	const ssize_t r = read(fd, p, 4);
	close(fd);
	return r == 4 ? 1 : 0;
}

int main(int argc, char** argv)
{
	const uint32_t rounds = argc > 1 ? (uint32_t)atoi(argv[1]) : 20000u;
	const int chain = argc > 2 ? atoi(argv[2]) : 0;
	CK(hipSetDevice(0));
	int wc_mhz = 0;
	CK(hipDeviceGetAttribute(&wc_mhz, hipDeviceAttributeWallClockRate, 0));  // kHz
	const unsigned long long max_ticks = 3ull * 1000ull * (unsigned long long)wc_mhz;
	const uint64_t nwords = 1ull << 27;  // 1 GiB
	uint64_t* table = nullptr;
	CK(hipMalloc(&table, nwords * 8));
	CK(hipMemset(table, 0, nwords * 8));
	uint32_t* ans = nullptr;
	CK(hipHostMalloc(&ans, 64, hipHostMallocDefault));
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	const char* names[4] = {"host", "dev", "dev_fine", "dev_unc"};
	for (int mode = 0; mode < 4; mode++) {
		uint32_t* req = nullptr;
		hipError_t e = hipSuccess;
		if (mode == 0)
			e = hipHostMalloc(&req, 64, hipHostMallocDefault);
		else if (mode == 1)
			e = hipMalloc(&req, 4096);
		else if (mode == 2)
			e = hipExtMallocWithFlags((void**)&req, 4096, hipDeviceMallocFinegrained);
		else
			e = hipExtMallocWithFlags((void**)&req, 4096, hipDeviceMallocUncached);
		if (e != hipSuccess) {
			printf("{\"request_in\": \"%s\", \"error\": \"%s\"}\n", names[mode], hipGetErrorString(e));
			(void)hipGetLastError();
			continue;
		}
		if (mode)
			CK(hipMemset(req, 0, 4096));
		CK(hipDeviceSynchronize());
		const int kw = kernel_can_write(req);
		if (!host_can_write(req)) {
			printf("{\"request_in\": \"%s\", \"error\": \"host cannot write\", \"read2_probe\": %d}\n", names[mode], kw);
			continue;
		}
		printf("{\"request_in\": \"%s\", \"host_can_write\": true, \"read2_probe\": %d}\n", names[mode], kw);
		*ans = 0;
		hipLaunchKernelGGL(k_pong, dim3(1), dim3(64), 0, s, req, ans, rounds, table, nwords, chain, max_ticks);
		CK(hipGetLastError());
		volatile uint32_t* vreq = req;
		volatile uint32_t* vans = ans;
		uint32_t done = 0;
		double worst = 0;
		const auto T0 = std::chrono::steady_clock::now();
		for (uint32_t i = 1; i <= rounds; i++) {
			const auto a = std::chrono::steady_clock::now();
			*vreq = i;
			_mm_sfence();
			bool got = false;
			for (;;) {
				if (__atomic_load_n(vans, __ATOMIC_ACQUIRE) == i) {
					got = true;
					break;
				}
				if (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() > 0.5)
					break;
			}
			if (!got)
				break;
			done = i;
			const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
			if (i > 100 && dt > worst)
				worst = dt;
		}
		const double T = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
		CK(hipStreamSynchronize(s));  // bounded: the kernel leaves after 3 s at the latest
		printf("{\"request_in\": \"%s\", \"rounds_answered\": %u, \"of\": %u, \"chain\": %d, \"us_per_round_trip\": %.3f, \"worst_us\": %.1f}\n", names[mode], done,
		       rounds, chain, done ? T / done * 1e6 : 0.0, worst * 1e6);
		fflush(stdout);
		if (mode == 0)
			CK(hipHostFree(req));
		else
			CK(hipFree(req));
	}
	return 0;
}
