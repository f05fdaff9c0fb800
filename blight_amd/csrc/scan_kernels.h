// scan_kernels.h -- exclusive prefix sums, hand-written (was rocprim::exclusive_scan): the (digit, tile) matrices of the build's radix sort
// and of the exact batch partition (32-bit), the read front end's position prefix (popcounts of bitmap words), the build's length and
// bitmap prefixes (64-bit).  Three small kernels: sums of blocks of 8192 entries, ONE block over those sums (in place), then every
// block scans its entries again behind its offset.  The input is a functor In: `T operator()(uint64_t i)`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blmx_internal.h"

namespace blmx {
namespace {

constexpr int kScanThreads = 1024, kScanPer = 8, kScanTile = kScanThreads * kScanPer;
__device__ __forceinline__ uint32_t scan_shfl_up(uint32_t v, int o) { return __shfl_up(v, o); }
__device__ __forceinline__ uint64_t scan_shfl_up(uint64_t v, int o)
{
	const uint32_t lo = __shfl_up((uint32_t)v, o), hi = __shfl_up((uint32_t)(v >> 32), o);
	return (uint64_t)lo | ((uint64_t)hi << 32);
}
template <typename T> __device__ __forceinline__ T block_exclusive_scan(T mine, T* s_w /* kScanThreads / 64 */, T& total)
{
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	T v = mine;
	for (int o = 1; o < 64; o <<= 1) {
		const T u = scan_shfl_up(v, o);
		if ((int)lane >= o)
			v += u;
	}
	if (lane == 63)
		s_w[wave] = v;
	__syncthreads();
	T before = 0, all = 0;
	for (uint32_t w = 0; w < blockDim.x / 64; w++) {
		const T x = s_w[w];
		if (w < wave)
			before += x;
		all += x;
	}
	__syncthreads();
	total = all;
	return before + v - mine;
}
// what is summed
template <typename T> struct ScanPlain {
	const T* p;
	__device__ __forceinline__ T operator()(uint64_t i) const { return p[i]; }
};
template <typename T> struct ScanPopc64 {  // popcounts of 64-bit bitmap words
	const unsigned long long* p;
	__device__ __forceinline__ T operator()(uint64_t i) const { return (T)__popcll(p[i]); }
};
template <typename In, typename T> __global__ __launch_bounds__(kScanThreads) void k_scan_sums(In in, uint64_t n, T* __restrict__ sums)
{
	__shared__ T s_w[kScanThreads / 64];
	const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer;
	T mine = 0;
#pragma unroll
	for (int j = 0; j < kScanPer; j++)
		mine += base + j < n ? (T)in(base + j) : (T)0;
	T total;
	(void)block_exclusive_scan<T>(mine, s_w, total);
	if (threadIdx.x == 0)
		sums[blockIdx.x] = total;
}
template <typename T> __global__ __launch_bounds__(kScanThreads) void k_scan_top(T* __restrict__ sums, uint32_t nsums)  // one block; in place, exclusive
{
	__shared__ T s_w[kScanThreads / 64];
	T carry = 0;
	for (uint32_t at = 0; at < nsums; at += kScanThreads) {
		const uint32_t i = at + threadIdx.x;
		const T mine = i < nsums ? sums[i] : (T)0;
		T total;
		const T ex = block_exclusive_scan<T>(mine, s_w, total);
		if (i < nsums)
			sums[i] = carry + ex;
		carry += total;
	}
}
template <typename In, typename T> __global__ __launch_bounds__(kScanThreads) void k_scan_apply(In in, uint64_t n, const T* __restrict__ sums, T* __restrict__ out)
{
	__shared__ T s_w[kScanThreads / 64];
	const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer;
	T x[kScanPer], mine = 0;
#pragma unroll
	for (int j = 0; j < kScanPer; j++) {
		x[j] = base + j < n ? (T)in(base + j) : (T)0;
		mine += x[j];
	}
	T total;
	T at = sums[blockIdx.x] + block_exclusive_scan<T>(mine, s_w, total);
#pragma unroll
	for (int j = 0; j < kScanPer; j++) {
		if (base + j < n)
			out[base + j] = at;
		at += x[j];
	}
}
inline size_t scan_sums_words(uint64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile) + 1; }
// in -> out (`out` may be the array `in` reads: every thread loads its entries before it stores them); d_sums: scan_sums_words(n)
// values of T of scratch
template <typename T, typename In> inline void exclusive_scan_of(In in, T* out, uint64_t n, T* d_sums, hipStream_t s)
{
	if (!n)
		return;
	const uint32_t nb = (uint32_t)((n + kScanTile - 1) / kScanTile);
	hipLaunchKernelGGL((k_scan_sums<In, T>), dim3(nb), dim3(kScanThreads), 0, s, in, n, d_sums);
	hipLaunchKernelGGL((k_scan_top<T>), dim3(1), dim3(kScanThreads), 0, s, d_sums, nb);
	hipLaunchKernelGGL((k_scan_apply<In, T>), dim3(nb), dim3(kScanThreads), 0, s, in, n, d_sums, out);
	BLMX_HIP(hipGetLastError());
}
inline void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* d_sums, hipStream_t s)
{
	exclusive_scan_of<uint32_t>(ScanPlain<uint32_t>{in}, out, n, d_sums, s);
}

}  // namespace
}  // namespace blmx
