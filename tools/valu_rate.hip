// valu_rate.hip -- issue rate of the integer VALU instructions the query path leans on (gfx950), relative to v_xor_b32.
// One wave per SIMD x 8 waves, 8 independent chains per lane, inline asm so that the compiler cannot fold the loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAIN8(OP)                                                                                       \
	asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                         \
	             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
	             : "v"(b), "s"(c))

#define OP_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define OP_MUL(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define OP_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 6\n"
#define OP_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define OP_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define OP_XOR3(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define OP_CMP(i) "v_cmp_eq_u32 vcc, %" #i ", %8\n"
#define OP_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CNDE64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define OP_CMPCND(i) "v_cmp_eq_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_STEP(i) "v_alignbit_b32 v40, %" #i ", %8, 6\nv_cmp_eq_u32 vcc, v40, %8\nv_cmp_eq_u32_e64 s[20:21], v40, %" #i "\ns_or_b64 s[20:21], vcc, s[20:21]\nv_cndmask_b32_e64 %" #i ", %" #i ", 7, s[20:21]\n"
#define OP_STEPMIN(i) "v_alignbit_b32 v40, %" #i ", %8, 6\nv_xor_b32 v41, v40, %8\nv_xor_b32 v40, v40, %" #i "\nv_min3_u32 %" #i ", %" #i ", v40, v41\n"
#define OP_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %8\n"
#define OP_SDWA(i) "v_xor_b32_sdwa %" #i ", %" #i ", %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"

template <int WHICH> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t b, uint32_t c)
{
	uint32_t a[8];
	for (int i = 0; i < 8; i++)
		a[i] = threadIdx.x * 8 + i;
	for (int it = 0; it < iters; it++) {
		if constexpr (WHICH == 0) CHAIN8(OP_XOR);
		if constexpr (WHICH == 1) CHAIN8(OP_MUL);
		if constexpr (WHICH == 2) CHAIN8(OP_MUL24);
		if constexpr (WHICH == 3) CHAIN8(OP_MAD24);
		if constexpr (WHICH == 4) CHAIN8(OP_ALIGN);
		if constexpr (WHICH == 5) CHAIN8(OP_BCNT);
		if constexpr (WHICH == 6) CHAIN8(OP_MULHI);
		if constexpr (WHICH == 7) CHAIN8(OP_ADD3);
		if constexpr (WHICH == 8) CHAIN8(OP_XOR3);
		if constexpr (WHICH == 9) asm volatile(OP_CMP(0) OP_CMP(1) OP_CMP(2) OP_CMP(3) OP_CMP(4) OP_CMP(5) OP_CMP(6) OP_CMP(7) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "s"(c) : "vcc");
		if constexpr (WHICH == 10) asm volatile("s_mov_b64 vcc, 0x5555\n" OP_CNDS(0) OP_CNDS(1) OP_CNDS(2) OP_CNDS(3) OP_CNDS(4) OP_CNDS(5) OP_CNDS(6) OP_CNDS(7) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "s"(c) : "vcc");
		if constexpr (WHICH == 11) CHAIN8(OP_SDWA);
		if constexpr (WHICH == 12) asm volatile(OP_CNDE64(0) OP_CNDE64(1) OP_CNDE64(2) OP_CNDE64(3) OP_CNDE64(4) OP_CNDE64(5) OP_CNDE64(6) OP_CNDE64(7) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "s"(c) : "s20", "s21");
		if constexpr (WHICH == 13) asm volatile(OP_CMPCND(0) OP_CMPCND(1) OP_CMPCND(2) OP_CMPCND(3) OP_CMPCND(4) OP_CMPCND(5) OP_CMPCND(6) OP_CMPCND(7) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "s"(c) : "vcc");
		if constexpr (WHICH == 14) asm volatile(OP_STEP(0) OP_STEP(1) OP_STEP(2) OP_STEP(3) OP_STEP(4) OP_STEP(5) OP_STEP(6) OP_STEP(7) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "s"(c) : "vcc", "scc", "s20", "s21", "v40", "v41");
		if constexpr (WHICH == 15) asm volatile(OP_STEPMIN(0) OP_STEPMIN(1) OP_STEPMIN(2) OP_STEPMIN(3) OP_STEPMIN(4) OP_STEPMIN(5) OP_STEPMIN(6) OP_STEPMIN(7) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "s"(c) : "vcc", "scc", "s20", "s21", "v40", "v41");
		if constexpr (WHICH == 16) CHAIN8(OP_MIN3);
	}
	uint32_t s = 0;
	for (int i = 0; i < 8; i++)
		s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WHICH> static void run(const char* name, double base_ms, double* ms_out)
{
	uint32_t* out;
	const int blocks = 256 * 8, iters = 4096;  // 8 waves per SIMD on every CU
	hipMalloc(&out, blocks * 256 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(256), 0, 0, out, 64, 12345u, 77u);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u, 77u);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double ops = (double)blocks * 256 * iters * 8;
	printf("{\"op\": \"%s\", \"ms\": %.3f, \"lane_ops_per_s\": %.4g, \"relative_cost\": %.2f}\n", name, ms, ops / (ms * 1e-3),
	       base_ms > 0 ? ms / base_ms : 1.0);
	fflush(stdout);
	if (ms_out)
		*ms_out = ms;
	hipFree(out);
}

int main()
{
	double base = 0;
	run<0>("v_xor_b32", 0, &base);
	run<1>("v_mul_lo_u32", base, nullptr);
	run<2>("v_mul_u32_u24", base, nullptr);
	run<3>("v_mad_u32_u24", base, nullptr);
	run<4>("v_alignbit_b32", base, nullptr);
	run<5>("v_bcnt_u32_b32", base, nullptr);
	run<6>("v_mul_hi_u32", base, nullptr);
	run<7>("v_add3_u32", base, nullptr);
	run<8>("v_and_or_b32", base, nullptr);
	run<9>("v_cmp_eq_u32", base, nullptr);
	run<10>("v_cndmask_b32", base, nullptr);
	run<11>("v_xor_b32_sdwa", base, nullptr);
	run<12>("v_cndmask_b32_e64 (sgpr mask)", base, nullptr);
	run<13>("v_cmp_eq + v_cndmask (2 instr)", base, nullptr);
	run<14>("filter step: alignbit+cmp+cmp+s_or+cndmask (4 VALU)", base, nullptr);
	run<15>("min step: alignbit+xor+xor+min3 (4 VALU)", base, nullptr);
	run<16>("v_min3_u32", base, nullptr);
	run<0>("v_xor_b32 (again, warm)", base, nullptr);
	return 0;
}
