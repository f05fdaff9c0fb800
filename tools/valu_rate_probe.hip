// valu_rate_probe -- issue rate of the integer VALU instructions the minimizer hash is made of, on the card at hand: every wave runs
// 8 independent chains of ONE instruction, 16 waves per CU on every CU; cycles per wave-instruction and SIMD from the kernel time and the
// shader clock.  One JSON line per instruction.     hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate_probe tools/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x)                                                        \
	do {                                                             \
		hipError_t e_ = (x);                                         \
		if (e_ != hipSuccess) {                                      \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
			return 1;                                                \
		}                                                            \
	} while (0)

constexpr int kIters = 2048, kChains = 8;

#define PROBE(NAME, ASM)                                                                          \
	__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t c)                        \
	{                                                                                             \
		uint32_t v[kChains];                                                                      \
		for (int j = 0; j < kChains; j++)                                                         \
			v[j] = threadIdx.x * 2654435761u + j;                                                 \
		for (int i = 0; i < kIters; i++) {                                                        \
			_Pragma("unroll") for (int u = 0; u < 8; u++)                                              \
			_Pragma("unroll") for (int j = 0; j < kChains; j++) asm volatile(ASM : "+v"(v[j]) : "v"(c)); \
		}                                                                                         \
		uint32_t s = 0;                                                                           \
		for (int j = 0; j < kChains; j++)                                                         \
			s ^= v[j];                                                                            \
		if (s == 0x12345)                                                                         \
			out[0] = s;                                                                           \
	}

PROBE(p_xor, "v_xor_b32 %0, %0, %1")
PROBE(p_and, "v_and_b32 %0, %0, %1")
PROBE(p_or, "v_or_b32 %0, %0, %1")
PROBE(p_lshr, "v_lshrrev_b32 %0, 3, %0")
PROBE(p_lshl, "v_lshlrev_b32 %0, 3, %0")
PROBE(p_lshr_v, "v_lshrrev_b32 %0, %1, %0")
PROBE(p_add_u32, "v_add_u32 %0, %0, %1")
PROBE(p_sub_u32, "v_sub_u32 %0, %0, %1")
PROBE(p_add_f32, "v_add_f32 %0, %0, %1")
PROBE(p_fma_f32, "v_fma_f32 %0, %0, %1, %0")
PROBE(p_mov, "v_mov_b32 %0, %1")
PROBE(p_not, "v_not_b32 %0, %0")
PROBE(p_bfrev, "v_bfrev_b32 %0, %0")
PROBE(p_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
PROBE(p_max_u32, "v_max_u32 %0, %0, %1")
PROBE(p_min_u32, "v_min_u32 %0, %0, %1")
PROBE(p_min_f32, "v_min_f32 %0, %0, %1")
PROBE(p_min3_i32, "v_min3_i32 %0, %0, %1, %0")
PROBE(p_and_or, "v_and_or_b32 %0, %0, %1, %0")
PROBE(p_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
PROBE(p_add3, "v_add3_u32 %0, %0, %1, %0")
PROBE(p_xad, "v_xad_u32 %0, %0, %1, %0")
PROBE(p_perm, "v_perm_b32 %0, %0, %1, %1")
PROBE(p_mul_i24, "v_mul_i32_i24 %0, %0, %1")
PROBE(p_xor_e64, "v_xor_b32_e64 %0, %0, %1")
PROBE(p_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
PROBE(p_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
PROBE(p_pk_lshr_b16, "v_pk_lshrrev_b16 %0, 3, %0")
PROBE(p_mul_lo, "v_mul_lo_u32 %0, %0, %1")
PROBE(p_mul_hi, "v_mul_hi_u32 %0, %0, %1")
PROBE(p_mul_u24, "v_mul_u32_u24 %0, %0, %1")
PROBE(p_mad_u24, "v_mad_u32_u24 %0, %0, %1, %0")
PROBE(p_mad_u16, "v_mad_u32_u16 %0, %0, %1, %0")
PROBE(p_mul_u24_sdwa, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")
PROBE(p_xor_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
PROBE(p_lshl_add, "v_lshl_add_u32 %0, %0, 16, %1")
PROBE(p_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
PROBE(p_bfe, "v_bfe_u32 %0, %0, 1, 17")
PROBE(p_bfi, "v_bfi_b32 %0, %1, %0, %1")
PROBE(p_min_i32, "v_min_i32 %0, %0, %1")
PROBE(p_mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
PROBE(p_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
PROBE(p_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %0")

int main()
{
	hipDeviceProp_t pr;
	CK(hipGetDeviceProperties(&pr, 0));
	const int cus = pr.multiProcessorCount;
	const double ghz = pr.clockRate * 1e-6;
	uint32_t* out;
	CK(hipMalloc(&out, 64));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	struct P {
		const char* name;
		void (*k)(uint32_t*, uint32_t);
	} probes[] = {{"v_xor_b32", p_xor}, {"v_and_b32", p_and}, {"v_or_b32", p_or}, {"v_lshrrev_b32 imm", p_lshr}, {"v_lshlrev_b32 imm", p_lshl},
	              {"v_lshrrev_b32 vgpr", p_lshr_v}, {"v_add_u32", p_add_u32}, {"v_sub_u32", p_sub_u32}, {"v_add_f32", p_add_f32}, {"v_fma_f32", p_fma_f32},
	              {"v_mov_b32", p_mov}, {"v_not_b32", p_not}, {"v_bfrev_b32", p_bfrev}, {"v_bcnt_u32_b32", p_bcnt}, {"v_max_u32", p_max_u32},
	              {"v_min_u32", p_min_u32}, {"v_min_f32", p_min_f32}, {"v_min3_i32", p_min3_i32}, {"v_and_or_b32", p_and_or}, {"v_lshl_or_b32", p_lshl_or},
	              {"v_add3_u32", p_add3}, {"v_xad_u32", p_xad}, {"v_perm_b32", p_perm}, {"v_mul_i32_i24", p_mul_i24}, {"v_xor_b32_e64", p_xor_e64},
	              {"v_pk_add_u16", p_pk_add_u16}, {"v_pk_min_i16", p_pk_min_i16}, {"v_pk_lshrrev_b16", p_pk_lshr_b16},
	              {"v_mul_lo_u32", p_mul_lo}, {"v_mul_hi_u32", p_mul_hi}, {"v_mul_u32_u24", p_mul_u24}, {"v_mad_u32_u24", p_mad_u24},
	              {"v_mad_u32_u16", p_mad_u16}, {"v_mul_u32_u24_sdwa", p_mul_u24_sdwa}, {"v_xor_b32_sdwa", p_xor_sdwa}, {"v_lshl_add_u32", p_lshl_add},
	              {"v_alignbit_b32", p_alignbit}, {"v_bfe_u32", p_bfe}, {"v_bfi_b32", p_bfi}, {"v_min_i32", p_min_i32}, {"v_mul_lo_u16", p_mul_lo_u16},
	              {"v_pk_mul_lo_u16", p_pk_mul_lo_u16}, {"v_pk_mad_u16", p_pk_mad_u16}};
	for (auto& p : probes) {
		for (int wps : {1, 4, 8}) {  // waves per SIMD: wps workgroups of 4 waves per CU
			const int blocks = cus * wps;
			hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, out, 0x2c1b3c6du);
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, out, 0x2c1b3c6du);
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			const double insts_per_simd = (double)wps * kIters * kChains * 8;
			const double cyc = ms * 1e-3 * ghz * 1e9 / insts_per_simd;
			printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"clock_ghz\": %.3f, \"cycles_per_wave_inst\": %.2f}\n", p.name, wps, ms, ghz, cyc);
		}
	}
	return 0;
}
