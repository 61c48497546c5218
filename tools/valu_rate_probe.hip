// valu_rate_probe.hip -- issue rate and dependent latency of the VALU instructions the 8-bit forward kernel is made of,
// per SIMD of an MI355X (gfx950).  Settles whether a wave64 instruction of each kind occupies its SIMD for 2 or 4
// cycles (MI355X_MICROARCH.md says 2 for v_fma_f32; the SQ counters of round 1 suggested 4 for the packed ops).
//   build: hipcc --offload-arch=gfx950 -O2 -o valu_rate_probe tools/valu_rate_probe.hip
//   run:   ./valu_rate_probe            (prints one line per instruction and waves-per-SIMD setting)
// Method: every wave runs ITER trips over a block of 64 instructions of one kind; `indep` uses 8 rotating
// destination registers (throughput), `dep` chains one register (latency).  Cycles come from s_memtime (shader
// clock) taken by each wave around its loop; the figure printed is cycles per instruction PER SIMD =
// wave cycles / (instructions per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REP8(X) X X X X X X X X
#define ITER 2000

#define KERNEL(name, INDEP_BODY, DEP_BODY) \
__global__ void __launch_bounds__(256) name(int dep, unsigned *out, unsigned long long *cyc){ \
	unsigned r0 = threadIdx.x, r1 = r0 * 3u + 1u, r2 = r0 ^ 0x55u, r3 = r0 + 7u, r4 = r0 * 5u, r5 = r0 | 3u, r6 = r0 + 11u, r7 = r0 * 9u; \
	unsigned a = r0 * 0x00010001u + 0x00030002u, b = 0x00010001u; \
	const unsigned long long t0 = __builtin_readcyclecounter(); \
	if(dep){ for(int it = 0; it < ITER; it++){ asm volatile(REP8(REP8(DEP_BODY)) : "+v"(r0), "+v"(a), "+v"(b)); } } \
	else { for(int it = 0; it < ITER; it++){ asm volatile(REP8(INDEP_BODY) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(a), "+v"(b)); } } \
	const unsigned long long t1 = __builtin_readcyclecounter(); \
	out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ a ^ b; \
	if((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; \
}

#define I8(op, tail) \
	op " %0, %0, " tail "\n" op " %1, %1, " tail "\n" op " %2, %2, " tail "\n" op " %3, %3, " tail "\n" \
	op " %4, %4, " tail "\n" op " %5, %5, " tail "\n" op " %6, %6, " tail "\n" op " %7, %7, " tail "\n"

KERNEL(k_pk_add_i16,  I8("v_pk_add_i16", "%8 clamp"),  "v_pk_add_i16 %0, %0, %1 clamp\n")
KERNEL(k_pk_sub_i16,  I8("v_pk_sub_i16", "%8"),        "v_pk_sub_i16 %0, %0, %1\n")
KERNEL(k_pk_max_i16,  I8("v_pk_max_i16", "%8"),        "v_pk_max_i16 %0, %0, %1\n")
KERNEL(k_pk_min_u16,  I8("v_pk_min_u16", "%9"),        "v_pk_min_u16 %0, %0, %1\n")
KERNEL(k_pk_mad_u16,  I8("v_pk_mad_u16", "%9, %8"),    "v_pk_mad_u16 %0, %0, %2, %1\n")
KERNEL(k_add_u32,     I8("v_add_u32", "%8"),           "v_add_u32 %0, %0, %1\n")
KERNEL(k_max_i32,     I8("v_max_i32", "%8"),           "v_max_i32 %0, %0, %1\n")
KERNEL(k_max3_i32,    I8("v_max3_i32", "%8, %9"),      "v_max3_i32 %0, %0, %1, %2\n")
KERNEL(k_and_b32,     I8("v_and_b32", "%8"),           "v_and_b32 %0, %0, %1\n")
KERNEL(k_bfi_b32,     I8("v_bfi_b32", "%8, %9"),       "v_bfi_b32 %0, %0, %1, %2\n")
KERNEL(k_perm_b32,    I8("v_perm_b32", "%8, %9"),      "v_perm_b32 %0, %0, %1, %2\n")
KERNEL(k_add3_u32,    I8("v_add3_u32", "%8, %9"),      "v_add3_u32 %0, %0, %1, %2\n")
KERNEL(k_lshl_or,     I8("v_lshl_or_b32", "1, %9"),    "v_lshl_or_b32 %0, %0, 1, %2\n")
KERNEL(k_fma_f32,     I8("v_fma_f32", "%8, %9"),       "v_fma_f32 %0, %0, %1, %2\n")
// DPP: a VALU write followed by a DPP read of the same register needs 2 wait states; the independent form reads
// registers written 8 instructions earlier, the dependent form carries the s_nop the compiler would insert
KERNEL(k_mov_dpp_shr1, \
	"v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
	"v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
	"v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
	"v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n", \
	"s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_max_i32_dpp, \
	"v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
	"v_max_i32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
	"v_max_i32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %6, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
	"v_max_i32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n", \
	"s_nop 1\n v_max_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n")
// the recurrence's shape: a dependent chain of packed ops (add -> max -> sub), as one wave sees it
KERNEL(k_chain_mix, \
	"v_pk_add_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %0\n v_pk_sub_i16 %2, %1, %9\n v_pk_add_i16 %3, %3, %8\n" \
	"v_pk_max_i16 %4, %4, %3\n v_pk_sub_i16 %5, %4, %9\n v_pk_add_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %6\n", \
	"v_pk_add_i16 %0, %0, %1\n")
// s_nop between VALU: does it take an issue slot of its own?
KERNEL(k_add_nop, \
	"v_add_u32 %0, %0, %8\n s_nop 0\n v_add_u32 %1, %1, %8\n s_nop 0\n v_add_u32 %2, %2, %8\n s_nop 0\n v_add_u32 %3, %3, %8\n s_nop 0\n" \
	"v_add_u32 %4, %4, %8\n s_nop 0\n v_add_u32 %5, %5, %8\n s_nop 0\n v_add_u32 %6, %6, %8\n s_nop 0\n v_add_u32 %7, %7, %8\n s_nop 0\n", \
	"v_add_u32 %0, %0, %1\n s_nop 1\n")

// round 2, second pass: which instructions are in the 2-cycle class (packed f16, plain VOP2 logic / shifts, f32)?
KERNEL(k_pk_add_f16,  I8("v_pk_add_f16", "%8"),        "v_pk_add_f16 %0, %0, %1\n")
KERNEL(k_pk_max_f16,  I8("v_pk_max_f16", "%8"),        "v_pk_max_f16 %0, %0, %1\n")
KERNEL(k_pk_min_f16,  I8("v_pk_min_f16", "%8"),        "v_pk_min_f16 %0, %0, %1\n")
KERNEL(k_pk_fma_f16,  I8("v_pk_fma_f16", "%8, %9"),    "v_pk_fma_f16 %0, %0, %1, %2\n")
KERNEL(k_pk_mul_f16,  I8("v_pk_mul_f16", "%8"),        "v_pk_mul_f16 %0, %0, %1\n")
KERNEL(k_pk_add_u16,  I8("v_pk_add_u16", "%8"),        "v_pk_add_u16 %0, %0, %1\n")
KERNEL(k_pk_max_u16,  I8("v_pk_max_u16", "%8"),        "v_pk_max_u16 %0, %0, %1\n")
KERNEL(k_pk_lshr_b16, I8("v_pk_lshrrev_b16", "%8"),    "v_pk_lshrrev_b16 %0, %1, %0\n")
KERNEL(k_or_b32,      I8("v_or_b32", "%8"),            "v_or_b32 %0, %0, %1\n")
KERNEL(k_xor_b32,     I8("v_xor_b32", "%8"),           "v_xor_b32 %0, %0, %1\n")
KERNEL(k_sub_u32,     I8("v_sub_u32", "%8"),           "v_sub_u32 %0, %0, %1\n")
KERNEL(k_lshrrev_b32, I8("v_lshrrev_b32", "%8"),       "v_lshrrev_b32 %0, %1, %0\n")
KERNEL(k_lshlrev_b32, I8("v_lshlrev_b32", "%8"),       "v_lshlrev_b32 %0, %1, %0\n")
KERNEL(k_ashrrev_i32, I8("v_ashrrev_i32", "%8"),       "v_ashrrev_i32 %0, %1, %0\n")
KERNEL(k_cndmask,     I8("v_cndmask_b32", "%8, vcc"),  "v_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL(k_min_u32,     I8("v_min_u32", "%8"),           "v_min_u32 %0, %0, %1\n")
KERNEL(k_max_f32,     I8("v_max_f32", "%8"),           "v_max_f32 %0, %0, %1\n")
KERNEL(k_add_f32,     I8("v_add_f32", "%8"),           "v_add_f32 %0, %0, %1\n")
KERNEL(k_max_f16,     I8("v_max_f16", "%8"),           "v_max_f16 %0, %0, %1\n")
KERNEL(k_add_u16,     I8("v_add_u16", "%8"),           "v_add_u16 %0, %0, %1\n")
KERNEL(k_max_u16,     I8("v_max_u16", "%8"),           "v_max_u16 %0, %0, %1\n")
KERNEL(k_mul_u32_u24, I8("v_mul_u32_u24", "%8"),       "v_mul_u32_u24 %0, %0, %1\n")
KERNEL(k_and_or_b32,  I8("v_and_or_b32", "%8, %9"),    "v_and_or_b32 %0, %0, %1, %2\n")
KERNEL(k_alignbit,    I8("v_alignbit_b32", "%8, 8"),   "v_alignbit_b32 %0, %0, %1, 8\n")
KERNEL(k_bfe_u32,     I8("v_bfe_u32", "8, 8"),         "v_bfe_u32 %0, %0, 8, 8\n")
KERNEL(k_or3_b32,     I8("v_or3_b32", "%8, %9"),       "v_or3_b32 %0, %0, %1, %2\n")
KERNEL(k_cvt_f32_f16, I8("v_cvt_f32_f16", ""),         "v_cvt_f32_f16 %0, %0\n")

// round 5: gfx950's three-operand packed f16 max / min, and the three-input boolean op
KERNEL(k_pk_maximum3_f16, I8("v_pk_maximum3_f16", "%8, %9"), "v_pk_maximum3_f16 %0, %0, %1, %2\n")
KERNEL(k_pk_minimum3_f16, I8("v_pk_minimum3_f16", "%8, %9"), "v_pk_minimum3_f16 %0, %0, %1, %2\n")
KERNEL(k_maximum3_f32,    I8("v_maximum3_f32", "%8, %9"),    "v_maximum3_f32 %0, %0, %1, %2\n")
KERNEL(k_bitop3_b32,      I8("v_bitop3_b32", "%8, %9 bitop3:0x96"), "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n")
KERNEL(k_pk_min_i16,      I8("v_pk_min_i16", "%8"),          "v_pk_min_i16 %0, %0, %1\n")
KERNEL(k_pk_mad_i16,      I8("v_pk_mad_i16", "%9, %8"),      "v_pk_mad_i16 %0, %0, %2, %1\n")
KERNEL(k_max3_i16,        I8("v_max3_i16", "%8, %9"),        "v_max3_i16 %0, %0, %1, %2\n")
KERNEL(k_cndmask_sgpr,    I8("v_cndmask_b32", "%8, s[10:11]"), "v_cndmask_b32 %0, %0, %1, s[10:11]\n")

// mixed streams: does a 2-cycle instruction between 4-cycle ones still cost 2?
KERNEL(k_mix_pk_sub, \
	"v_pk_max_i16 %0, %0, %8\n v_sub_u32 %1, %1, %9\n v_pk_max_i16 %2, %2, %8\n v_sub_u32 %3, %3, %9\n" \
	"v_pk_max_i16 %4, %4, %8\n v_sub_u32 %5, %5, %9\n v_pk_max_i16 %6, %6, %8\n v_sub_u32 %7, %7, %9\n", \
	"v_pk_max_i16 %0, %0, %1\n v_sub_u32 %0, %0, %2\n")
KERNEL(k_mix_pk_pk_sub_sub, \
	"v_pk_max_i16 %0, %0, %8\n v_pk_min_u16 %1, %1, %9\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %9\n" \
	"v_pk_max_i16 %4, %4, %8\n v_pk_min_u16 %5, %5, %9\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %9\n", \
	"v_pk_max_i16 %0, %0, %1\n v_sub_u32 %0, %0, %2\n")
KERNEL(k_mix_dep_chain, \
	"v_pk_max_u16 %0, %0, %8\n v_sub_u32 %0, %0, %9\n v_pk_max_u16 %1, %1, %8\n v_sub_u32 %1, %1, %9\n" \
	"v_pk_max_u16 %2, %2, %8\n v_sub_u32 %2, %2, %9\n v_pk_max_u16 %3, %3, %8\n v_sub_u32 %3, %3, %9\n", \
	"v_pk_max_u16 %0, %0, %1\n v_sub_u32 %0, %0, %2\n")
KERNEL(k_mix_dep_chain_pk, \
	"v_pk_max_u16 %0, %0, %8\n v_pk_sub_i16 %0, %0, %9\n v_pk_max_u16 %1, %1, %8\n v_pk_sub_i16 %1, %1, %9\n" \
	"v_pk_max_u16 %2, %2, %8\n v_pk_sub_i16 %2, %2, %9\n v_pk_max_u16 %3, %3, %8\n v_pk_sub_i16 %3, %3, %9\n", \
	"v_pk_max_u16 %0, %0, %1\n v_pk_sub_i16 %0, %0, %2\n")

typedef void (*kern_t)(int, unsigned*, unsigned long long*);
struct Case { const char *name; kern_t k; int per_block_indep; };

int main(){
	hipDeviceProp_t pr;
	if(hipGetDeviceProperties(&pr, 0) != hipSuccess){ fprintf(stderr, "no device\n"); return 1; }
	const int cus = pr.multiProcessorCount;
	printf("# %s, %d CUs, clockRate %d kHz\n", pr.name, cus, pr.clockRate);
	unsigned *out; unsigned long long *cyc;
	const int maxblocks = cus * 8;
	hipMalloc(&out, (size_t)maxblocks * 256 * 4); hipMalloc(&cyc, (size_t)maxblocks * 4 * 8);
	unsigned long long *hc = (unsigned long long*)malloc((size_t)maxblocks * 4 * 8);
	const Case cases[] = {
		{"v_pk_add_i16 clamp", k_pk_add_i16, 8}, {"v_pk_sub_i16", k_pk_sub_i16, 8}, {"v_pk_max_i16", k_pk_max_i16, 8},
		{"v_pk_min_u16", k_pk_min_u16, 8}, {"v_pk_mad_u16", k_pk_mad_u16, 8}, {"v_add_u32", k_add_u32, 8}, {"v_max_i32", k_max_i32, 8},
		{"v_max3_i32", k_max3_i32, 8}, {"v_and_b32", k_and_b32, 8}, {"v_bfi_b32", k_bfi_b32, 8}, {"v_perm_b32", k_perm_b32, 8},
		{"v_add3_u32", k_add3_u32, 8}, {"v_lshl_or_b32", k_lshl_or, 8}, {"v_fma_f32", k_fma_f32, 8},
		{"v_mov_b32_dpp row_shr:1", k_mov_dpp_shr1, 8}, {"v_max_i32_dpp row_shr:1", k_max_i32_dpp, 8},
		{"pk add/max/sub mix", k_chain_mix, 8}, {"v_add_u32 + s_nop 0 (per pair)", k_add_nop, 8},
		{"v_pk_add_f16", k_pk_add_f16, 8}, {"v_pk_max_f16", k_pk_max_f16, 8}, {"v_pk_min_f16", k_pk_min_f16, 8}, {"v_pk_fma_f16", k_pk_fma_f16, 8},
		{"v_pk_mul_f16", k_pk_mul_f16, 8}, {"v_pk_add_u16", k_pk_add_u16, 8}, {"v_pk_max_u16", k_pk_max_u16, 8}, {"v_pk_lshrrev_b16", k_pk_lshr_b16, 8},
		{"v_or_b32", k_or_b32, 8}, {"v_xor_b32", k_xor_b32, 8}, {"v_sub_u32", k_sub_u32, 8}, {"v_lshrrev_b32", k_lshrrev_b32, 8},
		{"v_lshlrev_b32", k_lshlrev_b32, 8}, {"v_ashrrev_i32", k_ashrrev_i32, 8}, {"v_cndmask_b32", k_cndmask, 8}, {"v_min_u32", k_min_u32, 8},
		{"v_max_f32", k_max_f32, 8}, {"v_add_f32", k_add_f32, 8}, {"v_max_f16", k_max_f16, 8}, {"v_add_u16", k_add_u16, 8}, {"v_max_u16", k_max_u16, 8},
		{"v_mul_u32_u24", k_mul_u32_u24, 8}, {"v_and_or_b32", k_and_or_b32, 8}, {"v_alignbit_b32", k_alignbit, 8}, {"v_bfe_u32", k_bfe_u32, 8},
		{"v_or3_b32", k_or3_b32, 8}, {"v_cvt_f32_f16", k_cvt_f32_f16, 8},
		{"v_pk_maximum3_f16", k_pk_maximum3_f16, 8}, {"v_pk_minimum3_f16", k_pk_minimum3_f16, 8}, {"v_maximum3_f32", k_maximum3_f32, 8},
		{"v_bitop3_b32", k_bitop3_b32, 8}, {"v_pk_min_i16", k_pk_min_i16, 8}, {"v_pk_mad_i16", k_pk_mad_i16, 8}, {"v_max3_i16", k_max3_i16, 8},
		{"v_cndmask_b32 (sgpr pair)", k_cndmask_sgpr, 8},
		{"mix pk_max, sub_u32 alternating", k_mix_pk_sub, 8}, {"mix pk, pk, sub, sub", k_mix_pk_pk_sub_sub, 8},
		{"chains: pk_max_u16 -> sub_u32 (x4)", k_mix_dep_chain, 8}, {"chains: pk_max_u16 -> pk_sub_i16 (x4)", k_mix_dep_chain_pk, 8},
	};
	printf("%-34s %6s %10s %10s %12s\n", "instruction", "w/SIMD", "indep cyc", "dep cyc", "ns/instr(i)");
	for(const Case &c : cases){
		for(int wps = 1; wps <= 8; wps *= 2){
			double res[2] = {0, 0}; double nsi = 0;
			for(int dep = 0; dep < 2; dep++){
				if(dep && wps > 1) continue;
				// wps waves per SIMD: blocks of 256 threads = 1 wave per SIMD of a CU; wps blocks per CU
				const int blocks = cus * wps;
				hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
				hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, dep, out, cyc);   // warm-up
				hipEventRecord(e0, 0);
				hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, dep, out, cyc);
				hipEventRecord(e1, 0); hipEventSynchronize(e1);
				float ms = 0; hipEventElapsedTime(&ms, e0, e1);
				hipMemcpy(hc, cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
				double sum = 0; for(int i = 0; i < blocks * 4; i++) sum += (double)hc[i];
				const double ninstr = (double)ITER * 64.0;
				res[dep] = sum / (blocks * 4) / ninstr / (dep ? 1 : wps);
				if(!dep) nsi = ms * 1e6 / ninstr / wps;
				hipEventDestroy(e0); hipEventDestroy(e1);
			}
			if(wps == 1) printf("%-34s %6d %10.2f %10.2f %12.3f\n", c.name, wps, res[0], res[1], nsi);
			else printf("%-34s %6d %10.2f %10s %12.3f\n", c.name, wps, res[0], "", nsi);
		}
	}
	return 0;
}
