// Micro-benchmark: what one SIMD of gfx950 sustains, by instruction and by the number of waves resident on it.
// Answers the questions the VALU roofline of bench.py rests on ("What's weak" 2) and prices every instruction kind of
// the traversal step (pt_kernel.hip.h: trav_step) for a cost model of the step:
//   * how many clocks does a wave64 v_fma_f32 / v_add_f32 / v_mul_f32 occupy its SIMD for -- one wave alone, and 2 or 4 waves taking turns;
//   * the packed FP32 forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) on their own (no MFMA nearby), dependent and independent;
//   * min/max/min3, compares, selects (v_cndmask from vcc / from an SGPR pair / fed by a compare), integer and bit ops, the IEEE-division
//     helpers, cross-lane moves, LDS reads/writes;
//   * scalar instructions (s_and_b64, s_and_saveexec_b64, branches) and what a wave pays when vector and scalar instructions alternate.
// One workgroup per CU (grid = number of CUs), 4 k waves per workgroup = k waves per SIMD (waves of a workgroup go to the SIMDs round robin).
// Every wave runs ITERS x 64 instructions of one kind from inline asm between two s_memtime reads; reported per (op, k):
//   clk_per_inst_wave   -- (t1 - t0) / instructions, the wave's own view (s_memtime ticks = shader clocks)
//   clk_per_inst_simd   -- the same divided by k: what the SIMD spends per instruction when k waves share it
//   g_wave_inst_per_s   -- chip-wide wave-instructions per second from hipEvents around the launch (independent of the tick unit)
// `fma_half` runs with 32 of the 64 lanes enabled (exec = 0x00000000FFFFFFFF): calibration of the lane-utilisation counters
// (rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU on this binary with --only fma / --only fma_half).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu.hip -o tools/ubench/valu ; run: tools/ubench/valu [json-lines file] [--only name]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// operands of every body: %0-%7 eight float accumulators, %8-%15 eight float2 accumulators, %16 an SGPR pair holding a lane mask,
// %17 a scratch SGPR pair, %18 %19 two float inputs, %20 %21 two float2 inputs, %22 this lane's LDS byte address, %23 a 32-bit SGPR value
#define OPERANDS                                                                                                                         \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), \
      "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+s"(smask), "+s"(stmp)                                                \
    : "v"(fx), "v"(fy), "v"(px), "v"(py), "v"(laddr), "s"(sval)                                                                           \
    : "vcc", "scc", "memory"

#define R8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define R8P(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define R64(M) R8(M) R8(M) R8(M) R8(M) R8(M) R8(M) R8(M) R8(M)
#define R64P(M) R8P(M) R8P(M) R8P(M) R8P(M) R8P(M) R8P(M) R8P(M) R8P(M)
#define R32(M) R8(M) R8(M) R8(M) R8(M)

#define KERNEL(NAME, HALF, BODY)                                                                                                   \
    __global__ __launch_bounds__(1024) void NAME(float *out, unsigned long long *cyc, unsigned long long *rt, int iters, float seed) {                     \
        __shared__ float lds[2 * 1024 + 64];                                                                                       \
        const float fx = seed + (float)threadIdx.x * 1e-9f, fy = 1.0f - seed * 1e-7f;                                              \
        float a[8];                                                                                                                \
        v2f p[8];                                                                                                                  \
        for (int i = 0; i < 8; i++) { a[i] = seed * (float)(i + 1); p[i].x = a[i]; p[i].y = a[i] + 1.0f; }                         \
        v2f px; px.x = fx; px.y = fx + 1e-9f;                                                                                      \
        v2f py; py.x = fy; py.y = fy;                                                                                              \
        lds[2 * threadIdx.x] = fx; lds[2 * threadIdx.x + 1] = fy;                                                                  \
        const unsigned laddr = (unsigned)(uintptr_t)&lds[2 * threadIdx.x];                                                         \
        unsigned long long smask = 0x5555555555555555ull, stmp = 0;                                                                \
        const unsigned sval = (unsigned)iters * 4u;                                                                                \
        __syncthreads();                                                                                                           \
        unsigned long long save_exec = 0;                                                                                          \
        if (HALF) asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffffffff" : "=s"(save_exec));                             \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int it = 0; it < iters; it++) asm volatile(BODY OPERANDS);                                                            \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                                            \
        if (HALF) asm volatile("s_mov_b64 exec, %0" : : "s"(save_exec));                                                           \
        float s = (float)(smask & 1ull) + (float)(stmp & 1ull);                                                                    \
        for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                            \
        if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                         \
                                       rt[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r1 - r0; }                        \
    }

// ---- vector ALU, independent (instruction j works on accumulator j mod 8) and dependent (all on accumulator 0)
#define I_FMA(n) "v_fma_f32 %" #n ", %18, %19, %" #n "\n\t"
#define D_FMA(n) "v_fma_f32 %0, %18, %19, %0\n\t"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %18, %19\n\t"
#define I_ADD(n) "v_add_f32 %" #n ", %18, %" #n "\n\t"
#define D_ADD(n) "v_add_f32 %0, %18, %0\n\t"
#define I_SUB(n) "v_sub_f32 %" #n ", %18, %" #n "\n\t"
#define I_MUL(n) "v_mul_f32 %" #n ", %18, %" #n "\n\t"
#define D_MUL(n) "v_mul_f32 %0, %18, %0\n\t"
#define I_MUL_S(n) "v_mul_f32 %" #n ", %23, %" #n "\n\t"  // one SGPR source
#define I_MAX(n) "v_max_f32 %" #n ", %18, %" #n "\n\t"
#define D_MAX(n) "v_max_f32 %0, %18, %0\n\t"
#define I_MIN(n) "v_min_f32 %" #n ", %18, %" #n "\n\t"
#define I_MIN3(n) "v_min3_f32 %" #n ", %18, %19, %" #n "\n\t"
#define I_MAX3(n) "v_max3_f32 %" #n ", %18, %19, %" #n "\n\t"
#define I_MED3(n) "v_med3_f32 %" #n ", %18, %19, %" #n "\n\t"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %20, %21, %" #n "\n\t"
#define D_PKFMA(n) "v_pk_fma_f32 %8, %20, %21, %8\n\t"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %20, %" #n "\n\t"
#define D_PKMUL(n) "v_pk_mul_f32 %8, %20, %8\n\t"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %20, %" #n "\n\t"
#define D_PKADD(n) "v_pk_add_f32 %8, %20, %8\n\t"
#define I_PKMUL_OPSEL(n) "v_pk_mul_f32 %" #n ", %20, %" #n " op_sel_hi:[0,1]\n\t"  // both results take the LOW half of source 0 (the broadcast a slab test would use)
#define I_PKMUL_S(n) "v_pk_mul_f32 %" #n ", %16, %" #n "\n\t"  // one SGPR-pair source (what a list scan would use: the record is in SGPRs)
#define I_PKADD_S(n) "v_pk_add_f32 %" #n ", %" #n ", %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"  // x - (SGPR pair)
#define I_SUB_S(n) "v_subrev_f32 %" #n ", %23, %" #n "\n\t"  // one SGPR source
#define I_FMA_S(n) "v_fma_f32 %" #n ", %23, %19, %" #n "\n\t"  // one SGPR source
#define I_PKMOV(n) "v_pk_mov_b32 %" #n ", %20, %21\n\t"
#define I_MOV(n) "v_mov_b32 %" #n ", %18\n\t"
#define I_MOV_DPP(n) "v_mov_b32_dpp %" #n ", %18 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_CND_VCC(n) "v_cndmask_b32 %" #n ", %18, %" #n ", vcc\n\t"              // vcc never written inside the loop
#define I_CND_SGPR(n) "v_cndmask_b32_e64 %" #n ", %18, %" #n ", %16\n\t"         // mask in an SGPR pair
#define I_CND_CONST(n) "v_cndmask_b32_e64 %" #n ", 0, 1, %16\n\t"                // the bool -> int materialisation the compiler emits
#define I_CMP_CND(n) "v_cmp_lt_f32 vcc, %18, %" #n "\n\tv_cndmask_b32 %" #n ", %19, %" #n ", vcc\n\t"  // compare feeding a select: 2 instructions
#define I_CMP_VCC(n) "v_cmp_lt_f32 vcc, %18, %" #n "\n\t"
#define I_CMP_SGPR(n) "v_cmp_lt_f32 %17, %18, %" #n "\n\t"
#define I_CMP_I32(n) "v_cmp_gt_i32 %17, 0, %" #n "\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %18, 3, %" #n "\n\t"
#define I_LSHLADD_S(n) "v_lshl_add_u32 %" #n ", %18, 6, %23\n\t"
#define I_ADDU(n) "v_add_u32 %" #n ", %18, %" #n "\n\t"
#define I_AND(n) "v_and_b32 %" #n ", %18, %" #n "\n\t"
#define I_NOT(n) "v_not_b32 %" #n ", %" #n "\n\t"
#define I_BFREV(n) "v_bfrev_b32 %" #n ", 1\n\t"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
#define I_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n\t"
#define I_DIVSCALE(n) "v_div_scale_f32 %" #n ", vcc, %18, %19, %18\n\t"
#define I_DIVFMAS(n) "v_div_fmas_f32 %" #n ", %18, %19, %" #n "\n\t"
#define I_DIVFIXUP(n) "v_div_fixup_f32 %" #n ", %18, %19, %" #n "\n\t"
#define I_READLANE(n) "v_readlane_b32 vcc_lo, %" #n ", 3\n\t"
#define I_READFIRST(n) "v_readfirstlane_b32 vcc_lo, %" #n "\n\t"
// exec-masked move as a select: 3 instructions (scalar, vector, scalar)
#define I_EXEC_MOV(n) "s_and_saveexec_b64 %17, %16\n\tv_mov_b32 %" #n ", %18\n\ts_mov_b64 exec, %17\n\t"
// ---- LDS
#define I_DSR64(n) "ds_read_b64 %" #n ", %22\n\t"
#define I_DSW64(n) "ds_write_b64 %22, %" #n "\n\t"
#define I_BPERM(n) "ds_bpermute_b32 %" #n ", %22, %" #n "\n\t"
#define WAIT_LGKM "s_waitcnt lgkmcnt(0)\n\t"
// ---- scalar
#define I_SAND(n) "s_and_b64 %17, %16, %17\n\t"
#define I_SOR(n) "s_or_b64 %17, %16, %17\n\t"
#define I_SMOV(n) "s_mov_b64 %17, %16\n\t"
#define I_SAVEEXEC(n) "s_and_saveexec_b64 %17, exec\n\t"          // exec unchanged (all lanes stay on)
#define I_SNOP(n) "s_nop 0\n\t"
#define I_SWAIT(n) "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define I_SBR_NOT(n) "s_cbranch_execz 1f\n\t"                     // never taken
#define I_SBR_TAKEN(n) "s_branch 2f\n\t2:\n\t"                    // always taken, to the next instruction
// ---- alternating vector / scalar in ONE wave
#define I_V_S(n) "v_mul_f32 %" #n ", %18, %" #n "\n\ts_and_b64 %17, %16, %17\n\t"
#define I_V_S_S(n) "v_mul_f32 %" #n ", %18, %" #n "\n\ts_and_b64 %17, %16, %17\n\ts_or_b64 %17, %16, %17\n\t"
#define I_MUL_MAX(n) "v_mul_f32 %" #n ", %18, %" #n "\n\tv_max_f32 %" #n ", %19, %" #n "\n\t"

KERNEL(k_fma, false, R64(I_FMA))
KERNEL(k_fma_dep, false, R64(D_FMA))
KERNEL(k_fma_half, true, R64(I_FMA))
KERNEL(k_fmac, false, R64(I_FMAC))
KERNEL(k_add, false, R64(I_ADD))
KERNEL(k_add_dep, false, R64(D_ADD))
KERNEL(k_sub, false, R64(I_SUB))
KERNEL(k_mul, false, R64(I_MUL))
KERNEL(k_mul_dep, false, R64(D_MUL))
KERNEL(k_mul_sgpr, false, R64(I_MUL_S))
KERNEL(k_max, false, R64(I_MAX))
KERNEL(k_max_dep, false, R64(D_MAX))
KERNEL(k_min, false, R64(I_MIN))
KERNEL(k_min3, false, R64(I_MIN3))
KERNEL(k_max3, false, R64(I_MAX3))
KERNEL(k_med3, false, R64(I_MED3))
KERNEL(k_pkfma, false, R64P(I_PKFMA))
KERNEL(k_pkfma_dep, false, R64P(D_PKFMA))
KERNEL(k_pkmul, false, R64P(I_PKMUL))
KERNEL(k_pkmul_s, false, R64P(I_PKMUL_S))
KERNEL(k_pkadd_s, false, R64P(I_PKADD_S))
KERNEL(k_sub_s, false, R64(I_SUB_S))
KERNEL(k_fma_s, false, R64(I_FMA_S))
KERNEL(k_pkmul_dep, false, R64P(D_PKMUL))
KERNEL(k_pkadd, false, R64P(I_PKADD))
KERNEL(k_pkadd_dep, false, R64P(D_PKADD))
KERNEL(k_pkmul_opsel, false, R64P(I_PKMUL_OPSEL))
KERNEL(k_pkmov, false, R64P(I_PKMOV))
KERNEL(k_mov, false, R64(I_MOV))
KERNEL(k_mov_dpp, false, R64(I_MOV_DPP))
KERNEL(k_cnd_vcc, false, R64(I_CND_VCC))
KERNEL(k_cnd_sgpr, false, R64(I_CND_SGPR))
KERNEL(k_cnd_const, false, R64(I_CND_CONST))
KERNEL(k_cmp_cnd, false, R32(I_CMP_CND))
KERNEL(k_cmp_vcc, false, R64(I_CMP_VCC))
KERNEL(k_cmp_sgpr, false, R64(I_CMP_SGPR))
KERNEL(k_cmp_i32, false, R64(I_CMP_I32))
KERNEL(k_lshladd, false, R64(I_LSHLADD))
KERNEL(k_lshladd_s, false, R64(I_LSHLADD_S))
KERNEL(k_addu, false, R64(I_ADDU))
KERNEL(k_and, false, R64(I_AND))
KERNEL(k_not, false, R64(I_NOT))
KERNEL(k_bfrev, false, R64(I_BFREV))
KERNEL(k_rcp, false, R64(I_RCP))
KERNEL(k_sqrt, false, R64(I_SQRT))
KERNEL(k_divscale, false, R64(I_DIVSCALE))
KERNEL(k_divfmas, false, R64(I_DIVFMAS))
KERNEL(k_divfixup, false, R64(I_DIVFIXUP))
KERNEL(k_readlane, false, R64(I_READLANE))
KERNEL(k_readfirst, false, R64(I_READFIRST))
KERNEL(k_exec_mov, false, R8(I_EXEC_MOV) R8(I_EXEC_MOV) R8(I_EXEC_MOV) R8(I_EXEC_MOV) R8(I_EXEC_MOV) R8(I_EXEC_MOV) R8(I_EXEC_MOV) R8(I_EXEC_MOV))
KERNEL(k_dsr64, false, R8P(I_DSR64) WAIT_LGKM R8P(I_DSR64) WAIT_LGKM R8P(I_DSR64) WAIT_LGKM R8P(I_DSR64) WAIT_LGKM R8P(I_DSR64) WAIT_LGKM R8P(I_DSR64) WAIT_LGKM R8P(I_DSR64) WAIT_LGKM R8P(I_DSR64) WAIT_LGKM)
KERNEL(k_dsw64, false, R64P(I_DSW64) WAIT_LGKM)
KERNEL(k_bperm, false, R8(I_BPERM) WAIT_LGKM R8(I_BPERM) WAIT_LGKM R8(I_BPERM) WAIT_LGKM R8(I_BPERM) WAIT_LGKM R8(I_BPERM) WAIT_LGKM R8(I_BPERM) WAIT_LGKM R8(I_BPERM) WAIT_LGKM R8(I_BPERM) WAIT_LGKM)
KERNEL(k_sand, false, R64(I_SAND))
KERNEL(k_sor, false, R64(I_SOR))
KERNEL(k_smov, false, R64(I_SMOV))
KERNEL(k_saveexec, false, R64(I_SAVEEXEC))
KERNEL(k_snop, false, R64(I_SNOP))
KERNEL(k_swait, false, R64(I_SWAIT))
KERNEL(k_sbr_not, false, R64(I_SBR_NOT) "1:\n\t")
KERNEL(k_sbr_taken, false, R64(I_SBR_TAKEN))
KERNEL(k_v_s, false, R64(I_V_S))
KERNEL(k_v_s_s, false, R64(I_V_S_S))
KERNEL(k_mul_max, false, R32(I_MUL_MAX))

typedef void (*Kern)(float *, unsigned long long *, unsigned long long *, int, float);
struct Entry { const char *key, *what; Kern k; int insts; };  // insts: instructions per loop trip
static const Entry kTable[] = {
    {"fma", "v_fma_f32", k_fma, 64}, {"fma_dep", "v_fma_f32, dependent chain", k_fma_dep, 64}, {"fma_half", "v_fma_f32, 32 of 64 lanes enabled", k_fma_half, 64},
    {"fmac", "v_fmac_f32", k_fmac, 64}, {"add", "v_add_f32", k_add, 64}, {"add_dep", "v_add_f32, dependent chain", k_add_dep, 64}, {"sub", "v_sub_f32", k_sub, 64},
    {"mul", "v_mul_f32", k_mul, 64}, {"mul_dep", "v_mul_f32, dependent chain", k_mul_dep, 64}, {"mul_sgpr", "v_mul_f32, one SGPR source", k_mul_sgpr, 64},
    {"max", "v_max_f32", k_max, 64}, {"max_dep", "v_max_f32, dependent chain", k_max_dep, 64}, {"min", "v_min_f32", k_min, 64}, {"min3", "v_min3_f32", k_min3, 64},
    {"max3", "v_max3_f32", k_max3, 64}, {"med3", "v_med3_f32", k_med3, 64},
    {"pk_fma", "v_pk_fma_f32", k_pkfma, 64}, {"pk_fma_dep", "v_pk_fma_f32, dependent chain", k_pkfma_dep, 64}, {"pk_mul", "v_pk_mul_f32", k_pkmul, 64},
    {"pk_mul_dep", "v_pk_mul_f32, dependent chain", k_pkmul_dep, 64}, {"pk_mul_sgpr", "v_pk_mul_f32, one SGPR-pair source", k_pkmul_s, 64},
    {"pk_add_sgpr", "v_pk_add_f32, minus an SGPR pair", k_pkadd_s, 64}, {"sub_sgpr", "v_subrev_f32, one SGPR source", k_sub_s, 64}, {"fma_sgpr", "v_fma_f32, one SGPR source", k_fma_s, 64}, {"pk_add", "v_pk_add_f32", k_pkadd, 64}, {"pk_add_dep", "v_pk_add_f32, dependent chain", k_pkadd_dep, 64},
    {"pk_mul_opsel", "v_pk_mul_f32 op_sel_hi:[0,1] (broadcast of one half)", k_pkmul_opsel, 64}, {"pk_mov", "v_pk_mov_b32", k_pkmov, 64},
    {"mov", "v_mov_b32", k_mov, 64}, {"mov_dpp", "v_mov_b32_dpp quad_perm", k_mov_dpp, 64},
    {"cndmask_vcc", "v_cndmask_b32 (vcc, not written in the loop)", k_cnd_vcc, 64}, {"cndmask_sgpr", "v_cndmask_b32_e64 (mask in an SGPR pair)", k_cnd_sgpr, 64},
    {"cndmask_const", "v_cndmask_b32_e64 v, 0, 1, s[..]", k_cnd_const, 64}, {"cmp_cndmask", "v_cmp_lt_f32 vcc + v_cndmask_b32 vcc pairs", k_cmp_cnd, 64},
    {"cmp_vcc", "v_cmp_lt_f32 to vcc", k_cmp_vcc, 64}, {"cmp_sgpr", "v_cmp_lt_f32 to an SGPR pair", k_cmp_sgpr, 64}, {"cmp_i32", "v_cmp_gt_i32 to an SGPR pair", k_cmp_i32, 64},
    {"lshl_add", "v_lshl_add_u32", k_lshladd, 64}, {"lshl_add_sgpr", "v_lshl_add_u32 with an SGPR addend", k_lshladd_s, 64}, {"add_u32", "v_add_u32", k_addu, 64},
    {"and", "v_and_b32", k_and, 64}, {"not", "v_not_b32", k_not, 64}, {"bfrev", "v_bfrev_b32 (constant materialisation)", k_bfrev, 64},
    {"rcp", "v_rcp_f32", k_rcp, 64}, {"sqrt", "v_sqrt_f32", k_sqrt, 64}, {"div_scale", "v_div_scale_f32", k_divscale, 64}, {"div_fmas", "v_div_fmas_f32", k_divfmas, 64},
    {"div_fixup", "v_div_fixup_f32", k_divfixup, 64}, {"readlane", "v_readlane_b32", k_readlane, 64}, {"readfirstlane", "v_readfirstlane_b32", k_readfirst, 64},
    {"exec_mov", "s_and_saveexec_b64 + v_mov_b32 + s_mov_b64 exec (a select by exec mask): per TRIPLE", k_exec_mov, 64},
    {"ds_read_b64", "ds_read_b64", k_dsr64, 64}, {"ds_write_b64", "ds_write_b64", k_dsw64, 64}, {"ds_bpermute", "ds_bpermute_b32 (__shfl)", k_bperm, 64},
    {"s_and", "s_and_b64", k_sand, 64}, {"s_or", "s_or_b64", k_sor, 64}, {"s_mov", "s_mov_b64", k_smov, 64}, {"s_saveexec", "s_and_saveexec_b64", k_saveexec, 64},
    {"s_nop", "s_nop 0", k_snop, 64}, {"s_waitcnt", "s_waitcnt (nothing outstanding)", k_swait, 64}, {"s_cbranch_not_taken", "s_cbranch_execz, not taken", k_sbr_not, 64},
    {"s_branch_taken", "s_branch to the next instruction", k_sbr_taken, 64},
    {"v_s", "v_mul_f32 + s_and_b64 alternating in one wave: per PAIR", k_v_s, 64}, {"v_s_s", "v_mul_f32 + s_and_b64 + s_or_b64 in one wave: per TRIPLE", k_v_s_s, 64},
    {"mul_max", "v_mul_f32 + v_max_f32 alternating: per instruction", k_mul_max, 64},
};

// Round 4: every (op, k) runs as a train of back-to-back launches (no host sync in between) lasting >= `sustain` seconds -- 2 s for the
// instructions the VALU roofline rests on (fma / add / mul and what the traversal step is made of), 0.5 s for the rest; --sustain S overrides --
// and the clock the chip holds under that load is measured IN the kernel: delta s_memtime / delta s_memrealtime x 100 MHz, median over the
// waves of the last launch (MI355X_MICROARCH.md, DVFS item 6).  G/s is at that clock; clk per instruction is the clock-free figure.
int main(int argc, char **argv) {
    FILE *js = nullptr;
    std::string only;
    double sustain = -1.0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
        else if (!strcmp(argv[i], "--sustain") && i + 1 < argc) sustain = atof(argv[++i]);
        else js = fopen(argv[i], "a");
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int iters = 2048;
    float *out; unsigned long long *cyc, *rt;
    CK(hipMalloc(&out, (size_t)n_cu * 1024 * 4)); CK(hipMalloc(&cyc, (size_t)n_cu * 16 * 8)); CK(hipMalloc(&rt, (size_t)n_cu * 16 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d CUs, nominal clock %.0f MHz (device property); in-kernel clock measured per case\n", n_cu, prop.clockRate / 1000.0);
    static const char *kKey[] = {"fma", "fma_half", "add", "mul", "max", "min3", "cmp_sgpr", "cndmask_sgpr", "lshl_add", "mul_max", "s_and", "v_s"};
    for (const Entry &e : kTable) {
        if (!only.empty() && only != e.key) continue;
        bool key = false;
        for (const char *kk : kKey) key = key || !strcmp(kk, e.key);
        const double T = sustain >= 0 ? sustain : (key ? 2.0 : 0.5);
        for (int k = 1; k <= 4; k *= 2) {
            const int threads = 256 * k;
            std::vector<unsigned long long> h((size_t)n_cu * 4 * k), hr((size_t)n_cu * 4 * k);
            float ms1;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.k, dim3(n_cu), dim3(threads), 0, 0, out, cyc, rt, iters, 0.37f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms1, e0, e1));
            const int n = std::max(3, (int)(T * 1e3 / std::max(ms1, 0.05f)) + 1);
            CK(hipEventRecord(e0));
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(e.k, dim3(n_cu), dim3(threads), 0, 0, out, cyc, rt, iters, 0.37f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms_all; CK(hipEventElapsedTime(&ms_all, e0, e1));
            const double ms = ms_all / n;
            CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), rt, hr.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> mhz(h.size());
            for (size_t i = 0; i < h.size(); i++) mhz[i] = hr[i] ? (double)h[i] / (double)hr[i] * 100.0 : 0.0;
            std::sort(mhz.begin(), mhz.end());
            const double clock_mhz = mhz[mhz.size() / 2];
            std::sort(h.begin(), h.end());
            const double clk_wave = (double)h[h.size() / 2] / ((double)iters * e.insts);  // median wave
            const double insts = (double)n_cu * 4 * k * iters * e.insts;
            printf("%-72s %d wave(s)/SIMD: %6.2f clk per wave, %6.2f clk per SIMD, in-kernel clock %6.0f MHz, %7.1f G/s chip-wide at that clock (%.3f ms x %d)\n", e.what, k,
                   clk_wave, clk_wave / k, clock_mhz, insts / (ms * 1e-3) * 1e-9, ms, n);
            fflush(stdout);
            if (js) fprintf(js, "{\"op\": \"%s\", \"what\": \"%s\", \"waves_per_simd\": %d, \"clk_per_inst_wave\": %.3f, \"clk_per_inst_simd\": %.3f, \"clock_mhz_in_kernel\": %.1f, "
                                "\"g_wave_inst_per_s\": %.2f, \"ms_per_launch\": %.4f, \"launches\": %d, \"simds\": %d}\n",
                            e.key, e.what, k, clk_wave, clk_wave / k, clock_mhz, insts / (ms * 1e-3) * 1e-9, ms, n, n_cu * 4);
        }
    }
    if (js) fclose(js);
    return 0;
}
