// Issue-rate probe for gfx950 vector instructions: cycles per wave-instruction on one SIMD at 1, 2,
// 4 and 8 resident waves per SIMD.  Development tool (not part of the library).
//   hipcc -O2 --offload-arch=gfx950 -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define REP8(X) X X X X X X X X
#define ITERS 2000

#define PROBE_KERNEL(NAME, ASM8)                                                                     \
    __global__ __launch_bounds__(512) void NAME(float* out, float seed, unsigned long long* cyc)    \
    {                                                                                                \
        if (seed < 0.0f && (threadIdx.x & 32)) return;      /* half_wave: upper 32 lanes leave */       \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = seed * 0.5f + 1.0f, c = seed + 0.25f;                                              \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, e = b;                                            \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int i = 0; i < ITERS; i++) {                                                            \
            asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                                "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)                               \
                              : "v"(b), "v"(c), "v"(e) : "vcc", "s2", "s3", "s4", "s10", "s11");                                     \
        }                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3); \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                     \
    }

// operands: %0..%7 floats, %8..%11 doubles, %12 b, %13 c, %14 e(double)
#define I8(OP, TAIL) \
    OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
    OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
#define I8_3(OP) \
    OP " %0, %0, %12, %13\n" OP " %1, %1, %12, %13\n" OP " %2, %2, %12, %13\n" OP " %3, %3, %12, %13\n" \
    OP " %4, %4, %12, %13\n" OP " %5, %5, %12, %13\n" OP " %6, %6, %12, %13\n" OP " %7, %7, %12, %13\n"
#define D4_3(OP) \
    OP " %8, %8, %14, %14\n" OP " %9, %9, %14, %14\n" OP " %10, %10, %14, %14\n" OP " %11, %11, %14, %14\n"
#define D4_2(OP) \
    OP " %8, %8, %14\n" OP " %9, %9, %14\n" OP " %10, %10, %14\n" OP " %11, %11, %14\n"

PROBE_KERNEL(k_fma_f32, REP8(I8_3("v_fma_f32")))
PROBE_KERNEL(k_add_f32, REP8(I8("v_add_f32", ", %12")))
PROBE_KERNEL(k_mul_f32, REP8(I8("v_mul_f32", ", %12")))
PROBE_KERNEL(k_max_f32, REP8(I8("v_max_f32", ", %12")))
PROBE_KERNEL(k_and_b32, REP8(I8("v_and_b32", ", %12")))
PROBE_KERNEL(k_lshl_b32, REP8(I8("v_lshlrev_b32", ", 1")))
PROBE_KERNEL(k_add_u32, REP8(I8("v_add_u32", ", %12")))
PROBE_KERNEL(k_mov_b32, REP8(I8("v_mov_b32", "")))
PROBE_KERNEL(k_cndmask, REP8(I8("v_cndmask_b32", ", %12, vcc")))
PROBE_KERNEL(k_cmp_f32, REP8("v_cmp_lt_f32 vcc, %0, %12\nv_cmp_lt_f32 vcc, %1, %12\nv_cmp_lt_f32 vcc, %2, %12\nv_cmp_lt_f32 vcc, %3, %12\nv_cmp_lt_f32 vcc, %4, %12\nv_cmp_lt_f32 vcc, %5, %12\nv_cmp_lt_f32 vcc, %6, %12\nv_cmp_lt_f32 vcc, %7, %12\n"))
PROBE_KERNEL(k_mul_lo_u32, REP8(I8("v_mul_lo_u32", ", %12")))
PROBE_KERNEL(k_mul_hi_u32, REP8(I8("v_mul_hi_u32", ", %12")))
PROBE_KERNEL(k_mad_u32_u24, REP8(I8_3("v_mad_u32_u24")))
PROBE_KERNEL(k_rcp_f32, REP8(I8("v_rcp_f32", "")))
PROBE_KERNEL(k_sqrt_f32, REP8(I8("v_sqrt_f32", "")))
PROBE_KERNEL(k_div_fixup, REP8(I8_3("v_div_fixup_f32")))
PROBE_KERNEL(k_div_fmas, REP8(I8_3("v_div_fmas_f32")))
PROBE_KERNEL(k_div_scale, REP8("v_div_scale_f32 %0, vcc, %0, %12, %13\nv_div_scale_f32 %1, vcc, %1, %12, %13\nv_div_scale_f32 %2, vcc, %2, %12, %13\nv_div_scale_f32 %3, vcc, %3, %12, %13\nv_div_scale_f32 %4, vcc, %4, %12, %13\nv_div_scale_f32 %5, vcc, %5, %12, %13\nv_div_scale_f32 %6, vcc, %6, %12, %13\nv_div_scale_f32 %7, vcc, %7, %12, %13\n"))
PROBE_KERNEL(k_pk_fma_f32, REP8(D4_3("v_pk_fma_f32") D4_3("v_pk_fma_f32")))
PROBE_KERNEL(k_pk_mul_f32, REP8(D4_2("v_pk_mul_f32") D4_2("v_pk_mul_f32")))
PROBE_KERNEL(k_fma_f64, REP8(D4_3("v_fma_f64") D4_3("v_fma_f64")))
PROBE_KERNEL(k_mul_f64, REP8(D4_2("v_mul_f64") D4_2("v_mul_f64")))
PROBE_KERNEL(k_add_f64, REP8(D4_2("v_add_f64") D4_2("v_add_f64")))
PROBE_KERNEL(k_cvt_f64_f32, REP8("v_cvt_f64_f32 %8, %0\nv_cvt_f64_f32 %9, %1\nv_cvt_f64_f32 %10, %2\nv_cvt_f64_f32 %11, %3\nv_cvt_f64_f32 %8, %4\nv_cvt_f64_f32 %9, %5\nv_cvt_f64_f32 %10, %6\nv_cvt_f64_f32 %11, %7\n"))
PROBE_KERNEL(k_cvt_f32_f64, REP8("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %8\nv_cvt_f32_f64 %5, %9\nv_cvt_f32_f64 %6, %10\nv_cvt_f32_f64 %7, %11\n"))
PROBE_KERNEL(k_cndmask_sgpr, REP8(I8("v_cndmask_b32", ", %12, s[2:3]")))
PROBE_KERNEL(k_cmp_sgpr, REP8("v_cmp_lt_f32 s[10:11], %0, %12\nv_cmp_lt_f32 s[10:11], %1, %12\nv_cmp_lt_f32 s[10:11], %2, %12\nv_cmp_lt_f32 s[10:11], %3, %12\nv_cmp_lt_f32 s[10:11], %4, %12\nv_cmp_lt_f32 s[10:11], %5, %12\nv_cmp_lt_f32 s[10:11], %6, %12\nv_cmp_lt_f32 s[10:11], %7, %12\n"))
PROBE_KERNEL(k_cmp_cnd_pair, REP8("v_cmp_lt_f32 vcc, %0, %12\nv_cndmask_b32 %1, %1, %12, vcc\nv_cmp_lt_f32 vcc, %2, %12\nv_cndmask_b32 %3, %3, %12, vcc\nv_cmp_lt_f32 vcc, %4, %12\nv_cndmask_b32 %5, %5, %12, vcc\nv_cmp_lt_f32 vcc, %6, %12\nv_cndmask_b32 %7, %7, %12, vcc\n"))
PROBE_KERNEL(k_min3_f32, REP8(I8_3("v_min3_f32")))
PROBE_KERNEL(k_med3_f32, REP8(I8_3("v_med3_f32")))
PROBE_KERNEL(k_min_f32, REP8(I8("v_min_f32", ", %12")))
PROBE_KERNEL(k_xor_b32, REP8(I8("v_xor_b32", ", %12")))
PROBE_KERNEL(k_or_b32, REP8(I8("v_or_b32", ", %12")))
PROBE_KERNEL(k_sub_f32, REP8(I8("v_sub_f32", ", %12")))
PROBE_KERNEL(k_fmac_f32, REP8(I8("v_fmac_f32", ", %12")))
PROBE_KERNEL(k_mul_f32_abs, REP8("v_mul_f32 %0, |%0|, %12\nv_mul_f32 %1, |%1|, %12\nv_mul_f32 %2, |%2|, %12\nv_mul_f32 %3, |%3|, %12\nv_mul_f32 %4, |%4|, %12\nv_mul_f32 %5, |%5|, %12\nv_mul_f32 %6, |%6|, %12\nv_mul_f32 %7, |%7|, %12\n"))
PROBE_KERNEL(k_mul_f32_sgpr, REP8("v_mul_f32 %0, s4, %0\nv_mul_f32 %1, s4, %1\nv_mul_f32 %2, s4, %2\nv_mul_f32 %3, s4, %3\nv_mul_f32 %4, s4, %4\nv_mul_f32 %5, s4, %5\nv_mul_f32 %6, s4, %6\nv_mul_f32 %7, s4, %7\n"))
PROBE_KERNEL(k_mul_f32_lit, REP8("v_mul_f32 %0, 0x3f8ccccd, %0\nv_mul_f32 %1, 0x3f8ccccd, %1\nv_mul_f32 %2, 0x3f8ccccd, %2\nv_mul_f32 %3, 0x3f8ccccd, %3\nv_mul_f32 %4, 0x3f8ccccd, %4\nv_mul_f32 %5, 0x3f8ccccd, %5\nv_mul_f32 %6, 0x3f8ccccd, %6\nv_mul_f32 %7, 0x3f8ccccd, %7\n"))
PROBE_KERNEL(k_fma_f32_sgpr, REP8("v_fma_f32 %0, %0, s4, %13\nv_fma_f32 %1, %1, s4, %13\nv_fma_f32 %2, %2, s4, %13\nv_fma_f32 %3, %3, s4, %13\nv_fma_f32 %4, %4, s4, %13\nv_fma_f32 %5, %5, s4, %13\nv_fma_f32 %6, %6, s4, %13\nv_fma_f32 %7, %7, s4, %13\n"))
PROBE_KERNEL(k_alignbit, REP8(I8_3("v_alignbit_b32")))
PROBE_KERNEL(k_lshrrev_b32, REP8(I8("v_lshrrev_b32", ", 3")))
PROBE_KERNEL(k_lshrrev_b64, REP8("v_lshrrev_b64 %8, 3, %8\nv_lshrrev_b64 %9, 3, %9\nv_lshrrev_b64 %10, 3, %10\nv_lshrrev_b64 %11, 3, %11\nv_lshrrev_b64 %8, 3, %8\nv_lshrrev_b64 %9, 3, %9\nv_lshrrev_b64 %10, 3, %10\nv_lshrrev_b64 %11, 3, %11\n"))
PROBE_KERNEL(k_cvt_f32_u32, REP8(I8("v_cvt_f32_u32", "")))
PROBE_KERNEL(k_add3_u32, REP8(I8_3("v_add3_u32")))
PROBE_KERNEL(k_lshl_add_u32, REP8(I8_3("v_lshl_add_u32")))
PROBE_KERNEL(k_add_co_u32, REP8("v_add_co_u32 %0, vcc, %0, %12\nv_add_co_u32 %1, vcc, %1, %12\nv_add_co_u32 %2, vcc, %2, %12\nv_add_co_u32 %3, vcc, %3, %12\nv_add_co_u32 %4, vcc, %4, %12\nv_add_co_u32 %5, vcc, %5, %12\nv_add_co_u32 %6, vcc, %6, %12\nv_add_co_u32 %7, vcc, %7, %12\n"))
PROBE_KERNEL(k_mov_b64, REP8("v_mov_b64 %8, %14\nv_mov_b64 %9, %14\nv_mov_b64 %10, %14\nv_mov_b64 %11, %14\nv_mov_b64 %8, %14\nv_mov_b64 %9, %14\nv_mov_b64 %10, %14\nv_mov_b64 %11, %14\n"))
PROBE_KERNEL(k_ldexp_f32, REP8(I8("v_ldexp_f32", ", %12")))
PROBE_KERNEL(k_cmp_class, REP8("v_cmp_class_f32 vcc, %0, %12\nv_cmp_class_f32 vcc, %1, %12\nv_cmp_class_f32 vcc, %2, %12\nv_cmp_class_f32 vcc, %3, %12\nv_cmp_class_f32 vcc, %4, %12\nv_cmp_class_f32 vcc, %5, %12\nv_cmp_class_f32 vcc, %6, %12\nv_cmp_class_f32 vcc, %7, %12\n"))
// dependent chain: every instruction reads the previous one's result
PROBE_KERNEL(k_fma_f32_dep, REP8("v_fma_f32 %0, %0, %12, %13\nv_fma_f32 %0, %0, %12, %13\nv_fma_f32 %0, %0, %12, %13\nv_fma_f32 %0, %0, %12, %13\nv_fma_f32 %0, %0, %12, %13\nv_fma_f32 %0, %0, %12, %13\nv_fma_f32 %0, %0, %12, %13\nv_fma_f32 %0, %0, %12, %13\n"))
PROBE_KERNEL(k_mad_u64_u32, REP8("v_mad_u64_u32 %8, vcc, %0, %12, %8\nv_mad_u64_u32 %9, vcc, %1, %12, %9\nv_mad_u64_u32 %10, vcc, %2, %12, %10\nv_mad_u64_u32 %11, vcc, %3, %12, %11\nv_mad_u64_u32 %8, vcc, %4, %12, %8\nv_mad_u64_u32 %9, vcc, %5, %12, %9\nv_mad_u64_u32 %10, vcc, %6, %12, %10\nv_mad_u64_u32 %11, vcc, %7, %12, %11\n"))

typedef void (*K)(float*, float, unsigned long long*);
struct Probe { const char* name; K k; };

int main()
{
    Probe probes[] = {
        {"v_fma_f32", k_fma_f32}, {"v_fma_f32 (dependent)", k_fma_f32_dep}, {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32},
        {"v_max_f32", k_max_f32}, {"v_and_b32", k_and_b32}, {"v_lshlrev_b32", k_lshl_b32}, {"v_add_u32", k_add_u32}, {"v_mov_b32", k_mov_b32},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32", k_cmp_f32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
        {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32},
        {"v_div_scale_f32", k_div_scale}, {"v_div_fmas_f32", k_div_fmas}, {"v_div_fixup_f32", k_div_fixup},
        {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64},
        {"v_add_f64", k_add_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr}, {"v_cmp_lt_f32 (sgpr dst)", k_cmp_sgpr}, {"v_cmp + v_cndmask pairs", k_cmp_cnd_pair},
        {"v_min3_f32", k_min3_f32}, {"v_med3_f32", k_med3_f32}, {"v_min_f32", k_min_f32}, {"v_xor_b32", k_xor_b32}, {"v_or_b32", k_or_b32},
        {"v_sub_f32", k_sub_f32}, {"v_fmac_f32", k_fmac_f32}, {"v_mul_f32 |abs|", k_mul_f32_abs}, {"v_mul_f32 sgpr", k_mul_f32_sgpr},
        {"v_mul_f32 literal", k_mul_f32_lit}, {"v_fma_f32 sgpr", k_fma_f32_sgpr}, {"v_alignbit_b32", k_alignbit}, {"v_lshrrev_b32", k_lshrrev_b32},
        {"v_lshrrev_b64", k_lshrrev_b64}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_add3_u32", k_add3_u32}, {"v_lshl_add_u32", k_lshl_add_u32},
        {"v_add_co_u32", k_add_co_u32}, {"v_mov_b64", k_mov_b64}, {"v_ldexp_f32", k_ldexp_f32}, {"v_cmp_class_f32", k_cmp_class},
    };
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)cus * 8 * 4 * 64 * 4 * 2);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs, clock %d kHz; wall-clock cycles per wave-instruction per SIMD at the rated clock, by resident waves per SIMD\n", cus, prop.clockRate);
    const float seed = (getenv("HALF_WAVE") ? -1.0f : 1.0f);
    if (seed < 0) printf("upper 32 lanes of every wave inactive\n");
    printf("%-24s %8s %8s %8s %8s\n", "instruction", "1", "2", "4", "8");
    const int insts = 64 * ITERS;
    for (auto& p : probes) {
        printf("%-24s", p.name);
        for (int w : {1, 2, 4, 8}) {
            // one workgroup of 4*w waves per CU puts w waves on every SIMD
            const int block = 64 * 4 * (w > 2 ? 2 : w), grid = cus * (w > 2 ? w / 2 : 1);
            hipLaunchKernelGGL(p.k, dim3(grid), dim3(block), 0, 0, out, seed, cyc);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(p.k, dim3(grid), dim3(block), 0, 0, out, seed, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double cycles = ms * 1e-3 * prop.clockRate * 1e3;
            printf(" %8.2f", cycles / ((double)insts * w));
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
