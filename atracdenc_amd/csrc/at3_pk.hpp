// Packed fp32 helpers of the gfx950 kernels (inline VOP3P assembly; see at3_common.hpp for how they are used).
#ifndef AT3_PK_HPP
#define AT3_PK_HPP
#include <hip/hip_runtime.h>

namespace at3 {

// v_pk_mul_f32 / v_pk_add_f32 (VOP3P) carry TWO independent IEEE fp32 operations per lane. The reference arithmetic has
// no fused multiply-add (-ffp-contract=off is part of the parity contract), so mul/add-bound code is issue bound; a packed
// instruction does NOT issue at the rate of a plain one, though: measured (tools/ubench/valu_lds_rates, profiles/
// r04_ubench_instruction_rates.txt, shader clock 2.1 - 2.4 GHz under load) a wavefront issues a plain fp32 instruction every
// ~6 - 8 cycles and a packed one every ~8.5 - 14, so packing buys between nothing and 40 % per flop depending on how many
// wavefronts share the SIMD - its surer gain is half the instruction count where the stream is latency bound. It never
// changes a rounding. `f2` maps to an aligned VGPR pair; plain vector expressions
// are selected as packed instructions by the compiler, the three complex forms whose lanes need DIFFERENT negate /
// half-select modifiers are spelled out below.
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 mk2(float x, float y)
{
    f2 v;
    v.x = x;
    v.y = y;
    return v;
}

// a * w, complex: (a.x w.x - a.y w.y, a.x w.y + a.y w.x) - the four products and two sums of C_MUL (_kiss_fft_guts.h)
__device__ __forceinline__ f2 pk_cmul(f2 a, f2 w)
{
    f2 t1, t2, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t1) : "v"(a), "v"(w));                  // (a.x w.x, a.x w.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t2) : "v"(a), "v"(w));     // (a.y w.y, a.y w.x)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(t1), "v"(t2));                    // (t1.x - t2.x, t1.y + t2.y)
    return r;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 pk_add_ib(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2 pk_sub_ib(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// r + the number of the eight values c0.x .. c1.w that are below `key`: a compare into a scalar pair and an add-with-carry
// of zero per value, four compares ahead of their adds (a vector instruction may read a scalar pair two instructions after
// the vector instruction that wrote it; spelled out here because the compiler, short of scalar registers in k_alloc_pack,
// paired every compare with a select and a wait state)
__device__ __forceinline__ int count_below8(float4 c0, float4 c1, float key, int r)
{
    unsigned long long m0, m1, m2, m3;
    asm("v_cmp_lt_f32_e64 %1, %5, %13\n\t"
        "v_cmp_lt_f32_e64 %2, %6, %13\n\t"
        "v_cmp_lt_f32_e64 %3, %7, %13\n\t"
        "v_cmp_lt_f32_e64 %4, %8, %13\n\t"
        "v_addc_co_u32_e64 %0, %1, 0, %0, %1\n\t"
        "v_addc_co_u32_e64 %0, %2, 0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, %3, 0, %0, %3\n\t"
        "v_addc_co_u32_e64 %0, %4, 0, %0, %4\n\t"
        "v_cmp_lt_f32_e64 %1, %9, %13\n\t"
        "v_cmp_lt_f32_e64 %2, %10, %13\n\t"
        "v_cmp_lt_f32_e64 %3, %11, %13\n\t"
        "v_cmp_lt_f32_e64 %4, %12, %13\n\t"
        "v_addc_co_u32_e64 %0, %1, 0, %0, %1\n\t"
        "v_addc_co_u32_e64 %0, %2, 0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, %3, 0, %0, %3\n\t"
        "v_addc_co_u32_e64 %0, %4, 0, %0, %4"
        : "+v"(r), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(c0.x), "v"(c0.y), "v"(c0.z), "v"(c0.w), "v"(c1.x), "v"(c1.y), "v"(c1.z), "v"(c1.w), "v"(key));
    return r;
}

// A copy of a per-lane value the optimiser cannot see through: address arithmetic derived from it is redone where it is
// used instead of being hoisted in front of a long loop and parked in registers for the whole kernel.
__device__ __forceinline__ int opaque_lane_value(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

}  // namespace at3
// occupancy hint for a kernel: register allocation for exactly n wavefronts per SIMD
#define AT3_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif  // AT3_PK_HPP
