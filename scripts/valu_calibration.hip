// valu_calibration.hip — what one gfx950 SIMD issues per cycle, measured (VERDICT r01 "Next 2").
//
// DESIGN.md (round 1) priced the traversal kernels against "a wave64 VALU instruction holds a SIMD for 4
// cycles"; MI355X_MICROARCH.md says 2 (SIMD-32).  This program settles it per instruction KIND: every
// workgroup runs a loop of INDEPENDENT vector instructions of one kind (inline asm, 16 accumulators,
// nothing for the compiler to fold), at 1 / 2 / 5 waves per SIMD, timed with hipEvents; in-kernel
// s_memtime / s_memrealtime gives the shader clock the loop really ran at.  The "a+b" rows interleave two
// kinds 1:1 to see whether their costs add or overlap.  Rows whose asm writes VCC or an SGPR declare that clobber,
// and the compiler then separates the statements by an s_nop (hazard recogniser): their figures include it.  SALU rows
// declare SCC (the loop's own compare lives there).
//
//   hipcc -O2 --offload-arch=gfx950 scripts/valu_calibration.hip -o scripts/valu_calibration
//   scripts/valu_calibration > profiles/r02_valu_calibration.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                     \
    do                                                                                               \
    {                                                                                                \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess)                                                                        \
        {                                                                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                  \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

constexpr int kAcc = 16;          // independent accumulators
constexpr int kInstsPerIter = 64; // 4 rounds over the accumulators

typedef float v2f __attribute__((ext_vector_type(2)));

// One row of the table: name, wave-instructions per slot, the asm of one slot on accumulator i.
// a = float acc, u = uint acc, p = packed pair acc; x, y floats; ux uint; xx, yy packed; m = 64-bit SGPR mask
#define KINDS(X)                                                                                                       \
    X(fma_f32, 1, "v_fma_f32 %0, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y))                                                 \
    X(mul_f32, 1, "v_mul_f32 %0, %0, %1", "+v"(a[i]), "v"(x), "v"(y))                                                     \
    X(add_f32, 1, "v_add_f32 %0, %0, %1", "+v"(a[i]), "v"(x), "v"(y))                                                     \
    X(sub_f32, 1, "v_sub_f32 %0, %0, %1", "+v"(a[i]), "v"(x), "v"(y))                                                     \
    X(fmac_f32, 1, "v_fmac_f32 %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y))                                                   \
    X(min_f32, 1, "v_min_f32 %0, %0, %1", "+v"(a[i]), "v"(x), "v"(y))                                                     \
    X(max3_f32, 1, "v_max3_f32 %0, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y))                                               \
    X(med3_f32, 1, "v_med3_f32 %0, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y))                                               \
    X(fma_mix_f32, 1, "v_fma_mix_f32 %0, %1, 1.0, -%0 op_sel_hi:[1,0,0]", "+v"(a[i]), "v"(ux), "v"(y))                    \
    X(cvt_f32_f16, 1, "v_cvt_f32_f16 %0, %0", "+v"(a[i]), "v"(x), "v"(y))                                                 \
    X(pk_fma_f32, 1, "v_pk_fma_f32 %0, %0, %1, %2", "+v"(p[i]), "v"(xx), "v"(yy))                                         \
    X(pk_mul_f32, 1, "v_pk_mul_f32 %0, %0, %1", "+v"(p[i]), "v"(xx), "v"(yy))                                             \
    X(pk_add_f32, 1, "v_pk_add_f32 %0, %0, %1", "+v"(p[i]), "v"(xx), "v"(yy))                                             \
    X(mov_b32, 1, "v_mov_b32 %0, %1", "+v"(a[i]), "v"(x), "v"(y))                                                         \
    X(add_u32, 1, "v_add_u32 %0, %0, %1", "+v"(u[i]), "v"(ux), "v"(ux))                                                   \
    X(and_b32, 1, "v_and_b32 %0, %0, %1", "+v"(u[i]), "v"(ux), "v"(ux))                                                   \
    X(xor_b32, 1, "v_xor_b32 %0, %0, %1", "+v"(u[i]), "v"(ux), "v"(ux))                                                   \
    X(lshlrev_b32, 1, "v_lshlrev_b32 %0, 1, %0", "+v"(u[i]), "v"(ux), "v"(ux))                                            \
    X(lshl_add_u32, 1, "v_lshl_add_u32 %0, %0, 1, %1", "+v"(u[i]), "v"(ux), "v"(ux))                                      \
    X(bfi_b32, 1, "v_bfi_b32 %0, %1, %0, %2", "+v"(u[i]), "v"(ux), "v"(ux))                                               \
    X(and_or_b32, 1, "v_and_or_b32 %0, %0, %1, %2", "+v"(u[i]), "v"(ux), "v"(ux))                                         \
    X(perm_b32, 1, "v_perm_b32 %0, %0, %1, %2", "+v"(u[i]), "v"(ux), "v"(ux))                                             \
    X(mul_lo_u32, 1, "v_mul_lo_u32 %0, %0, %1", "+v"(u[i]), "v"(ux), "v"(ux))                                             \
    X(mul_u32_u24, 1, "v_mul_u32_u24 %0, %0, %1", "+v"(u[i]), "v"(ux), "v"(ux))                                           \
    X(mad_u32_u24, 1, "v_mad_u32_u24 %0, %0, %1, %2", "+v"(u[i]), "v"(ux), "v"(ux))                                       \
    X(cndmask_vcc, 1, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(u[i]), "v"(ux), "v"(ux))                                      \
    X(cndmask_sgpr, 1, "v_cndmask_b32_e64 %0, %0, %1, %2", "+v"(u[i]), "v"(ux), "s"(m))                                   \
    X(cmp_lt_f32_vcc, 1, "v_cmp_lt_f32 vcc, %0, %1", "+v"(a[i]), "v"(x), "v"(y), "vcc")                                          \
    X(cmp_lt_f32_sgpr, 1, "v_cmp_lt_f32_e64 s[20:21], %0, %1", "+v"(a[i]), "v"(x), "v"(y), "s20", "s21")                                \
    X(cmp_then_cndmask, 2, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc", "+v"(a[i]), "v"(x), "v"(y), "vcc")        \
    X(cmp_then_4cndmask, 5, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(a[i]), "v"(x), "v"(y), "vcc") \
    X(cmp_sgpr_then_4cndmask, 5, "v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "+v"(a[i]), "v"(x), "v"(y), "s20", "s21") \
    X(cndmask_vcc_set_once, 1, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(u[i]), "v"(ux), "v"(ux))                                 \
    X(cndmask_e64_vcc, 1, "v_cndmask_b32_e64 %0, %0, %1, vcc", "+v"(u[i]), "v"(ux), "v"(ux))                                  \
    X(cndmask_vcc_nop_between, 1, "v_cndmask_b32 %0, %0, %1, vcc\n s_nop 0", "+v"(u[i]), "v"(ux), "v"(ux))                  \
    X(cndmask_vcc_nop4_between, 1, "v_cndmask_b32 %0, %0, %1, vcc\n s_nop 3", "+v"(u[i]), "v"(ux), "v"(ux))                 \
    X(cndmask_vcc_2_then_fma, 3, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_fma_f32 %0, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y)) \
    X(cndmask_alternating_sgprs, 2, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, s[22:23]", "+v"(a[i]), "v"(x), "v"(y)) \
    X(cmp_e64_vcc_then_4cndmask_e64, 5, "v_cmp_lt_f32_e64 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %0, %0, %1, vcc", "+v"(a[i]), "v"(x), "v"(y), "vcc") \
    X(addc_vcc_chain, 1, "v_addc_co_u32 %0, vcc, %0, %1, vcc", "+v"(u[i]), "v"(ux), "v"(ux), "vcc")                          \
    X(min_u32, 1, "v_min_u32 %0, %0, %1", "+v"(u[i]), "v"(ux), "v"(ux))                                                     \
    X(max_f32, 1, "v_max_f32 %0, %0, %1", "+v"(a[i]), "v"(x), "v"(y))                                                       \
    X(min3_f32, 1, "v_min3_f32 %0, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y))                                                 \
    X(mov_dpp_quad, 1, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "+v"(a[i]), "v"(x), "v"(y))   \
    X(ds_swizzle, 1, "ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)", "+v"(a[i]), "v"(x), "v"(y))    \
    X(rcp_f32, 1, "v_rcp_f32 %0, %0", "+v"(a[i]), "v"(x), "v"(y))                                                         \
    X(sqrt_f32, 1, "v_sqrt_f32 %0, %0", "+v"(a[i]), "v"(x), "v"(y))                                                       \
    X(div_scale_f32, 1, "v_div_scale_f32 %0, vcc, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y), "vcc")                                \
    X(div_fmas_f32, 1, "v_div_fmas_f32 %0, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y))                                       \
    X(div_fixup_f32, 1, "v_div_fixup_f32 %0, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y))                                     \
    X(cvt_f32_u32, 1, "v_cvt_f32_u32 %0, %0", "+v"(u[i]), "v"(ux), "v"(ux))                                               \
    X(readfirstlane, 1, "v_readfirstlane_b32 s22, %0", "+v"(u[i]), "v"(ux), "v"(ux), "s22")                                      \
    X(fma_dependent, 1, "v_fma_f32 %0, %0, %1, %2", "+v"(a[0]), "v"(x), "v"(y))                                           \
    X(fma_plus_cndmask, 2, "v_fma_f32 %0, %0, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(a[i]), "v"(x), "v"(y))        \
    X(fma_plus_max3, 2, "v_fma_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %1, %2", "+v"(a[i]), "v"(x), "v"(y))               \
    X(fma_plus_fma_mix, 2, "v_fma_f32 %0, %0, %1, %2\n v_fma_mix_f32 %0, %0, 1.0, -%1 op_sel_hi:[1,0,0]", "+v"(a[i]), "v"(x), "v"(y)) \
    X(fma_plus_mov, 2, "v_fma_f32 %0, %0, %1, %2\n v_mov_b32 %0, %0", "+v"(a[i]), "v"(x), "v"(y))                         \
    X(salu_s_or_b64, 1, "s_or_b64 s[20:21], s[20:21], s[22:23]", "+v"(a[i]), "v"(x), "v"(y), "s20", "s21", "scc")                              \
    X(fma_plus_salu, 2, "v_fma_f32 %0, %0, %1, %2\n s_or_b64 s[20:21], s[20:21], s[22:23]", "+v"(a[i]), "v"(x), "v"(y), "s20", "s21", "scc")   \
    X(saveexec_pair, 2, "s_and_saveexec_b64 s[20:21], -1\n s_or_b64 exec, exec, s[20:21]", "+v"(a[i]), "v"(x), "v"(y), "s20", "s21", "scc")

enum Kind
{
#define X(name, n, text, o, i0, i1, ...) k_##name,
    KINDS(X)
#undef X
        kKinds
};
static const char *kNames[kKinds] = {
#define X(name, n, text, o, i0, i1, ...) #name,
    KINDS(X)
#undef X
};
// wave-instructions one "slot" of the loop issues (the two-instruction rows count 2)
static const int kPerSlot[kKinds] = {
#define X(name, n, text, o, i0, i1, ...) n,
    KINDS(X)
#undef X
};

template <int KIND>
__global__ __launch_bounds__(256) void valu_loop(float *out, unsigned long long *clocks, int iters, float x, float y)
{
    float a[kAcc];
    v2f p[kAcc];
    unsigned int u[kAcc];
#pragma unroll
    for (int i = 0; i < kAcc; ++i)
    {
        a[i] = x * (float)(threadIdx.x + i);
        p[i] = v2f{a[i], y * (float)(threadIdx.x + 2 * i)};
        u[i] = threadIdx.x * 3u + i;
    }
    const v2f xx = v2f{x, x}, yy = v2f{y, y};
    const unsigned int ux = __builtin_bit_cast(unsigned int, x);
    const unsigned long long m = __ballot(threadIdx.x & 1);
    asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 s[22:23], 0" ::: "s20", "s21", "s22", "s23");
    if constexpr (KIND == k_cndmask_vcc_set_once || KIND == k_cndmask_e64_vcc || KIND == k_cndmask_vcc_nop_between || KIND == k_cndmask_vcc_nop4_between || KIND == k_cndmask_vcc_2_then_fma) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(x), "v"(y) : "vcc");

    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int r = 0; r < kInstsPerIter / kAcc; ++r)
        {
#pragma unroll
            for (int i = 0; i < kAcc; ++i)
            {
#define X(name, n, text, o, i0, i1, ...)                                                                                    \
    if constexpr (KIND == k_##name) asm volatile(text : o : i0, i1 : __VA_ARGS__);
                KINDS(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    unsigned int us = 0;
#pragma unroll
    for (int i = 0; i < kAcc; ++i)
    {
        s += a[i] + p[i].x + p[i].y;
        us += u[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)us;
    if (threadIdx.x == 0)
    {
        clocks[2 * blockIdx.x] = t1 - t0;
        clocks[2 * blockIdx.x + 1] = r1 - r0;
    }
}

typedef void (*KernelFn)(float *, unsigned long long *, int, float, float);
static KernelFn kKernels[kKinds] = {
#define X(name, n, text, o, i0, i1, ...) valu_loop<k_##name>,
    KINDS(X)
#undef X
};

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 10000;
    const char *only = argc > 2 ? argv[2] : nullptr; // run only the kinds whose name contains this
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int maxBlocksPerCu = 8;
    float *out = nullptr;
    unsigned long long *clocks = nullptr;
    CHECK(hipMalloc((void **)&out, (size_t)cus * maxBlocksPerCu * 256 * sizeof(float)));
    CHECK(hipMalloc((void **)&clocks, (size_t)cus * maxBlocksPerCu * 2 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> h((size_t)cus * maxBlocksPerCu * 2);

    printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"clockRate_kHz\": %d, \"iters\": %d,\n", prop.gcnArchName, cus,
           cus * 4, prop.clockRate, iters);
    printf(" \"note\": \"256-thread workgroups (one wave per SIMD each), k workgroups per CU = k waves per SIMD; every wave "
           "issues iters*64 slots of one kind (two-instruction rows: 2 wave-instructions per slot), 16 independent "
           "accumulators; ns_per_inst_per_simd = hipEvent time of the launch / (wave-instructions per wave * k): what one "
           "SIMD spends per wave64 instruction when k waves share it; cycles = that x ghz, ghz = median over workgroups of "
           "s_memtime / s_memrealtime * 0.1 inside the loop\",\n \"results\": [\n");
    bool first = true;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kKernels[0], dim3(cus * 4), dim3(256), 0, 0, out, clocks, iters, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    for (int kind = 0; kind < kKinds; ++kind)
    {
        if (only && !strstr(kNames[kind], only)) continue;
        const int wavesList[] = {1, 2, 5};
        for (int wi = 0; wi < 3; ++wi)
        {
            const int k = wavesList[wi];
            const int blocks = cus * k;
            hipLaunchKernelGGL(kKernels[kind], dim3(blocks), dim3(256), 0, 0, out, clocks, iters / 4, 1.0001f, 0.5f);
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kKernels[kind], dim3(blocks), dim3(256), 0, 0, out, clocks, iters, 1.0001f, 0.5f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), clocks, (size_t)blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<double> ghzs(blocks);
            for (int b = 0; b < blocks; ++b) ghzs[b] = h[2 * b + 1] ? (double)h[2 * b] / (double)h[2 * b + 1] * 0.1 : 0.0;
            std::nth_element(ghzs.begin(), ghzs.begin() + blocks / 2, ghzs.end());
            const double ghz = ghzs[blocks / 2];
            const double insts = (double)iters * kInstsPerIter * kPerSlot[kind];
            const double ns = ms * 1e6 / (insts * k);
            printf("%s  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ghz\": %.3f, \"ns_per_inst_per_simd\": %.4f, "
                   "\"cycles_per_inst_per_simd\": %.3f, \"wave_insts_per_s_chip\": %.4e}",
                   first ? "" : ",\n", kNames[kind], k, ms, ghz, ns, ns * ghz, insts * k * cus * 4 / (ms * 1e-3));
            first = false;
            fflush(stdout);
        }
    }
    printf("\n ]}\n");
    return 0;
}
