// gather_microbench.hip — how fast does a CU fetch random 80-byte BVH nodes? (measurement tooling, gfx950)
//
//   hipcc -O3 --offload-arch=gfx950 scripts/gather_microbench.hip -o scripts/gather_microbench && scripts/gather_microbench
//
// Every lane walks its own pseudo-random sequence of node indices over a table of `nodes` 80-byte records (the
// layout wf_trace reads) and fetches each record in one of these ways:
//   lane     five 16-byte loads by the lane itself (what the traversal does)
//   quad     the four lanes of a quad fetch each other's records: lane j loads bytes [16 j, 16 j + 16) of the record
//            of quad lane r, r = 0..3 (64 contiguous bytes per quad and instruction), plus one own load for bytes 64..79,
//            and the pieces travel to their owner through DPP quad permutes
//   quad64   as quad, records padded to 128 bytes (never straddle a cache line)
// Output: records per microsecond for the whole chip, for a few table sizes (L2-resident ... HBM).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                                       \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess)                                                                                           \
        {                                                                                                              \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                                     \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v)
{
    const uint32_t s = v * 747796405u + 2891336453u;
    const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

// MODE 0: lane, 1: quad (pieces are only folded into a checksum: the memory side of the idea), 2: quad through LDS
// (global_load_lds_dwordx4 drops each quad's 64 bytes into LDS, the owner reads its record back: the whole idea)
template <int MODE, int STRIDE16>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ table, uint32_t nodes, uint32_t iters, uint32_t *out)
{
    __shared__ uint4 stage[MODE == 2 ? 4 * 4 * 65 : 1]; // per wave: 4 rounds x (64 lanes + 1 pad) x 16 B
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t j = threadIdx.x & 3u;
    uint32_t acc = 0;
    uint32_t state = pcg(tid);
    for (uint32_t it = 0; it < iters; ++it)
    {
        state = pcg(state + it);
        const uint32_t idx = state % nodes;
        if (MODE == 0)
        {
            const uint4 *p = table + (size_t)idx * STRIDE16;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = STRIDE16 == 4 ? p[0] : p[4];
            acc ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ a.w ^ b.z ^ c.y ^ d.x ^ e.w;
        }
        else
        {
            // indices of the quad's four lanes
            const uint32_t i0 = quad_perm<0x00>(idx), i1 = quad_perm<0x55>(idx), i2 = quad_perm<0xAA>(idx),
                           i3 = quad_perm<0xFF>(idx);
            const uint4 *p0 = table + (size_t)i0 * STRIDE16 + j, *p1 = table + (size_t)i1 * STRIDE16 + j,
                        *p2 = table + (size_t)i2 * STRIDE16 + j, *p3 = table + (size_t)i3 * STRIDE16 + j;
            const uint4 e = table[(size_t)idx * STRIDE16 + 4];
            if (MODE == 1)
            {
                const uint4 r0 = *p0, r1 = *p1, r2 = *p2, r3 = *p3;
                acc ^= r0.x ^ r1.y ^ r2.z ^ r3.w ^ e.x ^ r0.w ^ r1.z ^ r2.y ^ r3.x ^ e.w;
            }
            else
            {
                typedef __attribute__((address_space(3))) void lds_void;
                typedef __attribute__((address_space(1))) const void global_void;
                uint4 *base = stage + wave * (4 * 65);
                // round r lands at base + r * 65 + lane (the instruction adds lane * 16 bytes itself)
                __builtin_amdgcn_global_load_lds((global_void *)p0, (lds_void *)(base + 0 * 65), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((global_void *)p1, (lds_void *)(base + 1 * 65), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((global_void *)p2, (lds_void *)(base + 2 * 65), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((global_void *)p3, (lds_void *)(base + 3 * 65), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // lane t = 4 q + j owns the record of round j: its pieces are the quad's four slots of that round
                const uint4 *mine = base + j * 65 + (lane & ~3u);
                const uint4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
                acc ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ a.w ^ b.z ^ c.y ^ d.x ^ e.w;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the next iteration overwrites the staging area
            }
        }
    }
    out[tid] = acc;
}

int main(int argc, char **argv)
{
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 512u;
    const uint32_t blocks = 256u * 5u * 4u; // 5 waves per SIMD (the traversal kernels' occupancy), 4 rounds
    uint32_t *out;
    CHECK(hipMalloc(&out, blocks * 256u * 4u));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t sizes[] = {8u, 4096u, 32768u, 65536u, 131072u, 1u << 20, 1u << 23};
    for (uint32_t nodes : sizes)
    {
        uint4 *table;
        const size_t bytes = (size_t)nodes * 128u + 256u;
        CHECK(hipMalloc(&table, bytes));
        CHECK(hipMemset(table, 0x5A, bytes));
        auto run = [&](const char *name, auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, table, nodes, 16u, out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, table, nodes, iters, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double recs = (double)blocks * 256.0 * iters;
            printf("nodes %8u (%7.1f MB at 80 B)  %-8s %8.1f records/us  %6.2f ms  %.2f TB/s of records\n", nodes,
                   nodes * 80.0 / 1e6, name, recs / (ms * 1e3), ms, recs * 80.0 / (ms * 1e-3) / 1e12);
            fflush(stdout);
        };
        run("lane64", gather<0, 4>); // 64-byte records, 64-byte aligned: half a cache line each
        run("lane", gather<0, 5>);
        run("quad", gather<1, 5>);
        run("quadlds", gather<2, 5>);
        run("lane128", gather<0, 8>);
        run("quad128", gather<1, 8>);
        run("quadlds128", gather<2, 8>);
        CHECK(hipFree(table));
    }
    return 0;
}
