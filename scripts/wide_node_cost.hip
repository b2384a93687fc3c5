// wide_node_cost.hip — what would a visit of an 8-wide node cost?  (round-2 verdict item 5b; tooling: compiled to ISA and
// counted by scripts/wide_node_cost.py, never linked into the product and never run)
//
// The 8-wide node of the experiment is the compressed wide node of Ylitie, Karras and Laine (2017) cut to what this
// traversal needs: 96 bytes = origin.xyz (f32) + three power-of-two scales (bytes) + for each of eight children six 8-bit
// planes + eight child references.  A plane is origin + q * 2^e, so a distance is (origin - o) * invd + q * (2^e * invd):
// per node three subtractions, six multiplications and three ldexp's, then per plane one byte -> float conversion and one
// fma - against the 4-wide node's one v_fma_mix_f32 + half a v_pk_mul_f32 per plane (pt_device.hpp intersect_node4).
// Two ways to take the children that were hit:
//   unit_node8_sorted   entry distances sorted by a 19-comparator network, the seven farther ones pushed (what the 4-wide
//                       node does with five comparators and three pushes)
//   unit_node8_mask     no distances: a hit mask, ONE stack entry (node index | mask) per visit, children taken in slot
//                       order by the popper (the compressed-wide-BVH scheme; the builder would place the children so that
//                       slot order ^ ray octant is front to back) - plus unit_node8_pop, what taking one child costs then
//   unit_node4          the product's own visit, through the same harness, as the yardstick
#include "../prosper_amd/csrc/pt_device.hpp"

using namespace ppt;

namespace
{
__device__ __forceinline__ uint32_t tid() { return blockIdx.x * blockDim.x + threadIdx.x; }
struct RayIn
{
    float4 a, b; // origin.xyz tMin | 1 / direction.xyz tMax
};
struct Node8
{
    uint4 q0;     // origin.xyz, scales (ex | ey << 8 | ez << 16, biased by 127)
    uint4 q1;     // lo.x[0..7] (two dwords), lo.y[0..7]
    uint4 q2;     // lo.z[0..7], hi.x[0..7]
    uint4 q3;     // hi.y[0..7], hi.z[0..7]
    uint4 q4, q5; // child[0..7]
};
__device__ __forceinline__ float ubyte(uint32_t w, int k)
{
    return (float)((w >> (8 * k)) & 255u); // v_cvt_f32_ubyte0 .. 3
}
struct Slabs8
{
    float tn[8], tf[8];
};
// the 48 plane distances of a node -> per child the entry and exit distance
__device__ __forceinline__ void node8_distances(const Node8 &n, f3 o, f3 invd, float tMin, float tMaxK, float entry[8], bool in[8])
{
    const float bx = (__builtin_bit_cast(float, n.q0.x) - o.x) * invd.x;
    const float by = (__builtin_bit_cast(float, n.q0.y) - o.y) * invd.y;
    const float bz = (__builtin_bit_cast(float, n.q0.z) - o.z) * invd.z;
    const float sx = __builtin_ldexpf(invd.x, (int)(n.q0.w & 255u) - 127);
    const float sy = __builtin_ldexpf(invd.y, (int)((n.q0.w >> 8) & 255u) - 127);
    const float sz = __builtin_ldexpf(invd.z, (int)((n.q0.w >> 16) & 255u) - 127);
    const float bxK = bx * kSlabTol, byK = by * kSlabTol, bzK = bz * kSlabTol;
    const float sxK = sx * kSlabTol, syK = sy * kSlabTol, szK = sz * kSlabTol;
    const bool nx = invd.x < 0.0f, ny = invd.y < 0.0f, nz = invd.z < 0.0f;
    // near planes: lo when the ray travels in +, hi otherwise (two dwords of eight bytes each)
    const uint32_t nxa = nx ? n.q2.z : n.q1.x, nxb = nx ? n.q2.w : n.q1.y, fxa = nx ? n.q1.x : n.q2.z, fxb = nx ? n.q1.y : n.q2.w;
    const uint32_t nya = ny ? n.q3.x : n.q1.z, nyb = ny ? n.q3.y : n.q1.w, fya = ny ? n.q1.z : n.q3.x, fyb = ny ? n.q1.w : n.q3.y;
    const uint32_t nza = nz ? n.q3.z : n.q2.x, nzb = nz ? n.q3.w : n.q2.y, fza = nz ? n.q2.x : n.q3.z, fzb = nz ? n.q2.y : n.q3.w;
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        const int k = c & 3;
        const float tnx = __builtin_fmaf(ubyte(c < 4 ? nxa : nxb, k), sx, bx), tfx = __builtin_fmaf(ubyte(c < 4 ? fxa : fxb, k), sxK, bxK);
        const float tny = __builtin_fmaf(ubyte(c < 4 ? nya : nyb, k), sy, by), tfy = __builtin_fmaf(ubyte(c < 4 ? fya : fyb, k), syK, byK);
        const float tnz = __builtin_fmaf(ubyte(c < 4 ? nza : nzb, k), sz, bz), tfz = __builtin_fmaf(ubyte(c < 4 ? fza : fzb, k), szK, bzK);
        const float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);
        in[c] = fmaxf(tn, tMin) <= fminf(tf, tMaxK);
        entry[c] = in[c] ? tn : kInf;
    }
}
} // namespace

extern "C" __global__ void unit_baseline_ray(const RayIn *in, float4 *out)
{
    const RayIn r = in[tid()];
    out[tid()] = make_float4(r.a.x + r.b.x, r.a.y + r.b.y, r.a.z + r.b.z, r.a.w + r.b.w);
}

extern "C" __global__ void unit_node4(DeviceScene s, const RayIn *in, const int32_t *nodeIdx, float4 *out, int32_t *stackMem)
{
    __shared__ int32_t lds[16 * 256];
    const RayIn r = in[tid()];
    const TraversalStack stack{(lds_int32 *)lds + (threadIdx.x >> 6) * (16u * 64u) + (threadIdx.x & 63u), stackMem + tid(), 16u, gridDim.x * 256u, 64u};
    const f3 o = f3{r.a.x, r.a.y, r.a.z}, invd = f3{r.b.x, r.b.y, r.b.z};
    const GlobalGeom g{s.nodes, s.triangles};
    const NodeData nd = g.node(nodeIdx[tid()]);
    float e[4];
    int32_t ref[4];
    int32_t sp = 0;
    intersect_node4(nd, o, make_ray_slabs(invd), r.a.w, r.b.w, e, ref);
    if (e[3] < kInf) stack.push(sp, ref[3]);
    if (e[2] < kInf) stack.push(sp, ref[2]);
    if (e[1] < kInf) stack.push(sp, ref[1]);
    out[tid()] = make_float4(e[0], (float)ref[0], (float)sp, 0.0f);
}

extern "C" __global__ void unit_node8_sorted(const Node8 *nodes, const RayIn *in, const int32_t *nodeIdx, float4 *out, int32_t *stackMem)
{
    __shared__ int32_t lds[16 * 256];
    const RayIn r = in[tid()];
    const TraversalStack stack{(lds_int32 *)lds + (threadIdx.x >> 6) * (16u * 64u) + (threadIdx.x & 63u), stackMem + tid(), 16u, gridDim.x * 256u, 64u};
    const f3 o = f3{r.a.x, r.a.y, r.a.z}, invd = f3{r.b.x, r.b.y, r.b.z};
    const Node8 n = nodes[nodeIdx[tid()]];
    float e[8];
    bool in8[8];
    node8_distances(n, o, invd, r.a.w, r.b.w * kSlabTol, e, in8);
    int32_t ref[8] = {(int32_t)n.q4.x, (int32_t)n.q4.y, (int32_t)n.q4.z, (int32_t)n.q4.w, (int32_t)n.q5.x, (int32_t)n.q5.y, (int32_t)n.q5.z, (int32_t)n.q5.w};
#define CS(i, j)                                                                                                       \
    {                                                                                                                  \
        const bool sw = e[j] < e[i];                                                                                   \
        const float te = sw ? e[j] : e[i];                                                                             \
        e[j] = sw ? e[i] : e[j];                                                                                       \
        e[i] = te;                                                                                                     \
        const int32_t tr = sw ? ref[j] : ref[i];                                                                       \
        ref[j] = sw ? ref[i] : ref[j];                                                                                 \
        ref[i] = tr;                                                                                                   \
    }
    // 19-comparator network for eight keys
    CS(0, 1) CS(2, 3) CS(4, 5) CS(6, 7) CS(0, 2) CS(1, 3) CS(4, 6) CS(5, 7) CS(1, 2) CS(5, 6) CS(0, 4) CS(3, 7) CS(1, 5) CS(2, 6) CS(1, 4) CS(3, 6) CS(2, 4) CS(3, 5) CS(3, 4)
#undef CS
    int32_t sp = 0;
#pragma unroll
    for (int c = 7; c >= 1; --c)
        if (e[c] < kInf) stack.push(sp, ref[c]);
    out[tid()] = make_float4(e[0], (float)ref[0], (float)sp, 0.0f);
}

extern "C" __global__ void unit_node8_mask(const Node8 *nodes, const RayIn *in, const int32_t *nodeIdx, float4 *out, int32_t *stackMem)
{
    __shared__ int32_t lds[16 * 256];
    const RayIn r = in[tid()];
    const TraversalStack stack{(lds_int32 *)lds + (threadIdx.x >> 6) * (16u * 64u) + (threadIdx.x & 63u), stackMem + tid(), 16u, gridDim.x * 256u, 64u};
    const f3 o = f3{r.a.x, r.a.y, r.a.z}, invd = f3{r.b.x, r.b.y, r.b.z};
    const int32_t self = nodeIdx[tid()];
    const Node8 n = nodes[self];
    float e[8];
    bool in8[8];
    node8_distances(n, o, invd, r.a.w, r.b.w * kSlabTol, e, in8);
    uint32_t mask = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) mask |= in8[c] ? 1u << c : 0u;
    int32_t sp = 0;
    // one entry stands for every child that was hit: (node, mask); the first child is taken right away
    int32_t next = -1;
    if (mask)
    {
        const uint32_t first = (uint32_t)__builtin_ctz(mask);
        mask &= mask - 1u;
        const uint32_t refs[8] = {n.q4.x, n.q4.y, n.q4.z, n.q4.w, n.q5.x, n.q5.y, n.q5.z, n.q5.w};
        uint32_t pick = refs[0];
#pragma unroll
        for (int c = 1; c < 8; ++c) pick = first == (uint32_t)c ? refs[c] : pick;
        next = (int32_t)pick;
        if (mask) stack.push(sp, (int32_t)(((uint32_t)self << 8) | mask));
    }
    out[tid()] = make_float4((float)next, (float)sp, 0.0f, 0.0f);
}

// taking the next child of a (node, mask) entry off the stack: the child reference has to be fetched again
extern "C" __global__ void unit_node8_pop(const Node8 *nodes, const RayIn *in, const int32_t *entries, float4 *out, int32_t *stackMem)
{
    __shared__ int32_t lds[16 * 256];
    const RayIn r = in[tid()];
    const TraversalStack stack{(lds_int32 *)lds + (threadIdx.x >> 6) * (16u * 64u) + (threadIdx.x & 63u), stackMem + tid(), 16u, gridDim.x * 256u, 64u};
    int32_t sp = 1;
    stack.lds[0] = entries[tid()];
    const uint32_t entry = (uint32_t)stack.pop(sp);
    uint32_t mask = entry & 255u;
    const uint32_t node = entry >> 8;
    const uint32_t c = (uint32_t)__builtin_ctz(mask);
    mask &= mask - 1u;
    if (mask) stack.push(sp, (int32_t)((node << 8) | mask));
    const int32_t child = reinterpret_cast<const int32_t *>(&nodes[node].q4)[c];
    out[tid()] = make_float4((float)child + r.a.x, (float)sp, 0.0f, 0.0f);
}
