// unit_costs.hip — one kernel per unit of path-tracing work, for a STATIC instruction count (scripts/unit_costs.py):
// each kernel reads its inputs from memory, runs exactly one unit through the product's own device functions
// (prosper_amd/csrc/pt_device.hpp, compiled with the product's flags) and stores every output, so that nothing is folded
// away.  unit_baseline_* do the same loads and stores without the work; a unit's cost = its VALU count minus its
// baseline's.  Both sides of a branch are counted (a static count cannot know which one a lane takes): the figures are
// upper estimates of what one lane needs, which makes the efficiency derived from them optimistic, never pessimistic.
// Never linked into the product and never run.
#include "../prosper_amd/csrc/pt_device.hpp"
#include "../prosper_amd/csrc/pt_render_common.hpp"

using namespace ppt;

namespace
{
__device__ __forceinline__ uint32_t tid() { return blockIdx.x * blockDim.x + threadIdx.x; }
struct RayIn
{
    float4 a, b; // origin.xyz tMin | direction.xyz tMax
};
} // namespace

extern "C" __global__ void unit_baseline_ray(const RayIn *in, float4 *out)
{
    const RayIn r = in[tid()];
    out[tid()] = make_float4(r.a.x + r.b.x, r.a.y + r.b.y, r.a.z + r.b.z, r.a.w + r.b.w);
}

// one closest-hit node visit: node fetch, four slab tests, distance sort, pushes of the far children
extern "C" __global__ void unit_node_visit_closest(DeviceScene s, const RayIn *in, const int32_t *nodeIdx, float4 *out, int32_t *stackMem)
{
    __shared__ int32_t lds[16 * 256];
    const RayIn r = in[tid()];
    const TraversalStack stack{(lds_int32 *)lds + (threadIdx.x >> 6) * (16u * 64u) + (threadIdx.x & 63u), stackMem + tid(), 16u, gridDim.x * 256u, 64u};
    const f3 o = f3{r.a.x, r.a.y, r.a.z}, invd = f3{r.b.x, r.b.y, r.b.z}; // 1 / direction is per-ray work, not per visit
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

// one any-hit (shadow) node visit: no sort, children in storage order
extern "C" __global__ void unit_node_visit_shadow(DeviceScene s, const RayIn *in, const int32_t *nodeIdx, float4 *out, int32_t *stackMem)
{
    __shared__ int32_t lds[16 * 256];
    const RayIn r = in[tid()];
    const TraversalStack stack{(lds_int32 *)lds + (threadIdx.x >> 6) * (16u * 64u) + (threadIdx.x & 63u), stackMem + tid(), 16u, gridDim.x * 256u, 64u};
    const f3 o = f3{r.a.x, r.a.y, r.a.z}, invd = f3{r.b.x, r.b.y, r.b.z};
    const GlobalGeom g{s.nodes, s.triangles};
    const NodeData nd = g.node(nodeIdx[tid()]);
    float e[4];
    int32_t ref[4];
    int32_t sp = 0, node = 0;
    intersect_node4<false>(nd, o, make_ray_slabs(invd), r.a.w, r.b.w, e, ref);
    const bool entered = descend_any(e, ref, stack, sp, node);
    out[tid()] = make_float4(entered ? 1.0f : 0.0f, (float)node, (float)sp, 0.0f);
}

// the cheap rejection of one triangle: fetch + three edge functions
extern "C" __global__ void unit_triangle_edge_functions(DeviceScene s, const RayIn *in, const uint32_t *triIdx, float4 *out)
{
    const RayIn r = in[tid()];
    const f3 o = f3{r.a.x, r.a.y, r.a.z}, d = f3{r.b.x, r.b.y, r.b.z};
    const GlobalGeom g{s.nodes, s.triangles};
    const TriangleData td = g.tri(triIdx[tid()]);
    const EdgeFunctions e = edge_functions(o, d, f3{td.a.x, td.a.y, td.a.z}, f3{td.b.x, td.b.y, td.b.z}, f3{td.c.x, td.c.y, td.c.z});
    out[tid()] = make_float4(e.U, e.V, e.W, e.pass ? e.det : 0.0f);
}

// the same + distance, range test, box guard, barycentrics of a triangle the ray goes through
extern "C" __global__ void unit_triangle_full(DeviceScene s, const RayIn *in, const uint32_t *triIdx, float4 *out)
{
    const RayIn r = in[tid()];
    const f3 o = f3{r.a.x, r.a.y, r.a.z}, d = f3{r.b.x, r.b.y, r.b.z};
    const f3 invd = f3{safe_rcp_dir(d.x), safe_rcp_dir(d.y), safe_rcp_dir(d.z)};
    const GlobalGeom g{s.nodes, s.triangles};
    const TriangleData td = g.tri(triIdx[tid()]);
    float t = 0.0f, bu = 0.0f, bv = 0.0f;
    const bool hit = intersect_triangle(
        o, d, invd, f3{td.a.x, td.a.y, td.a.z}, f3{td.b.x, td.b.y, td.b.z}, f3{td.c.x, td.c.y, td.c.z}, r.a.w, r.b.w, t, bu, bv);
    out[tid()] = make_float4(hit ? t : -1.0f, bu, bv, 0.0f);
}

// any-hit settled by the material's alpha bounds / with the exact texel path behind it
extern "C" __global__ void unit_any_hit_settle(DeviceScene s, const RayIn *in, const uint32_t *alphaIdx, float4 *out)
{
    const RayIn r = in[tid()];
    LaneCounters cnt = {};
    AlphaFootprint fp = {};
    const uint32_t v = any_hit_settle<false>(s, alphaIdx[tid()], f2{r.a.x, r.a.y}, __builtin_bit_cast(uint32_t, r.a.z), cnt, fp);
    out[tid()] = make_float4((float)v, 0.0f, 0.0f, 0.0f);
}
extern "C" __global__ void unit_any_hit_with_texels(DeviceScene s, const RayIn *in, const uint32_t *alphaIdx, float4 *out)
{
    const RayIn r = in[tid()];
    LaneCounters cnt = {};
    const bool v = any_hit_record<false>(s, alphaIdx[tid()], f2{r.a.x, r.a.y}, __builtin_bit_cast(uint32_t, r.a.z), cnt);
    out[tid()] = make_float4(v ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);
}

// one camera path start: RNG, jitter, pinhole ray (start_path without depth of field, the configurations' case)
extern "C" __global__ void unit_camera_ray(RenderParams p, const RayIn *in, float4 *out, uint4 *outRng)
{
    p.pc.flags &= ~(uint32_t)PROSPER_PC_FLAG_DEPTH_OF_FIELD;
    const RayIn r = in[tid()];
    PathState st;
    LaneCounters cnt = {};
    start_path<false>(p, __builtin_bit_cast(uint32_t, r.a.x), __builtin_bit_cast(uint32_t, r.a.y), __builtin_bit_cast(uint32_t, r.a.z), st, cnt);
    out[2 * tid()] = make_float4(st.o.x, st.o.y, st.o.z, 0.0f);
    out[2 * tid() + 1] = make_float4(st.d.x, st.d.y, st.d.z, 0.0f);
    outRng[tid()] = make_uint4(st.rng.x, st.rng.y, st.rng.z, 0u);
}

// one sky lookup of an escaping path (bilinear fetch from the RGBA16F cube)
extern "C" __global__ void unit_sky_lookup(DeviceScene s, const RayIn *in, float4 *out)
{
    const RayIn r = in[tid()];
    const f3 c = sample_skybox(s, f3{r.b.x, r.b.y, r.b.z});
    out[tid()] = make_float4(c.x + r.a.x, c.y + r.a.y, c.z + r.a.z, r.a.w + r.b.w);
}

// one closest hit in wf_shade: surface from the shading record, material, light pick, BRDF, both lobes of the bounce,
// roulette, offset origin (main.rgen:146-223,90-144,269-283)
extern "C" __global__ void unit_shade_hit(DeviceScene s, RenderParams p, const RayIn *in, const uint4 *hits, float4 *out, uint32_t bounce)
{
    const RayIn r = in[tid()];
    const uint4 h = hits[tid()];
    LaneCounters cnt = {};
    Hit hit;
    hit.drawInstance = h.x;
    hit.primitive = h.y;
    hit.bary = f2{__builtin_bit_cast(float, h.z), __builtin_bit_cast(float, h.w)};
    hit.t = 0.0f;
    Rng rng{__builtin_bit_cast(uint32_t, r.a.x), __builtin_bit_cast(uint32_t, r.a.y), __builtin_bit_cast(uint32_t, r.a.z)};
    const f3 throughput = f3{r.b.x, r.b.y, r.b.z};
    const Surface sf = evaluate_surface<false, false>(s, f3{r.b.x, r.b.y, r.b.w}, hit, cnt);
    f3 l, irradiance;
    float d;
    f3 shC1 = {}, c0 = {};
    bool want = false;
    if (prepare_direct_lighting<false>(s, sf, throughput, rng, l, d, irradiance, cnt))
    {
        const f3 brdf = eval_brdf_times_nol(l, sf);
        shC1 = direct_lighting_value(s, throughput, irradiance, brdf, 1.0f);
        c0 = direct_lighting_value(s, throughput, irradiance, brdf, 0.0f);
        want = shadow_ray_matters(shC1, c0);
    }
    f3 rd, tp = throughput;
    importance_sample_bounce(sf, rng, tp, rd);
    bool alive = !throughput_is_zero(tp);
    if (alive && bounce > p.pc.rouletteStartBounce) alive = !(rng.rnd01() < fmax_(0.05f, 1.0f - max3(tp)));
    const f3 nO = offset_ray(sf.positionWS, sf.normalWS);
    out[4 * tid()] = make_float4(shC1.x, shC1.y, shC1.z, want ? d : 0.0f);
    out[4 * tid() + 1] = make_float4(nO.x, nO.y, nO.z, alive ? 1.0f : 0.0f);
    out[4 * tid() + 2] = make_float4(rd.x, rd.y, rd.z, l.x + l.y + l.z);
    out[4 * tid() + 3] = make_float4(tp.x, tp.y, tp.z, __builtin_bit_cast(float, rng.x ^ rng.y ^ rng.z));
}
