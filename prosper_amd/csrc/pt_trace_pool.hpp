// pt_trace_pool.hpp — BVH traversal of a ray stream out of a per-wave ray POOL in LDS (gfx950).
//
// trace_stream() (pt_trace_stream.hpp) ties a ray to a lane for its whole life: a step of the node phase runs for
// the lanes whose ray happens to be at a node, and the wave's other lanes idle (measured lane utilisation 0.49 on
// C2, 0.29 on C4: profiles/r02_*_pmc.json).  Here the rays of a wave live in LDS, P >= 64 of them at a time, and a
// lane owns nothing between steps:
//
//     queues (LDS, one byte per entry): node | triangle | any-hit | finished | free       - each ray slot is in one
//     step of phase X:   pop up to 64 slots from queue X -> load what X needs -> run X for one visit ->
//                        store what changed -> push every slot on the queue of its new state
//
// so a step runs with min(64, |queue X|) lanes, and with P = 2 x 64 rays in the pool the busiest queue nearly always
// holds a full wave.  Queue heads are wave-uniform scalars; ranks come from ballots; nothing is atomic and the
// order rays finish in does not matter (the caller's commit compacts with ballots, and every result is a function
// of its own ray alone: DESIGN.md "hit contract"), so results equal trace_stream()'s and trace<>()'s bit for bit.
//
// LDS per wave: the 9 dwords of ray state a node step touches + S stack entries per slot + 5 queue bytes per slot; the
// rest of the ray, the accepted hit and the pending any-hit candidate (15 dwords per slot, touched a few times per
// ray) live in a global scratch block of the wave, as do stack entries deeper than S.
#pragma once

#include "pt_device.hpp"
#include "pt_trace_stream.hpp"

namespace ppt
{

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;

// LDS: what a node step reads and writes
enum : uint32_t
{
    kPoolOx = 0,
    kPoolOy,
    kPoolOz,
    kPoolIx,
    kPoolIy,
    kPoolIz,
    kPoolT,      // distance of the accepted hit, tMax while there is none
    kPoolCursor, // node state: node index; triangle / any-hit state: first << 4 | count
    kPoolSp,
    kPoolFields,
};
enum : uint32_t
{
    kQueueNode = 0,
    kQueueTri,
    kQueueAny,
    kQueueFinished,
    kQueueFree,
    kPoolQueues,
};
// global scratch of a wave, dword k of slot i at scratch[k * P + i]: the rest of the ray (read next to the triangle
// fetch of a triangle step, which has the same latency), the accepted hit and the pending any-hit candidate
enum : uint32_t
{
    kScratchDx = 0,
    kScratchDy,
    kScratchDz,
    kScratchTMax,
    kScratchSeed,
    kScratchRay, // stream position
    kScratchHitDi,
    kScratchHitPrim,
    kScratchHitBu,
    kScratchHitBv,
    kScratchCandDi,
    kScratchCandPrim,
    kScratchCandT,
    kScratchCandBu,
    kScratchCandBv,
    kPoolScratchFields,
};

template <uint32_t P, uint32_t S>
struct RayPool
{
    static constexpr uint32_t kLdsDwords = (kPoolFields + S) * P + (kPoolQueues * P + 3u) / 4u;
    lds_u32 *fields;   // this wave's block: dword k of slot i at fields[k * P + i]
    lds_int32 *stack;  // entry e of slot i at stack[e * P + i]
    lds_u8 *queues;    // entry j of queue k at queues[k * P + j]
    uint32_t *scratch; // global, this wave's: kPoolScratchFields * P dwords, then the stack overflow
    uint32_t overflowEntries;

    PPT_D static RayPool carve(uint32_t *ldsBlock, uint32_t waveInGroup, uint32_t *scratchBase, uint32_t waveGlobal,
                               uint32_t overflowEntries)
    {
        lds_u32 *base = (lds_u32 *)ldsBlock + waveInGroup * kLdsDwords;
        RayPool r;
        r.fields = base;
        r.stack = (lds_int32 *)(base + kPoolFields * P);
        r.queues = (lds_u8 *)(base + (kPoolFields + S) * P);
        r.scratch = scratchBase + (size_t)waveGlobal * scratch_dwords(overflowEntries);
        r.overflowEntries = overflowEntries;
        return r;
    }
    PPT_HD static uint32_t scratch_dwords(uint32_t overflowEntries) { return (kPoolScratchFields + overflowEntries) * P; }
    PPT_D TraversalStack stack_of(uint32_t slot) const
    {
        return TraversalStack{stack + slot, (int32_t *)scratch + kPoolScratchFields * P + slot, S, P, P};
    }
};

// Same contract as trace_stream(); every ray of the stream has the same tMin.
template <bool ANY, bool COUNT, uint32_t B, uint32_t P, uint32_t S, class Geom, class Fetch, class Commit>
PPT_D void trace_pool(
    const Geom &g, const DeviceScene &s, uint32_t nRays, float tMin, const RayPool<P, S> &pool, LaneCounters &cnt,
    Fetch &&fetch, Commit &&commit)
{
    // the stream length is the same in every lane; tell the compiler (scalar queue arithmetic, uniform branches)
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)nRays);
    static_assert(P >= 64u && P <= 256u && P % 32u == 0u, "queue entries are bytes; fetch and commit steps are 64 wide");
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    lds_u32 *const F = pool.fields;
    lds_u8 *const Q = pool.queues;
    uint32_t *const X = pool.scratch;

    // wave-uniform queue lengths and stream cursor
    uint32_t nNode = 0, nTri = 0, nAny = 0, nFin = 0, nFree = P, next = 0;
    for (uint32_t i = lane; i < P; i += 64u) Q[kQueueFree * P + i] = (uint8_t)i;

    auto fld = [&](uint32_t k, uint32_t slot) -> lds_u32 & { return F[k * P + slot]; };
    auto fldf = [&](uint32_t k, uint32_t slot) { return __builtin_bit_cast(float, (uint32_t)F[k * P + slot]); };
    auto setf = [&](uint32_t k, uint32_t slot, float v) { F[k * P + slot] = __builtin_bit_cast(uint32_t, v); };
    auto xf = [&](uint32_t k, uint32_t slot) { return __builtin_bit_cast(float, X[k * P + slot]); };
    auto setxf = [&](uint32_t k, uint32_t slot, float v) { X[k * P + slot] = __builtin_bit_cast(uint32_t, v); };
    // takes the top min(64, count) entries of a queue; lanes beyond that get act = false
    auto take = [&](uint32_t queue, uint32_t &count, bool &act) {
        const uint32_t k = count < 64u ? count : 64u;
        count -= k;
        act = lane < k;
        return act ? (uint32_t)Q[queue * P + count + lane] : 0u;
    };
    auto give = [&](bool pred, uint32_t slot, uint32_t queue, uint32_t &count) {
        const unsigned long long m = __ballot(pred);
        if (pred) Q[queue * P + count + (uint32_t)__builtin_popcountll(m & below)] = (uint8_t)slot;
        count += (uint32_t)__builtin_popcountll(m);
    };
    auto decode_leaf = [](int32_t node) {
        const uint32_t ref = (uint32_t)~node;
        return ((ref >> 3) << 4) | ((ref & 7u) + 1u);
    };

    while (true)
    {
        __builtin_amdgcn_wave_barrier();
        const uint32_t left = n - next;
        const uint32_t canFetch = nFree < left ? nFree : left;
        const uint32_t work = nNode + nTri + nAny;
        if (work + nFin + canFetch == 0u) break;

        // the phase with a full wave waiting, finished rays and fresh ones first (they keep the pool full);
        // otherwise the longest queue
        uint32_t pick;
        if (nFin >= 64u)
            pick = kQueueFinished;
        else if (canFetch >= 64u)
            pick = kQueueFree;
        else
        {
            uint32_t best = nNode;
            pick = kQueueNode;
            if (nTri >= 64u || nTri > best)
            {
                best = nTri;
                pick = kQueueTri;
            }
            if (best < 64u && nAny > best)
            {
                best = nAny;
                pick = kQueueAny;
            }
            if (best < 64u)
            {
                if (nFin > best)
                {
                    best = nFin;
                    pick = kQueueFinished;
                }
                if (canFetch > best) pick = kQueueFree;
            }
        }

        if (pick == kQueueNode)
        {
            // B batches of <= 64 rays per step: all their loads are issued before the first batch is computed, so a
            // wave has up to B x 64 node fetches in flight (the traversal is a pointer chase: when the nodes come from
            // L2 / HBM the time goes into their latency, not into instructions)
            do
            {
                bool act[B];
                uint32_t slot[B], state[B];
                f3 o[B], invd[B];
                float tCur[B];
                int32_t node[B], sp[B];
                NodeData nd[B];
#pragma unroll
                for (uint32_t k = 0; k < B; ++k)
                {
                    slot[k] = take(kQueueNode, nNode, act[k]);
                    o[k] = f3{fldf(kPoolOx, slot[k]), fldf(kPoolOy, slot[k]), fldf(kPoolOz, slot[k])};
                    invd[k] = f3{fldf(kPoolIx, slot[k]), fldf(kPoolIy, slot[k]), fldf(kPoolIz, slot[k])};
                    tCur[k] = fldf(kPoolT, slot[k]);
                    node[k] = act[k] ? (int32_t)fld(kPoolCursor, slot[k]) : 0;
                    sp[k] = (int32_t)fld(kPoolSp, slot[k]);
                }
#pragma unroll
                for (uint32_t k = 0; k < B; ++k) nd[k] = g.node(node[k]);
#pragma unroll
                for (uint32_t k = 0; k < B; ++k)
                {
                    if constexpr (COUNT) cnt.nodePhaseSteps += (lane == 0 && __any(act[k])) ? 1u : 0u;
                    state[k] = kLaneIdle;
                    if (act[k])
                    {
                        const TraversalStack stack = pool.stack_of(slot[k]);
                        if constexpr (COUNT) cnt.nodeVisits++;
                        float e[4];
                        int32_t ref[4];
                        bool entered;
                        if constexpr (ANY)
                        {
                            intersect_node4<false>(nd[k], o[k], make_ray_slabs(invd[k]), tMin, tCur[k], e, ref);
                            entered = descend_any(e, ref, stack, sp[k], node[k]);
                        }
                        else
                        {
                            intersect_node4(nd[k], o[k], make_ray_slabs(invd[k]), tMin, tCur[k], e, ref);
                            if (e[3] < kInf) stack.push(sp[k], ref[3]);
                            if (e[2] < kInf) stack.push(sp[k], ref[2]);
                            if (e[1] < kInf) stack.push(sp[k], ref[1]);
                            entered = e[0] < kInf;
                            if (entered) node[k] = ref[0];
                        }
                        state[k] = kLaneNode;
                        if (!entered)
                        {
                            if (sp[k] == 0)
                                state[k] = kLaneFinished;
                            else
                                node[k] = stack.pop(sp[k]);
                        }
                        uint32_t cursor = (uint32_t)node[k];
                        if (state[k] == kLaneNode && node[k] < 0)
                        {
                            cursor = decode_leaf(node[k]);
                            state[k] = kLaneTri;
                        }
                        fld(kPoolCursor, slot[k]) = cursor;
                        fld(kPoolSp, slot[k]) = (uint32_t)sp[k];
                    }
                }
#pragma unroll
                for (uint32_t k = 0; k < B; ++k)
                {
                    give(state[k] == kLaneNode, slot[k], kQueueNode, nNode);
                    give(state[k] == kLaneTri, slot[k], kQueueTri, nTri);
                    give(state[k] == kLaneFinished, slot[k], kQueueFinished, nFin);
                }
                __builtin_amdgcn_wave_barrier();
            } while (nNode >= 64u && nTri < 64u);
        }
        else if (pick == kQueueTri)
        {
            do
            {
                if constexpr (COUNT) cnt.trianglePhaseSteps += lane == 0 ? 1u : 0u;
                bool act;
                const uint32_t slot = take(kQueueTri, nTri, act);
                uint32_t state = kLaneIdle;
                uint32_t cursor = act ? (uint32_t)fld(kPoolCursor, slot) : 0u;
                // two triangles per step when some ray has two left in its leaf (see trace_stream)
                const bool two = act && (cursor & 15u) >= 2u;
                const bool anyTwo = __any(two);
                if (act)
                {
                    const f3 o = f3{fldf(kPoolOx, slot), fldf(kPoolOy, slot), fldf(kPoolOz, slot)};
                    const f3 d = f3{xf(kScratchDx, slot), xf(kScratchDy, slot), xf(kScratchDz, slot)};
                    const f3 invd = f3{fldf(kPoolIx, slot), fldf(kPoolIy, slot), fldf(kPoolIz, slot)};
                    const float tMaxIn = xf(kScratchTMax, slot);
                    const float tCur = fldf(kPoolT, slot);
                    uint32_t triFirst = cursor >> 4, triCount = cursor & 15u;
                    TriangleData td = g.tri(triFirst);
                    EdgeFunctions e = edge_functions(o, d, xyz3(td.a), xyz3(td.b), xyz3(td.c));
                    uint32_t consumed = 1u;
                    if (anyTwo)
                    {
                        const TriangleData td2 = g.tri(two ? triFirst + 1u : triFirst);
                        EdgeFunctions e2 = edge_functions(o, d, xyz3(td2.a), xyz3(td2.b), xyz3(td2.c));
                        e2.pass = e2.pass && two;
                        if (two && !(e.pass && e2.pass)) consumed = 2u;
                        if (!e.pass && e2.pass)
                        {
                            td = td2;
                            e = e2;
                        }
                    }
                    const float4 a = td.a, b = td.b, c4 = td.c;
                    triFirst += consumed;
                    triCount -= consumed;
                    if constexpr (COUNT)
                    {
                        cnt.triangleTests += consumed;
                        cnt.shortIndexTriangleTests +=
                            (__builtin_bit_cast(uint32_t, c4.w) & kTriFlagShortIndices) ? consumed : 0u;
                    }
                    float t, bu, bv;
                    bool candidate =
                        e.pass && finish_triangle(e, o, d, invd, xyz3(a), xyz3(b), xyz3(c4), tMin, tMaxIn, t, bu, bv);
                    const uint32_t di = __builtin_bit_cast(uint32_t, a.w);
                    const uint32_t prim = __builtin_bit_cast(uint32_t, b.w);
                    const uint32_t flags = __builtin_bit_cast(uint32_t, c4.w);
                    if (candidate && !ANY && tCur < tMaxIn) // a hit was accepted before (its t is < tMaxIn)
                    {
                        if (t > tCur) candidate = false;
                        if (t == tCur)
                        {
                            const uint32_t hDi = X[kScratchHitDi * P + slot], hPrim = X[kScratchHitPrim * P + slot];
                            if (!(di < hDi || (di == hDi && prim < hPrim))) candidate = false;
                        }
                    }
                    state = kLaneTri;
                    if (candidate)
                    {
                        if (flags & kTriFlagOpaque)
                        {
                            X[kScratchHitDi * P + slot] = di;
                            X[kScratchHitPrim * P + slot] = prim;
                            X[kScratchHitBu * P + slot] = __builtin_bit_cast(uint32_t, bu);
                            X[kScratchHitBv * P + slot] = __builtin_bit_cast(uint32_t, bv);
                            setf(kPoolT, slot, t);
                            if (ANY) state = kLaneFinished;
                        }
                        else
                        {
                            X[kScratchCandDi * P + slot] = di;
                            X[kScratchCandPrim * P + slot] = prim;
                            X[kScratchCandT * P + slot] = __builtin_bit_cast(uint32_t, t);
                            X[kScratchCandBu * P + slot] = __builtin_bit_cast(uint32_t, bu);
                            X[kScratchCandBv * P + slot] = __builtin_bit_cast(uint32_t, bv);
                            state = kLaneAny;
                        }
                    }
                    cursor = (triFirst << 4) | triCount;
                    if (state == kLaneTri && triCount == 0u)
                    {
                        int32_t sp = (int32_t)fld(kPoolSp, slot);
                        if (sp == 0)
                            state = kLaneFinished;
                        else
                        {
                            const int32_t node = pool.stack_of(slot).pop(sp);
                            fld(kPoolSp, slot) = (uint32_t)sp;
                            if (node >= 0)
                            {
                                cursor = (uint32_t)node;
                                state = kLaneNode;
                            }
                            else
                                cursor = decode_leaf(node);
                        }
                    }
                    fld(kPoolCursor, slot) = cursor;
                }
                give(state == kLaneTri, slot, kQueueTri, nTri);
                give(state == kLaneNode, slot, kQueueNode, nNode);
                give(state == kLaneAny, slot, kQueueAny, nAny);
                give(state == kLaneFinished, slot, kQueueFinished, nFin);
                __builtin_amdgcn_wave_barrier();
            } while (nTri >= 64u);
        }
        else if (pick == kQueueAny)
        {
            bool act;
            const uint32_t slot = take(kQueueAny, nAny, act);
            uint32_t state = kLaneIdle;
            if (act)
            {
                const uint32_t cDi = X[kScratchCandDi * P + slot], cPrim = X[kScratchCandPrim * P + slot];
                const uint32_t cT = X[kScratchCandT * P + slot], cBu = X[kScratchCandBu * P + slot],
                               cBv = X[kScratchCandBv * P + slot];
                const uint32_t seed = X[kScratchSeed * P + slot];
                uint32_t cursor = fld(kPoolCursor, slot);
                state = kLaneTri;
                if (any_hit<COUNT>(
                        s, cDi, cPrim, f2{__builtin_bit_cast(float, cBu), __builtin_bit_cast(float, cBv)}, seed, cnt))
                {
                    X[kScratchHitDi * P + slot] = cDi;
                    X[kScratchHitPrim * P + slot] = cPrim;
                    X[kScratchHitBu * P + slot] = cBu;
                    X[kScratchHitBv * P + slot] = cBv;
                    fld(kPoolT, slot) = cT;
                    if (ANY) state = kLaneFinished;
                }
                if (state == kLaneTri && (cursor & 15u) == 0u)
                {
                    int32_t sp = (int32_t)fld(kPoolSp, slot);
                    if (sp == 0)
                        state = kLaneFinished;
                    else
                    {
                        const int32_t node = pool.stack_of(slot).pop(sp);
                        fld(kPoolSp, slot) = (uint32_t)sp;
                        if (node >= 0)
                        {
                            cursor = (uint32_t)node;
                            state = kLaneNode;
                        }
                        else
                            cursor = decode_leaf(node);
                        fld(kPoolCursor, slot) = cursor;
                    }
                }
            }
            give(state == kLaneTri, slot, kQueueTri, nTri);
            give(state == kLaneNode, slot, kQueueNode, nNode);
            give(state == kLaneFinished, slot, kQueueFinished, nFin);
        }
        else if (pick == kQueueFinished)
        {
            bool act;
            const uint32_t slot = take(kQueueFinished, nFin, act);
            uint32_t ray = 0;
            bool found = false;
            Hit hit = {};
            f3 d = {};
            if (act)
            {
                ray = X[kScratchRay * P + slot];
                hit.t = fldf(kPoolT, slot);
                found = hit.t < xf(kScratchTMax, slot);
                d = f3{xf(kScratchDx, slot), xf(kScratchDy, slot), xf(kScratchDz, slot)};
                hit.drawInstance = kMissIndex;
                hit.primitive = kMissIndex;
                if (found && !ANY)
                {
                    hit.drawInstance = X[kScratchHitDi * P + slot];
                    hit.primitive = X[kScratchHitPrim * P + slot];
                    hit.bary = f2{__builtin_bit_cast(float, X[kScratchHitBu * P + slot]),
                                  __builtin_bit_cast(float, X[kScratchHitBv * P + slot])};
                }
            }
            commit(act, ray, found, hit, d);
            give(act, slot, kQueueFree, nFree);
        }
        else
        {
            const uint32_t k = canFetch < 64u ? canFetch : 64u;
            nFree -= k;
            const bool act = lane < k;
            uint32_t slot = 0;
            if (act)
            {
                slot = Q[kQueueFree * P + nFree + lane];
                const uint32_t idx = next + lane;
                const StreamRay r = fetch(idx);
                setf(kPoolOx, slot, r.o.x);
                setf(kPoolOy, slot, r.o.y);
                setf(kPoolOz, slot, r.o.z);
                setxf(kScratchDx, slot, r.d.x);
                setxf(kScratchDy, slot, r.d.y);
                setxf(kScratchDz, slot, r.d.z);
                setf(kPoolIx, slot, safe_rcp_dir(r.d.x));
                setf(kPoolIy, slot, safe_rcp_dir(r.d.y));
                setf(kPoolIz, slot, safe_rcp_dir(r.d.z));
                setxf(kScratchTMax, slot, r.tMax);
                setf(kPoolT, slot, r.tMax);
                X[kScratchSeed * P + slot] = r.seed;
                X[kScratchRay * P + slot] = idx;
                fld(kPoolCursor, slot) = 0u; // the root is always an inner node
                fld(kPoolSp, slot) = 0u;
            }
            next += k;
            give(act, slot, kQueueNode, nNode);
        }
    }
}

} // namespace ppt
