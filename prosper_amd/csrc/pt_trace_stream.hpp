// pt_trace_stream.hpp — wave-scheduled BVH traversal over a stream of rays (gfx950).
//
// A 64-lane wave that traces 64 incoherent rays in lockstep runs at ~25 % lane utilisation
// (measured, profiles/): the node loop lasts as long as its slowest lane, the (long) triangle code
// runs for the few lanes sitting in a leaf, the (longer) any-hit code for the odd lane on alpha
// geometry.  Here a wave owns a STREAM of n rays and every lane is a small state machine:
//
//     idle -> node* -> tri* -> (any-hit)? -> ... -> finished -> (commit, refill) -> node ...
//
// Each scheduling step the wave takes a ballot of the lane states and runs ONE phase — the one most
// lanes are waiting for — so node steps, triangle tests and any-hit evaluations each execute with a
// well-filled wave, and a lane whose ray is done is refilled from the stream instead of idling until
// the slowest lane of its batch finishes.  Hit selection is unchanged (DESIGN.md "hit contract":
// it is defined by arithmetic, not by traversal order), so results equal trace<>() bit for bit.
#pragma once

#include "pt_device.hpp"

// A phase keeps stepping while more than a third of the lanes that entered it are still in it.  Swept on
// C2/C3/C4 (profiles/r01_occupancy_ab.txt): leaving earlier to follow the majority costs more in censuses
// and half-empty first steps than the emptier late steps do (wf_trace -10 % on C2, -5 % on C3/C4 against the
// previous "2c > runner-up + c0").
// Finished rays are committed and lanes refilled once half of the wave waits for it (or nothing else can
// run): a refill step runs the commit and fetch code for the waiting lanes only, so it should be a full one.
// wf_trace C2 481 -> 457 us, C3 2901 -> 2871 us against "more lanes than any other phase" (C4 +3 %).
// Node steps feed triangle steps, so the triangle phase waits until it clearly leads (1.5x): -2 % on C2,
// neutral on C3 against a plain majority.
#ifndef PPT_TRI_BEATS_NODE
#define PPT_TRI_BEATS_NODE(t, n) (2u * (t) > 3u * (n))
#endif
#ifndef PPT_REFILL
#define PPT_REFILL(r, best) ((r) >= 32u)
#endif
#ifndef PPT_CONTINUE
#define PPT_CONTINUE(c, c0, other) (3u * (c) > (c0))
#endif
// The any-hit phase (rt/scene.rahit on the lanes holding a non-opaque candidate) used to run when those lanes
// outnumbered the node and triangle lanes.  Since the alpha bounds settle most candidates in a few dozen instructions the
// phase is cheap, and a lane it returns to the triangle phase early is worth more than a fuller any-hit step: it runs
// once a quarter as many lanes wait for it as for the fullest other phase.  wf_trace on C4, alone: 6603 us (a > best),
// 6291 (2a), 6223 (4a, kept), 6250 (8a), 6316 (16a), 6350 (a > 0); FlightHelmet 381 -> 365; C2 unchanged
// (profiles/r03_alpha_bounds.txt).
#ifndef PPT_ANY_BEATS
#define PPT_ANY_BEATS(a, best) (4u * (a) > (best))
#endif

namespace ppt
{

PPT_D f3 xyz3(float4 v) { return f3{v.x, v.y, v.z}; }

enum : uint32_t
{
    kLaneIdle = 0,
    kLaneNode = 1,
    kLaneTri = 2,
    kLaneAny = 3,
    kLaneFinished = 4,
};

struct StreamRay
{
    f3 o, d;
    float tMin, tMax;
    uint32_t seed;
};

// fetch(i) -> StreamRay for stream position i (called by the lanes that take a new ray);
// commit(pred, i, found, hit, dir) is called by ALL lanes in converged code; lanes with pred set
// hand over the result of ray i with direction dir (so commit may compact with ballots).
template <bool ANY, bool COUNT, class Geom, class Fetch, class Commit>
PPT_D void trace_stream(
    const Geom &g, const DeviceScene &s, uint32_t n, const TraversalStack &stack, LaneCounters &cnt, Fetch &&fetch,
    Commit &&commit)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;

    uint32_t next = 0; // wave-uniform: next stream position to hand out
    uint32_t state = kLaneIdle;
    uint32_t ray = 0;
    f3 o = {}, d = {}, invd = {};
    RaySlabs rs = {};
    float tMin = 0.0f, tMaxIn = 0.0f;
    uint32_t seed = 0;
    int32_t node = 0, sp = 0;
    uint32_t triFirst = 0, triCount = 0;
    Hit hit = {};
    // candidate waiting for its any-hit evaluation: its alpha record (which also names drawInstance and primitive)
    uint32_t cAlpha = 0;
    float cT = 0.0f, cBu = 0.0f, cBv = 0.0f;

    // after a leaf is exhausted or a node misses: pop the next entry or finish
    auto pop = [&]() {
        if (sp == 0)
            state = kLaneFinished;
        else
        {
            node = stack.pop(sp);
            if (node >= 0)
                state = kLaneNode;
            else
            {
                const uint32_t ref = (uint32_t)~node;
                triFirst = ref >> 3;
                triCount = (ref & 7u) + 1u;
                state = kLaneTri;
            }
        }
    };
    auto accept = [&](uint32_t di, uint32_t prim, float t, float bu, float bv) {
        hit.drawInstance = di;
        hit.primitive = prim;
        hit.bary = f2{bu, bv};
        hit.t = t;
        if (ANY) state = kLaneFinished;
    };

    while (true)
    {
        const uint32_t nNode = (uint32_t)__builtin_popcountll(__ballot(state == kLaneNode));
        const uint32_t nTri = (uint32_t)__builtin_popcountll(__ballot(state == kLaneTri));
        const uint32_t nAny = (uint32_t)__builtin_popcountll(__ballot(state == kLaneAny));
        const unsigned long long mFin = __ballot(state == kLaneFinished);
        const uint32_t nFin = (uint32_t)__builtin_popcountll(mFin);
        const uint32_t nIdle = 64u - nNode - nTri - nAny - nFin;
        const bool raysLeft = next < n;
        const uint32_t nRefill = nFin + (raysLeft ? nIdle : 0u);
        const uint32_t nWork = nNode + nTri + nAny;
        if (nWork == 0 && nRefill == 0) break;

        // phase with the most lanes waiting; refill also runs when nothing else can
        uint32_t best = nNode, pick = kLaneNode;
        if (PPT_TRI_BEATS_NODE(nTri, nNode))
        {
            best = nTri;
            pick = kLaneTri;
        }
        if (PPT_ANY_BEATS(nAny, best))
        {
            best = nAny;
            pick = kLaneAny;
        }
        if (PPT_REFILL(nRefill, best) || nWork == 0) pick = kLaneFinished;

        // the picked phase then steps several times in a row (PPT_CONTINUE above): saves the five-ballot census
        // per step
        if (pick == kLaneNode)
        {
            const uint32_t other = nTri > nAny ? (nTri > nRefill ? nTri : nRefill) : (nAny > nRefill ? nAny : nRefill);
            (void)other;
            uint32_t c = nNode;
            do
            {
                if constexpr (COUNT) cnt.nodePhaseSteps += lane == 0 ? 1u : 0u;
#ifdef PPT_EXPERIMENT_COUNT_DRAIN_STEPS
                // measurement only (profiles/r04_stream_tails.txt): node steps taken after the stream ran out of rays to hand out
                if constexpr (COUNT) cnt.historyReads += (lane == 0 && next >= n) ? 1u : 0u;
#endif
#ifdef PPT_EXPERIMENT_COUNT_IDLE_LANES
                // measurement only: lanes WITHOUT A RAY (idle, nothing left to refill from) summed over the node steps of the
                // shadow stream (historyReads) and of the closest-hit stream (pixelsWritten): what a merged stream could fill
                if constexpr (COUNT)
                {
                    const uint32_t idle = (uint32_t)__builtin_popcountll(__ballot(state == kLaneIdle || state == kLaneFinished));
                    if (lane == 0 && next >= n) (ANY ? cnt.historyReads : cnt.pixelsWritten) += idle;
                }
#endif
#ifdef PPT_EXPERIMENT_COUNT_NARROW_STEPS
                if constexpr (COUNT) // measurement only (profiles/r02_exec_mask_microbench.txt): node steps with <= 8 lanes
                {
                    const int active = __builtin_popcountll(__ballot(state == kLaneNode));
                    cnt.historyReads += (lane == 0 && active <= 8) ? 1u : 0u;
                }
#endif
                if (state == kLaneNode)
                {
                    if constexpr (COUNT) cnt.nodeVisits++;
                    float e[4];
                    int32_t ref[4];
                    bool entered;
#ifdef PPT_EXPERIMENT_UNSORTED_CLOSEST
                    // measurement only (profiles/r04_child_order.txt): closest-hit rays take the children in storage order
                    // like the any-hit rays - what the five-comparator sort buys in node visits against what it costs
                    if constexpr (true)
#else
                    if constexpr (ANY)
#endif
                    {
                        g.template test_node<false>(node, o, rs, tMin, hit.t, e, ref);
                        entered = descend_any(e, ref, stack, sp, node);
                    }
                    else
                    {
                        g.template test_node<true>(node, o, rs, tMin, hit.t, e, ref);
                        stack.push_hit_children(sp, e, ref);
                        entered = e[0] < kInf;
                        if (entered) node = ref[0];
                    }
                    if (entered)
                    {
                        if (node < 0)
                        {
                            const uint32_t leaf = (uint32_t)~node;
                            triFirst = leaf >> 3;
                            triCount = (leaf & 7u) + 1u;
                            state = kLaneTri;
                        }
                    }
                    else
                        pop();
                }
                c = (uint32_t)__builtin_popcountll(__ballot(state == kLaneNode));
            } while (PPT_CONTINUE(c, nNode, other));
        }
        else if (pick == kLaneTri)
        {
            const uint32_t other = nNode > nAny ? (nNode > nRefill ? nNode : nRefill) : (nAny > nRefill ? nAny : nRefill);
            (void)other;
            uint32_t c = nTri;
            do
            {
                if constexpr (COUNT) cnt.trianglePhaseSteps += lane == 0 ? 1u : 0u;
#ifdef PPT_EXPERIMENT_COUNT_DRAIN_STEPS
                if constexpr (COUNT) cnt.pixelsWritten += (lane == 0 && next >= n) ? 1u : 0u; // triangle steps after the stream ran out
#endif
#ifdef PPT_EXPERIMENT_COUNT_NARROW_STEPS
                if constexpr (COUNT) // triangle steps with <= 8 lanes
                {
                    const int active = __builtin_popcountll(__ballot(state == kLaneTri));
                    cnt.pixelsWritten += (lane == 0 && active <= 8) ? 1u : 0u;
                }
#endif
                // One step tests the next TWO triangles of the leaf when some lane has two left (SAH leaves are
                // mostly pairs: a quad, a box face): both edge-function sets (the cheap rejection), then ONE pass
                // through distance / guard / acceptance for whichever of them the ray goes through.  A ray through
                // both (the shared diagonal) takes the first now and meets the second again next step.
                const bool inTri = state == kLaneTri;
                const bool two = inTri && triCount >= 2u;
                const bool anyTwo = __any(two);
                if (inTri)
                {
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
                    if (candidate && !ANY && hit.drawInstance != kMissIndex)
                    {
                        if (t > hit.t) candidate = false;
                        if (t == hit.t && !(di < hit.drawInstance || (di == hit.drawInstance && prim < hit.primitive)))
                            candidate = false;
                    }
                    if (candidate)
                    {
                        if (flags & kTriFlagOpaque)
                            accept(di, prim, t, bu, bv);
                        else
                        {
                            // rt/scene.rahit:18-39 runs in a phase of its own (kLaneAny)
#ifdef PPT_EXPERIMENT_SETTLE_IN_TRI_PHASE
                            // measured slower (profiles/r03_alpha_bounds.txt): the two dependent loads lengthen every
                            // triangle step of the wave by more than the any-hit phases they save
                            AlphaFootprint fp;
                            const uint32_t verdict =
                                any_hit_settle<COUNT>(s, flags >> kTriAlphaShift, f2{bu, bv}, seed, cnt, fp);
                            if (verdict == kAlphaAccept)
                                accept(di, prim, t, bu, bv);
                            else if (verdict == kAlphaUndecided)
#endif
                            {
                                cAlpha = flags >> kTriAlphaShift;
                                cT = t;
                                cBu = bu;
                                cBv = bv;
                                state = kLaneAny;
                            }
                        }
                    }
                    if (state == kLaneTri && triCount == 0) pop();
                }
                c = (uint32_t)__builtin_popcountll(__ballot(state == kLaneTri));
            } while (PPT_CONTINUE(c, nTri, other));
        }
        else if (pick == kLaneAny)
        {
#ifdef PPT_EXPERIMENT_COUNT_ANY_STEPS
            if constexpr (COUNT) cnt.historyReads += lane == 0 ? 1u : 0u; // measurement only: any-hit phase steps
#endif
            if (state == kLaneAny)
            {
                state = kLaneTri;
#ifdef PPT_EXPERIMENT_SETTLE_IN_TRI_PHASE
                // the footprint again (a lane keeps only the candidate between phases), then the exact evaluation
                AlphaFootprint fp;
                LaneCounters uncounted = {};
                (void)any_hit_settle<false>(s, cAlpha, f2{cBu, cBv}, seed, uncounted, fp);
                if (any_hit_exact<COUNT>(fp, cnt))
#else
                if (any_hit_record<COUNT>(s, cAlpha, f2{cBu, cBv}, seed, cnt))
#endif
                {
                    const AlphaTriangle *rec = s.alphaTriangles + cAlpha;
                    accept(rec->drawInstance, rec->primitive, cT, cBu, cBv);
                }
                if (state == kLaneTri && triCount == 0) pop();
            }
        }
        else
        {
            // hand finished rays over (converged call), then refill finished + idle lanes
            commit(state == kLaneFinished, ray, hit.drawInstance != kMissIndex, hit, d);
            if (state == kLaneFinished) state = kLaneIdle;
            const unsigned long long mIdle = __ballot(state == kLaneIdle);
            const uint32_t rank = (uint32_t)__builtin_popcountll(mIdle & below);
            const uint32_t idx = next + rank;
            if (state == kLaneIdle && idx < n)
            {
                ray = idx;
                const StreamRay r = fetch(idx);
                o = r.o;
                d = r.d;
                tMin = r.tMin;
                tMaxIn = r.tMax;
                seed = r.seed;
                invd = f3{safe_rcp_dir(d.x), safe_rcp_dir(d.y), safe_rcp_dir(d.z)};
                rs = make_ray_slabs(invd);
                hit.drawInstance = kMissIndex;
                hit.primitive = kMissIndex;
                hit.bary = f2{0.0f, 0.0f};
                hit.t = tMaxIn;
                sp = 0;
                node = 0; // the root is always an inner node
                state = kLaneNode;
            }
            const uint32_t handed = (uint32_t)__builtin_popcountll(mIdle);
            next = (n - next < handed) ? n : next + handed;
        }
    }
}

} // namespace ppt
