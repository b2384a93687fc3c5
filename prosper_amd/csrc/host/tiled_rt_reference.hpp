// host/tiled_rt_reference.hpp — render::TiledRtReference: the pass on one rank of a multi-GPU job.
//
// prosper renders the whole image on one GPU (RtReference::record asserts renderArea.offset == 0,
// src/render/RtReference.cpp:327).  Here one process (or thread) per GPU owns a TiledRtReference: record()
// renders this rank's interleaved 16-pixel stripes with the wrapped render::RtReference (same push constants,
// same accumulation state machine) and then hands the rank's RGBA32F tile to the ONE collective of the path,
// the RCCL gather to the root, where a HIP kernel de-interleaves the ranks' tiles into the full image
// (prosper_pt_gather_tiles).  Pixels are independent, so nothing else is exchanged.
#pragma once

#include <cstdint>

#include "rt_reference.hpp"

namespace render
{

class TiledRtReference
{
  public:
    static constexpr uint32_t sStripeWidth = 16;
    static constexpr uint32_t sCommIdBytes = PROSPER_PT_COMM_ID_BYTES;

    // One rank calls this and distributes the bytes (any transport: a file, MPI, a torch store).
    static void createCommId(uint8_t id[sCommIdBytes]);

    // Collective over the ranks (ncclCommInitRank).  Throws std::runtime_error without a gfx950 device or RCCL.
    void init(int32_t deviceOrdinal, uint32_t rank, uint32_t ranks, const uint8_t commId[sCommIdBytes],
              uint32_t root = 0, uint32_t createFlags = 0);

    [[nodiscard]] RtReference &pass() { return m_pass; }
    [[nodiscard]] uint32_t rank() const { return m_tile.stripeIndex; }
    [[nodiscard]] uint32_t ranks() const { return m_tile.stripeCount; }
    [[nodiscard]] bool isRoot() const { return m_tile.stripeIndex == m_root; }

    struct Output
    {
        RtReference::Output tile;           // this rank's stripes (device pointer, localWidth x height)
        const float *illumination{nullptr}; // root only: the gathered width x height image (device pointer), valid
                                            // for readers once waitForGather() has been enqueued on their stream
        uint32_t width{0};
        uint32_t height{0};
    };
    // RtReference::record for this rank's stripes, then the gather (enqueued on the context's communication stream:
    // it overlaps the next record()'s path stages when renderFlags has PROSPER_PT_RENDER_PIPELINED).
    [[nodiscard]] Output record(
        void *stream, scene::World &world, const scene::Camera &cam, const Rect2D &renderArea,
        const RtReference::Options &options, uint32_t nextFrame, uint32_t frameCount = 1, uint32_t renderFlags = 0);
    // Makes `stream` wait for the last record()'s gather and de-interleave.
    void waitForGather(void *stream);

  private:
    RtReference m_pass;
    prosper_pt_tile_desc m_tile{sStripeWidth, 0, 1};
    uint32_t m_root{0};
};

} // namespace render
