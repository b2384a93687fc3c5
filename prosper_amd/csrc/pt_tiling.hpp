// pt_tiling.hpp — multi-GPU side of the pass: image stripes per rank + ONE gather of the per-rank HDR tiles over
// RCCL (xGMI) + a de-interleave kernel on the root (north star: "the image is tiled across the 8 GPUs of
// one node with an RCCL gather over xGMI of per-tile HDR buffers").  The reference never splits the image
// (asserts renderArea.offset == 0, src/render/RtReference.cpp:327): this is code the build owns.
#pragma once

#include "pt_context.hpp"

namespace ppt
{

// Frees the communicator (if owned), the comm stream, events and the staging buffer.
void destroy_tiling(prosper_pt_ctx *ctx);
// Called by prosper_pt_render_frames before it enqueues the accumulate kernel on `stream`: the tile may still be
// read by a gather in flight on the comm stream.
void wait_for_gather_before_writing_tile(prosper_pt_ctx *ctx, hipStream_t stream);
// Texels of rank r's tile for an image `width` wide cut into `stripeWidth` stripes over `ranks` ranks
uint32_t tile_local_width(uint32_t width, uint32_t stripeWidth, uint32_t rank, uint32_t ranks);

} // namespace ppt
