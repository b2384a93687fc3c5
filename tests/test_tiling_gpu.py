"""The product-side multi-GPU path on ONE GPU (prosper_pt_comm_*, prosper_pt_gather_tiles, prosper_pt_deinterleave_tiles,
render::TiledRtReference): N rank tiles rendered one after the other into the layout the root's receive buffer has,
de-interleaved by the HIP kernel, equal the whole-image render bit for bit; and a real RCCL communicator (one rank - two
ranks cannot share a GPU under RCCL) takes the gather path end to end.  N > 1 ranks with data exchange: the gloo test
tests/test_multi_rank_cpu.py here, bench.py --gpus N on a multi-GPU node."""
import ctypes

import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import structs as S, tiling

pytestmark = pytest.mark.gpu


def _camera(oracle, world, w, h):
    c = world.camera
    return oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)


class _DeviceBuffer:
    def __init__(self, nbytes):
        self.hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library already uses
        self.ptr = ctypes.c_void_p()
        self.nbytes = nbytes
        assert self.hip.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(nbytes)) == 0

    def upload(self, offset, array):
        a = np.ascontiguousarray(array)
        assert self.hip.hipMemcpy(ctypes.c_void_p(self.ptr.value + offset), ctypes.c_void_p(a.ctypes.data),
                                  ctypes.c_size_t(a.nbytes), 1) == 0

    def download(self, shape):
        out = np.empty(shape, np.float32)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), self.ptr, ctypes.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


@pytest.mark.parametrize("ranks,w,h", [(2, 256, 72), (4, 320, 40), (8, 384, 48), (3, 200, 33), (5, 1920, 16)])
def test_deinterleave_kernel_rebuilds_the_whole_image(gpu_ctx, oracle, cornell_world, ranks, w, h):
    """ranks = 3 / 5: stripe counts that do not divide (the per-rank widths differ, the last stripe is partial)."""
    cam, fl = _camera(oracle, cornell_world, w, h)
    gpu_ctx.upload_scene(cornell_world)
    pc = default_pc(S, fl, max_bounces=3, ibl=True)
    gpu_ctx.render(pc, cam, w, h, frames=2)
    whole = gpu_ctx.read_hdr()
    staging = _DeviceBuffer(w * h * 16)
    full = _DeviceBuffer(w * h * 16)
    offset = 0
    for r in range(ranks):
        gpu_ctx.render(pc, cam, w, h, tile=tiling.tile_for_rank(r, ranks), frames=2)
        t = gpu_ctx.read_hdr()
        assert t.shape == (h, tiling.local_width(w, r, ranks), 4)
        staging.upload(offset, t)
        offset += t.nbytes
    assert offset == w * h * 16
    gpu_ctx.deinterleave_tiles(staging.ptr.value, ranks, tiling.STRIPE_WIDTH, w, h, full.ptr.value)
    got = full.download((h, w, 4))
    staging.free()
    full.free()
    assert same_bits(got, whole).all()


def test_rccl_gather_path_with_a_real_communicator(oracle, cornell_world):
    """prosper_pt_comm_get_unique_id -> comm_init (ncclCommInitRank, one rank) -> render -> gather_tiles (ncclGather
    into the staging buffer on the communicator's stream + the de-interleave kernel) -> read_gathered."""
    from prosper_amd import capi
    w, h = 320, 200
    cam, fl = _camera(oracle, cornell_world, w, h)
    ctx = capi.Context(device=0)
    try:
        ctx.upload_scene(cornell_world)
        ctx.comm_init(capi.Context.comm_unique_id(), 0, 1)
        with pytest.raises(capi.ProsperPtError):
            ctx.comm_init(capi.Context.comm_unique_id(), 0, 1)  # one communicator per context
        pc = default_pc(S, fl, max_bounces=3, ibl=True)
        for k in range(3):  # frames in flight: the gather of frame k overlaps the path stages of frame k + 1
            p = default_pc(S, fl, frame_index=1 + 2 * k, max_bounces=3, ibl=True, skip_history=(k == 0))
            ctx.render(p, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
            ctx.gather_tiles(root=0)
        got = ctx.read_gathered()
        tile = ctx.read_hdr()
        assert same_bits(got, tile).all() and (got[..., 3] == 6).all()
        want = None
        osc = oracle.OracleScene(cornell_world, brute_force=True)
        for f in range(1, 7):
            want, _ = osc.render(default_pc(S, fl, frame_index=f, max_bounces=3, ibl=True, skip_history=(f == 1)), cam, w, h,
                                 history=want)
        assert same_bits(got, want).all()
        # a tile that does not match the communicator is refused
        ctx.render(pc, cam, w, h, tile=tiling.tile_for_rank(1, 2))
        with pytest.raises(capi.ProsperPtError):
            ctx.gather_tiles(root=0)
        ctx.comm_destroy()
    finally:
        ctx.close()


def test_tiled_rt_reference_host_class_single_rank(oracle, cornell_world):
    """render::TiledRtReference with one rank: record() = RtReference::record + gather (a copy), same state machine."""
    from prosper_amd.rt_reference import Camera, RtReference, TiledRtReference
    w, h = 160, 96
    tiled = TiledRtReference(0, 0, 1)
    plain = RtReference()
    plain.init(0)
    try:
        tiled.set_world(cornell_world)
        plain.set_world(cornell_world)
        cams = [Camera.from_world(cornell_world, w, h) for _ in range(2)]
        for pass_, cam in ((tiled, cams[0]), (plain, cams[1])):
            pass_.record(cam, w, h, RtReference.Options(ibl=True), frame_count=2)
            pass_.record(cam, w, h, RtReference.Options(ibl=True))
        a = tiled.context.read_gathered()
        b = plain.context.read_hdr()
        assert same_bits(a, b).all() and (a[..., 3] == 3).all()
    finally:
        tiled.close()
        plain.close()
