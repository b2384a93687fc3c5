"""`.prosper_mesh` v4 cache reader/writer (DeferredLoadingContext.cpp:505-802) round trip."""
import struct

import numpy as np
import pytest

from prosper_amd import mesh_cache, scenes, structs as S
from prosper_amd.world import World


def _mesh(n_quads, seed):
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(n_quads):
        c = rng.normal(size=3)
        parts.append(scenes.quad(c, c + (1, 0, 0), c + (1, 1, 0.2), c + (0, 1, 0.2), uv_scale=2.0))
    return scenes.merge(parts)


@pytest.mark.parametrize("with_tangents_uvs", [True, False])
def test_cache_round_trip_gives_the_same_geometry_buffer(tmp_path, with_tangents_uvs):
    p, n, t, uv, idx = _mesh(5, 1)
    direct = World()
    direct.add_mesh(p, idx, 0, normals=n, tangents=t if with_tangents_uvs else None, uvs=uv if with_tangents_uvs else None)
    header, blob = mesh_cache.pack_cache(p, idx, n, t if with_tangents_uvs else None, uv if with_tangents_uvs else None)
    path = tmp_path / "0.prosper_mesh"
    mesh_cache.write_mesh_cache(str(path), header, blob, source_write_time=123456789)
    raw = path.read_bytes()
    assert struct.unpack_from("<QI", raw) == (0x48534D5250535250, 4) and raw[:8] == b"PRSPRMSH"
    header2, blob2 = mesh_cache.read_mesh_cache(str(path))
    assert header2["sourceWriteTime"] == 123456789 and header2["blobByteCount"] == blob.size * 4
    assert header2["tangentsOffset"] == (header["tangentsOffset"] if with_tangents_uvs else mesh_cache.ABSENT)
    cached = World()
    mesh_cache.add_cached_mesh(cached, header2, blob2, 0)
    a, b = direct.freeze(), cached.freeze()
    assert np.array_equal(a["geometry_buffers"][0], b["geometry_buffers"][0])
    ma, mb = direct.metadatas[0], cached.metadatas[0]
    for name, _ in S.GeometryMetadata._fields_:
        if not name.startswith("meshlet"):
            assert getattr(ma, name) == getattr(mb, name), name
    assert (direct.mesh_infos[0].vertexCount, direct.mesh_infos[0].indexCount) == (20, 30)
    assert (cached.mesh_infos[0].vertexCount, cached.mesh_infos[0].indexCount) == (20, 30)


def test_second_blob_lands_after_the_first_and_renders_identically(oracle, tmp_path):
    from conftest import default_pc
    worlds = []
    for use_cache in (False, True):
        w = World()
        mat = w.add_material(base_color=(0.8, 0.6, 0.4, 1.0), metallic=0.0, roughness=0.7)
        for k in range(2):
            p, n, t, uv, idx = _mesh(4, 10 + k)
            if use_cache:
                path = tmp_path / ("%d.prosper_mesh" % k)
                mesh_cache.write_mesh_cache(str(path), *mesh_cache.pack_cache(p, idx, n, t, uv))
                mi = mesh_cache.add_cached_mesh(w, *mesh_cache.read_mesh_cache(str(path)), mat)
            else:
                mi = w.add_mesh(p, idx, mat, normals=n, tangents=t, uvs=uv)
            w.add_instance(w.add_model([(mi, mat)]))
        w.camera = dict(eye=(0.5, 0.5, 6.0), target=(0.5, 0.5, 0.0), up=(0, 1, 0), fov=0.9, zN=0.1, zF=100.0)
        worlds.append(w)
    assert worlds[1].metadatas[1].positionsOffset > worlds[1].metadatas[0].texCoord0sOffset
    c = worlds[0].camera
    cam, fl = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], 48, 32)
    imgs = [oracle.OracleScene(w, brute_force=True).render(default_pc(S, fl, max_bounces=2), cam, 48, 32)[0] for w in worlds]
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32)) and imgs[0][..., :3].max() > 0


def test_rejects_foreign_or_stale_files(tmp_path):
    header, blob = mesh_cache.pack_cache(*[_mesh(1, 3)[i] for i in (0, 4, 1)])
    path = tmp_path / "m.prosper_mesh"
    mesh_cache.write_mesh_cache(str(path), header, blob)
    raw = bytearray(path.read_bytes())
    bad = bytearray(raw)
    bad[0] ^= 1
    (tmp_path / "magic").write_bytes(bytes(bad))
    with pytest.raises(mesh_cache.MeshCacheError, match="magic"):
        mesh_cache.read_mesh_cache(str(tmp_path / "magic"))
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 8, 3)
    (tmp_path / "v3").write_bytes(bytes(bad))
    with pytest.raises(mesh_cache.MeshCacheError, match="version"):
        mesh_cache.read_mesh_cache(str(tmp_path / "v3"))
    (tmp_path / "short").write_bytes(bytes(raw[:-8]))
    with pytest.raises(mesh_cache.MeshCacheError, match="blob size"):
        mesh_cache.read_mesh_cache(str(tmp_path / "short"))
