#!/usr/bin/env python3
"""Writes tests/golden/tiny_scene.gltf: a hand-made glTF that exercises the ingest conventions
(prosper_amd/gltf.py) - two meshes / three primitives, u16 and u32 indices, an interleaved vertex
buffer, a normalised-u16 TEXCOORD_0, a primitive without TANGENT/TEXCOORD_0, an embedded PNG, a
non-default sampler, MASK and BLEND materials, a node tree whose LIFO traversal order differs from its
storage order, TRS components inside the 1e-3 "identity" threshold, a matrix node, KHR_lights_punctual
sun/point/spot lights and a camera.  Everything is authored here; nothing comes from the reference."""
import base64
import json
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def png_rgba(px):
    h, w, _ = px.shape
    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def main():
    buf = bytearray()
    views, accessors = [], []

    def add_view(data, stride=None):
        while len(buf) % 4:
            buf.append(0)
        v = {"buffer": 0, "byteOffset": len(buf), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        buf.extend(data)
        views.append(v)
        return len(views) - 1

    def add_accessor(view, ctype, count, kind, offset=0, normalized=False, **extra):
        a = {"bufferView": view, "componentType": ctype, "count": count, "type": kind}
        if offset:
            a["byteOffset"] = offset
        if normalized:
            a["normalized"] = True
        a.update(extra)
        accessors.append(a)
        return len(accessors) - 1

    # --- mesh 0, primitive 0: textured quad, interleaved POSITION|NORMAL (stride 24), TANGENT, u16-normalised uv ---
    pos = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (4, 1))
    inter = np.concatenate([pos, nrm], axis=1).astype(np.float32)
    v_inter = add_view(inter.tobytes(), stride=24)
    a_pos0 = add_accessor(v_inter, 5126, 4, "VEC3", min=[-1, -1, 0], max=[1, 1, 0])
    a_nrm0 = add_accessor(v_inter, 5126, 4, "VEC3", offset=12)
    tan = np.tile(np.array([1, 0, 0, 1], np.float32), (4, 1))
    a_tan0 = add_accessor(add_view(tan.tobytes()), 5126, 4, "VEC4")
    uv16 = np.array([[0, 0], [65535, 0], [65535, 65535], [0, 65535]], np.uint16)
    a_uv0 = add_accessor(add_view(uv16.tobytes()), 5123, 4, "VEC2", normalized=True)
    a_idx0 = add_accessor(add_view(np.array([0, 1, 2, 0, 2, 3], np.uint16).tobytes()), 5123, 6, "SCALAR")
    # --- mesh 0, primitive 1: one triangle, no TANGENT / TEXCOORD_0, u32 indices ---
    pos1 = np.array([[0, 0, 0.5], [0.5, 0, 0.5], [0, 0.5, 0.5]], np.float32)
    a_pos1 = add_accessor(add_view(pos1.tobytes()), 5126, 3, "VEC3", min=[0, 0, 0.5], max=[0.5, 0.5, 0.5])
    a_nrm1 = add_accessor(add_view(np.tile(np.array([0, 0, 1], np.float32), (3, 1)).tobytes()), 5126, 3, "VEC3")
    a_idx1 = add_accessor(add_view(np.array([0, 1, 2], np.uint32).tobytes()), 5125, 3, "SCALAR")
    # --- mesh 1: a floor quad ---
    pos2 = np.array([[-3, -1, 3], [3, -1, 3], [3, -1, -3], [-3, -1, -3]], np.float32)
    a_pos2 = add_accessor(add_view(pos2.tobytes()), 5126, 4, "VEC3", min=[-3, -1, -3], max=[3, -1, 3])
    a_nrm2 = add_accessor(add_view(np.tile(np.array([0, 1, 0], np.float32), (4, 1)).tobytes()), 5126, 4, "VEC3")
    a_idx2 = add_accessor(add_view(np.array([0, 1, 2, 0, 2, 3], np.uint8).tobytes()), 5121, 6, "SCALAR")

    rng = np.random.default_rng(7)
    tex = rng.integers(0, 256, size=(4, 4, 4), dtype=np.uint8)
    tex[..., 3] = np.where((np.arange(4)[:, None] + np.arange(4)[None, :]) % 2 == 0, 255, 40)
    png = png_rgba(tex)

    doc = {
        "asset": {"version": "2.0", "generator": "tests/golden/make_tiny_gltf.py"},
        "extensionsUsed": ["KHR_lights_punctual"],
        "extensions": {"KHR_lights_punctual": {"lights": [
            {"type": "point", "color": [1.0, 0.5, 0.25], "intensity": 50.0},
            {"type": "spot", "color": [1.0, 1.0, 1.0], "intensity": 80.0, "range": 12.0,
             "spot": {"innerConeAngle": 0.3, "outerConeAngle": 0.6}},
            {"type": "directional", "color": [1.0, 0.9, 0.8], "intensity": 3.0},
        ]}},
        "buffers": [{"byteLength": 0, "uri": ""}],
        "bufferViews": views,
        "accessors": accessors,
        "images": [{"uri": "data:image/png;base64," + base64.b64encode(png).decode("ascii")}],
        "samplers": [{"magFilter": 9728, "minFilter": 9986, "wrapS": 33648, "wrapT": 33071}],
        "textures": [{"source": 0, "sampler": 0}, {"source": 0}],
        "materials": [
            {"name": "masked", "alphaMode": "MASK", "alphaCutoff": 0.4,
             "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "baseColorFactor": [0.9, 0.8, 0.7, 1.0],
                                      "metallicFactor": 0.0, "roughnessFactor": 0.6}},
            {"name": "blend", "alphaMode": "BLEND",
             "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.9, 0.5], "metallicRoughnessTexture": {"index": 1}},
             "normalTexture": {"index": 1}},
            {"name": "floor", "pbrMetallicRoughness": {"baseColorFactor": [0.7, 0.7, 0.7, 1.0], "metallicFactor": 0.0}},
        ],
        "meshes": [
            {"primitives": [
                {"attributes": {"POSITION": a_pos0, "NORMAL": a_nrm0, "TANGENT": a_tan0, "TEXCOORD_0": a_uv0},
                 "indices": a_idx0, "material": 0},
                {"attributes": {"POSITION": a_pos1, "NORMAL": a_nrm1}, "indices": a_idx1, "material": 1}]},
            {"primitives": [{"attributes": {"POSITION": a_pos2, "NORMAL": a_nrm2}, "indices": a_idx2, "material": 2}]},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.05, "zfar": 50.0}}],
        "nodes": [
            {"name": "root", "children": [1, 2, 3, 4], "translation": [0.0004, -0.0002, 0.0]},   # inside the threshold
            {"name": "quads", "mesh": 0, "translation": [0.5, 0.25, 0.0], "scale": [1.0005, 0.9996, 1.0]},
            {"name": "floor", "mesh": 1, "children": [5]},
            {"name": "camera", "camera": 0, "translation": [0.0, 0.5, 4.0]},
            {"name": "lamp", "translation": [0.0, 2.5, 0.5], "extensions": {"KHR_lights_punctual": {"light": 0}}},
            {"name": "second quads", "mesh": 0,
             "matrix": [0.0, 0.0, -2.0, 0.0, 0.0, 2.0, 0.0, 0.0, 2.0, 0.0, 0.0, 0.0, -2.0, 0.0, -1.0, 1.0]},
            {"name": "spot", "translation": [1.0, 2.0, 2.0], "rotation": [-0.5, 0.0, 0.0, 0.8660254],
             "extensions": {"KHR_lights_punctual": {"light": 1}}},
            {"name": "sun", "rotation": [-0.3826834, 0.0, 0.0, 0.9238795],
             "extensions": {"KHR_lights_punctual": {"light": 2}}},
        ],
        "scenes": [{"nodes": [0, 6, 7]}],
        "scene": 0,
    }
    doc["buffers"][0] = {"byteLength": len(buf), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(buf)).decode("ascii")}
    with open(os.path.join(HERE, "tiny_scene.gltf"), "w") as f:
        json.dump(doc, f, indent=1)
    np.save(os.path.join(HERE, "tiny_scene_texture.npy"), tex)
    print("wrote tiny_scene.gltf (%d buffer bytes)" % len(buf))


if __name__ == "__main__":
    main()
