"""The one place the oracle (and the kernels) depart from the TEXT of res/shader/rt/reference/main.rgen:241-283: a path
whose throughput is exactly zero ends.  Measured here, not argued: the oracle's literal mode runs
the loop as written; per scene x {clampIndirect on, off} x {IBL on, off} every pixel that is finite in the literal image
must be bit-equal with the rule's, and the number of pixels where the literal image is NOT finite (0 * inf, NaN ray
directions - undefined behaviour in the GLSL) equals the committed count (tests/golden/literal_glsl_counts.json)."""
import json
import os

import numpy as np
import pytest

import literal_glsl_cases as L
from conftest import same_bits
from prosper_amd import structs as S

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "literal_glsl_counts.json")) as f:
    COUNTS = json.load(f)["cases"]

_worlds = {}


def _world(scene):
    if scene not in _worlds:
        _worlds[scene] = L.build_world(scene)
    return _worlds[scene]


@pytest.mark.parametrize("case", L.CASES, ids=L.case_id)
def test_literal_oracle_differs_from_the_rule_only_where_it_is_not_finite(oracle, case):
    scene, clamp, ibl = case
    world = _world(scene)
    lit1, lit = L.render_oracle(oracle, world, scene, clamp, ibl, True)
    rule1, rule = L.render_oracle(oracle, world, scene, clamp, ibl, False)
    want = COUNTS[L.case_id(case)]
    n1, d1 = L.compare(lit1, rule1)
    n, d = L.compare(lit, rule)
    assert d1 == 0 and d == 0, "finite literal pixels differ from the rule's: %d / %d" % (d1, d)
    assert n1 == want["nonfinite_after_1_frame"] and n == want["nonfinite_after_%d_frames" % L.FRAMES]
    assert np.isfinite(rule).all()
    if clamp or not ibl:
        # the reference's default (clampIndirect on) scrubs every NaN the literal loop produces: identical images
        assert n == 0


def test_the_only_nonfinite_case_is_clamp_off_with_ibl():
    """The committed measurement itself: which cases have pixels the GLSL as written leaves undefined."""
    bad = sorted(k for k, v in COUNTS.items() if v["nonfinite_after_%d_frames" % L.FRAMES])
    assert bad == ["cornell-clamp_off-ibl_on"]
    assert all(v["finite_but_different_after_%d_frames" % L.FRAMES] == 0 for v in COUNTS.values())


@pytest.mark.gpu
@pytest.mark.parametrize("case", L.CASES, ids=L.case_id)
def test_hip_path_equals_the_literal_glsl_wherever_it_is_finite(gpu_ctx, oracle, case):
    """The HIP path (which implements the rule) against the LITERAL oracle: bit-equal on every pixel the literal loop
    leaves finite, the committed number of pixels elsewhere; and bit-equal with the rule oracle everywhere."""
    scene, clamp, ibl = case
    world = _world(scene)
    _, w, h, mb = L.SCENES[scene]
    c = world.camera
    cam, focal = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
    gpu_ctx.upload_scene(world)
    for pc in L.pcs(S, focal, clamp, ibl, mb):
        gpu_ctx.render(pc, cam, w, h)
    got = gpu_ctx.read_hdr()
    _, lit = L.render_oracle(oracle, world, scene, clamp, ibl, True)
    _, rule = L.render_oracle(oracle, world, scene, clamp, ibl, False)
    assert same_bits(got, rule).all()
    n, d = L.compare(lit, got)
    assert d == 0 and n == COUNTS[L.case_id(case)]["nonfinite_after_%d_frames" % L.FRAMES]
