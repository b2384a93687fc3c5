#!/usr/bin/env python3
"""Writes tests/golden/literal_glsl_counts.json: per case of tests/literal_glsl_cases.py, the number of pixels where
the oracle's LITERAL restatement of main.rgen:241-283 (no zero-throughput rule) is not finite after 1 and after
FRAMES accumulated frames, and the number of finite literal pixels whose bits differ from the rule's (must be 0).

    python tests/golden/make_literal_counts.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import literal_glsl_cases as L  # noqa: E402
from oracle import binding as oracle  # noqa: E402


def main():
    out = {"note": "pixels of a %s-frame accumulation; literal = ora_scene_set_literal_glsl(1); 'nonfinite' = some channel "
                   "of the literal image is NaN/inf; 'finite_but_different' = finite in the literal image and not "
                   "bit-equal to the zero-throughput-rule image" % L.FRAMES, "cases": {}}
    worlds = {}
    for case in L.CASES:
        scene, clamp, ibl = case
        world = worlds.setdefault(scene, L.build_world(scene))
        lit1, lit = L.render_oracle(oracle, world, scene, clamp, ibl, True)
        rule1, rule = L.render_oracle(oracle, world, scene, clamp, ibl, False)
        n1, d1 = L.compare(lit1, rule1)
        n, d = L.compare(lit, rule)
        out["cases"][L.case_id(case)] = {
            "pixels": int(lit.shape[0] * lit.shape[1]), "nonfinite_after_1_frame": n1, "nonfinite_after_%d_frames" % L.FRAMES: n,
            "finite_but_different_after_1_frame": d1, "finite_but_different_after_%d_frames" % L.FRAMES: d,
            "rule_image_nonfinite": int((~__import__("numpy").isfinite(rule).all(axis=2)).sum())}
        print(L.case_id(case), out["cases"][L.case_id(case)])
    with open(os.path.join(HERE, "literal_glsl_counts.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
