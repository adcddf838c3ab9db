#!/usr/bin/env python3
"""bf_scan_nearest (HIP) against the reference's closest-point search restated in its own float32 arithmetic
(oracle/nearest_ref.c: mesh_grid_kernel.cu:12-109, 239-353 + matrix.h).  GPU box tool, one JSON object per data set.

For every query: face id equal?  If not: does the reference's rule give the HIP face the SAME distance bit for bit (an exact
tie - the reference's own answer then depends on the order an atomicCAS race left in its cell lists) or a larger one (a
decision that differs: reported with its relative gap)?  Plus max |point - point_ref| and |coeff - coeff_ref| over the queries
whose faces agree.  Every figure is given against both builds of the restatement (source order / multiply-adds fused) and
between those two builds - the reference's own latitude under its compiler.

usage: python tools/measure_nearest_parity.py [--quick] [--out FILE]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402
from oracle import mesh_oracle as MO, nearest_ref as NR, adversarial as ADV   # noqa: E402


def compare(name, verts, faces, queries, got, ref, other_label):
    """got / ref = (face, point, coeff); -> dict"""
    gf, gp, gc = got
    rf, rp, rc, rd = ref[:4]
    diff = np.nonzero(gf != rf)[0]
    out = {"vs": other_label, "queries": int(len(rf)), "face_mismatch": int(len(diff))}
    if len(diff):
        # the reference's rule on the face the other side chose
        _, d_other, _ = NR.rule(verts, faces, np.clip(gf[diff], 0, len(faces) - 1), queries[diff])
        tie = d_other.view(np.uint32) == rd[diff].view(np.uint32)
        gap = (d_other.astype(np.float64) - rd[diff]) / np.maximum(rd[diff].astype(np.float64), 1e-30)
        out["exact_ties"] = int(tie.sum())
        out["decisions_that_differ"] = int((~tie).sum())
        if (~tie).any():
            g = gap[~tie]
            out["gap_rel_median"] = float(np.median(g)); out["gap_rel_max"] = float(g.max())
            out["gap_negative"] = int((g < 0).sum())        # the other side's face is CLOSER by the reference's own rule (never for an argmin of it)
            out["point_distance_between_answers_max"] = float(np.abs(gp[diff][~tie] - rp[diff][~tie]).max())
    same = gf == rf
    if same.any():
        out["same_face_point_maxabs"] = float(np.abs(gp[same] - rp[same]).max())
        out["same_face_coeff_maxabs"] = float(np.abs(gc[same] - rc[same]).max())
        out["same_face_point_bitequal"] = float(np.mean((gp[same].view(np.uint32) == rp[same].view(np.uint32)).all(1)))
        out["same_face_coeff_bitequal"] = float(np.mean((gc[same].view(np.uint32) == rc[same].view(np.uint32)).all(1)))
    return out


def run_set(name, verts, faces, queries, allfaces=False):
    verts = np.ascontiguousarray(verts, np.float32); faces = np.ascontiguousarray(faces, np.int32)
    queries = np.ascontiguousarray(queries, np.float32)
    scan = N.Scan(verts, faces)
    pts, ids, bary = scan.nearest_points(queries)
    dims, origin, step = scan.grid_info()
    tri_num, tri_idx = scan.grid_lists()
    scan.close()
    grid = (step, dims, origin, tri_num, tri_idx)
    t0 = time.perf_counter()
    strict = NR.search_nearest(verts, faces, queries, grid, stats=True)
    t_ref = time.perf_counter() - t0
    fused = NR.search_nearest(verts, faces, queries, grid, fused=True)
    rec = {"set": name, "triangles": int(len(faces)), "queries": int(len(queries)), "oracle_seconds": round(t_ref, 3),
           "oracle_rule_evaluations_per_query": float(strict[4][0] / max(1, len(queries))),
           "hip_vs_source_order": compare(name, verts, faces, queries, (ids, pts, bary), strict, "source order"),
           "hip_vs_fused": compare(name, verts, faces, queries, (ids, pts, bary), fused, "fused multiply-adds"),
           "fused_vs_source_order": compare(name, verts, faces, queries, (fused[0], fused[1], fused[2]), strict, "oracle, fused, against source order")}
    if allfaces:       # the walk against the rule over all faces: must agree wherever there is no exact tie
        af = NR.nearest_allfaces(verts, faces, queries)
        d = np.nonzero(af[0] != strict[0])[0]
        rec["walk_vs_allfaces"] = {"face_mismatch": int(len(d)), "of_which_exact_ties": int((af[3][d].view(np.uint32) == strict[3][d].view(np.uint32)).sum())}
    print(json.dumps(rec), flush=True)
    return rec


def sliver_soup(scale, seed=0, n=600):
    """needles, slivers, obtuse and regular triangles with edges around `scale` metres, scattered in a 40-scale box, + queries in all
    Voronoi regions at distances 0.01 .. 3 scale (oracle/adversarial.py scaled)"""
    d = ADV.soup(seed=seed, min_margin=0.0)            # (no query dropped for being close to a branch decision: those are the point here)
    return (d["verts"].astype(np.float64) * scale).astype(np.float32), d["faces"], (d["queries"].astype(np.float64) * scale).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    recs = []
    # (1) config 5 as stated: SMPL-X scans, the body's vertices as queries at the distances the fit sees
    model = S.make_model("smplx", seed=0)
    frames = 1 if a.quick else 8
    for f in range(frames):
        _, sv, sf = S.make_scan_problem_smplx(model, frame=f, n_views=2)
        nv = model["v_template"].shape[0]
        rng = np.random.default_rng(100 + f)
        for label, sigma in (("1cm", 0.01), ("1mm", 0.001), ("5cm", 0.05)):
            q = sv[:nv] + rng.normal(0.0, sigma, (nv, 3))
            recs.append(run_set("cfg5 frame %d, body vertices + N(0, %s)" % (f, label), sv, sf, q, allfaces=(f == 0 and label == "1cm" and not a.quick)))
    # (2) adversarial soup at 1 mm .. 10 cm edges (where the reference's absolute 1e-9 rank tests bite) and at the original size
    for scale in (1.0, 0.1, 0.01, 0.001):
        v, fcs, q = sliver_soup(scale)
        recs.append(run_set("adversarial soup, scale %g" % scale, v, fcs, q, allfaces=True))
    tot = {"set": "TOTAL"}
    for key in ("hip_vs_source_order", "hip_vs_fused", "fused_vs_source_order"):
        tot[key] = {k: int(sum(r[key].get(k, 0) for r in recs)) for k in ("queries", "face_mismatch", "exact_ties", "decisions_that_differ")}
    print(json.dumps(tot), flush=True)
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(recs + [tot], fh, indent=1)


if __name__ == "__main__":
    main()
