"""CPU experiment (oracle only): what a four-wide node that drops the intermediate boxes would cost in parity.

A four-wide record holding only the grandchildren's boxes never runs the reference's fp16 box test of the node in
between.  That test is not conservative, so skipping it finds hits the reference misses.  The oracle's
PO_FLAG_SKIP_ODD_INTERIOR walks the reference tree in reference order with exactly those tests forced to pass;
this script reports how many rays change their answer."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers, oracle
import pathtracer_amd as pt

for ntri in (60_000, 1_000_000):
    pos, aux, idx = pt.scenes.courtyard(ntri, seed=13)
    nodes, ridx, depth = pt.build_bvh(pos, idx, os.cpu_count() or 1)
    osc = oracle.Scene(pos, aux, ridx, nodes)
    rays = helpers.random_rays(nodes[0]["a"], nodes[0]["b"], 1 << 21, seed=3)
    ref = osc.trace(rays)
    alt = osc.trace(rays, flags=oracle.FLAG_REJECT_NAN | oracle.FLAG_SKIP_ODD_INTERIOR)
    same = (ref["i1"] == alt["i1"]) & (ref["i2"] == alt["i2"]) & (ref["i3"] == alt["i3"])
    print(f"{ntri} triangles, depth {depth}: {(~same).mean():.2e} of rays differ "
          f"(miss -> hit {((ref['i1'] < 0) & (alt['i1'] >= 0)).mean():.2e}, closer hit {((alt['t'] < ref['t']) & ~same).mean():.2e}, "
          f"farther hit {((alt['t'] > ref['t']) & ~same).mean():.2e})")
