#!/usr/bin/env python3
"""Algorithmic work per path of the benchmark configurations -> bench/work_model.json.

Renders a bounded sample of each configuration with the counting build of the CPU checker
(oracle/liboracle_count.so, -DPT_ORACLE_COUNT) and divides its counters by the paths traced.
Two kinds of entries:

* flat walks (cornell, random, final, simple_light): the reference's own evaluation -- flat world
  list, the reference's BVH with both children always visited, every accepted hit record completed;
* tree walks (random_lbvh, final_lbvh, trimesh_lbvh): the same scenes searched the way the product
  searches them, through the LBVH the library built (copied back from the GPU with
  pt_scene_lbvh_nodes and replayed on the CPU: near child first, pruned by the best distance where
  that is allowed), plus HBM-side bytes: 64 per node visited + 64 per leaf record tested.

Convention: one FLOP = one fp32 add/sub/mul/compare/min/max/div/sqrt, no FMA credit; libm calls
and integer operations are reported beside it, not inside it.

The flat entries need no GPU; the tree entries need one (the tree is built on the device) and are
kept from the existing file when none is visible.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # the repository root
sys.path.insert(0, ROOT)
import gpupathtracer_amd as pt                     # noqa: E402
from gpupathtracer_amd import abi                  # noqa: E402
from tests import _oracle_stats                    # noqa: E402

OUT = os.path.join(ROOT, "bench", "work_model.json")
# (key, scene, aspect, nx, ny, spp of the sample, note)
FLAT = [
    ("cornell", "cornell", 1.0, 256, 256, 64, "BASELINE configs[2]: Cornell box, depth 25"),
    ("random", "random", 16 / 9, 320, 180, 8, "BASELINE configs[1]: RTIOW, the reference's flat list of 487 spheres"),
    ("final", "final", 16 / 9, 320, 180, 8, "RTNW final scene, reference BVH over 400 boxes + media"),
    ("simple_light", "simple_light", 2.0, 200, 100, 16, "two noise spheres, sphere + rectangle lights"),
]


def entry(c, note, sample):
    paths = c["paths"]
    e = {"note": note, "sample": sample,
         "flop_per_path": c["flop"] / paths, "libm_calls_per_path": c["libm"] / paths,
         "integer_ops_per_path": c["iop"] / paths, "rng_draws_per_path": c["draws"] / paths,
         "closest_hit_searches_per_path": c["world_hits"] / paths, "rect_tests_per_path": c["rect_tests"] / paths,
         "sphere_tests_per_path": c["sphere_tests"] / paths, "triangle_tests_per_path": c["triangle_tests"] / paths,
         "aabb_tests_per_path": c["aabb_tests"] / paths, "diffuse_scatters_per_path": c["diffuse_scatters"] / paths,
         "accepted_hits_per_path": c["accepted_hits"] / paths}
    if c["tree_nodes"]:
        e["tree_nodes_visited_per_path"] = c["tree_nodes"] / paths
        e["tree_leaves_tested_per_path"] = c["tree_leaves"] / paths
        e["tree_nodes_visited_per_search"] = c["tree_nodes"] / c["world_hits"]
        e["tree_leaves_tested_per_search"] = c["tree_leaves"] / c["world_hits"]
        e["bytes_per_path"] = 64.0 * (c["tree_nodes"] + c["tree_leaves"]) / paths
    return e


def tree_configs():
    sys.path.insert(0, os.path.join(ROOT, "bench"))
    import configs as bench_configs
    out = []
    d = pt.builtin_scene("random", 16 / 9, accel=abi.ACCEL_LBVH_BINARY)
    out.append(("random_lbvh", d, d.c.n_world, 1, 0, 320, 180, 8, "BASELINE configs[1] as the product runs it: RTIOW, 487 spheres in the library's LBVH"))
    d = pt.builtin_scene("final", 16 / 9, earth_rgb=bench_configs.synthetic_earthmap(), accel=abi.ACCEL_LBVH_BINARY)
    prefix = next(i for i in range(d.c.n_world) if d.c.world[i].type == abi.OBJ_MEDIUM)
    out.append(("final_lbvh", d, prefix, 1, 0, 320, 180, 8, "BASELINE configs[3] as the product runs it: RTNW final with a 512x256 earth map; the 400 boxes of the "
                "reference BVH and the shapes before the first medium in the library's LBVH, the media and what follows walked in order"))
    d = pt.builtin_scene("final", 16 / 9, accel=abi.ACCEL_LBVH_BINARY)
    prefix = next(i for i in range(d.c.n_world) if d.c.world[i].type == abi.OBJ_MEDIUM)
    out.append(("final_lbvh_no_earth", d, prefix, 1, 0, 320, 180, 8, "RTNW final as the reference's device branch builds it (no earth sphere), through the library's LBVH"))
    d = pt.trimesh_scene(1000, 16 / 9)
    d.c.accel = abi.ACCEL_LBVH_BINARY
    out.append(("trimesh_lbvh", d, d.c.n_world, 0, 1, 480, 270, 1, "BASELINE configs[4]: 2 000 000 triangles + Cornell enclosure in the library's LBVH "
                "(no pruning by distance: the reference's triangle t is not one); the replay's own hits drive the paths"))
    return out


def main():
    old = {}
    if os.path.exists(OUT):
        old = json.load(open(OUT)).get("configs", {})
    out = {"convention": "FLOP = fp32 add/sub/mul/compare/min/max/div/sqrt as the reference evaluates them, no FMA credit; libm and integer "
                         "ops separate.  Flat entries: the reference's own walk (flat lists, its BVH with both children always visited).  "
                         "*_lbvh entries: the search priced as a replay of the library's exported LBVH (27 FLOP per box test), "
                         "bytes_per_path = 64 per node visited + 64 per leaf record tested; *_lbvh_wide: the same tree walked in its "
                         "4-wide form (every second level folded away), 64 bytes per wide node",
           "generator": "tests/make_work_model.py (oracle/pt_oracle.c built with -DPT_ORACLE_COUNT)", "configs": {}}
    for key, scene, aspect, nx, ny, ns, note in FLAT:
        desc = pt.builtin_scene(scene, aspect)
        c, _ = _oracle_stats.counters(desc, nx, ny, ns)
        assert c["paths"] == nx * ny * ns
        out["configs"][key] = entry(c, note, "%dx%d at %d spp, depth 25" % (nx, ny, ns))
        print(key, json.dumps(out["configs"][key]), flush=True)
    if pt.device_count() > 0:
        pt.set_device(0)
        for key, desc, prefix, cull, trust, nx, ny, ns, note in tree_configs():
            sc = pt.Scene(desc)
            nodes = sc.lbvh_nodes()
            assert len(nodes) > 0
            c, _ = _oracle_stats.counters(desc, nx, ny, ns, threads=16, tree=(nodes, 0, cull, prefix, trust))
            out["configs"][key] = entry(c, note, "%dx%d at %d spp, depth 25" % (nx, ny, ns))
            print(key, json.dumps(out["configs"][key]), flush=True)
            # the same search through the 4-wide form the product walks by default: its own node visits and bytes
            c, _ = _oracle_stats.counters(desc, nx, ny, ns, threads=16, tree=(nodes, 0, cull, prefix, trust), wide=True)
            out["configs"][key + "_wide"] = entry(c, note + " -- priced as the 4-wide walk (64-byte nodes of up to four child boxes)",
                                                  "%dx%d at %d spp, depth 25" % (nx, ny, ns))
            print(key + "_wide", json.dumps(out["configs"][key + "_wide"]), flush=True)
    else:
        for key in ("random_lbvh", "final_lbvh", "final_lbvh_no_earth", "trimesh_lbvh",
                    "random_lbvh_wide", "final_lbvh_wide", "final_lbvh_no_earth_wide", "trimesh_lbvh_wide"):
            if key in old:
                out["configs"][key] = old[key]
                print(key, "(kept: no GPU to build the tree on)", flush=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
