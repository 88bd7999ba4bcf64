"""Ad-hoc parity check of the HIP path against the CPU oracle (development aid)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as ra
import oracle_lib as ol

def check(name, scale, depth, spheres=False, seed=0, threads=16):
    t0 = time.time()
    if spheres:
        g = ra.Scene.generate_random_spheres(scale, seed); o = ol.OracleScene.spheres(scale, seed)
    else:
        g = ra.Scene.from_file(f"{ROOT}/scenes/{name}.yaml", scale, seed); o = ol.OracleScene.from_file(f"{ROOT}/scenes/{name}.yaml", scale, seed)
    seeds = g.row_seeds(); oseeds = o.row_seeds()
    assert (seeds == oseeds).all()
    trav = int(os.environ.get("RT_TRAVERSAL", "0"))
    if trav:
        g.set_option(ra.OPT_TRAVERSAL, trav)
        name = f"{name}[traversal {trav}]"
    t1 = time.time()
    canvas, st = g.render(depth, seeds)
    t2 = time.time()
    canvas, st = g.render(depth, seeds)
    t3 = time.time()
    ref = o.render_rows(depth, oseeds, threads=threads)
    t4 = time.time()
    ref8 = ol.to_u8(ref)
    bad8 = int((canvas.rgb8 != ref8).any(axis=2).sum())
    den = np.maximum(np.abs(ref), 1e-300)
    rel = np.abs(canvas.pixels - ref) / den
    rel[ref == canvas.pixels] = 0
    exact = int((canvas.pixels == ref).all(axis=2).sum())
    print(f"{name} {g.width}x{g.height} depth {depth}: u8 mismatches {bad8}, f64 bit-exact px {exact}/{g.width*g.height}, max rel {rel.max():.3e}; "
          f"gpu {st['ms_total']:.2f} ms (trace {st['ms_trace']:.2f} shadow {st['ms_shadow']:.2f} shade {st['ms_shade']:.2f}, first call {1e3*(t2-t1):.0f} ms wall, second {1e3*(t3-t2):.0f} ms), "
          f"oracle {t4-t3:.2f}s x{threads}thr; rays p {st['primary_rays']} s {st['secondary_rays']} sh {st['shadow_rays']} levels {st['levels']} chunks {st['chunks']} "
          f"oracle counters {o.counters()}", flush=True)
    return bad8

if __name__ == "__main__":
    ra.init(0)
    which = sys.argv[1:] or ["c6"]
    if "c6" in which: check("chapter6", 2.0, 5)
    if "c11s" in which: check("chapter11", 1.0, 30)
    if "c8" in which: check("chapter8", 1.0, 5)
    if "c9" in which: check("chapter9", 1.0, 5)
    if "c12" in which: check("chapter12", 1.0, 5)
    if "c13" in which: check("chapter13", 1.0, 5)
    if "c14" in which: check("chapter14", 1.0, 5)
    if "c16" in which: check("chapter16", 1.0, 5)
    if "water" in which: check("chapter11-water", 1.0, 5)
    if "spheres" in which: check("spheres", 1.0, 20, spheres=True)
    if "c15" in which: check("chapter15", 0.5, 5)
    if "area" in which: check("area-light", 0.5, 5)
    if "dragon" in which: check("chapter15-dragon", 0.5, 5)
    if "bbox" in which: check("bounding-box", 0.5, 5)
