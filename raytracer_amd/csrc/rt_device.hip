// MI355X (gfx950) rendering core: the device side of include/rt_api.h.
//
// One launch per recursion level.  A level kernel takes one ray per lane
// (level 0 builds it from the pixel, deeper levels read a ray queue), walks
// the flattened object tree as a wavefront (the node index is wave-uniform,
// node and leaf records arrive through scalar loads, each lane carries a
// "skip until" index for the subtrees its own bounding-box tests rejected),
// keeps the closest hit in registers, shades it (shadow rays walk the same
// tree), writes one 64-byte node record and appends the reflected / refracted
// rays to the next level's queue with one atomic per wave.  After the last
// level a bottom-up pass folds the records into colours with exactly the
// association order of the recursive reference:
//   surface + reflected * reflectance + refracted * (1 - reflectance).
//
// All arithmetic is f64 with contraction off; expressions keep the operand
// order of the reference (cited per function, paths relative to the
// reference's src/).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_api.h"

namespace {

// ---------------------------------------------------------------- basics
constexpr double EPS = 2.220446049250313e-16;  // f64::EPSILON
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int CSG_CAP = 32;        // intersections buffered inside one outermost csg
constexpr int CSG_DEPTH = 6;       // nested csg levels
constexpr int CONT_CAP = 32;       // intersections at or before a refractive hit
constexpr int PAT_DEPTH = 6;       // nested pattern levels
constexpr int WAVE = 64;

#define DEV __device__ __forceinline__
#define CADDR __attribute__((address_space(4)))

// Loads through the constant address space: with a wave-uniform index the
// compiler emits scalar loads, so node / leaf data sits in SGPRs.
DEV double cld(const double* p, uint32_t i) { return ((const CADDR double*)(uintptr_t)p)[i]; }
DEV uint32_t clu(const uint32_t* p, uint32_t i) { return ((const CADDR uint32_t*)(uintptr_t)p)[i]; }

struct V3 { double x, y, z; };
DEV V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
DEV V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
DEV V3 operator/(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
DEV V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // math/vector.rs:55-57
DEV V3 cross(V3 a, V3 b) {                                                 // math/vector.rs:60-66
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DEV bool approx0(double x) { return fabs(x) <= EPS; }  // approx_eq!(x, 0.0), math/float.rs:10-24
// float-cmp 0.9.0 ApproxEq for f64, default margin (epsilon f64::EPSILON, ulps 4)
DEV bool approx_eq(double a, double b) {
    if (a == b) return true;
    if (fabs(a - b) <= EPS) return true;
    long long d = (long long)((unsigned long long)__double_as_longlong(a) - (unsigned long long)__double_as_longlong(b));
    if (d == (long long)0x8000000000000000ULL) return false;
    if (d < 0) d = -d;
    return d <= 4;
}
DEV V3 normalise(V3 a) {  // math/vector.rs:44-52
    double m = sqrt(dot(a, a));
    if (approx0(m)) return mk(0.0, 0.0, 0.0);
    return a / m;
}
DEV V3 reflect(V3 v, V3 n) { return v - (n * 2.0) * dot(v, n); }  // math/vector.rs:69-71

struct Ray { V3 o, d; };

// ---------------------------------------------------------------- device scene
struct DevScene {
    uint32_t n_nodes, n_leaves, n_lights, class_words;
    const uint32_t* node_meta;  // n_nodes*4: kind, skip, arg, 0
    const double* node_bbox;    // n_nodes*6
    const double* leaf_inv;     // n_leaves*12
    const double* leaf_prm;     // n_leaves*4: minimum, maximum, closed, 0
    const uint32_t* leaf_meta;  // n_leaves*4: shape, tri, material, class | casts_shadow<<31
    const double* tri;          // n_tris*18
    const uint32_t* csg_op;
    const uint64_t* csg_bits;
    const double* mat;          // n_materials*8: 7 scalars, pattern root (as integer value)
    const uint32_t* pat_meta;   // n_patterns*4: kind, a, b, 0
    const double* pat_inv;      // n_patterns*12
    const double* pat_col;      // n_patterns*4
    const double* light;        // n_lights*12
    const uint32_t* light_meta; // n_lights*4: kind, usteps, vsteps, 0
};

// Algorithmic FLOP constants (f64 add/sub/mul/div/sqrt/compare in the
// reference's formulas); the device sums them per lane actually tested.
constexpr uint32_t F_XFORM = 33, F_SPHERE = 32, F_PLANE = 3, F_BOX = 16, F_CYL = 40, F_CONE = 46, F_TRI = 45,
                   F_PRIMARY = 50, F_PREPARE = 95, F_LIGHT = 45, F_SHADOW = 14, F_SCHLICK = 20, F_REFRACT = 25,
                   F_PATTERN = 18;

// ---------------------------------------------------------------- wave helpers
DEV uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, WAVE);
        v = o < v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
DEV uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, WAVE);
    return v;
}

// ---------------------------------------------------------------- boxes and shapes
// BoundingBox::check_axis (object/bounding_box/mod.rs:92-112)
DEV void check_axis(double origin, double direction, double mn, double mx, double& tmin, double& tmax) {
    double min_numerator = mn - origin, max_numerator = mx - origin;
    double a, b;
    if (approx0(direction)) {
        a = min_numerator * __builtin_inf();
        b = max_numerator * __builtin_inf();
    } else {
        a = min_numerator / direction;
        b = max_numerator / direction;
    }
    if (a > b) { tmin = b; tmax = a; } else { tmin = a; tmax = b; }
}
// BoundingBox::intersect (object/bounding_box/mod.rs:57-89); fmax/fmin drop a NaN operand like f64::max/min.
DEV bool box_intersect(const Ray& r, double x0, double y0, double z0, double x1, double y1, double z1, double& tmin,
                       double& tmax) {
    double ax, bx, ay, by, az, bz;
    check_axis(r.o.x, r.d.x, x0, x1, ax, bx);
    check_axis(r.o.y, r.d.y, y0, y1, ay, by);
    check_axis(r.o.z, r.d.z, z0, z1, az, bz);
    tmin = fmax(fmax(ax, ay), az);
    tmax = fmin(fmin(bx, by), bz);
    return !(tmin > tmax || tmax < 0.0);
}

struct LeafHits { int n; double t[4]; };

// Ray into object space (object/shape.rs:49-57, math/transformation/matrix.rs:279-318);
// the direction is not renormalised.
DEV Ray to_object_u(const DevScene& S, uint32_t leaf, const Ray& r) {  // leaf wave-uniform
    const double* m = S.leaf_inv;
    uint32_t b = leaf * 12;
    Ray o;
    o.o.x = cld(m, b + 0) * r.o.x + cld(m, b + 1) * r.o.y + cld(m, b + 2) * r.o.z + cld(m, b + 3);
    o.o.y = cld(m, b + 4) * r.o.x + cld(m, b + 5) * r.o.y + cld(m, b + 6) * r.o.z + cld(m, b + 7);
    o.o.z = cld(m, b + 8) * r.o.x + cld(m, b + 9) * r.o.y + cld(m, b + 10) * r.o.z + cld(m, b + 11);
    o.d.x = cld(m, b + 0) * r.d.x + cld(m, b + 1) * r.d.y + cld(m, b + 2) * r.d.z;
    o.d.y = cld(m, b + 4) * r.d.x + cld(m, b + 5) * r.d.y + cld(m, b + 6) * r.d.z;
    o.d.z = cld(m, b + 8) * r.d.x + cld(m, b + 9) * r.d.y + cld(m, b + 10) * r.d.z;
    return o;
}
DEV V3 mul_point_v(const double* m, V3 p) {  // per-lane matrix pointer
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
DEV V3 mul_vector_v(const double* m, V3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// The six primitives in object space; t values in the reference's emit order.
DEV void sphere_hits(const Ray& r, LeafHits& h) {  // object/shapes/sphere.rs:15-35
    V3 s2r = r.o - mk(0.0, 0.0, 0.0);
    double a = dot(r.d, r.d);
    double b = 2.0 * dot(r.d, s2r);
    double c = dot(s2r, s2r) - 1.0;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return;
    double sq = sqrt(disc);
    double a2 = 2.0 * a;
    h.t[0] = (-b - sq) / a2;
    h.t[1] = (-b + sq) / a2;
    h.n = 2;
}
DEV void plane_hits(const Ray& r, LeafHits& h) {  // object/shapes/plane.rs:17-23
    if (approx0(r.d.y)) return;
    h.t[0] = -r.o.y / r.d.y;
    h.n = 1;
}
DEV void cube_hits(const Ray& r, LeafHits& h) {  // object/shapes/cube.rs:15-21
    double t0, t1;
    if (!box_intersect(r, -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, t0, t1)) return;
    h.t[0] = t0; h.t[1] = t1;
    h.n = 2;
}
DEV void push_hit(LeafHits& h, double t) {
    // n is at most 3 here; written as selects so the array stays in registers
    if (h.n == 0) h.t[0] = t; else if (h.n == 1) h.t[1] = t; else if (h.n == 2) h.t[2] = t; else h.t[3] = t;
    h.n++;
}
DEV void cylinder_hits(const Ray& r, double mn, double mx, bool closed, LeafHits& h) {  // object/shapes/cylinder.rs:26-95
    double a = r.d.x * r.d.x + r.d.z * r.d.z;
    if (!approx0(a)) {
        double b = 2.0 * (r.o.x * r.d.x + r.o.z * r.d.z);
        double c = r.o.x * r.o.x + r.o.z * r.o.z - 1.0;
        double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return;  // caps are skipped too
        double sq = sqrt(disc);
        double a2 = 2.0 * a;
        double t0 = (-b - sq) / a2, t1 = (-b + sq) / a2;
        double y0 = r.o.y + t0 * r.d.y;
        if (mn < y0 && y0 < mx) push_hit(h, t0);
        double y1 = r.o.y + t1 * r.d.y;
        if (mn < y1 && y1 < mx) push_hit(h, t1);
    }
    if (closed && !approx0(r.d.y)) {
        double t = (mn - r.o.y) / r.d.y;
        double x = r.o.x + t * r.d.x, z = r.o.z + t * r.d.z;
        if (x * x + z * z <= 1.0) push_hit(h, t);
        t = (mx - r.o.y) / r.d.y;
        x = r.o.x + t * r.d.x; z = r.o.z + t * r.d.z;
        if (x * x + z * z <= 1.0) push_hit(h, t);
    }
}
DEV void cone_hits(const Ray& r, double mn, double mx, bool closed, LeafHits& h) {  // object/shapes/cone.rs:26-100
    double a = r.d.x * r.d.x - r.d.y * r.d.y + r.d.z * r.d.z;
    double b = 2.0 * (r.o.x * r.d.x - r.o.y * r.d.y + r.o.z * r.d.z);
    double c = r.o.x * r.o.x - r.o.y * r.o.y + r.o.z * r.o.z;
    if (approx0(a)) {
        if (!approx0(b)) push_hit(h, -c / (2.0 * b));
    } else {
        double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return;
        double sq = sqrt(disc);
        double a2 = 2.0 * a;
        double t0 = (-b - sq) / a2, t1 = (-b + sq) / a2;
        double y0 = r.o.y + t0 * r.d.y;
        if (mn < y0 && y0 < mx) push_hit(h, t0);
        double y1 = r.o.y + t1 * r.d.y;
        if (mn < y1 && y1 < mx) push_hit(h, t1);
    }
    if (closed && !approx0(r.d.y)) {
        double t = (mn - r.o.y) / r.d.y;
        double x = r.o.x + t * r.d.x, z = r.o.z + t * r.d.z;
        if (x * x + z * z <= mn * mn) push_hit(h, t);
        t = (mx - r.o.y) / r.d.y;
        x = r.o.x + t * r.d.x; z = r.o.z + t * r.d.z;
        if (x * x + z * z <= mx * mx) push_hit(h, t);
    }
}
// Moller-Trumbore (object/shapes/triangle.rs:68-96); u, v returned for the normal.
DEV bool triangle_hit(const Ray& r, V3 p1, V3 e1, V3 e2, double& t, double& u, double& v) {
    V3 dce2 = cross(r.d, e2);
    double det = dot(e1, dce2);
    if (approx0(det)) return false;
    double f = 1.0 / det;
    V3 p1o = r.o - p1;
    u = f * dot(p1o, dce2);
    if (!(0.0 <= u && u <= 1.0)) return false;
    V3 oce1 = cross(p1o, e1);
    v = f * dot(r.d, oce1);
    if (v < 0.0 || (u + v) > 1.0) return false;
    t = f * dot(e2, oce1);
    return true;
}

// All intersections of one leaf (wave-uniform leaf index) with the lane's ray.
DEV void leaf_hits_u(const DevScene& S, uint32_t leaf, const Ray& world, LeafHits& h, uint32_t& flops, bool count) {
    h.n = 0;
    uint32_t shape = clu(S.leaf_meta, leaf * 4 + 0);
    Ray r = to_object_u(S, leaf, world);
    uint32_t f = F_XFORM;
    switch (shape) {
        case RT_SPHERE: sphere_hits(r, h); f += F_SPHERE; break;
        case RT_PLANE: plane_hits(r, h); f += F_PLANE; break;
        case RT_CUBE: cube_hits(r, h); f += F_BOX; break;
        case RT_CYLINDER:
            cylinder_hits(r, cld(S.leaf_prm, leaf * 4), cld(S.leaf_prm, leaf * 4 + 1), cld(S.leaf_prm, leaf * 4 + 2) != 0.0, h);
            f += F_CYL;
            break;
        case RT_CONE:
            cone_hits(r, cld(S.leaf_prm, leaf * 4), cld(S.leaf_prm, leaf * 4 + 1), cld(S.leaf_prm, leaf * 4 + 2) != 0.0, h);
            f += F_CONE;
            break;
        default: {  // RT_TRIANGLE
            uint32_t tb = clu(S.leaf_meta, leaf * 4 + 1) * 18;
            V3 p1 = mk(cld(S.tri, tb + 0), cld(S.tri, tb + 1), cld(S.tri, tb + 2));
            V3 e1 = mk(cld(S.tri, tb + 3), cld(S.tri, tb + 4), cld(S.tri, tb + 5));
            V3 e2 = mk(cld(S.tri, tb + 6), cld(S.tri, tb + 7), cld(S.tri, tb + 8));
            double t, u, v;
            if (triangle_hit(r, p1, e1, e2, t, u, v)) { h.t[0] = t; h.n = 1; }
            f += F_TRI;
            break;
        }
    }
    if (count) flops += f;
}

// ---------------------------------------------------------------- sinks
// What the reference does with the sorted world list, restated for a stream
// of intersections that arrive in insertion order (world.objects order, group
// children in stored order, a csg's filtered list in its sorted order):
//  * List::hit (intersection/list.rs:24-32) = first minimum among t > 0, so a
//    strict `<` against the running best keeps the reference's tie-break;
//  * the container walk (intersection/mod.rs:62-94) needs every intersection
//    that sorts before the hit; ContSink collects them in a second walk.
struct HitSink {
    double t;
    uint32_t leaf;
    uint32_t arrival, hit_arrival;
    bool nan_seen;
    DEV void init() { t = 0.0; leaf = NONE; arrival = 0; hit_arrival = 0; nan_seen = false; }
    DEV void emit(bool on, double tt, uint32_t lf) {
        if (on) {
            if (tt != tt) nan_seen = true;
            if (tt > 0.0 && (leaf == NONE || tt < t)) { t = tt; leaf = lf; hit_arrival = arrival; }
            arrival++;
        }
    }
};
struct ContSink {
    double hit_t;
    uint32_t hit_arrival, arrival;
    int n;
    bool overflow;
    double* ts;      // CONT_CAP entries (lane-private)
    uint32_t* leafs;
    DEV void emit(bool on, double tt, uint32_t lf) {
        if (on) {
            if (tt < hit_t || (tt == hit_t && arrival < hit_arrival)) {
                if (n < CONT_CAP) { ts[n] = tt; leafs[n] = lf; n++; } else overflow = true;
            }
            arrival++;
        }
    }
};

// ---------------------------------------------------------------- traversal
// World::intersect + Object::intersect for every lane of the wave at once
// (world.rs:90-106, object/mod.rs:100-106, object/group/mod.rs:29-47,
// object/csg/mod.rs:57-106).  `i` is wave-uniform and only grows; a lane whose
// box test fails sets skip_until to the end of that subtree.
template <bool HAS_CSG, class Sink>
DEV void traverse(const DevScene& S, const Ray& ray, bool lane_on, Sink& sink, uint32_t& flops, bool count,
                  uint32_t& err) {
    uint32_t skip_until = lane_on ? 0u : NONE;
    const uint32_t n = S.n_nodes;
    // lane-private csg state (only instantiated for scenes that have a csg)
    double ct[HAS_CSG ? CSG_CAP : 1];
    uint32_t cl[HAS_CSG ? CSG_CAP : 1];
    uint32_t cs_end[HAS_CSG ? CSG_DEPTH : 1], cs_csg[HAS_CSG ? CSG_DEPTH : 1];
    int cs_start[HAS_CSG ? CSG_DEPTH : 1];
    int clen = 0, cs_top = 0;
    (void)ct; (void)cl; (void)cs_end; (void)cs_csg; (void)cs_start; (void)clen;

    uint32_t i = 0;
    for (;;) {
        if constexpr (HAS_CSG) {
            // Close every csg whose subtree ends at or before node i.
            while (cs_top > 0 && cs_end[cs_top - 1] <= i) {
                cs_top--;
                int start = cs_start[cs_top];
                uint32_t csg = cs_csg[cs_top];
                // stable insertion sort by t (List::sort, intersection/list.rs:39-50)
                for (int a = start + 1; a < clen; a++) {
                    double ta = ct[a];
                    uint32_t la = cl[a];
                    if (ta != ta) err |= 4u;
                    int b = a - 1;
                    while (b >= start && ct[b] > ta) { ct[b + 1] = ct[b]; cl[b + 1] = cl[b]; b--; }
                    ct[b + 1] = ta; cl[b + 1] = la;
                }
                if (clen - start >= 2 && ct[start] != ct[start]) err |= 4u;
                // Csg::filter_intersections (object/csg/mod.rs:57-85)
                uint32_t op = S.csg_op[csg];
                const uint64_t* bits = S.csg_bits + (size_t)csg * S.class_words;
                bool in_left = false, in_right = false;
                int w = start;
                for (int a = start; a < clen; a++) {
                    uint32_t cls = S.leaf_meta[cl[a] * 4 + 3] & 0x7FFFFFFFu;
                    bool left_hit = (bits[cls >> 6] >> (cls & 63)) & 1ull;
                    bool allowed;
                    if (op == RT_CSG_DIFFERENCE) allowed = (left_hit && !in_right) || (!left_hit && in_left);
                    else if (op == RT_CSG_INTERSECTION) allowed = (left_hit && in_right) || (!left_hit && in_left);
                    else allowed = (left_hit && !in_right) || (!left_hit && !in_left);
                    if (allowed) { ct[w] = ct[a]; cl[w] = cl[a]; w++; }
                    if (left_hit) in_left = !in_left; else in_right = !in_right;
                }
                if (cs_top == 0) {
                    for (int a = start; a < w; a++) sink.emit(true, ct[a], cl[a]);
                    clen = start;
                } else {
                    clen = w;
                }
            }
        }
        if (i >= n) break;
        bool act = skip_until <= i;
        if (__ballot(act) == 0ull) {
            i = wave_min_u32(skip_until);
            if (i > n) i = n;
            continue;
        }
        uint32_t kind = clu(S.node_meta, i * 4 + 0);
        if (kind == RT_NODE_LEAF) {
            uint32_t leaf = clu(S.node_meta, i * 4 + 2);
            LeafHits h;
            leaf_hits_u(S, leaf, ray, h, flops, act && count);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                bool on = act && k < h.n;
                if constexpr (HAS_CSG) {
                    if (on && cs_top > 0) {
                        if (clen < CSG_CAP) { ct[clen] = h.t[k]; cl[clen] = leaf; clen++; } else err |= 2u;
                        on = false;
                    }
                }
                sink.emit(on, h.t[k], leaf);
            }
            i++;
        } else {
            // Group or csg: cull by the stored box (group/mod.rs:30-32, csg/mod.rs:89-91)
            double t0, t1;
            bool hit = box_intersect(ray, cld(S.node_bbox, i * 6 + 0), cld(S.node_bbox, i * 6 + 1),
                                     cld(S.node_bbox, i * 6 + 2), cld(S.node_bbox, i * 6 + 3),
                                     cld(S.node_bbox, i * 6 + 4), cld(S.node_bbox, i * 6 + 5), t0, t1);
            if (act && count) flops += F_BOX;
            uint32_t end = clu(S.node_meta, i * 4 + 1);
            if (act && !hit) skip_until = end;
            if constexpr (HAS_CSG) {
                if (kind == RT_NODE_CSG && act && hit) {
                    if (cs_top < CSG_DEPTH) {
                        cs_end[cs_top] = end;
                        cs_csg[cs_top] = clu(S.node_meta, i * 4 + 2);
                        cs_start[cs_top] = clen;
                        cs_top++;
                    } else {
                        err |= 2u;
                        skip_until = end;
                    }
                }
            }
            i++;
        }
    }
}

// ---------------------------------------------------------------- patterns
// Pattern::sub_pattern_at and the kinds (pattern/mod.rs:83-94, pattern/stripe.rs:11-17,
// ring.rs:12-18, checker.rs:12-18, gradient.rs:12-19, radial_gradient.rs:12-20, blend.rs:11-13).
struct PatTables { const uint32_t* pat_meta; const double* pat_inv; const double* pat_col; };
template <int DEPTH>
__device__ __noinline__ V3 sub_pattern_at(PatTables S, uint32_t node, V3 point) {
    V3 pp = mul_point_v(S.pat_inv + (size_t)node * 12, point);
    uint32_t kind = S.pat_meta[node * 4 + 0];
    if (kind == RT_PAT_SOLID) return mk(S.pat_col[node * 4 + 0], S.pat_col[node * 4 + 1], S.pat_col[node * 4 + 2]);
    uint32_t a = S.pat_meta[node * 4 + 1], b = S.pat_meta[node * 4 + 2];
    if constexpr (DEPTH <= 0) {
        return mk(0.0, 0.0, 0.0);  // deeper trees are rejected on the host
    } else {
        switch (kind) {
            case RT_PAT_STRIPE:
                return sub_pattern_at<DEPTH - 1>(S, fmod(floor(pp.x), 2.0) == 0.0 ? a : b, pp);
            case RT_PAT_RING:
                return sub_pattern_at<DEPTH - 1>(S, fmod(floor(hypot(pp.x, pp.z)), 2.0) == 0.0 ? a : b, pp);
            case RT_PAT_CHECKER:
                return sub_pattern_at<DEPTH - 1>(S, fmod(floor(pp.x) + floor(pp.y) + floor(pp.z), 2.0) == 0.0 ? a : b, pp);
            case RT_PAT_GRADIENT: {
                V3 ca = sub_pattern_at<DEPTH - 1>(S, a, pp);
                V3 distance = sub_pattern_at<DEPTH - 1>(S, b, pp) - ca;
                double fraction = pp.x - floor(pp.x);
                return ca + distance * fraction;
            }
            case RT_PAT_RADIAL_GRADIENT: {
                V3 ca = sub_pattern_at<DEPTH - 1>(S, a, pp);
                V3 distance = sub_pattern_at<DEPTH - 1>(S, b, pp) - ca;
                double rd = hypot(pp.x, pp.z);
                double fraction = rd - floor(rd);
                return ca + distance * fraction;
            }
            default:  // RT_PAT_BLEND
                return (sub_pattern_at<DEPTH - 1>(S, a, pp) + sub_pattern_at<DEPTH - 1>(S, b, pp)) / 2.0;
        }
    }
}

// ---------------------------------------------------------------- per-level work
struct RayRec {  // 64 B queue entry
    double ox, oy, oz, dx, dy, dz;
    uint32_t pad[4];
};
struct NodeRec {  // 64 B: one per traced ray
    double r, g, b;              // surface colour, later the folded colour
    double reflective, transparency;
    double reflectance;          // schlick value, or < 0 when the plain sum applies
    uint32_t child_refl, child_refr;  // record indices or NONE
    uint32_t pad[2];
};

struct Counters {  // device-side statistics
    unsigned long long secondary, shadow, shade, flop;
    uint32_t next_count;  // rays appended to the next level
    uint32_t error;       // bit0 queue/record overflow, bit1 list overflow, bit2 NaN distance
};

struct LevelArgs {
    // rays of this level
    uint32_t n_rays;          // level >= 1 (and caller-supplied rays)
    const RayRec* in_rays;
    // primary mode: tile geometry of the chunk
    uint32_t width, chunk_row0, chunk_rows, tiles_x, n_tiles;
    uint32_t part_block, part_rank, part_world;  // row partition (global y of a local row)
    double cam_inv[12];
    double half_width, half_height, pixel_size;
    // outputs
    NodeRec* recs;            // records of the whole chunk
    uint32_t rec_base;        // first record of this level
    uint32_t next_rec_base;   // first record of the next level
    RayRec* out_rays;
    uint32_t out_cap;         // capacity of out_rays
    uint32_t rec_cap;         // capacity of recs
    uint32_t depth_remaining; // the reference's `depth` argument at this level
    Counters* counters;
};

struct Comps {  // intersection/computations.rs:12-24
    V3 point, over_point, under_point, eye, normal, reflectv;
    double n1, n2;
    bool inside;
};

// Shape::normal_at (object/shape.rs:60-70) with the per-shape normals
// (sphere.rs:38-40, plane.rs:26-32, cube.rs:24-40, cylinder.rs:98-108, cone.rs:103-118,
// triangle.rs:99-104).  `leaf` differs per lane here.
DEV V3 normal_at(const DevScene& S, uint32_t leaf, V3 world_point, const Ray& world_ray) {
    const double* m = S.leaf_inv + (size_t)leaf * 12;
    V3 p = mul_point_v(m, world_point);
    uint32_t shape = S.leaf_meta[leaf * 4 + 0];
    V3 n;
    if (shape == RT_SPHERE) {
        n = p - mk(0.0, 0.0, 0.0);
    } else if (shape == RT_PLANE) {
        n = mk(0.0, 1.0, 0.0);
    } else if (shape == RT_CUBE) {
        double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
        if (ax >= ay) {
            if (ax >= az) n = mk(p.x, 0.0, 0.0); else n = mk(0.0, 0.0, p.z);
        } else if (ay >= az) {
            n = mk(0.0, p.y, 0.0);
        } else {
            n = mk(0.0, 0.0, p.z);
        }
    } else if (shape == RT_CYLINDER || shape == RT_CONE) {
        double mn = S.leaf_prm[leaf * 4 + 0], mx = S.leaf_prm[leaf * 4 + 1];
        double dist = p.x * p.x + p.z * p.z;
        if (dist < 1.0 && p.y >= mx - EPS) n = mk(0.0, 1.0, 0.0);
        else if (dist < 1.0 && p.y <= mn + EPS) n = mk(-0.0, -1.0, -0.0);
        else if (shape == RT_CYLINDER) n = mk(p.x, 0.0, p.z);
        else {
            double y = sqrt(dist);
            if (p.y > 0.0) y = -y;
            n = mk(p.x, y, p.z);
        }
    } else {
        // Triangle: u, v are recomputed with the same arithmetic that found the hit.
        const double* td = S.tri + (size_t)S.leaf_meta[leaf * 4 + 1] * 18;
        Ray r;
        r.o = mul_point_v(m, world_ray.o);
        r.d = mul_vector_v(m, world_ray.d);
        double t, u = 0.0, v = 0.0;
        triangle_hit(r, mk(td[0], td[1], td[2]), mk(td[3], td[4], td[5]), mk(td[6], td[7], td[8]), t, u, v);
        V3 n1 = mk(td[9], td[10], td[11]), n2 = mk(td[12], td[13], td[14]), n3 = mk(td[15], td[16], td[17]);
        n = n2 * u + n3 * v + n1 * (1.0 - u - v);
    }
    // transpose(inverse) * n, rows 0-2 (object/shape.rs:44-46)
    V3 wn = mk(m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
               m[2] * n.x + m[6] * n.y + m[10] * n.z);
    return normalise(wn);
}

// The container walk of prepare_computations (intersection/mod.rs:62-94) over
// the collected intersections that sort before the hit, then the hit itself.
DEV void container_walk(const DevScene& S, double* ts, uint32_t* leafs, int n, double hit_t, uint32_t hit_leaf,
                        double& n1, double& n2, uint32_t& err) {
    // stable insertion sort by t (arrival order is the index order)
    for (int a = 1; a < n; a++) {
        double ta = ts[a];
        uint32_t la = leafs[a];
        int b = a - 1;
        while (b >= 0 && ts[b] > ta) { ts[b + 1] = ts[b]; leafs[b + 1] = leafs[b]; b--; }
        ts[b + 1] = ta; leafs[b + 1] = la;
    }
    uint32_t cont[CONT_CAP + 1];
    int cn = 0;
    uint32_t hit_cls = S.leaf_meta[hit_leaf * 4 + 3] & 0x7FFFFFFFu;
    n1 = __builtin_nan("");
    n2 = __builtin_nan("");
    for (int a = 0; a <= n; a++) {
        double t = a < n ? ts[a] : hit_t;
        uint32_t lf = a < n ? leafs[a] : hit_leaf;
        uint32_t cls = S.leaf_meta[lf * 4 + 3] & 0x7FFFFFFFu;
        bool is_hit = (a == n) || (cls == hit_cls && approx_eq(t, hit_t));
        if (is_hit) n1 = cn == 0 ? 1.0 : S.mat[(size_t)S.leaf_meta[cont[cn - 1] * 4 + 2] * 8 + 6];
        int k = 0;
        for (; k < cn; k++)
            if ((S.leaf_meta[cont[k] * 4 + 3] & 0x7FFFFFFFu) == cls) break;
        if (k < cn) {
            for (int j = k; j + 1 < cn; j++) cont[j] = cont[j + 1];
            cn--;
        } else {
            cont[cn++] = lf;
        }
        if (is_hit) {
            n2 = cn == 0 ? 1.0 : S.mat[(size_t)S.leaf_meta[cont[cn - 1] * 4 + 2] * 8 + 6];
            break;
        }
    }
    (void)err;
}

// Computations::schlick (intersection/computations.rs:28-45); powi(5) = x*((x*x)*(x*x)).
DEV double schlick(const Comps& c) {
    double cosv = dot(c.eye, c.normal);
    if (c.n1 > c.n2) {
        double n_ratio = c.n1 / c.n2;
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cosv * cosv);
        if (sin2_t > 1.0) return 1.0;
        cosv = sqrt(1.0 - sin2_t);
    }
    double r = (c.n1 - c.n2) / (c.n1 + c.n2);
    double r0 = r * r;
    double x = 1.0 - cosv;
    double x2 = x * x;
    return r0 + (1.0 - r0) * (x * (x2 * x2));
}

// Trace one ray per lane to its closest hit and, for refractive hits, find
// n1 / n2.  Returns the prepared computations for lanes with a hit.
template <bool HAS_CSG>
DEV bool trace_and_prepare(const DevScene& S, const Ray& ray, bool lane_on, double& hit_t, uint32_t& hit_leaf,
                           Comps& c, uint32_t& n_list, uint32_t& flops, uint32_t& err) {
    HitSink hs;
    hs.init();
    traverse<HAS_CSG>(S, ray, lane_on, hs, flops, true, err);
    if (hs.nan_seen && hs.arrival >= 2) err |= 4u;
    bool hit = lane_on && hs.leaf != NONE;
    hit_t = hs.t;
    hit_leaf = hs.leaf;
    n_list = hs.arrival;
    c.n1 = 1.0; c.n2 = 1.0;
    uint32_t matb = hit ? S.leaf_meta[hs.leaf * 4 + 2] * 8 : 0;
    double transparency = hit ? S.mat[matb + 5] : 0.0;
    bool need = hit && (!approx0(transparency) || transparency > 0.0);
    if (__ballot(need) != 0ull) {
        double cts[CONT_CAP];
        uint32_t cls[CONT_CAP];
        ContSink cs;
        cs.hit_t = hs.t; cs.hit_arrival = hs.hit_arrival; cs.arrival = 0; cs.n = 0; cs.overflow = false;
        cs.ts = cts; cs.leafs = cls;
        uint32_t dummy = 0;
        traverse<HAS_CSG>(S, ray, need, cs, dummy, false, err);
        if (need) {
            if (cs.overflow) err |= 2u;
            container_walk(S, cts, cls, cs.n, hs.t, hs.leaf, c.n1, c.n2, err);
        }
    }
    if (hit) {  // intersection/mod.rs:45-109
        c.point = ray.o + ray.d * hs.t;
        c.eye = -ray.d;
        c.normal = normal_at(S, hs.leaf, c.point, ray);
        if (dot(c.normal, c.eye) < 0.0) { c.normal = c.normal * -1.0; c.inside = true; } else c.inside = false;
        V3 off = (c.normal * 100000.0) * EPS;
        c.over_point = c.point + off;
        c.under_point = c.point - off;
        c.reflectv = reflect(ray.d, c.normal);
        flops += F_PREPARE;
    }
    return hit;
}

// World::is_shadowed (world.rs:109-126): closest hit only.
template <bool HAS_CSG>
DEV bool is_shadowed(const DevScene& S, V3 light_position, V3 point, bool lane_on, uint32_t& flops, uint32_t& err) {
    V3 vec = light_position - point;
    double distance = sqrt(dot(vec, vec));
    Ray sr;
    sr.o = point;
    sr.d = normalise(vec);
    HitSink hs;
    hs.init();
    traverse<HAS_CSG>(S, sr, lane_on, hs, flops, true, err);
    if (hs.nan_seen && hs.arrival >= 2) err |= 4u;
    if (lane_on) flops += F_SHADOW;
    bool shadowed = false;
    if (lane_on && hs.leaf != NONE) {
        bool casts = (S.leaf_meta[hs.leaf * 4 + 3] >> 31) != 0;
        shadowed = casts && hs.t < distance;
    }
    return shadowed;
}

// Material::lighting for one light position list of length 1 (material.rs:47-87,
// light/point.rs:18-39).  `colour` is pattern_at(...) * light.intensity.
DEV V3 lighting_point(V3 colour, V3 light_intensity, double ambient_k, double diffuse_k, double specular_k,
                      double shininess, V3 light_position, V3 point, V3 eye, V3 normal, double intensity) {
    V3 ambient = colour * ambient_k;
    V3 diffuse = mk(0.0, 0.0, 0.0), specular = mk(0.0, 0.0, 0.0);
    V3 lv = normalise(light_position - point);
    double ldn = dot(lv, normal);
    if (ldn >= 0.0) {
        diffuse = diffuse + (colour * diffuse_k) * ldn;
        V3 rv = -reflect(lv, normal);
        double rde = dot(rv, eye);
        if (rde > 0.0) {
            double factor = pow(rde, shininess);
            specular = specular + (light_intensity * specular_k) * factor;
        }
    }
    return ambient + ((diffuse + specular) / 1.0) * intensity;
}

// Camera::ray_for_pixel (camera.rs:194-209)
DEV Ray ray_for_pixel(const LevelArgs& A, uint32_t x, uint32_t y) {
    double x_offset = ((double)x + 0.5) * A.pixel_size;
    double y_offset = ((double)y + 0.5) * A.pixel_size;
    double world_x = A.half_width - x_offset;
    double world_y = A.half_height - y_offset;
    const double* m = A.cam_inv;
    V3 pixel = mk(m[0] * world_x + m[1] * world_y + m[2] * -1.0 + m[3], m[4] * world_x + m[5] * world_y + m[6] * -1.0 + m[7],
                  m[8] * world_x + m[9] * world_y + m[10] * -1.0 + m[11]);
    V3 origin = mk(m[0] * 0.0 + m[1] * 0.0 + m[2] * 0.0 + m[3], m[4] * 0.0 + m[5] * 0.0 + m[6] * 0.0 + m[7],
                   m[8] * 0.0 + m[9] * 0.0 + m[10] * 0.0 + m[11]);
    Ray r;
    r.o = origin;
    r.d = normalise(pixel - origin);
    return r;
}

DEV uint32_t global_row(const LevelArgs& A, uint32_t local_row) {
    if (A.part_world <= 1) return local_row;
    uint32_t blk = local_row / A.part_block, in = local_row % A.part_block;
    return (blk * A.part_world + A.part_rank) * A.part_block + in;
}

// One recursion level: colour_at for every ray of the level, minus the
// recursion (world.rs:33-87,129-177).
template <bool PRIMARY, bool HAS_CSG>
__global__ __launch_bounds__(256) void k_level(DevScene S, LevelArgs A) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t flops = 0, err = 0;
    bool on;
    uint32_t rec_idx;  // record of this ray (absolute)
    Ray ray;
    if (PRIMARY) {
        // 8x8 pixel tile per wave; tiles are dealt to the 8 XCDs in contiguous screen bands
        // (blocks b and b+8 share an XCD and therefore an L2)
        uint32_t nt = A.n_tiles;
        uint32_t nb = gridDim.x, q = nb / 8, r = nb % 8, xcd = blockIdx.x % 8, k = blockIdx.x / 8;
        uint32_t vblock = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        uint32_t tile = vblock * 4 + (threadIdx.x >> 6);
        bool tile_ok = tile < nt;
        uint32_t tx = tile_ok ? tile % A.tiles_x : 0, ty = tile_ok ? tile / A.tiles_x : 0;
        uint32_t x = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
        on = tile_ok && x < A.width && ly < A.chunk_rows;
        rec_idx = on ? ly * A.width + x : 0;
        uint32_t gy = global_row(A, A.chunk_row0 + (on ? ly : 0));
        ray = ray_for_pixel(A, on ? x : 0, gy);
        if (on) flops += F_PRIMARY;
    } else {
        uint32_t idx = wave * 64 + lane;
        on = idx < A.n_rays;
        rec_idx = A.rec_base + (on ? idx : 0);
        const RayRec* q = A.in_rays + (on ? idx : 0);
        ray.o = mk(q->ox, q->oy, q->oz);
        ray.d = mk(q->dx, q->dy, q->dz);
    }

    double hit_t;
    uint32_t hit_leaf, n_list;
    Comps c;
    bool hit = trace_and_prepare<HAS_CSG>(S, ray, on, hit_t, hit_leaf, c, n_list, flops, err);

    // material of the hit
    uint32_t matb = hit ? S.leaf_meta[hit_leaf * 4 + 2] * 8 : 0;
    double m_ambient = S.mat[matb + 0], m_diffuse = S.mat[matb + 1], m_specular = S.mat[matb + 2],
           m_shininess = S.mat[matb + 3], m_reflective = S.mat[matb + 4], m_transparency = S.mat[matb + 5];

    // shade_hit (world.rs:52-70): lights in order, intensity_at before lighting
    V3 surface = mk(0.0, 0.0, 0.0);
    uint32_t n_shadow = 0;
    if (__ballot(hit) != 0ull) {
        V3 pat = mk(0.0, 0.0, 0.0);
        if (hit) {
            V3 object_point = mul_point_v(S.leaf_inv + (size_t)hit_leaf * 12, c.over_point);
            uint32_t root = (uint32_t)S.mat[matb + 7];
            PatTables pt;
            pt.pat_meta = S.pat_meta; pt.pat_inv = S.pat_inv; pt.pat_col = S.pat_col;
            pat = sub_pattern_at<PAT_DEPTH>(pt, root, object_point);
            flops += F_PATTERN;
        }
        for (uint32_t l = 0; l < S.n_lights; l++) {
            V3 lpos = mk(cld(S.light, l * 12 + 0), cld(S.light, l * 12 + 1), cld(S.light, l * 12 + 2));
            V3 lint = mk(cld(S.light, l * 12 + 3), cld(S.light, l * 12 + 4), cld(S.light, l * 12 + 5));
            bool shadowed = is_shadowed<HAS_CSG>(S, lpos, c.over_point, hit, flops, err);
            if (hit) {
                n_shadow++;
                double intensity = shadowed ? 0.0 : 1.0;
                V3 colour = cmul(pat, lint);
                surface = surface + lighting_point(colour, lint, m_ambient, m_diffuse, m_specular, m_shininess, lpos,
                                                   c.over_point, c.eye, c.normal, intensity);
                flops += F_LIGHT;
            }
        }
    }

    // reflected_colour / refracted_colour spawn tests (world.rs:129-177)
    bool spawn_refl = hit && A.depth_remaining > 0 && m_reflective > 0.0;
    bool spawn_refr = false;
    V3 refr_dir = mk(0.0, 0.0, 0.0);
    if (hit && A.depth_remaining > 0 && !approx0(m_transparency)) {
        double n_ratio = c.n1 / c.n2;
        double cos_i = dot(c.eye, c.normal);
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (!(sin2_t > 1.0)) {
            double cos_t = sqrt(1.0 - sin2_t);
            refr_dir = c.normal * (n_ratio * cos_i - cos_t) - c.eye * n_ratio;
            spawn_refr = true;
        }
        flops += F_REFRACT;
    }
    double reflectance = -1.0;
    if (hit && m_reflective > 0.0 && m_transparency > 0.0) {
        reflectance = schlick(c);
        flops += F_SCHLICK;
    }

    // append children to the next level: one atomic per wave
    unsigned long long b_refl = __ballot(spawn_refl), b_refr = __ballot(spawn_refr);
    uint32_t n_refl = __popcll(b_refl), n_refr = __popcll(b_refr);
    uint32_t child_refl = NONE, child_refr = NONE;
    if (n_refl + n_refr > 0) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&A.counters->next_count, n_refl + n_refr);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        unsigned long long below = (1ull << lane) - 1ull;
        if (spawn_refl) {
            uint32_t slot = base + __popcll(b_refl & below);
            if (slot < A.out_cap && A.next_rec_base + slot < A.rec_cap) {
                RayRec* o = A.out_rays + slot;
                o->ox = c.over_point.x; o->oy = c.over_point.y; o->oz = c.over_point.z;
                o->dx = c.reflectv.x; o->dy = c.reflectv.y; o->dz = c.reflectv.z;
                child_refl = A.next_rec_base + slot;
            } else err |= 1u;
        }
        if (spawn_refr) {
            uint32_t slot = base + n_refl + __popcll(b_refr & below);
            if (slot < A.out_cap && A.next_rec_base + slot < A.rec_cap) {
                RayRec* o = A.out_rays + slot;
                o->ox = c.under_point.x; o->oy = c.under_point.y; o->oz = c.under_point.z;
                o->dx = refr_dir.x; o->dy = refr_dir.y; o->dz = refr_dir.z;
                child_refr = A.next_rec_base + slot;
            } else err |= 1u;
        }
    }

    if (on) {
        NodeRec* r = A.recs + rec_idx;
        r->r = surface.x; r->g = surface.y; r->b = surface.z;
        r->reflective = m_reflective;
        r->transparency = m_transparency;
        r->reflectance = reflectance;
        r->child_refl = child_refl;
        r->child_refr = child_refr;
    }

    // statistics: one atomic per wave and counter
    uint32_t s_shade = wave_sum_u32(hit ? 1u : 0u);
    uint32_t s_shadow = wave_sum_u32(n_shadow);
    uint32_t s_flops_lo = wave_sum_u32(flops & 0xFFFFu), s_flops_hi = wave_sum_u32(flops >> 16);
    unsigned long long any_err = __ballot(err != 0);
    if (lane == 0) {
        if (n_refl + n_refr) atomicAdd(&A.counters->secondary, (unsigned long long)(n_refl + n_refr));
        if (s_shade) atomicAdd(&A.counters->shade, (unsigned long long)s_shade);
        if (s_shadow) atomicAdd(&A.counters->shadow, (unsigned long long)s_shadow);
        atomicAdd(&A.counters->flop, (unsigned long long)s_flops_lo + ((unsigned long long)s_flops_hi << 16));
    }
    if (any_err && err) atomicOr(&A.counters->error, err);
}

// Colour::to_u8 (colour.rs:63-72): clamp, * 255, round half away from zero, NaN -> 0.
DEV uint8_t to_u8(double c) {
    if (c != c) return 0;
    double cl = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
    return (uint8_t)round(cl * 255.0);
}

// Fold one level of records bottom-up (world.rs:72-86,144,176).  With
// `emit` set (level 0) the folded colour also goes to the frame buffers.
__global__ __launch_bounds__(256) void k_combine(NodeRec* recs, uint32_t base, uint32_t n, double* out_rgb,
                                                 uint8_t* out_rgb8, size_t out_offset) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    NodeRec* r = recs + base + i;
    V3 surface = mk(r->r, r->g, r->b);
    V3 reflected = mk(0.0, 0.0, 0.0), refracted = mk(0.0, 0.0, 0.0);
    uint32_t cr = r->child_refl, ct = r->child_refr;
    if (cr != NONE) reflected = mk(recs[cr].r, recs[cr].g, recs[cr].b) * r->reflective;
    if (ct != NONE) refracted = mk(recs[ct].r, recs[ct].g, recs[ct].b) * r->transparency;
    double R = r->reflectance;
    V3 colour = R >= 0.0 ? surface + reflected * R + refracted * (1.0 - R) : surface + reflected + refracted;
    r->r = colour.x; r->g = colour.y; r->b = colour.z;
    if (out_rgb) {
        double* o = out_rgb + (out_offset + i) * 3;
        o[0] = colour.x; o[1] = colour.y; o[2] = colour.z;
    }
    if (out_rgb8) {
        uint8_t* o = out_rgb8 + (out_offset + i) * 3;
        o[0] = to_u8(colour.x); o[1] = to_u8(colour.y); o[2] = to_u8(colour.z);
    }
}

// rt_probe: prepare_computations of the closest hit, for parity checks.
template <bool HAS_CSG>
__global__ __launch_bounds__(256) void k_probe(DevScene S, uint32_t n, const RayRec* rays, double* out, Counters* counters) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool on = idx < n;
    const RayRec* q = rays + (on ? idx : 0);
    Ray ray;
    ray.o = mk(q->ox, q->oy, q->oz);
    ray.d = mk(q->dx, q->dy, q->dz);
    double hit_t;
    uint32_t hit_leaf, n_list, flops = 0, err = 0;
    Comps c;
    bool hit = trace_and_prepare<HAS_CSG>(S, ray, on, hit_t, hit_leaf, c, n_list, flops, err);
    double sh[4] = {0.0, 0.0, 0.0, 0.0};
    for (uint32_t l = 0; l < S.n_lights && l < 4; l++) {
        V3 lpos = mk(cld(S.light, l * 12 + 0), cld(S.light, l * 12 + 1), cld(S.light, l * 12 + 2));
        bool s = is_shadowed<HAS_CSG>(S, lpos, c.over_point, hit, flops, err);
        double v = s ? 1.0 : 0.0;
        if (l == 0) sh[0] = v; else if (l == 1) sh[1] = v; else if (l == 2) sh[2] = v; else sh[3] = v;
    }
    if (on) {
        double* o = out + (size_t)idx * 32;
        for (int k = 0; k < 32; k++) o[k] = 0.0;
        o[0] = -1.0;
        o[24] = (double)n_list;
        if (hit) {
            o[0] = (double)hit_leaf; o[1] = hit_t;
            o[2] = c.point.x; o[3] = c.point.y; o[4] = c.point.z;
            o[5] = c.over_point.x; o[6] = c.over_point.y; o[7] = c.over_point.z;
            o[8] = c.under_point.x; o[9] = c.under_point.y; o[10] = c.under_point.z;
            o[11] = c.eye.x; o[12] = c.eye.y; o[13] = c.eye.z;
            o[14] = c.normal.x; o[15] = c.normal.y; o[16] = c.normal.z;
            o[17] = c.inside ? 1.0 : 0.0;
            o[18] = c.reflectv.x; o[19] = c.reflectv.y; o[20] = c.reflectv.z;
            // n1 / n2 are only defined by the walk for refractive hits here; opaque hits report them too
            o[21] = c.n1; o[22] = c.n2;
            o[23] = schlick(c);
            o[25] = sh[0]; o[26] = sh[1]; o[27] = sh[2]; o[28] = sh[3];
        }
    }
    if (err) atomicOr(&counters->error, err);
}

// ---------------------------------------------------------------- host side of the ABI
thread_local std::string g_error;
int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

template <class T>
int upload(const std::vector<T>& host, T** dev) {
    size_t bytes = (host.size() ? host.size() : 1) * sizeof(T);
    HIP_TRY(hipMalloc((void**)dev, bytes));
    if (!host.empty()) HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

}  // namespace

struct rt_scene {
    DevScene dev{};
    std::vector<void*> allocations;
    bool has_csg = false, has_area = false;
    int device = 0;
    // workspace, grown on demand and kept between calls
    NodeRec* recs = nullptr;
    size_t rec_cap = 0;
    RayRec* queue[2] = {nullptr, nullptr};
    size_t queue_cap = 0;
    Counters* counters = nullptr;
    Counters* counters_host = nullptr;  // pinned
    double* stage_rgb = nullptr;        // device staging for host outputs
    uint8_t* stage_rgb8 = nullptr;
    size_t stage_pixels = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> launch_ev;  // start/stop pairs around every level kernel
    size_t chunk_pixels = (size_t)4 << 20;
    size_t recs_per_pixel = 24;
};

namespace {

int ensure_workspace(rt_scene* s, size_t chunk_px, size_t n_first) {
    size_t want_q = std::max<size_t>(2 * std::max(chunk_px, n_first), 1024);
    size_t want_r = std::max<size_t>(s->recs_per_pixel * std::max(chunk_px, n_first), 4096);
    if (want_r > 0x7FFFFFF0ull) want_r = 0x7FFFFFF0ull;
    if (want_q > s->queue_cap) {
        for (int k = 0; k < 2; k++) {
            if (s->queue[k]) HIP_TRY(hipFree(s->queue[k]));
            s->queue[k] = nullptr;
            HIP_TRY(hipMalloc((void**)&s->queue[k], want_q * sizeof(RayRec)));
        }
        s->queue_cap = want_q;
    }
    if (want_r > s->rec_cap) {
        if (s->recs) HIP_TRY(hipFree(s->recs));
        s->recs = nullptr;
        HIP_TRY(hipMalloc((void**)&s->recs, want_r * sizeof(NodeRec)));
        s->rec_cap = want_r;
    }
    if (!s->counters) {
        HIP_TRY(hipMalloc((void**)&s->counters, sizeof(Counters)));
        HIP_TRY(hipHostMalloc((void**)&s->counters_host, sizeof(Counters)));
        for (auto& e : s->ev) HIP_TRY(hipEventCreate(&e));
    }
    return RT_OK;
}

struct RunTotals {
    uint64_t secondary = 0, shadow = 0, shade = 0, flop = 0, launches = 0;
    uint32_t levels = 0;
    double ms_trace = 0, ms_combine = 0;
};

// Run the levels for one batch of first-level rays (a pixel chunk, or rays
// already placed in queue[0]) and fold the records.  Returns RT_ERR_CAPACITY
// with *overflow set when a queue or the record buffer was too small.
template <bool HAS_CSG>
int run_levels(rt_scene* s, LevelArgs base, bool primary, uint32_t n_first, uint32_t depth, double* out_rgb,
               uint8_t* out_rgb8, size_t out_offset, RunTotals& tot, bool* overflow) {
    *overflow = false;
    std::vector<uint32_t> level_base, level_n;
    HIP_TRY(hipMemsetAsync(s->counters, 0, sizeof(Counters), 0));
    uint32_t n = n_first, rec_base = 0;
    int qi = 0;
    HIP_TRY(hipEventRecord(s->ev[0], 0));
    for (uint32_t level = 0;; level++) {
        LevelArgs A = base;
        A.n_rays = n;
        A.in_rays = s->queue[qi];
        A.out_rays = s->queue[qi ^ 1];
        A.out_cap = (uint32_t)std::min<size_t>(s->queue_cap, 0xFFFFFFF0u);
        A.recs = s->recs;
        A.rec_cap = (uint32_t)s->rec_cap;
        A.rec_base = rec_base;
        A.next_rec_base = rec_base + n;
        A.depth_remaining = depth >= level ? depth - level : 0;
        A.counters = s->counters;
        if ((size_t)rec_base + n > s->rec_cap) { *overflow = true; return RT_ERR_CAPACITY; }
        level_base.push_back(rec_base);
        level_n.push_back(n);
        while (s->launch_ev.size() < 2 * ((size_t)level + 1)) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            s->launch_ev.push_back(e);
        }
        HIP_TRY(hipEventRecord(s->launch_ev[2 * level], 0));
        if (primary && level == 0) {
            uint32_t waves = A.n_tiles;
            uint32_t blocks = (waves + 3) / 4;
            hipLaunchKernelGGL((k_level<true, HAS_CSG>), dim3(blocks), dim3(256), 0, 0, s->dev, A);
        } else {
            uint32_t blocks = (n + 255) / 256;
            hipLaunchKernelGGL((k_level<false, HAS_CSG>), dim3(blocks), dim3(256), 0, 0, s->dev, A);
        }
        tot.launches++;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(s->launch_ev[2 * level + 1], 0));
        HIP_TRY(hipMemcpyAsync(s->counters_host, s->counters, sizeof(Counters), hipMemcpyDeviceToHost, 0));
        HIP_TRY(hipStreamSynchronize(0));
        uint32_t next = s->counters_host->next_count;
        uint32_t error = s->counters_host->error;
        if (error & 1u) { *overflow = true; return RT_ERR_CAPACITY; }
        if (error & 2u)
            return fail(RT_ERR_CAPACITY, "a per-ray intersection list overflowed (csg list " + std::to_string(CSG_CAP) +
                                             ", container list " + std::to_string(CONT_CAP) + ")");
        if (error & 4u) return fail(RT_ERR_NAN, "NaN intersection distance (the reference panics while sorting)");
        if (next > A.out_cap) { *overflow = true; return RT_ERR_CAPACITY; }
        rec_base += n;
        if (next == 0 || level >= depth) break;
        // reset the append counter for the next level
        HIP_TRY(hipMemsetAsync(&s->counters->next_count, 0, sizeof(uint32_t), 0));
        n = next;
        qi ^= 1;
    }
    HIP_TRY(hipEventRecord(s->ev[1], 0));
    // bottom-up fold; level 0 writes the outputs
    for (int level = (int)level_n.size() - 1; level >= 0; level--) {
        uint32_t ln = level_n[level];
        uint32_t blocks = (ln + 255) / 256;
        bool emit = level == 0;
        hipLaunchKernelGGL(k_combine, dim3(blocks), dim3(256), 0, 0, s->recs, level_base[level], ln,
                           emit ? out_rgb : nullptr, emit ? out_rgb8 : nullptr, out_offset);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev[2], 0));
    HIP_TRY(hipStreamSynchronize(0));
    float ms = 0;
    for (size_t level = 0; level < level_n.size(); level++) {
        HIP_TRY(hipEventElapsedTime(&ms, s->launch_ev[2 * level], s->launch_ev[2 * level + 1]));
        tot.ms_trace += ms;
    }
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
    tot.ms_combine += ms;
    tot.secondary += s->counters_host->secondary;
    tot.shadow += s->counters_host->shadow;
    tot.shade += s->counters_host->shade;
    tot.flop += s->counters_host->flop;
    tot.levels = std::max<uint32_t>(tot.levels, (uint32_t)level_n.size());
    return RT_OK;
}

int run_levels_dispatch(rt_scene* s, const LevelArgs& base, bool primary, uint32_t n_first, uint32_t depth,
                        double* out_rgb, uint8_t* out_rgb8, size_t out_offset, RunTotals& tot, bool* overflow) {
    if (s->has_csg) return run_levels<true>(s, base, primary, n_first, depth, out_rgb, out_rgb8, out_offset, tot, overflow);
    return run_levels<false>(s, base, primary, n_first, depth, out_rgb, out_rgb8, out_offset, tot, overflow);
}

uint32_t local_rows(const rt_camera* cam, const rt_partition* p) {
    if (!p || p->world <= 1) return cam->height;
    uint32_t rows = 0;
    for (uint32_t b = 0; b * p->block_rows < cam->height; b++)
        if (b % p->world == p->rank) rows += std::min(p->block_rows, cam->height - b * p->block_rows);
    return rows;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }
uint32_t rt_max_csg_list(void) { return CSG_CAP; }
uint32_t rt_max_container_list(void) { return CONT_CAP; }

int rt_init(int device) {
    HIP_TRY(hipSetDevice(device));
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc* d, rt_scene** out) {
    if (!d || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (d->n_nodes == 0 || d->n_leaves == 0) return fail(RT_ERR_INVALID, "No objects were defined");
    if (d->n_lights == 0) return fail(RT_ERR_INVALID, "No lights were defined");
    // validate the tree
    uint32_t leaf_seen = 0, csg_seen = 0;
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        if (d->node_skip[i] <= i || d->node_skip[i] > d->n_nodes) return fail(RT_ERR_INVALID, "node_skip out of range");
        if (d->node_kind[i] == RT_NODE_LEAF) {
            if (d->node_skip[i] != i + 1) return fail(RT_ERR_INVALID, "a leaf node must have no subtree");
            if (d->node_arg[i] != leaf_seen++) return fail(RT_ERR_INVALID, "leaves must be numbered in node order");
        } else if (d->node_kind[i] == RT_NODE_CSG) {
            if (d->node_arg[i] != csg_seen++) return fail(RT_ERR_INVALID, "csgs must be numbered in node order");
            if (i + 1 >= d->n_nodes || d->node_skip[i + 1] >= d->node_skip[i])
                return fail(RT_ERR_INVALID, "a csg needs a left and a right subtree");
            if (d->node_skip[d->node_skip[i + 1]] != d->node_skip[i])
                return fail(RT_ERR_INVALID, "a csg must have exactly two children");
        } else if (d->node_kind[i] != RT_NODE_GROUP) {
            return fail(RT_ERR_INVALID, "unknown node kind");
        }
    }
    if (leaf_seen != d->n_leaves || csg_seen != d->n_csgs) return fail(RT_ERR_INVALID, "leaf / csg count mismatch");
    for (uint32_t i = 0; i < d->n_leaves; i++) {
        if (d->leaf_shape[i] > RT_TRIANGLE) return fail(RT_ERR_INVALID, "unknown shape kind");
        if (d->leaf_shape[i] == RT_TRIANGLE && d->leaf_tri[i] >= d->n_tris) return fail(RT_ERR_INVALID, "triangle index out of range");
        if (d->leaf_material[i] >= d->n_materials) return fail(RT_ERR_INVALID, "material index out of range");
        if (d->leaf_class[i] >= d->n_classes) return fail(RT_ERR_INVALID, "class id out of range");
    }
    // pattern trees: known kinds, children in range, bounded depth
    std::vector<int> pdepth(d->n_patterns, 0);
    for (uint32_t i = 0; i < d->n_patterns; i++) {
        uint32_t k = d->pat_kind[i];
        if (k > RT_PAT_BLEND)
            return fail(RT_ERR_UNSUPPORTED, "pattern kind " + std::to_string(k) + " is not available on the device (perturbed patterns need libnoise)");
        if (k != RT_PAT_SOLID) {
            if (d->pat_a[i] >= i || d->pat_b[i] >= i) return fail(RT_ERR_INVALID, "pattern children must precede their parent");
            pdepth[i] = 1 + std::max(pdepth[d->pat_a[i]], pdepth[d->pat_b[i]]);
            if (pdepth[i] > PAT_DEPTH) return fail(RT_ERR_UNSUPPORTED, "pattern nesting deeper than " + std::to_string(PAT_DEPTH));
        }
    }
    for (uint32_t i = 0; i < d->n_materials; i++)
        if (d->material_pattern[i] >= d->n_patterns) return fail(RT_ERR_INVALID, "pattern index out of range");

    auto* s = new rt_scene();
    hipGetDevice(&s->device);
    auto keep = [&](void* p) { s->allocations.push_back(p); };
    int rc = RT_OK;
#define UP(vec, field)                                   \
    do {                                                 \
        decltype(vec)::value_type* p_ = nullptr;         \
        rc = upload(vec, &p_);                           \
        if (rc != RT_OK) { rt_scene_destroy(s); return rc; } \
        keep(p_);                                        \
        s->dev.field = p_;                               \
    } while (0)

    std::vector<uint32_t> node_meta(d->n_nodes * 4, 0);
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        node_meta[i * 4 + 0] = d->node_kind[i];
        node_meta[i * 4 + 1] = d->node_skip[i];
        node_meta[i * 4 + 2] = d->node_arg[i];
        if (d->node_kind[i] == RT_NODE_CSG) s->has_csg = true;
    }
    std::vector<double> node_bbox(d->node_bbox, d->node_bbox + (size_t)d->n_nodes * 6);
    std::vector<double> leaf_inv(d->leaf_inv, d->leaf_inv + (size_t)d->n_leaves * 12);
    std::vector<double> leaf_prm((size_t)d->n_leaves * 4, 0.0);
    std::vector<uint32_t> leaf_meta((size_t)d->n_leaves * 4, 0);
    for (uint32_t i = 0; i < d->n_leaves; i++) {
        for (int k = 0; k < 3; k++) leaf_prm[(size_t)i * 4 + k] = d->leaf_param[(size_t)i * 3 + k];
        leaf_meta[(size_t)i * 4 + 0] = d->leaf_shape[i];
        leaf_meta[(size_t)i * 4 + 1] = d->leaf_tri[i];
        leaf_meta[(size_t)i * 4 + 2] = d->leaf_material[i];
        leaf_meta[(size_t)i * 4 + 3] = (d->leaf_class[i] & 0x7FFFFFFFu) | (d->leaf_casts_shadow[i] ? 0x80000000u : 0u);
    }
    std::vector<double> tri(d->tri_data, d->tri_data + (size_t)d->n_tris * 18);
    std::vector<uint32_t> csg_op(d->csg_op, d->csg_op + d->n_csgs);
    size_t words = ((size_t)d->n_classes + 63) / 64;
    std::vector<uint64_t> csg_bits(d->csg_left_bits, d->csg_left_bits + (size_t)d->n_csgs * words);
    std::vector<double> mat((size_t)d->n_materials * 8, 0.0);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        for (int k = 0; k < 7; k++) mat[(size_t)i * 8 + k] = d->material[(size_t)i * 7 + k];
        mat[(size_t)i * 8 + 7] = (double)d->material_pattern[i];
    }
    std::vector<uint32_t> pat_meta((size_t)d->n_patterns * 4, 0);
    std::vector<double> pat_col((size_t)d->n_patterns * 4, 0.0);
    for (uint32_t i = 0; i < d->n_patterns; i++) {
        pat_meta[(size_t)i * 4 + 0] = d->pat_kind[i];
        pat_meta[(size_t)i * 4 + 1] = d->pat_a[i];
        pat_meta[(size_t)i * 4 + 2] = d->pat_b[i];
        for (int k = 0; k < 3; k++) pat_col[(size_t)i * 4 + k] = d->pat_colour[(size_t)i * 3 + k];
    }
    std::vector<double> pat_inv(d->pat_inv, d->pat_inv + (size_t)d->n_patterns * 12);
    std::vector<double> light(d->light_data, d->light_data + (size_t)d->n_lights * 12);
    std::vector<uint32_t> light_meta((size_t)d->n_lights * 4, 0);
    for (uint32_t i = 0; i < d->n_lights; i++) {
        light_meta[(size_t)i * 4 + 0] = d->light_kind[i];
        light_meta[(size_t)i * 4 + 1] = d->light_steps[(size_t)i * 2];
        light_meta[(size_t)i * 4 + 2] = d->light_steps[(size_t)i * 2 + 1];
        if (d->light_kind[i] == RT_LIGHT_AREA) s->has_area = true;
        else if (d->light_kind[i] != RT_LIGHT_POINT) { rt_scene_destroy(s); return fail(RT_ERR_INVALID, "unknown light kind"); }
    }
    s->dev.n_nodes = d->n_nodes; s->dev.n_leaves = d->n_leaves; s->dev.n_lights = d->n_lights;
    s->dev.class_words = (uint32_t)words;
    UP(node_meta, node_meta); UP(node_bbox, node_bbox); UP(leaf_inv, leaf_inv); UP(leaf_prm, leaf_prm);
    UP(leaf_meta, leaf_meta); UP(tri, tri); UP(csg_op, csg_op); UP(csg_bits, csg_bits); UP(mat, mat);
    UP(pat_meta, pat_meta); UP(pat_inv, pat_inv); UP(pat_col, pat_col); UP(light, light); UP(light_meta, light_meta);
#undef UP
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    for (void* p : s->allocations) (void)hipFree(p);
    if (s->recs) (void)hipFree(s->recs);
    for (auto& q : s->queue) if (q) (void)hipFree(q);
    if (s->counters) (void)hipFree(s->counters);
    if (s->counters_host) (void)hipHostFree(s->counters_host);
    if (s->stage_rgb) (void)hipFree(s->stage_rgb);
    if (s->stage_rgb8) (void)hipFree(s->stage_rgb8);
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : s->launch_ev) if (e) (void)hipEventDestroy(e);
    delete s;
}

int rt_render(rt_scene* s, const rt_camera* cam, uint32_t depth, const uint64_t* row_seeds, const rt_partition* part,
              double* out_rgb, uint8_t* out_rgb8, int out_on_device, rt_stats* stats) {
    if (!s || !cam) return fail(RT_ERR_INVALID, "null argument");
    if (cam->width == 0 || cam->height == 0) return fail(RT_ERR_INVALID, "empty image");
    rt_partition p = part ? *part : rt_partition{0, 0, 1};
    if (p.world == 0) p.world = 1;
    if (p.world > 1 && (p.block_rows == 0 || p.block_rows % 8 != 0 || p.rank >= p.world))
        return fail(RT_ERR_INVALID, "partition needs block_rows a multiple of 8 and rank < world");
    if (s->has_area) return fail(RT_ERR_UNSUPPORTED, "area lights are not available on the device path yet");
    (void)row_seeds;
    HIP_TRY(hipSetDevice(s->device));

    const uint32_t W = cam->width;
    const uint32_t rows = local_rows(cam, &p);
    RunTotals tot;
    uint32_t chunks = 0;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));

    // device outputs (stage when the caller's buffers are host memory)
    double* d_rgb = out_rgb;
    uint8_t* d_rgb8 = out_rgb8;
    size_t total_px = (size_t)rows * W;
    if (!out_on_device && total_px > 0) {
        if (s->stage_pixels < total_px) {
            if (s->stage_rgb) HIP_TRY(hipFree(s->stage_rgb));
            if (s->stage_rgb8) HIP_TRY(hipFree(s->stage_rgb8));
            s->stage_rgb = nullptr; s->stage_rgb8 = nullptr;
            HIP_TRY(hipMalloc((void**)&s->stage_rgb, total_px * 3 * sizeof(double)));
            HIP_TRY(hipMalloc((void**)&s->stage_rgb8, total_px * 3));
            s->stage_pixels = total_px;
        }
        d_rgb = out_rgb ? s->stage_rgb : nullptr;
        d_rgb8 = out_rgb8 ? s->stage_rgb8 : nullptr;
    }

    // row chunks: multiples of 8 rows, at most chunk_pixels pixels
    uint32_t chunk_rows = (uint32_t)std::max<size_t>(8, (s->chunk_pixels / W) & ~(size_t)7);
    std::vector<std::pair<uint32_t, uint32_t>> work;  // (row0, rows)
    for (uint32_t r0 = 0; r0 < rows; r0 += chunk_rows) work.push_back({r0, std::min(chunk_rows, rows - r0)});
    std::reverse(work.begin(), work.end());
    while (!work.empty()) {
        auto [r0, nr] = work.back();
        work.pop_back();
        size_t px = (size_t)nr * W;
        int rc = ensure_workspace(s, px, 0);
        if (rc != RT_OK) return rc;
        LevelArgs A{};
        A.width = W; A.chunk_row0 = r0; A.chunk_rows = nr;
        A.tiles_x = (W + 7) / 8;
        A.n_tiles = A.tiles_x * ((nr + 7) / 8);
        A.part_block = p.block_rows; A.part_rank = p.rank; A.part_world = p.world;
        std::memcpy(A.cam_inv, cam->inv, sizeof A.cam_inv);
        A.half_width = cam->half_width; A.half_height = cam->half_height; A.pixel_size = cam->pixel_size;
        bool overflow = false;
        rc = run_levels_dispatch(s, A, true, (uint32_t)px, depth, d_rgb, d_rgb8, (size_t)r0 * W, tot, &overflow);
        if (rc == RT_ERR_CAPACITY && overflow) {
            if (nr <= 8) {
                // cannot split further: grow the per-pixel record budget instead
                if (s->recs_per_pixel >= 4096) return fail(RT_ERR_CAPACITY, "ray queues overflow even for an 8-row chunk");
                s->recs_per_pixel *= 4;
                work.push_back({r0, nr});
                continue;
            }
            uint32_t half = ((nr / 2) + 7) & ~7u;
            work.push_back({r0 + half, nr - half});
            work.push_back({r0, half});
            continue;
        }
        if (rc != RT_OK) return rc;
        chunks++;
    }
    if (!out_on_device) {
        if (out_rgb) HIP_TRY(hipMemcpy(out_rgb, d_rgb, total_px * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (out_rgb8) HIP_TRY(hipMemcpy(out_rgb8, d_rgb8, total_px * 3, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->primary_rays = total_px;
        stats->secondary_rays = tot.secondary;
        stats->shadow_rays = tot.shadow;
        stats->shade_hits = tot.shade;
        stats->levels = tot.levels;
        stats->chunks = chunks;
        stats->ms_total = ms;
        stats->ms_trace = tot.ms_trace;
        stats->ms_combine = tot.ms_combine;
        stats->trace_launches = tot.launches;
        stats->algorithmic_flop = tot.flop;
    }
    return RT_OK;
}

int rt_colour_at(rt_scene* s, uint32_t n, const double* rays, uint32_t depth, const uint64_t* seeds, double* out_rgb) {
    if (!s || (n && (!rays || !out_rgb))) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    if (s->has_area) return fail(RT_ERR_UNSUPPORTED, "area lights are not available on the device path yet");
    (void)seeds;
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_workspace(s, 0, n);
    if (rc != RT_OK) return rc;
    std::vector<RayRec> host(n);
    for (uint32_t i = 0; i < n; i++) {
        std::memset(&host[i], 0, sizeof(RayRec));
        host[i].ox = rays[6 * (size_t)i + 0]; host[i].oy = rays[6 * (size_t)i + 1]; host[i].oz = rays[6 * (size_t)i + 2];
        host[i].dx = rays[6 * (size_t)i + 3]; host[i].dy = rays[6 * (size_t)i + 4]; host[i].dz = rays[6 * (size_t)i + 5];
    }
    HIP_TRY(hipMemcpy(s->queue[0], host.data(), (size_t)n * sizeof(RayRec), hipMemcpyHostToDevice));
    double* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, (size_t)n * 3 * sizeof(double)));
    LevelArgs A{};
    RunTotals tot;
    bool overflow = false;
    for (;;) {
        rc = run_levels_dispatch(s, A, false, n, depth, d_out, nullptr, 0, tot, &overflow);
        if (rc == RT_ERR_CAPACITY && overflow && s->recs_per_pixel < 4096) {
            s->recs_per_pixel *= 4;
            int rc2 = ensure_workspace(s, 0, n);
            if (rc2 != RT_OK) { (void)hipFree(d_out); return rc2; }
            HIP_TRY(hipMemcpy(s->queue[0], host.data(), (size_t)n * sizeof(RayRec), hipMemcpyHostToDevice));
            continue;
        }
        break;
    }
    if (rc == RT_OK) {
        hipError_t e = hipMemcpy(out_rgb, d_out, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, hipGetErrorString(e));
    } else if (rc == RT_ERR_CAPACITY && overflow) {
        fail(RT_ERR_CAPACITY, "ray queues overflow");
    }
    (void)hipFree(d_out);
    return rc;
}

int rt_probe(rt_scene* s, uint32_t n, const double* rays, double* out) {
    if (!s || (n && (!rays || !out))) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_workspace(s, 0, n);
    if (rc != RT_OK) return rc;
    std::vector<RayRec> host(n);
    for (uint32_t i = 0; i < n; i++) {
        std::memset(&host[i], 0, sizeof(RayRec));
        host[i].ox = rays[6 * (size_t)i + 0]; host[i].oy = rays[6 * (size_t)i + 1]; host[i].oz = rays[6 * (size_t)i + 2];
        host[i].dx = rays[6 * (size_t)i + 3]; host[i].dy = rays[6 * (size_t)i + 4]; host[i].dz = rays[6 * (size_t)i + 5];
    }
    HIP_TRY(hipMemcpy(s->queue[0], host.data(), (size_t)n * sizeof(RayRec), hipMemcpyHostToDevice));
    double* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, (size_t)n * 32 * sizeof(double)));
    HIP_TRY(hipMemset(s->counters, 0, sizeof(Counters)));
    uint32_t blocks = (n + 255) / 256;
    if (s->has_csg) hipLaunchKernelGGL((k_probe<true>), dim3(blocks), dim3(256), 0, 0, s->dev, n, s->queue[0], d_out, s->counters);
    else hipLaunchKernelGGL((k_probe<false>), dim3(blocks), dim3(256), 0, 0, s->dev, n, s->queue[0], d_out, s->counters);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 32 * sizeof(double), hipMemcpyDeviceToHost);
    Counters c{};
    if (e == hipSuccess) e = hipMemcpy(&c, s->counters, sizeof c, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    if (c.error & 2u) return fail(RT_ERR_CAPACITY, "a per-ray intersection list overflowed");
    if (c.error & 4u) return fail(RT_ERR_NAN, "NaN intersection distance (the reference panics while sorting)");
    return RT_OK;
}

}  // extern "C"
