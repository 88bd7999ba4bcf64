// MI355X (gfx950) rendering core: the device side of include/rt_api.h.
//
// The recursion of World::colour_at becomes level-synchronous waves of rays.  A frame is cut into
// chunks of whole image rows; for each chunk the host queues, without waiting for the GPU,
//   k_chunk_begin, one k_trace per recursion level, [k_shadow], one k_shade per level from the
//   deepest up, k_chunk_end,
// on one of a few streams (chunks alternate over them, each with its own workspace).  How many
// rays a level holds lives in device memory (LevelState), so every kernel is a fixed grid whose
// blocks stride over the level's rays.
//
//   k_trace  one ray per lane: closest hit (tree walk), n1/n2 for refractive hits, the hit's
//            geometry, the reflected / refracted rays appended to the next level (one atomic per
//            block), the hit record, and - unless an area light or a lane-walked tree forbids
//            it - one shadow walk per light.
//   k_shade  pattern, Material::lighting and
//              surface + reflected * reflectance + refracted * (1 - reflectance)
//            with exactly the association order of the recursive reference; level 0 writes the
//            frame buffers.
//
// The tree is a depth-first stream of 64-byte units with skip links.  Small trees are walked by
// the 64 lanes of a wave together (position in an SGPR, node data through scalar loads, a
// per-lane "skip until" for rejected subtrees); large ones lane by lane; a mesh is walked lane by
// lane in either case, by a hierarchy of its own (mesh_walk), so a tree that is small apart from
// its meshes takes the wave's walk down to them.  Intersections arrive in the reference's insertion
// order (ties inside a mesh go to the lower stream position, which is the same thing), so a
// strict `<` against the running best reproduces its stable sort + first-positive-t rule
// without sorting.
//
// Scenes with area lights take a deferred route, because the reference threads one rng stream
// through a whole image row in depth-first order: the level kernels only trace (ray paths never
// depend on the rng), a few small passes number the shade points of each row in the reference's
// order and step the row's generator to each of them, then k_shadow and k_shade give every
// point its own generator state.
//
// A ray keeps its csg and candidate lists in fixed-size fast memory; a chunk in which some ray
// needs more is rendered again by variants of the kernels that keep the lists in global memory.
//
// All arithmetic is f64 with contraction off; expressions keep the operand order of the
// reference (cited per function, paths relative to the reference's src/).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_api.h"

namespace {

// ---------------------------------------------------------------- basics
constexpr double EPS = 2.220446049250313e-16;  // f64::EPSILON
// Safety margins of the device-only culling tests: far above the rounding error of an intersection
// distance (~1e-15 relative), far below anything visible.
constexpr double CULL_REL = 1e-9, CULL_ABS = 1e-9;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int CSG_CAP = 32;        // intersections buffered inside one outermost csg
constexpr int CSG_DEPTH = 6;       // nested csg levels
constexpr int CONT_CAP = 32;       // intersections at or before a refractive hit
constexpr int PAT_DEPTH = 16;      // nested pattern levels
constexpr int WAVE = 64;
constexpr uint32_t SPAWN_BINS = 32;  // groups the rays a block spawns in one step are sorted into (power of two, <= 64)
#ifndef TRACE_WAVES
#define TRACE_WAVES 4     // occupancy the compiler must reach for k_trace (waves per SIMD; csg variants: 2)
#endif
#ifndef CSG_WAVES
#define CSG_WAVES 4       // the same for the walking kernels of scenes with a csg (measured on chapter16: 2 -> 4 is +50 %, spills and all)
#endif
#ifndef SHADOW_WAVES
#define SHADOW_WAVES 5    // the same for k_shadow
#endif
#ifndef LANE_TRACE_WAVES
#define LANE_TRACE_WAVES 4   // k_trace / k_shadow when every lane walks on its own
#endif
#ifndef LANE_SHADOW_WAVES
#define LANE_SHADOW_WAVES 4
#endif
#ifndef SHADE_WAVES
#define SHADE_WAVES 3     // and for k_shade
#endif
#ifndef TRACE_BLOCK
#define TRACE_BLOCK 256   // threads per k_trace block: one append atomic per block and step
#endif
#ifndef PREFETCH_RAY
#define PREFETCH_RAY 1    // k_trace fetches the next step's ray before this step's shadow walk
#endif
#ifndef REC_CAP
#define REC_CAP 8         // candidates per lane recorded in LDS during the closest-hit walk (16 B each)
#endif

#define DEV __device__ __forceinline__
// At the top of a grid-stride loop: keeps the compiler from hoisting the body's loads out of the
// loop, which would hold their registers across the whole body and cost a wave of occupancy.
#define LOOP_FENCE() asm volatile("" ::: "memory")
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
// The tree is one stream of 64-byte units in depth-first order.  A record
// starts with four dwords:
//   w0  kind (bits 0-7: RT_SPHERE..RT_TRIANGLE, K_GROUP, K_CSG) | casts_shadow << 8 | closed << 9
//       | leaf has a bounding sphere in `cull` << 10 | leaf may be skipped when wholly behind the
//       origin: << 11 in a recording walk, << 16 in a plain one (groups: MESH_BIT .. see below)
//   w1  group / csg: unit position of the first record after the subtree; leaf: material index
//   w2  leaf index / csg index
//   w3  leaf: structural-equality class
// followed by
//   group / csg (1 unit):  stored box, 6 f64
//   sphere .. cone (2 units): inverse rows 0-2 (12 f64), minimum, maximum
//   triangle (2 units): w4 transform id, w5 triangle index, then p1, e1, e2 (9 f64 from dword 8);
//       its inverse lives in `xform` (meshes share one matrix, so a wave transforms its rays once
//       per matrix, not once per triangle) and its vertex normals in `tri_normals`.
constexpr uint32_t K_GROUP = 6, K_CSG = 7;
// Device-only: a "cluster" header (1 unit) ahead of a run of sibling leaves that all have bounding
// spheres.  w1 = unit position after the run, w2 = algorithmic FLOP of the run's tests for one ray,
// bits 11 / 16 as for a leaf (set when every member has them); its own bounding sphere, which holds
// the members' spheres, sits in `cull`.  A ray that cannot reach it skips the whole run with one test.
constexpr uint32_t K_CLUSTER = 8;
constexpr uint32_t MESH_BIT = 1u << 12;  // group header w0: the subtree is a mesh with a hierarchy of its own (w2 root, w3 matrix)
constexpr uint32_t MESH_ALL_CAST = 1u << 14, MESH_NONE_CAST = 1u << 15;  // mesh root: every / no triangle casts a shadow
constexpr uint32_t NESTED_BIT = 1u << 13;
constexpr uint32_t GROUP_SOUND = 1u << 17;  // group: finite box over bounded shapes only, so it holds everything below it  // group inside a mesh: its box lies within the box of every group up to the mesh's root
struct DevScene {
    uint32_t n_units, n_lights, class_words, draws_per_shade;
    uint32_t n_materials, n_patterns;
    uint32_t flags;             // bit0: some material has a nonzero transparency; bit1: shadow rays may stop at their first
                                // hit; bit2: shadow rays walk twice (see NoRecord)
    const uint32_t* stream;
    const double* xform;        // n_xforms*12
    const double* tri_normals;  // n_tris*9
    const double* cull;         // n_units*6, at a leaf's first unit: bounding sphere centre, radius, radius^2 (world space)
    const uint32_t* csg_op;
    const uint64_t* csg_bits;
    const double* mat;          // n_materials*8: 7 scalars, pattern root (as integer value)
    const uint32_t* pat_meta;   // n_patterns*4: kind, a, b, 0
    const double* pat_inv;      // n_patterns*12
    const double* pat_col;      // n_patterns*4
    const double* light;        // n_lights*12
    const uint32_t* light_meta; // n_lights*4: kind, usteps, vsteps, 0
    const float* bvh;           // 32 words per inner node of the mesh hierarchies: four children's boxes and references (see mesh_walk)
    const double* bvh_tris;     // 10 f64 per triangle of the hierarchies' leaves: p1, e1, e2, unit position (low word)
    const uint32_t* parent;     // n_units: unit position of the enclosing group / csg, NONE at the top
    // The "big list" kernels only (a chunk whose per-ray lists overflowed the fixed ones is rendered again
    // with them): per thread, big_csg_cap entries of a csg list and big_cont_cap of a candidate list.
    double* big_ct; uint32_t* big_cl; double* big_cont_t; uint32_t *big_cont_pos, *big_cont_cls;
    uint32_t big_csg_cap, big_cont_cap;
};
constexpr int BIG_CSG_DEPTH = 64;          // nested csg levels of the big-list kernels
constexpr uint32_t BIG_LIST_MAX = 4096;    // entries per thread and list they are given at most
constexpr uint32_t BIG_GRID = 256;         // their blocks (of 256 threads): the lists are per thread

// Algorithmic FLOP constants (f64 add/sub/mul/div/sqrt/compare in the
// reference's formulas); the device sums them per lane actually tested.
constexpr uint32_t F_XFORM = 33, F_SPHERE = 32, F_PLANE = 3, F_BOX = 16, F_CYL = 40, F_CONE = 46, F_TRI = 45,
                   F_PRIMARY = 50, F_PREPARE = 95, F_LIGHT = 45, F_SHADOW = 14, F_SCHLICK = 20, F_REFRACT = 25,
                   F_PATTERN = 18, F_COMBINE = 12;

// ---------------------------------------------------------------- wave helpers
DEV uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, WAVE);
        v = o < v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
DEV uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, WAVE);
    return v;
}

// ---------------------------------------------------------------- rng
// Xoshiro256PlusPlus of rand_xoshiro 0.6.0 and gen_range(0.0..=1.0) of rand 0.8.5
// (third-party crates of the reference; call sites camera.rs:144, light/area.rs:50-51).
struct Rng { unsigned long long s0, s1, s2, s3; };
DEV unsigned long long rotl64(unsigned long long x, int k) { return (x << k) | (x >> (64 - k)); }
DEV Rng rng_seed(unsigned long long state) {
    unsigned long long s[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {  // SplitMix64
        state += 0x9e3779b97f4a7c15ULL;
        unsigned long long z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        s[i] = z ^ (z >> 31);
    }
    Rng r; r.s0 = s[0]; r.s1 = s[1]; r.s2 = s[2]; r.s3 = s[3];
    return r;
}
DEV unsigned long long rng_next(Rng& r) {
    unsigned long long result = rotl64(r.s0 + r.s3, 23) + r.s0;
    unsigned long long t = r.s1 << 17;
    r.s2 ^= r.s0; r.s3 ^= r.s1; r.s1 ^= r.s2; r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rotl64(r.s3, 45);
    return result;
}
// UniformFloat<f64>::new_inclusive(0.0, 1.0).sample(): scale = 1 / (1 - 2^-52) = 1 + 2^-52 exactly representable
DEV double rng_unit(Rng& r) {
    unsigned long long bits = (rng_next(r) >> 12) | 0x3FF0000000000000ULL;
    double v = __longlong_as_double((long long)bits) - 1.0;
    return v * 0x1.0000000000001p+0 + 0.0;
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

DEV V3 mul_point_v(const double* m, V3 p) {  // math/transformation/matrix.rs:279-298
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
DEV V3 mul_vector_v(const double* m, V3 v) {  // matrix.rs:308-318
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// Ray into object space with a wave-uniform matrix (object/shape.rs:49-57); the direction is not renormalised.
DEV Ray to_object_u(const double* m, uint32_t b, const Ray& r) {
    Ray o;
    o.o.x = cld(m, b + 0) * r.o.x + cld(m, b + 1) * r.o.y + cld(m, b + 2) * r.o.z + cld(m, b + 3);
    o.o.y = cld(m, b + 4) * r.o.x + cld(m, b + 5) * r.o.y + cld(m, b + 6) * r.o.z + cld(m, b + 7);
    o.o.z = cld(m, b + 8) * r.o.x + cld(m, b + 9) * r.o.y + cld(m, b + 10) * r.o.z + cld(m, b + 11);
    o.d.x = cld(m, b + 0) * r.d.x + cld(m, b + 1) * r.d.y + cld(m, b + 2) * r.d.z;
    o.d.y = cld(m, b + 4) * r.d.x + cld(m, b + 5) * r.d.y + cld(m, b + 6) * r.d.z;
    o.d.z = cld(m, b + 8) * r.d.x + cld(m, b + 9) * r.d.y + cld(m, b + 10) * r.d.z;
    return o;
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

// ---------------------------------------------------------------- the tree walk
// What the reference does with the sorted world list, restated for a stream
// of intersections that arrive in insertion order (world.objects order, group
// children in stored order, a csg's filtered list in its sorted order):
//  * List::hit (intersection/list.rs:24-32) = first minimum among t > 0, so a
//    strict `<` against the running best keeps the reference's tie-break;
//  * the container walk (intersection/mod.rs:62-94) needs every intersection
//    that sorts before the hit.  Only an intersection below the running best can
//    do so (an equal t that arrives later sorts after the hit), so the walk
//    records those and the caller drops the ones that the final hit undercut.
// What a lane's walks add to the statistics: algorithmic FLOP (`+=` takes those), and for the
// meshes' own hierarchies the nodes visited and triangles tested (their fetches: 128 and 80 bytes).
struct Work {
    // `flops` is a sum over the wave, kept in scalar registers: add() is called where the wave's lanes
    // are together and counts f once for every lane for which `cond` holds (f must be the same in all
    // lanes - a constant, or data of a node the whole wave is at).  Code in which the lanes have gone
    // separate ways (the lane walk, a mesh's hierarchy) counts per lane.
    unsigned long long flops;
    uint32_t lane_flops, nodes, tris;
    DEV void add(uint32_t f, bool cond) { flops += (unsigned long long)f * (uint32_t)__popcll(__ballot(cond)); }
    DEV void add_lane(uint32_t f) { lane_flops += f; }
};
struct Walk {
    double t;              // closest hit so far
    uint32_t pos;          // unit position of the hit leaf's record, or NONE
    uint32_t w0, mat, cls; // of that leaf: header word 0 (shape, casts_shadow, closed), material, class
    uint32_t arrival;      // intersections seen (the length of the reference's list)
    bool nan_seen;
    int n;                 // recorded candidates
    bool overflow;
};
DEV Walk walk_init() {
    Walk w;
    w.t = 0.0; w.pos = NONE; w.w0 = 0; w.mat = 0; w.cls = 0; w.arrival = 0; w.nan_seen = false; w.n = 0; w.overflow = false;
    return w;
}
constexpr uint32_t LIMIT = 0xFFFFFFFEu;  // Walk::pos of a walk that only has a distance bound so far
DEV Walk walk_init_below(double bound) {
    Walk w = walk_init();
    w.t = bound; w.pos = LIMIT;
    return w;
}

// Where a walk keeps the (t, unit position, class) of its candidates: entry k of this lane is at
// index k*stride.
struct ContBuf {
    double* t;
    uint32_t* pos;
    uint32_t* cls;
    int stride, cap;
    uint32_t off;  // this lane's column: entry k is at k * stride + off (the pointers themselves may be the same for all lanes)
    DEV int at(int k) const { return k * stride + (int)off; }
    DEV void put(int k, double tt, uint32_t p, uint32_t c) const { t[at(k)] = tt; pos[at(k)] = p; cls[at(k)] = c; }
    DEV double& T(int k) const { return t[at(k)]; }
    DEV uint32_t& P(int k) const { return pos[at(k)]; }
    DEV uint32_t& C(int k) const { return cls[at(k)]; }
};
// The first entries go to LDS (lanes interleaved, stride = block size), the rare rest to
// lane-private scratch at the same index.
struct RecordBuf {
    ContBuf fast, big;
    static constexpr bool enabled = true;
    static constexpr bool any_hit = false;
    static constexpr uint32_t filter = 0;
    DEV int cap() const { return big.cap; }
    DEV void put(int k, double tt, uint32_t p, uint32_t c) const {
        if (k < fast.cap) fast.put(k, tt, p, c); else big.put(k, tt, p, c);
    }
    DEV double t_at(int k) const { return k < fast.cap ? fast.T(k) : big.T(k); }
    DEV uint32_t pos_at(int k) const { return k < fast.cap ? fast.P(k) : big.P(k); }
    DEV uint32_t cls_at(int k) const { return k < fast.cap ? fast.C(k) : big.C(k); }
};
// Shadow rays only need the closest hit below the light's distance: their walk starts with that
// distance as the bound to beat (position LIMIT = nothing found yet), so everything beyond the
// light is culled like anything beyond a hit.  World::is_shadowed (world.rs:117-123) asks whether
// the closest hit casts a shadow and lies before the light; a closest hit at or beyond the light
// answers no, as does finding nothing.  When every leaf of the scene casts a shadow the first hit
// found before the light already settles it: with `any_hit` a lane that walks on its own stops
// there (the wave-uniform walk does not look: the test costs its loop more than it saves).
// A scene where nearly everything casts a shadow but not all of it (a glass case around a mesh) gets
// two walks per shadow ray instead: the first only takes leaves that cast none and finds the
// closest of them before the light; the second only takes leaves that do, starts from that result
// and stops at the first hit that beats it - there is one iff the closest hit of all casts a
// shadow.  `filter`: 0 every leaf, 1 only leaves that cast no shadow, 2 only leaves that cast one.
struct NoRecord {
    bool any_hit;
    uint32_t filter;
    static constexpr bool enabled = false;
    DEV int cap() const { return 0; }
    DEV void put(int, double, uint32_t, uint32_t) const {}
    DEV double t_at(int) const { return 0.0; }
    DEV uint32_t pos_at(int) const { return 0u; }
    DEV uint32_t cls_at(int) const { return 0u; }
};

// Close every csg of this lane whose subtree ends at or before unit position i: stable sort of
// what its children delivered, Csg::filter_intersections, then hand the survivors to the enclosing
// csg's list or (outermost) to `emit`.
template <class Emit>
DEV void close_csgs(const DevScene& S, uint32_t i, double* ct, uint32_t* cl, uint32_t* cs_end, uint32_t* cs_csg,
                    int* cs_start, int& clen, int& cs_top, uint32_t& err, Emit& emit, bool want_class) {
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
        int wr = start;
        for (int a = start; a < clen; a++) {
            uint32_t cls = S.stream[(size_t)cl[a] * 16 + 3];
            bool left_hit = (bits[cls >> 6] >> (cls & 63)) & 1ull;
            bool allowed;
            if (op == RT_CSG_DIFFERENCE) allowed = (left_hit && !in_right) || (!left_hit && in_left);
            else if (op == RT_CSG_INTERSECTION) allowed = (left_hit && in_right) || (!left_hit && in_left);
            else allowed = (left_hit && !in_right) || (!left_hit && !in_left);
            if (allowed) { ct[wr] = ct[a]; cl[wr] = cl[a]; wr++; }
            if (left_hit) in_left = !in_left; else in_right = !in_right;
        }
        if (cs_top == 0) {
            for (int a = start; a < wr; a++) {
                const uint32_t* lr = S.stream + (size_t)cl[a] * 16;
                emit(true, ct[a], cl[a], lr[3], lr[0], lr[1]);
            }
            clen = start;
        } else {
            clen = wr;
        }
    }
}

// Device-only culling with a bounding sphere (centre c, radius^2 r2, world space): the ray cannot
// reach it, or reach it before its closest hit so far, or (`skip_behind`: shapes that are always
// entered and left) it lies wholly behind the origin.  Intersections inside such a sphere can
// neither become the hit nor sort before it.
DEV bool sphere_culled(const Ray& ray, V3 c, double r2, bool skip_behind, const Walk& w) {
    const V3 oc = c - ray.o;
    const double b = dot(oc, ray.d), cc = dot(oc, oc), dd = dot(ray.d, ray.d);
    const bool outside = cc > r2;
    const bool miss = outside && b * b < dd * (cc - r2);
    const bool behind = outside && b < 0.0 && skip_behind;
    // entry parameter >= b/dd - r/|d|: beyond the hit so far iff b - t*dd > r*|d|, squared to avoid |d|
    const double past = b - (w.t * (1.0 + CULL_REL) + CULL_ABS) * dd;
    const bool beyond = w.pos != NONE && past > 0.0 && past * past > r2 * dd;
    return miss || behind || beyond;
}

// World::intersect + Object::intersect for every lane of the wave at once
// (world.rs:90-106, object/mod.rs:100-106, object/group/mod.rs:29-47,
// object/csg/mod.rs:57-106).  `record` is wave-uniform.  With CULL the walk skips tests that cannot
// change the closest hit or what sorts before it (w.arrival then no longer counts every
// intersection of the reference's list; rt_probe, which reports that count, walks without it).
template <class Buf>
DEV void mesh_walk(const DevScene& S, const Ray& ray, uint32_t root_pos, uint32_t root_ref, uint32_t xf, V3 mag, Walk& w,
                   bool record, Buf cb, bool any_hit, Work& flops, bool count);

// With HYBRID the wave walks the tree together down to the meshes and every lane walks a mesh its ray reaches on
// its own, by the mesh's hierarchy (mesh_walk): for scenes whose tree is small once the meshes are set aside.
template <bool HAS_CSG, bool CULL, bool HYBRID, class Buf>
DEV void traverse(const DevScene& S, const Ray& ray, bool lane_on, bool record, Walk& w, Buf cb, Work& flops,
                  bool count, uint32_t& err) {
    uint32_t skip_until = lane_on ? 0u : NONE;
    const uint32_t n = S.n_units;
    const double* streamd = (const double*)S.stream;
    // lane-private csg state (only instantiated for scenes that have a csg)
    double ct[HAS_CSG ? CSG_CAP : 1];
    uint32_t cl[HAS_CSG ? CSG_CAP : 1];
    uint32_t cs_end[HAS_CSG ? CSG_DEPTH : 1], cs_csg[HAS_CSG ? CSG_DEPTH : 1];
    int cs_start[HAS_CSG ? CSG_DEPTH : 1];
    int clen = 0, cs_top = 0;
    (void)ct; (void)cl; (void)cs_end; (void)cs_csg; (void)cs_start; (void)clen; (void)cs_top;

    // `cls` is the structural-equality class of the leaf at unit position p.  Only the hit's distance
    // and position are tracked here; its header words are read back when the walk is over.
    auto emit = [&](bool on, double tt, uint32_t p, uint32_t cls, uint32_t lw0) {
        if constexpr (HYBRID && !Buf::enabled) {  // a filtered shadow walk (see NoRecord)
            if (cb.filter) on = on && (((lw0 >> 8) & 1u) != 0) == (cb.filter == 2u);
        }
        if (on) {
            if (tt != tt) w.nan_seen = true;
            bool below = w.pos == NONE || tt < w.t;
            if constexpr (HYBRID && !Buf::enabled) below = below || (cb.filter == 2u && tt == w.t && p < w.pos && w.pos < LIMIT);
            if constexpr (Buf::enabled) {
                if (record && below) {
                    if (w.n < cb.cap()) { cb.put(w.n, tt, p, cls); w.n++; } else w.overflow = true;
                }
            }
            if (tt > 0.0 && below) { w.t = tt; w.pos = p; }
            w.arrival++;
        }
    };

    uint32_t i = 0;
    for (;;) {
        if constexpr (HAS_CSG) {
            auto emit_csg = [&](bool on, double tt, uint32_t p, uint32_t cls, uint32_t lw0, uint32_t) { emit(on, tt, p, cls, lw0); };
            close_csgs(S, i, ct, cl, cs_end, cs_csg, cs_start, clen, cs_top, err, emit_csg, Buf::enabled && record);
        }
        if (i >= n) break;
        bool act = skip_until <= i;
        if (__ballot(act) == 0ull) {
            i = wave_min_u32(skip_until);
            if (i > n) i = n;
            continue;
        }
        // The whole node (128 bytes: every kind keeps what the walk needs in its first two units)
        // and its culling sphere are requested together and waited for once; fetching the header
        // first and the rest by kind would put four to six dependent scalar-load round trips
        // into every leaf.
        double q[16], cs[5];
#pragma unroll
        for (int k = 0; k < 16; k++) q[k] = cld(streamd, i * 8 + k);
#pragma unroll
        for (int k = 0; k < 5; k++) cs[k] = CULL ? cld(S.cull, i * 6 + k) : 0.0;
        asm volatile("" ::"s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]), "s"(q[6]), "s"(q[7]), "s"(q[8]),
                     "s"(q[9]), "s"(q[10]), "s"(q[11]), "s"(q[12]), "s"(q[13]), "s"(q[14]), "s"(q[15]), "s"(cs[0]),
                     "s"(cs[1]), "s"(cs[2]), "s"(cs[3]), "s"(cs[4]));
        const uint32_t w0 = (uint32_t)__double2loint(q[0]), w1 = (uint32_t)__double2hiint(q[0]);
        const uint32_t w2 = (uint32_t)__double2loint(q[1]), w3 = (uint32_t)__double2hiint(q[1]);
        const uint32_t kind = w0 & 0xFFu;
        if (kind == K_CLUSTER) {
            if constexpr (CULL) {
                // lanes that cannot use anything in the run pass over it (the reference would have run
                // its tests: they still count as algorithmic work); when that is every lane, the
                // ballot at the top of the loop jumps there
                const bool pass_over = act && sphere_culled(ray, mk(cs[0], cs[1], cs[2]), cs[4], (w0 & (record ? 1u << 11 : 1u << 16)) != 0, w);
                if (pass_over) skip_until = w1;
                if (count) flops.add(w2, pass_over);
            }
            i += 1;
            continue;
        }
        if (kind >= K_GROUP) {
            // Group or csg: cull by the stored box (group/mod.rs:30-32, csg/mod.rs:89-91)
            double t0, t1;
            bool hit = box_intersect(ray, q[2], q[3], q[4], q[5], q[6], q[7], t0, t1);
            if (count) flops.add(F_BOX, act);
            uint32_t end = w1;
            // Beyond the reference's test: a box entered after the closest hit so far cannot hold a
            // closer one (nor anything that sorts before it), so its subtree is skipped as well.  Only
            // for groups whose box holds their children (object/group/mod.rs:109-115,131-139; GROUP_SOUND
            // leaves out those with an unbounded shape below them, whose boxes come out of arithmetic
            // with infinities); a csg keeps the box it was built with when it is moved
            // (object/csg/mod.rs:117-121).
            if (CULL && kind == K_GROUP && (w0 & GROUP_SOUND) && hit && w.pos != NONE && t0 > w.t * (1.0 + CULL_REL) + CULL_ABS) hit = false;
            if (act && !hit) skip_until = end;
            if constexpr (HYBRID && CULL) {
                if (kind == K_GROUP && (w0 & MESH_BIT)) {  // meshes never lie inside a csg (rt_scene_create)
                    bool skip_mesh = false, any_hit = false;
                    if constexpr (!Buf::enabled) {
                        skip_mesh = (cb.filter == 1u && (w0 & MESH_ALL_CAST)) || (cb.filter == 2u && (w0 & MESH_NONE_CAST));
                        any_hit = cb.any_hit;
                    }
                    if (act && hit) {
                        skip_until = end;
                        if (!skip_mesh) {
                            const V3 mag = mk(2.0 * fmax(fabs(q[2]), fabs(q[5])), 2.0 * fmax(fabs(q[3]), fabs(q[6])),
                                              2.0 * fmax(fabs(q[4]), fabs(q[7])));
                            const uint32_t pos_before = w.pos;
                            mesh_walk(S, ray, i, w2, w3, mag, w, record, cb, any_hit, flops, count);
                            if (any_hit && w.pos != pos_before && ((w.w0 >> 8) & 1u)) skip_until = NONE;  // settled
                        }
                    }
                }
            }
            if constexpr (HAS_CSG) {
                if (kind == K_CSG && act && hit) {
                    if (cs_top < CSG_DEPTH) {
                        cs_end[cs_top] = end;
                        cs_csg[cs_top] = w2;
                        cs_start[cs_top] = clen;
                        cs_top++;
                    } else {
                        err |= 2u;
                        skip_until = end;
                    }
                }
            }
            i += 1;
            continue;
        }
        // a leaf
        if (CULL && (w0 & (1u << 10))) {
            // Device-only culling with the leaf's bounding sphere (centre c, radius r, world space):
            // no lane's ray can reach it, or reach it before its closest hit so far, or (for shapes
            // that are always entered and left) it lies wholly behind the origin.  Such intersections
            // can neither become the hit nor sort before it, so the test is skipped for the wave.
            const bool culled = sphere_culled(ray, mk(cs[0], cs[1], cs[2]), cs[4], (w0 & (record ? 1u << 11 : 1u << 16)) != 0, w);
            if (__ballot(act && !culled) == 0ull) {
                // the reference would have run the test: it still counts as algorithmic work
                if (count)
                    flops.add(F_XFORM + (kind == RT_SPHERE ? F_SPHERE : kind == RT_CUBE ? F_BOX : kind == RT_CYLINDER ? F_CYL : F_CONE), act);
                i += 2;
                continue;
            }
        }
        LeafHits h;
        h.n = 0;
        uint32_t f = F_XFORM;
        int max_hits = 4;
        if (kind == RT_TRIANGLE) {
            max_hits = 1;
            // the mesh's matrix through scalar loads; the transformed ray is not kept from one triangle
            // to the next: twelve registers held across the whole walk cost more than the arithmetic
            const Ray tri_ray = to_object_u(S.xform, (uint32_t)__double2loint(q[2]) * 12, ray);
            V3 p1 = mk(q[4], q[5], q[6]), e1 = mk(q[7], q[8], q[9]), e2 = mk(q[10], q[11], q[12]);
            double t, u, v;
            if (triangle_hit(tri_ray, p1, e1, e2, t, u, v)) { h.t[0] = t; h.n = 1; }
            f += F_TRI;
        } else if (kind == RT_PLANE) {
            // only the y components of the object-space ray matter (shape.rs:49-57, plane.rs:17-23)
            const double oy = q[6] * ray.o.x + q[7] * ray.o.y + q[8] * ray.o.z + q[9];
            const double dy = q[6] * ray.d.x + q[7] * ray.d.y + q[8] * ray.d.z;
            bool want = !approx0(dy);
#ifndef PLANE_PRETEST
#define PLANE_PRETEST 1
#endif
            if (PLANE_PRETEST && CULL && !(Buf::enabled && record)) {
                // A walk that records nothing only uses an intersection with 0 < t < the closest so far:
                // the signs settle the first, a product (with the culling margin) the second, and most
                // rays are done without the division.  NaNs go on to the division and are reported.
                const bool unknown = oy != oy || dy != dy;
                const bool ahead = (oy < 0.0) == (dy > 0.0) && oy != 0.0;  // t > 0
                const bool within = w.pos == NONE || !(fabs(oy) > (w.t * (1.0 + CULL_REL) + CULL_ABS) * fabs(dy));
                // Inside a csg an intersection behind the origin, however far, still flips the csg's
                // inside state for everything in front of it.
                bool behind_matters = false;
                if constexpr (HAS_CSG) behind_matters = cs_top > 0;
                want = want && (unknown || (ahead ? within : behind_matters));
            }
            if (act && want) { h.t[0] = -oy / dy; h.n = 1; }
            f += F_PLANE;
            max_hits = 1;
        } else {
            Ray r;  // Shape::intersect's ray in object space (shape.rs:49-57), rows 0-2 of the inverse in q[2..13]
            r.o = mk(q[2] * ray.o.x + q[3] * ray.o.y + q[4] * ray.o.z + q[5], q[6] * ray.o.x + q[7] * ray.o.y + q[8] * ray.o.z + q[9],
                     q[10] * ray.o.x + q[11] * ray.o.y + q[12] * ray.o.z + q[13]);
            r.d = mk(q[2] * ray.d.x + q[3] * ray.d.y + q[4] * ray.d.z, q[6] * ray.d.x + q[7] * ray.d.y + q[8] * ray.d.z,
                     q[10] * ray.d.x + q[11] * ray.d.y + q[12] * ray.d.z);
            if (kind == RT_SPHERE) { sphere_hits(r, h); f += F_SPHERE; max_hits = 2; }
            else if (kind == RT_CUBE) { cube_hits(r, h); f += F_BOX; max_hits = 2; }
            else if (kind == RT_CYLINDER) { cylinder_hits(r, q[14], q[15], (w0 >> 9) & 1u, h); f += F_CYL; }
            else { cone_hits(r, q[14], q[15], (w0 >> 9) & 1u, h); f += F_CONE; }
        }
        if (count) flops.add(f, act);
#pragma unroll
        for (int k = 0; k < 4; k++) {
#ifndef UNIFORM_HITS
#define UNIFORM_HITS 1
#endif
            if (UNIFORM_HITS && k >= max_hits) break;  // wave-uniform: the node's kind bounds its intersections
            bool on = act && k < h.n;
            if constexpr (HAS_CSG) {
                if (on && cs_top > 0) {
                    if (clen < CSG_CAP) { ct[clen] = h.t[k]; cl[clen] = i; clen++; } else err |= 2u;
                    on = false;
                }
            }
            emit(on, h.t[k], i, w3, w0);
        }
        i += 2;
    }
    // the hit leaf's header words (shape, casts_shadow, closed; material; class)
    if (w.pos < LIMIT) {
        const uint32_t* lr = S.stream + (size_t)w.pos * 16;
        w.w0 = lr[0]; w.mat = lr[1]; w.cls = lr[3];
    }
}

// ---------------------------------------------------------------- meshes
// A group whose subtree holds nothing but triangles under one matrix (an OBJ mesh, divided or not)
// carries a second, device-only hierarchy over the same triangles: built with the surface-area
// heuristic in the mesh's object space as a binary tree, four triangles or fewer per leaf, then
// folded into nodes of up to four children whose boxes are kept in f32 (rounded outwards), and walked
// nearest child first with a small stack.  The reference's own tree is whatever Group::divide
// (object/group/mod.rs:50-139) made of the file's order - children that straddle a split stay in
// the parent, the walk order is fixed - so a ray through it tests many times more boxes and
// triangles than it has to.
//
// What the reference's list would hold is reproduced exactly:
//  * the triangle test is the reference's, on the same object-space ray;
//  * an intersection only counts if the ray also passes, by the reference's own predicate, the
//    stored box of every group between the triangle and the mesh's root (a ray can graze a triangle
//    on the rim of a box that rounding makes it miss); the root's box was tested on the way in;
//  * intersections arrive in another order than the reference's, so ties in t go to the lower
//    stream position, which is the earlier arrival of the reference's walk.
// The hierarchy's boxes are padded and tested with slack: they only have to be conservative.
// A walk that records what sorts before the hit (every intersection below the running best,
// behind the origin too) puts the mesh's entries into stream order afterwards, which is the order
// the reference's walk would have delivered them in.
constexpr int BVH_STACK = 64;
constexpr uint32_t BVH_LEAF = 0x80000000u;  // reference to a leaf: BVH_LEAF | (count - 1) << 27 | first entry of bvh_tris

// Does the ray pass the stored boxes of the groups strictly between unit position `pos` and `root`?
// `mag` bounds the magnitude of the root box's coordinates per axis.  A hit point well inside a
// box (by a margin far above the rounding of either computation) passes the reference's test for
// it; when that box is NESTED it also lies well inside every box further up, which ends the climb.
// Otherwise the reference's own test decides, level by level.
DEV bool mesh_ancestors_pass(const DevScene& S, const Ray& ray, double t, uint32_t pos, uint32_t root, V3 mag) {
    const V3 P = ray.o + ray.d * t;
    const double mx = 1e-9 * (mag.x + fabs(ray.o.x) + fabs(P.x));
    const double my = 1e-9 * (mag.y + fabs(ray.o.y) + fabs(P.y));
    const double mz = 1e-9 * (mag.z + fabs(ray.o.z) + fabs(P.z));
    uint32_t g = S.parent[pos];
    for (int k = 0; k < BVH_STACK && g != root && g != NONE; k++) {
        const uint32_t* hdr = S.stream + (size_t)g * 16;
        const double* b = (const double*)hdr;
        const uint32_t gw0 = hdr[0], up = S.parent[g];
        const double x0 = b[2], y0 = b[3], z0 = b[4], x1 = b[5], y1 = b[6], z1 = b[7];
        const bool inside = t > 0.0 && P.x > x0 + mx && P.x < x1 - mx && P.y > y0 + my && P.y < y1 - my && P.z > z0 + mz &&
                            P.z < z1 - mz;
        if (inside && (gw0 & NESTED_BIT)) return true;
        if (!inside) {
            double ta, tb;
            if (!box_intersect(ray, x0, y0, z0, x1, y1, z1, ta, tb)) return false;
        }
        g = up;
    }
    return true;
}

// One lane's ray against a mesh.  `ray` is the world-space ray.
// An inner node has up to four children; their boxes are stored in f32, rounded outwards, and tested in f32
// with every rounding of that arithmetic added back as slack (the boxes only have to be conservative; the
// triangle tests stay the reference's, in f64).  With u = 2^-24: the computed slab distance
// fl(fl(b - fl(o)) * fl(1/d)) differs from (b - o)/d by at most |o/d| * 1.01u + |t| * 3.1u, so every axis is
// widened by e = |o/d| * 2u and the interval by 8u of its ends.  An axis whose terms come out NaN or
// infinite (d = 0, o or 1/d beyond f32) drops out of the fmin / fmax chain and constrains nothing.
template <class Buf>
DEV void mesh_walk(const DevScene& S, const Ray& ray, uint32_t root_pos, uint32_t root_ref, uint32_t xf, V3 mag, Walk& w,
                   bool record, Buf cb, bool any_hit, Work& flops, bool count) {
    const bool rec = Buf::enabled && record;
    uint32_t filter = 0;
    if constexpr (!Buf::enabled) filter = cb.filter;
    const int n_before = w.n;
    const double* m = S.xform + (size_t)xf * 12;
    Ray r;
    r.o = mul_point_v(m, ray.o);
    r.d = mul_vector_v(m, ray.d);
    constexpr float U2 = 1.1920929e-7f, U8 = 4.7683716e-7f;  // 2u, 8u
    const float ox = (float)r.o.x, oy = (float)r.o.y, oz = (float)r.o.z;
    const float ix = (float)(1.0 / r.d.x), iy = (float)(1.0 / r.d.y), iz = (float)(1.0 / r.d.z);
    const float ex = fabsf(ox * ix) * U2, ey = fabsf(oy * iy) * U2, ez = fabsf(oz * iz) * U2;
    uint32_t stack[BVH_STACK];
    float stack_t[BVH_STACK];
    int sp = 0;
    uint32_t cur = root_ref;
    const double huge = __builtin_huge_val();
    const float inf = __builtin_huge_valf();
    // The lanes that walk this mesh together alternate between two phases, each a loop over one kind of work:
    // down through inner nodes until every lane holds a leaf (or has run out), then through the leaves'
    // triangles.  A lane that reaches its leaf early waits for the others; in exchange the wave never
    // executes both kinds of step in the same iteration.
    for (;;) {
        for (;;) {
            const double hi = w.pos == NONE ? huge : w.t * (1.0 + CULL_REL) + CULL_ABS;
            const float hi_f = (float)hi * (1.0f + U2) + 1e-30f;  // never below hi
            if (cur == NONE) {
                while (sp > 0) {
                    sp--;
                    if (!(stack_t[sp] > hi_f)) { cur = stack[sp]; break; }
                }
            }
            const bool inner = cur != NONE && !(cur & BVH_LEAF);
            if (__ballot(inner) == 0ull) break;
            if (inner) {
                const float4* nd = (const float4*)(S.bvh + (size_t)cur * 32);
                const float4 lx = nd[0], ly = nd[1], lz = nd[2], hx = nd[3], hy = nd[4], hz = nd[5];
                const uint4 refs = *(const uint4*)(nd + 6);
                if (count) { flops.add_lane(4 * F_BOX); flops.nodes++; }
                float key[4];
                uint32_t ref[4];
#define RT_CHILD(c, f)                                                                                             \
                {                                                                                                  \
                    const float ax = (lx.f - ox) * ix, bx = (hx.f - ox) * ix;                                      \
                    const float ay = (ly.f - oy) * iy, by = (hy.f - oy) * iy;                                      \
                    const float az = (lz.f - oz) * iz, bz = (hz.f - oz) * iz;                                      \
                    const float tn = fmaxf(fmaxf(fminf(ax, bx) - ex, fminf(ay, by) - ey), fminf(az, bz) - ez);     \
                    const float tf = fminf(fminf(fmaxf(ax, bx) + ex, fmaxf(ay, by) + ey), fmaxf(az, bz) + ez);     \
                    const float slack = U8 * (fminf(fabsf(tn), 1e30f) + fminf(fabsf(tf), 1e30f)) + 1e-30f;         \
                    const float lo_t = tn - slack, hi_t = tf + slack;                                              \
                    const bool hit = refs.f != NONE && lo_t <= hi_t && lo_t <= hi_f && (rec || hi_t >= 0.0f);      \
                    key[c] = hit ? lo_t : inf;                                                                     \
                    ref[c] = hit ? refs.f : NONE;                                                                  \
                }
                RT_CHILD(0, x) RT_CHILD(1, y) RT_CHILD(2, z) RT_CHILD(3, w)
#undef RT_CHILD
                // nearest first: a five-exchange sorting network over (entry distance, child); misses sort last
#define RT_SWAP(a, b)                                                                                              \
                {                                                                                                  \
                    const bool sw = key[b] < key[a];                                                               \
                    const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];                              \
                    const uint32_t ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];                           \
                    key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb;                                            \
                }
                RT_SWAP(0, 1) RT_SWAP(2, 3) RT_SWAP(0, 2) RT_SWAP(1, 3) RT_SWAP(1, 2)
#undef RT_SWAP
                // the build keeps every hierarchy shallow enough for three entries a level
                if (ref[3] != NONE && sp < BVH_STACK) { stack[sp] = ref[3]; stack_t[sp] = key[3]; sp++; }
                if (ref[2] != NONE && sp < BVH_STACK) { stack[sp] = ref[2]; stack_t[sp] = key[2]; sp++; }
                if (ref[1] != NONE && sp < BVH_STACK) { stack[sp] = ref[1]; stack_t[sp] = key[1]; sp++; }
                cur = ref[0];
            }
        }
        if (__ballot(cur != NONE) == 0ull) break;  // every lane has run out of nodes
        for (;;) {  // here `cur` is a leaf reference or NONE
            const bool leaf = cur != NONE;
            if (__ballot(leaf) == 0ull) break;
            if (leaf) {
                const uint32_t at = cur & 0x07FFFFFFu, left = (cur >> 27) & 15u;
                cur = left ? (BVH_LEAF | ((left - 1u) << 27) | (at + 1u)) : NONE;
                const double* ld = S.bvh_tris + (size_t)at * 10;
                double td[10];
#pragma unroll
                for (int k = 0; k < 10; k++) td[k] = ld[k];
                const uint32_t pos = (uint32_t)__double2loint(td[9]);
                double t, u, v;
                const bool got = triangle_hit(r, mk(td[0], td[1], td[2]), mk(td[3], td[4], td[5]), mk(td[6], td[7], td[8]), t, u, v);
                if (count) { flops.add_lane(F_XFORM + F_TRI); flops.tris++; }
                if (got) {
                    if (t != t) w.nan_seen = true;
                    w.arrival++;
                    const bool below = w.pos == NONE || t < w.t || (t == w.t && pos < w.pos && w.pos < LIMIT);
                    const uint32_t* lr = S.stream + (size_t)pos * 16;
                    const bool wanted = filter == 0 || (((lr[0] >> 8) & 1u) != 0) == (filter == 2u);
                    if ((t > 0.0 || rec) && below && wanted && mesh_ancestors_pass(S, ray, t, pos, root_pos, mag)) {
                        if (rec) {
                            if (w.n < cb.cap()) { cb.put(w.n, t, pos, lr[3]); w.n++; } else w.overflow = true;
                        }
                        if (t > 0.0) {
                            w.t = t; w.pos = pos; w.w0 = lr[0]; w.mat = lr[1]; w.cls = lr[3];
                            if (any_hit) { cur = NONE; sp = 0; }  // settled: this lane is done with the mesh
                        }
                    }
                }
            }
        }
    }
    if (rec) {  // the mesh's entries into stream order (insertion sort: there are seldom more than two)
        for (int a = n_before + 1; a < w.n; a++) {
            const double ta = cb.t_at(a);
            const uint32_t pa = cb.pos_at(a), ca = cb.cls_at(a);
            int b = a - 1;
            while (b >= n_before && cb.pos_at(b) > pa) { cb.put(b + 1, cb.t_at(b), cb.pos_at(b), cb.cls_at(b)); b--; }
            cb.put(b + 1, ta, pa, ca);
        }
    }
}

// The same walk with one position per lane, for trees too large for the wave to walk together
// (meshes): each lane follows its own path through the skip links, loading nodes with vector
// loads, so a lane only pays for the boxes and leaves its own ray reaches.  Lanes at different
// kinds of node take turns within an iteration.
template <bool HAS_CSG, bool CULL, bool BIG, class Buf>
DEV void traverse_lane(const DevScene& S, const Ray& ray, bool lane_on, bool record, Walk& w, Buf cb, Work& flops,
                       bool count, uint32_t& err) {
    const uint32_t n = S.n_units;
    // the csg lists: fixed and lane-private, or (BIG) this thread's slice of the launch's global lists
    constexpr int DEPTH = BIG ? BIG_CSG_DEPTH : CSG_DEPTH;
    double ct_own[HAS_CSG && !BIG ? CSG_CAP : 1];
    uint32_t cl_own[HAS_CSG && !BIG ? CSG_CAP : 1];
    double* ct = ct_own;
    uint32_t* cl = cl_own;
    int csg_cap = CSG_CAP;
    if constexpr (BIG) {
        const size_t slice = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * S.big_csg_cap;
        ct = S.big_ct + slice; cl = S.big_cl + slice; csg_cap = (int)S.big_csg_cap;
    }
    uint32_t cs_end[HAS_CSG ? DEPTH : 1], cs_csg[HAS_CSG ? DEPTH : 1];
    int cs_start[HAS_CSG ? DEPTH : 1];
    int clen = 0, cs_top = 0;
    (void)ct; (void)cl; (void)cs_end; (void)cs_csg; (void)cs_start; (void)clen; (void)cs_top; (void)csg_cap;
    auto emit = [&](bool on, double tt, uint32_t p, uint32_t cls, uint32_t lw0, uint32_t lmat) {
        if constexpr (!Buf::enabled) {
            // a filtered shadow walk: the other kind of leaf does not exist; a walk that starts from
            // another walk's result can meet an equal t at an earlier position, which sorts first
            if (cb.filter) on = on && (((lw0 >> 8) & 1u) != 0) == (cb.filter == 2u);
        }
        if (on) {
            if (tt != tt) w.nan_seen = true;
            bool below = w.pos == NONE || tt < w.t;
            if constexpr (!Buf::enabled) below = below || (cb.filter == 2u && tt == w.t && p < w.pos && w.pos < LIMIT);
            if constexpr (Buf::enabled) {
                if (record && below) {
                    if (w.n < cb.cap()) { cb.put(w.n, tt, p, cls); w.n++; } else w.overflow = true;
                }
            }
            if (tt > 0.0 && below) { w.t = tt; w.pos = p; w.w0 = lw0; w.mat = lmat; w.cls = cls; }
            w.arrival++;
        }
    };
    uint32_t cur_xform = NONE;  // per lane: the mesh matrix tri_ray was made with
    Ray tri_ray = ray;
    uint32_t i = lane_on ? 0u : n;
    for (;;) {
        if constexpr (HAS_CSG) close_csgs(S, i, ct, cl, cs_end, cs_csg, cs_start, clen, cs_top, err, emit, Buf::enabled && record);
        if constexpr (!Buf::enabled) {
            if (cb.any_hit && w.pos < LIMIT && ((w.w0 >> 8) & 1u)) i = n;
        }
        const bool act = i < n;
        if (__ballot(act) == 0ull) break;
        const uint32_t* rec = S.stream + (size_t)(act ? i : 0u) * 16;
        const double* recd = (const double*)rec;
        // the node's first unit in one batch of loads (header and box of a group, header and the
        // start of a leaf's data): a group then costs one memory round trip, not two in a row
        double u0[8];
#pragma unroll
        for (int k = 0; k < 8; k++) u0[k] = recd[k];
        asm volatile("" ::"v"(u0[0]), "v"(u0[1]), "v"(u0[2]), "v"(u0[3]), "v"(u0[4]), "v"(u0[5]), "v"(u0[6]), "v"(u0[7]));
        const uint32_t w0 = (uint32_t)__double2loint(u0[0]), w1 = (uint32_t)__double2hiint(u0[0]);
        const uint32_t w2 = (uint32_t)__double2loint(u0[1]), w3 = (uint32_t)__double2hiint(u0[1]);
        const uint32_t kind = w0 & 0xFFu;
        if (act && kind == K_CLUSTER) {
            bool culled = false;
            if constexpr (CULL) {
                const double* cs = S.cull + (size_t)i * 6;
                culled = sphere_culled(ray, mk(cs[0], cs[1], cs[2]), cs[4], (w0 & (record ? 1u << 11 : 1u << 16)) != 0, w);
                if (culled && count) flops.add_lane(w2);
            }
            i = culled ? w1 : i + 1;
        } else if (act && kind >= K_GROUP) {
            double t0, t1;
            bool hit = box_intersect(ray, u0[2], u0[3], u0[4], u0[5], u0[6], u0[7], t0, t1);
            if (count) flops.add_lane(F_BOX);
            if (CULL && kind == K_GROUP && (w0 & GROUP_SOUND) && hit && w.pos != NONE && t0 > w.t * (1.0 + CULL_REL) + CULL_ABS) hit = false;
            uint32_t next = hit ? i + 1 : w1;
            if constexpr (CULL) {
                bool in_csg = false;
                if constexpr (HAS_CSG) in_csg = cs_top > 0;
                bool skip_mesh = false;
                if constexpr (!Buf::enabled)
                    skip_mesh = (cb.filter == 1u && (w0 & MESH_ALL_CAST)) || (cb.filter == 2u && (w0 & MESH_NONE_CAST));
                if (hit && (w0 & MESH_BIT) && !in_csg && skip_mesh) next = w1;
                if (hit && (w0 & MESH_BIT) && !in_csg && !skip_mesh) {
                    bool any_hit = false;
                    if constexpr (!Buf::enabled) any_hit = cb.any_hit;
                    const V3 mag = mk(2.0 * fmax(fabs(u0[2]), fabs(u0[5])), 2.0 * fmax(fabs(u0[3]), fabs(u0[6])),
                                      2.0 * fmax(fabs(u0[4]), fabs(u0[7])));
                    mesh_walk(S, ray, i, w2, w3, mag, w, record, cb, any_hit, flops, count);
                    next = w1;
                }
            }
            if constexpr (HAS_CSG) {
                if (kind == K_CSG && hit) {
                    if (cs_top < DEPTH) {
                        cs_end[cs_top] = w1; cs_csg[cs_top] = w2; cs_start[cs_top] = clen;
                        cs_top++;
                    } else {
                        err |= 2u;
                        next = w1;
                    }
                }
            }
            i = next;
        } else if (act) {
            bool culled = false;
            if (CULL && (w0 & (1u << 10))) {  // see the wave-uniform walk
                const double* cs = S.cull + (size_t)i * 6;
                culled = sphere_culled(ray, mk(cs[0], cs[1], cs[2]), cs[4], (w0 & (record ? 1u << 11 : 1u << 16)) != 0, w);
                if (culled && count)
                    flops.add_lane(F_XFORM + (kind == RT_SPHERE ? F_SPHERE : kind == RT_CUBE ? F_BOX : kind == RT_CYLINDER ? F_CYL : F_CONE));
            }
            if (!culled) {
                LeafHits h;
                h.n = 0;
                uint32_t f = F_XFORM;
                if (kind == RT_TRIANGLE) {
                    const uint32_t xf = (uint32_t)__double2loint(u0[2]);
                    if (xf != cur_xform) {
                        cur_xform = xf;
                        const double* m = S.xform + (size_t)xf * 12;
                        tri_ray.o = mul_point_v(m, ray.o);
                        tri_ray.d = mul_vector_v(m, ray.d);
                    }
                    V3 p1 = mk(u0[4], u0[5], u0[6]), e1 = mk(u0[7], recd[8], recd[9]), e2 = mk(recd[10], recd[11], recd[12]);
                    double t, u, v;
                    if (triangle_hit(tri_ray, p1, e1, e2, t, u, v)) { h.t[0] = t; h.n = 1; }
                    f += F_TRI;
                } else {
                    Ray r;
                    r.o = mul_point_v(recd + 2, ray.o);
                    r.d = mul_vector_v(recd + 2, ray.d);
                    if (kind == RT_SPHERE) { sphere_hits(r, h); f += F_SPHERE; }
                    else if (kind == RT_PLANE) { plane_hits(r, h); f += F_PLANE; }
                    else if (kind == RT_CUBE) { cube_hits(r, h); f += F_BOX; }
                    else if (kind == RT_CYLINDER) { cylinder_hits(r, recd[14], recd[15], (w0 >> 9) & 1u, h); f += F_CYL; }
                    else { cone_hits(r, recd[14], recd[15], (w0 >> 9) & 1u, h); f += F_CONE; }
                }
                if (count) flops.add_lane(f);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    bool on = k < h.n;
                    if constexpr (HAS_CSG) {
                        if (on && cs_top > 0) {
                            if (clen < csg_cap) { ct[clen] = h.t[k]; cl[clen] = i; clen++; } else err |= 2u;
                            on = false;
                        }
                    }
                    emit(on, h.t[k], i, w3, w0, w1);
                }
            }
            i += 2;
        }
    }
}

// Chooses the walk: small trees are walked by the wave together (scalar loads, no divergence),
// large ones lane by lane.
template <bool HAS_CSG, bool CULL, bool PER_LANE, class Buf, bool BIG = false, bool HYBRID = false>
DEV void walk_tree(const DevScene& S, const Ray& ray, bool lane_on, bool record, Walk& w, Buf cb, Work& flops,
                   bool count, uint32_t& err) {
    static_assert(!BIG || PER_LANE, "the big-list kernels walk lane by lane");
    static_assert(!HYBRID || !PER_LANE, "the hybrid walk is the wave's walk with the meshes lane by lane");
    if constexpr (PER_LANE) traverse_lane<HAS_CSG, CULL, BIG>(S, ray, lane_on, record, w, cb, flops, count, err);
    else traverse<HAS_CSG, CULL, HYBRID>(S, ray, lane_on, record, w, cb, flops, count, err);
}

// ---------------------------------------------------------------- patterns
// Pattern::sub_pattern_at and the kinds (pattern/mod.rs:83-94, pattern/stripe.rs:11-17,
// ring.rs:12-18, checker.rs:12-18, gradient.rs:12-19, radial_gradient.rs:12-20, blend.rs:11-13),
// evaluated with an explicit stack instead of recursion.  A frame is a two-child node
// (gradient, radial gradient, blend) waiting for its children; choosing kinds (stripe, ring,
// checker) just descend.  Trees deeper than PAT_DEPTH are rejected by rt_scene_create.
// `v % 2.0 == 0.0` of the reference (Rust's `%` is fmod) for an integer-valued v: fmod is exact,
// so it is zero exactly when v is an even integer.  Every double of magnitude >= 2^53 is an even
// integer; NaN and the infinities give NaN % 2.0, which is not zero.
DEV bool is_even(double v) {
    const double a = fabs(v);
    if (!(a < 9007199254740992.0)) return a == a && a < __builtin_huge_val();
    return (((long long)a) & 1ll) == 0;
}
struct PatTables { const uint32_t* pat_meta; const double* pat_inv; const double* pat_col; };
DEV V3 pattern_eval(PatTables S, uint32_t root, V3 object_point) {
    uint32_t f_node[PAT_DEPTH + 1];
    uint32_t f_stage[PAT_DEPTH + 1];      // 0: waiting for child a, 1: waiting for child b
    V3 f_pp[PAT_DEPTH + 1], f_ca[PAT_DEPTH + 1];
    int top = 0;
    uint32_t node = root;
    V3 point = object_point;
    V3 value = mk(0.0, 0.0, 0.0);
    for (;;) {
        // descend until a solid colour is reached
        for (;;) {
            V3 pp = mul_point_v(S.pat_inv + (size_t)node * 12, point);
            uint32_t kind = S.pat_meta[node * 4 + 0];
            uint32_t a = S.pat_meta[node * 4 + 1], b = S.pat_meta[node * 4 + 2];
            if (kind == RT_PAT_SOLID) {
                value = mk(S.pat_col[node * 4 + 0], S.pat_col[node * 4 + 1], S.pat_col[node * 4 + 2]);
                break;
            }
            if (kind == RT_PAT_STRIPE) node = is_even(floor(pp.x)) ? a : b;
            else if (kind == RT_PAT_RING) node = is_even(floor(hypot(pp.x, pp.z))) ? a : b;
            else if (kind == RT_PAT_CHECKER) node = is_even(floor(pp.x) + floor(pp.y) + floor(pp.z)) ? a : b;
            else {
                if (top > PAT_DEPTH) return mk(0.0, 0.0, 0.0);
                f_node[top] = node; f_stage[top] = 0; f_pp[top] = pp; top++;
                node = a;
            }
            point = pp;
        }
        // hand the value to the waiting frames
        for (;;) {
            if (top == 0) return value;
            int t = top - 1;
            uint32_t n = f_node[t];
            if (f_stage[t] == 0) {  // child a done: evaluate child b at the same pattern-space point
                f_ca[t] = value;
                f_stage[t] = 1;
                node = S.pat_meta[n * 4 + 2];
                point = f_pp[t];
                break;
            }
            uint32_t kind = S.pat_meta[n * 4 + 0];
            V3 ca = f_ca[t], cb = value, pp = f_pp[t];
            if (kind == RT_PAT_GRADIENT) {
                V3 distance = cb - ca;
                double fraction = pp.x - floor(pp.x);
                value = ca + distance * fraction;
            } else if (kind == RT_PAT_RADIAL_GRADIENT) {
                V3 distance = cb - ca;
                double rd = hypot(pp.x, pp.z);
                double fraction = rd - floor(rd);
                value = ca + distance * fraction;
            } else {  // RT_PAT_BLEND
                value = (ca + cb) / 2.0;
            }
            top--;
        }
    }
}

// ---------------------------------------------------------------- per-level work
// One record per traced ray, kept as a structure of arrays over the records of a chunk so that a
// wave's loads and stores are contiguous.  The levels of the recursion occupy consecutive index
// ranges; a spawned ray is written straight into its own record.
// Records are laid out in groups of 64 (one wave's worth): inside a group every column is a
// contiguous run (512 bytes of f64, 256 of u32), so a wave's loads and stores stay coalesced, and a
// record's columns lie within one 9 KB stretch of memory instead of twenty places gigabytes apart
// (one page to translate per wave and step, and the columns are reached by immediate offsets from
// one address).
constexpr size_t REC_GROUP = 64, REC_GROUP_BYTES = REC_GROUP * (16 * sizeof(double) + 4 * sizeof(uint32_t));
template <class T, uint32_t OFFSET>
struct RecColumn {
    unsigned char* base;
    __host__ __device__ __forceinline__ T& operator[](size_t i) const {
        return *(T*)(base + (i / REC_GROUP) * REC_GROUP_BYTES + OFFSET + (i % REC_GROUP) * sizeof(T));
    }
};
struct Recs {
    unsigned char* base;  // capacity: a multiple of REC_GROUP records
#define RT_COL(name, k) __host__ __device__ __forceinline__ RecColumn<double, (k) * 512u> name() const { return {base}; }
    RT_COL(ox, 0) RT_COL(oy, 1) RT_COL(oz, 2) RT_COL(dx, 3) RT_COL(dy, 4) RT_COL(dz, 5)  // the ray (primary rays store the direction only)
    RT_COL(px, 6) RT_COL(py, 7) RT_COL(pz, 8)      // over_point of the hit
    RT_COL(nx, 9) RT_COL(ny, 10) RT_COL(nz, 11)    // normal at the hit
    RT_COL(reflectance, 12)                        // schlick value, where the material is reflective and transparent
    RT_COL(r, 13) RT_COL(g, 14) RT_COL(b, 15)      // colour_at of the ray, filled bottom-up
#undef RT_COL
#define RT_UCOL(name, k) __host__ __device__ __forceinline__ RecColumn<uint32_t, 8192u + (k) * 256u> name() const { return {base}; }
    RT_UCOL(hit_pos, 0)      // unit position of the hit leaf, NONE for a miss
    RT_UCOL(child_refl, 1)   // record of the spawned rays or NONE (valid for hits)
    RT_UCOL(child_refr, 2)
    RT_UCOL(order, 3)        // area lights: subtree size, then position in the row's rng order
#undef RT_UCOL
};
constexpr size_t REC_BYTES = 16 * sizeof(double) + 4 * sizeof(uint32_t);

struct Counters {
    uint32_t error;       // bit0 record arrays full, bit1 list overflow, bit2 NaN distance
};
// The levels of one chunk, kept on the device so that a level can be launched without the host
// knowing how many rays the previous one spawned: level L traces records [base, base + n).
struct LevelState { uint32_t n, base; };
struct ChunkStatus {
    uint32_t error, total_recs, levels, pad;
    unsigned long long secondary, shadow, shade, flop_trace, flop_shadow, flop_light, mesh_nodes, mesh_tris;  // this chunk's statistics
};
// Statistics are spread over STAT_SHARDS cache lines (by block index): every wave adds to them once,
// and atomics to a single line serialise at ~85 wave-operations per microsecond on this chip.
constexpr uint32_t STAT_SHARDS = 256;
struct __attribute__((aligned(64))) StatShard {
    unsigned long long secondary, shadow, shade, flop_trace, flop_shadow, flop_light, mesh_nodes, mesh_tris;
#ifdef RT_PHASE_TIMERS
    unsigned long long phase[8];  // shader-clock ticks per k_trace phase, summed over waves
#endif
};
// Diagnostic build (-DRT_PHASE_TIMERS): k_trace drains its memory counters at each PHASE mark and
// adds the elapsed shader clock to StatShard::phase; rt_render prints the sums to stderr.
#ifdef RT_PHASE_TIMERS
#define PHASE(k) do { __builtin_amdgcn_s_waitcnt(0); phase_t[k] = __builtin_readcyclecounter(); } while (0)
#else
#define PHASE(k) do {} while (0)
#endif

struct LevelArgs {
    LevelState* levels;       // per chunk; entry L+1 is filled while level L runs
    uint32_t level;           // the level this launch works on (k_trace, k_shade)
    uint32_t n_levels;        // levels[n_levels].base = records of the whole chunk (k_shadow)
    // primary mode: tile geometry of the chunk
    uint32_t width, chunk_row0, chunk_rows, tiles_x, n_tiles;
    uint32_t part_block, part_rank, part_world;  // row partition (global y of a local row)
    double cam_inv[12];
    double half_width, half_height, pixel_size;
    Recs R;                   // records of the whole chunk
    double* intensity;        // Lightable::intensity_at of light l for record i at [l * rec_cap + i]
    const Rng* rng_states;    // area lights only: one generator state per shade point of the chunk
    uint32_t rec_cap;         // capacity of the record arrays
    uint32_t depth_remaining; // the reference's `depth` argument at this level
    // frame buffers (level 0 of k_shade), indexed from out_offset
    double* out_rgb;
    uint8_t* out_rgb8;
    size_t out_offset;
    Counters* counters;
    StatShard* stats;
};

// ---- per-ray helpers, kept out of line on purpose: each runs once per ray (not once per node),
// and keeping their registers out of the tree walk's allocation is worth far more than the calls.

DEV const double* leaf_inverse(const uint32_t* stream, const double* xform, uint32_t pos) {
    const uint32_t* rec = stream + (size_t)pos * 16;
    if ((rec[0] & 0xFFu) == RT_TRIANGLE) return xform + (size_t)rec[4] * 12;
    return (const double*)rec + 2;
}

struct HitGeom { V3 point, normal; int inside; };


// point, normal and inside flag of Intersection::prepare_computations (intersection/mod.rs:45-60) with
// Shape::normal_at (object/shape.rs:60-70) and the per-shape normals (sphere.rs:38-40, plane.rs:26-32,
// cube.rs:24-40, cylinder.rs:98-108, cone.rs:103-118, triangle.rs:99-104).  `pos` differs per lane.
DEV HitGeom hit_geometry(const uint32_t* stream, const double* xform, const double* tri_normals,
                                             Ray ray, double t, uint32_t pos) {
    HitGeom g;
    g.point = ray.o + ray.d * t;
    const uint32_t* rec = stream + (size_t)pos * 16;
    const double* recd = (const double*)rec;
    const double* m = leaf_inverse(stream, xform, pos);
    V3 p = mul_point_v(m, g.point);
    uint32_t shape = rec[0] & 0xFFu;
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
        double mn = recd[14], mx = recd[15];
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
        Ray r;
        r.o = mul_point_v(m, ray.o);
        r.d = mul_vector_v(m, ray.d);
        double tt, u = 0.0, v = 0.0;
        triangle_hit(r, mk(recd[4], recd[5], recd[6]), mk(recd[7], recd[8], recd[9]), mk(recd[10], recd[11], recd[12]), tt, u, v);
        const double* nd = tri_normals + (size_t)rec[5] * 9;
        V3 n1 = mk(nd[0], nd[1], nd[2]), n2 = mk(nd[3], nd[4], nd[5]), n3 = mk(nd[6], nd[7], nd[8]);
        n = n2 * u + n3 * v + n1 * (1.0 - u - v);
    }
    // transpose(inverse) * n, rows 0-2 (object/shape.rs:44-46)
    V3 wn = mk(m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
               m[2] * n.x + m[6] * n.y + m[10] * n.z);
    g.normal = normalise(wn);
    V3 eye = -ray.d;
    if (dot(g.normal, eye) < 0.0) { g.normal = g.normal * -1.0; g.inside = 1; } else g.inside = 0;
    return g;
}
// over_point / under_point (intersection/mod.rs:100-101)
DEV V3 over_point_of(V3 point, V3 normal) { return point + (normal * 100000.0) * EPS; }
DEV V3 under_point_of(V3 point, V3 normal) { return point - (normal * 100000.0) * EPS; }

struct N1N2 { double n1, n2; };
// The container walk of prepare_computations (intersection/mod.rs:62-94).  `cb` holds n candidate
// intersections in arrival order; those with t < hit_t sort before the hit (an equal t arrived
// after it).  Objects are compared by class id, which stands for the reference's structural `==`.
// The containers list is built in place over the entries already consumed.
DEV N1N2 container_walk(const uint32_t* stream, const double* mat, ContBuf cb, int n, double hit_t, uint32_t hit_pos,
                        uint32_t hit_cls) {
    (void)cb.stride;
    int m = 0;
    for (int a = 0; a < n; a++) {
        double ta = cb.T(a);
        if (ta < hit_t) {
            if (m != a) cb.put(m, ta, cb.P(a), cb.C(a));
            m++;
        }
    }
    // stable insertion sort by t (List::sort, intersection/list.rs:39-50)
    for (int a = 1; a < m; a++) {
        double ta = cb.T(a);
        if (!(cb.T((a - 1)) > ta)) continue;
        uint32_t la = cb.P(a), ca = cb.C(a);
        int b = a - 1;
        while (b >= 0 && cb.T(b) > ta) { cb.put(b + 1, cb.T(b), cb.P(b), cb.C(b)); b--; }
        cb.put(b + 1, ta, la, ca);
    }
    int cn = 0;  // containers are entries [0, cn) of pos/cls, oldest first; cn <= a
    uint32_t n1_pos = NONE, n2_pos = NONE;
    for (int a = 0; a <= m; a++) {
        double t = hit_t;
        uint32_t lf = hit_pos, cls = hit_cls;
        if (a < m) { t = cb.T(a); lf = cb.P(a); cls = cb.C(a); }
        const bool is_hit = (a == m) || (cls == hit_cls && approx_eq(t, hit_t));
        int k = 0;
        while (k < cn && cb.C(k) != cls) k++;
        if (is_hit) {
            n1_pos = cn == 0 ? NONE : cb.P((cn - 1));
            if (k == cn) n2_pos = lf;                                   // appended: the hit object itself
            else if (k < cn - 1) n2_pos = cb.P((cn - 1));         // removed below the top
            else n2_pos = cn >= 2 ? cb.P((cn - 2)) : NONE;        // the top was removed
            break;
        }
        if (k < cn) {
            for (int j = k; j + 1 < cn; j++) { cb.P(j) = cb.P((j + 1)); cb.C(j) = cb.C((j + 1)); }
            cn--;
        } else {
            cb.P(cn) = lf; cb.C(cn) = cls; cn++;
        }
    }
    N1N2 r;
    r.n1 = n1_pos == NONE ? 1.0 : mat[(size_t)stream[(size_t)n1_pos * 16 + 1] * 8 + 6];
    r.n2 = n2_pos == NONE ? 1.0 : mat[(size_t)stream[(size_t)n2_pos * 16 + 1] * 8 + 6];
    return r;
}
// The same result for lists that fit the LDS buffer, computed in registers without sorting.  In the
// reference's sorted walk an object is in `containers` iff it has appeared an odd number of times,
// and the list's last element is the member that was pushed most recently, i.e. whose latest
// appearance has the greatest (t, arrival) key.  So n1 comes from the odd-count object with the
// greatest latest key among the entries before the hit; for n2 the hit object is toggled first.
// An entry that equals the hit (same class, approx_eq t; intersection/mod.rs:64) ends the walk
// early: it and everything after it in sorted order drop out.  `need` is false for lanes that only
// keep the wave company.
DEV N1N2 container_walk_regs(const uint32_t* stream, const double* mat, ContBuf cb, int n, bool need, double hit_t,
                             uint32_t hit_pos, uint32_t hit_cls, bool few_classes) {
    constexpr int N = REC_CAP;
    (void)cb.stride;
    // the longest list of the wave bounds every loop below (wave-uniform, so the unrolled
    // iterations past it are skipped by a scalar branch)
    int nmax = 0;
#pragma unroll
    for (int k = N; k >= 1; k--)
        if (nmax == 0 && __ballot(need && n >= k) != 0ull) nmax = k;
    double t[N];
    uint32_t p[N], c[N];
    bool v[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        t[k] = 0.0; p[k] = NONE; c[k] = NONE; v[k] = false;
        if (k < nmax) {
            t[k] = cb.T(k); p[k] = cb.P(k); c[k] = cb.C(k);
            v[k] = need && k < n && t[k] < hit_t;
        }
    }
    // early end of the walk (rare): the first entry in sorted order that equals the hit
    // (a cheap superset of approx_eq first: 4 ulps or f64::EPSILON are below 1e-15 relative + 3e-16)
    bool m[N], any_m = false;
    const double near = fabs(hit_t) * 1e-15 + 3e-16;
#pragma unroll
    for (int k = 0; k < N; k++) {
        m[k] = false;
        if (k < nmax) { m[k] = v[k] && c[k] == hit_cls && hit_t - t[k] <= near; any_m = any_m || m[k]; }
    }
    if (__ballot(any_m) != 0ull) {
#pragma unroll
        for (int k = 0; k < N; k++) m[k] = m[k] && approx_eq(t[k], hit_t);
        bool have = false;
        double st = 0.0;
#pragma unroll
        for (int k = 0; k < N; k++) {  // ascending index: a strict `<` keeps the earliest arrival on ties
            bool take = m[k] && (!have || t[k] < st);
            if (take) { st = t[k]; have = true; }
        }
        bool passed = false;  // entries at or after the stop entry in (t, arrival) order
#pragma unroll
        for (int k = 0; k < N; k++) {
            // the stop entry is the first index with t == st among m; later indices with t == st follow it
            if (have && m[k] && t[k] == st) passed = true;
            bool drop = have && (t[k] > st || (t[k] == st && passed));
            v[k] = v[k] && !drop;
        }
    }
    bool cand[N];
    if (few_classes) {
        // class ids fit a 64-bit set: one pass for the parities, and an entry is a candidate iff
        // its class has odd parity (the greatest key among them is necessarily its class's latest)
        unsigned long long parity = 0ull;
#pragma unroll
        for (int i = 0; i < N; i++)
            if (i < nmax && v[i]) parity ^= 1ull << (c[i] & 63u);
#pragma unroll
        for (int i = 0; i < N; i++) cand[i] = i < nmax && v[i] && ((parity >> (c[i] & 63u)) & 1ull);
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) {
            cand[i] = false;
            if (i < nmax) {
                bool odd = false, last = true;
#pragma unroll
                for (int j = 0; j < N; j++) {
                    if (j < nmax) {
                        bool same = v[j] && c[j] == c[i];
                        odd = odd != same;
                        if (j != i) {
                            bool later = j > i ? t[j] >= t[i] : t[j] > t[i];
                            if (same && later) last = false;
                        }
                    }
                }
                cand[i] = v[i] && odd && last;
            }
        }
    }
    bool have1 = false, have2 = false, hit_present = false;
    double t1 = 0.0, t2 = 0.0;
    uint32_t n1_pos = NONE, n2_pos = NONE;
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (i < nmax) {
            bool is_hit_cls = c[i] == hit_cls;
            if (cand[i] && (!have1 || t[i] >= t1)) { t1 = t[i]; n1_pos = p[i]; have1 = true; }
            if (cand[i] && is_hit_cls) hit_present = true;
            if (cand[i] && !is_hit_cls && (!have2 || t[i] >= t2)) { t2 = t[i]; n2_pos = p[i]; have2 = true; }
        }
    }
    if (!hit_present) n2_pos = hit_pos;  // the hit object was pushed last
    N1N2 r;
    r.n1 = 1.0; r.n2 = 1.0;
    if (need) {
        if (n1_pos != NONE) r.n1 = mat[(size_t)stream[(size_t)n1_pos * 16 + 1] * 8 + 6];
        if (n2_pos != NONE) r.n2 = mat[(size_t)stream[(size_t)n2_pos * 16 + 1] * 8 + 6];
    }
    return r;
}
// Lane-private scratch buffers make this one big; keep it out of line.
__device__ __noinline__ N1N2 container_walk_scratch(const uint32_t* stream, const double* mat, double* ts, uint32_t* poss,
                                                    uint32_t* clss, int n, double hit_t, uint32_t hit_pos, uint32_t hit_cls) {
    ContBuf cb = {ts, poss, clss, 1, CONT_CAP, 0u};
    return container_walk(stream, mat, cb, n, hit_t, hit_pos, hit_cls);
}

// Computations::schlick (intersection/computations.rs:28-45); powi(5) = x*((x*x)*(x*x)).
DEV double schlick(V3 eye, V3 normal, double n1, double n2) {
    double cosv = dot(eye, normal);
    if (n1 > n2) {
        double n_ratio = n1 / n2;
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cosv * cosv);
        if (sin2_t > 1.0) return 1.0;
        cosv = sqrt(1.0 - sin2_t);
    }
    double r = (n1 - n2) / (n1 + n2);
    double r0 = r * r;
    double x = 1.0 - cosv;
    double x2 = x * x;
    return r0 + (1.0 - r0) * (x * (x2 * x2));
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

// ---------------------------------------------------------------- the three per-ray kernels
// k_trace  (one launch per recursion level): closest hit, n1/n2, spawn, hit record
// k_shadow (one launch per chunk, all levels at once): Lightable::intensity_at for every light
// k_shade  (one launch per level, deepest first): pattern + Material::lighting + the sum with the
//          spawned rays' colours; no tree walk, so the register-hungry pow() stays out of the
//          walking kernels' allocation
// All three run as a fixed-size grid whose blocks stride over the work: how many rays a level
// holds is only known on the device (LevelState).

// The area-light rng stream of one shade point (world.rs:60-70): per light, 2*samples draws for
// intensity_at's positions (light/area.rs:83) and then 2*samples for lighting's (material.rs:64).
DEV void rng_skip(Rng& r, uint32_t draws) { for (uint32_t k = 0; k < draws; k++) (void)rng_next(r); }

// Lightable::intensity_at of one shade point for every light (light/point.rs:27-38,
// light/area.rs:75-90): one closest-hit shadow ray per light sample (World::is_shadowed,
// world.rs:109-126).  Called by all lanes of a wave; `on` marks the lanes that hold a shade point.
template <bool HAS_CSG, bool PER_LANE, bool CULL = true, bool BIG = false, bool HYBRID = false>
DEV void intensities(const DevScene& S, const LevelArgs& A, bool on, uint32_t rec_idx, V3 over, V3 normal, Rng rng,
                     Work& flops, uint32_t& err, uint32_t& n_shadow, uint32_t l_begin = 0, uint32_t l_end = NONE) {
    if (__ballot(on) == 0ull) return;
    l_end = min(l_end, S.n_lights);  // a wave-uniform range of lights (all of them unless told otherwise)
    for (uint32_t l = l_begin; l < l_end; l++) {
        const uint32_t usteps = clu(S.light_meta, l * 4 + 1), vsteps = clu(S.light_meta, l * 4 + 2);
        const uint32_t samples = usteps * vsteps;
        const bool is_area = clu(S.light_meta, l * 4 + 0) == RT_LIGHT_AREA;
        const V3 l0 = mk(cld(S.light, l * 12 + 0), cld(S.light, l * 12 + 1), cld(S.light, l * 12 + 2));
        const V3 uc = mk(cld(S.light, l * 12 + 3), cld(S.light, l * 12 + 4), cld(S.light, l * 12 + 5));
        const V3 vc = mk(cld(S.light, l * 12 + 6), cld(S.light, l * 12 + 7), cld(S.light, l * 12 + 8));
        double unshadowed = 0.0;
        for (uint32_t k = 0; k < samples; k++) {
            V3 lp = l0;
            if (is_area) {  // light/area.rs:48-67: v outer, u inner, r1 before r2
                double r1 = rng_unit(rng);
                V3 p = l0 + uc * ((double)(k % usteps) + r1);
                double r2 = rng_unit(rng);
                lp = p + vc * ((double)(k / usteps) + r2);
            }
            V3 vec = lp - over;
            double distance = sqrt(dot(vec, vec));
            Ray sr;
            sr.o = over;
            sr.d = normalise(vec);
            // A point light behind the surface: Material::lighting takes neither diffuse nor specular
            // from it (light_dot_normal < 0, material.rs:70-83, the same expression as in k_shade), so
            // whatever is_shadowed answers is multiplied by zero.  The ray still counts as cast.
            const bool moot = CULL && !is_area && dot(sr.d, normal) < 0.0;
            const bool cast = on && !moot;
            Walk w = CULL ? walk_init_below(distance) : walk_init();
            if (__ballot(cast) != 0ull) {
                if (CULL && (PER_LANE || HYBRID) && (S.flags & 4u)) {
                    walk_tree<HAS_CSG, CULL, PER_LANE, NoRecord, BIG, HYBRID>(S, sr, cast, false, w, NoRecord{false, 1u}, flops, true, err);
                    bool nan_seen = w.nan_seen && w.arrival >= 2;
                    w.nan_seen = false; w.arrival = 0;
                    walk_tree<HAS_CSG, CULL, PER_LANE, NoRecord, BIG, HYBRID>(S, sr, cast, false, w, NoRecord{true, 2u}, flops, true, err);
                    if (nan_seen) err |= 4u;
                } else {
                    walk_tree<HAS_CSG, CULL, PER_LANE, NoRecord, BIG, HYBRID>(S, sr, cast, false, w, NoRecord{CULL && (S.flags & 2u) != 0, 0u}, flops, true, err);
                }
                if (w.nan_seen && w.arrival >= 2) err |= 4u;
            }
            bool shadowed = moot;  // reported as intensity 0: k_shade then skips the light's loop as well
            if (cast && w.pos < LIMIT) shadowed = ((w.w0 >> 8) & 1u) && w.t < distance;
            if (!shadowed) unshadowed += 1.0;
            flops.add(F_SHADOW, on);
            n_shadow += (uint32_t)__popcll(__ballot(on));
        }
        if (on) {
            double intensity = is_area ? unshadowed / (double)samples : (unshadowed != 0.0 ? 1.0 : 0.0);
            A.intensity[(size_t)l * A.rec_cap + rec_idx] = intensity;
        }
        if (is_area) rng_skip(rng, 2 * samples);  // lighting's own draws for this light
    }
}

// Colour::to_u8 (colour.rs:63-72): clamp, * 255, round half away from zero, NaN -> 0.
DEV uint8_t to_u8(double c) {
    if (c != c) return 0;
    double cl = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
    return (uint8_t)round(cl * 255.0);
}

// pow() is by far the largest register user of the shading code; out of line it only costs its
// registers while it runs.
__device__ __noinline__ double shade_pow(double x, double y) { return pow(x, y); }

// Material::lighting for one light (material.rs:47-87); `pat` is pattern_at(object, over_point)
// (pattern/mod.rs:83-87), `intensity` the light's intensity_at for this point.  With an area light
// `rng` stands at the first of intensity_at's draws for this light (world.rs:61-69) and is left
// behind lighting's own draws.
DEV V3 lighting_one(const DevScene& S, uint32_t l, V3 pat, double m_ambient, double m_diffuse, double m_specular,
                    double m_shininess, V3 over, V3 normal, V3 eye, double intensity, Rng& rng, uint32_t& flops,
                    bool after_intensity_at = true) {
    // wave-uniform light data through the scalar cache
    const uint32_t usteps = clu(S.light_meta, l * 4 + 1), vsteps = clu(S.light_meta, l * 4 + 2);
    const uint32_t samples = usteps * vsteps;
    const bool is_area = clu(S.light_meta, l * 4 + 0) == RT_LIGHT_AREA;
    const V3 l0 = mk(cld(S.light, l * 12 + 0), cld(S.light, l * 12 + 1), cld(S.light, l * 12 + 2));
    const V3 uc = mk(cld(S.light, l * 12 + 3), cld(S.light, l * 12 + 4), cld(S.light, l * 12 + 5));
    const V3 vc = mk(cld(S.light, l * 12 + 6), cld(S.light, l * 12 + 7), cld(S.light, l * 12 + 8));
    const V3 lint = is_area ? mk(cld(S.light, l * 12 + 9), cld(S.light, l * 12 + 10), cld(S.light, l * 12 + 11)) : uc;
    if (is_area && after_intensity_at) rng_skip(rng, 2 * samples);  // intensity_at's draws come first (world.rs:61-69)
    V3 col = cmul(pat, lint);
    V3 ambient = col * m_ambient;
    V3 diffuse = mk(0.0, 0.0, 0.0), specular = mk(0.0, 0.0, 0.0);
    // With intensity 0 the diffuse and specular sums are multiplied by zero (they are finite
    // for a non-negative shininess): the loop is skipped, only its rng draws are taken.
    const bool dark = intensity == 0.0 && m_shininess >= 0.0;
    if (dark && is_area) rng_skip(rng, 2 * samples);
    // A specular weight of zero makes every term of the specular sum (intensity * 0) * factor: for a
    // factor in [0, 1] that is the same signed zero (or NaN) whatever the factor, so pow() is skipped.
    const bool no_gloss = m_specular == 0.0 && m_shininess >= 0.0;
    for (uint32_t k = 0; k < (dark ? 0u : samples); k++) {
        V3 lp = l0;
        if (is_area) {  // a fresh jitter: positions() is called again (material.rs:64)
            double r1 = rng_unit(rng);
            V3 p = l0 + uc * ((double)(k % usteps) + r1);
            double r2 = rng_unit(rng);
            lp = p + vc * ((double)(k / usteps) + r2);
        }
        V3 lv = normalise(lp - over);
        double ldn = dot(lv, normal);
        if (ldn >= 0.0) {
            diffuse = diffuse + (col * m_diffuse) * ldn;
            V3 rv = -reflect(lv, normal);
            double rde = dot(rv, eye);
            if (rde > 0.0) {
                double factor = (no_gloss && rde <= 1.0) ? 1.0 : shade_pow(rde, m_shininess);
                specular = specular + (lint * m_specular) * factor;
            }
        }
    }
    flops += F_LIGHT * samples;
    return ambient + ((diffuse + specular) / (double)samples) * intensity;
}

// shade_hit's sum (world.rs:72-86); a ray that was not spawned contributes black
DEV V3 combine(V3 surface, V3 reflected, V3 refracted, double m_reflective, double m_transparency, double reflectance) {
    if (m_reflective > 0.0 && m_transparency > 0.0) return surface + reflected * reflectance + refracted * (1.0 - reflectance);
    return surface + reflected + refracted;
}

// colour_at for every ray of one level, minus the shading and the recursion (world.rs:33-49,129-177;
// intersection/mod.rs:45-109).  With FUSED the kernel also casts the hit's shadow rays (their
// arithmetic fills the issue slots this kernel leaves idle while it waits for memory); otherwise
// k_shadow does.  Computing Material::lighting here as well was measured: the kernel then no longer
// fits its 128 registers without spills and loses more than k_shade saves.
// Without CULL every walk is the reference's, test for test (a slow mode the others are checked against).
//
// The rays a block spawns in one step are appended to the next level together: one atomic on the
// level's counter per block and step (a single word serialises at ~85 atomics per microsecond, so
// per-wave atomics would), and inside the block's range the children are grouped by (object they
// start on, reflected / refracted), so that the next level's waves hold rays that meet the same
// objects and a test culled for one lane is more often culled for all 64.
template <bool PRIMARY, bool HAS_CSG, bool PER_LANE, bool FUSED, bool CULL = true, bool BIG = false, bool HYBRID = false>
__global__ __launch_bounds__(TRACE_BLOCK, HAS_CSG ? CSG_WAVES : (PER_LANE || HYBRID) ? LANE_TRACE_WAVES : TRACE_WAVES) void k_trace(DevScene S, LevelArgs A) {
    constexpr bool LANES_DIVERGE = PER_LANE || HYBRID;  // walks of very different length side by side in one wave
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_block = threadIdx.x >> 6;
    constexpr uint32_t WAVES = TRACE_BLOCK / 64;
    __shared__ double s_rec_t[REC_CAP * TRACE_BLOCK];
    __shared__ uint32_t s_rec_pos[REC_CAP * TRACE_BLOCK], s_rec_cls[REC_CAP * TRACE_BLOCK];
    __shared__ uint32_t s_hist[2][SPAWN_BINS], s_off[2][SPAWN_BINS];
    __shared__ uint32_t s_base[2], s_abort;
    // an earlier level ran out of records: the chunk is retried.  Blocks of this launch set the same
    // bit when they end, so the decision is taken once per block.
    if (threadIdx.x == 0) s_abort = A.counters->error & 1u;
    if (threadIdx.x < SPAWN_BINS) s_hist[0][threadIdx.x] = 0u;
    __syncthreads();
    if (s_abort) return;
    const LevelState lv = A.levels[A.level];
    const uint32_t n_rays = lv.n, rec_base = lv.base, next_rec_base = lv.base + lv.n;
    if (blockIdx.x == 0 && threadIdx.x == 0) A.levels[A.level + 1].base = next_rec_base;
    double big_t_own[BIG ? 1 : CONT_CAP];
    uint32_t big_pos_own[BIG ? 1 : CONT_CAP], big_cls_own[BIG ? 1 : CONT_CAP];
    double* big_t = big_t_own;
    uint32_t *big_pos = big_pos_own, *big_cls = big_cls_own;
    int cont_cap = CONT_CAP;
    if constexpr (BIG) {  // this thread's slice of the launch's global candidate lists
        const size_t slice = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * S.big_cont_cap;
        big_t = S.big_cont_t + slice; big_pos = S.big_cont_pos + slice; big_cls = S.big_cont_cls + slice;
        cont_cap = (int)S.big_cont_cap;
    }
    RecordBuf buf;
    buf.fast = {s_rec_t, s_rec_pos, s_rec_cls, TRACE_BLOCK, REC_CAP, threadIdx.x};  // block-wide bases, this lane's column
    buf.big = {big_t, big_pos, big_cls, 1, cont_cap, 0u};
    const bool record = (S.flags & 1u) != 0;  // scenes with a transparent material
    Work flops{}, flops_shadow{};
    uint32_t err = 0, n_shade = 0, n_spawned = 0, n_shadow = 0;
    // one step = TRACE_BLOCK rays (primary: TRACE_BLOCK / 64 tiles of 8x8 pixels)
    // A lane-walked level with few rays spreads them over more waves (`lanes` active lanes in each, a
    // power of two): lanes of one wave at different kinds of node take turns, so a wave's walk costs
    // the sum of its lanes' branches, and such a level lasts as long as its slowest wave.
    uint32_t lanes = 64;
    if constexpr (LANES_DIVERGE && !PRIMARY) {
#ifndef SPREAD_WAVES
#define SPREAD_WAVES 4096u
#endif
        while (lanes > 1 && n_rays <= (lanes / 2) * SPREAD_WAVES) lanes >>= 1;
    }
    const uint32_t rays_per_step = WAVES * lanes;
    const uint32_t n_steps = PRIMARY ? (A.n_tiles + WAVES - 1) / WAVES : (n_rays + rays_per_step - 1) / rays_per_step;
    uint32_t parity = 0;
    // The ray of the step after this one is requested before this step's shadow walk, so that its
    // trip through memory (behind this step's stores) is over by the time the walk ends.
    auto fetch_ray = [&](uint32_t step_) {
        const uint32_t idx = step_ * rays_per_step + wave_in_block * lanes + lane;
        const uint32_t r = rec_base + ((lane < lanes && idx < n_rays) ? idx : 0);
        Ray f;
        f.o = mk(A.R.ox()[r], A.R.oy()[r], A.R.oz()[r]);
        f.d = mk(A.R.dx()[r], A.R.dy()[r], A.R.dz()[r]);
        return f;
    };
    Ray fetched;
    fetched.o = fetched.d = mk(0.0, 0.0, 0.0);
    if constexpr (!PRIMARY && PREFETCH_RAY) {
        if (blockIdx.x < n_steps) fetched = fetch_ray(blockIdx.x);
    }
    for (uint32_t step = blockIdx.x; step < n_steps; step += gridDim.x, parity ^= 1u) {
    LOOP_FENCE();
    bool on;
    uint32_t rec_idx;  // record of this ray (absolute)
#ifdef RT_PHASE_TIMERS
    unsigned long long phase_t[8];
    phase_t[0] = __builtin_readcyclecounter();
    for (int k = 1; k < 8; k++) phase_t[k] = phase_t[0];
#endif
    Ray ray;
    if constexpr (PRIMARY) {
        // an 8x8 pixel tile per wave
        uint32_t tile = step * WAVES + wave_in_block;
        bool tile_ok = tile < A.n_tiles;
        uint32_t tx = tile_ok ? tile % A.tiles_x : 0, ty = tile_ok ? tile / A.tiles_x : 0;
        uint32_t x = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
        on = tile_ok && x < A.width && ly < A.chunk_rows;
        rec_idx = on ? ly * A.width + x : 0;
        uint32_t gy = global_row(A, A.chunk_row0 + (on ? ly : 0));
        ray = ray_for_pixel(A, on ? x : 0, gy);
        flops.add(F_PRIMARY, on);
        if (on) {
            A.R.dx()[rec_idx] = ray.d.x; A.R.dy()[rec_idx] = ray.d.y; A.R.dz()[rec_idx] = ray.d.z;
        }
    } else {
        uint32_t idx = step * rays_per_step + wave_in_block * lanes + lane;
        on = lane < lanes && idx < n_rays;
        rec_idx = rec_base + (on ? idx : 0);
        if constexpr (PREFETCH_RAY) ray = fetched;
        else ray = fetch_ray(step);
    }

    PHASE(1);
    Walk w = walk_init();
    // closest hit, recording what may sort before it
    walk_tree<HAS_CSG, CULL, PER_LANE, RecordBuf, BIG, HYBRID>(S, ray, on, record, w, buf, flops, true, err);
    PHASE(2);
    if (w.nan_seen && w.arrival >= 2) err |= 4u;
    const bool hit = on && w.pos != NONE;
    const double hit_t = w.t;
    const uint32_t hit_pos = w.pos;
    const uint32_t hit_cls = w.cls;
    const double* mat_rec = S.mat + (hit ? (size_t)w.mat * 8 : 0);
    const double m_reflective = mat_rec[4], m_transparency = mat_rec[5];
    PHASE(3);
    // for refractive hits: n1, n2 from the intersections that sort before the hit
    double n1 = 1.0, n2 = 1.0;
    const bool need_cont = hit && (!approx0(m_transparency) || m_transparency > 0.0);
    if (__ballot(need_cont) != 0ull) {
        const bool small = need_cont && w.n <= REC_CAP;
        N1N2 nn = container_walk_regs(S.stream, S.mat, buf.fast, w.n, small, hit_t, hit_pos, hit_cls, S.class_words <= 1);
        if (need_cont && !small) {  // rare: continue in the lane-private buffer
            if (w.overflow) err |= 2u;
            for (int k = 0; k < REC_CAP; k++)
                buf.big.put(k, buf.fast.T(k), buf.fast.P(k), buf.fast.C(k));
            nn = container_walk_scratch(S.stream, S.mat, big_t, big_pos, big_cls, w.n, hit_t, hit_pos, hit_cls);
        }
        if (need_cont) { n1 = nn.n1; n2 = nn.n2; }
    }
    PHASE(4);
    V3 point = mk(0.0, 0.0, 0.0), normal = mk(0.0, 0.0, 0.0);
    if (hit) {
        HitGeom g = hit_geometry(S.stream, S.xform, S.tri_normals, ray, hit_t, hit_pos);
        point = g.point; normal = g.normal;
    }
    flops.add(F_PREPARE, hit);
    PHASE(5);
    const V3 eye = -ray.d;
    const V3 over = over_point_of(point, normal);

    // reflected_colour / refracted_colour spawn tests (world.rs:129-177)
    const bool spawn_refl = hit && A.depth_remaining > 0 && m_reflective > 0.0;
    bool spawn_refr = false;
    // the refracted direction is normal * a - eye * b (world.rs:171): the two factors are kept, not the vector
    double refr_a = 0.0, refr_b = 0.0;
    const bool refracts = hit && A.depth_remaining > 0 && !approx0(m_transparency);
    flops.add(F_REFRACT, refracts);
    if (refracts) {
        double n_ratio = n1 / n2;
        double cos_i = dot(eye, normal);
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (!(sin2_t > 1.0)) {
            double cos_t = sqrt(1.0 - sin2_t);
            refr_a = n_ratio * cos_i - cos_t;
            refr_b = n_ratio;
            spawn_refr = true;
        }
    }
    double reflectance = -1.0;
    const bool both = hit && m_reflective > 0.0 && m_transparency > 0.0;
    flops.add(F_SCHLICK, both);
    if (both) reflectance = schlick(eye, normal, n1, n2);

    // Append the children to the next level (see above): ranks inside their groups, one scan and one
    // atomic per block, then the stores.
    const uint32_t bin_refl = ((hit_pos >> 1) * 2u) & (SPAWN_BINS - 1u), bin_refr = (bin_refl + 1u) & (SPAWN_BINS - 1u);
    uint32_t child_refl = NONE, child_refr = NONE, rank_refl = 0, rank_refr = 0;
    if (threadIdx.x < SPAWN_BINS) s_hist[parity ^ 1u][threadIdx.x] = 0u;  // for the next step (see the barriers)
    if (spawn_refl) rank_refl = atomicAdd(&s_hist[parity][bin_refl], 1u);
    if (spawn_refr) rank_refr = atomicAdd(&s_hist[parity][bin_refr], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the bins by the first wave
        uint32_t c = threadIdx.x < SPAWN_BINS ? s_hist[parity][threadIdx.x] : 0u, incl = c;
#pragma unroll
        for (int off = 1; off < (int)SPAWN_BINS; off <<= 1) {
            uint32_t o = (uint32_t)__shfl_up((int)incl, off, WAVE);
            if ((int)lane >= off) incl += o;
        }
        if (threadIdx.x < SPAWN_BINS) s_off[parity][threadIdx.x] = incl - c;
        if (threadIdx.x == SPAWN_BINS - 1u) s_base[parity] = incl ? atomicAdd(&A.levels[A.level + 1].n, incl) : 0u;
    }
    __syncthreads();
    PHASE(6);
    if (spawn_refl) {
        uint32_t c = next_rec_base + s_base[parity] + s_off[parity][bin_refl] + rank_refl;
        if (c < A.rec_cap) {
            V3 d_ = reflect(ray.d, normal);  // intersection/mod.rs:105
            A.R.ox()[c] = over.x; A.R.oy()[c] = over.y; A.R.oz()[c] = over.z;
            A.R.dx()[c] = d_.x; A.R.dy()[c] = d_.y; A.R.dz()[c] = d_.z;
            child_refl = c;
        } else err |= 1u;
    }
    if (spawn_refr) {
        uint32_t c = next_rec_base + s_base[parity] + s_off[parity][bin_refr] + rank_refr;
        if (c < A.rec_cap) {
            V3 o_ = under_point_of(point, normal);
            V3 d_ = normal * refr_a - eye * refr_b;  // world.rs:171
            A.R.ox()[c] = o_.x; A.R.oy()[c] = o_.y; A.R.oz()[c] = o_.z;
            A.R.dx()[c] = d_.x; A.R.dy()[c] = d_.y; A.R.dz()[c] = d_.z;
            child_refr = c;
        } else err |= 1u;
    }
    if (on) {
        A.R.hit_pos()[rec_idx] = hit ? hit_pos : NONE;
        if (hit) {
            A.R.child_refl()[rec_idx] = child_refl;
            A.R.child_refr()[rec_idx] = child_refr;
            if (reflectance >= 0.0) A.R.reflectance()[rec_idx] = reflectance;
            A.R.px()[rec_idx] = over.x; A.R.py()[rec_idx] = over.y; A.R.pz()[rec_idx] = over.z;
            A.R.nx()[rec_idx] = normal.x; A.R.ny()[rec_idx] = normal.y; A.R.nz()[rec_idx] = normal.z;
        }
    }
    PHASE(7);

    if constexpr (!PRIMARY && PREFETCH_RAY) {
        if (step + gridDim.x < n_steps) fetched = fetch_ray(step + gridDim.x);
    }
    // the shadow rays of this hit, unless the lights need the chunk's rng order first (k_shadow)
    if constexpr (FUSED) {
        Rng none;
        none.s0 = none.s1 = none.s2 = none.s3 = 0;
        intensities<HAS_CSG, PER_LANE, CULL, BIG, HYBRID>(S, A, hit, rec_idx, over, normal, none, flops_shadow, err, n_shadow);
    }
    n_spawned += (uint32_t)__popcll(__ballot(spawn_refl)) + (uint32_t)__popcll(__ballot(spawn_refr));
    n_shade += (uint32_t)__popcll(__ballot(hit));
#if defined(RT_PHASE_TIMERS) && !defined(RT_PHASE_SHADE)
    {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long t_end = __builtin_readcyclecounter();
        if (lane == 0) {
            for (int k = 1; k < 8; k++) atomicAdd(&A.stats[blockIdx.x % STAT_SHARDS].phase[k], phase_t[k] - phase_t[k - 1]);
            atomicAdd(&A.stats[blockIdx.x % STAT_SHARDS].phase[0], t_end - phase_t[7]);
        }
    }
#endif
    }  // steps

    // statistics: one atomic per wave and counter (the counters are sums over the wave already)
    const unsigned long long f_trace = flops.flops + wave_sum_u32(flops.lane_flops);
    unsigned long long any_err = __ballot(err != 0);
    if constexpr (LANES_DIVERGE) {
        const uint32_t s_nodes = wave_sum_u32(flops.nodes + flops_shadow.nodes), s_tris = wave_sum_u32(flops.tris + flops_shadow.tris);
        if (lane == 0 && (s_nodes | s_tris)) {
            StatShard* st = A.stats + (blockIdx.x % STAT_SHARDS);
            atomicAdd(&st->mesh_nodes, (unsigned long long)s_nodes);
            atomicAdd(&st->mesh_tris, (unsigned long long)s_tris);
        }
    }
    if (lane == 0) {
        StatShard* st = A.stats + (blockIdx.x % STAT_SHARDS);
        if (n_spawned) atomicAdd(&st->secondary, (unsigned long long)n_spawned);
        if (n_shade) atomicAdd(&st->shade, (unsigned long long)n_shade);
        atomicAdd(&st->flop_trace, f_trace);
    }
    if constexpr (FUSED) {
        const unsigned long long f_shadow = flops_shadow.flops + wave_sum_u32(flops_shadow.lane_flops);
        if (lane == 0) {
            StatShard* st = A.stats + (blockIdx.x % STAT_SHARDS);
            if (n_shadow) atomicAdd(&st->shadow, (unsigned long long)n_shadow);
            atomicAdd(&st->flop_shadow, f_shadow);
        }
    }
    if (any_err && err) atomicOr(&A.counters->error, err);
}

// The shadow rays of every record of a chunk, for scenes whose lights need the rng order of the
// whole chunk first (area lights); otherwise k_trace casts them itself.
template <bool HAS_CSG, bool PER_LANE, bool CULL = true, bool BIG = false, bool HYBRID = false>
__global__ __launch_bounds__(256, HAS_CSG ? CSG_WAVES : (PER_LANE || HYBRID) ? LANE_SHADOW_WAVES : SHADOW_WAVES) void k_shadow(DevScene S, LevelArgs A) {
    constexpr bool LANES_DIVERGE = PER_LANE || HYBRID;
    const uint32_t lane = threadIdx.x & 63;
    if (A.counters->error & 1u) return;
    const uint32_t total = A.levels[A.n_levels].base;  // records of the whole chunk
    Work flops{};
    uint32_t err = 0, n_shadow = 0;
    // Without an area light a shade point's lights are independent (no rng stream runs through
    // them): a lane-walked tree takes one (block of records, light) pair per step, which shortens
    // each lane's chain of walks and multiplies the number of waves in flight.
    const bool per_light = LANES_DIVERGE && !A.rng_states && S.n_lights > 1;
    const unsigned long long n_blocks = (total + 255u) / 256u, n_steps = n_blocks * (per_light ? S.n_lights : 1u);
    for (unsigned long long step = blockIdx.x; step < n_steps; step += gridDim.x) {
        LOOP_FENCE();
        const uint32_t light = per_light ? (uint32_t)(step / n_blocks) : 0u;
        const uint32_t first = (uint32_t)(step % n_blocks) * 256u;
        const uint32_t idx = first + threadIdx.x;
        bool on = idx < total;
        const uint32_t rec_idx = on ? idx : 0;
        const uint32_t hp = A.R.hit_pos()[rec_idx];
        const V3 over = mk(A.R.px()[rec_idx], A.R.py()[rec_idx], A.R.pz()[rec_idx]);  // fetched beside hit_pos, not after it
        on = on && hp != NONE;
        Rng rng;
        rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
        if (A.rng_states && on) rng = A.rng_states[A.R.order()[rec_idx]];
        const V3 normal = mk(A.R.nx()[rec_idx], A.R.ny()[rec_idx], A.R.nz()[rec_idx]);
        if (per_light) intensities<HAS_CSG, PER_LANE, CULL, BIG, HYBRID>(S, A, on, rec_idx, over, normal, rng, flops, err, n_shadow, light, light + 1u);
        else intensities<HAS_CSG, PER_LANE, CULL, BIG, HYBRID>(S, A, on, rec_idx, over, normal, rng, flops, err, n_shadow);
    }
    const unsigned long long f_shadow = flops.flops + wave_sum_u32(flops.lane_flops);
    unsigned long long any_err = __ballot(err != 0);
    if constexpr (LANES_DIVERGE) {
        const uint32_t s_nodes = wave_sum_u32(flops.nodes), s_tris = wave_sum_u32(flops.tris);
        if (lane == 0 && (s_nodes | s_tris)) {
            StatShard* st = A.stats + (blockIdx.x % STAT_SHARDS);
            atomicAdd(&st->mesh_nodes, (unsigned long long)s_nodes);
            atomicAdd(&st->mesh_tris, (unsigned long long)s_tris);
        }
    }
    if (lane == 0) {
        StatShard* st = A.stats + (blockIdx.x % STAT_SHARDS);
        if (n_shadow) atomicAdd(&st->shadow, (unsigned long long)n_shadow);
        atomicAdd(&st->flop_shadow, f_shadow);
    }
    if (any_err && err) atomicOr(&A.counters->error, err);
}

// colour_at of every ray of one level, deepest level first: the surface colour, Σ over lights of
// Material::lighting (world.rs:58-70, material.rs:47-87) with pattern_at(object, over_point)
// (pattern/mod.rs:83-87), plus the colours of the spawned rays (world.rs:72-86,144,176).  No tree
// walk here, so the register-hungry pow() stays out of the walking kernels' allocation.  Level 0
// writes the frame buffers instead of the record.
// With TABLES_IN_LDS every block first copies the scene's small tables (materials, pattern nodes,
// the node stream) into LDS: the per-lane lookups hit-leaf -> material -> pattern node -> child
// node are a chain of dependent gathers, and from LDS each link costs ~100 cycles instead of a
// round trip to the L2.  Scenes whose tables do not fit keep reading them from memory.
constexpr uint32_t SHADE_LDS_DOUBLES = 5120;  // 40 KB
struct ShadeTables { uint32_t mat, pat_inv, pat_col, pat_meta, stream, total; };  // offsets in doubles
__host__ __device__ inline ShadeTables shade_tables(uint32_t n_mat, uint32_t n_pat, uint32_t n_units) {
    ShadeTables t;
    t.mat = 0;
    t.pat_inv = t.mat + n_mat * 8;
    t.pat_col = t.pat_inv + n_pat * 12;
    t.pat_meta = t.pat_col + n_pat * 4;
    t.stream = t.pat_meta + (n_pat * 4 + 1) / 2;
    t.total = t.stream + (n_units + 2) * 8;
    return t;
}

template <bool TABLES_IN_LDS>
__global__ __launch_bounds__(256, SHADE_WAVES) void k_shade(DevScene S, LevelArgs A) {
    if (A.counters->error & 1u) return;
    const LevelState lv = A.levels[A.level];
    __shared__ double s_tab[TABLES_IN_LDS ? SHADE_LDS_DOUBLES : 1];
    const double* t_mat = S.mat;
    const uint32_t* t_stream = S.stream;
    PatTables pt;
    pt.pat_meta = S.pat_meta; pt.pat_inv = S.pat_inv; pt.pat_col = S.pat_col;
    if constexpr (TABLES_IN_LDS) {
        const ShadeTables T = shade_tables(S.n_materials, S.n_patterns, S.n_units);
        for (uint32_t k = threadIdx.x; k < S.n_materials * 8; k += 256) s_tab[T.mat + k] = S.mat[k];
        for (uint32_t k = threadIdx.x; k < S.n_patterns * 12; k += 256) s_tab[T.pat_inv + k] = S.pat_inv[k];
        for (uint32_t k = threadIdx.x; k < S.n_patterns * 4; k += 256) s_tab[T.pat_col + k] = S.pat_col[k];
        uint32_t* meta = (uint32_t*)(s_tab + T.pat_meta);
        for (uint32_t k = threadIdx.x; k < S.n_patterns * 4; k += 256) meta[k] = S.pat_meta[k];
        const double* sd = (const double*)S.stream;
        for (uint32_t k = threadIdx.x; k < S.n_units * 8; k += 256) s_tab[T.stream + k] = sd[k];
        __syncthreads();
        t_mat = s_tab + T.mat;
        t_stream = (const uint32_t*)(s_tab + T.stream);
        pt.pat_meta = meta; pt.pat_inv = s_tab + T.pat_inv; pt.pat_col = s_tab + T.pat_col;
    }
    uint32_t flops_sum = 0;
    for (uint32_t first = blockIdx.x * 256; first < lv.n; first += gridDim.x * 256) {
    LOOP_FENCE();
    const uint32_t idx = first + threadIdx.x;
    const bool on = idx < lv.n;
    const uint32_t rec = lv.base + (on ? idx : 0);
    // every column of the record is fetched before the hit test so that the loads overlap
    // (a miss wastes them, but misses are rare and the kernel is bound by load latency)
    const uint32_t hit_pos = on ? A.R.hit_pos()[rec] : NONE;
    const V3 over = mk(A.R.px()[rec], A.R.py()[rec], A.R.pz()[rec]), normal = mk(A.R.nx()[rec], A.R.ny()[rec], A.R.nz()[rec]);
    const V3 eye = -mk(A.R.dx()[rec], A.R.dy()[rec], A.R.dz()[rec]);
    const uint32_t cr_raw = A.R.child_refl()[rec], ct_raw = A.R.child_refr()[rec];
    // the first light's intensity and the Schlick reflectance as well: loaded where they are used they would
    // each add a trip through memory to the end of the record's chain of dependent steps (the reflectance is
    // only meaningful, and only used, for a hit on a material that both reflects and refracts).  Fetching the
    // next record's columns while this one is shaded was measured too: the registers it takes cost more
    // (16.4 -> 18.5 ms per frame).
    const double intensity0 = A.intensity[rec], reflectance_raw = A.R.reflectance()[rec];
#ifdef RT_PHASE_TIMERS
    unsigned long long phase_t[8];
    phase_t[0] = __builtin_readcyclecounter();
    for (int k = 1; k < 8; k++) phase_t[k] = phase_t[0];
#endif
    PHASE(1);
    uint32_t flops = 0;
    V3 colour = mk(0.0, 0.0, 0.0);
    if (hit_pos != NONE) {
        const uint32_t cr = cr_raw, ct = ct_raw;
        // the spawned rays' colours are a scattered read from the level below: requested now,
        // needed only after the lighting
        V3 child_r = mk(0.0, 0.0, 0.0), child_t = mk(0.0, 0.0, 0.0);
        if (cr != NONE) child_r = mk(A.R.r()[cr], A.R.g()[cr], A.R.b()[cr]);
        if (ct != NONE) child_t = mk(A.R.r()[ct], A.R.g()[ct], A.R.b()[ct]);
        const double* mat_rec = t_mat + (size_t)t_stream[(size_t)hit_pos * 16 + 1] * 8;
        const double m_ambient = mat_rec[0], m_diffuse = mat_rec[1], m_specular = mat_rec[2], m_shininess = mat_rec[3];
        const double m_reflective = mat_rec[4], m_transparency = mat_rec[5];
        // a solid root pattern ignores the point (pattern/solid.rs:19): no object-space transform needed
        const uint32_t pat_root = (uint32_t)mat_rec[7];
        V3 pat;
        if (pt.pat_meta[pat_root * 4] == RT_PAT_SOLID) {
            pat = mk(pt.pat_col[pat_root * 4 + 0], pt.pat_col[pat_root * 4 + 1], pt.pat_col[pat_root * 4 + 2]);
        } else {
            V3 object_point = mul_point_v(leaf_inverse(t_stream, S.xform, hit_pos), over);
            pat = pattern_eval(pt, pat_root, object_point);
        }
        PHASE(2);
        PHASE(3);
        Rng rng;
        rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
        if (A.rng_states) rng = A.rng_states[A.R.order()[rec]];
        V3 surface = mk(0.0, 0.0, 0.0);
        flops = F_PATTERN;
        for (uint32_t l = 0; l < S.n_lights; l++) {
            const double intensity = l == 0 ? intensity0 : A.intensity[(size_t)l * A.rec_cap + rec];
            surface = surface + lighting_one(S, l, pat, m_ambient, m_diffuse, m_specular, m_shininess, over, normal, eye, intensity, rng, flops);
        }
        PHASE(4);
        // shade_hit's sum (world.rs:72-86); a ray that was not spawned contributes black
        V3 reflected = mk(0.0, 0.0, 0.0), refracted = mk(0.0, 0.0, 0.0);
        if (cr != NONE) reflected = child_r * m_reflective;
        if (ct != NONE) refracted = child_t * m_transparency;
        const bool both = m_reflective > 0.0 && m_transparency > 0.0;
        colour = combine(surface, reflected, refracted, m_reflective, m_transparency, both ? reflectance_raw : 0.0);
        flops += both ? F_COMBINE + 7 : F_COMBINE;
    }
    PHASE(5);
    if (on) {
        if (A.out_rgb || A.out_rgb8) {
            if (A.out_rgb) {
                double* o = A.out_rgb + (A.out_offset + idx) * 3;
                o[0] = colour.x; o[1] = colour.y; o[2] = colour.z;
            }
            if (A.out_rgb8) {
                uint8_t* o = A.out_rgb8 + (A.out_offset + idx) * 3;
                o[0] = to_u8(colour.x); o[1] = to_u8(colour.y); o[2] = to_u8(colour.z);
            }
        } else {
            A.R.r()[rec] = colour.x; A.R.g()[rec] = colour.y; A.R.b()[rec] = colour.z;
        }
    }
    PHASE(6);
#ifdef RT_PHASE_TIMERS
#ifdef RT_PHASE_SHADE
    if ((threadIdx.x & 63) == 0)
        for (int k = 1; k < 8; k++) atomicAdd(&A.stats[blockIdx.x % STAT_SHARDS].phase[k], phase_t[k] - phase_t[k - 1]);
#endif
#endif
    flops_sum += flops;
    }  // records
    const uint32_t flops = flops_sum;
    uint32_t s_lo = wave_sum_u32(flops & 0xFFFFu), s_hi = wave_sum_u32(flops >> 16);
    if ((threadIdx.x & 63) == 0 && (s_lo | s_hi))
        atomicAdd(&A.stats[blockIdx.x % STAT_SHARDS].flop_light, (unsigned long long)s_lo + ((unsigned long long)s_hi << 16));
}

// Start of a chunk: level 0 holds n0 rays in records [0, n0); everything else is empty.
__global__ void k_chunk_begin(LevelState* levels, uint32_t n_entries, uint32_t n0, Counters* counters) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_entries) { levels[i].n = i == 0 ? n0 : 0u; levels[i].base = 0u; }
    if (i == 0) counters->error = 0u;
}
// End of a chunk: what the host needs to know about it, with the statistics lines added up and
// cleared for the next chunk (a chunk that is retried must not count twice).  One block of
// STAT_SHARDS threads.
__global__ __launch_bounds__(STAT_SHARDS) void k_chunk_end(const LevelState* levels, uint32_t n_levels,
                                                           const Counters* counters, StatShard* stats, ChunkStatus* out) {
    __shared__ unsigned long long sums[STAT_SHARDS / 64][8];
    __shared__ uint32_t used_by_wave[STAT_SHARDS / 64];
    StatShard* st = stats + threadIdx.x;
    unsigned long long v[8] = {st->secondary, st->shadow, st->shade, st->flop_trace, st->flop_shadow, st->flop_light,
                               st->mesh_nodes, st->mesh_tris};
    st->secondary = 0; st->shadow = 0; st->shade = 0; st->flop_trace = 0; st->flop_shadow = 0; st->flop_light = 0;
    st->mesh_nodes = 0; st->mesh_tris = 0;
    uint32_t used = 0;  // one more than the deepest level that held a ray
    for (uint32_t l = threadIdx.x; l < n_levels; l += STAT_SHARDS)
        if (levels[l].n) used = l + 1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v[k], off, WAVE), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v[k] >> 32), off, WAVE);
            v[k] += ((unsigned long long)hi << 32) | lo;
        }
        used = max(used, (uint32_t)__shfl_xor((int)used, off, WAVE));
    }
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 8; k++) sums[threadIdx.x >> 6][k] = v[k];
        used_by_wave[threadIdx.x >> 6] = used;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned long long total[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    used = 0;
    for (uint32_t w = 0; w < STAT_SHARDS / 64; w++) {
        for (int k = 0; k < 8; k++) total[k] += sums[w][k];
        used = max(used, used_by_wave[w]);
    }
    out->error = counters->error;
    out->total_recs = levels[n_levels].base;
    out->levels = used;
    out->pad = 0;
    out->secondary = total[0]; out->shadow = total[1]; out->shade = total[2];
    out->flop_trace = total[3]; out->flop_shadow = total[4]; out->flop_light = total[5];
    out->mesh_nodes = total[6]; out->mesh_tris = total[7];
}

// ---- deferred shading order (area lights): the reference consumes one rng stream per row, pixel
// after pixel, and inside a pixel depth-first with the reflected subtree before the refracted one
// (camera.rs:143-157, world.rs:60-74).
// (1) bottom-up: order = number of shade points in the subtree
__global__ __launch_bounds__(256) void k_order_sizes(Recs R, uint32_t base, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rec = base + i;
    uint32_t s = 0;
    if (R.hit_pos()[rec] != NONE) {
        s = 1;
        const uint32_t cr = R.child_refl()[rec], ct = R.child_refr()[rec];
        if (cr != NONE) s += R.order()[cr];
        if (ct != NONE) s += R.order()[ct];
    }
    R.order()[rec] = s;
}
// (2) one block per row: running offsets of the row's pixels; row_total[row] = shade points in the row
__global__ __launch_bounds__(256) void k_order_rows(Recs R, uint32_t width, uint32_t rows, uint32_t* pixel_base,
                                                    uint32_t* row_total) {
    const uint32_t row = blockIdx.x;
    if (row >= rows) return;
    __shared__ uint32_t s_sum[256];
    // every thread takes a run of consecutive pixels; the runs' sums are scanned through LDS
    const uint32_t per = (width + 255) / 256;
    const uint32_t x0 = min(threadIdx.x * per, width), x1 = min(x0 + per, width);
    uint32_t mine = 0;
    for (uint32_t x = x0; x < x1; x++) mine += R.order()[(size_t)row * width + x];
    s_sum[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t add = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0u;
        __syncthreads();
        s_sum[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t acc = s_sum[threadIdx.x] - mine;
    for (uint32_t x = x0; x < x1; x++) {
        pixel_base[(size_t)row * width + x] = acc;
        acc += R.order()[(size_t)row * width + x];
    }
    if (threadIdx.x == 255) row_total[row] = s_sum[255];
}
// (3) top-down: order = position of the shade point in the chunk's state array
__global__ __launch_bounds__(256) void k_order_assign(Recs R, uint32_t base, uint32_t n, uint32_t width,
                                                      const uint32_t* pixel_base, const uint32_t* row_base, int level0) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rec = base + i;
    if (R.hit_pos()[rec] == NONE) return;
    uint32_t mine = R.order()[rec];  // level 0: still the subtree size; deeper: already the position
    if (level0) mine = row_base[i / width] + pixel_base[i];
    uint32_t next = mine + 1;
    const uint32_t cr = R.child_refl()[rec], ct = R.child_refr()[rec];
    if (cr != NONE && R.hit_pos()[cr] != NONE) { uint32_t sz = R.order()[cr]; R.order()[cr] = next; next += sz; }
    if (ct != NONE && R.hit_pos()[ct] != NONE) R.order()[ct] = next;
    R.order()[rec] = mine;
}
// (4) the generator state at every shade point.  Shade point k of a row starts k * draws steps
// after the row's seed, and the state transition of xoshiro256 is linear over GF(2) (the ++ only
// scrambles the output), so that state is T^(draws * k) applied to the seeded state, with
// precomputed matrices jump[b] = T^(draws * 2^b).  A matrix is stored by columns, 4 words each;
// applying it XORs the columns selected by the state's bits.  A few blocks per row (stepping the
// generator along the row instead takes one lane 15 ms for 320 000 steps).  A wave works on 64
// consecutive shade points with one state that all its lanes share: every lane XORs the four
// columns of "its" state bits and a butterfly over the wave folds the 64 partial sums.  It jumps to
// the first point by the set bits of its index, then walks on by jump[0] (kept in LDS), leaving
// the state with the lane of the same number.
constexpr uint32_t RNG_JUMPS = 26;  // shade points per row < 2^26
DEV void rng_apply(const unsigned long long* m, unsigned long long* u, uint32_t lane) {
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
    for (int wd = 0; wd < 4; wd++) {
        const unsigned long long mask = 0ull - ((u[wd] >> lane) & 1ull);
        const unsigned long long* c = m + (size_t)(wd * 64 + lane) * 4;
        p0 ^= c[0] & mask; p1 ^= c[1] & mask; p2 ^= c[2] & mask; p3 ^= c[3] & mask;
    }
    unsigned long long p[4] = {p0, p1, p2, p3};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int lo = __shfl_xor((int)(uint32_t)p[j], off, WAVE), hi = __shfl_xor((int)(uint32_t)(p[j] >> 32), off, WAVE);
            p[j] ^= ((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) u[j] = p[j];
}
__global__ __launch_bounds__(256) void k_rng_rows(Rng* states, const uint32_t* row_base, const uint32_t* row_total,
                                                  const unsigned long long* seeds, uint32_t rows, uint32_t chunk_row0,
                                                  uint32_t part_block, uint32_t part_rank, uint32_t part_world,
                                                  const unsigned long long* jump) {
    const uint32_t row = blockIdx.x;
    if (row >= rows) return;
    __shared__ unsigned long long s_step[1024];  // jump[0]: from one shade point to the next
    for (uint32_t k = threadIdx.x; k < 1024; k += blockDim.x) s_step[k] = jump[k];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t local = chunk_row0 + row, gy = local;
    if (part_world > 1) gy = ((local / part_block) * part_world + part_rank) * part_block + local % part_block;
    const Rng seeded = rng_seed(seeds[gy]);
    const uint32_t n = row_total[row];
    Rng* out = states + row_base[row];
    // the row's groups of 64 points are dealt over the blocks of the row (gridDim.y) and their waves
    for (uint32_t kw = (blockIdx.y * (blockDim.x / 64) + wave) * 64; kw < n; kw += gridDim.y * blockDim.x) {
        unsigned long long u[4] = {seeded.s0, seeded.s1, seeded.s2, seeded.s3};
        for (uint32_t b = 6; b < RNG_JUMPS; b++)
            if ((kw >> b) & 1u) rng_apply(jump + (size_t)b * 1024, u, lane);
        unsigned long long mine[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < 64; i++) {
            if (lane == i) { mine[0] = u[0]; mine[1] = u[1]; mine[2] = u[2]; mine[3] = u[3]; }
            if (kw + i + 1 >= n) break;
            rng_apply(s_step, u, lane);
        }
        if (kw + lane < n) {
            Rng r;
            r.s0 = mine[0]; r.s1 = mine[1]; r.s2 = mine[2]; r.s3 = mine[3];
            out[kw + lane] = r;
        }
    }
}

// rt_probe: prepare_computations of the closest hit, for parity checks.
template <bool HAS_CSG>
__global__ __launch_bounds__(256) void k_probe(DevScene S, uint32_t n, Recs R, double* out, Counters* counters) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool on = idx < n;
    const uint32_t q = on ? idx : 0;
    Ray ray;
    ray.o = mk(R.ox()[q], R.oy()[q], R.oz()[q]);
    ray.d = mk(R.dx()[q], R.dy()[q], R.dz()[q]);
    Work flops{};
    uint32_t err = 0;
    V3 point = mk(0.0, 0.0, 0.0), normal = mk(0.0, 0.0, 0.0);
    int inside = 0;
    double n1 = 1.0, n2 = 1.0;
    __shared__ double s_rec_t[REC_CAP * 256];
    __shared__ uint32_t s_rec_pos[REC_CAP * 256], s_rec_cls[REC_CAP * 256];
    double big_t[CONT_CAP];
    uint32_t big_pos[CONT_CAP], big_cls[CONT_CAP];
    RecordBuf buf;
    buf.fast = {s_rec_t, s_rec_pos, s_rec_cls, 256, REC_CAP, threadIdx.x};
    buf.big = {big_t, big_pos, big_cls, 1, CONT_CAP, 0u};
    Walk w = walk_init();
    bool hit = false, need = false;
    double hit_t = 0.0;
    uint32_t hit_pos = NONE, n_list = 0;
    double sh[4] = {0.0, 0.0, 0.0, 0.0};
    uint32_t n_l = S.n_lights < 4 ? S.n_lights : 4;
    for (uint32_t job = 0; job < 1 + n_l; job++) {
        Ray jr = ray;
        bool jon = on;
        double dist = 0.0;
        if (job >= 1) {
            uint32_t l = job - 1;
            V3 lpos = mk(cld(S.light, l * 12 + 0), cld(S.light, l * 12 + 1), cld(S.light, l * 12 + 2));
            V3 over = over_point_of(point, normal);
            V3 vec = lpos - over;
            dist = sqrt(dot(vec, vec));
            jr.o = over;
            jr.d = normalise(vec);
            jon = hit;
            w = walk_init();
            traverse<HAS_CSG, true, false>(S, jr, jon, false, w, NoRecord{}, flops, false, err);
        } else {
            traverse<HAS_CSG, false, false>(S, jr, jon, true, w, buf, flops, false, err);
        }
        if (job == 0) {
            if (w.nan_seen && w.arrival >= 2) err |= 4u;
            hit = on && w.pos != NONE;
            hit_t = w.t; hit_pos = w.pos; n_list = w.arrival;
            if (hit) {
                HitGeom g = hit_geometry(S.stream, S.xform, S.tri_normals, ray, hit_t, hit_pos);
                point = g.point; normal = g.normal; inside = g.inside;
            }
            double tr = hit ? S.mat[(size_t)S.stream[(size_t)hit_pos * 16 + 1] * 8 + 5] : 0.0;
            need = hit && (!approx0(tr) || tr > 0.0);
            if (__ballot(need) != 0ull) {  // n1 / n2 from the recorded candidates
                const uint32_t hit_cls = need ? S.stream[(size_t)hit_pos * 16 + 3] : 0u;
                const bool small = need && w.n <= REC_CAP;
                N1N2 nn = container_walk_regs(S.stream, S.mat, buf.fast, w.n, small, hit_t, hit_pos, hit_cls, S.class_words <= 1);
                if (need && !small) {
                    if (w.overflow) err |= 2u;
                    for (int k = 0; k < REC_CAP; k++)
                        buf.big.put(k, buf.fast.T(k), buf.fast.P(k), buf.fast.C(k));
                    nn = container_walk_scratch(S.stream, S.mat, big_t, big_pos, big_cls, w.n, hit_t, hit_pos, hit_cls);
                }
                if (need) { n1 = nn.n1; n2 = nn.n2; }
            }
        } else {
            bool s = false;
            if (hit && w.pos != NONE) s = ((S.stream[(size_t)w.pos * 16] >> 8) & 1u) && w.t < dist;
            double v = s ? 1.0 : 0.0;
            uint32_t l = job - 1;
            if (l == 0) sh[0] = v; else if (l == 1) sh[1] = v; else if (l == 2) sh[2] = v; else sh[3] = v;
        }
    }
    if (on) {
        double* o = out + (size_t)idx * 32;
        for (int k = 0; k < 32; k++) o[k] = 0.0;
        o[0] = -1.0;
        o[24] = (double)n_list;
        if (hit) {
            V3 over = over_point_of(point, normal), under = under_point_of(point, normal);
            V3 eye = -ray.d, refl = reflect(ray.d, normal);
            o[0] = (double)S.stream[(size_t)hit_pos * 16 + 2]; o[1] = hit_t;
            o[2] = point.x; o[3] = point.y; o[4] = point.z;
            o[5] = over.x; o[6] = over.y; o[7] = over.z;
            o[8] = under.x; o[9] = under.y; o[10] = under.z;
            o[11] = eye.x; o[12] = eye.y; o[13] = eye.z;
            o[14] = normal.x; o[15] = normal.y; o[16] = normal.z;
            o[17] = inside ? 1.0 : 0.0;
            o[18] = refl.x; o[19] = refl.y; o[20] = refl.z;
            // n1 / n2 come from the container walk for refractive hits; opaque hits report 1.0
            o[21] = n1; o[22] = n2;
            o[23] = schlick(eye, normal, n1, n2);
            o[25] = sh[0]; o[26] = sh[1]; o[27] = sh[2]; o[28] = sh[3];
        }
    }
    if (err) atomicOr(&counters->error, err);
}

// rt_lighting: Material::lighting (material.rs:47-87) for caller-supplied shade points, one per
// thread.  Per point 12 f64: unit position of the object's leaf, light index, point, eye vector,
// normal vector, intensity.  The rng of point i is seeded with seeds[i] and stands at lighting's
// own first draw, as in a direct call of the reference's function.
__global__ __launch_bounds__(64) void k_lighting(DevScene S, uint32_t n, const double* in, const unsigned long long* seeds,
                                                 double* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* q = in + (size_t)i * 12;
    const uint32_t pos = (uint32_t)q[0], light = (uint32_t)q[1];
    const V3 point = mk(q[2], q[3], q[4]), eye = mk(q[5], q[6], q[7]), normal = mk(q[8], q[9], q[10]);
    const double* mat_rec = S.mat + (size_t)S.stream[(size_t)pos * 16 + 1] * 8;
    PatTables pt;
    pt.pat_meta = S.pat_meta; pt.pat_inv = S.pat_inv; pt.pat_col = S.pat_col;
    const V3 object_point = mul_point_v(leaf_inverse(S.stream, S.xform, pos), point);
    const V3 pat = pattern_eval(pt, (uint32_t)mat_rec[7], object_point);
    Rng rng = rng_seed(seeds ? seeds[i] : 0ull);
    uint32_t flops = 0;
    const V3 c = lighting_one(S, light, pat, mat_rec[0], mat_rec[1], mat_rec[2], mat_rec[3], point, normal, eye, q[11], rng, flops, false);
    out[(size_t)i * 3 + 0] = c.x; out[(size_t)i * 3 + 1] = c.y; out[(size_t)i * 3 + 2] = c.z;
}

// rt_intensity_at: Lightable::intensity_at (light/point.rs:27-38, light/area.rs:75-90) of one light
// for n points, one call after the other on ONE generator, by one wave (the walk is the wave's; only
// lane 0 holds a point).  The reference's walk, without any device-only shortcut.
__global__ __launch_bounds__(64) void k_intensity_at(DevScene S, uint32_t light, uint32_t n, const double* points,
                                                     unsigned long long seed, double* out, Counters* counters) {
    LevelArgs A{};
    A.intensity = out - (size_t)light * n;  // intensities() writes light l of record i at [l * rec_cap + i]
    A.rec_cap = n;
    const uint32_t samples = clu(S.light_meta, light * 4 + 1) * clu(S.light_meta, light * 4 + 2);
    const bool is_area = clu(S.light_meta, light * 4 + 0) == RT_LIGHT_AREA;
    Rng rng = rng_seed(seed);
    Work flops{};
    uint32_t err = 0, n_shadow = 0;
    for (uint32_t i = 0; i < n; i++) {
        const V3 p = mk(points[(size_t)i * 3 + 0], points[(size_t)i * 3 + 1], points[(size_t)i * 3 + 2]);
        intensities<true, false, false>(S, A, threadIdx.x == 0, i, p, mk(0.0, 0.0, 0.0), rng, flops, err, n_shadow, light, light + 1u);
        if (is_area) rng_skip(rng, 2 * samples);  // what positions() drew (light/area.rs:57-67)
    }
    if (err) atomicOr(&counters->error, err);
}

// ---------------------------------------------------------------- static scratch (rt_colour_at_noalloc)
// Device memory for callers that must not allocate: the record arrays, level table, counters and
// statistics lines of one small batch of rays are carved out of this array instead of a Workspace's
// hipMalloc'ed buffers.  Every GPU the library runs on holds its own copy, at its own address, until
// the process ends; the name is only a host-side handle, the address comes from hipGetSymbolAddress.
}  // namespace
constexpr size_t SCRATCH_BYTES = (size_t)64 << 20;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
namespace {
constexpr int SCRATCH_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[SCRATCH_MAX_DEVICES] = {};  // per device: where that GPU's copy of `scratch` lives

// rays given as n * 6 f64 (origin, direction) into the record columns the first level reads
__global__ __launch_bounds__(256) void k_load_rays(const double* rays, uint32_t n, Recs R) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = rays + (size_t)i * 6;
    R.ox()[i] = r[0]; R.oy()[i] = r[1]; R.oz()[i] = r[2];
    R.dx()[i] = r[3]; R.dy()[i] = r[4]; R.dz()[i] = r[5];
}
// a fill kernel instead of hipMemsetAsync: safe to record into a caller's graph
__global__ __launch_bounds__(256) void k_zero_words(unsigned long long* p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ull;
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

// Address of the current device's copy of `scratch`, looked up once per device with that device current.
int scratch_of_current_device(unsigned char** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) return fail(RT_ERR_INVALID, "device ordinal beyond the static scratch table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress returned no address for the static scratch");
        g_scratch_ptr[dev] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[dev];
    return RT_OK;
}

// World-space bounding sphere {centre, radius, radius^2} of a bounded leaf, from rows 0-2 of its
// inverse transformation; false for unbounded shapes (plane, untruncated cylinder / cone) and
// degenerate matrices.  The radius is padded: culling only has to be conservative.
bool bounding_sphere(uint32_t shape, const double* inv, double mn, double mx, double* out) {
    if (shape == RT_PLANE || shape == RT_TRIANGLE) return false;
    double lo[3] = {-1.0, -1.0, -1.0}, hi[3] = {1.0, 1.0, 1.0};
    if (shape == RT_CYLINDER || shape == RT_CONE) {
        if (!std::isfinite(mn) || !std::isfinite(mx)) return false;
        double rad = shape == RT_CYLINDER ? 1.0 : std::max(std::fabs(mn), std::fabs(mx));
        lo[0] = lo[2] = -rad; hi[0] = hi[2] = rad;
        lo[1] = std::min(mn, mx); hi[1] = std::max(mn, mx);
    }
    // forward matrix A (3x3) and translation c: world = A * object + c
    const double a = inv[0], b = inv[1], c = inv[2], e = inv[4], f = inv[5], g = inv[6], k = inv[8], l = inv[9], m = inv[10];
    const double det = a * (f * m - g * l) - b * (e * m - g * k) + c * (e * l - f * k);
    if (!(std::fabs(det) > 1e-300) || !std::isfinite(det)) return false;
    const double A[3][3] = {{(f * m - g * l) / det, (c * l - b * m) / det, (b * g - c * f) / det},
                            {(g * k - e * m) / det, (a * m - c * k) / det, (c * e - a * g) / det},
                            {(e * l - f * k) / det, (b * k - a * l) / det, (a * f - b * e) / det}};
    const double t[3] = {inv[3], inv[7], inv[11]};
    double centre_obj[3] = {(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, (lo[2] + hi[2]) / 2};
    double centre[3];
    for (int r = 0; r < 3; r++) {
        centre[r] = 0.0;
        for (int q = 0; q < 3; q++) centre[r] += A[r][q] * (centre_obj[q] - t[q]);  // A * (p - t') with t' = inv translation
    }
    // world = A * (object - inv_translation_in_object_terms): inv is X_obj = I3 * X_world + t, so X_world = A * (X_obj - t)
    double radius = 0.0;
    if (shape == RT_SPHERE) {
        // largest singular value of A by power iteration on A^T A, from three starting vectors
        double sigma2 = 0.0;
        const double starts[3][3] = {{1.0, 0.3, 0.1}, {0.1, 1.0, 0.3}, {0.3, 0.1, 1.0}};
        for (const auto& s0 : starts) {
            double v[3] = {s0[0], s0[1], s0[2]};
            for (int it = 0; it < 64; it++) {
                double n0 = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                if (!(n0 > 0.0)) break;
                for (int r = 0; r < 3; r++) v[r] /= n0;
                double w[3] = {0, 0, 0}, u[3] = {0, 0, 0};
                for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) w[r] += A[r][q] * v[q];
                for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) u[r] += A[q][r] * w[q];
                double n = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
                sigma2 = std::max(sigma2, n);
                for (int r = 0; r < 3; r++) v[r] = u[r];
            }
        }
        // the iteration converges from below; Frobenius bounds it from above
        double fro = 0.0;
        for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) fro += A[r][q] * A[r][q];
        radius = std::min(std::sqrt(sigma2) * 1.01, std::sqrt(fro));
        if (!(radius > 0.0)) radius = std::sqrt(fro);
    } else {
        for (int corner = 0; corner < 8; corner++) {
            double p[3] = {(corner & 1 ? hi[0] : lo[0]) - centre_obj[0], (corner & 2 ? hi[1] : lo[1]) - centre_obj[1],
                           (corner & 4 ? hi[2] : lo[2]) - centre_obj[2]};
            double q2 = 0.0;
            for (int r = 0; r < 3; r++) {
                double x = A[r][0] * p[0] + A[r][1] * p[1] + A[r][2] * p[2];
                q2 += x * x;
            }
            radius = std::max(radius, std::sqrt(q2));
        }
    }
    radius = radius * (1.0 + 1e-6) + 1e-9;
    if (!std::isfinite(radius) || !std::isfinite(centre[0]) || !std::isfinite(centre[1]) || !std::isfinite(centre[2])) return false;
    out[0] = centre[0]; out[1] = centre[1]; out[2] = centre[2];
    out[3] = radius; out[4] = radius * radius; out[5] = 0.0;
    return true;
}

// ---- jump matrices of the area lights' generator (see k_rng_rows)
struct Gf2Matrix { uint64_t col[256][4]; };   // column j = image of the state with only bit j set
void gf2_apply(const Gf2Matrix& m, const uint64_t* v, uint64_t* out) {
    uint64_t a[4] = {0, 0, 0, 0};
    for (int j = 0; j < 256; j++)
        if ((v[j >> 6] >> (j & 63)) & 1ull)
            for (int w = 0; w < 4; w++) a[w] ^= m.col[j][w];
    for (int w = 0; w < 4; w++) out[w] = a[w];
}
void gf2_mul(const Gf2Matrix& a, const Gf2Matrix& b, Gf2Matrix& out) {  // out = a * b (first b, then a)
    Gf2Matrix r;
    for (int j = 0; j < 256; j++) gf2_apply(a, b.col[j], r.col[j]);
    out = r;
}
// T^(draws * 2^b) for b = 0 .. RNG_JUMPS-1, 1024 words each, where T is one step of xoshiro256's
// state (rand_xoshiro 0.6.0: s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= s1_old << 17; s3 = rotl(s3, 45)).
std::vector<unsigned long long> rng_jump_matrices(uint64_t draws) {
    auto step = [](uint64_t* s) {
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = (s[3] << 45) | (s[3] >> 19);
    };
    std::vector<Gf2Matrix> m(3);  // 0: T^(2^i), 1: accumulated power, 2: scratch
    Gf2Matrix &sq = m[0], &acc = m[1];
    for (int j = 0; j < 256; j++) {
        uint64_t e[4] = {0, 0, 0, 0};
        e[j >> 6] = 1ull << (j & 63);
        for (int w = 0; w < 4; w++) acc.col[j][w] = e[w];  // identity
        step(e);
        for (int w = 0; w < 4; w++) sq.col[j][w] = e[w];
    }
    for (uint64_t d = draws; d; d >>= 1) {
        if (d & 1) gf2_mul(sq, acc, acc);
        if (d >> 1) gf2_mul(sq, sq, sq);
    }
    std::vector<unsigned long long> out((size_t)RNG_JUMPS * 1024);
    for (uint32_t b = 0; b < RNG_JUMPS; b++) {
        for (int j = 0; j < 256; j++)
            for (int w = 0; w < 4; w++) out[(size_t)b * 1024 + (size_t)j * 4 + w] = acc.col[j][w];
        if (b + 1 < RNG_JUMPS) gf2_mul(acc, acc, acc);
    }
    return out;
}

// ---- the meshes' own hierarchies (see mesh_walk)
struct MeshTri { double lo[3], hi[3], c[3]; uint32_t pos; };
struct BinNode { double lo[2][3], hi[2][3]; uint32_t c[2]; };  // a node of the binary tree the wide one is made from
struct MeshBuild { std::vector<BinNode> bin; std::vector<float> nodes; std::vector<double> tris; const uint32_t* stream; };
#ifndef MESH_LEAF_TRIS_N
#define MESH_LEAF_TRIS_N 4
#endif
constexpr uint32_t MESH_MIN_TRIS = 4, MESH_LEAF_TRIS = MESH_LEAF_TRIS_N;

void mesh_bounds(const MeshTri* t, size_t n, double* lo, double* hi) {
    for (int a = 0; a < 3; a++) { lo[a] = HUGE_VAL; hi[a] = -HUGE_VAL; }
    for (size_t k = 0; k < n; k++)
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], t[k].lo[a]); hi[a] = std::max(hi[a], t[k].hi[a]); }
}
double half_area(const double* lo, const double* hi) {
    double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * y + y * z + z * x;
}
// Builds the subtree over t[b, e) and returns its reference.  Splits by the surface-area heuristic
// over 16 centroid bins per axis; below `depth` 32 a split that fails to separate anything, and
// every split beyond, halves the range at the median, which bounds the depth by 32 + log2(n).
uint32_t mesh_build(std::vector<MeshTri>& t, size_t b, size_t e, int depth, MeshBuild& out) {
    const size_t n = e - b;
    if (n <= MESH_LEAF_TRIS) {
        uint32_t at = (uint32_t)(out.tris.size() / 10);
        std::sort(t.begin() + b, t.begin() + e, [](const MeshTri& x, const MeshTri& y) { return x.pos < y.pos; });
        for (size_t k = b; k < e; k++) {
            const double* v = (const double*)(out.stream + (size_t)t[k].pos * 16) + 4;  // p1, e1, e2
            out.tris.insert(out.tris.end(), v, v + 9);
            double tail = 0.0;
            std::memcpy(&tail, &t[k].pos, 4);
            out.tris.push_back(tail);
        }
        return BVH_LEAF | ((uint32_t)(n - 1) << 27) | at;
    }
    double clo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, chi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (size_t k = b; k < e; k++)
        for (int a = 0; a < 3; a++) { clo[a] = std::min(clo[a], t[k].c[a]); chi[a] = std::max(chi[a], t[k].c[a]); }
    size_t mid = b;
    bool split = false;
    if (depth < 32) {
        constexpr int BINS = 16;
        double best = HUGE_VAL, best_edge = 0.0;
        int best_axis = -1;
        for (int a = 0; a < 3; a++) {
            const double ext = chi[a] - clo[a];
            if (!(ext > 0.0)) continue;
            size_t cnt[BINS] = {0};
            double blo[BINS][3], bhi[BINS][3];
            for (int i = 0; i < BINS; i++) for (int q = 0; q < 3; q++) { blo[i][q] = HUGE_VAL; bhi[i][q] = -HUGE_VAL; }
            for (size_t k = b; k < e; k++) {
                int bin = std::min(BINS - 1, std::max(0, (int)((t[k].c[a] - clo[a]) / ext * BINS)));
                cnt[bin]++;
                for (int q = 0; q < 3; q++) { blo[bin][q] = std::min(blo[bin][q], t[k].lo[q]); bhi[bin][q] = std::max(bhi[bin][q], t[k].hi[q]); }
            }
            double rarea[BINS];
            size_t rcnt[BINS];
            double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
            size_t c = 0;
            for (int i = BINS - 1; i > 0; i--) {
                for (int q = 0; q < 3; q++) { lo[q] = std::min(lo[q], blo[i][q]); hi[q] = std::max(hi[q], bhi[i][q]); }
                c += cnt[i];
                rarea[i] = c ? half_area(lo, hi) : 0.0;
                rcnt[i] = c;
            }
            for (int q = 0; q < 3; q++) { lo[q] = HUGE_VAL; hi[q] = -HUGE_VAL; }
            c = 0;
            for (int i = 0; i + 1 < BINS; i++) {
                for (int q = 0; q < 3; q++) { lo[q] = std::min(lo[q], blo[i][q]); hi[q] = std::max(hi[q], bhi[i][q]); }
                c += cnt[i];
                if (c == 0 || rcnt[i + 1] == 0) continue;
                double cost = half_area(lo, hi) * (double)c + rarea[i + 1] * (double)rcnt[i + 1];
                if (cost < best) { best = cost; best_axis = a; best_edge = clo[a] + ext * (double)(i + 1) / BINS; }
            }
        }
        if (best_axis >= 0) {
            const int a = best_axis;
            const double ext = chi[a] - clo[a];
            auto it = std::partition(t.begin() + b, t.begin() + e, [&](const MeshTri& x) {
                int bin = std::min(15, std::max(0, (int)((x.c[a] - clo[a]) / ext * 16)));
                return clo[a] + ext * (double)(bin + 1) / 16 <= best_edge;
            });
            mid = (size_t)(it - t.begin());
            split = mid > b && mid < e;
        }
    }
    if (!split) {
        int a = 0;
        for (int q = 1; q < 3; q++) if (chi[q] - clo[q] > chi[a] - clo[a]) a = q;
        mid = b + n / 2;
        std::nth_element(t.begin() + b, t.begin() + mid, t.begin() + e,
                         [a](const MeshTri& x, const MeshTri& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.pos < y.pos); });
    }
    const size_t node = out.bin.size();
    out.bin.emplace_back();
    const uint32_t c0 = mesh_build(t, b, mid, depth + 1, out), c1 = mesh_build(t, mid, e, depth + 1, out);
    BinNode& nd = out.bin[node];
    for (int c = 0; c < 2; c++) {
        double lo[3], hi[3];
        mesh_bounds(&t[c ? mid : b], c ? e - mid : mid - b, lo, hi);
        for (int q = 0; q < 3; q++) {
            const double pad = 1e-9 * std::max(std::max(std::fabs(lo[q]), std::fabs(hi[q])), hi[q] - lo[q]) + 1e-30;
            nd.lo[c][q] = lo[q] - pad;
            nd.hi[c][q] = hi[q] + pad;
        }
    }
    nd.c[0] = c0; nd.c[1] = c1;
    return (uint32_t)node;
}

// f32 not above / not below a double
float round_down(double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -HUGE_VALF) : f; }
float round_up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, HUGE_VALF) : f; }

// The wide node for binary node `ref`: its two children, the one with the largest box replaced by its own
// two children until there are four (or only leaves are left).  32 words: lo.x[4] lo.y[4] lo.z[4] hi.x[4]
// hi.y[4] hi.z[4] (f32, rounded outwards), the four references (NONE: no child), four unused.
// Returns the reference for the walk and reports the depth of the wide tree.
uint32_t mesh_collapse(MeshBuild& out, uint32_t ref, int depth, int& max_depth) {
    if (ref & BVH_LEAF) return ref;
    max_depth = std::max(max_depth, depth + 1);
    struct Kid { double lo[3], hi[3]; uint32_t ref; };
    Kid kids[4];
    int n = 0;
    auto take = [&](uint32_t from, int slot) {
        for (int c = 0; c < 2; c++) {
            Kid k;
            for (int q = 0; q < 3; q++) { k.lo[q] = out.bin[from].lo[c][q]; k.hi[q] = out.bin[from].hi[c][q]; }
            k.ref = out.bin[from].c[c];
            if (c == 0) kids[slot] = k; else kids[n++] = k;
        }
    };
    n = 1;
    take(ref, 0);
    while (n < 4) {
        int widest = -1;
        for (int k = 0; k < n; k++)
            if (!(kids[k].ref & BVH_LEAF) && (widest < 0 || half_area(kids[k].lo, kids[k].hi) > half_area(kids[widest].lo, kids[widest].hi)))
                widest = k;
        if (widest < 0) break;
        take(kids[widest].ref, widest);
    }
    const size_t node = out.nodes.size() / 32;
    out.nodes.resize(out.nodes.size() + 32, 0.0f);
    uint32_t refs[4] = {NONE, NONE, NONE, NONE};
    for (int k = 0; k < n; k++) refs[k] = mesh_collapse(out, kids[k].ref, depth + 1, max_depth);
    float* nd = &out.nodes[node * 32];
    for (int k = 0; k < n; k++)
        for (int q = 0; q < 3; q++) {
            nd[q * 4 + k] = round_down(kids[k].lo[q]);
            nd[12 + q * 4 + k] = round_up(kids[k].hi[q]);
        }
    std::memcpy(nd + 24, refs, sizeof refs);
    return (uint32_t)node;
}

template <class T>
int upload(const std::vector<T>& host, T** dev) {
    size_t bytes = (host.size() ? host.size() : 1) * sizeof(T);
    HIP_TRY(hipMalloc((void**)dev, bytes));
    if (!host.empty()) HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

template <class T>
int grow(T** buf, size_t* cap, size_t want) {
    if (want <= *cap) return RT_OK;
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void**)buf, want * sizeof(T)));
    *cap = want;
    return RT_OK;
}

}  // namespace

// Everything one in-flight chunk needs: the record arrays, the level table and the stream its
// kernels are queued on.  Two workspaces let two chunks overlap on the GPU.
struct Workspace {
    hipStream_t stream = nullptr;
    void* rec_slab = nullptr;           // one allocation carved into the arrays of R
    Recs R{};
    size_t rec_cap = 0;
    double* intensity = nullptr;
    size_t intensity_cap = 0;
    LevelState* levels = nullptr;
    size_t levels_cap = 0;
    Counters* counters = nullptr;
    StatShard* stats = nullptr;
    ChunkStatus* status = nullptr;      // one entry per chunk queued on this workspace (page-locked host memory)
    size_t status_cap = 0;
    std::vector<hipEvent_t> ev;         // four per queued chunk: start, traced, shadowed, shaded
    // rng order (area lights)
    Rng* rng_states = nullptr;
    size_t rng_cap = 0;
    uint32_t* pixel_base = nullptr;
    size_t pixel_base_cap = 0;
    uint32_t *row_total = nullptr, *row_base = nullptr;
    size_t row_cap = 0, row_base_cap = 0;
    // the big-list kernels' per-thread lists (allocated when a chunk first overflows the fixed ones)
    void* big_slab = nullptr;
    size_t big_bytes = 0;
    uint32_t big_csg_cap = 0, big_cont_cap = 0;
};
constexpr int MAX_STREAMS = 4;
constexpr uint32_t RT_NOALLOC_MAX_RAYS = 1u << 16, RT_NOALLOC_MAX_DEPTH = 1022;  // limits of rt_colour_at_noalloc
constexpr uint32_t PER_LANE_MIN_UNITS = 512;  // trees above this many 64-byte units are walked lane by lane

struct rt_scene {
    DevScene dev{};
    std::vector<void*> allocations;
    bool has_csg = false, has_area = false;
    unsigned long long* rng_jump = nullptr;  // area lights: the jump matrices of k_rng_rows
    bool spawns = true;  // some material reflects or refracts; without one every level but the first stays empty
    bool shade_lds = false;             // the tables k_shade looks things up in fit in LDS
    int device = 0;
    Workspace ws[MAX_STREAMS];
    int n_streams = 2;
    int traversal = 0;                  // 0: by tree size, 1: the wave walks together, 2: lane by lane, 3: the reference's walk,
                                        // 4: the wave together, meshes lane by lane
    uint32_t mesh_units = 0;            // units inside meshes that have a hierarchy of their own
    uint32_t grid_blocks = 0;           // blocks of every level kernel (they stride over the rays); 0: see level_grid
    double* stage_rgb = nullptr;        // device staging for host outputs
    uint8_t* stage_rgb8 = nullptr;
    size_t stage_pixels = 0;
    hipStream_t copy_stream = nullptr;  // host outputs: finished chunks go to the caller's memory beside the next one's kernels
    std::vector<hipEvent_t> copy_ev;
    bool pin_outputs = true;            // page-lock pageable host outputs for the length of the call, where the driver can
    void* bounce[2] = {nullptr, nullptr};   // page-locked staging for pageable host outputs (f64, u8)
    size_t bounce_bytes[2] = {0, 0};
    size_t chunk_pixels = (size_t)26 << 20;
    size_t recs_per_pixel = 24;
    uint32_t min_chunks = 2;
    unsigned long long* seeds = nullptr;
    size_t seeds_cap = 0;
    std::vector<uint32_t> leaf_unit;    // unit position of every leaf's record (rt_lighting takes leaf indices)
    // the most intersections one ray can leave in the list of an outermost csg / in the world's list
    uint64_t csg_list_bound = 0, world_list_bound = 0;
};

namespace {

int ensure_workspace(rt_scene* s, Workspace& w, size_t chunk_px, size_t n_first, uint32_t depth, size_t n_chunks) {
    if (!w.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        HIP_TRY(hipMalloc((void**)&w.counters, sizeof(Counters)));
        HIP_TRY(hipMalloc((void**)&w.stats, STAT_SHARDS * sizeof(StatShard)));
        HIP_TRY(hipMemsetAsync(w.stats, 0, STAT_SHARDS * sizeof(StatShard), w.stream));
    }
    size_t want_r = std::max<size_t>(s->recs_per_pixel * std::max(chunk_px, n_first), 4096);
    if (want_r > 0x7FFFFFF0ull) want_r = 0x7FFFFFF0ull;
    if (want_r > w.rec_cap) {
        HIP_TRY(hipStreamSynchronize(w.stream));
        if (w.rec_slab) HIP_TRY(hipFree(w.rec_slab));
        w.rec_slab = nullptr;
        w.rec_cap = 0;
        const size_t cap = (want_r + 63) & ~(size_t)63;  // every array starts 256-byte aligned
        HIP_TRY(hipMalloc(&w.rec_slab, cap * REC_BYTES));
        w.R.base = (unsigned char*)w.rec_slab;
        w.rec_cap = cap;
    }
    if (w.rec_cap * s->dev.n_lights > w.intensity_cap || (size_t)depth + 2 > w.levels_cap || n_chunks > w.status_cap)
        HIP_TRY(hipStreamSynchronize(w.stream));
    int rc = grow(&w.intensity, &w.intensity_cap, w.rec_cap * s->dev.n_lights);
    if (rc == RT_OK) rc = grow(&w.levels, &w.levels_cap, (size_t)depth + 2);
    if (rc != RT_OK) return rc;
    if (n_chunks > w.status_cap) {
        // page-locked host memory the device writes straight into: the host reads a chunk's status after the
        // stream has drained, with no copy to wait for
        if (w.status) HIP_TRY(hipHostFree(w.status));
        w.status = nullptr; w.status_cap = 0;
        const size_t cap = std::max<size_t>(n_chunks, 16);
        HIP_TRY(hipHostMalloc((void**)&w.status, cap * sizeof(ChunkStatus), hipHostMallocDefault));
        w.status_cap = cap;
    }
    while (w.ev.size() < 4 * n_chunks) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        w.ev.push_back(e);
    }
    return RT_OK;
}

void free_workspace(Workspace& w) {
    if (w.stream) (void)hipStreamSynchronize(w.stream);
    void* bufs[] = {w.rec_slab, w.intensity, w.levels, w.counters, w.stats,
                    w.rng_states, w.pixel_base, w.row_total, w.row_base, w.big_slab};
    for (void* p : bufs) if (p) (void)hipFree(p);
    if (w.status) (void)hipHostFree(w.status);
    for (auto& e : w.ev) if (e) (void)hipEventDestroy(e);
    if (w.stream) (void)hipStreamDestroy(w.stream);
    w = Workspace{};
}

// The lists of the big-list kernels: BIG_GRID x 256 threads, each with room for what the scene can produce at
// most (up to BIG_LIST_MAX entries a list).
int ensure_big(rt_scene* s, Workspace& w) {
    const uint32_t csg_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(s->csg_list_bound, 1), BIG_LIST_MAX);
    const uint32_t cont_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(s->world_list_bound, CONT_CAP), BIG_LIST_MAX);
    const size_t threads = (size_t)BIG_GRID * 256;
    const size_t bytes = threads * ((size_t)csg_cap * 12 + (size_t)cont_cap * 16);
    if (w.big_slab && w.big_bytes >= bytes && w.big_csg_cap == csg_cap && w.big_cont_cap == cont_cap) return RT_OK;
    if (w.big_slab) { HIP_TRY(hipStreamSynchronize(w.stream)); HIP_TRY(hipFree(w.big_slab)); }
    w.big_slab = nullptr; w.big_bytes = 0;
    HIP_TRY(hipMalloc(&w.big_slab, bytes));
    w.big_bytes = bytes; w.big_csg_cap = csg_cap; w.big_cont_cap = cont_cap;
    return RT_OK;
}
DevScene big_scene(const rt_scene* s, const Workspace& w) {
    DevScene d = s->dev;
    const size_t threads = (size_t)BIG_GRID * 256;
    unsigned char* p = (unsigned char*)w.big_slab;
    d.big_ct = (double*)p; p += threads * w.big_csg_cap * 8;
    d.big_cont_t = (double*)p; p += threads * w.big_cont_cap * 8;
    d.big_cl = (uint32_t*)p; p += threads * w.big_csg_cap * 4;
    d.big_cont_pos = (uint32_t*)p; p += threads * w.big_cont_cap * 4;
    d.big_cont_cls = (uint32_t*)p;
    d.big_csg_cap = w.big_csg_cap; d.big_cont_cap = w.big_cont_cap;
    return d;
}

struct RunTotals {
    uint64_t secondary = 0, shadow = 0, shade = 0, flop_trace = 0, flop_shadow = 0, flop_light = 0, mesh_nodes = 0, mesh_tris = 0;
    uint64_t trace_launches = 0;
    uint32_t levels = 0;
    double ms_trace = 0, ms_shadow = 0, ms_shade = 0;
};

struct FirstLevel {   // where the first level's rays come from
    bool primary;     // true: pixels of a chunk; false: rays already in records [0, n)
    uint32_t n;       // number of first-level rays
    uint32_t width, rows;            // primary only
    const unsigned long long* seeds; // device: per global row (primary) or per ray; area lights only
};

// k_trace casts the shadow rays itself unless an area light needs the chunk's rng order first, or
// the tree is walked lane by lane: there the deep levels hold few rays with long walks, and a
// shadow walk behind each of them doubles the serial chain, while k_shadow runs all the chunk's
// shadow rays side by side.
// 0: the wave walks the tree together; 1: every lane on its own; 2: the wave together down to the meshes, every
// lane on its own inside them (trees that are small once their meshes are set aside)
int walk_mode(const rt_scene* s) {
    if (s->traversal == 1 || s->traversal == 3) return 0;
    if (s->traversal == 2) return 1;
    if (s->traversal == 4) return 2;
    if (s->dev.n_units <= PER_LANE_MIN_UNITS) return 0;
    return s->mesh_units > 0 && s->dev.n_units - s->mesh_units <= PER_LANE_MIN_UNITS ? 2 : 1;
}
bool uses_lane_walk(const rt_scene* s) { return walk_mode(s) != 0; }  // lanes of a wave walk (partly) on their own
bool reference_walk(const rt_scene* s) { return s->traversal == 3; }  // no device-only shortcut anywhere
bool fused_shadows(const rt_scene* s, bool per_lane) { return !s->has_area && !per_lane && !reference_walk(s); }
// Blocks per level kernel.  A k_trace block is 4 waves of the 16 a CU holds, so 1024 blocks are one
// full GPU of them.  A chunk rendered alone wants several rounds of blocks (the hardware hands a
// freed slot to the next block, which evens out walks of different length); with chunks on two or
// more streams, one round per stream leaves the other stream's kernels room to run beside it.
uint32_t level_grid(const rt_scene* s, bool trace) {
    if (s->grid_blocks) return s->grid_blocks;
    // measured on the 10000x5000 headline frame: 64.8 -> 63.7 ms; lane walks differ too much in length
    // for a single round of blocks (3000x1200 bounding-box frame: 4.4 -> 4.8 ms)
    if (s->n_streams > 1 && !uses_lane_walk(s)) return trace ? 1024u : 2048u;
    return 4096u;
}
template <bool PRIMARY, bool HAS_CSG, bool PER_LANE, bool HYBRID>
void launch_trace(rt_scene* s, dim3 grid, hipStream_t st, const LevelArgs& A) {
    if constexpr (HAS_CSG && !PER_LANE && !HYBRID) {
        if (reference_walk(s)) {
            hipLaunchKernelGGL((k_trace<PRIMARY, true, false, false, false>), grid, dim3(TRACE_BLOCK), 0, st, s->dev, A);
            return;
        }
    }
    if constexpr (HYBRID) {
        hipLaunchKernelGGL((k_trace<PRIMARY, HAS_CSG, false, false, true, false, true>), grid, dim3(TRACE_BLOCK), 0, st, s->dev, A);
    } else if constexpr (PER_LANE) {
        hipLaunchKernelGGL((k_trace<PRIMARY, HAS_CSG, true, false>), grid, dim3(TRACE_BLOCK), 0, st, s->dev, A);
    } else {
        if (fused_shadows(s, false)) hipLaunchKernelGGL((k_trace<PRIMARY, HAS_CSG, false, true>), grid, dim3(TRACE_BLOCK), 0, st, s->dev, A);
        else hipLaunchKernelGGL((k_trace<PRIMARY, HAS_CSG, false, false>), grid, dim3(TRACE_BLOCK), 0, st, s->dev, A);
    }
}

// Queue every kernel of one batch of first-level rays on the workspace's stream: the levels, the
// shadow rays, then the colours from the deepest level up (level 0 writes the outputs).  Nothing
// here waits for the GPU unless the scene has an area light (its rng order needs the level sizes
// on the host).  `slot` selects the workspace's ChunkStatus entry and event quadruple.
template <bool HAS_CSG, bool PER_LANE, bool HYBRID = false>
int queue_chunk(rt_scene* s, Workspace& w, LevelArgs base, const FirstLevel& first, uint32_t depth, double* out_rgb,
                uint8_t* out_rgb8, size_t out_offset, size_t slot, RunTotals& tot, bool big) {
    hipStream_t st = w.stream;
    // `big`: the chunk overflowed a fixed per-ray list; walk it lane by lane with the lists in global memory
    DevScene big_dev{};
    if (big) {
        int rc = ensure_big(s, w);
        if (rc != RT_OK) return rc;
        big_dev = big_scene(s, w);
    }
    const uint32_t n_levels = depth + 1;
    const dim3 grid(level_grid(s, false)), grid_trace(level_grid(s, true)), block(256);
    LevelArgs A = base;
    A.levels = w.levels;
    A.n_levels = n_levels;
    A.R = w.R;
    A.rec_cap = (uint32_t)w.rec_cap;
    A.intensity = w.intensity;
    A.rng_states = nullptr;
    A.counters = w.counters;
    A.stats = w.stats;
    A.out_rgb = nullptr; A.out_rgb8 = nullptr; A.out_offset = 0;
    hipLaunchKernelGGL(k_chunk_begin, dim3((n_levels + 1 + 255) / 256), dim3(256), 0, st, w.levels, n_levels + 1, first.n,
                       w.counters);
    const bool timed = w.ev.size() >= 4 * (slot + 1);  // the allocation-free entry point brings no events
    if (timed) HIP_TRY(hipEventRecord(w.ev[4 * slot + 0], st));
    for (uint32_t level = 0; level < n_levels; level++) {
        A.level = level;
        A.depth_remaining = depth - level;
        if (big) {
            if (first.primary && level == 0)
                hipLaunchKernelGGL((k_trace<true, true, true, false, true, true>), dim3(BIG_GRID), dim3(TRACE_BLOCK), 0, st, big_dev, A);
            else
                hipLaunchKernelGGL((k_trace<false, true, true, false, true, true>), dim3(BIG_GRID), dim3(TRACE_BLOCK), 0, st, big_dev, A);
        } else if (first.primary && level == 0)
            launch_trace<true, HAS_CSG, PER_LANE, HYBRID>(s, grid_trace, st, A);
        else
            launch_trace<false, HAS_CSG, PER_LANE, HYBRID>(s, grid_trace, st, A);
    }
    tot.trace_launches += n_levels;
    HIP_TRY(hipGetLastError());
    if (timed) HIP_TRY(hipEventRecord(w.ev[4 * slot + 1], st));

    if (s->has_area) {
        // number the shade points in the reference's rng order and step the row generators to them
        std::vector<LevelState> lv(n_levels + 1);
        HIP_TRY(hipMemcpyAsync(lv.data(), w.levels, lv.size() * sizeof(LevelState), hipMemcpyDeviceToHost, st));
        Counters c{};
        HIP_TRY(hipMemcpyAsync(&c, w.counters, sizeof c, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!(c.error & 1u)) {
            const uint32_t n0 = lv[0].n;
            const uint32_t width = first.primary ? first.width : 1, rows = first.primary ? first.rows : n0;
            for (int level = (int)n_levels - 1; level >= 0; level--)
                if (lv[level].n)
                    hipLaunchKernelGGL(k_order_sizes, dim3((lv[level].n + 255) / 256), block, 0, st, w.R, lv[level].base,
                                       lv[level].n);
            int rc = grow(&w.pixel_base, &w.pixel_base_cap, (size_t)n0);
            if (rc == RT_OK) rc = grow(&w.row_total, &w.row_cap, (size_t)rows);
            if (rc == RT_OK) rc = grow(&w.row_base, &w.row_base_cap, (size_t)rows);
            if (rc != RT_OK) return rc;
            hipLaunchKernelGGL(k_order_rows, dim3(rows), dim3(256), 0, st, w.R, width, rows, w.pixel_base,
                               w.row_total);
            std::vector<uint32_t> totals(rows), bases(rows);
            HIP_TRY(hipMemcpyAsync(totals.data(), w.row_total, rows * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            uint64_t acc = 0;
            for (uint32_t r = 0; r < rows; r++) {
                bases[r] = (uint32_t)acc;
                acc += totals[r];
                if (totals[r] >> RNG_JUMPS) return fail(RT_ERR_CAPACITY, "too many shade points in one image row");
            }
            if (acc > 0xFFFFFFF0ull) return fail(RT_ERR_CAPACITY, "too many shade points in one chunk");
            HIP_TRY(hipMemcpyAsync(w.row_base, bases.data(), rows * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));  // `bases` goes out of scope
            rc = grow(&w.rng_states, &w.rng_cap, std::max<size_t>((size_t)acc, 1));
            if (rc != RT_OK) return rc;
            for (uint32_t level = 0; level < n_levels; level++)
                if (lv[level].n)
                    hipLaunchKernelGGL(k_order_assign, dim3((lv[level].n + 255) / 256), block, 0, st, w.R, lv[level].base,
                                       lv[level].n, width, w.pixel_base, w.row_base, level == 0 ? 1 : 0);
            uint32_t most = 0;
            for (uint32_t r = 0; r < rows; r++) most = std::max(most, totals[r]);
            const uint32_t blocks_per_row = std::min<uint32_t>(std::max<uint32_t>((most + 255) / 256, 1), 65535);
            hipLaunchKernelGGL(k_rng_rows, dim3(rows, blocks_per_row), dim3(256), 0, st, w.rng_states, w.row_base, w.row_total,
                               first.seeds, rows, first.primary ? base.chunk_row0 : 0u,
                               first.primary ? base.part_block : 0u, first.primary ? base.part_rank : 0u,
                               first.primary ? base.part_world : 1u, s->rng_jump);
            HIP_TRY(hipGetLastError());
            A.rng_states = w.rng_states;
        }
    }

    bool shadows_queued = false;
    if (big) {
        hipLaunchKernelGGL((k_shadow<true, true, true, true>), dim3(BIG_GRID), block, 0, st, big_dev, A);
        shadows_queued = true;
    }
    if constexpr (HAS_CSG && !PER_LANE && !HYBRID) {
        if (!shadows_queued && reference_walk(s)) {
            hipLaunchKernelGGL((k_shadow<true, false, false>), grid, block, 0, st, s->dev, A);
            shadows_queued = true;
        }
    }
    if (!shadows_queued && !fused_shadows(s, PER_LANE || HYBRID)) {
        if constexpr (HYBRID) hipLaunchKernelGGL((k_shadow<HAS_CSG, false, true, false, true>), grid, block, 0, st, s->dev, A);
        else hipLaunchKernelGGL((k_shadow<HAS_CSG, PER_LANE>), grid, block, 0, st, s->dev, A);
    }
    if (timed) HIP_TRY(hipEventRecord(w.ev[4 * slot + 2], st));
    for (int level = (int)n_levels - 1; level >= 0; level--) {
        A.level = (uint32_t)level;
        if (level == 0) { A.out_rgb = out_rgb; A.out_rgb8 = out_rgb8; A.out_offset = out_offset; }
        if (s->shade_lds) hipLaunchKernelGGL(k_shade<true>, grid, block, 0, st, s->dev, A);
        else hipLaunchKernelGGL(k_shade<false>, grid, block, 0, st, s->dev, A);
    }
    if (timed) HIP_TRY(hipEventRecord(w.ev[4 * slot + 3], st));
    hipLaunchKernelGGL(k_chunk_end, dim3(1), dim3(STAT_SHARDS), 0, st, w.levels, n_levels, w.counters, w.stats,
                       w.status + slot);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int queue_chunk_dispatch(rt_scene* s, Workspace& w, const LevelArgs& base, const FirstLevel& first, uint32_t depth,
                         double* out_rgb, uint8_t* out_rgb8, size_t out_offset, size_t slot, RunTotals& tot, bool big = false) {
    const int mode = walk_mode(s);
    if (big) return queue_chunk<true, true>(s, w, base, first, depth, out_rgb, out_rgb8, out_offset, slot, tot, true);
    if (reference_walk(s)) return queue_chunk<true, false>(s, w, base, first, depth, out_rgb, out_rgb8, out_offset, slot, tot, false);
    if (s->has_csg) {
        if (mode == 2) return queue_chunk<true, false, true>(s, w, base, first, depth, out_rgb, out_rgb8, out_offset, slot, tot, false);
        if (mode == 1) return queue_chunk<true, true>(s, w, base, first, depth, out_rgb, out_rgb8, out_offset, slot, tot, false);
        return queue_chunk<true, false>(s, w, base, first, depth, out_rgb, out_rgb8, out_offset, slot, tot, false);
    }
    if (mode == 2) return queue_chunk<false, false, true>(s, w, base, first, depth, out_rgb, out_rgb8, out_offset, slot, tot, false);
    if (mode == 1) return queue_chunk<false, true>(s, w, base, first, depth, out_rgb, out_rgb8, out_offset, slot, tot, false);
    return queue_chunk<false, false>(s, w, base, first, depth, out_rgb, out_rgb8, out_offset, slot, tot, false);
}

// Wait for a workspace's stream and fetch the status of its first `n_slots` chunks.
int collect(Workspace& w, size_t n_slots, std::vector<ChunkStatus>& out, RunTotals& tot) {
    out.assign(n_slots, ChunkStatus{});
    if (n_slots == 0) return RT_OK;
    HIP_TRY(hipStreamSynchronize(w.stream));
    std::memcpy(out.data(), w.status, n_slots * sizeof(ChunkStatus));  // page-locked host memory (ensure_workspace)
    float ms = 0;
    for (size_t k = 0; k < n_slots; k++) {
        HIP_TRY(hipEventElapsedTime(&ms, w.ev[4 * k + 0], w.ev[4 * k + 1]));
        tot.ms_trace += ms;
        HIP_TRY(hipEventElapsedTime(&ms, w.ev[4 * k + 1], w.ev[4 * k + 2]));
        tot.ms_shadow += ms;
        HIP_TRY(hipEventElapsedTime(&ms, w.ev[4 * k + 2], w.ev[4 * k + 3]));
        tot.ms_shade += ms;
    }
    return RT_OK;
}

void add_stats(RunTotals& tot, const ChunkStatus& c) {
    tot.secondary += c.secondary; tot.shadow += c.shadow; tot.shade += c.shade;
    tot.flop_trace += c.flop_trace; tot.flop_shadow += c.flop_shadow; tot.flop_light += c.flop_light;
    tot.mesh_nodes += c.mesh_nodes; tot.mesh_tris += c.mesh_tris;
    tot.levels = std::max(tot.levels, c.levels);
}

// `big`: the status of a chunk that ran with the big-list kernels
int check_status(const ChunkStatus& c, bool big = false) {
    if ((c.error & 2u) && big)
        return fail(RT_ERR_CAPACITY, "a per-ray intersection list overflowed even in global memory (" + std::to_string(BIG_LIST_MAX) +
                                         " entries a list, " + std::to_string(BIG_CSG_DEPTH) + " nested csgs)");
    if (c.error & 2u)
        return fail(RT_ERR_CAPACITY, "a per-ray intersection list overflowed (csg list " + std::to_string(CSG_CAP) + ", " +
                                         std::to_string(CSG_DEPTH) + " nested csgs, container list " + std::to_string(CONT_CAP) +
                                         "); this entry point has no slow path for it");
    if (c.error & 4u) return fail(RT_ERR_NAN, "NaN intersection distance (the reference panics while sorting)");
    return RT_OK;
}

#ifdef RT_PHASE_TIMERS
// Diagnostic build: print and clear the per-phase shader-clock sums of k_trace.
void print_phases(Workspace& w) {
    std::vector<StatShard> shards(STAT_SHARDS);
    if (hipMemcpy(shards.data(), w.stats, STAT_SHARDS * sizeof(StatShard), hipMemcpyDeviceToHost) != hipSuccess) return;
    (void)hipMemset(w.stats, 0, STAT_SHARDS * sizeof(StatShard));
    unsigned long long ph[8] = {0};
    for (const StatShard& sh : shards) for (int k = 0; k < 8; k++) ph[k] += sh.phase[k];
    static const char* names[8] = {"shadows", "ray", "walk", "material", "containers", "geometry", "spawn+sync", "stores"};
    for (int k = 0; k < 8; k++) std::fprintf(stderr, "phase %-11s %.3e ticks\n", names[k], (double)ph[k]);
}
#endif

// memcpy over a few threads: one thread moves about 10 GB/s, a frame of f64 pixels is 1.2 GB.
void parallel_copy(void* dst, const void* src, size_t bytes) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n = bytes < ((size_t)8 << 20) ? 1 : std::min<size_t>(8, hw ? hw : 1);
    if (n <= 1) { std::memcpy(dst, src, bytes); return; }
    const size_t slice = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
    std::vector<std::thread> pool;
    for (size_t off = slice; off < bytes; off += slice)
        pool.emplace_back([=] { std::memcpy((char*)dst + off, (const char*)src + off, std::min(slice, bytes - off)); });
    std::memcpy(dst, src, std::min(slice, bytes));
    for (auto& t : pool) t.join();
}

// Whether the HIP runtime knows `p` as page-locked host memory (hipHostMalloc, hipHostRegister): copies into
// such memory can be asynchronous and run at the link's rate.
bool host_page_locked(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

uint32_t local_rows(const rt_camera* cam, const rt_partition* p) {
    if (!p || p->world <= 1) return cam->height;
    uint32_t rows = 0;
    for (uint32_t b = 0; b * p->block_rows < cam->height; b++)
        if (b % p->world == p->rank) rows += std::min(p->block_rows, cam->height - b * p->block_rows);
    return rows;
}

int upload_seeds(rt_scene* s, const uint64_t* seeds, size_t n) {
    int rc = grow(&s->seeds, &s->seeds_cap, std::max<size_t>(n, 1));
    if (rc != RT_OK) return rc;
    std::vector<unsigned long long> tmp(n, 0);
    if (seeds) for (size_t i = 0; i < n; i++) tmp[i] = seeds[i];
    HIP_TRY(hipMemcpy(s->seeds, tmp.data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice));
    return RT_OK;
}

int upload_rays(Workspace& w, uint32_t n, const double* rays) {
    double* staged = nullptr;
    HIP_TRY(hipMalloc((void**)&staged, (size_t)n * 6 * sizeof(double)));
    hipError_t e = hipMemcpyAsync(staged, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, w.stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_load_rays, dim3((n + 255) / 256), dim3(256), 0, w.stream, staged, n, w.R);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(w.stream);
    (void)hipFree(staged);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }
uint32_t rt_max_csg_list(void) { return BIG_LIST_MAX; }
uint32_t rt_max_container_list(void) { return BIG_LIST_MAX; }

int rt_init(int device) {
    HIP_TRY(hipSetDevice(device));
    unsigned char* p = nullptr;
    return scratch_of_current_device(&p);  // rt_colour_at_noalloc's memory on this GPU
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
            if (d->node_arg[i] < d->n_csgs && d->csg_op[d->node_arg[i]] > RT_CSG_UNION) return fail(RT_ERR_INVALID, "unknown csg operation");
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
    uint64_t draws = 0;
    bool has_area = false;
    for (uint32_t i = 0; i < d->n_lights; i++) {
        if (d->light_kind[i] == RT_LIGHT_AREA) {
            has_area = true;
            uint64_t samples = (uint64_t)d->light_steps[(size_t)i * 2] * d->light_steps[(size_t)i * 2 + 1];
            if (samples == 0 || samples > (1u << 20)) return fail(RT_ERR_INVALID, "area light step counts out of range");
            draws += 4 * samples;  // positions() twice per shade point, two draws per sample (area.rs:57-67,83; material.rs:64)
        } else if (d->light_kind[i] != RT_LIGHT_POINT) {
            return fail(RT_ERR_INVALID, "unknown light kind");
        }
    }

    if (draws > 0xFFFFFFFFull) return fail(RT_ERR_INVALID, "the area lights take more than 2^32 rng draws per shade point");
    // ---- enclosing node of every node, and whether it lies inside a csg
    std::vector<uint32_t> parent_node(d->n_nodes, NONE);
    std::vector<uint8_t> in_csg(d->n_nodes, 0);
    {
        std::vector<uint32_t> open;
        uint32_t open_csgs = 0;
        for (uint32_t i = 0; i < d->n_nodes; i++) {
            while (!open.empty() && d->node_skip[open.back()] <= i) {
                if (d->node_kind[open.back()] == RT_NODE_CSG) open_csgs--;
                open.pop_back();
            }
            parent_node[i] = open.empty() ? NONE : open.back();
            if (!open.empty() && d->node_skip[i] > d->node_skip[open.back()])
                return fail(RT_ERR_INVALID, "a node's subtree must end inside its parent's");
            in_csg[i] = open_csgs > 0;
            if (d->node_kind[i] != RT_NODE_LEAF) {
                open.push_back(i);
                if (d->node_kind[i] == RT_NODE_CSG) open_csgs++;
            }
        }
    }
    // ---- the longest lists one ray can produce: every leaf delivers at most as many intersections as its shape
    // has (sphere, cube 2; cylinder, cone 4; plane, triangle 1); an outermost csg buffers those of its subtree
    uint64_t csg_list_bound = 0, world_list_bound = 0;
    {
        uint64_t in_outermost = 0;
        uint32_t outermost_end = 0;
        for (uint32_t i = 0; i < d->n_nodes; i++) {
            if (d->node_kind[i] == RT_NODE_CSG && !in_csg[i]) { in_outermost = 0; outermost_end = d->node_skip[i]; }
            if (d->node_kind[i] != RT_NODE_LEAF) continue;
            const uint32_t shape = d->leaf_shape[d->node_arg[i]];
            const uint64_t hits = shape == RT_SPHERE || shape == RT_CUBE ? 2 : shape == RT_CYLINDER || shape == RT_CONE ? 4 : 1;
            world_list_bound += hits;
            if (i < outermost_end) { in_outermost += hits; csg_list_bound = std::max(csg_list_bound, in_outermost); }
        }
    }
    // ---- bounding spheres of the bounded leaves, and the clusters: runs of three or more sibling
    // leaves with bounding spheres, outside any csg, get a header unit of their own (see K_CLUSTER)
    std::vector<double> leaf_sphere((size_t)d->n_nodes * 6, 0.0);
    std::vector<uint8_t> has_sphere(d->n_nodes, 0);
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        if (d->node_kind[i] != RT_NODE_LEAF) continue;
        const uint32_t leaf = d->node_arg[i];
        has_sphere[i] = bounding_sphere(d->leaf_shape[leaf], d->leaf_inv + (size_t)leaf * 12, d->leaf_param[(size_t)leaf * 3 + 0],
                                        d->leaf_param[(size_t)leaf * 3 + 1], &leaf_sphere[(size_t)i * 6]);
    }
#ifndef CLUSTER_MIN_LEAVES
#define CLUSTER_MIN_LEAVES 3
#endif
    constexpr uint32_t CLUSTER_MIN = CLUSTER_MIN_LEAVES;
    std::vector<uint32_t> cluster_end(d->n_nodes, 0);  // at the first node of a run: the node after it
    for (uint32_t i = 0; i < d->n_nodes;) {
        uint32_t j = i;
        while (j < d->n_nodes && d->node_kind[j] == RT_NODE_LEAF && has_sphere[j] && !in_csg[j] && parent_node[j] == parent_node[i]) j++;
        if (j - i >= CLUSTER_MIN) cluster_end[i] = j;
        i = std::max(j, i + 1);
    }

    // ---- node stream
    std::vector<uint32_t> unit_of(d->n_nodes + 1);
    uint32_t units = 0;
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        if (cluster_end[i]) units += 1;  // the cluster's header comes right before its first leaf
        unit_of[i] = units;
        units += d->node_kind[i] == RT_NODE_LEAF ? 2 : 1;
    }
    unit_of[d->n_nodes] = units;
    // where a subtree that ends at node k ends in the stream: a cluster header belongs to the run behind it
    auto end_unit = [&](uint32_t k) { return k < d->n_nodes && cluster_end[k] ? unit_of[k] - 1 : unit_of[k]; };
    std::vector<uint32_t> stream((size_t)(units + 4) * 16, 0);  // padded: the walk may read one record ahead
    std::vector<double> xforms, tri_normals((size_t)d->n_tris * 9, 0.0), cull((size_t)(units + 4) * 6, 0.0);
    std::map<std::vector<uint64_t>, uint32_t> xform_ids;
    bool has_csg = false;
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        uint32_t* rec = &stream[(size_t)unit_of[i] * 16];
        double* recd = (double*)rec;
        if (d->node_kind[i] == RT_NODE_LEAF) {
            uint32_t leaf = d->node_arg[i];
            uint32_t shape = d->leaf_shape[leaf];
            bool closed = d->leaf_param[(size_t)leaf * 3 + 2] != 0.0;
            rec[0] = shape | (d->leaf_casts_shadow[leaf] ? 1u << 8 : 0u) | (closed ? 1u << 9 : 0u);
            rec[1] = d->leaf_material[leaf];
            rec[2] = leaf;
            rec[3] = d->leaf_class[leaf];
            const double* inv = d->leaf_inv + (size_t)leaf * 12;
            if (shape == RT_TRIANGLE) {
                std::vector<uint64_t> key(12);
                std::memcpy(key.data(), inv, 96);
                auto it = xform_ids.find(key);
                uint32_t id;
                if (it == xform_ids.end()) {
                    id = (uint32_t)xform_ids.size();
                    xform_ids.emplace(key, id);
                    xforms.insert(xforms.end(), inv, inv + 12);
                } else id = it->second;
                uint32_t tri = d->leaf_tri[leaf];
                rec[4] = id;
                rec[5] = tri;
                for (int k = 0; k < 9; k++) recd[4 + k] = d->tri_data[(size_t)tri * 18 + k];
                for (int k = 0; k < 9; k++) tri_normals[(size_t)tri * 9 + k] = d->tri_data[(size_t)tri * 18 + 9 + k];
            } else {
                for (int k = 0; k < 12; k++) recd[2 + k] = inv[k];
                recd[14] = d->leaf_param[(size_t)leaf * 3 + 0];
                recd[15] = d->leaf_param[(size_t)leaf * 3 + 1];
                if (has_sphere[i]) {
                    std::memcpy(&cull[(size_t)unit_of[i] * 6], &leaf_sphere[(size_t)i * 6], 6 * sizeof(double));
                    rec[0] |= 1u << 10;
                    // Wholly behind the origin a leaf only has intersections with t < 0.  A walk that
                    // records wants those too, unless they cancel: a sphere or cube outside any csg is
                    // always entered and left, which toggles the container list twice (bit 11).  A walk
                    // that only wants the closest hit can skip the leaf (bit 16) - but inside a csg only
                    // if it is entered and left: an open cylinder or a cone can be crossed once, which
                    // flips the csg's inside state for everything in front of the origin.
                    const bool pairs = shape == RT_SPHERE || shape == RT_CUBE;
                    if (pairs && !in_csg[i]) rec[0] |= 1u << 11;
                    if (pairs || !in_csg[i]) rec[0] |= 1u << 16;
                }
            }
        } else {
            bool csg = d->node_kind[i] == RT_NODE_CSG;
            has_csg |= csg;
            rec[0] = csg ? K_CSG : K_GROUP;
            rec[1] = end_unit(d->node_skip[i]);
            rec[2] = csg ? d->node_arg[i] : 0;
            for (int k = 0; k < 6; k++) recd[2 + k] = d->node_bbox[(size_t)i * 6 + k];
        }
    }

    // cluster headers: a sphere around the members' bounding spheres, the algorithmic cost of the
    // members' tests, and the members' common "may be skipped when wholly behind the origin" bits
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        if (!cluster_end[i]) continue;
        const uint32_t n = cluster_end[i] - i;
        double c[3] = {0.0, 0.0, 0.0};
        for (uint32_t j = i; j < cluster_end[i]; j++)
            for (int a = 0; a < 3; a++) c[a] += leaf_sphere[(size_t)j * 6 + a] / (double)n;
        double radius = 0.0;
        uint32_t bits = (1u << 11) | (1u << 16), cost = 0;
        for (uint32_t j = i; j < cluster_end[i]; j++) {
            const double* ls = &leaf_sphere[(size_t)j * 6];
            const double dx = ls[0] - c[0], dy = ls[1] - c[1], dz = ls[2] - c[2];
            radius = std::max(radius, std::sqrt(dx * dx + dy * dy + dz * dz) + ls[3]);
            bits &= stream[(size_t)unit_of[j] * 16];
            const uint32_t shape = d->leaf_shape[d->node_arg[j]];
            cost += F_XFORM + (shape == RT_SPHERE ? F_SPHERE : shape == RT_CUBE ? F_BOX : shape == RT_CYLINDER ? F_CYL : F_CONE);
        }
        radius = radius * (1.0 + 1e-9) + 1e-9;
        uint32_t* rec = &stream[(size_t)(unit_of[i] - 1) * 16];
        double* cs = &cull[(size_t)(unit_of[i] - 1) * 6];
        const bool ok = std::isfinite(radius) && std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]);
        rec[0] = K_CLUSTER | (ok ? bits : 0u);
        rec[1] = end_unit(cluster_end[i]);
        rec[2] = cost;
        // a header that must never cull: a sphere nothing is outside of
        cs[0] = ok ? c[0] : 0.0; cs[1] = ok ? c[1] : 0.0; cs[2] = ok ? c[2] : 0.0;
        cs[3] = ok ? radius : HUGE_VAL; cs[4] = ok ? radius * radius : HUGE_VAL;
    }

    // groups whose stored box certainly holds their subtree: all six coordinates finite and no
    // unbounded shape below (a rotated endless cylinder's box is made of inf - inf)
    {
        std::vector<uint8_t> bounded(d->n_nodes, 1);
        for (uint32_t i = d->n_nodes; i-- > 0;) {  // children have higher indices than their parent
            if (d->node_kind[i] == RT_NODE_LEAF) {
                const uint32_t leaf = d->node_arg[i], shape = d->leaf_shape[leaf];
                const double mn = d->leaf_param[(size_t)leaf * 3 + 0], mx = d->leaf_param[(size_t)leaf * 3 + 1];
                if (shape == RT_PLANE) bounded[i] = 0;
                if ((shape == RT_CYLINDER || shape == RT_CONE) && !(std::isfinite(mn) && std::isfinite(mx))) bounded[i] = 0;
                const double* inv = d->leaf_inv + (size_t)leaf * 12;
                for (int k = 0; k < 12; k++) if (!std::isfinite(inv[k])) bounded[i] = 0;
            } else {
                for (uint32_t j = i + 1; j < d->node_skip[i]; j = d->node_skip[j]) bounded[i] = bounded[i] && bounded[j];
                bool finite = true;
                for (int k = 0; k < 6; k++) finite = finite && std::isfinite(d->node_bbox[(size_t)i * 6 + k]);
                if (d->node_kind[i] == RT_NODE_GROUP && bounded[i] && finite) stream[(size_t)unit_of[i] * 16] |= GROUP_SOUND;
            }
        }
    }

    // enclosing node of every unit, and the meshes: the topmost groups outside any csg whose
    // leaves are all triangles under one matrix
    std::vector<uint32_t> parent_unit((size_t)units + 4, NONE);
    MeshBuild meshes;
    meshes.stream = stream.data();
    uint32_t mesh_units = 0;
    {
        for (uint32_t i = 0; i < d->n_nodes; i++) {
            uint32_t pu = parent_node[i] == NONE ? NONE : unit_of[parent_node[i]];
            parent_unit[unit_of[i]] = pu;
            if (d->node_kind[i] == RT_NODE_LEAF) parent_unit[unit_of[i] + 1] = pu;
            if (cluster_end[i]) parent_unit[unit_of[i] - 1] = pu;
        }
        constexpr uint32_t MIXED = 0xFFFFFFFEu;           // more than one matrix, or something that is no triangle
        std::vector<uint32_t> matrix(d->n_nodes, NONE);   // NONE: nothing below yet
        std::vector<uint32_t> n_below(d->n_nodes, 0), levels(d->n_nodes, 0);
        for (uint32_t i = d->n_nodes; i-- > 0;) {         // children come after their parent: fold them in backwards
            if (d->node_kind[i] == RT_NODE_LEAF) {
                bool tri = d->leaf_shape[d->node_arg[i]] == RT_TRIANGLE;
                matrix[i] = tri ? stream[(size_t)unit_of[i] * 16 + 4] : MIXED;
                n_below[i] = tri ? 1 : 0;
            } else if (d->node_kind[i] == RT_NODE_CSG) {
                matrix[i] = MIXED;
            }
            uint32_t par = parent_node[i];
            if (par != NONE) {
                if (matrix[i] != NONE) matrix[par] = (matrix[par] == NONE || matrix[par] == matrix[i]) ? matrix[i] : MIXED;
                n_below[par] += n_below[i];
                levels[par] = std::max(levels[par], levels[i] + 1);
            }
        }
        for (uint32_t i = 0; i < d->n_nodes; i++) {
            if (d->node_kind[i] != RT_NODE_GROUP || matrix[i] == NONE || matrix[i] == MIXED) continue;
            if (parent_node[i] != NONE && matrix[parent_node[i]] != MIXED) continue;   // part of a larger mesh
            if (n_below[i] < MESH_MIN_TRIS || n_below[i] >= (1u << 27) || levels[i] >= (uint32_t)BVH_STACK - 2) continue;
            if (in_csg[i]) continue;
            std::vector<MeshTri> tris;
            tris.reserve(n_below[i]);
            bool finite = true;
            for (uint32_t j = i + 1; j < d->node_skip[i]; j++) {
                if (d->node_kind[j] != RT_NODE_LEAF) continue;
                const double* v = (const double*)&stream[(size_t)unit_of[j] * 16] + 4;  // p1, e1, e2
                MeshTri mt;
                mt.pos = unit_of[j];
                for (int a = 0; a < 3; a++) {
                    const double p0 = v[a], p1 = v[a] + v[3 + a], p2 = v[a] + v[6 + a];
                    mt.lo[a] = std::min(p0, std::min(p1, p2));
                    mt.hi[a] = std::max(p0, std::max(p1, p2));
                    mt.c[a] = 0.5 * (mt.lo[a] + mt.hi[a]);
                    finite = finite && std::fabs(mt.lo[a]) < 1e30 && std::fabs(mt.hi[a]) < 1e30;  // also false for NaN
                }
                tris.push_back(mt);
            }
            if (!finite) continue;
            // the binary tree first, then four children a node; a hierarchy too deep for the walk's stack (three
            // entries a level) or with coordinates beyond f32 is dropped and the mesh walked like any group
            meshes.bin.clear();
            const size_t nodes_before = meshes.nodes.size(), tris_before = meshes.tris.size();
            const uint32_t bin_root = mesh_build(tris, 0, tris.size(), 0, meshes);
            int wide_depth = 0;
            const uint32_t wide_root = mesh_collapse(meshes, bin_root, 0, wide_depth);
            if (3 * wide_depth + 1 > BVH_STACK) {
                meshes.nodes.resize(nodes_before);
                meshes.tris.resize(tris_before);
                continue;
            }
            uint32_t* rec = &stream[(size_t)unit_of[i] * 16];
            rec[2] = wide_root;
            mesh_units += rec[1] - unit_of[i] - 1;
            rec[3] = matrix[i];
            rec[0] |= MESH_BIT;
            bool all_cast = true, none_cast = true;
            for (uint32_t j = i + 1; j < d->node_skip[i]; j++) {
                if (d->node_kind[j] != RT_NODE_LEAF) continue;
                const bool casts = d->leaf_casts_shadow[d->node_arg[j]] != 0;
                all_cast = all_cast && casts;
                none_cast = none_cast && !casts;
            }
            rec[0] |= (all_cast ? MESH_ALL_CAST : 0u) | (none_cast ? MESH_NONE_CAST : 0u);
            // groups whose box lies within the box of every group up to the root (always, for boxes
            // that Group::bounding_box made); parents come first, so one pass settles the chain
            for (uint32_t j = i + 1; j < d->node_skip[i]; j++) {
                if (d->node_kind[j] != RT_NODE_GROUP) continue;
                const uint32_t par = parent_node[j];
                const double* bj = d->node_bbox + (size_t)j * 6;
                const double* bp = d->node_bbox + (size_t)par * 6;
                bool within = true;
                for (int a = 0; a < 3; a++) within = within && bj[a] >= bp[a] && bj[3 + a] <= bp[3 + a];
                const bool par_ok = par == i || (stream[(size_t)unit_of[par] * 16] & NESTED_BIT);
                if (within && par_ok) stream[(size_t)unit_of[j] * 16] |= NESTED_BIT;
            }
        }
    }

    auto* s = new rt_scene();
    (void)hipGetDevice(&s->device);
    s->has_csg = has_csg;
    s->mesh_units = mesh_units;
    s->csg_list_bound = csg_list_bound;
    s->world_list_bound = world_list_bound;
    s->has_area = has_area;
    s->leaf_unit.assign(d->n_leaves, 0);
    for (uint32_t i = 0; i < d->n_nodes; i++)
        if (d->node_kind[i] == RT_NODE_LEAF) s->leaf_unit[d->node_arg[i]] = unit_of[i];
    auto keep = [&](void* p) { s->allocations.push_back(p); };
    int rc = RT_OK;
#define UP(vec, field)                                       \
    do {                                                     \
        decltype(vec)::value_type* p_ = nullptr;             \
        rc = upload(vec, &p_);                               \
        if (rc != RT_OK) { rt_scene_destroy(s); return rc; } \
        keep(p_);                                            \
        s->dev.field = p_;                                   \
    } while (0)

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
        bool area = d->light_kind[i] == RT_LIGHT_AREA;
        light_meta[(size_t)i * 4 + 0] = d->light_kind[i];
        light_meta[(size_t)i * 4 + 1] = area ? d->light_steps[(size_t)i * 2] : 1;
        light_meta[(size_t)i * 4 + 2] = area ? d->light_steps[(size_t)i * 2 + 1] : 1;
    }
    s->dev.n_units = units; s->dev.n_lights = d->n_lights;
    s->dev.n_materials = d->n_materials; s->dev.n_patterns = d->n_patterns;
    s->shade_lds = shade_tables(d->n_materials, d->n_patterns, units).total <= SHADE_LDS_DOUBLES;
    s->dev.class_words = (uint32_t)words;
    s->dev.draws_per_shade = (uint32_t)draws;
    s->dev.flags = 0;
    s->spawns = false;
    for (uint32_t i = 0; i < d->n_materials; i++) {
        if (d->material[(size_t)i * 7 + 5] != 0.0) s->dev.flags |= 1u;
        // reflected_colour / refracted_colour spawn for reflective > 0 and |transparency| > EPSILON (world.rs:129-177)
        if (d->material[(size_t)i * 7 + 4] > 0.0 || !(std::fabs(d->material[(size_t)i * 7 + 5]) <= EPS)) s->spawns = true;
    }
    {
        bool all_cast = true;
        for (uint32_t i = 0; i < d->n_leaves; i++) all_cast = all_cast && d->leaf_casts_shadow[i] != 0;
        // Stopping at the first hit pays in the long walks through a mesh; in a scene of a few objects
        // the wave finishes together anyway, and there the work counters keep describing the
        // reference's full walk.
        if (all_cast && units > PER_LANE_MIN_UNITS) s->dev.flags |= 2u;
        // nearly everything casts a shadow: two filtered walks (see NoRecord) beat one full one
        uint32_t n_dark = 0;
        for (uint32_t i = 0; i < d->n_leaves; i++) n_dark += d->leaf_casts_shadow[i] ? 0u : 1u;
        if (!all_cast && units > PER_LANE_MIN_UNITS && (uint64_t)n_dark * 20 <= d->n_leaves) s->dev.flags |= 4u;
    }
    UP(stream, stream); UP(xforms, xform); UP(tri_normals, tri_normals); UP(cull, cull); UP(csg_op, csg_op); UP(csg_bits, csg_bits);
    UP(mat, mat); UP(pat_meta, pat_meta); UP(pat_inv, pat_inv); UP(pat_col, pat_col); UP(light, light);
    UP(light_meta, light_meta);
    UP(meshes.nodes, bvh); UP(meshes.tris, bvh_tris); UP(parent_unit, parent);
#undef UP
    if (has_area) {
        const std::vector<unsigned long long> jump = rng_jump_matrices(draws);
        rc = upload(jump, &s->rng_jump);
        if (rc != RT_OK) { rt_scene_destroy(s); return rc; }
        keep(s->rng_jump);
    }
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    for (auto& w : s->ws) free_workspace(w);
    for (void* p : s->allocations) (void)hipFree(p);
    if (s->stage_rgb) (void)hipFree(s->stage_rgb);
    if (s->stage_rgb8) (void)hipFree(s->stage_rgb8);
    if (s->seeds) (void)hipFree(s->seeds);
    for (hipEvent_t e : s->copy_ev) (void)hipEventDestroy(e);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    for (void* b : s->bounce) if (b) (void)hipHostFree(b);
    delete s;
}

int rt_host_alloc(void** out, size_t bytes) {
    if (!out || !bytes) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return RT_OK;
}

int rt_host_free(void* p) {
    if (!p) return RT_OK;
    HIP_TRY(hipHostFree(p));
    return RT_OK;
}

int rt_set_option(rt_scene* s, int option, uint64_t value) {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    switch (option) {
        case RT_OPT_STREAMS:
            if (value < 1 || value > MAX_STREAMS) return fail(RT_ERR_INVALID, "streams must be 1..4");
            s->n_streams = (int)value;
            return RT_OK;
        case RT_OPT_CHUNK_PIXELS:
            if (value < 64) return fail(RT_ERR_INVALID, "chunk_pixels must be at least 64");
            s->chunk_pixels = (size_t)value;
            return RT_OK;
        case RT_OPT_RECS_PER_PIXEL:
            if (value < 1 || value > 4096) return fail(RT_ERR_INVALID, "recs_per_pixel must be in 1..4096");
            s->recs_per_pixel = (size_t)value;
            for (auto& w : s->ws) {  // shrink on the next call
                if (w.stream) (void)hipStreamSynchronize(w.stream);
                if (w.rec_slab) (void)hipFree(w.rec_slab);
                w.rec_slab = nullptr; w.rec_cap = 0;
                if (w.intensity) (void)hipFree(w.intensity);
                w.intensity = nullptr; w.intensity_cap = 0;
            }
            return RT_OK;
        case RT_OPT_TRAVERSAL:
            if (value > 4) return fail(RT_ERR_INVALID, "traversal must be 0 (auto), 1 (wave), 2 (lane), 3 (reference) or 4 (wave, meshes by lane)");
            s->traversal = (int)value;
            return RT_OK;
        case RT_OPT_MIN_CHUNKS:
            if (value < 1 || value > 64) return fail(RT_ERR_INVALID, "min_chunks must be in 1..64");
            s->min_chunks = (uint32_t)value;
            return RT_OK;
        case RT_OPT_PIN_OUTPUTS:
            s->pin_outputs = value != 0;
            return RT_OK;
        case RT_OPT_GRID_BLOCKS:
            if (value > 65536) return fail(RT_ERR_INVALID, "grid_blocks must be in 0..65536 (0: chosen by the number of streams)");
            s->grid_blocks = (uint32_t)value;
            return RT_OK;
    }
    return fail(RT_ERR_INVALID, "unknown option");
}

int rt_render(rt_scene* s, const rt_camera* cam, uint32_t depth, const uint64_t* row_seeds, const rt_partition* part,
              double* out_rgb, uint8_t* out_rgb8, int out_on_device, rt_stats* stats) {
    if (!s || !cam) return fail(RT_ERR_INVALID, "null argument");
    if (cam->width == 0 || cam->height == 0) return fail(RT_ERR_INVALID, "empty image");
    rt_partition p = part ? *part : rt_partition{0, 0, 1};
    if (p.world == 0) p.world = 1;
    if (p.world > 1 && (p.block_rows == 0 || p.block_rows % 8 != 0 || p.rank >= p.world))
        return fail(RT_ERR_INVALID, "partition needs block_rows a multiple of 8 and rank < world");
    if (s->has_area && !row_seeds) return fail(RT_ERR_INVALID, "a scene with an area light needs the per-row seeds");
    if (!s->spawns) depth = 0;  // no ray is ever spawned: the deeper levels' launches would find nothing to do
    HIP_TRY(hipSetDevice(s->device));
    if (s->has_area) {
        int rc = upload_seeds(s, row_seeds, cam->height);
        if (rc != RT_OK) return rc;
    }

    const uint32_t W = cam->width;
    const uint32_t rows = local_rows(cam, &p);
    RunTotals tot;
    uint32_t chunks = 0;
    HIP_TRY(hipDeviceSynchronize());  // the outputs may have work queued on them by the caller
    const auto t_start = std::chrono::steady_clock::now();

    // device outputs (stage when the caller's buffers are host memory)
    double* d_rgb = out_rgb;
    uint8_t* d_rgb8 = out_rgb8;
    size_t total_px = (size_t)rows * W;
    // Host outputs: page-locked memory (the caller's own, or locked here for the length of the call) takes each
    // chunk as it finishes, on a stream of its own, while the next chunk's kernels run; memory that cannot be
    // locked is filled by one pageable copy at the end.
    bool host_async = false;
    void* locked_here[2] = {nullptr, nullptr};
    unsigned char* async_dst[2] = {nullptr, nullptr};  // where the asynchronous copies land (f64, u8)
    bool via_bounce[2] = {false, false};
    if (!out_on_device && total_px > 0) {
        if (s->stage_pixels < total_px) {
            if (s->stage_rgb) HIP_TRY(hipFree(s->stage_rgb));
            if (s->stage_rgb8) HIP_TRY(hipFree(s->stage_rgb8));
            s->stage_rgb = nullptr; s->stage_rgb8 = nullptr;
            s->stage_pixels = 0;
            HIP_TRY(hipMalloc((void**)&s->stage_rgb, total_px * 3 * sizeof(double)));
            HIP_TRY(hipMalloc((void**)&s->stage_rgb8, total_px * 3));
            s->stage_pixels = total_px;
        }
        d_rgb = out_rgb ? s->stage_rgb : nullptr;
        d_rgb8 = out_rgb8 ? s->stage_rgb8 : nullptr;
        void* outs[2] = {out_rgb, out_rgb8};
        const size_t bytes[2] = {total_px * 3 * sizeof(double), total_px * 3};
        host_async = true;
        for (int k = 0; k < 2 && host_async; k++) {
            async_dst[k] = (unsigned char*)outs[k];
            if (!outs[k] || host_page_locked(outs[k])) continue;
            if (s->pin_outputs && hipHostRegister(outs[k], bytes[k], hipHostRegisterDefault) == hipSuccess) { locked_here[k] = outs[k]; continue; }
            (void)hipGetLastError();
            // not lockable in place (some drivers refuse ordinary memory): stage through the scene's own page-locked buffer
            if (s->bounce_bytes[k] < bytes[k]) {
                if (s->bounce[k]) (void)hipHostFree(s->bounce[k]);
                s->bounce[k] = nullptr; s->bounce_bytes[k] = 0;
                if (hipHostMalloc(&s->bounce[k], bytes[k], hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); s->bounce[k] = nullptr; host_async = false; break; }
                s->bounce_bytes[k] = bytes[k];
            }
            async_dst[k] = (unsigned char*)s->bounce[k];
            via_bounce[k] = true;
        }
        if (host_async && !s->copy_stream) {
            if (hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); host_async = false; }
        }
    }
    const size_t recs_per_pixel_before = s->recs_per_pixel;
    size_t copies = 0;
    uint32_t big_chunks = 0;

    auto frame = [&]() -> int {
    // row chunks: multiples of 8 rows, at most chunk_pixels pixels.  Each pass queues every
    // pending chunk (alternating over the workspaces so two chunks overlap on the GPU), then
    // waits; chunks whose rays did not fit their record arrays are split and go round again.
    // As few and as large chunks as fit (a level of many rays runs closer to the kernels' peak than
    // several small ones), but at least min_chunks for a frame of two megapixels or more, so that
    // one chunk's k_shade launches run beside the next one's k_trace, and a multiple of the
    // number of streams, so that the streams end together.  The record arrays of all workspaces
    // may take 80 % of the device's memory, and 90 % of what is free of it.
    const int n_ws = std::max(1, std::min(s->n_streams, MAX_STREAMS));
    const size_t frame_px = (size_t)rows * W;
    uint32_t chunk_rows = 8;
    auto plan = [&](size_t chunk_px) {
        size_t n_chunks = (frame_px + chunk_px - 1) / chunk_px;
        if (frame_px >= ((size_t)2 << 20)) n_chunks = std::max<size_t>(n_chunks, s->min_chunks);
        if (n_chunks > 1) n_chunks = (n_chunks + n_ws - 1) / n_ws * n_ws;
        chunk_rows = std::max(8u, (uint32_t)(((rows + n_chunks - 1) / n_chunks + 7) & ~(size_t)7));
    };
    size_t chunk_px = std::max<size_t>(s->chunk_pixels, 64);
    plan(chunk_px);
    // the device's memory is only asked about when a workspace would have to grow for this plan
    bool grows = false;
    for (int k = 0; k < n_ws; k++)
        grows = grows || s->ws[k].rec_cap < std::max<size_t>(s->recs_per_pixel * (size_t)std::min(chunk_rows, rows) * W, 4096);
    if (grows) {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && mem_total) {
            const size_t per_rec = REC_BYTES + 8 * (size_t)s->dev.n_lights, per_px = s->recs_per_pixel * per_rec;
            size_t held = 0;  // what this scene's workspaces hold already counts as available
            for (int k = 0; k < n_ws; k++) held += s->ws[k].rec_cap * per_rec;
            const size_t budget = std::min(mem_total / 10 * 8, (mem_free + held) / 10 * 9);
            chunk_px = std::max<size_t>(std::min(chunk_px, budget / ((size_t)n_ws * per_px)), (size_t)8 * W);
            plan(chunk_px);
        }
    }
    struct Job { uint32_t r0, nr; bool big; };  // big: rendered again with the lists in global memory
    struct Arrival { hipEvent_t copied; size_t at, n; };  // a chunk's outputs: first element and element count
    std::vector<Job> pending;
    std::vector<Arrival> arrivals;
    for (uint32_t r0 = 0; r0 < rows; r0 += chunk_rows) pending.push_back({r0, std::min(chunk_rows, rows - r0), false});
    while (!pending.empty()) {
        copies = 0;
        std::vector<Job> queued[MAX_STREAMS];
        size_t max_px = 0;
        for (const Job& j : pending) max_px = std::max(max_px, (size_t)j.nr * W);
        for (int k = 0; k < n_ws; k++) {
            size_t share = (pending.size() + n_ws - 1) / n_ws;
            int rc = ensure_workspace(s, s->ws[k], max_px, 0, depth, share);
            if (rc != RT_OK) return rc;
        }
        for (size_t i = 0; i < pending.size(); i++) {
            const Job& j = pending[i];
            const int k = (int)(i % n_ws);
            LevelArgs A{};
            A.width = W; A.chunk_row0 = j.r0; A.chunk_rows = j.nr;
            A.tiles_x = (W + 7) / 8;
            A.n_tiles = A.tiles_x * ((j.nr + 7) / 8);
            A.part_block = p.block_rows; A.part_rank = p.rank; A.part_world = p.world;
            std::memcpy(A.cam_inv, cam->inv, sizeof A.cam_inv);
            A.half_width = cam->half_width; A.half_height = cam->half_height; A.pixel_size = cam->pixel_size;
            FirstLevel first{true, (uint32_t)((size_t)j.nr * W), W, j.nr, s->seeds};
            int rc = queue_chunk_dispatch(s, s->ws[k], A, first, depth, d_rgb, d_rgb8, (size_t)j.r0 * W,
                                          queued[k].size(), tot, j.big);
            if (rc != RT_OK) return rc;
            queued[k].push_back(j);
            if (host_async) {
                // a chunk that has to go round again is copied again behind this copy, on the same stream
                while (s->copy_ev.size() <= copies + 1) {
                    hipEvent_t e = nullptr;
                    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    s->copy_ev.push_back(e);
                }
                hipEvent_t done = s->copy_ev[copies], copied = s->copy_ev[copies + 1];
                copies += 2;
                HIP_TRY(hipEventRecord(done, s->ws[k].stream));
                HIP_TRY(hipStreamWaitEvent(s->copy_stream, done, 0));
                const size_t at = (size_t)j.r0 * W * 3, n = (size_t)j.nr * W * 3;
                if (out_rgb) HIP_TRY(hipMemcpyAsync(async_dst[0] + at * sizeof(double), d_rgb + at, n * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
                if (out_rgb8) HIP_TRY(hipMemcpyAsync(async_dst[1] + at, d_rgb8 + at, n, hipMemcpyDeviceToHost, s->copy_stream));
                HIP_TRY(hipEventRecord(copied, s->copy_stream));
                arrivals.push_back({copied, at, n});
            }
        }
        // staged outputs: move each chunk on to the caller's memory as it arrives, while later chunks render
        if (via_bounce[0] || via_bounce[1]) {
            for (const Arrival& a : arrivals) {
                HIP_TRY(hipEventSynchronize(a.copied));
                if (via_bounce[0] && out_rgb) parallel_copy(out_rgb + a.at, async_dst[0] + a.at * sizeof(double), a.n * sizeof(double));
                if (via_bounce[1] && out_rgb8) parallel_copy(out_rgb8 + a.at, async_dst[1] + a.at, a.n);
            }
        }
        arrivals.clear();
        pending.clear();
        bool grow_budget = false;
        for (int k = 0; k < n_ws; k++) {
            std::vector<ChunkStatus> st;
            int rc = collect(s->ws[k], queued[k].size(), st, tot);
            if (rc != RT_OK) return rc;
            for (size_t i = 0; i < st.size(); i++) {
                const Job& j = queued[k][i];
                if (st[i].error & 1u) {
                    if (j.nr <= 8) {
                        grow_budget = true;  // cannot split further: grow the per-pixel record budget
                        pending.push_back(j);
                    } else {
                        uint32_t half = ((j.nr / 2) + 7) & ~7u;
                        pending.push_back({j.r0, half, j.big});
                        pending.push_back({j.r0 + half, j.nr - half, j.big});
                    }
                    continue;
                }
                if ((st[i].error & 2u) && !j.big) {  // a fixed per-ray list overflowed: the slow path
                    pending.push_back({j.r0, j.nr, true});
                    big_chunks++;
                    continue;
                }
                rc = check_status(st[i], j.big);
                if (rc != RT_OK) return rc;
                add_stats(tot, st[i]);
                chunks++;
            }
        }
        if (grow_budget) {
            // kept for the scene's later frames when this one succeeds: they would overflow the same way
            if (s->recs_per_pixel >= 4096) return fail(RT_ERR_CAPACITY, "the record arrays overflow even for an 8-row chunk");
            s->recs_per_pixel *= 4;
        }
    }
#ifdef RT_PHASE_TIMERS
    for (int k = 0; k < n_ws; k++) print_phases(s->ws[k]);
#endif
    if (host_async) {
        HIP_TRY(hipStreamSynchronize(s->copy_stream));
    } else if (!out_on_device) {
        if (out_rgb) HIP_TRY(hipMemcpy(out_rgb, d_rgb, total_px * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (out_rgb8) HIP_TRY(hipMemcpy(out_rgb8, d_rgb8, total_px * 3, hipMemcpyDeviceToHost));
    }
    return RT_OK;
    };
    int frame_rc = frame();
    std::string frame_error;
    if (frame_rc != RT_OK) {
        // Nothing of this frame may still run when the caller gets its buffers back: wait for every stream that
        // has work of it, whatever the waits themselves return, and keep the first error's message.
        frame_error = g_error;
        for (int k = 0; k < MAX_STREAMS; k++) if (s->ws[k].stream) (void)hipStreamSynchronize(s->ws[k].stream);
        if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
        (void)hipGetLastError();
        s->recs_per_pixel = recs_per_pixel_before;
    }
    for (void* q : locked_here) if (q) (void)hipHostUnregister(q);
    if (frame_rc != RT_OK) { g_error = frame_error; return frame_rc; }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
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
        stats->ms_shadow = tot.ms_shadow;
        stats->ms_shade = tot.ms_shade;
        stats->trace_launches = tot.trace_launches;
        stats->flop_trace = tot.flop_trace;
        stats->flop_shadow = tot.flop_shadow;
        stats->flop_light = tot.flop_light;
        stats->mesh_nodes = tot.mesh_nodes;
        stats->mesh_tris = tot.mesh_tris;
        stats->fused_shadows = fused_shadows(s, uses_lane_walk(s)) ? 1 : 0;
        stats->big_chunks = big_chunks;
    }
    return RT_OK;
}

// One frame over several scenes at once, each a replica of the same world on a GPU of its own: scene k renders
// the row blocks b with b % n == k into its own device buffers, and the blocks are then copied to their places
// in the caller's frame (one strided copy per scene and output).  Scenes on different devices run side by side,
// one host thread per device; scenes that share a device take turns.
int rt_render_multi(rt_scene* const* scenes, uint32_t n, const rt_camera* cam, uint32_t depth, const uint64_t* row_seeds,
                    uint32_t block_rows, double* out_rgb, uint8_t* out_rgb8, int out_on_device, rt_stats* stats) {
    if (!scenes || !cam || n == 0) return fail(RT_ERR_INVALID, "null argument");
    for (uint32_t k = 0; k < n; k++) if (!scenes[k]) return fail(RT_ERR_INVALID, "null scene");
    if (cam->width == 0 || cam->height == 0) return fail(RT_ERR_INVALID, "empty image");
    if (block_rows == 0 || block_rows % 8 != 0) return fail(RT_ERR_INVALID, "block_rows must be a multiple of 8");
    if (n == 1) return rt_render(scenes[0], cam, depth, row_seeds, nullptr, out_rgb, out_rgb8, out_on_device, stats);
    for (uint32_t k = 0; k < n; k++)
        for (uint32_t j = 0; j < k; j++)
            if (scenes[j] == scenes[k]) return fail(RT_ERR_INVALID, "every rank needs a scene of its own");
    const uint32_t W = cam->width, H = cam->height;
    const uint32_t n_blocks = (H + block_rows - 1) / block_rows;
    const int frame_device = scenes[0]->device;  // device outputs live on the first scene's GPU

    auto render_one = [&](uint32_t k, std::string& error) -> int {
        rt_scene* s = scenes[k];
        const rt_partition part{block_rows, k, n};
        int rc = hipSetDevice(s->device) == hipSuccess ? RT_OK : fail(RT_ERR_HIP, "hipSetDevice failed");
        const size_t px = (size_t)local_rows(cam, &part) * W;
        if (rc == RT_OK && px == 0) {
            if (stats) std::memset(&stats[k], 0, sizeof stats[k]);
            return RT_OK;
        }
        auto grow_stage = [&]() -> int {
            if (s->stage_pixels >= px) return RT_OK;
            if (s->stage_rgb) HIP_TRY(hipFree(s->stage_rgb));
            if (s->stage_rgb8) HIP_TRY(hipFree(s->stage_rgb8));
            s->stage_rgb = nullptr; s->stage_rgb8 = nullptr; s->stage_pixels = 0;
            HIP_TRY(hipMalloc((void**)&s->stage_rgb, px * 3 * sizeof(double)));
            HIP_TRY(hipMalloc((void**)&s->stage_rgb8, px * 3));
            s->stage_pixels = px;
            return RT_OK;
        };
        if (rc == RT_OK) rc = grow_stage();
        if (rc == RT_OK)
            rc = rt_render(s, cam, depth, row_seeds, &part, out_rgb ? s->stage_rgb : nullptr, out_rgb8 ? s->stage_rgb8 : nullptr, 1,
                           stats ? &stats[k] : nullptr);
        // this scene's blocks to their rows of the frame: local block j is global block k + j * n
        auto place = [&](unsigned char* dst, const unsigned char* src, size_t bytes_per_px) -> int {
            const size_t block_bytes = (size_t)block_rows * W * bytes_per_px;
            const uint32_t mine = (n_blocks + n - 1 - k) / n;                     // blocks of this rank
            const bool ragged = H % block_rows != 0 && (n_blocks - 1) % n == k;   // the frame's short last block is among them
            const uint32_t whole = ragged ? mine - 1 : mine;
            if (whole)
                HIP_TRY(hipMemcpy2D(dst + (size_t)k * block_bytes, (size_t)n * block_bytes, src, block_bytes, block_bytes, whole,
                                    hipMemcpyDefault));
            if (ragged)
                HIP_TRY(hipMemcpy(dst + (size_t)(n_blocks - 1) * block_bytes, src + (size_t)whole * block_bytes,
                                  (size_t)(H % block_rows) * W * bytes_per_px, hipMemcpyDefault));
            return RT_OK;
        };
        if (rc == RT_OK && out_rgb) rc = place((unsigned char*)out_rgb, (const unsigned char*)s->stage_rgb, 3 * sizeof(double));
        if (rc == RT_OK && out_rgb8) rc = place(out_rgb8, s->stage_rgb8, 3);
        if (rc != RT_OK) error = g_error;
        return rc;
    };

    if (out_on_device)  // let the other GPUs write into the frame's GPU where the hardware links them
        for (uint32_t k = 1; k < n; k++)
            if (scenes[k]->device != frame_device && hipSetDevice(scenes[k]->device) == hipSuccess) {
                (void)hipDeviceEnablePeerAccess(frame_device, 0);
                (void)hipGetLastError();
            }
    std::vector<int> rcs(n, RT_OK);
    std::vector<std::string> errors(n);
    std::vector<int> devices;
    for (uint32_t k = 0; k < n; k++)
        if (std::find(devices.begin(), devices.end(), scenes[k]->device) == devices.end()) devices.push_back(scenes[k]->device);
    auto run_device = [&](int device) {
        for (uint32_t k = 0; k < n; k++)
            if (scenes[k]->device == device) rcs[k] = render_one(k, errors[k]);
    };
    std::vector<std::thread> pool;
    for (size_t d = 1; d < devices.size(); d++) pool.emplace_back(run_device, devices[d]);
    run_device(devices[0]);
    for (auto& t : pool) t.join();
    (void)hipSetDevice(frame_device);
    for (uint32_t k = 0; k < n; k++)
        if (rcs[k] != RT_OK) return fail(rcs[k], "rank " + std::to_string(k) + ": " + errors[k]);
    return RT_OK;
}

int rt_colour_at(rt_scene* s, uint32_t n, const double* rays, uint32_t depth, const uint64_t* seeds, double* out_rgb) {
    if (!s || (n && (!rays || !out_rgb))) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    if (!s->spawns) depth = 0;
    HIP_TRY(hipSetDevice(s->device));
    if (s->has_area) {
        int rc = upload_seeds(s, seeds, n);  // a missing list seeds every ray's generator with 0
        if (rc != RT_OK) return rc;
    }
    Workspace& w = s->ws[0];
    double* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, (size_t)n * 3 * sizeof(double)));
    LevelArgs A{};
    A.part_world = 1;
    RunTotals tot;
    int rc = RT_OK;
    bool big = false;
    for (;;) {
        rc = ensure_workspace(s, w, 0, n, depth, 1);
        if (rc == RT_OK) rc = upload_rays(w, n, rays);
        if (rc != RT_OK) break;
        FirstLevel first{false, n, 1, n, s->seeds};
        rc = queue_chunk_dispatch(s, w, A, first, depth, d_out, nullptr, 0, 0, tot, big);
        if (rc != RT_OK) break;
        std::vector<ChunkStatus> st;
        rc = collect(w, 1, st, tot);
        if (rc != RT_OK) break;
        if (st[0].error & 1u) {
            if (s->recs_per_pixel >= 4096) { rc = fail(RT_ERR_CAPACITY, "the record arrays overflow"); break; }
            s->recs_per_pixel *= 4;
            continue;
        }
        if ((st[0].error & 2u) && !big) { big = true; continue; }  // a fixed per-ray list overflowed: the slow path
        rc = check_status(st[0], big);
        break;
    }
    if (rc == RT_OK) {
        hipError_t e = hipMemcpy(out_rgb, d_out, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_out);
    return rc;
}

int rt_lighting(rt_scene* s, uint32_t n, const double* in, const uint64_t* seeds, double* out_rgb) {
    if (!s || (n && (!in || !out_rgb))) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    std::vector<double> host(in, in + (size_t)n * 12);
    std::vector<unsigned long long> host_seeds(n, 0ull);
    for (uint32_t i = 0; i < n; i++) {
        double* q = &host[(size_t)i * 12];
        if (!(q[0] >= 0.0 && q[0] < (double)s->leaf_unit.size())) return fail(RT_ERR_INVALID, "leaf index out of range");
        if (!(q[1] >= 0.0 && q[1] < (double)s->dev.n_lights)) return fail(RT_ERR_INVALID, "light index out of range");
        q[0] = (double)s->leaf_unit[(size_t)q[0]];
        if (seeds) host_seeds[i] = seeds[i];
    }
    double *d_in = nullptr, *d_out = nullptr;
    unsigned long long* d_seeds = nullptr;
    hipError_t e = hipMalloc((void**)&d_in, host.size() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, (size_t)n * 3 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_seeds, (size_t)n * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpy(d_in, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_seeds, host_seeds.data(), (size_t)n * sizeof(unsigned long long), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_lighting, dim3((n + 63) / 64), dim3(64), 0, nullptr, s->dev, n, d_in, d_seeds, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out_rgb, d_out, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_seeds);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

int rt_intensity_at(rt_scene* s, uint32_t light, uint32_t n, const double* points, uint64_t seed, double* out) {
    if (!s || (n && (!points || !out))) return fail(RT_ERR_INVALID, "null argument");
    if (light >= s->dev.n_lights) return fail(RT_ERR_INVALID, "light index out of range");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    double *d_pts = nullptr, *d_out = nullptr;
    Counters* d_counters = nullptr;
    hipError_t e = hipMalloc((void**)&d_pts, (size_t)n * 3 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_counters, sizeof(Counters));
    if (e == hipSuccess) e = hipMemset(d_counters, 0, sizeof(Counters));
    if (e == hipSuccess) e = hipMemcpy(d_pts, points, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_intensity_at, dim3(1), dim3(64), 0, nullptr, s->dev, light, n, d_pts, (unsigned long long)seed, d_out, d_counters);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    Counters c{};
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&c, d_counters, sizeof c, hipMemcpyDeviceToHost);
    (void)hipFree(d_pts); (void)hipFree(d_out); (void)hipFree(d_counters);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    if (c.error & 2u) return fail(RT_ERR_CAPACITY, "a per-ray intersection list overflowed");
    if (c.error & 4u) return fail(RT_ERR_NAN, "NaN intersection distance (the reference panics while sorting)");
    return RT_OK;
}

uint32_t rt_noalloc_max_rays(void) { return RT_NOALLOC_MAX_RAYS; }

// rt_colour_at with every device buffer carved out of the static `scratch` array: no hipMalloc, no
// hipFree, no stream or event creation.  Same kernels, same results.
int rt_colour_at_noalloc(rt_scene* s, uint32_t n, const double* rays_dev, uint32_t depth, double* out_rgb_dev) {
    if (!s || (n && (!rays_dev || !out_rgb_dev))) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    if (n > RT_NOALLOC_MAX_RAYS)
        return fail(RT_ERR_CAPACITY, "rt_colour_at_noalloc takes at most " + std::to_string(RT_NOALLOC_MAX_RAYS) +
                                         " rays per call (the static scratch is " + std::to_string(SCRATCH_BYTES >> 20) + " MiB)");
    if (depth > RT_NOALLOC_MAX_DEPTH)
        return fail(RT_ERR_CAPACITY, "rt_colour_at_noalloc takes a depth of at most " + std::to_string(RT_NOALLOC_MAX_DEPTH));
    if (s->has_area)
        return fail(RT_ERR_UNSUPPORTED, "rt_colour_at_noalloc: a scene with an area light sizes its rng-order buffers from "
                                        "device-side counts, which needs allocations");
    if (!s->spawns) depth = 0;
    HIP_TRY(hipSetDevice(s->device));
    unsigned char* base = nullptr;
    int rc = scratch_of_current_device(&base);
    if (rc != RT_OK) return rc;

    // carve the scratch: fixed-size parts first, the record columns take the rest
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char* p = base + off; off += (bytes + 255) & ~(size_t)255; return p; };
    Workspace w;                      // lives on this stack frame only; owns nothing
    w.stream = nullptr;               // the null stream
    w.counters = (Counters*)take(sizeof(Counters));
    w.stats = (StatShard*)take(STAT_SHARDS * sizeof(StatShard));
    w.status = (ChunkStatus*)take(sizeof(ChunkStatus));
    w.levels = (LevelState*)take(((size_t)RT_NOALLOC_MAX_DEPTH + 2) * sizeof(LevelState));
    const size_t per_rec = REC_BYTES + 8 * (size_t)s->dev.n_lights;
    const size_t cap = ((SCRATCH_BYTES - off - 1024) / per_rec) & ~(size_t)63;  // 64 records = 512 bytes of a f64 column; room for the rounding in take()
    if (cap < n) return fail(RT_ERR_CAPACITY, "the static scratch cannot hold one record per ray for this scene's lights");
    w.R.base = take(cap * REC_BYTES);
    w.intensity = (double*)take(cap * s->dev.n_lights * sizeof(double));
    if (off > SCRATCH_BYTES) return fail(RT_ERR_CAPACITY, "static scratch layout overflow");
    w.rec_cap = cap;

    // the array keeps the previous call's contents: the statistics lines must start at zero (the
    // level table and the error word are reset by k_chunk_begin)
    constexpr uint32_t stat_words = STAT_SHARDS * sizeof(StatShard) / sizeof(unsigned long long);
    hipLaunchKernelGGL(k_zero_words, dim3((stat_words + 255) / 256), dim3(256), 0, w.stream, (unsigned long long*)w.stats, stat_words);
    hipLaunchKernelGGL(k_load_rays, dim3((n + 255) / 256), dim3(256), 0, w.stream, rays_dev, n, w.R);
    LevelArgs A{};
    A.part_world = 1;
    RunTotals tot;
    FirstLevel first{false, n, 1, n, nullptr};
    rc = queue_chunk_dispatch(s, w, A, first, depth, out_rgb_dev, nullptr, 0, 0, tot);
    hipError_t e = hipStreamSynchronize(w.stream);  // also after a failed queueing: nothing may still write into the scratch
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    ChunkStatus st{};
    HIP_TRY(hipMemcpy(&st, w.status, sizeof st, hipMemcpyDeviceToHost));
    if (st.error & 1u)
        return fail(RT_ERR_CAPACITY, "rt_colour_at_noalloc: the rays spawned more than the " + std::to_string(cap) +
                                         " ray records the static scratch holds; pass fewer rays per call or a smaller depth");
    return check_status(st);
}

int rt_probe(rt_scene* s, uint32_t n, const double* rays, double* out) {
    if (!s || (n && (!rays || !out))) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    Workspace& w = s->ws[0];
    int rc = ensure_workspace(s, w, 0, n, 0, 1);
    if (rc != RT_OK) return rc;
    rc = upload_rays(w, n, rays);
    if (rc != RT_OK) return rc;
    double* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, (size_t)n * 32 * sizeof(double)));
    HIP_TRY(hipMemsetAsync(w.counters, 0, sizeof(Counters), w.stream));
    uint32_t blocks = (n + 255) / 256;
    if (s->has_csg) hipLaunchKernelGGL((k_probe<true>), dim3(blocks), dim3(256), 0, w.stream, s->dev, n, w.R, d_out, w.counters);
    else hipLaunchKernelGGL((k_probe<false>), dim3(blocks), dim3(256), 0, w.stream, s->dev, n, w.R, d_out, w.counters);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(w.stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 32 * sizeof(double), hipMemcpyDeviceToHost);
    Counters c{};
    if (e == hipSuccess) e = hipMemcpy(&c, w.counters, sizeof c, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    if (c.error & 2u) return fail(RT_ERR_CAPACITY, "a per-ray intersection list overflowed");
    if (c.error & 4u) return fail(RT_ERR_NAN, "NaN intersection distance (the reference panics while sorting)");
    return RT_OK;
}

}  // extern "C"
