// Host-side scene model: the object tree exactly as the reference builds it
// (push-down of group transforms, stored bounding boxes, divide), materials,
// patterns, lights and the camera. The device never sees this tree; it sees
// the SoA buffers that flatten.h derives from it.
//
// reference: src/object/mod.rs:36-151, src/object/shape.rs:24-108,
// src/object/shapes/*.rs (bounding boxes), src/object/bounding_box/mod.rs:28-210,
// src/object/group/mod.rs:29-153, src/object/group/helper.rs:42-65,
// src/object/csg/mod.rs:23-150, src/material.rs:16-105, src/pattern/mod.rs:38-118,
// src/light/area.rs:22-45, src/camera.rs:35-87.
#pragma once

#include <cmath>
#include <limits>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "rmath.h"

namespace rt {

constexpr double INF = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------- patterns
enum class PatKind : uint32_t {
    Solid = 0, Stripe = 1, Ring = 2, Checker = 3, Gradient = 4, RadialGradient = 5,
    Blend = 6, Perturbed = 7
};

struct Pattern {
    PatKind kind = PatKind::Solid;
    Mat4 transformation = Mat4::identity();
    Mat4 inverse = Mat4::identity();
    V3 colour{1, 1, 1};                       // Solid
    std::shared_ptr<const Pattern> a, b;      // two-pattern kinds; Perturbed uses `a`
    double scale = 0.0;                       // Perturbed
};
using PatternPtr = std::shared_ptr<const Pattern>;

inline PatternPtr solid_pattern(V3 c) {
    auto p = std::make_shared<Pattern>();
    p->kind = PatKind::Solid;
    p->colour = c;
    return p;
}
inline PatternPtr make_pattern(PatKind kind, PatternPtr a, PatternPtr b, const Mat4& t) {
    auto p = std::make_shared<Pattern>();
    p->kind = kind;
    p->a = std::move(a);
    p->b = std::move(b);
    p->transformation = t;
    p->inverse = invert(t);
    return p;
}

// reference src/pattern/mod.rs:106-108 and the per-kind impl_approx_eq!s.
inline bool approx_eq(const Pattern& l, const Pattern& r) {
    if (l.kind != r.kind) return false;
    switch (l.kind) {
        case PatKind::Solid:
            if (!approx_eq(l.colour, r.colour)) return false;
            break;
        case PatKind::Perturbed:  // noise itself is ignored (perturbed.rs:43)
            if (!approx_eq(l.scale, r.scale) || !approx_eq(*l.a, *r.a)) return false;
            break;
        default:
            if (!approx_eq(*l.a, *r.a) || !approx_eq(*l.b, *r.b)) return false;
    }
    return approx_eq(l.transformation, r.transformation) && approx_eq(l.inverse, r.inverse);
}

// ---------------------------------------------------------------- material
struct Material {  // defaults: reference src/material.rs:16-31
    PatternPtr pattern = solid_pattern({1, 1, 1});
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
    double reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
};
inline Material glass_material() {  // src/material.rs:36-43
    Material m;
    m.ambient = 0.01; m.diffuse = 0.01; m.transparency = 1.0; m.refractive_index = 1.5;
    return m;
}
inline bool approx_eq(const Material& l, const Material& r) {
    return approx_eq(*l.pattern, *r.pattern) && approx_eq(l.ambient, r.ambient) &&
           approx_eq(l.diffuse, r.diffuse) && approx_eq(l.specular, r.specular) &&
           approx_eq(l.shininess, r.shininess) && approx_eq(l.reflective, r.reflective) &&
           approx_eq(l.transparency, r.transparency) &&
           approx_eq(l.refractive_index, r.refractive_index);
}

// ---------------------------------------------------------------- bounding box
struct BBox {  // default = inverted-infinite (bounding_box/mod.rs:203-210)
    V3 min{INF, INF, INF};
    V3 max{-INF, -INF, -INF};

    // f64::min/max ignore a NaN operand, like fmin/fmax (mod.rs:28-36).
    void add_point(V3 p) {
        min.x = std::fmin(min.x, p.x); min.y = std::fmin(min.y, p.y); min.z = std::fmin(min.z, p.z);
        max.x = std::fmax(max.x, p.x); max.y = std::fmax(max.y, p.y); max.z = std::fmax(max.z, p.z);
    }
    void add_box(const BBox& b) { add_point(b.min); add_point(b.max); }
    bool contains(V3 p) const {
        return min.x <= p.x && p.x <= max.x && min.y <= p.y && p.y <= max.y &&
               min.z <= p.z && p.z <= max.z;
    }
    bool contains_box(const BBox& b) const { return contains(b.min) && contains(b.max); }
    // mod.rs:182-201: the eight corners in this order
    BBox apply(const Mat4& t) const {
        BBox r;
        r.add_point(mul_point(t, min));
        r.add_point(mul_point(t, {min.x, min.y, max.z}));
        r.add_point(mul_point(t, {min.x, max.y, min.z}));
        r.add_point(mul_point(t, {min.x, max.y, max.z}));
        r.add_point(mul_point(t, {max.x, min.y, min.z}));
        r.add_point(mul_point(t, {max.x, min.y, max.z}));
        r.add_point(mul_point(t, {max.x, max.y, min.z}));
        r.add_point(mul_point(t, max));
        return r;
    }
    // mod.rs:115-141
    std::pair<BBox, BBox> split() const {
        double dx = std::fabs(max.x - min.x), dy = std::fabs(max.y - min.y),
               dz = std::fabs(max.z - min.z);
        double x0 = min.x, y0 = min.y, z0 = min.z, x1 = max.x, y1 = max.y, z1 = max.z;
        if (dx >= dy && dx >= dz) { x0 += dx / 2.0; x1 = x0; }
        else if (dy >= dz)        { y0 += dy / 2.0; y1 = y0; }
        else                      { z0 += dz / 2.0; z1 = z0; }
        BBox l, r;
        l.min = min; l.max = {x1, y1, z1};
        r.min = {x0, y0, z0}; r.max = max;
        return {l, r};
    }
};

// ---------------------------------------------------------------- objects
enum class ShapeKind : uint32_t { Sphere = 0, Plane = 1, Cube = 2, Cylinder = 3, Cone = 4, Triangle = 5 };
enum class CsgOp : uint32_t { Difference = 0, Intersection = 1, Union = 2 };
enum class ObjType : uint32_t { Shape = 0, Group = 1, Csg = 2 };

struct Object {
    ObjType type = ObjType::Shape;

    // Shape
    ShapeKind shape = ShapeKind::Sphere;
    double minimum = -INF, maximum = INF;  // cylinder / cone
    bool closed = false;
    V3 p1, p2, p3, e1, e2, n1, n2, n3;     // triangle
    Mat4 transformation = Mat4::identity();
    Mat4 inverse = Mat4::identity();
    Material material;
    bool casts_shadow = true;

    // Group: children in stored order.  Csg: children[0] = left, children[1] = right.
    std::vector<Object> children;
    CsgOp op = CsgOp::Union;
    BBox bbox;  // the *stored* box of a Group / Csg (may be stale for a Csg)
};

inline Object make_shape(ShapeKind k, const Mat4& t = Mat4::identity(),
                         const Material& m = Material(), bool casts_shadow = true) {
    Object o;
    o.type = ObjType::Shape;
    o.shape = k;
    o.transformation = t;
    o.inverse = invert(t);
    o.material = m;
    o.casts_shadow = casts_shadow;
    return o;
}
inline Object make_cyl_or_cone(ShapeKind k, double mn, double mx, bool closed,
                               const Mat4& t = Mat4::identity(), const Material& m = Material(),
                               bool casts_shadow = true) {
    Object o = make_shape(k, t, m, casts_shadow);
    o.minimum = mn; o.maximum = mx; o.closed = closed;
    return o;
}
// reference src/object/shapes/triangle.rs:23-63
inline Object make_triangle(V3 p1, V3 p2, V3 p3, V3 n1, V3 n2, V3 n3) {
    Object o = make_shape(ShapeKind::Triangle);
    o.p1 = p1; o.p2 = p2; o.p3 = p3;
    o.e1 = p2 - p1; o.e2 = p3 - p1;
    o.n1 = n1; o.n2 = n2; o.n3 = n3;
    return o;
}
inline Object make_flat_triangle(V3 p1, V3 p2, V3 p3) {
    V3 e1 = p2 - p1, e2 = p3 - p1;
    V3 n = normalise(cross(e2, e1));
    return make_triangle(p1, p2, p3, n, n, n);
}

// Object-space box of a primitive (each shapes/*.rs `impl Bounded`).
inline BBox shape_bounds(const Object& o) {
    BBox b;
    switch (o.shape) {
        case ShapeKind::Sphere:
        case ShapeKind::Cube:
            b.min = {-1, -1, -1}; b.max = {1, 1, 1};
            break;
        case ShapeKind::Plane:
            b.min = {-INF, 0, -INF}; b.max = {INF, 0, INF};
            break;
        case ShapeKind::Cylinder:
            b.min = {-1, o.minimum, -1}; b.max = {1, o.maximum, 1};
            break;
        case ShapeKind::Cone: {
            double limit = std::fmax(std::fabs(o.minimum), std::fabs(o.maximum));
            b.min = {-limit, o.minimum, -limit}; b.max = {limit, o.maximum, limit};
            break;
        }
        case ShapeKind::Triangle:
            b.add_point(o.p1); b.add_point(o.p2); b.add_point(o.p3);
            break;
    }
    return b;
}

// `Bounded::bounding_box`: always recomputed from the leaves, never the
// stored field (shape.rs:88-95, group/mod.rs:130-140, csg/mod.rs:134-139).
inline BBox bounding_box(const Object& o) {
    switch (o.type) {
        case ObjType::Shape:
            return shape_bounds(o).apply(o.transformation);
        case ObjType::Group: {
            BBox b;
            for (const Object& c : o.children) b.add_box(bounding_box(c));
            return b;
        }
        case ObjType::Csg: {
            BBox l = bounding_box(o.children[0]), r = bounding_box(o.children[1]);
            BBox b;  // `Add for BoundingBox`: component-wise min / max
            b.min = {std::fmin(l.min.x, r.min.x), std::fmin(l.min.y, r.min.y), std::fmin(l.min.z, r.min.z)};
            b.max = {std::fmax(l.max.x, r.max.x), std::fmax(l.max.y, r.max.y), std::fmax(l.max.z, r.max.z)};
            return b;
        }
    }
    return BBox();
}

// `Updatable` (shape.rs:73-86, group/mod.rs:108-128, csg/mod.rs:117-132).
inline void update_transformation(Object& o, const Mat4& t) {
    switch (o.type) {
        case ObjType::Shape:
            o.transformation = t * o.transformation;
            o.inverse = invert(o.transformation);
            break;
        case ObjType::Group:
            for (Object& c : o.children) update_transformation(c, t);
            o.bbox = bounding_box(o);
            break;
        case ObjType::Csg:  // the stored box is NOT refreshed (reference quirk)
            update_transformation(o.children[0], t);
            update_transformation(o.children[1], t);
            break;
    }
}
inline void replace_material(Object& o, const Material& m) {
    if (o.type == ObjType::Shape) { o.material = m; return; }
    for (Object& c : o.children) replace_material(c, m);
}
inline void update_casts_shadow(Object& o, bool s) {
    if (o.type == ObjType::Shape) { o.casts_shadow = s; return; }
    for (Object& c : o.children) update_casts_shadow(c, s);
}

// GroupBuilder::build (group/helper.rs:42-65)
inline Object build_group(std::vector<Object> objects, const Mat4& t = Mat4::identity(),
                          const Material* material = nullptr, const bool* casts_shadow = nullptr) {
    Object g;
    g.type = ObjType::Group;
    g.children = std::move(objects);
    update_transformation(g, t);
    if (material) replace_material(g, *material);
    if (casts_shadow) update_casts_shadow(g, *casts_shadow);
    return g;
}
// Csg::new (csg/mod.rs:23-34)
inline Object make_csg(CsgOp op, Object left, Object right) {
    Object c;
    c.type = ObjType::Csg;
    c.op = op;
    c.children.push_back(std::move(left));
    c.children.push_back(std::move(right));
    c.bbox = bounding_box(c);
    return c;
}

// Group::divide / partition (group/mod.rs:50-105), Csg::divide (csg/mod.rs:109-114)
inline Object divide(Object o, uint32_t threshold) {
    if (o.type == ObjType::Shape) return o;
    if (o.type == ObjType::Csg) {
        o.children[0] = divide(std::move(o.children[0]), threshold);
        o.children[1] = divide(std::move(o.children[1]), threshold);
        return o;
    }
    if (o.children.size() >= (size_t)threshold) {
        auto halves = o.bbox.split();
        std::vector<Object> left, right, neither;
        for (Object& c : o.children) {
            BBox b = bounding_box(c);
            if (halves.first.contains_box(b)) left.push_back(std::move(c));
            else if (halves.second.contains_box(b)) right.push_back(std::move(c));
            else neither.push_back(std::move(c));
        }
        o.children = std::move(neither);
        if (!left.empty()) o.children.push_back(build_group(std::move(left)));
        if (!right.empty()) o.children.push_back(build_group(std::move(right)));
    }
    for (Object& c : o.children) c = divide(std::move(c), threshold);
    return o;
}

// Structural equality of two leaves: shape kind/params, transformation and
// material; casts_shadow and the inverse are not compared (shape.rs:108).
inline bool shape_approx_eq(const Object& l, const Object& r) {
    if (l.shape != r.shape) return false;
    switch (l.shape) {
        case ShapeKind::Cylinder:
        case ShapeKind::Cone:
            if (l.closed != r.closed || !approx_eq(l.minimum, r.minimum) ||
                !approx_eq(l.maximum, r.maximum))
                return false;
            break;
        case ShapeKind::Triangle:
            if (!approx_eq(l.p1, r.p1) || !approx_eq(l.p2, r.p2) || !approx_eq(l.p3, r.p3) ||
                !approx_eq(l.n1, r.n1) || !approx_eq(l.n2, r.n2) || !approx_eq(l.n3, r.n3))
                return false;
            break;
        default:
            break;
    }
    return approx_eq(l.transformation, r.transformation) && approx_eq(l.material, r.material);
}

// ---------------------------------------------------------------- lights
struct Light {
    bool is_area = false;
    V3 position;               // point light
    V3 corner, u, v;           // area light: u, v are already divided by the steps
    uint32_t u_steps = 1, v_steps = 1, samples = 1;
    V3 intensity{1, 1, 1};
};
inline Light point_light(V3 at, V3 intensity) {
    Light l;
    l.position = at; l.intensity = intensity;
    return l;
}
// Area::new (light/area.rs:22-45)
inline Light area_light(V3 corner, V3 u, uint32_t us, V3 v, uint32_t vs, V3 intensity) {
    Light l;
    l.is_area = true;
    l.corner = corner;
    l.u = u / (double)us; l.v = v / (double)vs;
    l.u_steps = us; l.v_steps = vs; l.samples = us * vs;
    l.intensity = intensity;
    return l;
}

// ---------------------------------------------------------------- camera
struct Camera {
    uint32_t hsize = 0, vsize = 0;
    double field_of_view = 0;
    Mat4 inverse = Mat4::identity();
    double half_width = 0, half_height = 0, pixel_size = 0;
    // tan() comes from the platform libm in the reference, so a render made on
    // another platform can sit one ulp away from glibc's value. This moves
    // tan(fov/2) by whole ulps; tests use it to reproduce such a render. 0 = off.
    int half_view_ulps = 0;

    // Camera::calculate (camera.rs:55-73)
    void calculate() {
        double half_view = std::tan(field_of_view / 2.0);
        for (int n = half_view_ulps; n > 0; n--) half_view = std::nextafter(half_view, INF);
        for (int n = half_view_ulps; n < 0; n++) half_view = std::nextafter(half_view, -INF);
        double hf = (double)hsize;
        double aspect = hf / (double)vsize;
        if (aspect > 1.0) { half_width = half_view; half_height = half_view / aspect; }
        else              { half_width = half_view * aspect; half_height = half_view; }
        pixel_size = half_width * 2.0 / hf;
    }
    static Camera make(uint32_t h, uint32_t v, double fov, const Mat4& transformation) {
        Camera c;
        c.hsize = h; c.vsize = v; c.field_of_view = fov;
        c.inverse = invert(transformation);
        c.calculate();
        return c;
    }
    // Camera::scale (camera.rs:78-87): truncating casts
    void scale(double s) {
        hsize = (uint32_t)((double)hsize * s);
        vsize = (uint32_t)((double)vsize * s);
        calculate();
    }
};

struct Scene {
    Camera camera;
    std::vector<Object> objects;  // world.objects order
    std::vector<Light> lights;
};

}  // namespace rt
