// ORACLE - TEST INFRASTRUCTURE ONLY.
//
// A plain CPU restatement of the reference's rendering path, recursive and
// list-based exactly like the Rust code, used as the checker for the HIP
// path and as the same-box CPU baseline.  Nothing in the product path may
// link or call this file; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py do.
//
// It walks the host object tree of raytracer_amd/csrc/host/scene.h (the
// scene-construction side, which is product code and is itself pinned by the
// reference's known answers and golden renders through this oracle).
//
// Parity is pinned by: the reference's known-answer unit tests (ported in
// tests/test_oracle_known_answers.py) and the reference's own renders
// images/*.png sampled into tests/golden/*.npz.
//
// Each function cites the reference lines it follows.  Build with
// -O2 -ffp-contract=off (Rust never contracts a*b+c).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../raytracer_amd/csrc/host/loader.h"
#include "../raytracer_amd/csrc/host/scene.h"

using namespace rt;

namespace {

struct Ray { V3 origin, direction; };
inline V3 position(const Ray& r, double t) { return r.origin + r.direction * t; }  // ray.rs:17-19

struct Isect {  // intersection/mod.rs:22-26
    const Object* object;
    double t;
    double u, v;
};
using List = std::vector<Isect>;

struct NanT {};

// List::sort (intersection/list.rs:39-50): stable; a NaN t panics.
void sort_list(List& l) {
    if (l.size() >= 2)  // a one-element sort never calls the comparator
        for (const Isect& i : l)
            if (std::isnan(i.t)) throw NanT();
    std::stable_sort(l.begin(), l.end(), [](const Isect& a, const Isect& b) { return a.t < b.t; });
}

// BoundingBox::check_axis / intersect (bounding_box/mod.rs:57-112).
inline void check_axis(double origin, double direction, double mn, double mx, double& tmin, double& tmax) {
    double min_numerator = mn - origin, max_numerator = mx - origin;
    double a, b;
    if (approx_eq(direction, 0.0)) { a = min_numerator * INF; b = max_numerator * INF; }
    else { a = min_numerator / direction; b = max_numerator / direction; }
    if (a > b) { tmin = b; tmax = a; } else { tmin = a; tmax = b; }
}
inline bool box_intersect(const Ray& r, V3 mn, V3 mx, double& tmin, double& tmax) {
    double x0, x1, y0, y1, z0, z1;
    check_axis(r.origin.x, r.direction.x, mn.x, mx.x, x0, x1);
    check_axis(r.origin.y, r.direction.y, mn.y, mx.y, y0, y1);
    check_axis(r.origin.z, r.direction.z, mn.z, mx.z, z0, z1);
    tmin = std::fmax(std::fmax(x0, y0), z0);  // f64::max ignores NaN
    tmax = std::fmin(std::fmin(x1, y1), z1);
    return !(tmin > tmax || tmax < 0.0);
}

// Shapes::intersect on the object-space ray; appends in the reference's emit order.
void shape_intersect(const Object& o, const Ray& r, List& out) {
    const V3 &ro = r.origin, &rd = r.direction;
    switch (o.shape) {
        case ShapeKind::Sphere: {  // sphere.rs:15-35
            V3 s2r = ro - V3{0, 0, 0};
            double a = dot(rd, rd);
            double b = 2.0 * dot(rd, s2r);
            double c = dot(s2r, s2r) - 1.0;
            double disc = b * b - 4.0 * a * c;
            if (disc < 0.0) return;
            double sq = std::sqrt(disc);
            double a2 = 2.0 * a;
            out.push_back({&o, (-b - sq) / a2, 0, 0});
            out.push_back({&o, (-b + sq) / a2, 0, 0});
            return;
        }
        case ShapeKind::Plane: {  // plane.rs:17-23
            if (approx_eq(rd.y, 0.0)) return;
            out.push_back({&o, -ro.y / rd.y, 0, 0});
            return;
        }
        case ShapeKind::Cube: {  // cube.rs:15-21
            double t0, t1;
            if (!box_intersect(r, {-1, -1, -1}, {1, 1, 1}, t0, t1)) return;
            out.push_back({&o, t0, 0, 0});
            out.push_back({&o, t1, 0, 0});
            return;
        }
        case ShapeKind::Cylinder: {  // cylinder.rs:26-95
            auto caps = [&]() {
                if (o.closed && !approx_eq(rd.y, 0.0)) {
                    auto check = [&](double t) {
                        double x = ro.x + t * rd.x, z = ro.z + t * rd.z;
                        return x * x + z * z <= 1.0;
                    };
                    double t = (o.minimum - ro.y) / rd.y;
                    if (check(t)) out.push_back({&o, t, 0, 0});
                    t = (o.maximum - ro.y) / rd.y;
                    if (check(t)) out.push_back({&o, t, 0, 0});
                }
            };
            double a = rd.x * rd.x + rd.z * rd.z;
            if (approx_eq(a, 0.0)) { caps(); return; }
            double b = 2.0 * (ro.x * rd.x + ro.z * rd.z);
            double c = ro.x * ro.x + ro.z * ro.z - 1.0;
            double disc = b * b - 4.0 * a * c;
            if (disc < 0.0) return;
            double sq = std::sqrt(disc);
            double a2 = 2.0 * a;
            double t0 = (-b - sq) / a2, t1 = (-b + sq) / a2;
            double y0 = ro.y + t0 * rd.y;
            if (o.minimum < y0 && y0 < o.maximum) out.push_back({&o, t0, 0, 0});
            double y1 = ro.y + t1 * rd.y;
            if (o.minimum < y1 && y1 < o.maximum) out.push_back({&o, t1, 0, 0});
            caps();
            return;
        }
        case ShapeKind::Cone: {  // cone.rs:26-100
            double a = rd.x * rd.x - rd.y * rd.y + rd.z * rd.z;
            double b = 2.0 * (ro.x * rd.x - ro.y * rd.y + ro.z * rd.z);
            double c = ro.x * ro.x - ro.y * ro.y + ro.z * ro.z;
            if (approx_eq(a, 0.0)) {
                if (!approx_eq(b, 0.0)) out.push_back({&o, -c / (2.0 * b), 0, 0});
            } else {
                double disc = b * b - 4.0 * a * c;
                if (disc < 0.0) return;
                double sq = std::sqrt(disc);
                double a2 = 2.0 * a;
                double t0 = (-b - sq) / a2, t1 = (-b + sq) / a2;
                double y0 = ro.y + t0 * rd.y;
                if (o.minimum < y0 && y0 < o.maximum) out.push_back({&o, t0, 0, 0});
                double y1 = ro.y + t1 * rd.y;
                if (o.minimum < y1 && y1 < o.maximum) out.push_back({&o, t1, 0, 0});
            }
            if (o.closed && !approx_eq(rd.y, 0.0)) {
                auto check = [&](double t, double radius) {
                    double x = ro.x + t * rd.x, z = ro.z + t * rd.z;
                    return x * x + z * z <= radius * radius;
                };
                double t = (o.minimum - ro.y) / rd.y;
                if (check(t, o.minimum)) out.push_back({&o, t, 0, 0});
                t = (o.maximum - ro.y) / rd.y;
                if (check(t, o.maximum)) out.push_back({&o, t, 0, 0});
            }
            return;
        }
        case ShapeKind::Triangle: {  // triangle.rs:68-96
            V3 dce2 = cross(rd, o.e2);
            double det = dot(o.e1, dce2);
            if (approx_eq(det, 0.0)) return;
            double f = 1.0 / det;
            V3 p1o = ro - o.p1;
            double u = f * dot(p1o, dce2);
            if (!(0.0 <= u && u <= 1.0)) return;
            V3 oce1 = cross(p1o, o.e1);
            double v = f * dot(rd, oce1);
            if (v < 0.0 || (u + v) > 1.0) return;
            double t = f * dot(o.e2, oce1);
            out.push_back({&o, t, u, v});
            return;
        }
    }
}

// Includes (shape.rs:97-106, group/mod.rs:142-153, csg/mod.rs:141-150)
bool includes(const Object& tree, const Object& leaf) {
    if (tree.type == ObjType::Shape) return shape_approx_eq(tree, leaf);
    for (const Object& c : tree.children)
        if (includes(c, leaf)) return true;
    return false;
}

// Object::intersect (object/mod.rs:100-106). Appends to `out`.
void object_intersect(const Object& o, const Ray& ray, List& out) {
    switch (o.type) {
        case ObjType::Shape: {  // shape.rs:49-57
            Ray local{mul_point(o.inverse, ray.origin), mul_vector(o.inverse, ray.direction)};
            shape_intersect(o, local, out);
            return;
        }
        case ObjType::Group: {  // group/mod.rs:29-47
            double t0, t1;
            if (!box_intersect(ray, o.bbox.min, o.bbox.max, t0, t1)) return;
            for (const Object& c : o.children) object_intersect(c, ray, out);
            return;
        }
        case ObjType::Csg: {  // csg/mod.rs:57-106
            double t0, t1;
            if (!box_intersect(ray, o.bbox.min, o.bbox.max, t0, t1)) return;
            List xs;
            object_intersect(o.children[0], ray, xs);
            object_intersect(o.children[1], ray, xs);
            sort_list(xs);
            bool in_left = false, in_right = false;
            for (const Isect& i : xs) {
                bool left_hit = includes(o.children[0], *i.object);
                bool allowed = false;
                switch (o.op) {  // csg/mod.rs:37-54
                    case CsgOp::Difference: allowed = (left_hit && !in_right) || (!left_hit && in_left); break;
                    case CsgOp::Intersection: allowed = (left_hit && in_right) || (!left_hit && in_left); break;
                    case CsgOp::Union: allowed = (left_hit && !in_right) || (!left_hit && !in_left); break;
                }
                if (allowed) out.push_back(i);
                if (left_hit) in_left = !in_left; else in_right = !in_right;
            }
            return;
        }
    }
}

// Shapes::normal_at in object space.
V3 shape_normal(const Object& o, V3 p, const Isect& hit) {
    switch (o.shape) {
        case ShapeKind::Sphere: return p - V3{0, 0, 0};
        case ShapeKind::Plane: return {0, 1, 0};
        case ShapeKind::Cube: {  // cube.rs:24-40
            double ax = std::fabs(p.x), ay = std::fabs(p.y), az = std::fabs(p.z);
            if (ax >= ay) {
                if (ax >= az) return {p.x, 0, 0};
                return {0, 0, p.z};
            } else if (ay >= az) {
                return {0, p.y, 0};
            }
            return {0, 0, p.z};
        }
        case ShapeKind::Cylinder: {  // cylinder.rs:98-108
            double dist = p.x * p.x + p.z * p.z;
            if (dist < 1.0 && p.y >= o.maximum - EPS) return {0, 1, 0};
            if (dist < 1.0 && p.y <= o.minimum + EPS) return {-0.0, -1.0, -0.0};
            return {p.x, 0.0, p.z};
        }
        case ShapeKind::Cone: {  // cone.rs:103-118
            double dist = p.x * p.x + p.z * p.z;
            if (dist < 1.0 && p.y >= o.maximum - EPS) return {0, 1, 0};
            if (dist < 1.0 && p.y <= o.minimum + EPS) return {-0.0, -1.0, -0.0};
            double y = std::sqrt(dist);
            if (p.y > 0.0) y = -y;
            return {p.x, y, p.z};
        }
        case ShapeKind::Triangle:  // triangle.rs:99-104
            return o.n2 * hit.u + o.n3 * hit.v + o.n1 * (1.0 - hit.u - hit.v);
    }
    return {0, 0, 0};
}

// Shape::normal_at (shape.rs:60-70): world normal = transpose(inverse) * n, normalised.
V3 normal_at(const Object& o, V3 world_point, const Isect& hit) {
    V3 op = mul_point(o.inverse, world_point);
    V3 on = shape_normal(o, op, hit);
    V3 wn = mul_vector(o.inverse.transpose(), on);
    return normalise(wn);
}

// Pattern::sub_pattern_at and the kinds (pattern/mod.rs:83-94, kind files).
V3 sub_pattern_at(const Pattern& p, V3 point) {
    V3 pp = mul_point(p.inverse, point);
    switch (p.kind) {
        case PatKind::Solid: return p.colour;
        case PatKind::Stripe:
            return std::fmod(std::floor(pp.x), 2.0) == 0.0 ? sub_pattern_at(*p.a, pp) : sub_pattern_at(*p.b, pp);
        case PatKind::Ring:
            return std::fmod(std::floor(std::hypot(pp.x, pp.z)), 2.0) == 0.0 ? sub_pattern_at(*p.a, pp)
                                                                             : sub_pattern_at(*p.b, pp);
        case PatKind::Checker:
            return std::fmod(std::floor(pp.x) + std::floor(pp.y) + std::floor(pp.z), 2.0) == 0.0
                       ? sub_pattern_at(*p.a, pp) : sub_pattern_at(*p.b, pp);
        case PatKind::Gradient: {
            V3 distance = sub_pattern_at(*p.b, pp) - sub_pattern_at(*p.a, pp);
            double fraction = pp.x - std::floor(pp.x);
            return sub_pattern_at(*p.a, pp) + distance * fraction;
        }
        case PatKind::RadialGradient: {
            V3 distance = sub_pattern_at(*p.b, pp) - sub_pattern_at(*p.a, pp);
            double rd = std::hypot(pp.x, pp.z);
            double fraction = rd - std::floor(rd);
            return sub_pattern_at(*p.a, pp) + distance * fraction;
        }
        case PatKind::Blend:
            return (sub_pattern_at(*p.a, pp) + sub_pattern_at(*p.b, pp)) / 2.0;
        case PatKind::Perturbed:
            throw std::runtime_error("perturbed patterns are outside the oracle (libnoise is not available)");
    }
    return {0, 0, 0};
}
inline V3 colour_mul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }

struct Comps {  // intersection/computations.rs:12-24
    const Object* object;
    double t;
    V3 point, over_point, under_point, eye, normal;
    bool inside;
    V3 reflect;
    double n1, n2;
};

// Computations::schlick (computations.rs:28-45); powi(5) = x*((x*x)*(x*x)).
double schlick(const Comps& c) {
    double cosv = dot(c.eye, c.normal);
    if (c.n1 > c.n2) {
        double n_ratio = c.n1 / c.n2;
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cosv * cosv);
        if (sin2_t > 1.0) return 1.0;
        cosv = std::sqrt(1.0 - sin2_t);
    }
    double r = (c.n1 - c.n2) / (c.n1 + c.n2);
    double r0 = r * r;
    double x = 1.0 - cosv;
    double x2 = x * x;
    return r0 + (1.0 - r0) * (x * (x2 * x2));
}

struct World {
    const Scene* scene;
    mutable std::atomic<uint64_t> n_secondary{0}, n_shadow{0}, n_shade{0};

    // World::intersect (world.rs:90-106)
    bool intersect(const Ray& ray, List& list) const {
        list.clear();
        for (const Object& o : scene->objects) object_intersect(o, ray, list);
        if (list.empty()) return false;
        sort_list(list);
        return true;
    }
    // List::hit (list.rs:24-32): first minimum among t > 0
    static const Isect* hit(const List& l) {
        const Isect* best = nullptr;
        for (const Isect& i : l)
            if (i.t > 0.0 && (!best || i.t < best->t)) best = &i;
        return best;
    }
    // Intersection::prepare_computations (intersection/mod.rs:45-109)
    static Comps prepare(const Isect& self, const Ray& ray, const List& xs) {
        Comps c;
        c.object = self.object;
        c.t = self.t;
        c.point = position(ray, self.t);
        c.eye = -ray.direction;
        c.normal = normal_at(*self.object, c.point, self);
        if (dot(c.normal, c.eye) < 0.0) { c.normal = c.normal * -1.0; c.inside = true; }
        else c.inside = false;
        std::vector<const Object*> container;
        c.n1 = NAN; c.n2 = NAN;
        for (const Isect& i : xs) {
            bool is_hit = shape_approx_eq(*i.object, *self.object) && approx_eq(i.t, self.t);
            if (is_hit) c.n1 = container.empty() ? 1.0 : container.back()->material.refractive_index;
            size_t k = 0;
            for (; k < container.size(); k++)
                if (shape_approx_eq(*container[k], *i.object)) break;
            if (k < container.size()) container.erase(container.begin() + k);
            else container.push_back(i.object);
            if (is_hit) {
                c.n2 = container.empty() ? 1.0 : container.back()->material.refractive_index;
                break;
            }
        }
        V3 off = (c.normal * 100000.0) * EPS;
        c.over_point = c.point + off;
        c.under_point = c.point - off;
        c.reflect = reflect(ray.direction, c.normal);
        return c;
    }
    // World::is_shadowed (world.rs:109-126): closest hit only
    bool is_shadowed(V3 light_position, V3 point, List& scratch) const {
        n_shadow.fetch_add(1, std::memory_order_relaxed);
        V3 vec = light_position - point;
        double distance = magnitude(vec);
        V3 direction = normalise(vec);
        Ray ray{point, direction};
        if (intersect(ray, scratch)) {
            const Isect* h = hit(scratch);
            if (h && h->object->casts_shadow && h->t < distance) return true;
        }
        return false;
    }
    // Lightable::positions (point.rs:19-21, area.rs:57-67)
    static void positions(const Light& l, Xoshiro256pp& rng, std::vector<V3>& out) {
        out.clear();
        if (!l.is_area) { out.push_back(l.position); return; }
        for (uint32_t v = 0; v < l.v_steps; v++)
            for (uint32_t u = 0; u < l.u_steps; u++) {
                double r1 = rng.gen_range_inclusive(0.0, 1.0);
                V3 p = l.corner + l.u * ((double)u + r1);
                double r2 = rng.gen_range_inclusive(0.0, 1.0);
                out.push_back(p + l.v * ((double)v + r2));
            }
    }
    // Lightable::intensity_at (point.rs:27-38, area.rs:75-90)
    double intensity_at(const Light& l, V3 point, Xoshiro256pp& rng, List& scratch) const {
        if (!l.is_area) return is_shadowed(l.position, point, scratch) ? 0.0 : 1.0;
        std::vector<V3> pos;
        positions(l, rng, pos);
        double intensity = 0.0;
        for (const V3& p : pos)
            if (!is_shadowed(p, point, scratch)) intensity += 1.0;
        return intensity / (double)l.samples;
    }
    // Material::lighting (material.rs:47-87)
    static V3 lighting(const Material& m, const Object& object, const Light& light, V3 point, V3 eye, V3 normal,
                       double intensity, Xoshiro256pp& rng) {
        V3 object_point = mul_point(object.inverse, point);
        V3 colour = colour_mul(sub_pattern_at(*m.pattern, object_point), light.intensity);
        V3 ambient = colour * m.ambient;
        V3 diffuse{0, 0, 0}, specular{0, 0, 0};
        std::vector<V3> pos;
        positions(light, rng, pos);
        double samples = (double)pos.size();
        for (const V3& lp : pos) {
            V3 lv = normalise(lp - point);
            double ldn = dot(lv, normal);
            if (ldn >= 0.0) {
                diffuse = diffuse + (colour * m.diffuse) * ldn;
                V3 rv = -reflect(lv, normal);
                double rde = dot(rv, eye);
                if (rde > 0.0) {
                    double factor = std::pow(rde, m.shininess);
                    specular = specular + (light.intensity * m.specular) * factor;
                }
            }
        }
        return ambient + ((diffuse + specular) / samples) * intensity;
    }
    // World::shade_hit (world.rs:52-87)
    V3 shade_hit(const Comps& c, uint32_t depth, Xoshiro256pp& rng) const {
        n_shade.fetch_add(1, std::memory_order_relaxed);
        const Material& m = c.object->material;
        V3 surface{0, 0, 0};
        List scratch;
        for (const Light& l : scene->lights) {
            double intensity = intensity_at(l, c.over_point, rng, scratch);
            surface = surface + lighting(m, *c.object, l, c.over_point, c.eye, c.normal, intensity, rng);
        }
        V3 reflected = reflected_colour(c, depth, rng);
        V3 refracted = refracted_colour(c, depth, rng);
        if (m.reflective > 0.0 && m.transparency > 0.0) {
            double reflectance = schlick(c);
            return surface + reflected * reflectance + refracted * (1.0 - reflectance);
        }
        return surface + reflected + refracted;
    }
    // World::reflected_colour (world.rs:129-145)
    V3 reflected_colour(const Comps& c, uint32_t depth, Xoshiro256pp& rng) const {
        if (depth == 0 || c.object->material.reflective <= 0.0) return {0, 0, 0};
        n_secondary.fetch_add(1, std::memory_order_relaxed);
        Ray r{c.over_point, c.reflect};
        return colour_at(r, depth - 1, rng) * c.object->material.reflective;
    }
    // World::refracted_colour (world.rs:148-177)
    V3 refracted_colour(const Comps& c, uint32_t depth, Xoshiro256pp& rng) const {
        const Material& m = c.object->material;
        if (depth == 0 || approx_eq(m.transparency, 0.0)) return {0, 0, 0};
        double n_ratio = c.n1 / c.n2;
        double cos_i = dot(c.eye, c.normal);
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (sin2_t > 1.0) return {0, 0, 0};
        double cos_t = std::sqrt(1.0 - sin2_t);
        V3 direction = c.normal * (n_ratio * cos_i - cos_t) - c.eye * n_ratio;
        n_secondary.fetch_add(1, std::memory_order_relaxed);
        Ray r{c.under_point, direction};
        return colour_at(r, depth - 1, rng) * m.transparency;
    }
    // World::colour_at (world.rs:33-49)
    V3 colour_at(const Ray& ray, uint32_t depth, Xoshiro256pp& rng) const {
        List xs;
        if (intersect(ray, xs)) {
            const Isect* h = hit(xs);
            if (h) {
                Comps c = prepare(*h, ray, xs);
                return shade_hit(c, depth, rng);
            }
        }
        return {0, 0, 0};
    }
};

// Camera::ray_for_pixel (camera.rs:194-209)
Ray ray_for_pixel(const Camera& cam, uint32_t x, uint32_t y) {
    double x_offset = ((double)x + 0.5) * cam.pixel_size;
    double y_offset = ((double)y + 0.5) * cam.pixel_size;
    double world_x = cam.half_width - x_offset;
    double world_y = cam.half_height - y_offset;
    V3 pixel = mul_point(cam.inverse, {world_x, world_y, -1.0});
    V3 origin = mul_point(cam.inverse, {0, 0, 0});
    return {origin, normalise(pixel - origin)};
}

struct OrcScene {
    Scene scene;
    Xoshiro256pp rng;  // the main rng, after scene construction
    World world;
    std::string error;
};

thread_local std::string g_error;

}  // namespace

extern "C" {

const char* orc_last_error() { return g_error.c_str(); }

// main.rs:27-45: seed the rng, build the scene with it.
void* orc_scene_from_file(const char* path, double scale, uint64_t seed, const char* base_dir) {
    try {
        auto s = std::make_unique<OrcScene>();
        s->rng = Xoshiro256pp::seed_from_u64(seed);
        s->scene = scene_from_file(path, scale, s->rng, base_dir ? base_dir : "");
        s->world.scene = &s->scene;
        return s.release();
    } catch (const std::exception& e) { g_error = e.what(); return nullptr; }
}
void* orc_scene_from_text(const char* text, double scale, uint64_t seed, const char* base_dir) {
    try {
        auto s = std::make_unique<OrcScene>();
        s->rng = Xoshiro256pp::seed_from_u64(seed);
        s->scene = scene_from_yaml_text(text, scale, s->rng, base_dir ? base_dir : "");
        s->world.scene = &s->scene;
        return s.release();
    } catch (const std::exception& e) { g_error = e.what(); return nullptr; }
}
void* orc_scene_spheres(double scale, uint64_t seed) {
    try {
        auto s = std::make_unique<OrcScene>();
        s->rng = Xoshiro256pp::seed_from_u64(seed);
        s->scene = generate_random_spheres(scale, s->rng);
        s->world.scene = &s->scene;
        return s.release();
    } catch (const std::exception& e) { g_error = e.what(); return nullptr; }
}
void orc_scene_free(void* h) { delete (OrcScene*)h; }

void orc_scene_size(void* h, uint32_t* w, uint32_t* hgt) {
    auto* s = (OrcScene*)h;
    *w = s->scene.camera.hsize; *hgt = s->scene.camera.vsize;
}
// Test knob: see Camera::half_view_ulps.
void orc_camera_nudge_half_view(void* h, int ulps) {
    auto* s = (OrcScene*)h;
    s->scene.camera.half_view_ulps = ulps;
    s->scene.camera.calculate();
}
// camera.rs:159-161: one u64 per row from the main rng
void orc_row_seeds(void* h, uint64_t* out) {
    auto* s = (OrcScene*)h;
    for (uint32_t y = 0; y < s->scene.camera.vsize; y++) out[y] = s->rng.next_u64();
}
void orc_counters(void* h, uint64_t* out3) {
    auto* s = (OrcScene*)h;
    out3[0] = s->world.n_secondary.load(); out3[1] = s->world.n_shadow.load(); out3[2] = s->world.n_shade.load();
}

// camera.rs:143-179 for rows [y0, y1): per-row rng, x serial; rows dynamic over threads.
int orc_render_rows(void* h, uint32_t depth, const uint64_t* seeds, uint32_t y0, uint32_t y1, int threads,
                    double* out_rgb) {
    auto* s = (OrcScene*)h;
    const Camera& cam = s->scene.camera;
    std::atomic<uint32_t> next{y0};
    std::atomic<int> failed{0};
    auto worker = [&]() {
        for (;;) {
            uint32_t y = next.fetch_add(1);
            if (y >= y1) break;
            try {
                Xoshiro256pp rng = Xoshiro256pp::seed_from_u64(seeds ? seeds[y] : 0);
                for (uint32_t x = 0; x < cam.hsize; x++) {
                    V3 c = s->world.colour_at(ray_for_pixel(cam, x, y), depth, rng);
                    double* px = out_rgb + ((size_t)(y - y0) * cam.hsize + x) * 3;
                    px[0] = c.x; px[1] = c.y; px[2] = c.z;
                }
            } catch (const NanT&) { failed = 1; }
            catch (const std::exception&) { failed = 2; }
        }
    };
    if (threads <= 1) worker();
    else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
    if (failed == 1) { g_error = "NaN intersection distance while sorting"; return -5; }
    if (failed) { g_error = "render failed"; return -1; }
    return 0;
}

// The same for an explicit list of rows (each row is complete, so the per-row
// rng stream is exactly the reference's).  out_rgb holds the rows in list order.
int orc_render_row_list(void* h, uint32_t depth, const uint64_t* seeds, uint32_t n_rows, const uint32_t* rows,
                        int threads, double* out_rgb) {
    auto* s = (OrcScene*)h;
    const Camera& cam = s->scene.camera;
    std::atomic<uint32_t> next{0};
    std::atomic<int> failed{0};
    auto worker = [&]() {
        for (;;) {
            uint32_t k = next.fetch_add(1);
            if (k >= n_rows) break;
            uint32_t y = rows[k];
            try {
                Xoshiro256pp rng = Xoshiro256pp::seed_from_u64(seeds ? seeds[y] : 0);
                for (uint32_t x = 0; x < cam.hsize; x++) {
                    V3 c = s->world.colour_at(ray_for_pixel(cam, x, y), depth, rng);
                    double* px = out_rgb + ((size_t)k * cam.hsize + x) * 3;
                    px[0] = c.x; px[1] = c.y; px[2] = c.z;
                }
            } catch (const NanT&) { failed = 1; }
            catch (const std::exception&) { failed = 2; }
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < std::max(1, threads); i++) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    if (failed == 1) { g_error = "NaN intersection distance while sorting"; return -5; }
    if (failed) { g_error = "render failed"; return -1; }
    return 0;
}

// Selected pixels (x, y) of scenes without area lights (no rng use).
int orc_render_pixels(void* h, uint32_t depth, uint32_t n, const uint32_t* xy, int threads, double* out_rgb) {
    auto* s = (OrcScene*)h;
    const Camera& cam = s->scene.camera;
    std::atomic<uint32_t> next{0};
    std::atomic<int> failed{0};
    auto worker = [&]() {
        Xoshiro256pp rng = Xoshiro256pp::seed_from_u64(0);
        for (;;) {
            uint32_t b = next.fetch_add(256);
            if (b >= n) break;
            uint32_t e = std::min(n, b + 256);
            try {
                for (uint32_t i = b; i < e; i++) {
                    V3 c = s->world.colour_at(ray_for_pixel(cam, xy[2 * i], xy[2 * i + 1]), depth, rng);
                    out_rgb[3 * (size_t)i] = c.x; out_rgb[3 * (size_t)i + 1] = c.y; out_rgb[3 * (size_t)i + 2] = c.z;
                }
            } catch (...) { failed = 1; }
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < std::max(1, threads); i++) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    if (failed) { g_error = "render failed"; return -1; }
    return 0;
}

int orc_colour_at(void* h, uint32_t n, const double* rays, uint32_t depth, const uint64_t* seeds, double* out_rgb) {
    auto* s = (OrcScene*)h;
    try {
        for (uint32_t i = 0; i < n; i++) {
            Xoshiro256pp rng = Xoshiro256pp::seed_from_u64(seeds ? seeds[i] : 0);
            const double* r = rays + 6 * (size_t)i;
            V3 c = s->world.colour_at({{r[0], r[1], r[2]}, {r[3], r[4], r[5]}}, depth, rng);
            out_rgb[3 * (size_t)i] = c.x; out_rgb[3 * (size_t)i + 1] = c.y; out_rgb[3 * (size_t)i + 2] = c.z;
        }
    } catch (const NanT&) { g_error = "NaN intersection distance while sorting"; return -5; }
    catch (const std::exception& e) { g_error = e.what(); return -1; }
    return 0;
}

void orc_ray_for_pixel(void* h, uint32_t x, uint32_t y, double* out6) {
    auto* s = (OrcScene*)h;
    Ray r = ray_for_pixel(s->scene.camera, x, y);
    out6[0] = r.origin.x; out6[1] = r.origin.y; out6[2] = r.origin.z;
    out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
}

// Same 32-double record per ray as rt_probe (include/rt_api.h).  `leaf_of`
// maps a leaf object to its index in insertion order.
static void index_leaves(const Object& o, std::vector<const Object*>& out) {
    if (o.type == ObjType::Shape) { out.push_back(&o); return; }
    for (const Object& c : o.children) index_leaves(c, out);
}
int orc_probe(void* h, uint32_t n, const double* rays, double* out) {
    auto* s = (OrcScene*)h;
    std::vector<const Object*> leaves;
    for (const Object& o : s->scene.objects) index_leaves(o, leaves);
    try {
        List xs, scratch;
        for (uint32_t i = 0; i < n; i++) {
            const double* r = rays + 6 * (size_t)i;
            double* o = out + 32 * (size_t)i;
            for (int k = 0; k < 32; k++) o[k] = 0.0;
            Ray ray{{r[0], r[1], r[2]}, {r[3], r[4], r[5]}};
            o[0] = -1.0;
            bool any = s->world.intersect(ray, xs);
            o[24] = any ? (double)xs.size() : 0.0;
            const Isect* hp = any ? World::hit(xs) : nullptr;
            if (!hp) continue;
            Comps c = World::prepare(*hp, ray, xs);
            o[0] = (double)(std::find(leaves.begin(), leaves.end(), c.object) - leaves.begin());
            o[1] = c.t;
            o[2] = c.point.x; o[3] = c.point.y; o[4] = c.point.z;
            o[5] = c.over_point.x; o[6] = c.over_point.y; o[7] = c.over_point.z;
            o[8] = c.under_point.x; o[9] = c.under_point.y; o[10] = c.under_point.z;
            o[11] = c.eye.x; o[12] = c.eye.y; o[13] = c.eye.z;
            o[14] = c.normal.x; o[15] = c.normal.y; o[16] = c.normal.z;
            o[17] = c.inside ? 1.0 : 0.0;
            o[18] = c.reflect.x; o[19] = c.reflect.y; o[20] = c.reflect.z;
            o[21] = c.n1; o[22] = c.n2;
            o[23] = schlick(c);
            for (size_t l = 0; l < s->scene.lights.size() && l < 4; l++) {
                const Light& lt = s->scene.lights[l];
                o[25 + l] = s->world.is_shadowed(lt.is_area ? lt.corner : lt.position, c.over_point, scratch) ? 1.0 : 0.0;
            }
        }
    } catch (const NanT&) { g_error = "NaN intersection distance while sorting"; return -5; }
    catch (const std::exception& e) { g_error = e.what(); return -1; }
    return 0;
}

// Material::lighting (material.rs:47-87) called directly, as the reference's unit tests do.  Per
// point 12 doubles: leaf index, light index, point, eye, normal, intensity; the rng of point i is
// seeded with seeds[i].
int orc_lighting(void* h, uint32_t n, const double* in, const uint64_t* seeds, double* out_rgb) {
    auto* s = (OrcScene*)h;
    std::vector<const Object*> leaves;
    for (const Object& o : s->scene.objects) index_leaves(o, leaves);
    try {
        for (uint32_t i = 0; i < n; i++) {
            const double* q = in + 12 * (size_t)i;
            const size_t leaf = (size_t)q[0], light = (size_t)q[1];
            if (leaf >= leaves.size() || light >= s->scene.lights.size()) { g_error = "index out of range"; return -1; }
            Xoshiro256pp rng = Xoshiro256pp::seed_from_u64(seeds ? seeds[i] : 0);
            const Object& o = *leaves[leaf];
            V3 c = World::lighting(o.material, o, s->scene.lights[light], {q[2], q[3], q[4]}, {q[5], q[6], q[7]},
                                   {q[8], q[9], q[10]}, q[11], rng);
            out_rgb[3 * (size_t)i] = c.x; out_rgb[3 * (size_t)i + 1] = c.y; out_rgb[3 * (size_t)i + 2] = c.z;
        }
    } catch (const std::exception& e) { g_error = e.what(); return -1; }
    return 0;
}

// Lightable::intensity_at (point.rs:27-38, area.rs:75-90) for n points, one call after the other on
// one generator seeded with `seed` (the reference's test: light/area.rs:164-198).
int orc_intensity_at(void* h, uint32_t light, uint32_t n, const double* points, uint64_t seed, double* out) {
    auto* s = (OrcScene*)h;
    if (light >= s->scene.lights.size()) { g_error = "index out of range"; return -1; }
    try {
        Xoshiro256pp rng = Xoshiro256pp::seed_from_u64(seed);
        List scratch;
        for (uint32_t i = 0; i < n; i++)
            out[i] = s->world.intensity_at(s->scene.lights[light], {points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]},
                                           rng, scratch);
    } catch (const NanT&) { g_error = "NaN intersection distance while sorting"; return -5; }
    catch (const std::exception& e) { g_error = e.what(); return -1; }
    return 0;
}

// Matrix::invert and determinant (matrix.rs:99-123,227-245) of a row-major 4x4; returns -1 for a
// matrix the reference refuses to invert.
int orc_matrix_inverse(const double* in16, double* out16, double* det) {
    Mat4 m;
    std::memcpy(m.m, in16, sizeof m.m);
    *det = det4(m);
    try {
        Mat4 r = invert(m);
        std::memcpy(out16, r.m, sizeof r.m);
    } catch (const std::exception& e) { g_error = e.what(); return -1; }
    return 0;
}

// Colour::to_u8 (colour.rs:63-72): clamp, *255, round half away from zero; NaN -> 0.
void orc_to_u8(const double* rgb, size_t n_values, uint8_t* out) {
    for (size_t i = 0; i < n_values; i++) {
        double c = rgb[i];
        if (std::isnan(c)) { out[i] = 0; continue; }
        double cl = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
        out[i] = (uint8_t)std::round(cl * 255.0);
    }
}

// RNG known answers (light/area.rs:136-160): expose the generator.
void orc_rng_gen_range(uint64_t seed, uint32_t n, double lo, double hi, double* out) {
    Xoshiro256pp r = Xoshiro256pp::seed_from_u64(seed);
    for (uint32_t i = 0; i < n; i++) out[i] = r.gen_range_inclusive(lo, hi);
}
void orc_rng_u64(uint64_t seed, uint32_t n, uint64_t* out) {
    Xoshiro256pp r = Xoshiro256pp::seed_from_u64(seed);
    for (uint32_t i = 0; i < n; i++) out[i] = r.next_u64();
}

}  // extern "C"
