// Scene construction on the host: the YAML scene format (add / define /
// extend, named transformations and materials, csg, groups, obj files), the
// Wavefront OBJ reader and the generated many-spheres scene.
//
// reference: src/scene/mod.rs:66-85,114-293, src/scene/list.rs:20-37,
// src/scene/add.rs:18-34, src/scene/define.rs:18-59, src/scene/shapes.rs:99-231,
// src/scene/material.rs:17-75, src/scene/transformations.rs:12-47,
// src/object/obj_parser.rs:37-232, src/math/angle.rs:75-106,
// src/math/transformation/mod.rs:96-197, src/camera.rs:221-252,
// src/light/mod.rs:58-95, src/material.rs:107-157, src/pattern/mod.rs:121-208.
#pragma once

#include <algorithm>
#include <cctype>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "scene.h"
#include "yaml.h"

namespace rt {

// ------------------------------------------------------------------ angles
// Expression strings such as "PI / 3" or "2 * PI / 3" are evaluated by the
// exmex 0.19.0 crate in the reference (third-party, not vendored). Restated
// here with exmex's default float operator priorities (+ 0, - 1, * 2, / 3,
// ^ 4, unary operators bind tightest), so "2 * PI / 3" is 2 * (PI / 3).
// The reference's tests pin only single-operator expressions
// (src/math/angle.rs:225-250): the association of mixed * and / is unpinned.
namespace expr_detail {
struct Eval {
    const std::string& s;
    size_t i = 0;
    explicit Eval(const std::string& str) : s(str) {}
    void ws() { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; }
    static int prio(char op) {
        switch (op) { case '+': return 0; case '-': return 1; case '*': return 2; case '/': return 3; case '^': return 4; }
        return -1;
    }
    static double apply(char op, double a, double b) {
        switch (op) {
            case '+': return a + b; case '-': return a - b; case '*': return a * b;
            case '/': return a / b; case '^': return std::pow(a, b);
        }
        return 0;
    }
    double operand() {
        ws();
        if (i >= s.size()) throw std::runtime_error("unexpected end of expression '" + s + "'");
        if (s[i] == '-') { i++; return -operand(); }
        if (s[i] == '+') { i++; return operand(); }
        if (s[i] == '(') {
            i++;
            double v = expression();
            ws();
            if (i >= s.size() || s[i] != ')') throw std::runtime_error("missing ')' in '" + s + "'");
            i++;
            return v;
        }
        if (std::isalpha((unsigned char)s[i])) {
            size_t b = i;
            while (i < s.size() && std::isalnum((unsigned char)s[i])) i++;
            std::string name = s.substr(b, i - b);
            if (name == "PI") return 3.14159265358979323846264338327950288;
            if (name == "E") return 2.71828182845904523536028747135266250;
            if (name == "TAU") return 6.28318530717958647692528676655900577;
            throw std::runtime_error("unknown name '" + name + "' in '" + s + "'");
        }
        const char* start = s.c_str() + i;
        char* end = nullptr;
        double v = std::strtod(start, &end);
        if (end == start) throw std::runtime_error("cannot parse expression '" + s + "'");
        i += (size_t)(end - start);
        return v;
    }
    double expression() {
        std::vector<double> vals;
        std::vector<char> ops;
        vals.push_back(operand());
        for (;;) {
            ws();
            if (i >= s.size() || prio(s[i]) < 0) break;
            char op = s[i++];
            while (!ops.empty() && prio(ops.back()) >= prio(op)) {
                double b = vals.back(); vals.pop_back();
                double a = vals.back(); vals.pop_back();
                vals.push_back(apply(ops.back(), a, b));
                ops.pop_back();
            }
            ops.push_back(op);
            vals.push_back(operand());
        }
        while (!ops.empty()) {
            double b = vals.back(); vals.pop_back();
            double a = vals.back(); vals.pop_back();
            vals.push_back(apply(ops.back(), a, b));
            ops.pop_back();
        }
        return vals[0];
    }
};
}  // namespace expr_detail

inline double eval_expression(const std::string& s) {
    expr_detail::Eval e(s);
    double v = e.expression();
    e.ws();
    if (e.i != s.size()) throw std::runtime_error("trailing characters in expression '" + s + "'");
    return v;
}

// Angle::deserialize (angle.rs:75-106): number = radians, string =
// expression, {degrees: x} = x.to_radians() = x * (PI / 180.0).
inline double parse_angle(const YVal& v) {
    auto parse = [](const YVal& x) -> double {
        if (x.is_number()) return x.as_f64();
        if (x.is_scalar()) return eval_expression(x.s);
        throw std::runtime_error("Unable to parse angle value as a float");
    };
    if (v.is_map()) {
        const YVal* d = v.get("degrees");
        if (!d) throw std::runtime_error("angle mapping needs a 'degrees' key");
        return parse(*d) * (3.14159265358979323846264338327950288 / 180.0);
    }
    return parse(v);
}

inline V3 parse_tuple3(const YVal& v, const char* what) {
    if (!v.is_seq() || v.seq.size() != 3) throw std::runtime_error(std::string("expected 3 numbers for ") + what);
    return {v.seq[0].as_f64(), v.seq[1].as_f64(), v.seq[2].as_f64()};
}

// Transformation::deserialize (transformation/mod.rs:96-197): every listed
// operation left-multiplies the running matrix.
inline Mat4 parse_transformation(const YVal& list) {
    if (!list.is_seq()) throw std::runtime_error("transform must be a list");
    Mat4 t = Mat4::identity();
    for (const YVal& item : list.seq) {
        if (!item.is_seq() || item.seq.empty() || !item.seq[0].is_scalar())
            throw std::runtime_error("Unable to parse transformation operator");
        const std::string& op = item.seq[0].s;
        size_t n = item.seq.size() - 1;
        auto need = [&](size_t len) {
            if (n != len)
                throw std::runtime_error("Transformation '" + op + "' requires " + std::to_string(len) +
                                         " arguments, found " + std::to_string(n));
        };
        auto f = [&](size_t k) { return item.seq[k + 1].as_f64(); };
        if (op == "rotate-x") { need(1); t = Mat4::rotate_x(parse_angle(item.seq[1])) * t; }
        else if (op == "rotate-y") { need(1); t = Mat4::rotate_y(parse_angle(item.seq[1])) * t; }
        else if (op == "rotate-z") { need(1); t = Mat4::rotate_z(parse_angle(item.seq[1])) * t; }
        else if (op == "scale") { need(3); t = Mat4::scale(f(0), f(1), f(2)) * t; }
        else if (op == "shear") { need(6); t = Mat4::shear(f(0), f(1), f(2), f(3), f(4), f(5)) * t; }
        else if (op == "translate") { need(3); t = Mat4::translate(f(0), f(1), f(2)) * t; }
        else throw std::runtime_error("Unknown operator '" + op + "'");
    }
    return t;
}

// ------------------------------------------------------------------ OBJ
// obj_parser.rs:37-232: `v`, `vn`, `f` (fan triangulation; a, a/b/c, a//c),
// `g`; groups are emitted sorted by name, empty ones dropped.
struct ObjData {
    std::vector<V3> vertices, normals;
    std::vector<Object> groups;  // one built Group per non-empty named group
    uint32_t ignored = 0;
};

inline ObjData parse_obj(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open OBJ file '" + path + "'");
    ObjData d;
    std::map<std::string, std::vector<Object>> groups;  // std::map iterates sorted by name
    groups["default"];
    std::vector<Object>* current = &groups["default"];
    auto split = [](const std::string& line) {
        std::vector<std::string> out;
        size_t i = 0;
        while (i < line.size()) {
            while (i < line.size() && line[i] == ' ') i++;
            size_t b = i;
            while (i < line.size() && line[i] != ' ') i++;
            if (i > b) out.push_back(line.substr(b, i - b));
        }
        return out;
    };
    auto num = [](const std::string& s) {
        char* end = nullptr;
        double v = std::strtod(s.c_str(), &end);
        if (s.empty() || !end || *end != '\0') throw std::runtime_error("invalid float literal '" + s + "'");
        return v;
    };
    auto idx = [](const std::string& s) -> size_t {
        char* end = nullptr;
        if (s.empty() || !std::isdigit((unsigned char)s[0])) throw std::runtime_error("invalid digit found in string '" + s + "'");
        unsigned long v = std::strtoul(s.c_str(), &end, 10);
        if (!end || *end != '\0') throw std::runtime_error("invalid digit found in string '" + s + "'");
        if (v == 0) throw std::runtime_error("attempt to subtract with overflow (index 0)");
        return (size_t)(v - 1);
    };
    std::string raw;
    while (std::getline(f, raw)) {
        std::string line = yaml_detail::trim(raw);
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.rfind("v ", 0) == 0) {
            auto it = split(line);
            if (it.size() != 4)
                throw std::runtime_error("Expected 'v' followed by 3 space separated numbers for a vertex.\nFound " +
                                         std::to_string(it.size()) + " items.");
            d.vertices.push_back({num(it[1]), num(it[2]), num(it[3])});
        } else if (line.rfind("vn ", 0) == 0) {
            auto it = split(line);
            if (it.size() != 4)
                throw std::runtime_error("Expected 'vn' followed by 3 space separated numbers for a normal.\nFound " +
                                         std::to_string(it.size()) + " items.");
            d.normals.push_back({num(it[1]), num(it[2]), num(it[3])});
        } else if (line.rfind("f ", 0) == 0) {
            auto it = split(line);
            if (it.size() < 4)
                throw std::runtime_error("Expected 'f' followed by at least 3 space separated numbers for a face.\nFound " +
                                         std::to_string(it.size()) + " items.");
            auto vertex_normal = [&](const std::string& item, size_t& v, bool& has_n, size_t& n) {
                std::vector<std::string> parts;
                size_t b = 0;
                for (;;) {
                    size_t s = item.find('/', b);
                    parts.push_back(item.substr(b, s == std::string::npos ? std::string::npos : s - b));
                    if (s == std::string::npos) break;
                    b = s + 1;
                }
                if (parts.size() != 1 && parts.size() != 3)
                    throw std::runtime_error("Expected face values to be either 'num' or 'num//num' or 'num/num/num'\nFound " + item + ".");
                v = idx(parts[0]);
                has_n = parts.size() == 3;
                n = has_n ? idx(parts[2]) : 0;
            };
            auto vert = [&](size_t i) -> V3 {
                if (i >= d.vertices.size()) throw std::runtime_error("face references a vertex that does not exist");
                return d.vertices[i];
            };
            auto norm = [&](size_t i) -> V3 {
                if (i >= d.normals.size()) throw std::runtime_error("face references a normal that does not exist");
                return d.normals[i];
            };
            size_t v1, n1; bool h1;
            vertex_normal(it[1], v1, h1, n1);
            for (size_t k = 2; k + 1 < it.size(); k++) {
                size_t v2, n2, v3, n3; bool h2, h3;
                vertex_normal(it[k], v2, h2, n2);
                vertex_normal(it[k + 1], v3, h3, n3);
                if (h1) {
                    if (!h2 || !h3)
                        throw std::runtime_error("If one vertex normal is specified, all faces must also provide vertex normals.");
                    current->push_back(make_triangle(vert(v1), vert(v2), vert(v3), norm(n1), norm(n2), norm(n3)));
                } else {
                    current->push_back(make_flat_triangle(vert(v1), vert(v2), vert(v3)));
                }
            }
        } else if (line.rfind("g ", 0) == 0) {
            std::string name = yaml_detail::trim(line.substr(1));
            if (groups.count(name)) throw std::runtime_error("Group " + name + " is repeated.");
            current = &groups[name];
        } else {
            d.ignored++;
        }
    }
    for (auto& kv : groups)
        if (!kv.second.empty()) d.groups.push_back(build_group(std::move(kv.second)));
    return d;
}

// ------------------------------------------------------------------ YAML scene
struct SceneLoader {
    struct MaterialRef { bool is_name = false; std::string name; YVal data; };

    Xoshiro256pp& rng;
    std::string base_dir;
    std::map<std::string, YVal> shapes;           // define name -> full add-mapping (with 'add')
    std::map<std::string, MaterialRef> materials;
    std::map<std::string, YVal> transformations;  // define name -> raw list
    bool have_camera = false;
    bool has_perturbed = false;
    Scene scene;

    SceneLoader(Xoshiro256pp& r, std::string base) : rng(r), base_dir(std::move(base)) {}

    // TransformationList::collect (transformations.rs:18-36)
    YVal collect(const YVal& list) {
        if (!list.is_seq()) throw std::runtime_error("transform must be a list");
        YVal out;
        out.kind = YVal::Seq;
        for (const YVal& item : list.seq) {
            if (item.is_scalar()) {
                auto it = transformations.find(item.s);
                if (it == transformations.end())
                    throw std::runtime_error("Reference to transformations '" + item.s + "' that was not defined");
                YVal sub = collect(it->second);
                for (auto& s : sub.seq) out.seq.push_back(s);
            } else {
                out.seq.push_back(item);
            }
        }
        return out;
    }

    // Pattern::deserialize (pattern/mod.rs:121-208)
    PatternPtr parse_colour_or_pattern(const YVal& v) {
        if (v.is_seq()) return solid_pattern(parse_tuple3(v, "a colour"));
        return parse_pattern(v);
    }
    PatternPtr parse_pattern(const YVal& v) {
        if (!v.is_map()) throw std::runtime_error("pattern must be a mapping");
        Mat4 t = Mat4::identity();
        if (const YVal* tr = v.get("transform")) t = parse_transformation(*tr);
        const YVal *kind = v.get("kind"), *a = v.get("a"), *b = v.get("b");
        if (kind && kind->is_scalar() && a && b) {
            static const std::pair<const char*, PatKind> kinds[] = {
                {"blend", PatKind::Blend}, {"checker", PatKind::Checker}, {"gradient", PatKind::Gradient},
                {"radial-gradient", PatKind::RadialGradient}, {"ring", PatKind::Ring}, {"stripe", PatKind::Stripe}};
            for (auto& k : kinds)
                if (kind->s == k.first)
                    return make_pattern(k.second, parse_colour_or_pattern(*a), parse_colour_or_pattern(*b), t);
            throw std::runtime_error("Unknown pattern '" + kind->s + "'");
        }
        const YVal *scale = v.get("scale"), *sub = v.get("pattern"), *seed = v.get("seed");
        if (scale && sub && seed) {
            has_perturbed = true;
            auto p = std::make_shared<Pattern>();
            p->kind = PatKind::Perturbed;
            p->scale = scale->as_f64();
            p->a = parse_pattern(*sub);
            p->transformation = t;
            p->inverse = invert(t);
            return p;
        }
        throw std::runtime_error("data did not match any variant of untagged enum PatternData");
    }

    YVal material_value(const MaterialRef& m) {  // Material::get_value
        if (!m.is_name) return m.data;
        auto it = materials.find(m.name);
        if (it == materials.end())
            throw std::runtime_error("Reference to material '" + m.name + "' that was not defined");
        return material_value(it->second);
    }
    static MaterialRef material_ref(const YVal& v) {
        MaterialRef m;
        if (v.is_scalar()) { m.is_name = true; m.name = v.s; }
        else m.data = v;
        return m;
    }
    // scene::Material::parse (scene/material.rs:17-47) + Material::deserialize
    // (material.rs:107-157). A material with a `pattern` key consumes exactly
    // one u64 from the scene rng.
    Material parse_material(const YVal& v) {
        YVal value = material_value(material_ref(v));
        if (!value.is_map()) throw std::runtime_error("material must be a mapping");
        Material m;
        const YVal* pattern = value.get("pattern");
        const YVal* colour = value.get("color");
        if (pattern) {
            if (!pattern->is_map()) throw std::runtime_error("pattern must be a mapping");
            YVal pm = *pattern;
            pm.set("seed", seed_value(rng.next_u64()));
            if (const YVal* tr = pm.get("transform")) pm.set("transform", collect(*tr));
            if (colour) throw std::runtime_error("Only one of pattern or colour can be set on a material");
            m.pattern = parse_pattern(pm);
        } else if (colour) {
            m.pattern = solid_pattern(parse_tuple3(*colour, "a colour"));
        }
        auto opt = [&](const char* key, double& dst) {
            if (const YVal* x = value.get(key)) dst = x->as_f64();
        };
        opt("ambient", m.ambient); opt("diffuse", m.diffuse); opt("specular", m.specular);
        opt("shininess", m.shininess); opt("reflective", m.reflective);
        opt("transparency", m.transparency); opt("refractive_index", m.refractive_index);
        return m;
    }
    static YVal seed_value(uint64_t s) {
        YVal v;
        v.kind = YVal::Scalar;
        v.s = std::to_string(s);
        return v;
    }

    // build_object! (scene/shapes.rs:66-97): material, then transform, then shadow.
    struct Common { bool has_m = false, has_t = false, has_s = false; Material m; Mat4 t = Mat4::identity(); bool s = true; };
    Common parse_common(const YVal& v) {
        Common c;
        if (const YVal* m = v.get("material")) if (m->kind != YVal::Null) { c.has_m = true; c.m = parse_material(*m); }
        if (const YVal* t = v.get("transform")) if (t->kind != YVal::Null) { c.has_t = true; c.t = parse_transformation(collect(*t)); }
        if (const YVal* s = v.get("shadow")) if (s->kind != YVal::Null) { c.has_s = true; c.s = s->as_bool(); }
        return c;
    }
    Object finish_shape(Object o, const Common& c) {
        if (c.has_m) o.material = c.m;
        if (c.has_t) { o.transformation = c.t; o.inverse = invert(c.t); }
        if (c.has_s) o.casts_shadow = c.s;
        return o;
    }
    Object finish_group(std::vector<Object> objects, const YVal& v) {
        Common c = parse_common(v);
        Object g = build_group(std::move(objects), c.t, c.has_m ? &c.m : nullptr, c.has_s ? &c.s : nullptr);
        if (const YVal* d = v.get("divide")) if (d->kind != YVal::Null) g = divide(std::move(g), (uint32_t)d->as_u64());
        return g;
    }

    std::string resolve(const std::string& file) {
        if (!file.empty() && file[0] == '/') return file;
        return base_dir.empty() ? file : base_dir + "/" + file;
    }

    // parse_shape (scene/shapes.rs:169-231). `value` is the mapping without its tag key.
    Object parse_shape(const std::string& tag, const YVal& value) {
        if (value.kind != YVal::Map && value.kind != YVal::Null) throw std::runtime_error("shape must be a mapping");
        auto optf = [&](const char* k, double dflt) {
            const YVal* x = value.get(k);
            return (x && x->kind != YVal::Null) ? x->as_f64() : dflt;
        };
        auto optb = [&](const char* k, bool dflt) {
            const YVal* x = value.get(k);
            return (x && x->kind != YVal::Null) ? x->as_bool() : dflt;
        };
        if (tag == "cone" || tag == "cylinder") {
            Object o = make_cyl_or_cone(tag == "cone" ? ShapeKind::Cone : ShapeKind::Cylinder, optf("min", -INF),
                                        optf("max", INF), optb("closed", false));
            return finish_shape(std::move(o), parse_common(value));
        }
        if (tag == "cube") return finish_shape(make_shape(ShapeKind::Cube), parse_common(value));
        if (tag == "plane") return finish_shape(make_shape(ShapeKind::Plane), parse_common(value));
        if (tag == "sphere") return finish_shape(make_shape(ShapeKind::Sphere), parse_common(value));
        if (tag == "group") {
            const YVal* children = value.get("children");
            if (!children || !children->is_seq()) throw std::runtime_error("missing field `children`");
            std::vector<Object> objects;
            for (const YVal& c : children->seq) objects.push_back(parse_add_shape(c));
            return finish_group(std::move(objects), value);
        }
        if (tag == "obj") {
            const YVal* file = value.get("file");
            if (!file || !file->is_scalar()) throw std::runtime_error("missing field `file`");
            ObjData d = parse_obj(resolve(file->s));
            return finish_group(std::move(d.groups), value);
        }
        if (tag == "csg") {
            const YVal *op = value.get("operation"), *l = value.get("left"), *r = value.get("right");
            if (!op || !l || !r || !op->is_scalar()) throw std::runtime_error("csg needs operation, left and right");
            CsgOp o;
            if (op->s == "difference") o = CsgOp::Difference;
            else if (op->s == "intersection") o = CsgOp::Intersection;
            else if (op->s == "union") o = CsgOp::Union;
            else throw std::runtime_error("unknown variant `" + op->s + "`");
            auto side = [&](const YVal& s) {
                const YVal* type = s.get("type");
                if (!type || !type->is_scalar()) throw std::runtime_error("missing field `type`");
                YVal rest = s;
                rest.remove("type");
                return parse_shape(type->s, rest);
            };
            Object left = side(*l);
            Object right = side(*r);
            return make_csg(o, std::move(left), std::move(right));
        }
        auto it = shapes.find(tag);
        if (it == shapes.end()) throw std::runtime_error("Reference to shape '" + tag + "' that was not defined");
        YVal define_values = it->second;
        YVal add_tag;
        define_values.remove("add", &add_tag);
        if (const YVal* t = value.get("transform")) {
            YVal transform = *t;
            YVal define_transform;
            if (define_values.remove("transform", &define_transform)) {  // TransformationList::combine
                if (!define_transform.is_seq() || !transform.is_seq()) throw std::runtime_error("transform must be a list");
                for (auto& x : transform.seq) define_transform.seq.push_back(x);
                transform = define_transform;
            }
            define_values.set("transform", transform);
        }
        if (const YVal* m = value.get("material")) define_values.set("material", *m);
        if (const YVal* s = value.get("shadow")) define_values.set("shadow", *s);
        return parse_shape(add_tag.s, define_values);
    }
    Object parse_add_shape(const YVal& v) {
        const YVal* add = v.get("add");
        if (!add || !add->is_scalar()) throw std::runtime_error("missing field `add`");
        YVal rest = v;
        rest.remove("add");
        return parse_shape(add->s, rest);
    }

    void parse_camera(const YVal& v) {
        auto req = [&](const char* k) -> const YVal& {
            const YVal* x = v.get(k);
            if (!x) throw std::runtime_error(std::string("missing field `") + k + "`");
            return *x;
        };
        uint32_t w = (uint32_t)req("width").as_u64(), h = (uint32_t)req("height").as_u64();
        double fov = parse_angle(req("field-of-view"));
        V3 from = parse_tuple3(req("from"), "from"), to = parse_tuple3(req("to"), "to"), up = parse_tuple3(req("up"), "up");
        scene.camera = Camera::make(w, h, fov, view_transformation(from, to, up));
    }
    void parse_light(const YVal& v) {
        const YVal *at = v.get("at"), *intensity = v.get("intensity");
        if (at && intensity) {
            scene.lights.push_back(point_light(parse_tuple3(*at, "at"), parse_tuple3(*intensity, "intensity")));
            return;
        }
        const YVal *corner = v.get("corner"), *u = v.get("uvec"), *us = v.get("usteps"), *vv = v.get("vvec"),
                   *vs = v.get("vsteps");
        if (corner && u && us && vv && vs && intensity) {
            scene.lights.push_back(area_light(parse_tuple3(*corner, "corner"), parse_tuple3(*u, "uvec"),
                                              (uint32_t)us->as_u64(), parse_tuple3(*vv, "vvec"),
                                              (uint32_t)vs->as_u64(), parse_tuple3(*intensity, "intensity")));
            return;
        }
        throw std::runtime_error("data did not match any variant of untagged enum Light");
    }

    // Define::parse (define.rs:18-59)
    void parse_define(const YVal& v) {
        const std::string& name = v.get("define")->s;
        const YVal* value = v.get("value");
        if (!value) throw std::runtime_error("missing field `value`");
        if (value->is_seq()) {
            if (transformations.count(name)) throw std::runtime_error("Transformations '" + name + "' already defined");
            transformations[name] = *value;
        } else if (value->is_map() && value->get("add") && value->get("add")->is_scalar()) {
            if (shapes.count(name)) throw std::runtime_error("Shape '" + name + "' already defined");
            shapes[name] = *value;
        } else if (value->is_map()) {
            MaterialRef m;
            const YVal* extend = v.get("extend");
            if (extend && extend->kind != YVal::Null) {
                auto it = materials.find(extend->s);
                if (it == materials.end())
                    throw std::runtime_error("Attempt to extend material '" + extend->s + "' which was not defined");
                m.data = it->second.data;  // Material::update: later keys override
                for (auto& kv : value->map) m.data.set(kv.first, kv.second);
            } else {
                m.data = *value;
            }
            if (materials.count(name)) throw std::runtime_error("Material '" + name + "' already defined");
            materials[name] = m;
        } else {
            throw std::runtime_error("Unable to parse define '" + name + "'");
        }
    }

    // List::parse (list.rs:20-37) + Add::parse (add.rs:18-34)
    void parse_list(const YVal& root) {
        if (!root.is_seq()) throw std::runtime_error("scene file must be a list of elements");
        for (const YVal& e : root.seq) {
            if (!e.is_map()) throw std::runtime_error("data did not match any variant of untagged enum Element");
            const YVal* add = e.get("add");
            if (add && add->is_scalar()) {
                if (add->s == "camera") {
                    if (have_camera) throw std::runtime_error("Only one camera can be added");
                    parse_camera(e);
                    have_camera = true;
                } else if (add->s == "light") {
                    parse_light(e);
                } else {
                    scene.objects.push_back(parse_add_shape(e));
                }
            } else if (e.get("define") && e.get("define")->is_scalar()) {
                parse_define(e);
            } else {
                throw std::runtime_error("data did not match any variant of untagged enum Element");
            }
        }
        if (!have_camera) throw std::runtime_error("A camera must be defined");
        if (scene.lights.empty()) throw std::runtime_error("No lights were defined");
        if (scene.objects.empty()) throw std::runtime_error("No objects were defined");
    }
};

// Scene::from_file (scene/mod.rs:66-85). `base_dir` is the directory that
// relative `file:` entries of `add: obj` are resolved against (the reference
// resolves them against the process working directory).
inline Scene scene_from_yaml_text(const std::string& text, double scale, Xoshiro256pp& rng,
                                  const std::string& base_dir, bool* has_perturbed = nullptr) {
    SceneLoader l(rng, base_dir);
    l.parse_list(yaml_parse(text));
    l.scene.camera.scale(scale);
    if (has_perturbed) *has_perturbed = l.has_perturbed;
    return std::move(l.scene);
}
inline Scene scene_from_file(const std::string& path, double scale, Xoshiro256pp& rng,
                             const std::string& base_dir, bool* has_perturbed = nullptr) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open scene file '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    return scene_from_yaml_text(ss.str(), scale, rng, base_dir, has_perturbed);
}

// Scene::generate_random_spheres (scene/mod.rs:114-293)
inline Scene generate_random_spheres(double scale, Xoshiro256pp& rng) {
    Scene sc;
    sc.camera = Camera::make((uint32_t)(1000.0 * scale), (uint32_t)(800.0 * scale), 1.04719755119659774615421446109316763,
                             view_transformation({0.0, 2.0, -1.0}, {0.0, 1.8, 0.0}, {0, 1, 0}));
    Material floor;
    floor.pattern = make_pattern(PatKind::Checker, solid_pattern({0.5, 0.5, 0.4}), solid_pattern({0.5, 0.4, 0.3}),
                                 Mat4::identity());
    sc.objects.push_back(make_shape(ShapeKind::Plane, Mat4::identity(), floor));

    auto generate_spheres = [&](size_t num, double mn, double mx) {
        std::vector<Object> spheres;
        std::vector<V3> locations;
        for (size_t k = 0; k < num; k++) {
            auto ok = [&](const V3& p) {
                for (const V3& l : locations) {
                    if (std::fabs(l.x - p.x) < 0.6) return false;
                    if (std::fabs(l.z - p.z) < 0.6) return false;
                }
                return true;
            };
            V3 loc;
            loc.x = rng.gen_range_inclusive(mn, mx); loc.y = 0.5; loc.z = rng.gen_range_inclusive(mn, mx);
            while (!ok(loc)) {
                loc.x = rng.gen_range_inclusive(mn, mx); loc.y = 0.5; loc.z = rng.gen_range_inclusive(mn, mx);
            }
            locations.push_back(loc);
            Material m;
            if (rng.gen_range_inclusive(0.0, 1.0) < 0.1) {
                m = glass_material();
            } else {
                double reflective = (rng.gen_range_inclusive(0.0, 1.0) < 0.4) ? 0.0 : rng.gen_range_inclusive(0.0, 1.0);
                V3 c;
                c.x = rng.gen_range_inclusive(0.0, 1.0);
                c.y = rng.gen_range_inclusive(0.0, 1.0);
                c.z = rng.gen_range_inclusive(0.0, 1.0);
                m.pattern = solid_pattern(c);
                m.ambient = rng.gen_range_inclusive(0.0, 1.0);
                m.diffuse = rng.gen_range_inclusive(0.0, 1.0);
                m.specular = rng.gen_range_inclusive(0.0, 1.0);
                m.shininess = rng.gen_range_inclusive(0.0, 250.0);
                m.reflective = reflective;
            }
            Mat4 t = Mat4::translate(loc.x, loc.y, loc.z) * (Mat4::scale(0.5, 0.5, 0.5) * Mat4::identity());
            spheres.push_back(make_shape(ShapeKind::Sphere, t, m));
        }
        return spheres;
    };
    struct G { size_t n; double mn, mx, tx, tz; };
    const G groups[] = {{20, -10, 10, -10, 35}, {20, -10, 10, 10, 35}, {20, -10, 10, -8, 25}, {20, -10, 10, 8, 25},
                        {10, -5, 5, -5, 10},    {10, -5, 5, 5, 10},    {10, -5, 5, -5, 0},    {10, -5, 5, 5, 0}};
    for (const G& g : groups) {
        auto spheres = generate_spheres(g.n, g.mn, g.mx);
        sc.objects.push_back(build_group(std::move(spheres), Mat4::translate(g.tx, 0.0, g.tz) * Mat4::identity()));
    }
    sc.lights.push_back(point_light({-100, 100, -100}, {0.5, 0.5, 0.5}));
    sc.lights.push_back(point_light({100, 100, 100}, {0.5, 0.5, 0.5}));
    return sc;
}

}  // namespace rt
