// Flattening of the host object tree into the SoA buffers of rt_scene_desc
// (include/rt_api.h): depth-first node list with skip links, leaves in
// insertion order, structural-equality classes, csg left-subtree class sets,
// de-duplicated materials and pattern nodes, lights and the camera block.
//
// Equality classes restate what the reference decides by value at run time:
// `approx_eq!(object, intersection.object)` in the container walk
// (src/intersection/mod.rs:77-79) and `left.includes(object)` in the csg
// filter (src/object/csg/mod.rs:67, src/object/shape.rs:97-108).
#pragma once

#include <algorithm>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "../../../include/rt_api.h"
#include "scene.h"

namespace rt {

struct FlatScene {
    std::vector<uint32_t> node_kind, node_skip, node_arg;
    std::vector<double> node_bbox;
    std::vector<uint32_t> leaf_shape, leaf_tri, leaf_class, leaf_material;
    std::vector<double> leaf_inv, leaf_param;
    std::vector<uint8_t> leaf_casts_shadow;
    std::vector<double> tri_data;
    std::vector<uint32_t> csg_op;
    std::vector<uint64_t> csg_left_bits;
    uint32_t n_classes = 0;
    std::vector<double> material;
    std::vector<uint32_t> material_pattern;
    std::vector<uint32_t> pat_kind, pat_a, pat_b;
    std::vector<double> pat_inv, pat_colour;
    std::vector<uint32_t> light_kind, light_steps;
    std::vector<double> light_data;
    bool has_perturbed = false;

    std::vector<const Object*> leaves;  // host objects in leaf order (not part of the desc)

    rt_scene_desc desc() const {
        rt_scene_desc d{};
        d.n_nodes = (uint32_t)node_kind.size();
        d.node_kind = node_kind.data(); d.node_skip = node_skip.data(); d.node_arg = node_arg.data();
        d.node_bbox = node_bbox.data();
        d.n_leaves = (uint32_t)leaf_shape.size();
        d.leaf_shape = leaf_shape.data(); d.leaf_inv = leaf_inv.data(); d.leaf_param = leaf_param.data();
        d.leaf_tri = leaf_tri.data(); d.leaf_class = leaf_class.data(); d.leaf_material = leaf_material.data();
        d.leaf_casts_shadow = leaf_casts_shadow.data();
        d.n_tris = (uint32_t)(tri_data.size() / 18);
        d.tri_data = tri_data.data();
        d.n_csgs = (uint32_t)csg_op.size();
        d.csg_op = csg_op.data();
        d.n_classes = n_classes;
        d.csg_left_bits = csg_left_bits.data();
        d.n_materials = (uint32_t)material_pattern.size();
        d.material = material.data(); d.material_pattern = material_pattern.data();
        d.n_patterns = (uint32_t)pat_kind.size();
        d.pat_kind = pat_kind.data(); d.pat_a = pat_a.data(); d.pat_b = pat_b.data();
        d.pat_inv = pat_inv.data(); d.pat_colour = pat_colour.data();
        d.n_lights = (uint32_t)light_kind.size();
        d.light_kind = light_kind.data(); d.light_data = light_data.data(); d.light_steps = light_steps.data();
        return d;
    }
};

namespace flatten_detail {

inline void push_rows3(std::vector<double>& dst, const Mat4& m) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) dst.push_back(m.m[r][c]);
}
inline void push_v3(std::vector<double>& dst, V3 v) { dst.push_back(v.x); dst.push_back(v.y); dst.push_back(v.z); }

struct Builder {
    FlatScene& f;
    std::unordered_map<const Pattern*, uint32_t> pattern_ids;   // by identity (shared trees)
    std::vector<const Material*> material_reps;                  // by value, see material_id
    std::unordered_map<const Pattern*, std::vector<uint32_t>> materials_by_pattern;

    explicit Builder(FlatScene& fs) : f(fs) {}

    uint32_t pattern_id(const Pattern* p) {
        auto it = pattern_ids.find(p);
        if (it != pattern_ids.end()) return it->second;
        uint32_t a = RT_NONE, b = RT_NONE;
        if (p->kind == PatKind::Perturbed) f.has_perturbed = true;
        if (p->a) a = pattern_id(p->a.get());
        if (p->b) b = pattern_id(p->b.get());
        uint32_t id = (uint32_t)f.pat_kind.size();
        f.pat_kind.push_back((uint32_t)p->kind);
        f.pat_a.push_back(a);
        f.pat_b.push_back(b);
        push_rows3(f.pat_inv, p->inverse);
        push_v3(f.pat_colour, p->colour);
        pattern_ids[p] = id;
        return id;
    }
    // Materials are copied into every leaf of a group (replace_material), so
    // identical bits are merged here: same pattern tree object and the same
    // seven scalars bit for bit.  This only saves memory; equality classes
    // are decided on the leaves themselves.
    uint32_t material_id(const Material& m) {
        auto& cands = materials_by_pattern[m.pattern.get()];
        const double vals[7] = {m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency,
                                m.refractive_index};
        for (uint32_t id : cands)
            if (std::memcmp(&f.material[(size_t)id * 7], vals, sizeof vals) == 0) return id;
        uint32_t id = (uint32_t)f.material_pattern.size();
        for (double v : vals) f.material.push_back(v);
        f.material_pattern.push_back(pattern_id(m.pattern.get()));
        cands.push_back(id);
        return id;
    }

    void node(const Object& o) {
        uint32_t idx = (uint32_t)f.node_kind.size();
        f.node_kind.push_back((uint32_t)o.type);
        f.node_skip.push_back(0);
        f.node_arg.push_back(RT_NONE);
        push_v3(f.node_bbox, o.bbox.min);
        push_v3(f.node_bbox, o.bbox.max);
        switch (o.type) {
            case ObjType::Shape: {
                uint32_t leaf = (uint32_t)f.leaf_shape.size();
                f.node_arg[idx] = leaf;
                f.leaves.push_back(&o);
                f.leaf_shape.push_back((uint32_t)o.shape);
                push_rows3(f.leaf_inv, o.inverse);
                f.leaf_param.push_back(o.minimum);
                f.leaf_param.push_back(o.maximum);
                f.leaf_param.push_back(o.closed ? 1.0 : 0.0);
                if (o.shape == ShapeKind::Triangle) {
                    f.leaf_tri.push_back((uint32_t)(f.tri_data.size() / 18));
                    push_v3(f.tri_data, o.p1); push_v3(f.tri_data, o.e1); push_v3(f.tri_data, o.e2);
                    push_v3(f.tri_data, o.n1); push_v3(f.tri_data, o.n2); push_v3(f.tri_data, o.n3);
                } else {
                    f.leaf_tri.push_back(RT_NONE);
                }
                f.leaf_class.push_back(0);
                f.leaf_material.push_back(material_id(o.material));
                f.leaf_casts_shadow.push_back(o.casts_shadow ? 1 : 0);
                break;
            }
            case ObjType::Group:
                for (const Object& c : o.children) node(c);
                break;
            case ObjType::Csg:
                f.node_arg[idx] = (uint32_t)f.csg_op.size();
                f.csg_op.push_back((uint32_t)o.op);
                node(o.children[0]);
                node(o.children[1]);
                break;
        }
        f.node_skip[idx] = (uint32_t)f.node_kind.size();
    }

    // Class ids: leaf i joins the class of the first earlier leaf it is
    // approx-equal to.  Candidates are found by a sorted sweep over one key
    // value (two approx-equal leaves have approx-equal keys), so the cost is
    // n log n plus the few leaves that share a key.
    void classes() {
        size_t n = f.leaves.size();
        std::vector<double> key(n);
        for (size_t i = 0; i < n; i++) {
            const Object& o = *f.leaves[i];
            double k = o.shape == ShapeKind::Triangle ? o.p1.x + o.transformation.m[0][3] : o.transformation.m[0][3];
            key[i] = std::isnan(k) ? 0.0 : k;
        }
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        std::vector<uint32_t> rep(n);  // smallest leaf index of the class
        std::iota(rep.begin(), rep.end(), 0u);
        for (size_t a = 0; a < n; a++) {
            uint32_t i = order[a];
            for (size_t b = a + 1; b < n; b++) {
                uint32_t j = order[b];
                // keys further apart than any approx_eq tolerance cannot match
                double tol = std::fmax(1e-9, std::fabs(key[i]) * 1e-9);
                if (key[j] - key[i] > tol) break;
                if (shape_approx_eq(*f.leaves[i], *f.leaves[j])) {
                    uint32_t r = std::min(rep[i], rep[j]);
                    rep[i] = r; rep[j] = r;
                }
            }
        }
        // resolve chains and number classes densely in leaf order
        std::vector<uint32_t> cls(n, RT_NONE);
        uint32_t next = 0;
        for (size_t i = 0; i < n; i++) {
            uint32_t r = (uint32_t)i;
            while (rep[r] != r) r = rep[r];
            if (cls[r] == RT_NONE) cls[r] = next++;
            f.leaf_class[i] = cls[r];
        }
        f.n_classes = next;
    }

    void csg_sets() {
        size_t words = (f.n_classes + 63) / 64;
        f.csg_left_bits.assign(f.csg_op.size() * words, 0);
        for (size_t i = 0; i < f.node_kind.size(); i++) {
            if (f.node_kind[i] != RT_NODE_CSG) continue;
            uint32_t csg = f.node_arg[i];
            uint32_t left_begin = (uint32_t)i + 1, left_end = f.node_skip[i + 1];
            for (uint32_t k = left_begin; k < left_end; k++)
                if (f.node_kind[k] == RT_NODE_LEAF) {
                    uint32_t c = f.leaf_class[f.node_arg[k]];
                    f.csg_left_bits[csg * words + c / 64] |= (uint64_t)1 << (c % 64);
                }
        }
    }
};

}  // namespace flatten_detail

// The returned FlatScene points into `scene` (leaves); keep `scene` alive.
inline void flatten_scene(const Scene& scene, FlatScene& f) {
    flatten_detail::Builder b(f);
    for (const Object& o : scene.objects) b.node(o);
    b.classes();
    b.csg_sets();
    for (const Light& l : scene.lights) {
        f.light_kind.push_back(l.is_area ? RT_LIGHT_AREA : RT_LIGHT_POINT);
        if (l.is_area) {
            flatten_detail::push_v3(f.light_data, l.corner);
            flatten_detail::push_v3(f.light_data, l.u);
            flatten_detail::push_v3(f.light_data, l.v);
            flatten_detail::push_v3(f.light_data, l.intensity);
        } else {
            flatten_detail::push_v3(f.light_data, l.position);
            flatten_detail::push_v3(f.light_data, l.intensity);
            for (int k = 0; k < 6; k++) f.light_data.push_back(0.0);
        }
        f.light_steps.push_back(l.u_steps);
        f.light_steps.push_back(l.v_steps);
    }
}

inline rt_camera flatten_camera(const Camera& c) {
    rt_camera r{};
    r.width = c.hsize; r.height = c.vsize;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) r.inv[i * 4 + j] = c.inverse.m[i][j];
    r.half_width = c.half_width; r.half_height = c.half_height; r.pixel_size = c.pixel_size;
    return r;
}

}  // namespace rt
