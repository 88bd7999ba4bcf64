// Host library (librt_host.so): the scene side of the renderer behind a small
// C API for language bindings.  It builds the scene exactly as the reference's
// host code does (src/main.rs:27-55: seed the rng, parse or generate the scene
// with it, draw the per-row seeds), flattens it and drives librt_hip.so through
// the C ABI of include/rt_api.h.  There is no CPU rendering path here: every
// pixel comes from the device library.
//
// Also holds the output writers: Canvas::to_ppm (src/canvas.rs:50-60) and the
// 8-bit RGB PNG the reference writes through the image crate (src/main.rs:63-71).
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/rt_api.h"
#include "flatten.h"
#include "loader.h"
#include "scene.h"

using namespace rt;

namespace {
thread_local std::string g_error;

struct HostScene {
    Scene scene;
    Xoshiro256pp rng;  // the main rng after scene construction
    FlatScene flat;
    rt_scene_desc desc{};
    rt_camera camera{};
    rt_scene* device = nullptr;
    bool has_perturbed = false;

    void finish() {
        flatten_scene(scene, flat);
        desc = flat.desc();
        camera = flatten_camera(scene.camera);
    }
    ~HostScene() {
        if (device) rt_scene_destroy(device);
    }
};

uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(uint32_t crc, const uint8_t* p, size_t n) {
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put32(out, (uint32_t)data.size());
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put32(out, crc32(0, body.data(), body.size()));
}
}  // namespace

extern "C" {

const char* rth_last_error() { return g_error.c_str(); }

void* rth_scene_from_file(const char* path, double scale, uint64_t seed, const char* base_dir) {
    try {
        auto s = std::make_unique<HostScene>();
        s->rng = Xoshiro256pp::seed_from_u64(seed);
        s->scene = scene_from_file(path, scale, s->rng, base_dir ? base_dir : "", &s->has_perturbed);
        s->finish();
        return s.release();
    } catch (const std::exception& e) { g_error = e.what(); return nullptr; }
}
void* rth_scene_from_text(const char* text, double scale, uint64_t seed, const char* base_dir) {
    try {
        auto s = std::make_unique<HostScene>();
        s->rng = Xoshiro256pp::seed_from_u64(seed);
        s->scene = scene_from_yaml_text(text, scale, s->rng, base_dir ? base_dir : "", &s->has_perturbed);
        s->finish();
        return s.release();
    } catch (const std::exception& e) { g_error = e.what(); return nullptr; }
}
// Scene::generate_random_spheres (src/scene/mod.rs:114-293), `--sphere-scene`
void* rth_scene_spheres(double scale, uint64_t seed) {
    try {
        auto s = std::make_unique<HostScene>();
        s->rng = Xoshiro256pp::seed_from_u64(seed);
        s->scene = generate_random_spheres(scale, s->rng);
        s->finish();
        return s.release();
    } catch (const std::exception& e) { g_error = e.what(); return nullptr; }
}
void rth_scene_free(void* h) { delete (HostScene*)h; }

// width, height, nodes, leaves, classes, csgs, lights, triangles, materials, patterns
void rth_scene_info(void* h, uint32_t* out10) {
    auto* s = (HostScene*)h;
    out10[0] = s->camera.width; out10[1] = s->camera.height;
    out10[2] = s->desc.n_nodes; out10[3] = s->desc.n_leaves; out10[4] = s->desc.n_classes;
    out10[5] = s->desc.n_csgs; out10[6] = s->desc.n_lights; out10[7] = s->desc.n_tris;
    out10[8] = s->desc.n_materials; out10[9] = s->desc.n_patterns;
}
const rt_scene_desc* rth_scene_desc(void* h) { return &((HostScene*)h)->desc; }
const rt_camera* rth_camera(void* h) { return &((HostScene*)h)->camera; }

// See Camera::half_view_ulps (scene.h): reproduces renders made with another libm.
void rth_camera_nudge_half_view(void* h, int ulps) {
    auto* s = (HostScene*)h;
    s->scene.camera.half_view_ulps = ulps;
    s->scene.camera.calculate();
    s->camera = flatten_camera(s->scene.camera);
}

// The `height` per-row seeds Camera::render draws from the caller's rng
// (src/camera.rs:159-161).  Advances the scene's rng like the reference does.
void rth_row_seeds(void* h, uint64_t* out) {
    auto* s = (HostScene*)h;
    for (uint32_t y = 0; y < s->camera.height; y++) out[y] = s->rng.next_u64();
}

// Upload the flattened world to the current HIP device (rt_scene_create).
int rth_upload(void* h) {
    auto* s = (HostScene*)h;
    if (s->device) return RT_OK;
    if (s->has_perturbed || s->flat.has_perturbed) {
        g_error = "perturbed patterns are not available on the device path (libnoise arithmetic is not vendored)";
        return RT_ERR_UNSUPPORTED;
    }
    int rc = rt_scene_create(&s->desc, &s->device);
    if (rc != RT_OK) g_error = rt_last_error();
    return rc;
}
rt_scene* rth_device_scene(void* h) { return ((HostScene*)h)->device; }

// Scene::render -> Camera::render (src/scene/mod.rs:92-100, src/camera.rs:105-191).
int rth_render(void* h, uint32_t depth, const uint64_t* row_seeds, const rt_partition* part, double* out_rgb,
               uint8_t* out_rgb8, int out_on_device, rt_stats* stats) {
    auto* s = (HostScene*)h;
    int rc = rth_upload(h);
    if (rc != RT_OK) return rc;
    rc = rt_render(s->device, &s->camera, depth, row_seeds, part, out_rgb, out_rgb8, out_on_device, stats);
    if (rc != RT_OK) g_error = rt_last_error();
    return rc;
}
int rth_colour_at(void* h, uint32_t n, const double* rays, uint32_t depth, const uint64_t* seeds, double* out_rgb) {
    auto* s = (HostScene*)h;
    int rc = rth_upload(h);
    if (rc != RT_OK) return rc;
    rc = rt_colour_at(s->device, n, rays, depth, seeds, out_rgb);
    if (rc != RT_OK) g_error = rt_last_error();
    return rc;
}
int rth_probe(void* h, uint32_t n, const double* rays, double* out) {
    auto* s = (HostScene*)h;
    int rc = rth_upload(h);
    if (rc != RT_OK) return rc;
    rc = rt_probe(s->device, n, rays, out);
    if (rc != RT_OK) g_error = rt_last_error();
    return rc;
}

// Material::lighting and Lightable::intensity_at for caller-supplied points (rt_lighting, rt_intensity_at)
int rth_lighting(void* h, uint32_t n, const double* in, const uint64_t* seeds, double* out_rgb) {
    auto* s = (HostScene*)h;
    int rc = rth_upload(h);
    if (rc != RT_OK) return rc;
    rc = rt_lighting(s->device, n, in, seeds, out_rgb);
    if (rc != RT_OK) g_error = rt_last_error();
    return rc;
}
int rth_intensity_at(void* h, uint32_t light, uint32_t n, const double* points, uint64_t seed, double* out) {
    auto* s = (HostScene*)h;
    int rc = rth_upload(h);
    if (rc != RT_OK) return rc;
    rc = rt_intensity_at(s->device, light, n, points, seed, out);
    if (rc != RT_OK) g_error = rt_last_error();
    return rc;
}
// Matrix::invert and determinant of a row-major 4x4 (src/math/transformation/matrix.rs:99-123,227-245):
// the routine every inverse handed to the device comes from, exposed so that the reference's known
// answers can be run against it.  Returns RT_ERR_INVALID for a matrix the reference refuses to invert.
int rth_matrix_inverse(const double* in16, double* out16, double* det) {
    Mat4 m;
    std::memcpy(m.m, in16, sizeof m.m);
    *det = det4(m);
    try {
        Mat4 r = invert(m);
        std::memcpy(out16, r.m, sizeof r.m);
    } catch (const std::exception& e) { g_error = e.what(); return RT_ERR_INVALID; }
    return RT_OK;
}

// Canvas::to_ppm (src/canvas.rs:50-60): "P3\nW H\n255\n" then one "r g b\n" per pixel.
int rth_write_ppm(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_error = std::string("cannot open '") + path + "' for writing"; return RT_ERR_INVALID; }
    std::string data = "P3\n" + std::to_string(width) + " " + std::to_string(height) + "\n255\n";
    data.reserve(data.size() + (size_t)width * height * 12);
    char buf[16];
    for (size_t i = 0; i < (size_t)width * height; i++) {
        int n = std::snprintf(buf, sizeof buf, "%u %u %u\n", rgb8[3 * i], rgb8[3 * i + 1], rgb8[3 * i + 2]);
        data.append(buf, (size_t)n);
    }
    size_t w = std::fwrite(data.data(), 1, data.size(), f);
    std::fclose(f);
    if (w != data.size()) { g_error = "short write"; return RT_ERR_INVALID; }
    return RT_OK;
}
// 8-bit RGB PNG (src/main.rs:63-71), written with stored deflate blocks.
int rth_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (width * 3 + 1));
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb8 + (size_t)y * width * 3, rgb8 + (size_t)(y + 1) * width * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        bool last = pos + n == raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        for (size_t i = 0; i < n; i++) {
            uint8_t c = raw[pos + i];
            z.push_back(c);
            a = (a + c) % 65521; b = (b + a) % 65521;
        }
        pos += n;
        if (last) break;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, width); put32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_error = std::string("cannot open '") + path + "' for writing"; return RT_ERR_INVALID; }
    size_t w = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (w != out.size()) { g_error = "short write"; return RT_ERR_INVALID; }
    return RT_OK;
}

}  // extern "C"
