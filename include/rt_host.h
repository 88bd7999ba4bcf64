/*
 * rt_host.h - C ABI of the host side (librt_host.so): the callers and data
 * formats either side of the rendering core of rt_api.h.
 *
 * It mirrors the reference's scene front end and output writers so that a
 * test can say "load this file, render it, write it" the way the reference's
 * main does (src/main.rs:18-75):
 *
 *   Scene::from_file                 src/scene/mod.rs:66-90   -> rth_scene_from_file / _text
 *   Scene::generate_random_spheres   src/scene/mod.rs:114-293 -> rth_scene_spheres
 *   Scene::render -> Camera::render  src/scene/mod.rs:92-100  -> rth_render
 *   Canvas::to_ppm                   src/canvas.rs:50-60      -> rth_write_ppm
 *   RgbImage::save (8-bit PNG)       src/main.rs:63-71        -> rth_write_png
 *
 * A scene handle owns the parsed object tree, its flattened form
 * (rt_scene_desc) and, after rth_upload, the device copy (rt_scene).  It also
 * owns the main rng (Xoshiro256++ seeded with `seed`, src/main.rs:31): parsing
 * consumes it exactly like the reference (one u64 per patterned material,
 * src/scene/material.rs:26-31), and rth_row_seeds continues the same stream.
 *
 * Functions returning a handle return NULL on failure; the others return 0 or
 * a negative RT_ERR_* code.  rth_last_error() holds the message; parser
 * messages are the reference's own wording (src/scene/*.rs, obj_parser.rs).
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *rth_last_error(void);

/* YAML scene file (src/scene/mod.rs:66-90); _text takes the file's contents.  `scale` is the --scale argument
 * (src/arguments.rs:29-31), `seed` the --seed argument.  `base_dir` resolves
 * relative `file:` entries of obj shapes; NULL = the process's directory. */
void *rth_scene_from_file(const char *path, double scale, uint64_t seed, const char *base_dir);
void *rth_scene_from_text(const char *yaml, double scale, uint64_t seed, const char *base_dir);
/* --sphere-scene (src/scene/mod.rs:114-293). */
void *rth_scene_spheres(double scale, uint64_t seed);
void rth_scene_free(void *scene);

/* out10: width, height, nodes, leaves, classes, csgs, lights, triangles,
 * materials, patterns. */
void rth_scene_info(void *scene, uint32_t *out10);
const rt_scene_desc *rth_scene_desc(void *scene);
const rt_camera *rth_camera(void *scene);

/* Moves tan(field_of_view / 2) by `ulps` units in the last place before the
 * camera scalars are derived (src/camera.rs:55-73).  Renders made with a libm
 * whose tan() rounds differently are reproduced with ulps = -1 or +1. */
void rth_camera_nudge_half_view(void *scene, int ulps);

/* The `height` u64 row seeds Camera::render draws from the caller's rng
 * (src/camera.rs:159-161).  Advances the scene's rng. */
void rth_row_seeds(void *scene, uint64_t *out);

/* rt_scene_create on the current device; idempotent.  Scenes with a perturbed
 * pattern fail with RT_ERR_UNSUPPORTED (src/pattern/perturbed.rs needs
 * libnoise arithmetic that is not available here). */
int rth_upload(void *scene);
rt_scene *rth_device_scene(void *scene);

/* Scene::render (src/scene/mod.rs:92-100); arguments as rt_render. */
int rth_render(void *scene, uint32_t depth, const uint64_t *row_seeds, const rt_partition *part,
               double *out_rgb, uint8_t *out_rgb8, int out_on_device, rt_stats *stats);
int rth_colour_at(void *scene, uint32_t n, const double *rays, uint32_t depth, const uint64_t *seeds,
                  double *out_rgb);
int rth_probe(void *scene, uint32_t n, const double *rays, double *out);
/* Material::lighting (src/material.rs:47-87) and Lightable::intensity_at
 * (src/light/area.rs:75-90) for caller-supplied points; arguments as rt_lighting
 * and rt_intensity_at. */
int rth_lighting(void *scene, uint32_t n, const double *in, const uint64_t *seeds, double *out_rgb);
int rth_intensity_at(void *scene, uint32_t light, uint32_t n, const double *points, uint64_t seed, double *out);
/* Matrix::invert and determinant of a row-major 4x4
 * (src/math/transformation/matrix.rs:99-123,227-245); RT_ERR_INVALID for a
 * matrix the reference refuses to invert. */
int rth_matrix_inverse(const double *in16, double *out16, double *det);

/* Output writers; rgb8 is width*height*3 bytes, row-major. */
int rth_write_ppm(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height);
int rth_write_png(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
