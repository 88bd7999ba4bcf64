/*
 * rt_api.h - C ABI of the MI355X rendering core (librt_hip.so).
 *
 * The library replaces one call of the reference:
 *
 *   Camera::render(&self, world, depth, single_threaded, output, rng) -> Canvas
 *       reference src/camera.rs:105-191 (reached from src/scene/mod.rs:92-100
 *       and src/main.rs:50-55)
 *
 * together with everything that call runs per pixel (World::colour_at and
 * below, src/world.rs:33-177).  The reference has no FFI of its own; these
 * entry points are what an `extern "C"` block in a Rust `-sys` crate would
 * bind.  All arguments are plain
 * pointers and sizes.  Every function returns 0 on success and a negative
 * code otherwise; the message is available from rt_last_error() on the
 * calling thread.  The caller owns every buffer it passes; the library owns
 * rt_scene.  Calls on one rt_scene must not overlap.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes */
#define RT_OK 0
#define RT_ERR_INVALID (-1)     /* malformed description / argument          */
#define RT_ERR_HIP (-2)         /* a HIP runtime call failed                 */
#define RT_ERR_UNSUPPORTED (-3) /* scene uses a feature the device path lacks */
#define RT_ERR_CAPACITY (-4)    /* a per-ray list overflowed its fixed bound */
#define RT_ERR_NAN (-5)         /* an intersection distance was NaN (the
                                   reference panics: src/intersection/list.rs:39-50) */

/* node kinds, shape kinds, csg operations, pattern kinds, light kinds */
enum { RT_NODE_LEAF = 0, RT_NODE_GROUP = 1, RT_NODE_CSG = 2 };
enum { RT_SPHERE = 0, RT_PLANE = 1, RT_CUBE = 2, RT_CYLINDER = 3, RT_CONE = 4, RT_TRIANGLE = 5 };
enum { RT_CSG_DIFFERENCE = 0, RT_CSG_INTERSECTION = 1, RT_CSG_UNION = 2 };
enum { RT_PAT_SOLID = 0, RT_PAT_STRIPE = 1, RT_PAT_RING = 2, RT_PAT_CHECKER = 3,
       RT_PAT_GRADIENT = 4, RT_PAT_RADIAL_GRADIENT = 5, RT_PAT_BLEND = 6 };
enum { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 1 };

#define RT_NONE 0xFFFFFFFFu

/*
 * Flattened world.  Replaces `World { objects: Vec<Object>, lights }`
 * (reference src/world.rs:13-16) and the Object tree under it
 * (src/object/mod.rs:36-40).  All arrays are host pointers and are copied by
 * rt_scene_create.
 *
 * Nodes are listed in depth-first pre-order of the tree whose implicit root
 * is the world: world.objects in order, each Group's children in stored
 * order, a Csg's left subtree before its right.  node_skip[i] is the index
 * of the first node after the subtree of node i, so the children of i are
 * i+1, node_skip[i+1], ... while < node_skip[i].  Leaves are numbered in the
 * same order (leaf k is the k-th leaf node met), which makes "smaller leaf
 * index" the insertion order the reference's stable sorts preserve
 * (src/world.rs:90-106, src/intersection/list.rs:24-50).
 */
typedef struct rt_scene_desc {
    uint32_t n_nodes;
    const uint32_t *node_kind; /* RT_NODE_*                                         */
    const uint32_t *node_skip; /* first node after this subtree                     */
    const uint32_t *node_arg;  /* leaf: leaf index; csg: csg index; group: unused   */
    const double *node_bbox;   /* n_nodes*6: min xyz, max xyz of the STORED box of a
                                  group / csg (src/object/group/mod.rs:20,
                                  src/object/csg/mod.rs:18); ignored for leaves.  A
                                  csg's box may be stale (csg/mod.rs:117-121) and only
                                  ever decides what the reference lets it decide; a
                                  group's finite box over bounded shapes must hold
                                  everything below it, as Group::bounding_box sees to
                                  (group/mod.rs:109-139): the walk skips such a group
                                  when it enters it beyond the closest hit so far     */

    uint32_t n_leaves;
    const uint32_t *leaf_shape;    /* RT_SPHERE ...                                 */
    const double *leaf_inv;        /* n_leaves*12: rows 0-2 of the inverse
                                      transformation, row-major (shape.rs:28)       */
    const double *leaf_param;      /* n_leaves*3: cylinder/cone minimum, maximum,
                                      closed (0.0 / 1.0); unused otherwise          */
    const uint32_t *leaf_tri;      /* triangle index or RT_NONE                     */
    const uint32_t *leaf_class;    /* structural-equality class (shape.rs:108)      */
    const uint32_t *leaf_material; /* material index                                */
    const uint8_t *leaf_casts_shadow;

    uint32_t n_tris;
    const double *tri_data; /* n_tris*18: p1, e1, e2, n1, n2, n3 (triangle.rs:12-21) */

    uint32_t n_csgs;
    const uint32_t *csg_op;       /* RT_CSG_*                                       */
    uint32_t n_classes;           /* class ids are < n_classes                      */
    const uint64_t *csg_left_bits; /* n_csgs * ceil(n_classes/64) words: bit c set
                                      when some leaf of class c is in the left
                                      subtree (csg/mod.rs:67 `left.includes`)      */

    uint32_t n_materials;
    const double *material;          /* n_materials*7: ambient, diffuse, specular,
                                        shininess, reflective, transparency,
                                        refractive_index (material.rs:15-32)       */
    const uint32_t *material_pattern; /* root pattern node of each material        */

    uint32_t n_patterns;
    const uint32_t *pat_kind; /* RT_PAT_*                                           */
    const uint32_t *pat_a;    /* child pattern nodes (RT_NONE for solid)            */
    const uint32_t *pat_b;
    const double *pat_inv;    /* n_patterns*12: rows 0-2 of the pattern's inverse   */
    const double *pat_colour; /* n_patterns*3: colour of a solid pattern            */

    uint32_t n_lights;
    const uint32_t *light_kind; /* RT_LIGHT_*                                       */
    const double *light_data;   /* n_lights*12: point: position, intensity, 6 unused;
                                   area: corner, u cell vector, v cell vector,
                                   intensity (light/area.rs:10-18: u, v already
                                   divided by the step counts)                     */
    const uint32_t *light_steps; /* n_lights*2: usteps, vsteps (1, 1 for a point)  */
} rt_scene_desc;

/* Camera (reference src/camera.rs:23-31): sizes, rows 0-2 of the inverse view
 * transformation and the three derived scalars of Camera::calculate. */
typedef struct rt_camera {
    uint32_t width, height;
    double inv[12];
    double half_width, half_height, pixel_size;
} rt_camera;

/* Which image rows a call renders: rows y with (y / block_rows) % world == rank.
 * {0, 0, 1} or a NULL pointer selects every row.  block_rows must be a
 * multiple of 8 when world > 1.  Output buffers then hold only the selected
 * rows, in increasing y. */
typedef struct rt_partition {
    uint32_t block_rows, rank, world;
} rt_partition;

typedef struct rt_stats {
    uint64_t primary_rays;   /* one per rendered pixel                              */
    uint64_t secondary_rays; /* colour_at calls from reflected_/refracted_colour
                                (src/world.rs:142,175)                             */
    uint64_t shadow_rays;    /* is_shadowed calls (src/world.rs:109)               */
    uint64_t shade_hits;     /* shade_hit calls (src/world.rs:52)                  */
    uint32_t levels;         /* deepest recursion level that held rays, plus one   */
    uint32_t chunks;         /* row chunks the frame was rendered in               */
    double ms_total;         /* wall time of the whole call                        */
    double ms_trace;         /* HIP-event time on the chunk's stream from the first
                                to the last k_trace launch (closest hit + n1/n2 +
                                spawn, one launch per level), summed over chunks;
                                with two streams the chunks overlap, so the three
                                sums can exceed ms_total                           */
    double ms_shadow;        /* the same for k_shadow (shadow rays, one per chunk;
                                only launched for scenes with an area light or a
                                tree that is walked lane by lane)                  */
    double ms_shade;         /* the same for the k_shade launches (patterns + Phong
                                + the sum with the spawned rays' colours)          */
    uint64_t trace_launches; /* number of k_trace launches                         */
    uint64_t flop_trace;     /* algorithmic f64 FLOP counted on the device in
                                k_trace / k_shadow / k_shade: op-class counts x the
                                per-op constants F_* of csrc/rt_device.hip         */
    uint64_t flop_shadow;    /* shadow-ray walks (inside k_trace when fused_shadows) */
    uint64_t flop_light;
    uint64_t fused_shadows;  /* 1: k_trace cast the shadow rays itself and ms_shadow
                                is 0; 0: k_shadow did (area lights, meshes)        */
    uint64_t mesh_nodes;     /* nodes of the meshes' own hierarchies visited (128 bytes
                                fetched each) and triangles tested in their leaves  */
    uint64_t mesh_tris;      /* (80 bytes each), by all rays of the frame           */
    uint64_t big_chunks;     /* chunks rendered a second time with per-ray lists in
                                global memory, after a fixed-size list overflowed    */
} rt_stats;

typedef struct rt_scene rt_scene;

/* Select the HIP device this thread renders on (one process per GPU). */
int rt_init(int device);

/* Upload a flattened world.  Rejects malformed trees (RT_ERR_INVALID) and
 * pattern kinds outside RT_PAT_* (RT_ERR_UNSUPPORTED). */
int rt_scene_create(const rt_scene_desc *desc, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);

/*
 * Render.  Replaces Camera::render (src/camera.rs:105-191).
 *   depth      the `depth` argument (src/arguments.rs:33-35)
 *   row_seeds  camera->height u64 values: the per-row seeds the reference
 *              draws from the caller's rng (src/camera.rs:159-161); may be
 *              NULL when the scene has no area light (no other consumer)
 *   part       row partition, or NULL for the whole image
 *   out_rgb    rows*width*3 f64, row-major linear colour (Canvas::pixels,
 *              src/canvas.rs:7-11), or NULL
 *   out_rgb8   rows*width*3 bytes of Colour::to_u8 (src/colour.rs:63-72), or NULL
 *   out_on_device  non-zero when the two output pointers are device memory
 *   stats      optional
 * Host outputs receive each chunk of rows as it is finished, beside the next
 * chunk's kernels.  Page-locked memory (rt_host_alloc, hipHostMalloc) is
 * written directly; ordinary memory is locked for the length of the call where
 * the driver allows that (RT_OPT_PIN_OUTPUTS) and is otherwise filled through
 * a page-locked staging buffer the scene keeps.  A call that fails returns
 * only after every kernel and copy it queued has ended.
 */
int rt_render(rt_scene *scene, const rt_camera *camera, uint32_t depth, const uint64_t *row_seeds,
              const rt_partition *part, double *out_rgb, uint8_t *out_rgb8, int out_on_device,
              rt_stats *stats);

/*
 * The same frame over n GPUs from one process: scenes[k] is a replica of the
 * world created on GPU k (rt_init(k), then rt_scene_create) and renders the
 * row blocks b with b % n == k, b = y / block_rows, side by side with the
 * others (the fan-out of src/camera.rs:166-179 over devices); every block is
 * then copied to its rows of the caller's frame, which is the whole Canvas
 * (height*width*3) in host memory or, with out_on_device, on scenes[0]'s GPU.
 * block_rows: a multiple of 8.  stats: n entries, or NULL.  Scenes may share a
 * GPU (they then take turns).  Nothing is exchanged while tracing.
 */
int rt_render_multi(rt_scene *const *scenes, uint32_t n, const rt_camera *camera, uint32_t depth,
                    const uint64_t *row_seeds, uint32_t block_rows, double *out_rgb, uint8_t *out_rgb8,
                    int out_on_device, rt_stats *stats);

/*
 * Page-locked host memory for the outputs of rt_render (the pixels of a Canvas,
 * src/canvas.rs:7-11): the device writes finished chunks straight into it, with
 * no staging copy.  No counterpart in the reference.
 */
int rt_host_alloc(void **out, size_t bytes);
int rt_host_free(void *buffer);

/*
 * World::colour_at for caller-supplied rays (src/world.rs:33-49): n rays of
 * 6 f64 (origin, direction), host memory; out_rgb n*3 f64, host memory.
 * Ray i uses an rng seeded with seeds[i] (area lights only; may be NULL).
 */
int rt_colour_at(rt_scene *scene, uint32_t n, const double *rays, uint32_t depth,
                 const uint64_t *seeds, double *out_rgb);

/*
 * Material::lighting (src/material.rs:47-87) for caller-supplied shade points,
 * as the reference's unit tests call it.  Per point 12 f64, host memory: leaf
 * index of the object, light index, point, eye vector, normal vector,
 * intensity.  The generator of point i is seeded with seeds[i] (area lights;
 * NULL: 0) and stands at lighting's own first draw.  out_rgb: n*3 f64.
 */
int rt_lighting(rt_scene *scene, uint32_t n, const double *in, const uint64_t *seeds, double *out_rgb);

/*
 * Lightable::intensity_at (src/light/point.rs:27-38, src/light/area.rs:75-90)
 * of light `light` at n points (n*3 f64, host memory), one call after the
 * other on ONE generator seeded with `seed`, as src/light/area.rs:164-198
 * does.  The reference's walk, without device-only shortcuts.  out: n f64.
 */
int rt_intensity_at(rt_scene *scene, uint32_t light, uint32_t n, const double *points, uint64_t seed,
                    double *out);

/*
 * rt_colour_at for callers that cannot allocate (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart
 * in the reference.  Same kernels and bit-identical colours as rt_colour_at, but
 * every device buffer the kernels need (ray records, level table, counters,
 * statistics) is carved out of one static 64 MiB __device__ array that each GPU
 * holds from the time the library is loaded; the call makes no hipMalloc /
 * hipFree and creates no stream or event (it launches on the null stream and
 * returns when the colours are written).
 *   rays_dev     n*6 f64 (origin, direction), DEVICE memory
 *   out_rgb_dev  n*3 f64, DEVICE memory
 * Limits: n <= rt_noalloc_max_rays() (65536), depth <= 1022, no area light
 * (RT_ERR_UNSUPPORTED), and the rays together may not spawn more ray records
 * than the array holds (about 64 MiB / (144 + 8 * lights) bytes):
 * RT_ERR_CAPACITY with a message that says so.  All calls on one GPU share the
 * array: calls that overlap on one GPU race, whatever scene they use.
 */
int rt_colour_at_noalloc(rt_scene *scene, uint32_t n, const double *rays_dev, uint32_t depth,
                         double *out_rgb_dev);
uint32_t rt_noalloc_max_rays(void);

/*
 * Intersection::prepare_computations for the hit of each ray
 * (src/intersection/mod.rs:45-109, src/world.rs:90-126).  Per ray 32 f64:
 *  [0] hit leaf index or -1, [1] t, [2..4] point, [5..7] over_point,
 *  [8..10] under_point, [11..13] eye, [14..16] normal, [17] inside,
 *  [18..20] reflect, [21] n1, [22] n2, [23] schlick, [24] number of
 *  intersections delivered to the world list, [25..28] is_shadowed towards
 *  the first four lights' positions (area light: its corner) from over_point,
 *  [29..31] unused.
 */
int rt_probe(rt_scene *scene, uint32_t n, const double *rays, double *out);

/*
 * Tuning knobs of a scene's renderer (no counterpart in the reference; the
 * defaults suit a 288 GB MI355X).  Results do not depend on them.
 *   RT_OPT_STREAMS         1..4: image-row chunks in flight at once (default 2)
 *   RT_OPT_CHUNK_PIXELS    most pixels in a chunk (default 26 Mi, less when the record
 *                          arrays would not fit the device's memory); a chunk is
 *                          whole rows, a multiple of 8 of them, and a frame is cut
 *                          into equal chunks, a multiple of the stream count
 *   RT_OPT_RECS_PER_PIXEL  ray records reserved per pixel of a chunk (default
 *                          24); a chunk that needs more is split and retried
 *   RT_OPT_GRID_BLOCKS     blocks per kernel launch (default 0 = 4096 with one stream;
 *                          with more, 1024 for the tracing and 2048 for the shading
 *                          kernels, so that chunks on different streams share the GPU)
 *   RT_OPT_TRAVERSAL       0: chosen by tree size (default); 1: the 64 lanes of a
 *                          wave walk the tree together; 2: every lane walks on
 *                          its own (large trees; meshes by a hierarchy of their
 *                          own); 3: the reference's walk, test for test, with none
 *                          of the device-only shortcuts (slow; what the others
 *                          are checked against); 4: the wave together down to
 *                          the meshes, every lane on its own inside them (trees
 *                          that are small apart from their meshes)
 *   RT_OPT_MIN_CHUNKS      a frame of two megapixels or more is cut into at least
 *                          this many chunks (default 2)
 *   RT_OPT_PIN_OUTPUTS     1 (default): rt_render tries to page-lock ordinary host
 *                          outputs for the length of the call; 0: it goes straight
 *                          to its staging buffer
 */
enum { RT_OPT_STREAMS = 1, RT_OPT_CHUNK_PIXELS = 2, RT_OPT_RECS_PER_PIXEL = 3, RT_OPT_GRID_BLOCKS = 4,
       RT_OPT_TRAVERSAL = 5, RT_OPT_MIN_CHUNKS = 6, RT_OPT_PIN_OUTPUTS = 7 };
int rt_set_option(rt_scene *scene, int option, uint64_t value);

/*
 * Limits of this build (per-ray list bounds).  The reference's lists are
 * unbounded Vecs (src/object/csg/mod.rs:88-106, src/intersection/mod.rs:62-94).
 * Here a ray keeps 32 entries of each list in fast memory; rt_render and
 * rt_colour_at render a chunk in which some ray needs more a second time with
 * kernels that keep the lists in global memory (rt_stats.big_chunks counts
 * them), up to the numbers below and 64 nested csgs; past those the call fails
 * with RT_ERR_CAPACITY.  rt_probe and rt_colour_at_noalloc have the fast lists
 * only.
 */
uint32_t rt_max_csg_list(void);       /* intersections inside one outermost csg    */
uint32_t rt_max_container_list(void); /* intersections at or before a refractive hit */

const char *rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
