/* rt_abi.h — the drop-in boundary: C ABI of the MI355X render library (librt_hip.so).
 *
 * This is what replaces the body of the reference's
 *     pub fn render(&self, world: &World) -> Canvas          (src/camera.rs:66-84)
 * The reference-side host (Rust; the C++ mirror in
 * raytracer_amd/csrc/host in this repo) flattens `World` into the plain arrays below and calls
 * rt_scene_create + rt_render.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *  - all floats are IEEE f32, little-endian; all indices are 32-bit;
 *  - every array is caller-owned and only read during the call that receives it
 *    (rt_scene_create copies the scene into HBM; the handle keeps no host pointers);
 *  - functions return RT_OK (0) or a negative RT_E_* code, never throw or abort;
 *    rt_last_error() returns a message for the last failure on that context;
 *  - a context is bound to one GPU and is single-threaded (one call at a time).
 *
 * Layout rules that make the device results equal to the reference's:
 *  - shapes are stored in the reference's INSERTION ORDER (World::intersect gather order,
 *    src/world.rs:44-62: csg leaves, then groups depth-first shapes-before-children
 *    (src/groups.rs:118-123), then loose shapes); the index of a shape is its tie-break rank
 *    and its object id;
 *  - group nodes are the reference's groups verbatim, depth-first, with the bounding boxes the
 *    reference itself would compute (src/groups.rs:166-188 incl. its quirks) — never rebuilt;
 *  - transforms are the exact cofactor inverses (src/matrices.rs:185-197), rows 0..2 only:
 *    Mul<Point>/Mul<Vector> discard the w row (src/matrices.rs:220-236).
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2u

enum {
  RT_OK = 0,
  RT_E_ABI = -1,     /* version/size mismatch, NULL argument, empty camera, frame beyond 65535 pixels a side */
  RT_E_DEVICE = -2,  /* HIP error */
  RT_E_OOM = -3,     /* device allocation failed */
  RT_E_SCENE = -4,   /* scene violates a device limit (chain depth, csg depth, ...) */
  RT_E_OVERFLOW = -5, /* a per-ray intersection list overflowed (see rt_render_device); the image is invalid */
  RT_E_LIMIT = -6     /* an argument exceeds a fixed capacity of the entry point (rt_color_at_static) */
};

/* shape kinds — order of the reference enum (src/shapes.rs:27-36) */
enum { RT_SPHERE = 0, RT_PLANE, RT_CUBE, RT_CYLINDER, RT_CONE, RT_TRIANGLE, RT_SMOOTH_TRIANGLE };
#define RT_SHAPE_KIND_MASK 0xFu
#define RT_SHAPE_SHADOW 0x10u /* Shape::shadow() (src/shapes.rs:294-304) */
#define RT_SHAPE_CLOSED 0x20u /* cylinder/cone `closed` */
#define RT_VCLASS_NEVER 0xFFFFFFFFu /* shape holds a NaN: it is != every shape, itself included */

/* rows 0..2 of an exact inverse transform, row-major; entry 0 of the table is the identity */
typedef struct {
  float m[12];
} RtXform;

/* One shape (32 B). Replaces the by-value `Shape` enum (src/shapes.rs:27-36). */
typedef struct {
  uint32_t kind_flags; /* RT_SHAPE_KIND_MASK | RT_SHAPE_SHADOW | RT_SHAPE_CLOSED */
  uint32_t xform;      /* index into xforms (0 = identity) */
  uint32_t material;   /* index into materials */
  int32_t parent;      /* node index of the owning group, -1 if none (src/shapes.rs:248-258) */
  uint32_t vclass;     /* value class: equal <=> reference `Shape == Shape` (parent ignored) */
  uint32_t prim;       /* (smooth) triangle: index into tri_geom / tri_normals */
  float mn, mx;        /* cylinder/cone min,max (src/shapes/cylinder.rs:22-23) */
} RtShape;

/* Triangle edges (src/shapes/triangle.rs:32-48): p1, e1 = p2-p1, e2 = p3-p1. 48 B, 16-aligned. */
typedef struct {
  float p1[3], pad0;
  float e1[3], pad1;
  float e2[3], pad2;
} RtTriGeom;
/* Flat triangle: n1 = precomputed normal (triangle.rs:35); smooth: n1,n2,n3 (smooth_triangle.rs:24-26). */
typedef struct {
  float n1[3], pad0;
  float n2[3], pad1;
  float n3[3], pad2;
} RtTriNormals;

#define RT_NODE_ALL_TRIS 0x1u /* every shape of the node is an identity-transform (smooth) triangle */
#define RT_NODE_PRUNABLE 0x2u /* the copied box really encloses the subtree: distance pruning allowed */
/* One group (64 B), depth-first order. Replaces GroupData (src/groups.rs:20-26). */
typedef struct {
  float bmin[3];
  uint32_t xform; /* 0 = identity */
  float bmax[3];
  uint32_t flags;
  uint32_t first_shape, n_shapes; /* this group's own shapes: shapes[first_shape .. +n_shapes) */
  uint32_t skip;                  /* index of the first node after this subtree */
  int32_t parent;                 /* parent node or -1 */
  uint32_t first_tri;             /* prim index of shapes[first_shape] when RT_NODE_ALL_TRIS */
  int32_t next_xf;                /* nearest node at-or-above this one with a non-identity transform, -1 none */
  uint32_t idrun;                 /* number of consecutive identity-transform nodes from here upwards */
  uint32_t n_children;
} RtNode;

/* csg ops — src/csg.rs:110-114 */
enum { RT_CSG_UNION = 0, RT_CSG_INTERSECT = 1, RT_CSG_DIFFERENCE = 2 };
#define RT_CSG_LEFT_IS_SHAPE 0x1u
#define RT_CSG_RIGHT_IS_SHAPE 0x2u
/* One Csg node (48 B), post-order within its tree (children before parents). src/csg.rs:7-11. */
typedef struct {
  float bmin[3];
  uint32_t op;
  float bmax[3];
  uint32_t flags;
  int32_t left, right;       /* shape index (shapes[]) or csg node index, per flags */
  uint32_t left_lo, left_hi; /* shape-index range of the leaves under `left` (Csg::includes, src/csg.rs:74-85) */
} RtCsgNode;
/* A top-level Csg of the world: csg_nodes[first .. first+count), root = last. */
typedef struct {
  uint32_t first, count;
} RtCsgRoot;

/* Material (src/materials.rs:6-33), 48 B. pattern = -1 for None. */
typedef struct {
  float color[3];
  float ambient;
  float diffuse, specular, shininess, reflective;
  float transparency, refractive_index;
  int32_t pattern;
  uint32_t pad;
} RtMaterial;

enum { RT_PAT_STRIPED = 0, RT_PAT_GRADIENT, RT_PAT_RING, RT_PAT_CHECKER, RT_PAT_XYZ_RGB, RT_PAT_TEXTURE_MAP, RT_PAT_CUBE_MAP };
enum { RT_UVMAP_SPHERICAL = 0, RT_UVMAP_PLANAR, RT_UVMAP_CYLINDRICAL };
/* Pattern (src/patterns.rs:25-33), 80 B. uv[]: texture map uses uv[0]; cube map uses
 * left,front,right,back,up,down (src/patterns/cube_map.rs:11-19). */
typedef struct {
  uint32_t kind;
  uint32_t xform; /* inverse of the pattern transform, index into xforms */
  uint32_t uv_mapping;
  uint32_t pad;
  float a[3], pad_a;
  float b[3], pad_b;
  int32_t uv[6];
  uint32_t pad2[2];
} RtPattern;

enum { RT_UV_CHECKER = 0, RT_UV_ALIGN_CHECK, RT_UV_IMAGE };
/* UvPattern (src/patterns/uv_pattern.rs:8-12), 80 B. */
typedef struct {
  uint32_t kind;
  float width, height; /* checker */
  int32_t image;       /* index into images */
  float c[5][3];       /* checker: a,b ; align-check: main,ul,ur,bl,br */
  uint32_t pad;
} RtUvPattern;
/* Canvas used as a texture (src/patterns/uv_pattern.rs:109-132): texels[offset ..] w*h*3 f32 */
typedef struct {
  uint32_t width, height;
  uint64_t offset; /* in floats into `texels` */
} RtImage;

enum { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 1 };
/* Light (src/lights.rs:36-40, :65-74), 96 B. uvec/vvec are already divided by the step counts. */
typedef struct {
  uint32_t kind;
  uint32_t usteps, vsteps;
  uint32_t pad;
  float intensity[3], pad_i;
  float position[3], pad_p;
  float corner[3], pad_c;
  float uvec[3], pad_u;
  float vvec[3], pad_v;
} RtLight;

/* The flattened World (src/world.rs:12-18). */
typedef struct {
  uint32_t abi_version;
  uint32_t recursion_limit; /* World::recursion_limit (src/world.rs:17, default 4) */
  uint32_t n_xforms, n_shapes, n_tris, n_nodes, n_csg_nodes, n_csg_roots;
  uint32_t n_materials, n_patterns, n_uvpatterns, n_images, n_lights;
  uint32_t n_loose_shapes; /* trailing shapes[] entries that sit directly in World::shapes */
  uint64_t n_texels;       /* floats */
  const RtXform* xforms;
  const RtShape* shapes;
  const RtTriGeom* tri_geom;
  const RtTriNormals* tri_normals;
  const RtNode* nodes;
  const RtCsgNode* csg_nodes;
  const RtCsgRoot* csg_roots;
  const RtMaterial* materials;
  const RtPattern* patterns;
  const RtUvPattern* uvpatterns;
  const RtImage* images;
  const float* texels;
  const RtLight* lights;
} RtScene;

/* Camera (src/camera.rs:7-16): derived values are computed by the host exactly as
 * Camera::new does (tanf on the host libm, src/camera.rs:19-38). */
typedef struct {
  uint32_t hsize, vsize;
  float pixel_size, half_width, half_height;
  float inv_transform[12]; /* rows 0..2 of transform.inverse() (src/camera.rs:58) */
} RtCamera;

enum { RT_JITTER_CONST = 0, RT_JITTER_COUNTER = 1 };
typedef struct {
  uint32_t jitter_mode; /* area-light jitter source (src/lights.rs:105-121), see DESIGN.md */
  float jitter_value;
  uint64_t seed;
  /* interleaved scanline tiles (multi-GPU): this call renders the rows r with
   * (r / tile_h) % n_parts == part, compacted in order.  n_parts = 1 renders everything.  Prefer a tile_h that is
   * a multiple of 8: the kernel works on 8x8 pixel tiles of the compacted rows. */
  uint32_t tile_h, n_parts, part;
  uint32_t flags; /* RT_RENDER_* */
} RtRenderOptions;
#define RT_RENDER_NO_PRUNE 0x1u   /* visit every node the reference visits (no distance pruning) */
#define RT_RENDER_COUNT_WORK 0x2u /* instrumented kernel: also fill the work counters of RtStats (slower) */
#define RT_RENDER_WORK_MAP 0x8u   /* with COUNT_WORK: RtAux.hit_t receives each pixel's traversal work (box + shape tests) */
#define RT_RENDER_SHORT_STACK 0x10u /* diagnostics: the traversal stack of the auxiliary trees holds 5 entries; a ray whose
                                     * traversal needs more takes the reference's walk through that subtree, which is
                                     * what happens (with 20 entries) in the kernel for worlds of groups only. Same image. */
#define RT_RENDER_NO_ACCEL 0x4u   /* test every own triangle of a group linearly, as the reference does
                                     (groups.rs:118-120), instead of through the library's auxiliary BVHs */

/* Optional per-pixel side outputs of the PRIMARY ray, same compaction as the image. */
typedef struct {
  uint8_t* hit_mask;  /* 1 = hit */
  int32_t* object_id; /* shape index (insertion rank) or -1; of equal shapes met at one t, the one the reference's
                         Intersections keeps (src/intersections.rs:66-73: the last of the first run of equal neighbours) */
  float* hit_t;
} RtAux;

/* Ray counters (one ray = one World::intersect-equivalent query) and timings. */
typedef struct {
  uint64_t primary, shadow, reflect, refract;
  uint64_t overflow;   /* intersection-list overflows (must be 0 for a valid image) */
  /* work counters, only with RT_RENDER_COUNT_WORK (else 0): what the roofline's algorithmic
   * bytes/flops are computed from (DESIGN.md "Measurement") */
  uint64_t node_visits, tri_tests, shape_tests, shaded_hits, node_xforms;
  uint64_t aux_tests;  /* box tests against the library's auxiliary BVHs */
  double kernel_ms;    /* device time of the render kernel(s), HIP events on the render stream */
  double total_ms;     /* wall time of the call */
  /* With RT_RENDER_COUNT_WORK: the rays the REFERENCE casts for the same pixels.  It re-traces every secondary ray
   * once per light (src/world.rs:66-86) where the device traces it once and weights it; equal to
   * primary+shadow+reflect+refract for worlds with one light. */
  uint64_t rays_reference;
} RtStats;

typedef struct RtContext RtContext;         /* one GPU */
typedef struct RtDeviceScene RtDeviceScene; /* a scene resident in that GPU's HBM */

uint32_t rt_abi_version(void);
/* device = HIP device ordinal. */
int rt_context_create(int device, RtContext** out);
void rt_context_destroy(RtContext*);
const char* rt_last_error(const RtContext*);

/* Upload a flattened world; validates limits (RT_E_SCENE). */
int rt_scene_create(RtContext*, const RtScene*, RtDeviceScene** out);
void rt_scene_destroy(RtContext*, RtDeviceScene*);

/* Diagnostics (host only, no GPU needed): the auxiliary BVHs rt_scene_create would build for this
 * scene (raytracer_amd/csrc/rt_accel.h documents the node/triangle layout). No counterpart in the
 * reference: the acceleration data is private to this library; tests use this view to check it. */
typedef struct {
  uint64_t n_nodes, n_tris, n_roots;
  const void* nodes;       /* n_nodes x 128 B */
  const void* tris;        /* n_tris x 48 B */
  const uint32_t* root_of; /* per RtScene node: header index of the BVH covering its subtree, or 0xFFFFFFFF */
  const uint32_t* gate_next; /* per RtScene node inside a covered subtree: next group whose box test is not implied */
  void* owner;
} RtAccelView;
int rt_accel_build(const RtScene*, RtAccelView* out);
/* The same view with the trees of 16384 triangles and more built on the context's GPU, which is how rt_scene_create
 * builds them (raytracer_amd/csrc/rt_accel_device.h: the same splits, boxes and leaves as the host builder's; tests
 * compare the two views). rt_last_error says so when the device builder declined a tree and the host built it. */
int rt_accel_build_device(RtContext*, const RtScene*, RtAccelView* out);
void rt_accel_free(RtAccelView*);

/* Number of rows a (camera, options) pair renders on this part (for buffer sizing). */
uint32_t rt_part_rows(const RtCamera*, const RtRenderOptions*);

/* Camera::render (src/camera.rs:66-84): writes rows*hsize*3 f32 (row-major, unclamped) to HOST memory. */
int rt_render(RtContext*, const RtDeviceScene*, const RtCamera*, const RtRenderOptions*, float* out_rgb,
              RtAux* aux /* host pointers, may be NULL */, RtStats* stats /* may be NULL */);

/* What Scene::render does with the canvas after Camera::render (src/scene.rs:215-224), on the device: ordered
 * dithering (Canvas::apply_dithering, src/canvas.rs:37-55, Bayer matrix of src/effects/dithering.rs:12-37) and the
 * 8-bit quantisation of Canvas::to_ppm (src/canvas.rs:73-75): u8 = round(clamp(c, 0, 1) * 255), half away from zero.
 * RT_DITHER_BAYER2..16 are `apply_dithering::<N>(false)` (black or white), RT_DITHER_BAYER_COLOR is
 * `apply_dithering::<4>(true)` (the matrix value is added to the colour), as src/scene.rs:216-221 maps the CLI flag. */
enum { RT_DITHER_NONE = 0, RT_DITHER_BAYER2 = 1, RT_DITHER_BAYER4 = 2, RT_DITHER_BAYER8 = 3, RT_DITHER_BAYER16 = 4,
       RT_DITHER_BAYER_COLOR = 5 };

/* Camera::render + apply_dithering + the quantisation of to_ppm: writes rows*hsize*3 u8 (row-major RGB) to HOST
 * memory -- a quarter of the bytes rt_render moves. */
int rt_render_u8(RtContext*, const RtDeviceScene*, const RtCamera*, const RtRenderOptions*, uint32_t dither,
                 uint8_t* out_u8, RtAux* aux /* host pointers, may be NULL */, RtStats* stats /* may be NULL */);

/* The same post-processing for a frame (or part: opts gives the interleave, so that a part dithers exactly like the
 * rows of the whole frame) already in device memory: d_rgb rows*hsize*3 f32 -> d_out_u8 rows*hsize*3 u8, asynchronous
 * on `stream` (NULL = the context's stream).  Multi-GPU hosts gather the u8 rows instead of the f32 ones. */
int rt_canvas_to_u8_device(RtContext*, const RtCamera*, const RtRenderOptions*, uint32_t dither, const float* d_rgb,
                           uint8_t* d_out_u8, void* stream);

/* Page-locked host memory for the buffers handed to rt_render / rt_render_u8 / rt_color_at (optional: any host pointer
 * works).  The copy of a frame into pageable memory runs at about 16 GB/s and keeps the calling thread busy; into
 * memory from rt_host_alloc it is a DMA at the link's speed that overlaps the next band's kernel (4K frame as u8:
 * 4.1 ms per call pageable, see DESIGN.md section 5 for the page-locked figure).  The reference's Canvas owns a Vec
 * (src/canvas.rs:10-18); a host that wants this allocates the Vec's storage here instead.  Returns NULL on failure. */
void* rt_host_alloc(RtContext*, uint64_t bytes);
void rt_host_free(RtContext*, void* p);

/* Same, but out_rgb / aux are DEVICE pointers on the context's GPU and nothing is copied back;
 * `stream` is a hipStream_t (NULL = the context's own stream, NOT the legacy null stream: a caller whose
 * other work is on the null stream, e.g. torch's default stream, must create a stream, pass it here and
 * run that other work on it, or synchronise). The call is asynchronous unless
 * stats != NULL (then it synchronises the stream to read the counters and event times).
 *
 * Launches in flight: every call takes its own counter block and work queue from a ring of 8 in the context, so
 * up to 8 asynchronous launches may be in flight on any mix of streams; the 9th waits (stream-side) for the 1st.
 * The calls themselves are still made from one host thread at a time (a context is single-threaded).
 *
 * Per-ray intersection lists.  The reference's list is unbounded (src/intersections.rs:66-73); the device keeps 96
 * entries per ray for the n1/n2 walk of a refracted hit (src/intersections.rs:141-160) and, for a ray that passes them,
 * continues in SIDE LISTS in device memory that belong to the context.  A context has none until one of its launches
 * overflows; then it allocates them (512 entries per ray and resident lane, 1.2 GB on an MI355X) and a further
 * overflow makes them four times as large, up to RT_SIDE_MAX entries (default 8192 = 19 GB; the environment variable
 * RT_SIDE_MAX lowers or raises it).  While a context has side lists its launches run one after another and always
 * through the persistent kernel.
 *   - A synchronous call (stats != NULL, rt_render, rt_render_u8, rt_color_at) does this by itself: it grows the side
 *     lists and runs again, and returns the right image.  RT_E_OVERFLOW from it means a ray passed RT_SIDE_MAX entries
 *     or the allocation failed.
 *   - An asynchronous call (stats == NULL) cannot: its overflow is kept in the context and returned, once, by the NEXT
 *     synchronising call on that context or by rt_check, which also grow the side lists so that rendering the lost
 *     frames again succeeds.  A caller that renders asynchronously must call rt_check (or a synchronous entry point)
 *     before trusting the frames rendered since the last check.
 *   - rt_color_at_static may not allocate: it uses side lists the context already has and reports RT_E_OVERFLOW
 *     otherwise.
 * CSG trees need no such thing: rt_scene_create adds up the most intersections the leaves of a tree can give one ray
 * and refuses (RT_E_SCENE) a tree that could pass the 32 entries of the device's csg list. */
int rt_render_device(RtContext*, const RtDeviceScene*, const RtCamera*, const RtRenderOptions*, float* d_out_rgb,
                     RtAux* d_aux, void* stream, RtStats* stats);

/* Camera::render (src/camera.rs:66-84) over SEVERAL GPUs of one host, called from one thread like the reference's:
 * ctxs[g] / scenes[g] are n contexts (1 <= n <= 16, each listed once) with the same world uploaded to each
 * (rt_scene_create per context).  Context g renders the rows r with (r / tile_h) % n == g (opts->tile_h, 8 if 0;
 * opts->n_parts / part are ignored) -- pixels are independent (src/camera.rs:73-77), area-light jitter is keyed by the
 * absolute pixel, so the canvas is the same bytes for every n -- all GPUs launched before any is waited for.
 *   rt_render_multi         writes the whole canvas, vsize*hsize*3 f32, to HOST memory: every GPU copies its stripes
 *                           straight to their places in out_rgb over its own PCIe link (no GPU-to-GPU traffic).  With
 *                           out_rgb from rt_host_alloc the copies are asynchronous DMAs and one thread drives all GPUs;
 *                           into pageable memory one thread per GPU is started for the call.
 *   rt_render_multi_device  leaves the whole canvas in d_out_rgb on ctxs[0]'s GPU: the parts are gathered there over
 *                           xGMI with RCCL (grouped ncclSend / ncclRecv; librccl.so is opened on first use, RT_RCCL_LIB
 *                           names another file) and put in order by a de-interleave kernel.  n == 1 renders in place.
 * stats (may be NULL): n entries, one per context.  Both are synchronous.  With n == 1 they equal rt_render /
 * rt_render_device byte for byte.  Contexts that share a GPU are accepted (tests on a one-GPU box); the device variant
 * then copies inside that GPU instead of going through RCCL, which takes a device once per communicator. */
#define RT_MULTI_MAX 16u
int rt_device_count(void); /* HIP devices visible to this process (0 if none), for hosts that do not link HIP themselves */
int rt_render_multi(RtContext* const* ctxs, const RtDeviceScene* const* scenes, uint32_t n, const RtCamera*,
                    const RtRenderOptions*, float* out_rgb, RtStats* stats);
int rt_render_multi_device(RtContext* const* ctxs, const RtDeviceScene* const* scenes, uint32_t n, const RtCamera*,
                           const RtRenderOptions*, float* d_out_rgb, RtStats* stats);

/* Synchronises `stream` (NULL: every stream of the context's GPU) and returns RT_E_OVERFLOW if any launch on this
 * context since the last synchronising call overflowed a per-ray list, RT_OK otherwise; clears the condition and gives
 * the context (larger) side lists, so that the same frames rendered again come out right.
 * Where the reference's Camera::render would simply have returned (its Intersections list is unbounded,
 * src/intersections.rs:66-73), this is how the asynchronous path reports that a frame must be rendered again. */
int rt_check(RtContext*, void* stream);

/* World::color_at(ray, remaining) (src/world.rs:91-99) and Intersections::hit()
 * (src/intersections.rs:94-96) for n caller-supplied rays (host pointers): origins/directions
 * n*3, out_rgb n*3, hit_t n (NaN on miss), hit_id n (-1 on miss), hit_uv n*2 (the (u,v) of a smooth-triangle
 * hit, src/intersections.rs:34-41; 0 for every other shape); outputs may be NULL. */
int rt_color_at(RtContext*, const RtDeviceScene*, const RtRenderOptions*, const float* origins,
                const float* directions, uint32_t n, uint32_t remaining, float* out_rgb, float* hit_t,
                int32_t* hit_id, float* hit_uv, RtStats* stats);

/* rt_color_at for callers that cannot allocate device memory.  NO COUNTERPART IN THE REFERENCE: it exists for callers
 * of this library that must not reach hipMalloc / hipFree -- tight serving loops that shade small ray batches, code
 * under HIP graph capture on another stream, harnesses that forbid allocation -- where rt_color_at, which allocates
 * and frees six staging buffers per call, cannot be used.  Same arguments, same results bit for bit; the rays and
 * results are staged through a fixed array in device memory that the library holds per GPU for the life of the
 * process (13 * 4 * RT_STATIC_MAX_RAYS bytes + one counter block = 3.3 MB).
 *   - n > RT_STATIC_MAX_RAYS is rejected with RT_E_LIMIT (split the batch);
 *   - calls no hipMalloc, hipFree or other allocating routine; synchronous like rt_color_at;
 *   - every call on a GPU shares that GPU's array, whatever the context: calls that overlap on one GPU race
 *     (serialise them); the array keeps the last call's contents, which the next call overwrites before reading. */
#define RT_STATIC_MAX_RAYS 65536u
int rt_color_at_static(RtContext*, const RtDeviceScene*, const RtRenderOptions*, const float* origins,
                       const float* directions, uint32_t n, uint32_t remaining, float* out_rgb, float* hit_t,
                       int32_t* hit_id, float* hit_uv, RtStats* stats);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
