/*
 * rtx.h -- C ABI of the MI355X trace-loop backend.
 *
 * The reference renderer has no plugin or FFI seam: Scene::render (src/Scene.cpp:185-229) builds
 * the scene, then runs an OpenMP loop that calls the private members traceRay / sceneIntersect /
 * calcLightning directly.  This interface is cut exactly where that loop begins: everything
 * before it (materials, primitives, textures, camera) is handed over as plain arrays, everything
 * inside it (src/Scene.cpp:214-224 and all it calls) happens behind rtx_render(), and what comes
 * back is the pixel store of the reference's bitmap_image: W*H*3 bytes, row 0 on top, B,G,R order
 * (include/bitmap_image.hpp:198-210).
 *
 * Conventions
 *   - every call returns 0 on success or a negative RTX_E_* code; rtx_last_error() gives the text
 *     of the last failure on the calling thread;
 *   - all array arguments are caller-owned host memory, copied during the call;
 *   - a context is driven by one host thread at a time;
 *   - all floating point is IEEE binary32 without fused multiply-add on the paths whose results
 *     reach a pixel, so that results track the reference's x86-64 build.
 *
 * The same header is implemented twice: raytracing_amd/csrc (HIP, gfx950 -- the product) and
 * oracle/rtx_oracle.cpp (CPU restatement -- the parity checker; test infrastructure only).
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_OK 0
#define RTX_E_INVALID (-1)  /* bad argument (null pointer, negative count, index out of range) */
#define RTX_E_STATE (-2)    /* call order: render before commit, scene change after commit ...  */
#define RTX_E_DEVICE (-3)   /* HIP runtime error, text in rtx_last_error()                      */
#define RTX_E_NOMEM (-4)
#define RTX_E_LIMIT (-5)    /* a documented maximum was exceeded                                */

#define RTX_MAX_DEPTH 8     /* recursion depth the per-lane ray stack is sized for              */

typedef struct rtx_ctx rtx_ctx;

/* Camera of Scene::render (src/Scene.cpp:195-202, 217-220).
 * dir = normalize(basis * (x, y, z) + (add_origin ? origin : 0)), ray origin = origin, with
 * x = (col + 0.5) - W/2, y = -(row + 0.5) + H/2, z = -H / (2 * tan_half_fov).
 * The reference adds the camera position to the un-normalised direction (src/Scene.cpp:220);
 * add_origin = 1 keeps that behaviour.
 * tan_half_fov is passed as a number, not derived from an angle behind the interface: the
 * reference's `tan(fov / 2.0f)` (src/Scene.cpp:219) has a constant argument, which its compiler
 * folds to the correctly rounded 0x1.279a74p-1, while glibc's tanf returns 0x1.279a76p-1 for the
 * same input at run time -- one ulp that moves almost every primary direction. */
typedef struct rtx_camera {
    float origin[3];
    float basis[9]; /* 3x3, column-major: basis[3*c + r] = M[c][r] as GLM stores it */
    float tan_half_fov;
    int add_origin;
} rtx_camera;

/* One frame, or the share of one frame a rank owns.  Rows row_first, row_first + row_step, ...
 * below H are rendered and stored compactly: output row k is image row row_first + k*row_step.
 * A whole frame is row_first = 0, row_step = 1. */
typedef struct rtx_frame {
    int width;
    int height;
    int depth;     /* recursion depth, the literal 4 at src/Scene.cpp:221 */
    int row_first;
    int row_step;
} rtx_frame;

/* Ray counts of the last render ("ray" = one sceneIntersect call, src/Scene.cpp:29).  Only filled
 * when the context was rendered with statistics enabled (rtx_set_option RTX_OPT_STATS = 1), which
 * costs one atomic per wavefront per counter.  The ray counts, sky_lookups and lit_lights are per
 * pixel whatever the schedule; in quad mode (RTX_OPT_QUAD, second and later frames of a view) the
 * per-test counters below them also count the tests of the three lanes that repeat a pixel's
 * closest-hit ray. */
typedef struct rtx_stats {
    uint64_t primary_rays;    /* traceRay calls at full depth                                  */
    uint64_t secondary_rays;  /* traceRay calls below it that reach sceneIntersect             */
    uint64_t shadow_rays;     /* sceneIntersect calls from calcLightning                       */
    uint64_t sky_lookups;     /* Sky::getColorAt calls                                         */
    uint64_t mesh_box_hits;   /* rays that passed the mesh bounding box                        */
    uint64_t bvh_node_tests;  /* inner-node (two-box) tests; oracle: non-empty grid cells      */
    uint64_t triangle_tests;  /* Moller-Trumbore tests on mesh triangles                       */
    uint64_t triangle_full;   /* ... of which got past the first (u) rejection                 */
    uint64_t sphere_tests;    /* Sphere::rayIntersect calls                                    */
    uint64_t loose_tri_tests; /* Triangle::rayIntersect calls on the loose triangles           */
    uint64_t box_tests;       /* Cube slab tests (scene boxes and the mesh gate)               */
    uint64_t plane_tests;     /* ray/plane tests (water plane, island plane)                   */
    uint64_t island_tests;    /* ellipse containment tests                                     */
    uint64_t lit_lights;      /* light evaluations that were not in shadow (Phong + powf)      */
    uint64_t node_wave_trips; /* HIP backend: trips of the node loop counted once per wavefront; 64 x
                                 this / bvh_node_tests = how many lane slots each useful node test
                                 costs (SIMT divergence of the mesh walk); 0 on the oracle       */
    uint64_t tri_wave_trips;  /* the same for the triangle test of the mesh walk               */
    double kernel_ms;         /* device time of the trace kernel(s), HIP events                */
} rtx_stats;

enum rtx_option {
    RTX_OPT_STATS = 1,        /* 0/1: count rays (slower)                                          */
    RTX_OPT_PRUNE = 2,        /* 0/1 (default 1): do not trace rays whose contribution is exactly
                                 nothing: children whose albedo weight is exactly 0 (the reference
                                 traces them and multiplies by 0, src/Scene.cpp:95-105) and shadow
                                 rays towards lights that can add to neither the diffuse nor the
                                 specular sum (src/Scene.cpp:71-78).  Same pixels as long as all
                                 colours stay finite (x + 0*c == x needs a finite c)                */
    RTX_OPT_FAR_CLIP = 3,     /* float bits: sceneIntersect's `minDist < 700` (src/Scene.cpp:57)   */
    RTX_OPT_SURFACE_BIAS = 4, /* float bits: fprec = 1e-3 (include/Scene.h:23)                     */
    RTX_OPT_BVH_BUILDER = 6,  /* before rtx_commit, who builds the mesh's bounding volume hierarchy: 0 (default) =
                                 the GPU, top-down by the binned surface-area heuristic; 1 = the GPU, the plain
                                 Morton-order hierarchy; 3 = the GPU, agglomerative clustering along the Morton
                                 order; 2 = the host, surface-area heuristic.  The tree only steers the walk:
                                 results do not change                                                 */
    RTX_OPT_SCHEDULE_FEEDBACK = 7, /* 0/1 (default 1): a frame whose camera is the previous frame's or
                                 close to it (and whose frame description is the same) starts its
                                 tiles in the order of what they cost in that frame; 0 = always the
                                 static order from projected bounding boxes.  Scheduling only: pixels
                                 do not change                                                       */
    RTX_OPT_QUAD = 8,         /* 0/1/2 (default 1): with schedule feedback, tiles that lasted long in
                                 the previous frame are traced with four lanes per pixel
                                 (the shadow rays of a node go out side by side): 0 never, 1 the tiles
                                 that would hold the frame up, 2 every tile.  Pixels do not change   */
    RTX_OPT_SPP = 9,          /* 1 (default) or 4: samples per pixel.  No counterpart in the reference
                                 (one ray through the pixel centre, src/Scene.cpp:217-218).  With 4,
                                 sample k goes through (col + 0.5 + dx[k], row + 0.5 + dy[k]) with
                                 dx = {.125, -.375, .375, -.125}, dy = {.375, .125, -.125, -.375} (a
                                 rotated grid); the pixel is (((c0 + c1) + c2) + c3) * 0.25 in
                                 binary32, then clamped and truncated to bytes as with one sample   */
    RTX_OPT_PIPELINE = 10,    /* 0: one kernel per frame, a lane follows its pixel's whole call tree
                                 (Scene::traceRay's recursion as a per-lane stack); 1: the frame is traced
                                 level by level of that tree -- closest-hit rays, shadow rays and shading
                                 of one recursion depth each in a launch of their own over compacted ray
                                 lists, colours summed bottom-up in the reference's order; 2 (default):
                                 whichever measured faster for the kind of scene (level by level when
                                 the spheres sit behind a BVH, one kernel otherwise).  Same pixels, same
                                 ray counts; level by level has no schedule feedback and no quad mode */
    RTX_OPT_PIPELINE_BUDGET_MB = 11, /* MiB of device memory the arrays of RTX_OPT_PIPELINE = 1 may take (default
                                 65536: an MI355X has 288 GB, and larger strips mean fewer launches with their
                                 ramps and tails -- 10 000 spheres at 4096 x 4096 x 4 samples: 97.0 ms within
                                 16 GiB, 92.5 within 64); a frame that needs more is traced in strips of tiles */
    RTX_OPT_RAY_POOL = 15,    /* 0/1 (default 1): in the level-by-level pipeline of a scene whose spheres sit behind a
                                 tree, the lanes of a wave draw their shadow rays from the list one by one as they
                                 finish, instead of working through 64 of them in step (same rays, same results) */
    RTX_OPT_TILE_PROBE = 13,  /* 0/1 (default 0): the first frame of a view (no costs of a previous frame to go by) puts
                                 its tiles into classes by one probe ray per tile through sceneIntersect (what it hits,
                                 where the children of that hit go) on top of the projected bounding boxes of the mesh
                                 and the mirror / glass spheres.  Measured: a first frame of the mesh over the water
                                 14 % shorter, of the full default scene 10 % longer (the probe launch costs 0.1 ms),
                                 hence off by default.  Scheduling only: pixels do not change                     */
    RTX_OPT_QUAD_KERNEL = 14, /* 0/1/2 (default 1): the quad-mode tiles of RTX_OPT_QUAD are traced by a launch of their
                                 own beside the frame's main launch (on a stream of the context's, forked off the
                                 render's stream and joined to it again before the call returns), in which the idle
                                 fourth lane of a pixel casts the node's first child ray during the shadow trip: a node
                                 then takes one trip instead of two.  0 = never (inside the main launch, without that),
                                 2 = whenever the tile list has quad-mode tiles, 1 = for frames small enough to be
                                 bounded by their longest chains (up to 16 tiles per resident wave: 1920x1080 on an
                                 MI355X, or an eighth of 3840x2160 -- 9 to 12 % shorter there, 8 % longer for a whole
                                 3840x2160 frame).  Renders that are being captured into a graph, and
                                 rtx_render_device_noalloc, always take the first form.  Pixels and ray counts do not
                                 change                                                                            */
    RTX_OPT_PIN_OUTPUT = 12,  /* 0/1 (default 0): rtx_render / rtx_group_render may page-lock the caller's output buffer
                                 (hipHostRegister) and keep it so while the same buffer keeps being passed: the frame
                                 then leaves the GPU at the speed of the link instead of through the runtime's pageable
                                 staging (three times as fast for a 3840x2160 frame).  The caller promises that the
                                 buffer stays allocated until it passes another one, clears the option or destroys the
                                 context.  A buffer that is page-locked already (rtx_host_alloc, hipHostMalloc) gets the
                                 fast copy without this option                                                    */
    RTX_OPT_MESH_GRID = 5     /* 0/1 (default 1), before rtx_commit: 1 = a mesh triangle is hit only
                                 by rays that pass one of the 7x7x7 grid cells holding its vertices,
                                 as the reference's BoundingBox walk has it (src/SceneObjects.cpp:
                                 176-209, 350-380; large triangles have holes); 0 = every triangle
                                 is tested for every ray (watertight mesh)                          */
};

const char *rtx_last_error(void);

/* replaces: `Scene scene;` + the SceneObjects aggregate (include/SceneObjects.h:127-154) */
int rtx_create(rtx_ctx **out, int device_id);
void rtx_destroy(rtx_ctx *ctx);
int rtx_set_option(rtx_ctx *ctx, int option, uint32_t value);

/* replaces: Materials table (src/Material.cpp:7-17).  m9 = n x {refractive, albedo[0..3],
 * diffuse r,g,b, specular exponent}.  Primitives refer to materials by index. */
int rtx_set_materials(rtx_ctx *ctx, const float *m9, int n);
/* replaces: Scene::loadLights (src/Scene.cpp:133-139).  xyzi = n x {x, y, z, intensity};
 * lights are applied in array order (src/Scene.cpp:64). */
int rtx_set_lights(rtx_ctx *ctx, const float *xyzi, int n);
/* replaces: Scene::loadSpheres + SceneObjects::spheresIntersect (src/Scene.cpp:108-115,
 * src/SceneObjects.cpp:15-26, 289-302).  Of two spheres hit at the same distance the one that comes
 * first in the arrays wins (the reference's loop keeps the first, `curDist < spheresDist`).  Up to
 * 16 spheres are tested one after the other, as the reference does; beyond that the HIP backend
 * walks a bounding volume hierarchy over them (same hits, same winner; rtx_stats.sphere_tests then
 * counts the tests the walk performs, not spheres x rays). */
int rtx_set_spheres(rtx_ctx *ctx, const float *cx, const float *cy, const float *cz, const float *r,
                    const int *mat, int n);
/* replaces: Scene::loadCubes + cubesIntersect (src/Scene.cpp:117-131, src/SceneObjects.cpp:75-174,
 * 382-397).  Corners may be given in any order (Cube::load sorts them per axis). */
int rtx_set_boxes(rtx_ctx *ctx, const float *corner_a_xyz, const float *corner_b_xyz, const int *mat,
                  int n);
/* replaces: Scene::loadTriangles + trianglesIntersect (src/Scene.cpp:141-149,
 * src/SceneObjects.cpp:28-49, 304-317).  v9 = n x {v0, v1, v2}; tested linearly, single-sided. */
int rtx_set_triangles(rtx_ctx *ctx, const float *v9, const int *mat, int n);
/* replaces: Model + BoundingBox + modelsIntersect (src/SceneObjects.cpp:176-255, 350-380).
 * verts = nv x {x,y,z} already scaled and offset; idx = nt x 3, zero-based.  The bounding box
 * that gates the mesh is the min/max over ALL nv vertices -/+ 1e-4 (src/SceneObjects.cpp:241-243).
 * The HIP backend builds an LBVH over the triangles on the GPU in rtx_commit(); the oracle builds
 * the reference's 7x7x7 vertex-binned grid. */
int rtx_set_mesh(rtx_ctx *ctx, const float *verts, int nv, const int *idx, int nt, int mat);
/* replaces: Scene::loadIslands + Island + islandsIntersect (src/Scene.cpp:159-166,
 * src/SceneObjects.cpp:273-287, 399-419; include/SceneObjects.h:101-113).
 * cx_cz_a_b = n x {centre x, centre z, semi-axis along x, semi-axis along z}. */
int rtx_set_islands(rtx_ctx *ctx, const float *cx_cz_a_b, int n, float plane_y, int mat,
                    const float *light_rgb, const float *dark_rgb);
/* replaces: the water plane and its texture (src/Scene.cpp:38-54, 235-249).  rgb = h x w x 3
 * bytes as stb_image returns them.  w = 0 removes the plane (the reference always has it). */
int rtx_set_water(rtx_ctx *ctx, const unsigned char *rgb, int w, int h, float plane_y, float scale,
                  int mat);
/* replaces: Sky (src/SceneObjects.cpp:421-445).  row_shift = the literal 100 at :443. */
int rtx_set_envmap(rtx_ctx *ctx, const unsigned char *rgb, int w, int h, int row_shift);

/* Uploads the scene and builds the acceleration structure.  The scene is frozen afterwards. */
int rtx_commit(rtx_ctx *ctx);

/* replaces: the pixel loop of Scene::render (src/Scene.cpp:214-224).
 * bgr_out: host buffer of rtx_frame_rows(frame) * width * 3 bytes.  stats may be NULL. */
int rtx_render(rtx_ctx *ctx, const rtx_camera *cam, const rtx_frame *frame, unsigned char *bgr_out,
               rtx_stats *stats);
/* Page-locked host memory for rtx_render's / rtx_group_render's output (NULL when there is none to be had): a frame is
 * copied into it at the speed of the link.  No counterpart in the reference (its pixel store is plain memory,
 * include/bitmap_image.hpp:83-92); the oracle returns ordinary memory. */
void *rtx_host_alloc(size_t bytes);
void rtx_host_free(void *p);
/* Same, for callers that own device memory (a torch tensor, an RCCL send buffer): d_bgr_out is a
 * device pointer on the context's GPU, hip_stream a hipStream_t (NULL = the context's own stream).
 * The call only enqueues work; the caller synchronises the stream.  One render of a context at a time: the
 * next call on a context must follow the previous one in stream order (same stream, or after the caller has
 * waited).  Exceptions to "only enqueues": the first frame of a larger size than the context has seen
 * allocates its scheduling memory after a device-wide wait, and so does the level-by-level pipeline
 * (RTX_OPT_PIPELINE) when its arrays have to grow -- it keeps the largest block it ever needed, stays within
 * RTX_OPT_PIPELINE_BUDGET_MB and three quarters of the GPU's free memory by tracing in strips, and hands the frame
 * to the one-kernel pipeline when not even one tile's arrays fit.  rtx_render_device_noalloc never allocates.
 * A stream that is being captured into a HIP graph gets no timing events (rtx_get_stats then reports no
 * kernel time).  The oracle has no device and returns RTX_E_STATE. */
int rtx_render_device(rtx_ctx *ctx, const rtx_camera *cam, const rtx_frame *frame, void *d_bgr_out,
                      void *hip_stream);
/* rtx_render_device for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.
 * Same arguments, same pixels; the call performs no hipMalloc / hipFree (nor any other allocating
 * routine): the memory its kernels schedule tiles with (work-queue heads, per-tile classes, the tile
 * list; 4.1 MiB) is a static device array that every GPU the library runs on holds until the process
 * ends.  Limits and rules:
 *   - frames of at most 524 288 tiles of 8x8 pixels (e.g. 7680x4320); larger ones return RTX_E_LIMIT;
 *   - every call on a GPU, from whatever context, shares that GPU's array: calls that overlap on one GPU
 *     (two streams, two threads) race -- serialise them;
 *   - the tile list is made afresh from projected bounding boxes on every call (there is no schedule
 *     feedback and no quad mode), so a frame costs what the first frame of a view costs.
 * The context must have been created on its GPU by rtx_create, which looks the array up. */
int rtx_render_device_noalloc(rtx_ctx *ctx, const rtx_camera *cam, const rtx_frame *frame, void *d_bgr_out,
                              void *hip_stream);
/* Ray counts and kernel time of the most recent render on this context (synchronises). */
int rtx_get_stats(rtx_ctx *ctx, rtx_stats *stats);

/* number of rows a frame description owns: ceil((height - row_first) / row_step) */
int rtx_frame_rows(const rtx_frame *frame);

/* Row-interleaved gather, second half (the first half is an all-gather of equal-sized rank
 * buffers over RCCL): d_gathered holds n_ranks blocks of rows_per_rank * width * 3 bytes, block r
 * being rank r's compact rows (row_first = r, row_step = n_ranks); writes the height x width x 3
 * frame to d_frame.  No counterpart in the reference (single process). */
int rtx_deinterleave_rows_device(rtx_ctx *ctx, const void *d_gathered, int n_ranks, int rows_per_rank,
                                 int width, int height, void *d_frame, void *hip_stream);

/* One frame over several GPUs from one process.  replaces: the OpenMP loop over the rows of the frame and its
 * thread count (src/Scene.cpp:214-215, `-threads`: src/Scene.cpp:232).  Rank r of the group renders rows
 * r, r + n, ... on GPU device_ids[r] (a row-interleaved partition: the cheap sky rows and the expensive rows
 * through the model are spread evenly); the ranks' rows are gathered on device_ids[0] over RCCL (send / recv
 * pairs straight into the root, point-to-point xGMI) and put in place by a de-interleave kernel.  The scene
 * is replicated: set it on every rank's context (rtx_group_ctx) with the rtx_set_* calls, then
 * rtx_group_commit.  A device id may be repeated (several ranks on one GPU, for tests); ranks that share the
 * root's GPU render straight into the gathered buffer.  librccl is loaded at run time, only for groups that
 * span more than one GPU.  One host thread drives the group; errors: rtx_group_last_error(). */
typedef struct rtx_group rtx_group;
int rtx_group_create(rtx_group **out, const int *device_ids, int n_devices);
void rtx_group_destroy(rtx_group *group);
int rtx_group_size(const rtx_group *group);
rtx_ctx *rtx_group_ctx(rtx_group *group, int rank);
int rtx_group_commit(rtx_group *group);
/* rtx_set_option on every rank's context; RTX_OPT_PIN_OUTPUT applies to rtx_group_render's output buffer. */
int rtx_group_set_option(rtx_group *group, int option, uint32_t value);
/* bgr_out: host buffer of height * width * 3 bytes; stats (may be NULL): ray counts summed over the ranks,
 * kernel_ms the slowest rank's.  Waits for the frame. */
int rtx_group_render(rtx_group *group, const rtx_camera *cam, int width, int height, int depth,
                     unsigned char *bgr_out, rtx_stats *stats);
/* The same frame left in the memory of the root's GPU (device_ids[0]): *d_frame_out points at height * width * 3
 * bytes there.  The call only enqueues -- renders on the ranks' streams, the exchange and the de-interleave on a
 * stream per GPU of their own -- and returns; the work that completes the frame is ordered on
 * rtx_group_root_stream(group) (a hipStream_t on the root's GPU): enqueue the consumer there, or record an event
 * there, or call rtx_group_wait.  All buffers exist twice and alternate: the exchange of one frame runs under the
 * render of the next, and a frame stays valid until the call after the next one.  Nothing is allocated after the
 * first frame of a size.  replaces: src/Scene.cpp:214-227 for a caller that keeps the frame on the GPU (display,
 * encoder); the oracle has no device and returns RTX_E_STATE. */
int rtx_group_render_device(rtx_group *group, const rtx_camera *cam, int width, int height, int depth,
                            void **d_frame_out);
void *rtx_group_root_stream(rtx_group *group);
/* waits for everything the group has enqueued */
int rtx_group_wait(rtx_group *group);
const char *rtx_group_last_error(void);

/* Traces caller-supplied rays through Scene::traceRay semantics (src/Scene.cpp:83-106) and
 * returns float RGB per ray, un-clamped.  Used by the parity tests to probe single rays.
 * orig/dir = n x 3, rgb_out = n x 3 (host memory). */
int rtx_trace_rays(rtx_ctx *ctx, const float *orig, const float *dir, int n, int depth, float *rgb_out);

/* Closest-hit query = Scene::sceneIntersect (src/Scene.cpp:29-58) on caller-supplied rays.
 * out16 = n x {hit flag, hit xyz, N xyz, refractive, albedo[0..3], diffuse rgb, specular}. */
int rtx_intersect_rays(rtx_ctx *ctx, const float *orig, const float *dir, int n, float *out16);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
