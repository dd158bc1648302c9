/*
 * rt_host.h -- host side of the renderer: what the reference does before and after its pixel loop.
 *
 * These calls sit ABOVE the rtx.h boundary and are plain C++ on the CPU: the material table, the
 * hard-coded scene, the model and texture loaders, the BMP writer and the command line.  They feed
 * a backend through the rtx_set_* calls of rtx.h and never touch a pixel themselves.
 * Citations are relative to the reference checkout.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* scene subset bits for rt_scene_load_default (the reference always loads all of them) */
#define RT_MASK_SPHERES 1u
#define RT_MASK_CUBES 2u
#define RT_MASK_TRIANGLES 4u
#define RT_MASK_MODEL 8u
#define RT_MASK_ISLANDS 16u
#define RT_MASK_ALL 31u

/* replaces: Materials::Materials (src/Material.cpp:7-17).  Writes 9 rows of
 * {refractive, albedo[0..3], diffuse r,g,b, specular}; returns the row count (9).
 * Row order = the reference's: none, glass, mirror, gold, silver, water_bottom, water_top,
 * island, wood. */
int rt_default_materials(float *m9);
/* replaces: Materials::get (src/Material.cpp:19-27): index of a material by name, or 0 (the
 * default material) with a message on stderr when the name is unknown. */
int rt_material_index(const char *name);

/* replaces: the camera set-up of Scene::render (src/Scene.cpp:195-202) */
int rt_default_camera(rtx_camera *cam);

/* replaces: Model::Model's parser (src/SceneObjects.cpp:211-255): "<nv> <nf>", nv x "v x y z",
 * nf x "f a b c" (1-based); coordinates parsed as double, narrowed to float, then * scale + offset.
 * Arrays are malloc'ed; release with rt_free.  idx is zero-based. */
int rt_load_model(const char *path, float scale, const float *offset_xyz, float **verts, int *nv, int **idx, int *nt);
void rt_free(void *p);

/* replaces: stbi_load(path, &w, &h, &n, 0) as used at src/Scene.cpp:236 and src/SceneObjects.cpp:423.
 * Baseline and progressive 8-bit JPEG, 1 or 3 components; output is always 3 bytes per pixel... */
int rt_decode_jpeg(const char *path, unsigned char **rgb, int *w, int *h, int *ncomp);

/* replaces: the scene construction of Scene::run + Scene::render (src/Scene.cpp:186-193, 235-249):
 * materials, lights, spheres, cubes, loose triangles, the bunny model, islands, water texture and
 * environment map, handed to the backend behind `ctx`.  resources_dir holds bunny.obj and water.jpg;
 * envmap_path is the panorama (the reference's literal ../resources/map2.jpg is not in its
 * checkout).  The caller commits. */
int rt_scene_load_default(rtx_ctx *ctx, const char *resources_dir, const char *envmap_path, unsigned mask);
/* Same, with textures already decoded (h x w x 3 bytes each). */
int rt_scene_load_default_tex(rtx_ctx *ctx, const char *bunny_obj_path, unsigned mask, const unsigned char *water_rgb,
                              int water_w, int water_h, const unsigned char *sky_rgb, int sky_w, int sky_h);

/* replaces: bitmap_image::save_image (include/bitmap_image.hpp:435-483) for a top-down BGR8 buffer */
int rt_save_bmp(const unsigned char *bgr, int width, int height, const char *path);

/* replaces: main + parseArgs (src/rt.cpp:7-49) with the same flags (-out, -scene, -threads; all
 * three or none) plus optional -w, -h, -depth, -envmap, -resources, -device, -gpus (the frame's rows
 * interleaved over that many GPUs, starting at -device: rtx_group_*).  Returns the exit code. */
int rt_main(int argc, char **argv);

const char *rt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
