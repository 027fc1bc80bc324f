/*
 * pt_host.h -- C-ABI of the host-side pieces of the path (libpt_host.so, plain C++20, no GPU):
 * scene ingest and image output, i.e. what main() does around the dispatch.
 *
 *   pth_load_obj        replaces loadFromFile (main.cpp:28-58): tinyobjloader + the per-index
 *                       de-indexing with Y negated (main.cpp:40-45) + per-face {Kd,Ke} (47-56)
 *   pth_write_*         replace the copy to the swapchain + present (main.cpp:661-679)
 *   pth_write_soup_obj  generator of BASELINE.json config 5 (1M-triangle random soup), not in
 *                       the reference
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene {
    float    *vertices; /* 3*n_verts: x, -y, z per mesh index (main.cpp:41-43)             */
    uint32_t  n_verts;
    uint32_t *indices;  /* 3*n_tris: 0,1,2,... (main.cpp:45)                               */
    uint32_t  n_tris;
    float    *faces;    /* 6*n_tris: Kd.rgb, Ke.rgb of the face's material (main.cpp:47-56) */
} pth_scene;

/* Returns 0 on success; on failure returns nonzero and writes a message into err (the
 * reference throws std::runtime_error(warn + err), main.cpp:35).  mtl_dir may be NULL = the
 * OBJ's directory (the reference passes "../assets", main.cpp:34).
 * Polygons are fan-triangulated; a declared material starts Kd = Ke = 0 and missing components of a Kd/Ke line
 * stay 0 (tinyobjloader's InitMaterial / parseReal3); faces with NO material get Kd = 0.6 grey, Ke = 0 (the
 * reference would index materials[-1], main.cpp:49 -- undefined there, defined here).      */
int  pth_load_obj(const char *obj_path, const char *mtl_dir, pth_scene *out, char *err, size_t err_len);
/* flags: PTH_QUAD_SHORTER_DIAGONAL cuts 4-gons along their shorter diagonal -- (0,1,2)(0,2,3) if |v0v2|^2 < |v1v3|^2,
 * else (0,1,3)(1,2,3) -- as newer tinyobjloader releases are understood to do (the reference's submodule is unpinned
 * and empty in the checkout, so which rule it was built with is unknown).  On the Cornell box both rules give the
 * same surfaces and, within the stated tolerance, the same image; only gl_PrimitiveID numbering and the last bits of
 * hit positions differ.  pth_load_obj = flags 0 = fan, which all fixtures use.                          */
enum { PTH_QUAD_SHORTER_DIAGONAL = 1u,
       PTH_SMALL_CHUNKS = 2u /* test hook: the text is read by up to 64 threads in chunks of ~256 bytes (the loader cuts big files into one
                                chunk per hardware thread; this makes short files cross chunk boundaries).  Same result. */ };
int  pth_load_obj_ex(const char *obj_path, const char *mtl_dir, uint32_t flags, pth_scene *out, char *err, size_t err_len);
void pth_free_scene(pth_scene *s);

/* bgra: w*h*4 bytes as read by pt_film_read_bgra8 -> binary PPM (P6, RGB).                 */
int pth_write_ppm_bgra8(const char *path, const uint8_t *bgra, uint32_t w, uint32_t h);
/* rgb: w*h*3 floats (pt_film_read_f32) -> PFM (little endian, bottom-up as the format wants) */
int pth_write_pfm(const char *path, const float *rgb, uint32_t w, uint32_t h);

/* Writes <path> (OBJ) and <path minus .obj>.mtl: n_tris random triangles, frozen recipe in
 * BASELINE.md section 4 / DESIGN.md section 9 (PCG stream seeded with `seed`).                     */
int pth_write_soup_obj(const char *obj_path, uint32_t n_tris, uint32_t seed);

/* The arrays pth_load_obj would return for that file, generated directly (free with pth_free_scene): scenes
 * larger than the Infinity Cache (bench.py --config c5x) without a gigabyte of OBJ text in between.      */
int pth_make_soup(uint32_t n_tris, uint32_t seed, pth_scene *out);

/* The "teapot in a stadium" stress scene (frozen recipe, host/image_io.cpp): a Cornell-sized room whose floor is a
 * floor_side x floor_side grid of jittered quads, two-triangle walls, the Cornell light, two boxes and a sphere of
 * sphere_seg x 2 sphere_seg quads -- primitive sizes over four orders of magnitude, what a Morton-median BVH handles
 * badly and ePreferFastTrace (main.cpp:419) is for.  Arrays as pth_load_obj returns them (free with pth_free_scene).
 * (384, 160) = 396 706 triangles: the tests' and the probe's scene; (640, 224) = 1.02 M.                            */
int pth_make_stadium(uint32_t floor_side, uint32_t sphere_seg, pth_scene *out);

#ifdef __cplusplus
}
#endif
#endif
