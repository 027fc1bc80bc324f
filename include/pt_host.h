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
 * Polygons are fan-triangulated; faces with no material get Kd = 0.6 grey, Ke = 0 (the
 * reference would index materials[-1], main.cpp:49 -- undefined there, defined here).      */
int  pth_load_obj(const char *obj_path, const char *mtl_dir, pth_scene *out, char *err, size_t err_len);
void pth_free_scene(pth_scene *s);

/* bgra: w*h*4 bytes as read by pt_film_read_bgra8 -> binary PPM (P6, RGB).                 */
int pth_write_ppm_bgra8(const char *path, const uint8_t *bgra, uint32_t w, uint32_t h);
/* rgb: w*h*3 floats (pt_film_read_f32) -> PFM (little endian, bottom-up as the format wants) */
int pth_write_pfm(const char *path, const float *rgb, uint32_t w, uint32_t h);

/* Writes <path> (OBJ) and <path minus .obj>.mtl: n_tris random triangles, frozen recipe in
 * BASELINE.md section 4 / DESIGN.md section 9 (PCG stream seeded with `seed`).                     */
int pth_write_soup_obj(const char *obj_path, uint32_t n_tris, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
