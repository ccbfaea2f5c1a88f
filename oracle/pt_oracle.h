/* pt_oracle.h — declarations of the CPU oracle (test infrastructure; see pt_oracle.c). */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/ptmi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_STACK_SIZE 64 /* STACK_SIZE, GpuPathTracer/cudaUtils.h:22 */

typedef pt_counters orc_counters;

uint64_t orc_wang64(uint64_t key);
float orc_rng_draw(uint64_t frame, uint64_t pixel, uint32_t draw);
void orc_sincos2pi(float u, float* c_out, float* s_out);
float orc_pow01(float x, float y);
void orc_camera_ray(const pt_camera* cam, int px, int py, int w, int h, float u0, float u1,
                    float* o_out, float* d_out);
void orc_accumulate(float* acc3, uint32_t* rgba, const float* sample3, uint64_t N);

void orc_trace_rays_bvh(const float* nodes, const float* tris, const int32_t* tidx,
                        const float* rays, size_t n_rays, int cull,
                        float* t_out, int32_t* tri_out, float* n_out, orc_counters* cnt);
void orc_trace_rays_brute(const float* verts, const int32_t* tri_vidx, size_t n_tris,
                          const float* rays, size_t n_rays, int cull,
                          float* t_out, int32_t* tri_out, float* n_out);
int orc_render(float* accum, uint32_t* rgba,
               const float* nodes, const float* tris, const int32_t* tidx,
               const pt_sphere* sph, size_t n_sph,
               const pt_camera* cam, const pt_params* P, uint32_t spp, orc_counters* cnt);
int orc_render_mat(float* accum, uint32_t* rgba,
                   const float* nodes, const float* tris, const int32_t* tidx,
                   const pt_sphere* sph, size_t n_sph,
                   const pt_material* mtab, const int32_t* tri_mat,
                   const pt_camera* cam, const pt_params* P, uint32_t spp, orc_counters* cnt);
/* PT_FLAG_NEE over emissive triangles: the light list ((v0, emi.r) (e1, emi.g) (e2, emi.b) per triangle whose material row
 * emits, ascending id) used by the following orc_render_mat calls; n = 0 clears it.  Not copied: keep the array alive. */
void orc_set_tri_lights(const float* lights12, size_t n);
int orc_sample_pixels(const float* nodes, const float* tris, const int32_t* tidx,
                      const float* verts, const int32_t* tri_vidx, size_t n_tris,
                      const pt_sphere* sph, size_t n_sph, const pt_camera* cam, const pt_params* P, uint32_t spp,
                      const int32_t* pixels_xy, size_t n, float* out_col, float* out_seg);
void orc_primary_rays(const pt_camera* cam, int W, int H, uint64_t frame, int jitter, float* rays8);

#ifdef __cplusplus
}
#endif
#endif
