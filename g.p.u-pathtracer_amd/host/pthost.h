/*
 * pthost.h — host-side scene preparation ("libpthost.so", CPU only, no HIP).
 *
 * In a real drop-in the reference's own host code stays (utilfun.cpp loadIndexedTris →
 * SceneMesh → BVH/SplitBVHBuilder → CudaBVH::createCompact, BasicScene.cpp:281-294) and
 * hands its three arrays to pt_upload_bvh.  The reference does not travel to the GPU box,
 * so this library re-creates that producer clean-room: mesh ingest (OBJ v/f records or the
 * committed .ptmesh fixtures), a SAH BVH builder (binned object splits + optional
 * Stich-et-al. spatial splits), and a flatten step that emits EXACTLY the reference's
 * "Compact" layout (GpuPathTracer/CudaBVH.cpp:121-270):
 *   nodes : 4 x vec4 per inner node  [c0.lo.x c0.hi.x c0.lo.y c0.hi.y]
 *                                    [c1.lo.x c1.hi.x c1.lo.y c1.hi.y]
 *                                    [c0.lo.z c0.hi.z c1.lo.z c1.hi.z]
 *                                    [link0 link1 0 0]   (int bits)
 *           link >= 0: BYTE offset of the child node; link < 0: ~(index of the leaf's
 *           first vec4 in the triangle array).  Root at offset 0.
 *   tris  : per leaf, 3 vec4 (v0,v1,v2 as xyz,0) per triangle reference, then one
 *           terminator vec4 of 0x80000000 words        (CudaBVH.cpp:184-205)
 *   index : int array parallel to tris: [triId,0,0] per reference, 0 per terminator
 *           (CudaBVH.cpp:201-212)
 * Differences from the reference producer, both fixes of crashes:
 * a root that is a leaf is wrapped in an inner node whose second child is an empty,
 * never-hit leaf (CudaBVH.cpp:141 asserts instead), and multi-object OBJ files load
 * (utilfun.cpp:474 asserts exactly one shape).
 */
#ifndef PTHOST_H
#define PTHOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_mesh pth_mesh;
typedef struct pth_bvh pth_bvh;

typedef struct pth_build_params {
    int32_t max_leaf_size;   /* Platform::m_maxLeafSize (bvh_util.hpp), default 0x7FFFFFF */
    int32_t min_leaf_size;   /* Platform::m_minLeafSize, default 1                        */
    int32_t max_depth;       /* SplitBVHBuilder MaxDepth = 64 (SplitBVHBuilder.hpp:15)     */
    int32_t n_bins;          /* object-split SAH bins per axis (0 = full sweep)             */
    float sah_node_cost;     /* Platform::m_SAHNodeCost = 1                                */
    float sah_tri_cost;      /* Platform::m_SAHTriangleCost = 1                            */
    float split_alpha;       /* BuildParams::splitAlpha = 1e-5; < 0 disables spatial splits */
    int32_t n_spatial_bins;  /* SplitBVHBuilder NumSpatialBins = 32                         */
    int32_t optimize_passes; /* EXTENSION (0 = off, the reference's tree as built): passes of insertion-based optimisation
                                over the finished hierarchy (Bittner et al. 2013): every node is re-inserted where the tree's
                                surface-area cost grows least; closest hits unchanged                                    */
} pth_build_params;

typedef struct pth_bvh_stats {
    uint64_t n_inner, n_leaves, n_tri_refs;
    uint32_t max_depth;      /* edges on the longest root-to-leaf path                      */
    float sah_cost;
    double build_ms;
    float opt_cost_before, opt_cost_after;   /* sum of inner-node areas / root area, before / after optimize_passes (0 if off) */
} pth_bvh_stats;

const char* pth_last_error(void);
void pth_default_build_params(pth_build_params* p);

/* One row of a per-triangle material table — same 32-byte layout as pt_material (include/ptmi.h),
 * so the array goes to pt_upload_tri_materials as it is. */
typedef struct pth_material {
    float col[3];
    float emi[3];
    int32_t mat;         /* 0 DIFF, 1 METAL, 2 SPEC, 3 REFR (Mat, CommomStructs.hpp:16) */
    float phong_expo;
} pth_material;

/* ---- meshes ---- */
pth_mesh* pth_mesh_create(const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris);
/* `v` / `f` records, polygons fan-triangulated, any number of o/g groups; `mtllib` + `usemtl`
 * give every triangle a material row (Kd → col, Ke → emi, illum 3/5/4,6,7 → METAL/SPEC/REFR) */
pth_mesh* pth_mesh_load_obj(const char* path);
pth_mesh* pth_mesh_load_ptmesh(const char* path);   /* the committed .ptmesh fixtures under assets/ */
int pth_mesh_save_ptmesh(const pth_mesh* m, const char* path);   /* PTMESH2 when the mesh has materials */
/* per-triangle materials: n = 0 / NULL when the mesh has none */
size_t pth_mesh_n_materials(const pth_mesh* m);
const pth_material* pth_mesh_materials(const pth_mesh* m);
const int32_t* pth_mesh_tri_materials(const pth_mesh* m);
int pth_mesh_set_materials(pth_mesh* m, const pth_material* table, size_t n, const int32_t* tri_material);
/* append src transformed by the row-major 3x4 affine m (NULL = identity) */
int pth_mesh_append(pth_mesh* dst, const pth_mesh* src, const float* m3x4);
size_t pth_mesh_n_verts(const pth_mesh* m);
size_t pth_mesh_n_tris(const pth_mesh* m);
const float* pth_mesh_verts(const pth_mesh* m);
const int32_t* pth_mesh_tris(const pth_mesh* m);
void pth_mesh_bounds(const pth_mesh* m, float lo[3], float hi[3]);
void pth_mesh_free(pth_mesh* m);

/* ---- BVH build + flatten to the Compact layout ---- */
pth_bvh* pth_bvh_build(const pth_mesh* mesh, const pth_build_params* params);
const float* pth_bvh_nodes(const pth_bvh* b);        size_t pth_bvh_n_node_vec4(const pth_bvh* b);
const float* pth_bvh_tris(const pth_bvh* b);         size_t pth_bvh_n_tri_vec4(const pth_bvh* b);
const int32_t* pth_bvh_index(const pth_bvh* b);      size_t pth_bvh_n_index(const pth_bvh* b);
void pth_bvh_get_stats(const pth_bvh* b, pth_bvh_stats* out);
void pth_bvh_free(pth_bvh* b);

/* ---- image output + progressive-state checkpoints ----
 * The reference only copies dev_drawRes into a GL texture (BasicScene.cpp:424-432).  Rows are
 * stored bottom-up in the frame buffers (row 0 = bottom of the picture, as GL has it).
 *   pth_write_ppm / pth_write_png : the display words (0x00BBGGRR, rgbToUint cudaUtils.h:99-105),
 *                                   written top row first; PNG uses stored (uncompressed) deflate
 *   pth_write_pfm                 : the float accumulator (PFM "PF", little endian, bottom-up)
 *   pth_checkpoint_save / _load   : accumulator + the loop state needed to continue a progressive
 *                                   render bit-identically: next frame number and constantPdf */
int pth_write_ppm(const char* path, const uint32_t* rgba, int width, int height);
int pth_write_png(const char* path, const uint32_t* rgba, int width, int height);
int pth_write_pfm(const char* path, const float* accum, int width, int height);
typedef struct pth_checkpoint_info {
    int32_t width, height;
    uint64_t next_frame;      /* frameNumber of the next launch (BasicScene.cpp:397)  */
    uint64_t constant_pdf;    /* samples folded so far (BasicScene.cpp:399)          */
    uint64_t scene_tag;       /* caller's fingerprint of scene + parameters          */
} pth_checkpoint_info;
int pth_checkpoint_save(const char* path, const pth_checkpoint_info* info, const float* accum);
/* reads the header; if accum is non-NULL also the pixels (width*height*3 floats, caller's buffer) */
int pth_checkpoint_load(const char* path, pth_checkpoint_info* info, float* accum);

/* uf::hash (utilfun.cpp:380-389): the per-frame seed the App loop passes down. */
uint64_t pth_frame_hash(uint64_t frame);

#ifdef __cplusplus
}
#endif
#endif
