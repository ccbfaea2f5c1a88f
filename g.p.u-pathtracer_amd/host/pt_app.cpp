// pt_app.cpp — headless "App render loop": what BasicScene::run() (GpuPathTracer/
// BasicScene.cpp:368-477) does per frame, minus window/GL/ImGui, over the C ABI.
//
//   per frame:  sync (BasicScene.cpp:395) → frame seed (:397) → constantPdf logic (:399)
//               → launchKernel (:404) → [display copy :424-432 → here: optional download]
//
// Usage: pt_app --mesh assets/cornell.ptmesh [--width 1280 --height 720 --frames 16 --spp 1
//               --depth 4 --mat 0..3 --no-spheres --no-materials --bk r g b --device 0
//               --out image.ppm|.png|.pfm  --checkpoint state.ckpt [--checkpoint-every N]
//               --resume state.ckpt --device-build --fix-estimators --nee --gpus N --tile ROWS]
// --gpus N splits the framebuffer over N contexts, one per GPU (devices device, device+1, ... modulo the number
// present, so N > 1 also runs on a one-GPU box): stripes of --tile rows (default 8) are dealt round-robin
// (pt_params.part_*), every context holds the whole scene and renders only its stripes of every frame — the random
// streams are keyed by the global pixel, so the merged image equals the one-GPU image bit for bit — and the stripes
// are gathered on the host when an image or a checkpoint is written (the reference is single-GPU).
// --device-build builds the BVH on the GPU (pt_build_bvh) instead of the host SAH/SBVH builder;
// --fix-estimators sets the PT_FLAG_* corrected-estimator switches (face-forward, cosine DIFF,
// glass fix, Russian roulette).
// --frames counts samples per pixel in total; --spp of them are folded per pt_render call.
// A mesh that carries materials (OBJ usemtl + .mtl, PTMESH2) is shaded with them
// (pt_upload_tri_materials) unless --no-materials.  --resume continues a checkpointed
// progressive render: the result is bit-identical to one uninterrupted run.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptmi.h"
#include "pthost.h"

static int die(const char* what, const char* msg) {
    std::fprintf(stderr, "pt_app: %s: %s\n", what, msg ? msg : "");
    return 1;
}

int main(int argc, char** argv) {
    std::string mesh_path, out_path, ckpt_path, resume_path;
    int W = 1280, H = 720, frames = 16, depth = 4, mat = PT_MAT_DIFF, device = 0, spp = 1, ckpt_every = 0, gpus = 1, tile = 8;
    bool spheres = true, use_materials = true, device_build = false, fix_estimators = false, nee = false;
    float bk[3] = {1.f, 1.f, 1.f};
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", name); std::exit(2); }
            return argv[++i];
        };
        if (a == "--mesh") mesh_path = next("--mesh");
        else if (a == "--out") out_path = next("--out");
        else if (a == "--width") W = std::atoi(next("--width"));
        else if (a == "--height") H = std::atoi(next("--height"));
        else if (a == "--frames") frames = std::atoi(next("--frames"));
        else if (a == "--depth") depth = std::atoi(next("--depth"));
        else if (a == "--mat") mat = std::atoi(next("--mat"));
        else if (a == "--device") device = std::atoi(next("--device"));
        else if (a == "--no-spheres") spheres = false;
        else if (a == "--no-materials") use_materials = false;
        else if (a == "--device-build") device_build = true;
        else if (a == "--fix-estimators") fix_estimators = true;
        else if (a == "--nee") nee = true;   // next-event estimation (implies the cosine-weighted DIFF lobe and keeps a path's light on a miss)
        else if (a == "--spp") spp = std::atoi(next("--spp"));
        else if (a == "--checkpoint") ckpt_path = next("--checkpoint");
        else if (a == "--checkpoint-every") ckpt_every = std::atoi(next("--checkpoint-every"));
        else if (a == "--resume") resume_path = next("--resume");
        else if (a == "--gpus") gpus = std::atoi(next("--gpus"));
        else if (a == "--tile") tile = std::atoi(next("--tile"));
        else if (a == "--bk") { for (int k = 0; k < 3; k++) bk[k] = (float)std::atof(next("--bk")); }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (mesh_path.empty()) return die("usage", "--mesh <file.obj|file.ptmesh> is required");
    if (spp < 1 || frames < 0 || W < 2 || H < 2) return die("usage", "--spp >= 1, --frames >= 0, --width/--height >= 2");
    if (gpus < 1 || gpus > 64 || tile < 8 || tile % 8) return die("usage", "--gpus 1..64, --tile a positive multiple of 8");

    const bool is_ptmesh = mesh_path.size() > 7 && mesh_path.substr(mesh_path.size() - 7) == ".ptmesh";
    pth_mesh* mesh = is_ptmesh ? pth_mesh_load_ptmesh(mesh_path.c_str()) : pth_mesh_load_obj(mesh_path.c_str());
    if (!mesh) return die("mesh", pth_last_error());
    // one context per GPU of the tile split (ctxs[0] = `ctx` is where the merged frame is assembled from)
    const int n_dev = pt_device_count();
    if (n_dev < 1) return die("pt_device_count", pt_last_error(nullptr));
    std::vector<pt_ctx*> ctxs((size_t)gpus, nullptr);
    for (int g = 0; g < gpus; g++)
        if (pt_create((device + g) % n_dev, &ctxs[(size_t)g]) != PT_OK) return die("pt_create", pt_last_error(nullptr));
    pt_ctx* ctx = ctxs[0];
    if (gpus > 1) std::printf("tile split over %d context(s) on %d device(s), stripes of %d rows\n", gpus, std::min(gpus, n_dev), tile);
    pth_bvh* bvh = nullptr;
    if (!device_build) {
        bvh = pth_bvh_build(mesh, nullptr);
        if (!bvh) return die("bvh", pth_last_error());
        pth_bvh_stats st;
        pth_bvh_get_stats(bvh, &st);
        std::printf("mesh %zu tris; bvh %llu inner, %llu leaves, depth %u, built in %.1f ms\n", pth_mesh_n_tris(mesh),
                    (unsigned long long)st.n_inner, (unsigned long long)st.n_leaves, st.max_depth, st.build_ms);
    }
    const bool has_materials = use_materials && pth_mesh_n_materials(mesh) > 0;
    for (pt_ctx* cx : ctxs) {   // the scene is replicated: every GPU needs all of it for its own pixels
        if (device_build) {
            if (pt_build_bvh(cx, pth_mesh_verts(mesh), pth_mesh_n_verts(mesh), pth_mesh_tris(mesh), pth_mesh_n_tris(mesh)) != PT_OK)
                return die("pt_build_bvh", pt_last_error(cx));
            float bms = 0.f;
            pt_last_build_ms(cx, &bms);
            if (cx == ctx) std::printf("mesh %zu tris; bvh built on the device in %.2f ms\n", pth_mesh_n_tris(mesh), bms);
        } else if (pt_upload_bvh(cx, pth_bvh_nodes(bvh), pth_bvh_n_node_vec4(bvh), pth_bvh_tris(bvh), pth_bvh_n_tri_vec4(bvh),
                                 pth_bvh_index(bvh), pth_bvh_n_index(bvh)) != PT_OK) {
            return die("pt_upload_bvh", pt_last_error(cx));
        }
        if (has_materials) {
            static_assert(sizeof(pth_material) == sizeof(pt_material), "one layout");
            if (pt_upload_tri_materials(cx, (const pt_material*)pth_mesh_materials(mesh), pth_mesh_n_materials(mesh),
                                        pth_mesh_tri_materials(mesh), pth_mesh_n_tris(mesh)) != PT_OK)
                return die("pt_upload_tri_materials", pt_last_error(cx));
        }
    }
    if (has_materials) std::printf("%zu materials from the mesh file\n", pth_mesh_n_materials(mesh));

    // the reference's sphere room, BasicScene.cpp:181-202
    std::vector<pt_sphere> sph;
    if (spheres) {
        const float rad = 600.f, px = 20.f, py = 15.f;
        const float a[3] = {197.f / 255.f, 153.f / 255.f, 92.f / 255.f};
        auto add = [&](float x, float y, float z, float r, const float* e, float cr, float cg, float cb, int m) {
            pt_sphere s{{x, y, z, r}, {e[0], e[1], e[2]}, {cr, cg, cb}, m};
            sph.push_back(s);
        };
        const float red[3] = {165.f / 255.f, 15.f / 255.f, 0.f}, green[3] = {30.f / 255.f, 76.f / 255.f, 14.f / 255.f};
        const float cyan[3] = {0.f, 1.f, 0.8f}, zero[3] = {0, 0, 0}, lamp[3] = {0.f, 1.f, 1.f};
        add(0, -py - rad, -20, rad, a, .5f, .5f, .5f, PT_MAT_DIFF);
        add(0, py + rad, -20, rad, a, .1f, .3f, .4f, PT_MAT_DIFF);
        add(px + rad, 0, -20, rad, red, red[0], red[1], red[2], PT_MAT_DIFF);
        add(-px - rad, 0, -20, rad, green, green[0], green[1], green[2], PT_MAT_DIFF);
        add(0, 0, -rad * 1.5f - 20, rad, a, 1, 1, 1, PT_MAT_DIFF);
        add(0, 0, rad * 1.5f + 20, rad, cyan, .5f, .5f, .5f, PT_MAT_DIFF);
        add(13, -8, -35, 6, zero, 1, 1, 1, PT_MAT_SPEC);
        add(10, -15, -68, 10, lamp, 1, 1, 1, PT_MAT_DIFF);
        for (pt_ctx* cx : ctxs)
            if (pt_upload_spheres(cx, sph.data(), sph.size()) != PT_OK) return die("pt_upload_spheres", pt_last_error(cx));
    }

    // kernel defaults (BasicScene.cpp:220-236) and camera (:239-259)
    pt_camera cam{};
    cam.front[2] = -1.f; cam.right[0] = 1.f; cam.up[1] = 1.f;
    cam.dist = (float)(H / 60);
    cam.aspect = W * 1.0f / H;
    cam.fov = 1.0f;
    pt_params p{};
    p.width = W; p.height = H; p.depth = (uint32_t)depth; p.cull_backfaces = 1;
    p.tri_mat = mat;
    p.tri_col[0] = 246.f / 256.f; p.tri_col[1] = 246.f / 255.f; p.tri_col[2] = 70.f / 255.f;
    p.bk_color[0] = bk[0]; p.bk_color[1] = bk[1]; p.bk_color[2] = bk[2];
    p.air_ior = 1.0f; p.glass_ior = 1.4f; p.phong_expo = 30.f;
    p.flags = PT_FLAG_WRITE_RGBA;
    if (fix_estimators) p.flags |= PT_FLAG_FACE_FORWARD | PT_FLAG_COSINE_DIFF | PT_FLAG_GLASS_FIX | PT_FLAG_RUSSIAN_ROULETTE;
    if (nee) p.flags |= PT_FLAG_COSINE_DIFF | PT_FLAG_NEE | PT_FLAG_MISS_KEEPS_PATH;
    p.part_count = gpus; p.part_rows = tile;

    // every context owns full-frame buffers (accum/rgba are always addressed by the global pixel); it touches only
    // its own stripes of them
    std::vector<void*> accums((size_t)gpus, nullptr), rgbas((size_t)gpus, nullptr);
    for (int g = 0; g < gpus; g++) {
        if (pt_malloc(ctxs[(size_t)g], (size_t)W * H * 12, &accums[(size_t)g]) != PT_OK || pt_malloc(ctxs[(size_t)g], (size_t)W * H * 4, &rgbas[(size_t)g]) != PT_OK)
            return die("pt_malloc", pt_last_error(ctxs[(size_t)g]));
        pt_memset(ctxs[(size_t)g], accums[(size_t)g], 0, (size_t)W * H * 12);
    }
    void *accum = accums[0], *rgba = rgbas[0];
    // gather: rows of stripe s belong to context s % gpus; `elem` bytes per pixel
    std::vector<unsigned char> part_buf;
    auto gather = [&](std::vector<void*>& dev, void* host, size_t elem) -> int {
        if (pt_download(ctxs[0], host, dev[0], (size_t)W * H * elem) != PT_OK) return die("pt_download", pt_last_error(ctxs[0]));
        for (int g = 1; g < gpus; g++) {
            part_buf.resize((size_t)W * H * elem);
            if (pt_download(ctxs[(size_t)g], part_buf.data(), dev[(size_t)g], part_buf.size()) != PT_OK) return die("pt_download", pt_last_error(ctxs[(size_t)g]));
            for (int s0 = g * tile; s0 < H; s0 += gpus * tile) {
                const size_t off = (size_t)s0 * W * elem, len = (size_t)std::min(tile, H - s0) * W * elem;
                std::memcpy((unsigned char*)host + off, part_buf.data() + off, len);
            }
        }
        return 0;
    };

    // what a checkpoint must agree on to be continued: geometry size, image size, the scalar parameters
    uint64_t tag = pth_frame_hash((uint64_t)pth_mesh_n_tris(mesh) * 1315423911ull + (uint64_t)W * 65537u + (uint64_t)H);
    tag = pth_frame_hash(tag ^ ((uint64_t)depth << 32 | (uint64_t)mat << 8 | (fix_estimators ? 4u : 0u) | (nee ? 8u : 0u) | (spheres ? 2u : 0u) | (has_materials ? 1u : 0u)));

    uint64_t frameNumber = 0, constantPdf = 0;   // constantPdf = samples folded so far
    std::vector<float> host_acc;
    if (!resume_path.empty()) {
        pth_checkpoint_info ci{};
        host_acc.resize((size_t)W * H * 3);
        if (pth_checkpoint_load(resume_path.c_str(), &ci, nullptr) != 0) return die("resume", pth_last_error());
        if (ci.width != W || ci.height != H || ci.scene_tag != tag) return die("resume", "checkpoint belongs to another scene / image size / parameters");
        if (pth_checkpoint_load(resume_path.c_str(), &ci, host_acc.data()) != 0) return die("resume", pth_last_error());
        for (int g = 0; g < gpus; g++)   // every context continues its own stripes
            if (pt_upload(ctxs[(size_t)g], accums[(size_t)g], host_acc.data(), host_acc.size() * 4) != PT_OK) return die("pt_upload", pt_last_error(ctxs[(size_t)g]));
        frameNumber = ci.next_frame;
        constantPdf = ci.constant_pdf;
        std::printf("resumed %s: %llu samples per pixel done\n", resume_path.c_str(), (unsigned long long)constantPdf);
    }
    auto save_checkpoint = [&]() -> int {
        host_acc.resize((size_t)W * H * 3);
        if (int rc = gather(accums, host_acc.data(), 12)) return rc;
        pth_checkpoint_info ci{W, H, frameNumber, constantPdf, tag};
        if (pth_checkpoint_save(ckpt_path.c_str(), &ci, host_acc.data()) != 0) return die("checkpoint", pth_last_error());
        return 0;
    };

    const uint64_t end_frame = frameNumber + (uint64_t)frames;
    int calls = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (frameNumber < end_frame) {
        for (pt_ctx* cx : ctxs)
            if (pt_sync(cx) != PT_OK) return die("pt_sync", pt_last_error(cx));     // :395
        const uint32_t n = (uint32_t)std::min<uint64_t>((uint64_t)spp, end_frame - frameNumber);
        p.frame = frameNumber;                                                      // :397
        p.sample_index = constantPdf + 1;                                           // :399 (1 on the first frame: overwrite)
        for (int g = 0; g < gpus; g++) {                                            // :404, once per GPU: asynchronous, so the GPUs run side by side
            p.part_index = g;
            if (pt_render(ctxs[(size_t)g], (float*)accums[(size_t)g], (uint32_t*)rgbas[(size_t)g], &cam, &p, n) != PT_OK)
                return die("pt_render", pt_last_error(ctxs[(size_t)g]));
        }
        frameNumber += n;
        constantPdf += n;
        calls++;
        if (!ckpt_path.empty() && ckpt_every > 0 && calls % ckpt_every == 0 && frameNumber < end_frame)
            if (int rc = save_checkpoint()) return rc;
    }
    for (pt_ctx* cx : ctxs) pt_sync(cx);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const double rays = (double)W * H * depth * frames;
    std::printf("%d samples/pixel %dx%d depth %d in %d calls: %.2f ms/sample, <= %.1f Mrays/s (closed scene bound)\n", frames, W, H,
                depth, calls, frames ? ms / frames : 0.0, ms > 0 ? rays / ms / 1e3 : 0.0);
    if (!ckpt_path.empty())
        if (int rc = save_checkpoint()) return rc;

    if (!out_path.empty()) {
        const std::string ext = out_path.size() > 4 ? out_path.substr(out_path.size() - 4) : std::string();
        int rc;
        if (ext == ".pfm") {
            host_acc.resize((size_t)W * H * 3);
            if (int grc = gather(accums, host_acc.data(), 12)) return grc;
            rc = pth_write_pfm(out_path.c_str(), host_acc.data(), W, H);
        } else {
            std::vector<uint32_t> img((size_t)W * H);
            if (frames == 0) {  // nothing rendered in this run (e.g. --resume only to convert): pack the accumulator here
                host_acc.resize((size_t)W * H * 3);
                if (int grc = gather(accums, host_acc.data(), 12)) return grc;
                for (size_t i = 0; i < img.size(); i++) {   // rgbToUint, cudaUtils.h:99-105
                    auto q = [&](float v) { v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v); return (uint32_t)(v * 255.f); };
                    img[i] = q(host_acc[3 * i]) | (q(host_acc[3 * i + 1]) << 8) | (q(host_acc[3 * i + 2]) << 16);
                }
            } else if (int grc = gather(rgbas, img.data(), 4)) {
                return grc;
            }
            rc = ext == ".png" ? pth_write_png(out_path.c_str(), img.data(), W, H) : pth_write_ppm(out_path.c_str(), img.data(), W, H);
        }
        if (rc != 0) return die("write", pth_last_error());
    }
    (void)accum; (void)rgba;
    for (int g = 0; g < gpus; g++) {
        pt_free(ctxs[(size_t)g], accums[(size_t)g]);
        pt_free(ctxs[(size_t)g], rgbas[(size_t)g]);
        pt_destroy(ctxs[(size_t)g]);
    }
    if (bvh) pth_bvh_free(bvh);
    pth_mesh_free(mesh);
    return 0;
}
