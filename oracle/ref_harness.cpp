// ref_harness.cpp — OUR driver around the REFERENCE's CPU tracer classes (test infrastructure).
//
// oracle/Makefile compiles the reference's own CpuRayTracer sources where they lie
// (/root/reference/CpuRayTracer/src/{objects,kdtree,scene,material,camera,texture}.cpp +
// lib/) and links them with this file into oracle/_ref/cpuraytracer_core.  The reference's
// renderer.cpp/main.cpp need a GLFW window and are not built; this file replays their
// logic headlessly:
//   hits   — Mesh::get_intersection (objects.cpp:151-158 → KDNode::hit kdtree.cpp:59-90 →
//            Triangle::intersect triangle.hpp:49-72) on a caller-supplied ray batch.
//            Pins the oracle's closest-hit distances against reference code run here.
//   render — the loop of Renderer::render (renderer.cpp:197-222) over the scene of
//            main.cpp:26-35 (5 spheres + one mesh); the timed CPU "reference" baseline.
//            Ray segments are counted through the reference's own Object interface
//            (a forwarding Object in front of the mesh sees every Scene::intersect).
#include <omp.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "camera.hpp"
#include "material.h"
#include "objects.h"
#include "scene.h"

namespace {

struct CountingObject : public Object {
    Object* inner;
    std::atomic<unsigned long long>* counter;
    CountingObject(Object* o, std::atomic<unsigned long long>* c) : inner(o), counter(c) {}
    ObjectIntersection get_intersection(const Ray& r) override {
        counter->fetch_add(1, std::memory_order_relaxed);
        return inner->get_intersection(r);
    }
};

int usage() {
    std::fprintf(stderr,
                 "usage: cpuraytracer_core hits <mesh.obj> <rays.f32> <n_rays> <out_t.f32>\n"
                 "       cpuraytracer_core render <mesh.obj> <W> <H> <spp> <px> <py> <pz> [out.f32]\n");
    return 2;
}

int run_hits(int argc, char** argv) {
    if (argc < 6) return usage();
    const long n = std::atol(argv[4]);
    std::vector<float> rays(8 * (size_t)n), out((size_t)n);
    FILE* f = std::fopen(argv[3], "rb");
    if (!f || std::fread(rays.data(), 4, rays.size(), f) != rays.size()) { std::fprintf(stderr, "bad ray file\n"); return 1; }
    std::fclose(f);
    Mesh mesh(glm::dvec3(0, 0, 0), argv[2], Material(DIFF, glm::dvec3(0.9, 0.9, 0.9)));
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < n; i++) {
        const float* r = &rays[8 * i];
        Ray ray(glm::dvec3(r[0], r[1], r[2]), glm::dvec3(r[4], r[5], r[6]));
        ObjectIntersection isct = mesh.get_intersection(ray);
        out[i] = isct.hit ? (float)isct.u : 3.402823466e+38f;
    }
    f = std::fopen(argv[5], "wb");
    if (!f) return 1;
    std::fwrite(out.data(), 4, out.size(), f);
    std::fclose(f);
    return 0;
}

int run_render(int argc, char** argv) {
    if (argc < 9) return usage();
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), samples = std::atoi(argv[5]);
    const glm::dvec3 p(std::atof(argv[6]), std::atof(argv[7]), std::atof(argv[8]));
    Camera camera(glm::dvec3(-2, -5, 2.5), glm::dvec3(0, 0, 0), W, H);  // main.cpp:26
    Scene scene;
    std::atomic<unsigned long long> segments(0);
    // main.cpp:30-35
    scene.add(new Sphere(glm::dvec3(0, 0, -1000), 1000, Material()));
    scene.add(new Sphere(glm::dvec3(-1004, 0, 0), 1000, Material(DIFF, glm::dvec3(0.85, 0.4, 0.4))));
    scene.add(new Sphere(glm::dvec3(1004, 0, 0), 1000, Material(DIFF, glm::dvec3(0.4, 0.4, 0.85))));
    scene.add(new Sphere(glm::dvec3(0, 1006, 0), 1000, Material()));
    scene.add(new Sphere(glm::dvec3(0, 0, 110), 100, Material(EMIT, glm::dvec3(1, 1, 1), glm::dvec3(2.2, 2.2, 2.2))));
    scene.add(new CountingObject(new Mesh(p, argv[2], Material(DIFF, glm::dvec3(0.9, 0.9, 0.9))), &segments));

    std::vector<float> img((size_t)W * H * 3);
    const double samples_recp = 1.0 / samples;
    auto t0 = std::chrono::steady_clock::now();
    // Renderer::render, renderer.cpp:197-222
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < H; y++) {
        unsigned short Xi[3] = {0, 0, (unsigned short)(y * y * y)};
        for (int x = 0; x < W; x++) {
            glm::dvec3 col = glm::dvec3();
            for (int a = 0; a < samples; a++) {
                Ray ray = camera.get_ray(x, y, a > 0, Xi);
                col = col + scene.trace_ray(ray, 0, Xi);
            }
            img[3 * ((size_t)y * W + x) + 0] = (float)(col.x * samples_recp);
            img[3 * ((size_t)y * W + x) + 1] = (float)(col.y * samples_recp);
            img[3 * ((size_t)y * W + x) + 2] = (float)(col.z * samples_recp);
        }
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double sum = 0;
    for (float v : img) sum += v;
    std::printf("{\"seconds\": %.6f, \"paths\": %lld, \"segments\": %llu, \"threads\": %d, \"mean\": %.6f}\n", sec,
                (long long)W * H * samples, segments.load(), omp_get_max_threads(), sum / img.size());
    if (argc > 9) {
        FILE* f = std::fopen(argv[9], "wb");
        if (f) { std::fwrite(img.data(), 4, img.size(), f); std::fclose(f); }
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) return usage();
    if (!std::strcmp(argv[1], "hits")) return run_hits(argc, argv);
    if (!std::strcmp(argv[1], "render")) return run_render(argc, argv);
    return usage();
}
