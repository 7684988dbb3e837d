// Host driver: stands where the reference's main() does (raytracer.cpp:843-868).
//
//   rt_render <scene> [out.png] [-w W] [-h H] [-depth D] [-eps E] [-tile ROWS] [-bvh] [-exact] [-stats]
//
// With no options it does what `./raytracer <scene> [out.png]` does: 1000 x 1000, eps 0.05,
// depth 6, output.png (raytracer.cpp:847-862).  Parsing goes through the host loader
// (include/rt_host.h), the trace loop through the C ABI (include/rt_abi.h), the PNG through
// rth_write_png.  There is no CPU rendering path here: without a GPU the program fails.
#include "rt_abi.h"
#include "rt_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <scene> [out.png] [-w W] [-h H] [-depth D] [-eps E] [-tile ROWS] [-bvh] [-exact] [-stats]\n",
                argv[0]);
        return 2;
    }
    std::string scene = argv[1], out = "output.png";
    int width = 1000, height = 1000, depth = -1, tile = 0;
    float eps = 0.f;
    bool bvh = false, exact = false, stats = false;
    int pos = 0;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "-w") width = atoi(need("-w"));
        else if (a == "-h") height = atoi(need("-h"));
        else if (a == "-depth") depth = atoi(need("-depth"));
        else if (a == "-eps") eps = (float)atof(need("-eps"));
        else if (a == "-tile") tile = atoi(need("-tile"));
        else if (a == "-bvh") bvh = true;
        else if (a == "-exact") exact = true;
        else if (a == "-stats") stats = true;
        else if (pos++ == 0) out = a;
        else { fprintf(stderr, "unexpected argument %s\n", a.c_str()); return 2; }
    }
    printf("Ray Tracer!\n");
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    auto t_load = now();
    rth_scene* sc = nullptr;
    if (rth_load_scene(&sc, scene.c_str(), width, height, depth, eps)) {
        fprintf(stderr, "%s\n", rth_last_error());
        return 1;
    }
    const double load_s = since(t_load);
    auto t_upload = now();
    rt_ctx* ctx = nullptr;
    if (rt_upload_scene(&ctx, rth_desc(sc))) {
        fprintf(stderr, "%s\n", rt_last_error());
        return 1;
    }
    const double upload_s = since(t_upload);
    if (exact && rt_set_option(ctx, RT_OPT_EXACT_ONLY, 1)) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    auto t_bvh = now();
    if (bvh && rt_set_option(ctx, RT_OPT_USE_BVH, 1)) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    const double bvh_s = since(t_bvh);   // the build is synchronous
    printf("Rendering Scene...\n");
    auto t0 = std::chrono::steady_clock::now();
    int rc = tile > 0 ? rt_render_tiles(ctx, 0, tile, 1, nullptr) : rt_render_tile(ctx, 0, height, nullptr);
    std::vector<uint8_t> image((size_t)width * height * 4);
    if (!rc) rc = rt_download(ctx, image.data(), image.size());
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    printf("Scene render completed.\n");
    if (stats) {
        rt_stats st;
        if (!rt_get_stats(ctx, &st)) {
            unsigned long long rays = st.rays_primary + st.rays_reflect + st.rays_shadow;
            printf("{\"render_s\": %.6f, \"rays_primary\": %llu, \"rays_reflect\": %llu, \"rays_shadow\": %llu, \"hits\": %llu, "
                   "\"Mrays_per_s\": %.3f, \"triangles\": %d, \"spheres\": %d}\n",
                   secs, (unsigned long long)st.rays_primary, (unsigned long long)st.rays_reflect,
                   (unsigned long long)st.rays_shadow, (unsigned long long)st.hits, rays / secs / 1e6, st.n_triangles,
                   st.n_spheres);
        }
    }
    printf("Writing to file...\n");
    auto t_png = now();
    if (rth_write_png(out.c_str(), image.data(), width, height)) {
        fprintf(stderr, "cannot write %s\n", out.c_str());
        return 1;
    }
    printf("Image saved as %s.\n", out.c_str());
    if (stats)   // render_s above covers the trace and the download of the frame
        printf("{\"load_s\": %.6f, \"upload_s\": %.6f, \"bvh_build_s\": %.6f, \"png_s\": %.6f}\n", load_s, upload_s, bvh_s,
               since(t_png));
    rt_free(ctx);
    rth_free_scene(sc);
    return 0;
}
