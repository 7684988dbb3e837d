// TEST INFRASTRUCTURE -- not part of the product path.
//
// Drives the UNMODIFIED reference renderer where it lies (the include path is
// given on the compiler command line by oracle/Makefile; nothing from the
// reference is copied into this repository).  The reference's `main` is
// renamed out of the way and this file supplies its own:
//
//   ref_render <scene> <W> <H> <depth> <out.rgba|-> [threads] [--dump desc.bin]
//
//  * parses with the reference's parseSceneFile (raytracer.cpp:705),
//  * sets the public Scene::maxDepth (raytracer.cpp:465),
//  * loops pixels itself, calling Camera::generateRay (raytracer.cpp:441),
//    Scene::traceRay (raytracer.cpp:514) and Film::writeToFilm
//    (raytracer.cpp:195) -- rows are interleaved over std::threads; traceRay
//    only reads the scene, so the output does not depend on the thread count,
//  * writes the RGBA8 image exactly as Film::writeFile (raytracer.cpp:205-230)
//    would hand it to lodepng: row 0 is the TOP scanline = film row H-1,
//  * optionally dumps the parsed scene as the flat "RTSD" description that
//    rt_upload_scene takes (this walk over the reference's public members is
//    the reference-side binding; the format is documented in include/rt_abi.h),
//  * prints one JSON line with the wall time of the trace loop only.
//
// The working directory is changed to the scene's directory first because the
// reference resolves `obj` file names against the cwd (raytracer.cpp:629).
#define main reference_main_unused
#include "raytracer.cpp"
#undef main

#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <unistd.h>

namespace {

void put(std::vector<unsigned char>& b, const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    b.insert(b.end(), c, c + n);
}
void put_i32(std::vector<unsigned char>& b, int32_t v) { put(b, &v, 4); }
void put_f32(std::vector<unsigned char>& b, float v) { put(b, &v, 4); }
void put_f64(std::vector<unsigned char>& b, double v) { put(b, &v, 8); }
void put_vec3(std::vector<unsigned char>& b, const vec3& v) {
    put_f64(b, v[0]); put_f64(b, v[1]); put_f64(b, v[2]);
}
void put_mat4(std::vector<unsigned char>& b, const mat4& m) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) put_f64(b, m[i][j]);
}
bool same_mat(const mat4& a, const mat4& b) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
    {
        double x = a[i][j], y = b[i][j];
        if (std::memcmp(&x, &y, 8) != 0) return false;
    }
    return true;
}
bool same_material(const Material& a, const Material& b) {
    return std::memcmp(&a.ka, &b.ka, sizeof(vec3)) == 0 && std::memcmp(&a.kd, &b.kd, sizeof(vec3)) == 0 &&
           std::memcmp(&a.ks, &b.ks, sizeof(vec3)) == 0 && std::memcmp(&a.kr, &b.kr, sizeof(vec3)) == 0 &&
           std::memcmp(&a.sp, &b.sp, sizeof(float)) == 0;
}

// Flatten the reference's object graph into the "RTSD" blob (include/rt_abi.h).
// Transforms and materials are de-duplicated by bit pattern; an object is
// compared with the previous object's entry first because the parser hands
// every face of a mesh the same transform and material.
bool dump_scene(Scene* s, int depth, const char* path) {
    std::vector<mat4> xf, xfi, xfit;
    std::vector<Material> mats;
    std::vector<int32_t> kind, mat_id, xf_id, interp;
    std::vector<double> geom, nrm;
    int last_x = -1, last_m = -1;
    for (size_t i = 0; i < s->objects.size(); i++) {
        Polygon* p = s->objects[i];
        int xi = -1;
        if (last_x >= 0 && same_mat(xf[last_x], p->transform)) xi = last_x;
        for (int k = (int)xf.size() - 1; xi < 0 && k >= 0; k--)
            if (same_mat(xf[k], p->transform)) xi = k;
        if (xi < 0) {
            xi = (int)xf.size();
            xf.push_back(p->transform); xfi.push_back(p->inverseTransform); xfit.push_back(p->inverseTranspose);
        }
        last_x = xi;
        int mi = -1;
        if (last_m >= 0 && same_material(mats[last_m], p->mat)) mi = last_m;
        for (int k = (int)mats.size() - 1; mi < 0 && k >= 0; k--)
            if (same_material(mats[k], p->mat)) mi = k;
        if (mi < 0) { mi = (int)mats.size(); mats.push_back(p->mat); }
        last_m = mi;
        xf_id.push_back(xi); mat_id.push_back(mi);
        if (Triangle* t = dynamic_cast<Triangle*>(p)) {
            kind.push_back(0);
            const vec3* v[3] = {&t->vertex1, &t->vertex2, &t->vertex3};
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) geom.push_back((*v[a])[c]);
            interp.push_back(t->needInterpolate ? 1 : 0);
            const vec3* n[3] = {&t->normal1, &t->normal2, &t->normal3};
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++)
                nrm.push_back(t->needInterpolate ? (*n[a])[c] : 0.0);
        } else {
            Sphere* sp = dynamic_cast<Sphere*>(p);
            kind.push_back(1);
            for (int c = 0; c < 3; c++) geom.push_back(sp->center[c]);
            geom.push_back((double)sp->radius);
            for (int c = 4; c < 9; c++) geom.push_back(0.0);
            interp.push_back(0);
            for (int c = 0; c < 9; c++) nrm.push_back(0.0);
        }
    }
    std::vector<unsigned char> b;
    put(b, "RTSD", 4);
    put_i32(b, 1);  // format version
    put_i32(b, s->width); put_i32(b, s->height); put_i32(b, depth);
    put_f32(b, s->eps); put_f32(b, 0.1f);  // shadow eps: raytracer.cpp:308, 390
    put_i32(b, (int32_t)s->objects.size()); put_i32(b, (int32_t)mats.size());
    put_i32(b, (int32_t)xf.size()); put_i32(b, (int32_t)s->lights.size());
    put_vec3(b, s->cam.eye); put_vec3(b, s->cam.ll); put_vec3(b, s->cam.lr);
    put_vec3(b, s->cam.ul); put_vec3(b, s->cam.ur);
    put_vec3(b, s->ambientLight.color);
    for (size_t i = 0; i < kind.size(); i++) put_i32(b, kind[i]);
    for (size_t i = 0; i < kind.size(); i++) put_i32(b, mat_id[i]);
    for (size_t i = 0; i < kind.size(); i++) put_i32(b, xf_id[i]);
    for (size_t i = 0; i < kind.size(); i++) put_i32(b, interp[i]);
    if (b.size() % 8) put_i32(b, 0);
    put(b, geom.data(), geom.size() * 8);
    put(b, nrm.data(), nrm.size() * 8);
    for (size_t i = 0; i < mats.size(); i++) {
        put_vec3(b, mats[i].ka); put_vec3(b, mats[i].kd); put_vec3(b, mats[i].ks); put_vec3(b, mats[i].kr);
        put_f64(b, (double)mats[i].sp);
    }
    for (size_t i = 0; i < xf.size(); i++) { put_mat4(b, xf[i]); put_mat4(b, xfi[i]); put_mat4(b, xfit[i]); }
    for (size_t i = 0; i < s->lights.size(); i++) {
        Light* l = s->lights[i];
        put_vec3(b, l->color); put_vec3(b, l->pos);
        put_i32(b, l->falloff); put_i32(b, l->pointLight ? 1 : 0);
    }
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    fwrite(b.data(), 1, b.size(), f);
    fclose(f);
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s scene W H depth out.rgba|- [threads] [--dump desc.bin] [--no-render]\n", argv[0]);
        return 2;
    }
    std::string scenePath = argv[1];
    int W = atoi(argv[2]), H = atoi(argv[3]), depth = atoi(argv[4]);
    std::string outPath = argv[5];
    int threads = 1;
    std::string dumpPath;
    bool noRender = false;
    for (int i = 6; i < argc; i++) {
        if (!strcmp(argv[i], "--dump") && i + 1 < argc) dumpPath = argv[++i];
        else if (!strcmp(argv[i], "--no-render")) noRender = true;
        else threads = atoi(argv[i]);
    }
    if (threads < 1) threads = 1;
    // absolute output paths before chdir
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return 2;
    auto absolutize = [&](std::string p) {
        if (p.empty() || p == "-" || p[0] == '/') return p;
        return std::string(cwd) + "/" + p;
    };
    outPath = absolutize(outPath);
    dumpPath = absolutize(dumpPath);
    size_t slash = scenePath.find_last_of('/');
    std::string sceneFile = scenePath;
    if (slash != std::string::npos) {
        if (chdir(scenePath.substr(0, slash).c_str()) != 0) { perror("chdir"); return 2; }
        sceneFile = scenePath.substr(slash + 1);
    }

    Scene* scene = parseSceneFile(sceneFile, 0.05f, W, H);  // eps: raytracer.cpp:849
    scene->maxDepth = depth;
    if (!dumpPath.empty() && !dump_scene(scene, depth, dumpPath.c_str())) {
        fprintf(stderr, "cannot write %s\n", dumpPath.c_str());
        return 2;
    }
    if (noRender) return 0;

    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        for (int y = tid; y < H; y += threads)
            for (int x = 0; x < W; x++) {
                Sample sm(x, y);
                Ray r = scene->cam.generateRay(sm);
                vec3 c = scene->traceRay(r, scene->maxDepth);
                scene->film.writeToFilm(sm, c);
            }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    if (outPath != "-") {
        std::vector<unsigned char> image((size_t)W * H * 4);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                size_t pos = 4 * ((size_t)y * W + x);
                const vec3& px = scene->film.film[(size_t)(H - y - 1) * W + x];
                image[pos] = (unsigned char)px[0];
                image[pos + 1] = (unsigned char)px[1];
                image[pos + 2] = (unsigned char)px[2];
                image[pos + 3] = 255;
            }
        FILE* f = fopen(outPath.c_str(), "wb");
        if (!f) { perror("fopen"); return 2; }
        fwrite(image.data(), 1, image.size(), f);
        fclose(f);
    }
    printf("{\"render_s\": %.6f, \"width\": %d, \"height\": %d, \"depth\": %d, \"threads\": %d, \"objects\": %zu, \"lights\": %zu}\n",
           secs, W, H, depth, threads, scene->objects.size(), scene->lights.size());
    return 0;
}
