// TEST INFRASTRUCTURE -- the reference program with its render() statement swapped for the MI355X library.
//
// This is the binding a maintainer of the reference would add: the
// reference's own parser (parseSceneFile, raytracer.cpp:705), the walk over the public members of its Scene
// that fills an rt_scene_desc, rt_upload_scene / rt_render_tile / rt_download in place of `scene->render()`
// (raytracer.cpp:857), and the reference's own PNG encoder (lodepng::encode, raytracer.cpp:231) on the vector
// that Film::writeFile would have built (raytracer.cpp:212-230).  The reference's sources are compiled where
// they lie (oracle/Makefile passes the include path); nothing of them is copied.  The reference's main is
// renamed out of the way; this file's main differs from it only in taking the frame size and depth from the
// command line (the reference hard-codes 1000 x 1000 and depth 6, raytracer.cpp:850-851, 477):
//
//   ref_main_gpu <scene> <out.png> [W H depth]
#define main reference_main_unused
#include "raytracer.cpp"
#undef main

#include "rt_abi.h"

#include <cstring>
#include <unistd.h>

static bool same_bits(const mat4& a, const mat4& b) {             // de-duplicate by bit pattern
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double x = a[i][j], y = b[i][j];
        if (memcmp(&x, &y, 8)) return false;
    }
    return true;
}

int render_on_mi355x(Scene* s, std::vector<unsigned char>& image) {
    std::vector<rt_transform> xf;  std::vector<mat4> xf_src;
    std::vector<rt_material> mats; std::vector<rt_light> lights;
    std::vector<int32_t> kind, mat_id, xf_id, interp;
    std::vector<double> geom, nrm;
    for (Polygon* p : s->objects) {                                // order = Scene::objects order
        int xi = -1;
        for (int k = (int)xf_src.size() - 1; xi < 0 && k >= 0; k--) if (same_bits(xf_src[k], p->transform)) xi = k;
        if (xi < 0) {
            rt_transform t;
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
                t.m[4*i+j] = p->transform[i][j];                   // raytracer.cpp:241
                t.inv[4*i+j] = p->inverseTransform[i][j];          // computed by the reference's own
                t.inv_t[4*i+j] = p->inverseTranspose[i][j];        // mat4::inverse(): bits must agree
            }
            xi = (int)xf.size(); xf.push_back(t); xf_src.push_back(p->transform);
        }
        rt_material m;                                              // raytracer.cpp:44-64
        memset(&m, 0, sizeof m);
        for (int c = 0; c < 3; c++) { m.ka[c] = p->mat.ka[c]; m.kd[c] = p->mat.kd[c];
                                      m.ks[c] = p->mat.ks[c]; m.kr[c] = p->mat.kr[c]; }
        m.sp = p->mat.sp;
        int mi = -1;
        for (int k = (int)mats.size() - 1; mi < 0 && k >= 0; k--) if (!memcmp(&mats[k], &m, sizeof m)) mi = k;
        if (mi < 0) { mi = (int)mats.size(); mats.push_back(m); }
        xf_id.push_back(xi); mat_id.push_back(mi);
        if (Triangle* t = dynamic_cast<Triangle*>(p)) {
            kind.push_back(0); interp.push_back(t->needInterpolate);
            const vec3* v[3] = {&t->vertex1, &t->vertex2, &t->vertex3};
            const vec3* n[3] = {&t->normal1, &t->normal2, &t->normal3};
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) {
                geom.push_back((*v[a])[c]);
                nrm.push_back(t->needInterpolate ? (*n[a])[c] : 0.0);
            }
        } else {
            Sphere* sp = static_cast<Sphere*>(p);
            kind.push_back(1); interp.push_back(0);
            for (int c = 0; c < 3; c++) geom.push_back(sp->center[c]);
            geom.push_back(sp->radius);
            for (int c = 4; c < 9; c++) geom.push_back(0.0);
            for (int c = 0; c < 9; c++) nrm.push_back(0.0);
        }
    }
    for (Light* l : s->lights) {                                    // raytracer.cpp:86-106
        rt_light o;
        memset(&o, 0, sizeof o);
        for (int c = 0; c < 3; c++) { o.color[c] = l->color[c]; o.pos[c] = l->pos[c]; }
        o.falloff = l->falloff; o.is_point = l->pointLight;
        lights.push_back(o);
    }
    rt_scene_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    d.width = s->width; d.height = s->height; d.max_depth = s->maxDepth;
    d.eps = s->eps; d.shadow_eps = 0.1f;                            // raytracer.cpp:308, 390
    for (int c = 0; c < 3; c++) {
        d.cam_eye[c] = s->cam.eye[c]; d.cam_ll[c] = s->cam.ll[c]; d.cam_lr[c] = s->cam.lr[c];
        d.cam_ul[c] = s->cam.ul[c]; d.cam_ur[c] = s->cam.ur[c];
        d.ambient[c] = s->ambientLight.color[c];                    // 0 unless an `lta` line appeared
    }
    d.n_objects = (int)kind.size(); d.n_materials = (int)mats.size();
    d.n_transforms = (int)xf.size(); d.n_lights = (int)lights.size();
    d.obj_kind = kind.data(); d.obj_material = mat_id.data(); d.obj_transform = xf_id.data();
    d.obj_interp = interp.data(); d.obj_geom = geom.data(); d.obj_normals = nrm.data();
    d.materials = mats.data(); d.transforms = xf.data(); d.lights = lights.data();

    rt_ctx* ctx = nullptr;
    if (rt_upload_scene(&ctx, &d)) { fprintf(stderr, "%s\n", rt_last_error()); return -1; }
    int rc = rt_render_tile(ctx, 0, s->height, /*stream*/ nullptr);
    image.assign((size_t)s->width * s->height * 4, 0);
    if (!rc) rc = rt_download(ctx, image.data(), image.size());
    if (rc) fprintf(stderr, "%s\n", rt_last_error());
    rt_free(ctx);
    return rc;          // then: lodepng::encode(outFile, image, s->width, s->height)  (raytracer.cpp:231)
}

int main(int argc, char* argv[]) {
    if (argc < 3) { fprintf(stderr, "usage: %s <scene> <out.png> [W H depth]\n", argv[0]); return 2; }
    std::string scene_path = argv[1], out_path = argv[2];
    const int W = argc > 3 ? atoi(argv[3]) : 1000, H = argc > 4 ? atoi(argv[4]) : 1000;     // raytracer.cpp:850-851
    const int depth = argc > 5 ? atoi(argv[5]) : 6;                                         // raytracer.cpp:477
    // the reference resolves `obj` file names against the working directory (raytracer.cpp:629)
    if (out_path[0] != '/') { char cwd[4096]; if (getcwd(cwd, sizeof cwd)) out_path = std::string(cwd) + "/" + out_path; }
    size_t slash = scene_path.find_last_of('/');
    std::string file = scene_path;
    if (slash != std::string::npos) {
        if (chdir(scene_path.substr(0, slash).c_str()) != 0) { perror("chdir"); return 1; }
        file = scene_path.substr(slash + 1);
    }
    Scene* scene = parseSceneFile(file, 0.05f, W, H);               // raytracer.cpp:849-853
    scene->maxDepth = depth;
    std::vector<unsigned char> image;
    if (render_on_mi355x(scene, image)) return 1;                   // in place of scene->render(), raytracer.cpp:857
    unsigned error = lodepng::encode(out_path, image, W, H);        // raytracer.cpp:231
    if (error) { fprintf(stderr, "encoder error %u: %s\n", error, lodepng_error_text(error)); return 1; }
    return 0;
}
