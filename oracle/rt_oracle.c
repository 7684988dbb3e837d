/* TEST INFRASTRUCTURE -- CPU restatement of the reference trace loop.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library; the product path (raytracer_amd/) never does.
 *
 * Plain C99 over the flat rt_scene_desc (include/rt_abi.h).  Every function
 * follows the reference function named in its comment, operation by
 * operation, including the points where a double is rounded to float and the
 * reference's quirks (wrong-space nearest-hit metric, untransformed shadow
 * rays, last object's ray used for shading, NaN triangles that shadow
 * everything, spheres returning the root of smaller |t|).  Built with
 * -ffp-contract=off.  Parity is pinned: tests/test_oracle.py checks this
 * restatement bit for bit against oracle/_ref (the unmodified reference) and
 * against the reference's committed golden PNGs.
 */
#include "rt_abi.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define RT_MISS FLT_MAX /* MAXFLOAT in the reference (raytracer.cpp:78, 308, 323) */

typedef struct { double x, y, z; } v3;

typedef struct rto_counters {
    uint64_t rays_primary, rays_reflect, rays_shadow, hits, closest_tests, shadow_tests;
} rto_counters;

/* ---- algebra3.h pieces the trace loop reaches ---- */
static v3 v3_make(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 v3_ptr(const double* p) { v3 r = {p[0], p[1], p[2]}; return r; }
static v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }  /* algebra3.h:675 */
static v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }  /* algebra3.h:678 */
static v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }                        /* algebra3.h:672 */
static v3 v3_scale(v3 a, double d) { return v3_make(d * a.x, d * a.y, d * a.z); }    /* algebra3.h:681 */
static double v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      /* algebra3.h:699 */
static v3 v3_cross(v3 a, v3 b) {                                                      /* algebra3.h:706 */
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static v3 v3_normalize(v3 a) {                                                        /* algebra3.h:638-662 */
    double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    double inv = 1. / len;
    return v3_make(a.x * inv, a.y * inv, a.z * inv);
}
static v3 v3_mulc(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }  /* raytracer.cpp:36 */
/* mat4 * vec4 (algebra3.h:853-858), m row-major */
static void m4_mulv(const double* m, double x, double y, double z, double w, double out[4]) {
    for (int i = 0; i < 4; i++) out[i] = m[4 * i] * x + m[4 * i + 1] * y + m[4 * i + 2] * z + m[4 * i + 3] * w;
}
/* vec3(M * vec4(p, 1)): divides by w (algebra3.h:610) */
static v3 m4_point(const double* m, v3 p) {
    double o[4];
    m4_mulv(m, p.x, p.y, p.z, 1.0, o);
    return v3_make(o[0] / o[3], o[1] / o[3], o[2] / o[3]);
}
/* vec3(M * vec4(v, 0), VW): drops w (algebra3.h:614-621) */
static v3 m4_dir(const double* m, v3 v) {
    double o[4];
    m4_mulv(m, v.x, v.y, v.z, 0.0, o);
    return v3_make(o[0], o[1], o[2]);
}
/* float distance(vec3&, vec3&) (raytracer.cpp:32-34) */
static float dist_f(v3 a, v3 b) {
    double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return (float)sqrt(dx * dx + dy * dy + dz * dz);
}
static v3 point_at(v3 pos, v3 vec, float t) { return v3_add(pos, v3_scale(vec, (double)t)); } /* raytracer.cpp:81 */

/* ---- primitives ---- */

/* Triangle::intersection (raytracer.cpp:312-357); g = vertex1, vertex2, vertex3 */
static float tri_intersection(const double* g, v3 pos, v3 vec, float eps) {
    v3 v1 = v3_ptr(g), v2 = v3_ptr(g + 3), vtx3 = v3_ptr(g + 6);
    v3 edge1 = v3_sub(vtx3, v1), edge2 = v3_sub(v2, v1);       /* raytracer.cpp:288-289 */
    v3 normal = v3_normalize(v3_cross(edge2, edge1));
    v3 w0 = v3_sub(pos, v1);
    float a = (float)-(v3_dot(normal, w0));
    float b = (float)v3_dot(normal, vec);
    float r = a / b;
    if (fabsf(b) < eps) {
        if (a == 0.0f) return r;
        return RT_MISS;
    }
    if (r < 0.0f) return RT_MISS;
    v3 isect = point_at(pos, vec, r);
    float uu = (float)v3_dot(edge1, edge1);
    float uv = (float)v3_dot(edge1, edge2);
    float vv = (float)v3_dot(edge2, edge2);
    v3 w = v3_sub(isect, v1);
    float wu = (float)v3_dot(w, edge1);
    float wv = (float)v3_dot(w, edge2);
    float D = uv * uv - uu * vv;
    float s = (uv * wv - vv * wu) / D;
    if (s < 0.0f || s > 1.0f) return RT_MISS;
    float t = (uv * wu - uu * wv) / D;
    if (t < 0.0f || (s + t) > 1.0f) return RT_MISS;
    if (r < eps) return RT_MISS;
    return r;
}

/* Triangle::normal (raytracer.cpp:359-373); n9 = normal1..3 */
static v3 tri_normal(const double* g, const double* n9, int interp, v3 point) {
    v3 v1 = v3_ptr(g), v2 = v3_ptr(g + 3), vtx3 = v3_ptr(g + 6);
    v3 edge1 = v3_sub(vtx3, v1), edge2 = v3_sub(v2, v1);
    if (!interp) return v3_cross(edge2, edge1);
    v3 normal = v3_normalize(v3_cross(edge2, edge1));
    float a1 = (float)v3_dot(normal, v3_cross(v3_sub(v2, v1), v3_sub(vtx3, v1)));
    float a2 = (float)v3_dot(normal, v3_cross(v3_sub(v2, point), v3_sub(vtx3, point)));
    float a3 = (float)v3_dot(normal, v3_cross(v3_sub(vtx3, point), v3_sub(v1, point)));
    float alpha = a2 / a1;
    float beta = a3 / a1;
    float gamma = 1.0f - alpha - beta;
    return v3_add(v3_add(v3_scale(v3_ptr(n9), (double)alpha), v3_scale(v3_ptr(n9 + 3), (double)beta)),
                  v3_scale(v3_ptr(n9 + 6), (double)gamma));
}

/* Sphere::intersection (raytracer.cpp:393-414); g = center[3], radius */
static float sph_intersection(const double* g, v3 pos, v3 vec, float eps) {
    v3 center = v3_ptr(g);
    float radius = (float)g[3];
    v3 p_c = v3_sub(pos, center);
    v3 d = vec;
    double dpc = v3_dot(d, p_c);
    float disc = (float)(dpc * dpc - (v3_dot(d, d) * (v3_dot(p_c, p_c) - (double)radius * (double)radius)));
    if (disc < 0) return RT_MISS;
    float sq = sqrtf(disc);                                   /* sqrt(float) -> float overload */
    float t1 = (float)((v3_dot(v3_neg(d), p_c) + (double)sq) / v3_dot(d, d));
    float t2 = (float)((v3_dot(v3_neg(d), p_c) - (double)sq) / v3_dot(d, d));
    if (t1 < eps && t2 < eps) return RT_MISS;
    v3 p1 = point_at(pos, vec, t1), p2 = point_at(pos, vec, t2);
    float d1 = dist_f(pos, p1), d2 = dist_f(pos, p2);
    return d1 < d2 ? t1 : t2;
}

static float obj_intersection(const rt_scene_desc* s, int i, v3 pos, v3 vec, float eps) {
    const double* g = s->obj_geom + 9 * (size_t)i;
    return s->obj_kind[i] == 0 ? tri_intersection(g, pos, vec, eps) : sph_intersection(g, pos, vec, eps);
}

/* ---- the trace loop ---- */

/* Scene::traceShadowRay (raytracer.cpp:551-558): the world-space ray is tested
 * against every object's untransformed geometry with eps 0.1 and no far limit. */
static int trace_shadow(const rt_scene_desc* s, v3 pos, v3 vec, int source, rto_counters* c) {
    if (c) c->rays_shadow++;
    for (int i = 0; i < s->n_objects; i++) {
        if (c) c->shadow_tests++;
        if (obj_intersection(s, i, pos, vec, s->shadow_eps) != RT_MISS && i != source) return 1;
    }
    return 0;
}

/* Scene::phongShade (raytracer.cpp:560-581) with the Light methods inlined
 * (raytracer.cpp:108-128). ray_vec is the LAST object's transformed direction. */
static v3 phong_shade(const rt_scene_desc* s, v3 ray_vec, int hit, v3 normal, v3 point, rto_counters* c) {
    const rt_material* m = &s->materials[s->obj_material[hit]];
    v3 total = v3_make(0, 0, 0);
    total = v3_add(total, v3_mulc(v3_ptr(s->ambient), v3_ptr(m->ka)));
    for (int i = 0; i < s->n_lights; i++) {
        const rt_light* L = &s->lights[i];
        v3 lpos = v3_ptr(L->pos), color = v3_ptr(L->color);
        v3 intensity = color;
        if (L->falloff != 0) {
            float d = dist_f(lpos, point);
            double inv = 1. / pow((double)d, (double)L->falloff);
            intensity = v3_make(color.x * inv, color.y * inv, color.z * inv);
        }
        v3 lvec = L->is_point ? v3_sub(lpos, point) : v3_neg(lpos);
        v3 sdir = v3_normalize(lvec);
        if (trace_shadow(s, point, sdir, hit, c)) continue;
        v3 l = lvec;
        v3 v = v3_neg(ray_vec);
        v3 r = v3_add(v3_neg(l), v3_scale(normal, 2 * v3_dot(l, normal)));
        l = v3_normalize(l);
        v = v3_normalize(v);
        r = v3_normalize(r);
        double ln = v3_dot(l, normal), rv = v3_dot(r, v);
        double diff = ln > 0.0f ? ln : 0.0f;
        double spec = pow(rv > 0.0f ? rv : 0.0f, m->sp);
        v3 coeffs = v3_add(v3_scale(v3_ptr(m->kd), diff), v3_scale(v3_ptr(m->ks), spec));
        total = v3_add(total, v3_mulc(intensity, coeffs));
    }
    return total;
}

typedef struct { int hit; float t, d; v3 point, ray_vec; } rto_hit;

/* The object loop of Scene::traceRay (raytracer.cpp:515-530). */
static rto_hit closest_hit(const rt_scene_desc* s, v3 pos, v3 vec, rto_counters* c) {
    rto_hit h;
    h.hit = -1; h.t = RT_MISS; h.d = RT_MISS;
    h.point = v3_make(0, 0, 0); h.ray_vec = v3_make(0, 0, 0);
    float min_dist = RT_MISS;
    for (int i = 0; i < s->n_objects; i++) {
        const rt_transform* T = &s->transforms[s->obj_transform[i]];
        v3 rpos = m4_point(T->inv, pos);              /* Polygon::transformRay, raytracer.cpp:251-259 */
        v3 rvec = m4_dir(T->inv, vec);
        h.ray_vec = rvec;
        if (c) c->closest_tests++;
        float t = obj_intersection(s, i, rpos, rvec, s->eps);
        v3 isect = m4_point(T->m, point_at(rpos, rvec, t));   /* transformIntersection, raytracer.cpp:267 */
        float d = dist_f(rpos, isect);
        if (t != RT_MISS && d < min_dist) {
            min_dist = d;
            h.hit = i; h.t = t; h.d = d; h.point = isect;
        }
    }
    return h;
}

/* Scene::traceRay (raytracer.cpp:514-549) */
static v3 trace_ray(const rt_scene_desc* s, v3 pos, v3 vec, int depth, rto_counters* c) {
    rto_hit h = closest_hit(s, pos, vec, c);
    if (h.hit < 0) return v3_make(0, 0, 0);
    if (c) c->hits++;
    int i = h.hit;
    const rt_transform* T = &s->transforms[s->obj_transform[i]];
    const double* g = s->obj_geom + 9 * (size_t)i;
    v3 local = m4_point(T->inv, h.point);             /* mat4 * vec3, algebra3.h:693 */
    v3 normal;
    if (s->obj_kind[i] == 0)
        normal = tri_normal(g, s->obj_normals ? s->obj_normals + 9 * (size_t)i : g, s->obj_interp[i], local);
    else
        normal = v3_sub(local, v3_ptr(g));           /* Sphere::normal, raytracer.cpp:416 */
    normal = m4_dir(T->inv_t, normal);               /* transformNormal, raytracer.cpp:260-266 */
    normal = v3_normalize(normal);
    v3 refl = v3_sub(h.ray_vec, v3_scale(normal, 2 * v3_dot(h.ray_vec, normal)));
    refl = v3_normalize(refl);
    v3 shade = phong_shade(s, h.ray_vec, i, normal, h.point, c);
    if (depth == 0) return shade;
    if (c) c->rays_reflect++;
    v3 kr = v3_ptr(s->materials[s->obj_material[i]].kr);
    return v3_add(shade, v3_mulc(trace_ray(s, h.point, refl, depth - 1, c), kr));
}

/* Camera::generateRay (raytracer.cpp:441-447) */
static void camera_ray(const rt_scene_desc* s, int x, int y, v3* pos, v3* vec) {
    float u = (float)x / (float)s->width;
    float v = (float)y / (float)s->height;
    v3 ll = v3_ptr(s->cam_ll), lr = v3_ptr(s->cam_lr), ul = v3_ptr(s->cam_ul), ur = v3_ptr(s->cam_ur);
    v3 bottom = v3_add(v3_scale(ll, (double)(1 - u)), v3_scale(lr, (double)u));
    v3 top = v3_add(v3_scale(ul, (double)(1 - u)), v3_scale(ur, (double)u));
    v3 cpt = v3_add(v3_scale(bottom, (double)(1 - v)), v3_scale(top, (double)v));
    *pos = v3_ptr(s->cam_eye);
    *vec = v3_sub(cpt, *pos);
}

/* floatToByte + Film::writeToFilm + the uchar store of Film::writeFile
 * (raytracer.cpp:174-203, 225): truncate, clamp the high side only.  Values the
 * int cast cannot hold behave as on x86-64 (INT_MIN); they do not occur with
 * finite non-negative colours. */
static unsigned char quantise(double cval) {
    double scaled = cval * 255.0f;
    int v = (scaled >= -2147483648.0 && scaled < 2147483648.0) ? (int)scaled : (-2147483647 - 1);
    if (v > 255) v = 255;
    return (unsigned char)(v & 0xFF);
}

typedef struct {
    const rt_scene_desc* s;
    int y0, y1, tid, nthreads;
    uint8_t* rgba;
    rto_counters c;
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    const rt_scene_desc* s = j->s;
    int W = s->width, H = s->height;
    for (int y = j->y0 + j->tid; y < j->y1; y += j->nthreads)
        for (int x = 0; x < W; x++) {
            v3 pos, vec;
            camera_ray(s, x, y, &pos, &vec);
            j->c.rays_primary++;
            v3 col = trace_ray(s, pos, vec, s->max_depth, &j->c);
            uint8_t* px = j->rgba + 4 * ((size_t)(H - 1 - y) * W + x);   /* flip: raytracer.cpp:225 */
            px[0] = quantise(col.x); px[1] = quantise(col.y); px[2] = quantise(col.z); px[3] = 255;
        }
    return NULL;
}

/* Scene::render (raytracer.cpp:497-512) for film rows [y0, y1) into a full
 * frame in Film::writeFile's layout (row 0 = top).  Rows outside the range
 * are left untouched. */
int rto_render(const rt_scene_desc* s, int y0, int y1, int nthreads, uint8_t* rgba, rto_counters* out) {
    if (!s || !rgba || y0 < 0 || y1 > s->height || y0 > y1) return RT_ERR_ARG;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t* jobs = (job_t*)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) {
        jobs[t].s = s; jobs[t].y0 = y0; jobs[t].y1 = y1; jobs[t].tid = t; jobs[t].nthreads = nthreads;
        jobs[t].rgba = rgba;
        if (t > 0) pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    worker(&jobs[0]);
    rto_counters sum;
    memset(&sum, 0, sizeof sum);
    for (int t = 0; t < nthreads; t++) {
        if (t > 0) pthread_join(th[t], NULL);
        sum.rays_primary += jobs[t].c.rays_primary; sum.rays_reflect += jobs[t].c.rays_reflect;
        sum.rays_shadow += jobs[t].c.rays_shadow; sum.hits += jobs[t].c.hits;
        sum.closest_tests += jobs[t].c.closest_tests; sum.shadow_tests += jobs[t].c.shadow_tests;
    }
    if (out) *out = sum;
    free(jobs); free(th);
    return 0;
}

/* Known-answer entry points for unit tests of the device functions. */

/* Nearest hit of n rays (orig/dir: [n][3]); hit = object index or -1. */
int rto_closest(const rt_scene_desc* s, const double* orig, const double* dir, int n,
                int32_t* hit, float* t, float* d, double* point) {
    for (int k = 0; k < n; k++) {
        rto_hit h = closest_hit(s, v3_ptr(orig + 3 * k), v3_ptr(dir + 3 * k), NULL);
        hit[k] = h.hit; t[k] = h.t; d[k] = h.d;
        point[3 * k] = h.point.x; point[3 * k + 1] = h.point.y; point[3 * k + 2] = h.point.z;
    }
    return 0;
}

/* Shadow query of n rays from object `source[k]` (or -1). */
int rto_occluded(const rt_scene_desc* s, const double* orig, const double* dir, const int32_t* source, int n,
                 int32_t* occluded) {
    for (int k = 0; k < n; k++)
        occluded[k] = trace_shadow(s, v3_ptr(orig + 3 * k), v3_ptr(dir + 3 * k), source[k], NULL);
    return 0;
}

/* Full colour (before quantisation) of n rays traced with `depth`. */
int rto_trace(const rt_scene_desc* s, const double* orig, const double* dir, int n, int depth, double* rgb) {
    for (int k = 0; k < n; k++) {
        v3 c = trace_ray(s, v3_ptr(orig + 3 * k), v3_ptr(dir + 3 * k), depth, NULL);
        rgb[3 * k] = c.x; rgb[3 * k + 1] = c.y; rgb[3 * k + 2] = c.z;
    }
    return 0;
}

/* One bounce of n rays, for per-ray known answers beyond the camera rays: the nearest hit
 * (raytracer.cpp:515-530), the hit point and the reflected ray's direction (raytracer.cpp:531-538), and
 * per light whether the shadow ray from the hit point is blocked (raytracer.cpp:565-567, 551-558).
 * hit[k] = object index or -1; point, refl: [n][3]; occluded: [n][words] with bit l of word l / 32 set when
 * light l is blocked, words = (n_lights + 31) / 32 (at least 1). */
int rto_bounce(const rt_scene_desc* s, const double* orig, const double* dir, int n,
               int32_t* hit, double* point, double* refl_out, uint32_t* occluded) {
    const int words = s->n_lights > 0 ? (s->n_lights + 31) / 32 : 1;
    for (int k = 0; k < n; k++) {
        rto_hit h = closest_hit(s, v3_ptr(orig + 3 * k), v3_ptr(dir + 3 * k), NULL);
        hit[k] = h.hit;
        for (int w = 0; w < words; w++) occluded[(size_t)k * words + w] = 0u;
        for (int a = 0; a < 3; a++) { point[3 * k + a] = 0.0; refl_out[3 * k + a] = 0.0; }
        if (h.hit < 0) continue;
        int i = h.hit;
        const rt_transform* T = &s->transforms[s->obj_transform[i]];
        const double* g = s->obj_geom + 9 * (size_t)i;
        v3 local = m4_point(T->inv, h.point);
        v3 normal;
        if (s->obj_kind[i] == 0)
            normal = tri_normal(g, s->obj_normals ? s->obj_normals + 9 * (size_t)i : g, s->obj_interp[i], local);
        else
            normal = v3_sub(local, v3_ptr(g));
        normal = m4_dir(T->inv_t, normal);
        normal = v3_normalize(normal);
        v3 refl = v3_sub(h.ray_vec, v3_scale(normal, 2 * v3_dot(h.ray_vec, normal)));
        refl = v3_normalize(refl);
        point[3 * k] = h.point.x; point[3 * k + 1] = h.point.y; point[3 * k + 2] = h.point.z;
        refl_out[3 * k] = refl.x; refl_out[3 * k + 1] = refl.y; refl_out[3 * k + 2] = refl.z;
        for (int l = 0; l < s->n_lights; l++) {
            const rt_light* L = &s->lights[l];
            v3 lvec = L->is_point ? v3_sub(v3_ptr(L->pos), h.point) : v3_neg(v3_ptr(L->pos));
            if (trace_shadow(s, h.point, v3_normalize(lvec), i, NULL)) occluded[(size_t)k * words + (l >> 5)] |= 1u << (l & 31);
        }
    }
    return 0;
}

/* Camera rays of film row y: orig/dir [width][3]. */
int rto_camera_row(const rt_scene_desc* s, int y, double* orig, double* dir) {
    for (int x = 0; x < s->width; x++) {
        v3 p, v;
        camera_ray(s, x, y, &p, &v);
        orig[3 * x] = p.x; orig[3 * x + 1] = p.y; orig[3 * x + 2] = p.z;
        dir[3 * x] = v.x; dir[3 * x + 1] = v.y; dir[3 * x + 2] = v.z;
    }
    return 0;
}
