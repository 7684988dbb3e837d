// rt_upload_scene's work: per-primitive records of both forms of the FP32 filter, their margins, the bounding
// tables of triangles and clusters, the order of the slots (install_filter) -- host precomputation in double.
#pragma once

namespace {

// Smallest sphere around a triangle (centre in its plane): the circumscribed circle, or the
// circle over the longest edge when the angle opposite to it is not acute.
void bounding_sphere(const TriExact& ex, Vec* centre, double* radius) {
    const Vec a{ex.v1[0], ex.v1[1], ex.v1[2]};
    const Vec ab{ex.e2[0], ex.e2[1], ex.e2[2]}, ac{ex.e1[0], ex.e1[1], ex.e1[2]};   // v2 - v1, v3 - v1
    const Vec bc = vsub(ac, ab);
    const double lab = vdot(ab, ab), lac = vdot(ac, ac), lbc = vdot(bc, bc);
    Vec rel;        // centre - a
    if (lab >= lac + lbc) rel = {ab.x * .5, ab.y * .5, ab.z * .5};
    else if (lac >= lab + lbc) rel = {ac.x * .5, ac.y * .5, ac.z * .5};
    else if (lbc >= lab + lac) rel = {(ab.x + ac.x) * .5, (ab.y + ac.y) * .5, (ab.z + ac.z) * .5};
    else {
        const Vec n = vcross(ab, ac);
        const double nn = vdot(n, n);
        const Vec t1 = vcross(n, ab), t2 = vcross(ac, n);
        rel = {(lac * t1.x + lab * t2.x) / (2 * nn), (lac * t1.y + lab * t2.y) / (2 * nn), (lac * t1.z + lab * t2.z) / (2 * nn)};
    }
    if (!(std::isfinite(rel.x) && std::isfinite(rel.y) && std::isfinite(rel.z))) rel = {(ab.x + ac.x) / 3, (ab.y + ac.y) / 3, (ab.z + ac.z) / 3};
    *centre = {a.x + rel.x, a.y + rel.y, a.z + rel.z};
    // the radius is measured, not derived: whatever the centre, the sphere holds the three vertices
    const Vec da = rel, db = vsub(rel, ab), dc = vsub(rel, ac);
    *radius = std::sqrt(std::max(vdot(da, da), std::max(vdot(db, db), vdot(dc, dc))));
}

// Cone from `apex` around the sphere (centre, radius): unit axis and cosine of the half-angle,
// lowered by the rounding of the FP32 test; cos = -2 (every direction passes) when the apex is
// inside the sphere or nearly so.
void bounding_cone(Vec apex, Vec centre, double radius, float out[4]) {
    const Vec d = vsub(centre, apex);
    const double dist = vlen(d);
    out[0] = 0.f; out[1] = 0.f; out[2] = 1.f; out[3] = -2.f;
    if (!(dist > 0) || !std::isfinite(dist) || !(radius < 0.999 * dist)) return;
    const double sn = radius / dist;
    out[0] = (float)(d.x / dist); out[1] = (float)(d.y / dist); out[2] = (float)(d.z / dist);
    out[3] = (float)(std::sqrt(1.0 - sn * sn) - 4e-6);
}

// Orders the triangles for the chosen form of the filter (regular ones by transform group and
// dominant axis, then the irregular ones), builds the groups, and uploads the triangle records,
// the groups and the scene header.  Called by rt_upload_scene and by RT_OPT_FILTER_FORM.
// The form the context should have installed now (see rt_ctx::auto_robust).
int chosen_form(const rt_ctx* c) {
    if (c->form_option) return c->form_option == 2 ? 1 : 0;
    return c->pretest ? c->auto_robust_bounded : c->auto_robust;
}

int install_filter(rt_ctx* c, int robust) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());          // no launch may still read the records replaced here
    const std::vector<HostTri>& tris = c->h_tris;
    auto regular = [&](const HostTri& t) { return robust ? t.ok_robust : t.ok_static; };
    // regular triangles by (transform, dominant axis), irregular ones by transform, file order
    // within a key: a counting sort, the keys are few
    std::vector<int> order_reg, order_irr;
    {
        int n_xf = 1;
        for (const HostTri& t : tris) n_xf = std::max(n_xf, t.xf + 1);
        std::vector<size_t> n_reg((size_t)n_xf * 3 + 1, 0), n_irr((size_t)n_xf + 1, 0);
        for (const HostTri& t : tris) {
            if (regular(t)) n_reg[(size_t)t.xf * 3 + t.axis + 1]++; else n_irr[(size_t)t.xf + 1]++;
        }
        for (size_t k = 1; k < n_reg.size(); k++) n_reg[k] += n_reg[k - 1];
        for (size_t k = 1; k < n_irr.size(); k++) n_irr[k] += n_irr[k - 1];
        order_reg.resize(n_reg.back()); order_irr.resize(n_irr.back());
        for (int k = 0; k < (int)tris.size(); k++) {
            const HostTri& t = tris[k];
            if (regular(t)) order_reg[n_reg[(size_t)t.xf * 3 + t.axis]++] = k; else order_irr[n_irr[(size_t)t.xf]++] = k;
        }
    }
    // Within a (transform, axis) run the scans ask their bounding questions of clusters of 64 and batches of 4
    // consecutive slots, so consecutive slots should be neighbours in space.  A mesh written patch by patch (every
    // mesh of the reference is) has that in file order already, and file order is kept: on the teapot and on
    // campanile + teddy it measured 3-5 % faster than the order along a space-filling curve through the centroids
    // (63-bit Morton codes in the run's own centroid box).  The curve takes over only where it is clearly better -- the
    // clusters' extents (largest distance of a member's centroid from the cluster's mean centroid) add up to less than
    // 0.7 of file order's -- i.e. for meshes whose faces come in no spatial order.  Nothing else depends on the order
    // of the slots (ties between hits go by object index, which travels with the record).
    // RAYTRACER_AMD_SLOT_ORDER=file / curve forces one or the other (A/B runs).
    {
        const char* env = std::getenv("RAYTRACER_AMD_SLOT_ORDER");
        const bool file_order = env && std::string(env) == "file", force_curve = env && std::string(env) == "curve";
        auto extent_sum = [&](const int* idx, size_t n) {
            double sum = 0;
            for (size_t b = 0; b < n; b += 64) {
                const size_t m = std::min<size_t>(64, n - b);
                double mean[3] = {0, 0, 0};
                for (size_t k = 0; k < m; k++) for (int q = 0; q < 3; q++) mean[q] += tris[(size_t)idx[b + k]].ex.v1[q] + (tris[(size_t)idx[b + k]].ex.e1[q] + tris[(size_t)idx[b + k]].ex.e2[q]) * (1.0 / 3.0);
                for (int q = 0; q < 3; q++) mean[q] /= (double)m;
                double far2 = 0;
                for (size_t k = 0; k < m; k++) {
                    double d2 = 0;
                    for (int q = 0; q < 3; q++) {
                        const double d = tris[(size_t)idx[b + k]].ex.v1[q] + (tris[(size_t)idx[b + k]].ex.e1[q] + tris[(size_t)idx[b + k]].ex.e2[q]) * (1.0 / 3.0) - mean[q];
                        d2 += d * d;
                    }
                    far2 = std::max(far2, d2);
                }
                sum += std::sqrt(far2);
            }
            return sum;
        };
        auto centroid = [&](int k, int a) { return tris[(size_t)k].ex.v1[a] + (tris[(size_t)k].ex.e1[a] + tris[(size_t)k].ex.e2[a]) * (1.0 / 3.0); };
        auto spread21 = [](unsigned long long v) {          // 21 bits -> every third bit
            v &= 0x1fffffull;
            v = (v | v << 32) & 0x1f00000000ffffull;
            v = (v | v << 16) & 0x1f0000ff0000ffull;
            v = (v | v << 8) & 0x100f00f00f00f00full;
            v = (v | v << 4) & 0x10c30c30c30c30c3ull;
            v = (v | v << 2) & 0x1249249249249249ull;
            return v;
        };
        size_t r = 0;
        std::vector<std::pair<unsigned long long, int>> keyed;
        while (!file_order && r < order_reg.size()) {
            const int x = tris[(size_t)order_reg[r]].xf, a = tris[(size_t)order_reg[r]].axis;
            size_t e = r;
            double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
            while (e < order_reg.size() && tris[(size_t)order_reg[e]].xf == x && tris[(size_t)order_reg[e]].axis == a) {
                for (int q = 0; q < 3; q++) { const double cq = centroid(order_reg[e], q); lo[q] = std::min(lo[q], cq); hi[q] = std::max(hi[q], cq); }
                e++;
            }
            double ext = 0;
            for (int q = 0; q < 3; q++) ext = std::max(ext, hi[q] - lo[q]);
            if (e - r > 64 && ext > 0 && std::isfinite(ext)) {
                keyed.resize(e - r);
                const double scale = 2097151.0 / ext;               // one scale for the three axes: cells are cubes
                parallel_ranges(e - r, host_parts(e - r), [&](size_t, size_t k0, size_t k1) {
                    for (size_t k = k0; k < k1; k++) {
                        unsigned long long code = 0;
                        for (int q = 0; q < 3; q++) {
                            const double v = (centroid(order_reg[r + k], q) - lo[q]) * scale;
                            const unsigned long long cell = (unsigned long long)std::min(2097151.0, std::max(0.0, v));
                            code |= spread21(cell) << q;
                        }
                        keyed[k] = {code, order_reg[r + k]};
                    }
                });
                std::sort(keyed.begin(), keyed.end());              // (ties: by index, i.e. file order)
                std::vector<int> curve(keyed.size());
                for (size_t k = 0; k < keyed.size(); k++) curve[k] = keyed[k].second;
                if (force_curve || extent_sum(curve.data(), curve.size()) < 0.7 * extent_sum(&order_reg[r], e - r))
                    for (size_t k = 0; k < curve.size(); k++) order_reg[r + k] = curve[k];
            }
            r = e;
        }
    }
    // Every (transform, axis) range starts at a multiple of 64 slots, so that clusters of 64 consecutive
    // slots never straddle two ranges: the slots between a range's last triangle and the next multiple of
    // 64 are holes (order_reg = -1).  No scan visits a hole (the ranges end at their last triangle); the
    // kernels that walk whole groups (full reference-precision scan, LBVH) skip them by TriExact::hole.
    c->n_regular_real = (int)order_reg.size();
    std::vector<int> slot_xf, slot_axis;             // per slot, holes included
    {
        std::vector<int> padded;
        padded.reserve(order_reg.size() + 192);
        size_t r = 0;
        while (r < order_reg.size()) {
            const int x = tris[order_reg[r]].xf, a = tris[order_reg[r]].axis;
            while (r < order_reg.size() && tris[order_reg[r]].xf == x && tris[order_reg[r]].axis == a) {
                padded.push_back(order_reg[r++]); slot_xf.push_back(x); slot_axis.push_back(a);
            }
            while (padded.size() % 64) { padded.push_back(-1); slot_xf.push_back(x); slot_axis.push_back(a); }
        }
        order_reg.swap(padded);
    }
    const int NR = (int)order_reg.size(), NI = (int)order_irr.size();
    const size_t parts = host_parts(tris.size());
    std::vector<v16f> filt((size_t)NR + 1);          // + one readable padding record
    std::memset(&filt[NR], 0, sizeof(v16f));
    std::vector<TriExact> exact((size_t)NR + NI);
    std::vector<int32_t>& slot = c->h_slot;
    parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            if (order_reg[k] < 0) {
                for (int j = 0; j < 16; j++) filt[k][j] = 0.f;
                std::memset(&exact[k], 0, sizeof(TriExact));
                exact[k].hole = 1; exact[k].obj = -1;
                continue;
            }
            const HostTri& t = tris[order_reg[k]];
            const float* f = robust ? t.f_robust : t.f_static;
            for (int j = 0; j < 16; j++) filt[k][j] = f[j];
            exact[k] = t.ex;
            exact[k].hole = 0;
            slot[t.ex.obj] = (int32_t)k;
        }
    });
    int n_void_normal = 0;
    // (RAYTRACER_AMD_TEST_VOID_NORMALS=1, a diagnostic switch: such triangles are tested against every ray like the others)
    const char* const void_env = std::getenv("RAYTRACER_AMD_TEST_VOID_NORMALS");
    const bool test_void = void_env && std::atoi(void_env) != 0;
    for (int k = 0; k < NI; k++) {
        const HostTri& t = tris[order_irr[k]];
        exact[(size_t)NR + k] = t.ex;
        // (zero-area triangle: (edge2 ^ edge1).normalize() is 0 / 0, raytracer.cpp:313)
        const bool void_normal = !test_void && (std::isnan(t.ex.n[0]) || std::isnan(t.ex.n[1]) || std::isnan(t.ex.n[2]));
        exact[(size_t)NR + k].hole = void_normal ? 2 : 0;
        if (void_normal) n_void_normal++;
        slot[t.ex.obj] = NR + k;
    }
    std::vector<Group> groups;
    {
        int r = 0, q = 0;
        while (r < NR || q < NI) {
            int xr = r < NR ? slot_xf[r] : INT32_MAX;
            int xq = q < NI ? tris[order_irr[q]].xf : INT32_MAX;
            int x = std::min(xr, xq);
            Group G{};
            G.xf = x; G.identity = c->h_xf_ident[x];
            G.q[0] = c->h_q0[x]; G.q[1] = c->h_q1[x];
            for (int a = 0; a < 3; a++) {
                G.fax[a] = r;
                G.fcnt[a] = 0;
                while (r < NR && slot_xf[r] == x && slot_axis[r] == a) { if (order_reg[r] >= 0) G.fcnt[a]++; r++; }
            }
            G.fax[3] = r;
            G.ibeg = NR + q;
            while (q < NI && tris[order_irr[q]].xf == x) q++;
            G.iend = NR + q;
            const int32_t grp_word = (int32_t)groups.size() | (G.identity ? RT_GRP_IDENTITY : 0);
            for (int k2 = G.fax[0]; k2 < G.fax[3]; k2++) exact[k2].grp = grp_word;
            for (int k2 = G.ibeg; k2 < G.iend; k2++) exact[k2].grp = grp_word;
            groups.push_back(G);
        }
    }

    c->h_groups = groups;
    c->all_identity = true;
    for (const Group& g : groups) if (!g.identity) c->all_identity = false;
    c->h_centroid_box.assign(6 * groups.size(), 0.f);
    for (size_t gi = 0; gi < groups.size(); gi++) {
        float* bx = &c->h_centroid_box[6 * gi];
        for (int a = 0; a < 3; a++) { bx[a] = FLT_MAX; bx[3 + a] = -FLT_MAX; }
        for (int k2 = groups[gi].fax[0]; k2 < groups[gi].fax[3]; k2++) {
            const TriExact& ex = exact[k2];
            if (ex.hole) continue;
            for (int a = 0; a < 3; a++) {
                float cc = (float)(ex.v1[a] + (ex.e1[a] + ex.e2[a]) * (1.0 / 3.0));
                bx[a] = std::min(bx[a], cc); bx[3 + a] = std::max(bx[3 + a], cc);
            }
        }
    }
    // a hierarchy built over the previous order is void
    c->bvh_built = false;
    c->bvh_nodes.release(); c->bvh_groups.release();

    // Constants shared by all rays of a launch, in double precision like the reference's own
    // dot products (raytracer.cpp:316-317): a = -(n . (eye - v1)) for camera rays (the eye in the
    // group's object space), 1 / (n . dir) for the rays towards each directional light
    // (world-space direction against the untransformed normal, raytracer.cpp:553).
    {
        std::vector<float> eye_a((size_t)NR + 1, 0.f);
        std::vector<float> cone_cam(4 * ((size_t)NR + 8), 0.f);      // readable a batch past the end
        for (size_t k = (size_t)NR; k < (size_t)NR + 8; k++) cone_cam[4 * k + 3] = -2.f;
        // bounding sphere of every regular triangle: what the reference can accept lies within
        // box_pad of the triangle; the rest covers the rounding of centre and radius
        std::vector<double> sphere(4 * (size_t)NR);
        const double sphere_pad = 64.0 * 1.1920928955078125e-7 * c->scene_bound;
        parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++) {
                Vec centre; double radius;
                bounding_sphere(exact[k], &centre, &radius);
                sphere[4 * k] = centre.x; sphere[4 * k + 1] = centre.y; sphere[4 * k + 2] = centre.z;
                sphere[4 * k + 3] = (radius + exact[k].box_pad) * (1 + 1e-6) + sphere_pad;
            }
        });
        // rays with nothing in common: sphere as (centre, rho^2 - |centre|^2 + rounding of the expanded
        // distance test, 64 eps R^2 with R the group's bound on |origin| and |vertex|), plane as (n, d0)
        std::vector<float> sphere_any(4 * ((size_t)NR + 8), 0.f), plane_any(4 * ((size_t)NR + 8), 0.f);
        for (size_t k = (size_t)NR; k < (size_t)NR + 8; k++) { sphere_any[4 * k + 3] = -INFINITY; plane_any[4 * k + 3] = INFINITY; }
        const Vec eye{c->hs.cam_eye[0], c->hs.cam_eye[1], c->hs.cam_eye[2]};
        for (const Group& G : groups) {
            const Vec o = G.identity ? eye : affine_point(&c->h_xf[48 * (size_t)G.xf + 16], eye);
            const size_t first = (size_t)G.fax[0], count = (size_t)(G.fax[3] - G.fax[0]);
            const double Rg = (double)G.q[0].e_a / (16.0 * 1.1920928955078125e-7) * (1 + 1e-3);
            const double slack_any = 64.0 * 1.1920928955078125e-7 * Rg * Rg;
            parallel_ranges(count, host_parts(count), [&](size_t, size_t k0, size_t k1) {
                for (size_t k = first + k0; k < first + k1; k++) {
                    const TriExact& ex = exact[k];
                    {
                        const double cx = sphere[4 * k], cy = sphere[4 * k + 1], cz = sphere[4 * k + 2], rho = sphere[4 * k + 3];
                        const double kk = rho * rho - (cx * cx + cy * cy + cz * cz) + slack_any;
                        sphere_any[4 * k] = (float)cx; sphere_any[4 * k + 1] = (float)cy; sphere_any[4 * k + 2] = (float)cz;
                        float kf = std::isfinite(kk) ? (float)kk : INFINITY;
                        if (std::isfinite(kf)) kf += std::fabs(kf) * 1e-6f;
                        if (!std::isfinite(sphere_any[4 * k]) || !std::isfinite(sphere_any[4 * k + 1]) || !std::isfinite(sphere_any[4 * k + 2])) {
                            sphere_any[4 * k] = sphere_any[4 * k + 1] = sphere_any[4 * k + 2] = 0.f; kf = INFINITY;
                        }
                        sphere_any[4 * k + 3] = kf;
                        for (int q = 0; q < 4; q++) plane_any[4 * k + q] = filt[k][q];
                    }
                    const Vec w0{o.x - ex.v1[0], o.y - ex.v1[1], o.z - ex.v1[2]};
                    eye_a[k] = (float)(-(ex.n[0] * w0.x + ex.n[1] * w0.y + ex.n[2] * w0.z));
                    bounding_cone(o, {sphere[4 * k], sphere[4 * k + 1], sphere[4 * k + 2]}, sphere[4 * k + 3], &cone_cam[4 * k]);
                    // the in-plane-origin case (raytracer.cpp:319-325) ignores where the triangle is
                    if (!(std::fabs(eye_a[k]) > G.q[0].a_zero)) cone_cam[4 * k + 3] = -2.f;
                }
            });
        }
        c->n_eye_plane = 0;
        for (const Group& G : groups)
            for (int k = G.fax[0]; k < G.fax[3]; k++)
                if (!exact[(size_t)k].hole && !(std::fabs(eye_a[(size_t)k]) > G.q[0].a_zero)) c->n_eye_plane++;
        // Per light (shadow rays meet the untransformed triangles in world space, raytracer.cpp:553):
        // point light: cone from the light, and the light's distance from the triangle's plane;
        // directional light: disc in the plane perpendicular to the light's direction, coordinates
        // about the centre of the scene.
        // (16 bytes per triangle and light: beyond 2 GiB the table is left out and shadow rays take the
        // triangle test for every pair)
        if (c->h_lights.size() * ((size_t)NR + 8) * 16 > ((size_t)2 << 30)) {
            c->bound_lights.release(); c->light_frame.release();
        } else {
            const size_t NL = c->h_lights.size(), stride = (size_t)NR + 8;
            std::vector<float> bound(NL * stride * 4, 0.f);
            std::vector<double> frame(std::max<size_t>(NL, 1) * 8, 0.0);
            const double half = c->scene_bound;                       // bound of |origin| and |vertex| for every ray and triangle
            for (size_t l = 0; l < NL; l++) {
                const DevLight& lt = c->h_lights[l];
                float* B = &bound[l * stride * 4];
                for (size_t k = (size_t)NR; k < stride; k++) { B[4 * k + 2] = lt.is_point ? 1.f : -INFINITY; B[4 * k + 3] = lt.is_point ? 2.f : 0.f; }
                if (lt.is_point) {
                    const Vec Lp{lt.pos[0], lt.pos[1], lt.pos[2]};
                    parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
                        for (size_t k = k0; k < k1; k++) {
                            bounding_cone(Lp, {sphere[4 * k], sphere[4 * k + 1], sphere[4 * k + 2]}, sphere[4 * k + 3], &B[4 * k]);
                        }
                    });
                } else {
                    // Light::lightRay: direction = (-pos).normalize() (raytracer.cpp:113-117)
                    const double x = -lt.pos[0], y = -lt.pos[1], z = -lt.pos[2];
                    const double len = std::sqrt(x * x + y * y + z * z), inv = 1. / len;
                    const Vec dir{x * inv, y * inv, z * inv};
                    // an orthonormal pair perpendicular to dir
                    Vec t = std::fabs(dir.x) <= std::fabs(dir.y) && std::fabs(dir.x) <= std::fabs(dir.z) ? Vec{1, 0, 0}
                          : (std::fabs(dir.y) <= std::fabs(dir.z) ? Vec{0, 1, 0} : Vec{0, 0, 1});
                    Vec e1 = vcross(dir, t);
                    const double l1 = vlen(e1);
                    e1 = {e1.x / l1, e1.y / l1, e1.z / l1};
                    const Vec e2 = vcross(dir, e1);
                    const bool usable = std::isfinite(e1.x) && std::isfinite(e1.y) && std::isfinite(e1.z) && std::isfinite(e2.x) &&
                                        std::isfinite(e2.y) && std::isfinite(e2.z);
                    double* F = &frame[8 * l];
                    F[0] = e1.x; F[1] = e1.y; F[2] = e1.z; F[3] = e2.x; F[4] = e2.y; F[5] = e2.z; F[6] = 0.0; F[7] = 0.0;
                    // |coordinates| <= half; rounding of the expanded form, see filter_range
                    const double rmax = half * (1 + 1e-3);
                    const double slack = 16.0 * 1.1920928955078125e-7 * rmax * rmax;
                    parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
                        for (size_t k = k0; k < k1; k++) {
                            const Vec ctr{sphere[4 * k], sphere[4 * k + 1], sphere[4 * k + 2]};
                            const double cu = vdot(ctr, e1), cv = vdot(ctr, e2), rho = sphere[4 * k + 3];
                            double kk = rho * rho + 3.0 * 1.1920928955078125e-7 * rho * rmax - cu * cu - cv * cv + slack;
                            if (!usable || !std::isfinite(kk)) kk = INFINITY;
                            B[4 * k] = (float)(-2.0 * cu); B[4 * k + 1] = (float)(-2.0 * cv);
                            B[4 * k + 2] = (kk == INFINITY) ? INFINITY : (float)kk + std::fabs((float)kk) * 1e-6f;
                            B[4 * k + 3] = 0.f;
                            if (!std::isfinite(B[4 * k]) || !std::isfinite(B[4 * k + 1])) { B[4 * k] = B[4 * k + 1] = 0.f; B[4 * k + 2] = INFINITY; }
                        }
                    });
                }
            }
            HIP_TRY(c->bound_lights.upload(bound));
            HIP_TRY(c->light_frame.upload(frame));
        }
        std::vector<int32_t> row(std::max<size_t>(c->h_lights.size(), 1), -1);
        std::vector<float> rinv;
        int n_rows = 0;
        for (size_t l = 0; l < c->h_lights.size(); l++) {
            const DevLight& lt = c->h_lights[l];
            if (lt.is_point) continue;
            // Light::lightRay: direction = (-pos).normalize() (raytracer.cpp:113-117, algebra3.h:638-662)
            const double x = -lt.pos[0], y = -lt.pos[1], z = -lt.pos[2];
            const double len = std::sqrt(x * x + y * y + z * z), inv = 1. / len;
            const Vec dir{x * inv, y * inv, z * inv};
            row[l] = n_rows++;
            rinv.resize((size_t)n_rows * ((size_t)NR + 1), 0.f);
            float* out = &rinv[(size_t)row[l] * ((size_t)NR + 1)];
            parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
                for (size_t k = k0; k < k1; k++) {
                    const TriExact& ex = exact[k];
                    out[k] = (float)(1.0 / (ex.n[0] * dir.x + ex.n[1] * dir.y + ex.n[2] * dir.z));
                }
            });
        }
        // ---- one level up: clusters of 64 consecutive slots ----
        {
            const size_t NC = (size_t)NR / 64, NL = c->h_lights.size();
            c->n_clusters = (int)NC;
            std::vector<float> cl_sphere(4 * (NC + 1), 0.f), cl_normal(4 * (NC + 1), 0.f), cl_cam(4 * (NC + 1), 0.f);
            std::vector<float> cl_lights(std::max<size_t>(NL, 1) * (NC + 1) * 4, 0.f);
            std::vector<int32_t> graze_slots, graze_start(std::max<size_t>(NL, 1) * (NC + 1), 0);
            std::vector<double> csph(4 * (NC + 1), 0.0);
            // which group a cluster belongs to (for the eye in that group's space)
            std::vector<int> cl_group(NC + 1, 0);
            for (size_t gi = 0; gi < groups.size(); gi++)
                for (int k = groups[gi].fax[0] / 64; k < groups[gi].fax[3] / 64; k++) cl_group[(size_t)k] = (int)gi;
            parallel_ranges(NC, host_parts(NC * 64), [&](size_t, size_t c0, size_t c1) {
                for (size_t cl = c0; cl < c1; cl++) {
                    // sphere: centre = mean of the members' centres, radius = what holds every member's sphere
                    double mx = 0, my = 0, mz = 0, nx = 0, ny = 0, nz = 0;
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++) {
                        if (exact[k].hole) continue;
                        mx += sphere[4 * k]; my += sphere[4 * k + 1]; mz += sphere[4 * k + 2];
                        nx += filt[k][0]; ny += filt[k][1]; nz += filt[k][2];
                    }
                    int members = 0;
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++) members += exact[k].hole ? 0 : 1;
                    mx /= members; my /= members; mz /= members;             // (a cluster holds at least one triangle)
                    double rad = 0;
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++) {
                        if (exact[k].hole) continue;
                        const double dx = sphere[4 * k] - mx, dy = sphere[4 * k + 1] - my, dz = sphere[4 * k + 2] - mz;
                        // (the radius of the triangle-level test carries the rounding slack of its expanded form)
                        const double rk = std::sqrt((double)sphere_any[4 * k + 3] + (double)sphere_any[4 * k] * sphere_any[4 * k] +
                                                    (double)sphere_any[4 * k + 1] * sphere_any[4 * k + 1] +
                                                    (double)sphere_any[4 * k + 2] * sphere_any[4 * k + 2]);
                        rad = std::max(rad, std::sqrt(dx * dx + dy * dy + dz * dz) + std::max(sphere[4 * k + 3], rk));
                    }
                    rad = rad * (1 + 1e-5) + sphere_pad;
                    const bool fin = std::isfinite(mx) && std::isfinite(my) && std::isfinite(mz) && std::isfinite(rad);
                    csph[4 * cl] = fin ? mx : 0; csph[4 * cl + 1] = fin ? my : 0; csph[4 * cl + 2] = fin ? mz : 0;
                    csph[4 * cl + 3] = fin ? rad : INFINITY;
                    for (int q = 0; q < 4; q++) cl_sphere[4 * cl + q] = (float)csph[4 * cl + q];
                    if (std::isfinite(cl_sphere[4 * cl + 3])) cl_sphere[4 * cl + 3] *= 1.0f + 1e-6f;
                    // normal cone: mean normal and the sine of the largest angle a member's normal makes with it
                    const double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
                    double sd = 1.0;
                    if (nl > 1e-6 && std::isfinite(nl)) {
                        nx /= nl; ny /= nl; nz /= nl;
                        double cmin = 1.0;
                        for (size_t k = 64 * cl; k < 64 * cl + 64; k++) {
                        if (exact[k].hole) continue;
                            const double len = std::sqrt((double)filt[k][0] * filt[k][0] + (double)filt[k][1] * filt[k][1] + (double)filt[k][2] * filt[k][2]);
                            cmin = std::min(cmin, (filt[k][0] * nx + filt[k][1] * ny + filt[k][2] * nz) / len);
                        }
                        sd = cmin > 0 ? std::sqrt(std::max(0.0, 1.0 - cmin * cmin)) + 1e-5 : 1.0;
                        if (!std::isfinite(sd)) sd = 1.0;
                    } else { nx = 0; ny = 0; nz = 1; }
                    cl_normal[4 * cl] = (float)nx; cl_normal[4 * cl + 1] = (float)ny; cl_normal[4 * cl + 2] = (float)nz;
                    cl_normal[4 * cl + 3] = (float)std::min(1.0, sd);
                    // camera rays: the cone from the eye (in the group's space); "every direction" when the eye
                    // lies in a member's plane (the in-plane-origin case ignores where the triangle is)
                    const Group& G = groups[(size_t)cl_group[cl]];
                    const Vec o = G.identity ? eye : affine_point(&c->h_xf[48 * (size_t)G.xf + 16], eye);
                    bounding_cone(o, {csph[4 * cl], csph[4 * cl + 1], csph[4 * cl + 2]}, csph[4 * cl + 3], &cl_cam[4 * cl]);
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++)
                        if (!exact[k].hole && !(std::fabs(eye_a[k]) > G.q[0].a_zero)) cl_cam[4 * cl + 3] = -2.f;
                }
            });
            cl_cam[4 * NC + 3] = -2.f;
            const double half = c->scene_bound;
            for (size_t l = 0; l < NL; l++) {
                const DevLight& lt = c->h_lights[l];
                float* B = &cl_lights[l * (NC + 1) * 4];
                int32_t* GS = &graze_start[l * (NC + 1)];
                const Vec Lp{lt.pos[0], lt.pos[1], lt.pos[2]};
                const double x = -lt.pos[0], y = -lt.pos[1], z = -lt.pos[2];
                const double len = std::sqrt(x * x + y * y + z * z), inv = 1. / len;
                const Vec dir{x * inv, y * inv, z * inv};
                // the light's plane coordinates as the per-triangle table has them
                Vec e1{0, 0, 0}, e2{0, 0, 0};
                bool usable = false;
                if (!lt.is_point) {
                    Vec t = std::fabs(dir.x) <= std::fabs(dir.y) && std::fabs(dir.x) <= std::fabs(dir.z) ? Vec{1, 0, 0}
                          : (std::fabs(dir.y) <= std::fabs(dir.z) ? Vec{0, 1, 0} : Vec{0, 0, 1});
                    e1 = vcross(dir, t);
                    const double l1 = vlen(e1);
                    e1 = {e1.x / l1, e1.y / l1, e1.z / l1};
                    e2 = vcross(dir, e1);
                    usable = std::isfinite(e1.x) && std::isfinite(e1.y) && std::isfinite(e1.z) && std::isfinite(e2.x) &&
                             std::isfinite(e2.y) && std::isfinite(e2.z);
                }
                const double rmax = half * (1 + 1e-3);
                const double slack = 16.0 * 1.1920928955078125e-7 * rmax * rmax;
                const double l1norm = std::fabs(Lp.x) + std::fabs(Lp.y) + std::fabs(Lp.z);
                for (size_t cl = 0; cl < NC; cl++) {
                    const Vec ctr{csph[4 * cl], csph[4 * cl + 1], csph[4 * cl + 2]};
                    const double rho = csph[4 * cl + 3];
                    // Graze list: what the plane guard of any task could name.  Its limit is b_hi * (largest distance
                    // from the task's origins to the light) + 2 a_zero + rounding for a point light -- origins lie
                    // within the scene's bound of the origin of coordinates --, b_hi for a directional light; the
                    // kernel evaluates the same quantities in FP32, whose rounding is allowed for here.
                    GS[cl] = (int32_t)graze_slots.size();
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++) {
                        if (exact[k].hole) continue;
                        const double nxk = filt[k][0], nyk = filt[k][1], nzk = filt[k][2], d0 = filt[k][3];
                        const double o = lt.is_point ? std::fabs(d0 - (nxk * Lp.x + nyk * Lp.y + nzk * Lp.z))
                                                     : std::fabs(nxk * dir.x + nyk * dir.y + nzk * dir.z);
                        const GroupFilter& q1 = c->h_q1[0];       // (shadow thresholds are the same for every transform)
                        const double far = std::sqrt(3.0) * half + vlen(Lp);
                        const double limit = lt.is_point ? (double)q1.b_hi * far * (1 + 1e-3) + 2.0 * q1.a_zero + 1e-5 * (l1norm + half) + 1e-6
                                                         : (double)q1.b_hi * (1 + 2e-3) + 1e-5;
                        if (!(o > limit)) graze_slots.push_back((int32_t)k);
                    }
                    if (lt.is_point) {
                        bounding_cone(Lp, ctr, rho, &B[4 * cl]);
                    } else {
                        const double cu = vdot(ctr, e1), cv = vdot(ctr, e2);
                        double kk = rho * rho + 3.0 * 1.1920928955078125e-7 * rho * rmax - cu * cu - cv * cv + slack;
                        if (!usable || !std::isfinite(kk)) kk = INFINITY;
                        B[4 * cl] = (float)(-2.0 * cu); B[4 * cl + 1] = (float)(-2.0 * cv);
                        B[4 * cl + 2] = (kk == INFINITY) ? INFINITY : (float)kk + std::fabs((float)kk) * 1e-6f;
                        B[4 * cl + 3] = 0.f;
                        if (!std::isfinite(B[4 * cl]) || !std::isfinite(B[4 * cl + 1])) { B[4 * cl] = B[4 * cl + 1] = 0.f; B[4 * cl + 2] = INFINITY; }
                    }
                }
            }
            HIP_TRY(c->cl_sphere.upload(cl_sphere)); HIP_TRY(c->cl_normal.upload(cl_normal)); HIP_TRY(c->cl_camera.upload(cl_cam));
            for (size_t l = 0; l < NL; l++) graze_start[l * (NC + 1) + NC] = (int32_t)((l + 1 < NL) ? graze_start[(l + 1) * (NC + 1)] : (int32_t)graze_slots.size());
            c->n_graze = graze_slots.size();
            graze_slots.push_back(0);                        // (never empty)
            HIP_TRY(c->cl_lights.upload(cl_lights)); HIP_TRY(c->graze_slots.upload(graze_slots)); HIP_TRY(c->graze_start.upload(graze_start));
        }
        HIP_TRY(c->eye_a.upload(eye_a));
        HIP_TRY(c->cone_camera.upload(cone_cam));
        HIP_TRY(c->sphere_any.upload(sphere_any)); HIP_TRY(c->plane_any.upload(plane_any));
        HIP_TRY(c->light_dir_row.upload(row));
        if (n_rows > 0) HIP_TRY(c->dir_rinv.upload(rinv)); else c->dir_rinv.release();
    }

    HIP_TRY(c->tri_filt.upload(filt)); HIP_TRY(c->tri_exact.upload(exact));
    HIP_TRY(c->groups.upload(groups)); HIP_TRY(c->obj_slot.upload(slot));
    DevScene& s = c->hs;
    s.tri_filt = c->tri_filt.p; s.tri_exact = c->tri_exact.p; s.groups = c->groups.p; s.obj_slot = c->obj_slot.p;
    s.n_regular = NR; s.n_irregular = NI; s.n_groups = (int)groups.size();
    s.n_prims_nominal = c->n_regular_real + NI + s.n_spheres;
    s.robust_filter = robust ? 1 : 0;
    s.n_void_normal = n_void_normal;
    c->n_void_normal = n_void_normal;
    std::vector<DevScene> one(1, s);
    HIP_TRY(c->d_scene.upload(one));
    c->robust = robust ? 1 : 0;
    return 0;
}

}  // namespace

static int upload_scene(rt_ctx** out, const rt_scene_desc* d) {
    if (!out || !d) return fail(RT_ERR_ARG, "rt_upload_scene: null argument");
    if (d->struct_size != sizeof(rt_scene_desc)) return fail(RT_ERR_ARG, "rt_upload_scene: struct_size mismatch");
    if (d->width <= 0 || d->height <= 0 || d->max_depth < 0 || d->max_depth > 64)
        return fail(RT_ERR_ARG, "rt_upload_scene: bad width/height/max_depth");
    if ((long long)d->width * d->height > (1ll << 30)) return fail(RT_ERR_ARG, "rt_upload_scene: frame too large");
    if (d->n_objects < 0 || d->n_lights < 0 || d->n_materials < 0 || d->n_transforms < 0)
        return fail(RT_ERR_ARG, "rt_upload_scene: negative count");
    const int N = d->n_objects;
    if (N > 0 && (!d->obj_kind || !d->obj_material || !d->obj_transform || !d->obj_interp || !d->obj_geom ||
                  !d->materials || !d->transforms))
        return fail(RT_ERR_ARG, "rt_upload_scene: null object array");
    if (d->n_lights > 0 && !d->lights) return fail(RT_ERR_ARG, "rt_upload_scene: null lights");
    for (int i = 0; i < N; i++) {
        if (d->obj_kind[i] != 0 && d->obj_kind[i] != 1) return fail(RT_ERR_ARG, "rt_upload_scene: bad obj_kind");
        if (d->obj_material[i] < 0 || d->obj_material[i] >= d->n_materials)
            return fail(RT_ERR_ARG, "rt_upload_scene: material index out of range");
        if (d->obj_transform[i] < 0 || d->obj_transform[i] >= d->n_transforms)
            return fail(RT_ERR_ARG, "rt_upload_scene: transform index out of range");
        if (d->obj_interp[i] && !d->obj_normals) return fail(RT_ERR_ARG, "rt_upload_scene: obj_normals missing");
    }

    // Every transform must be affine: the bounds and margins behind the FP32 filter are derived
    // without a perspective divide (what parseSceneFile can build -- translations, rotations,
    // scalings and their products, raytracer.cpp:806-824 -- always is).
    for (int t = 0; t < d->n_transforms; t++) {
        const double* rows[2] = {d->transforms[t].m + 12, d->transforms[t].inv + 12};
        for (const double* r : rows)
            if (!(r[0] == 0.0 && r[1] == 0.0 && r[2] == 0.0 && r[3] == 1.0))
                return fail(RT_ERR_ARG, "rt_upload_scene: transform " + std::to_string(t) +
                                            " is not affine (the last row of m and of inv must be 0 0 0 1)");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RT_ERR_DEVICE, "rt_upload_scene: no HIP device (the trace loop has no CPU fallback)");
    rt_ctx* c = new rt_ctx();
    auto bail = [&](int code) { rt_free(c); return code; };
    hipDeviceProp_t prop;
    if (hipGetDevice(&c->device) != hipSuccess || hipGetDeviceProperties(&prop, c->device) != hipSuccess)
        return bail(fail(RT_ERR_DEVICE, "rt_upload_scene: cannot query the current device"));
    c->n_cu = prop.multiProcessorCount;
    c->clock_khz = prop.clockRate;

    const int NX = d->n_transforms;
    // RT_UPLOAD_TIMING in the environment: print how long each phase of the upload took (stderr)
    const bool timing = std::getenv("RT_UPLOAD_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "rt_upload_scene: %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    // ---- transforms ----
    std::vector<double> xf(48 * (size_t)NX);
    std::vector<int32_t> xf_ident(NX);
    for (int t = 0; t < NX; t++) {
        std::memcpy(&xf[48 * (size_t)t], d->transforms[t].m, 16 * 8);
        std::memcpy(&xf[48 * (size_t)t + 16], d->transforms[t].inv, 16 * 8);
        std::memcpy(&xf[48 * (size_t)t + 32], d->transforms[t].inv_t, 16 * 8);
        int ident = 1;
        for (int k = 0; k < 48; k++) {
            double want = ((k % 16) % 5 == 0) ? 1.0 : 0.0;
            if (std::memcmp(&xf[48 * (size_t)t + k], &want, 8) != 0) ident = 0;
        }
        xf_ident[t] = ident;
    }

    // ---- world bounds: every point a ray can start from ----
    Box world;
    world.add({d->cam_eye[0], d->cam_eye[1], d->cam_eye[2]});
    std::vector<double> xf_extent(std::max(NX, 1), 0.0);   // largest |coordinate| of a transform's object-space vertices
    {
        const size_t parts = host_parts((size_t)N);
        std::vector<Box> part_box(parts);
        std::vector<std::vector<double>> part_extent(parts, std::vector<double>(std::max(NX, 1), 0.0));
        parallel_ranges((size_t)N, parts, [&](size_t part, size_t i0, size_t i1) {
            Box& box = part_box[part];
            std::vector<double>& extent = part_extent[part];
            for (size_t i = i0; i < i1; i++) {
                const double* g = d->obj_geom + 9 * i;
                const double* M = &xf[48 * (size_t)d->obj_transform[i]];
                if (d->obj_kind[i] == 0) {
                    for (int k = 0; k < 3; k++) {
                        Vec v{g[3 * k], g[3 * k + 1], g[3 * k + 2]};
                        box.add(affine_point(M, v));
                        double& ext = extent[d->obj_transform[i]];
                        ext = std::max(ext, max_abs(v));
                    }
                } else {
                    double r = std::fabs(g[3]);
                    for (int k = 0; k < 8; k++) {
                        Vec v{g[0] + ((k & 1) ? r : -r), g[1] + ((k & 2) ? r : -r), g[2] + ((k & 4) ? r : -r)};
                        box.add(affine_point(M, v));
                    }
                }
            }
        });
        for (size_t part = 0; part < parts; part++) {
            for (int a = 0; a < 3; a++) {
                world.lo[a] = std::min(world.lo[a], part_box[part].lo[a]);
                world.hi[a] = std::max(world.hi[a], part_box[part].hi[a]);
            }
            for (int t = 0; t < NX; t++) xf_extent[t] = std::max(xf_extent[t], part_extent[part][t]);
        }
    }

    lap("bounds");
    // ---- per transform and kind of query: the magnitudes the filter's operands can have ----
    // Nearest-hit rays reach a group in its object space (raytracer.cpp:251-259): origins are the
    // world box under M^-1, camera directions are unnormalised, later directions have unit
    // length in world space.  Shadow rays stay in world space and meet the untransformed
    // vertices (raytracer.cpp:553); their directions are normalised (raytracer.cpp:108-119).
    auto clamp_big = [](double x, double cap) { return x < cap ? x : cap; };   // NaN -> cap
    auto widen = [](double x) { return x * (1.0 + 1e-3) + 1e-3; };
    Vec cam_dir[4];
    double cam_len = 1.0;
    {
        Vec eye{d->cam_eye[0], d->cam_eye[1], d->cam_eye[2]};
        const double* corners[4] = {d->cam_ll, d->cam_lr, d->cam_ul, d->cam_ur};
        for (int k = 0; k < 4; k++) {
            cam_dir[k] = vsub({corners[k][0], corners[k][1], corners[k][2]}, eye);
            cam_len = std::max(cam_len, vlen(cam_dir[k]));
        }
    }
    double world_ob = 0.0;
    for (int k = 0; k < 8; k++) world_ob = std::max(world_ob, max_abs(world.corner(k)));
    world_ob = widen(clamp_big(world_ob, 1e30));
    struct Mag { double ob, vm, R; };
    std::vector<Mag> mag0(std::max(NX, 1)), mag1(std::max(NX, 1));
    double R_max = std::sqrt(3.0) * world_ob;
    for (int t = 0; t < NX; t++) {
        double ob = world_ob, vm = cam_len;
        if (!xf_ident[t]) {
            const double* I = &xf[48 * (size_t)t + 16];
            ob = 0.0;
            for (int k = 0; k < 8; k++) ob = std::max(ob, max_abs(affine_point(I, world.corner(k))));
            ob = widen(clamp_big(ob, 1e30));
            double fro = 0;
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) fro += I[4 * r + cc] * I[4 * r + cc];
            vm = std::max(1.0, std::sqrt(fro));
            for (int k = 0; k < 4; k++) vm = std::max(vm, vlen(linear_dir(I, cam_dir[k])));
        }
        vm = clamp_big(vm, 1e15) * (1.0 + 1e-3);
        const double ext = widen(clamp_big(xf_extent[t], 1e30));
        mag0[t] = {ob, vm, std::sqrt(3.0) * std::max(ob, ext)};
        mag1[t] = {world_ob, 1.0 + 1e-3, std::sqrt(3.0) * std::max(world_ob, ext)};
        R_max = std::max(R_max, std::max(mag0[t].R, mag1[t].R));
    }

    // ---- filter thresholds (every constant is rounded up) ----
    const double EPSF = 1.1920928955078125e-7;   // 2^-23
    // amp_out: position-error amplification of the most grazing ray the reference still accepts
    auto thresholds = [&](float eps, const Mag& m, double* amp_out) {
        GroupFilter f{};
        const double Eb = 8 * EPSF * m.vm, Ea = 16 * EPSF * m.R;
        double b_lo = (double)eps * (1 - 1e-3) - Eb;
        double r_lo = (double)eps * (1 - 1e-3) - 4 * (Ea + (double)eps * Eb) / std::max(b_lo, 1e-30);
        if (!(b_lo > 0)) { b_lo = 0; r_lo = -FLT_MAX; }      // nothing is certain about grazing rays
        f.r_lo = (float)std::max(r_lo, -(double)FLT_MAX);
        f.a_zero = (float)(2 * Ea * (1 + 1e-6));
        f.b_hi = (float)(((double)std::fabs(eps) + Eb) * (1 + 1e-3));
        f.e_a = (float)(Ea * (1 + 1e-6)); f.e_b = (float)(Eb * (1 + 1e-6));
        f.r_hi = (float)((double)std::fabs(eps) * (1 + 1e-3));
        f.origin_bound = (float)(m.ob * (1 - 1e-6));
        f.dir_len2_bound = (float)(m.vm * m.vm * (1 - 1e-6));
        *amp_out = 18.0 * m.vm / std::max(b_lo, 1e-30) + 20.0;
        return f;
    };
    c->h_q0.resize(std::max(NX, 1)); c->h_q1.resize(std::max(NX, 1));
    std::vector<double> amp0(std::max(NX, 1)), amp1(std::max(NX, 1));
    for (int t = 0; t < NX; t++) {
        c->h_q0[t] = thresholds(d->eps, mag0[t], &amp0[t]);
        c->h_q1[t] = thresholds(d->shadow_eps, mag1[t], &amp1[t]);
    }

    // ---- triangles: reference-precision records and both forms of the filter record ----
    std::vector<HostTri>& tris = c->h_tris;
    std::vector<SphereRec> sph;
    std::vector<SphereFilt> sph_filt;
    c->h_slot.assign(N, -1);
    std::vector<int> tri_objects;            // Scene::objects indices of the triangles, in order
    for (int i = 0; i < N; i++) {
        const double* g = d->obj_geom + 9 * (size_t)i;
        if (d->obj_kind[i] == 1) {
            SphereRec s{};
            s.c[0] = g[0]; s.c[1] = g[1]; s.c[2] = g[2];
            float radius = (float)g[3];
            s.r2 = (double)radius * (double)radius;
            s.obj = i; s.xf = d->obj_transform[i]; s.identity = xf_ident[s.xf];
            c->h_slot[i] = (int32_t)sph.size();
            sph.push_back(s);
            SphereFilt f{};
            const double* I = &xf[48 * (size_t)s.xf + 16];
            for (int r = 0; r < 3; r++) {
                double row = 0;
                for (int q = 0; q < 4; q++) { f.m[4 * r + q] = (float)I[4 * r + q]; if (q < 3) row += std::fabs(I[4 * r + q]); }
                f.norm = std::max(f.norm, (float)(row * (1 + 1e-6)));
                f.tabs += (float)(std::fabs(I[4 * r + 3]) * (1 + 1e-6));
            }
            for (int q = 0; q < 3; q++) { f.c[q] = (float)g[q]; f.tabs += (float)(std::fabs(g[q]) * (1 + 1e-6)); }
            f.r2 = (float)s.r2;
            f.identity = s.identity;
            // A ball around the sphere's image in world space: centre M c, radius r * |M| with the 3 x 3 part's spectral
            // norm bounded by sqrt(|M|_1 |M|_inf).  The rays meet the sphere through the inverse the reference computed
            // (raytracer.cpp:246-247), so M must be that matrix's inverse to well within the test's slack; if not (a
            // transform close to singular), no ball: the sphere is never dropped for a whole wave.
            {
                const double* M = &xf[48 * (size_t)s.xf];
                double col[3] = {0, 0, 0}, n1 = 0, ninf = 0, off = 0;
                for (int r = 0; r < 3; r++) {
                    double row = 0;
                    for (int q = 0; q < 3; q++) { row += std::fabs(M[4 * r + q]); col[q] += std::fabs(M[4 * r + q]); }
                    ninf = std::max(ninf, row);
                    f.wc[r] = (float)(M[4 * r] * g[0] + M[4 * r + 1] * g[1] + M[4 * r + 2] * g[2] + M[4 * r + 3]);
                    for (int q = 0; q < 4; q++) {
                        double e = (r == q) ? -1.0 : 0.0;
                        for (int k = 0; k < 4; k++) e += I[4 * r + k] * M[4 * k + q];
                        off = std::max(off, std::fabs(e));
                    }
                }
                for (int q = 0; q < 3; q++) n1 = std::max(n1, col[q]);
                const double scale = s.identity ? 1.0 : std::sqrt(n1 * ninf) * (1 + 1e-6);
                f.wr = (off < 1e-6) ? (float)(std::fabs((double)radius) * scale * (1 + 1e-6)) : INFINITY;      // (false for NaNs: no ball)
                if (s.identity) { f.wc[0] = f.c[0]; f.wc[1] = f.c[1]; f.wc[2] = f.c[2]; }
            }
            // (a transform or sphere whose floats are not finite: the pre-test keeps every pair, NaNs compare false)
            sph_filt.push_back(f);
        } else {
            tri_objects.push_back(i);
        }
    }
    tris.assign(tri_objects.size(), HostTri{});
    // one record per triangle, independent of the others: large meshes are spread over host threads
    auto build_records = [&](size_t k_begin, size_t k_end) {
      for (size_t k_tri = k_begin; k_tri < k_end; k_tri++) {
        const int i = tri_objects[k_tri];
        const double* g = d->obj_geom + 9 * (size_t)i;
        HostTri t{};
        t.xf = d->obj_transform[i];
        Vec v1{g[0], g[1], g[2]}, v2{g[3], g[4], g[5]}, v3{g[6], g[7], g[8]};
        Vec e1 = vsub(v3, v1), e2 = vsub(v2, v1);                  // raytracer.cpp:288-289
        Vec cr = vcross(e2, e1);
        double len = std::sqrt(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
        double inv = 1. / len;                                      // algebra3.h:638-662
        Vec n{cr.x * inv, cr.y * inv, cr.z * inv};
        float uu = (float)vdot(e1, e1), uv = (float)vdot(e1, e2), vv = (float)vdot(e2, e2);
        float D = uv * uv - uu * vv;                                // raytracer.cpp:340 (float arithmetic)
        TriExact& ex = t.ex;
        ex.v1[0] = v1.x; ex.v1[1] = v1.y; ex.v1[2] = v1.z;
        ex.e1[0] = e1.x; ex.e1[1] = e1.y; ex.e1[2] = e1.z;
        ex.e2[0] = e2.x; ex.e2[1] = e2.y; ex.e2[2] = e2.z;
        ex.n[0] = n.x; ex.n[1] = n.y; ex.n[2] = n.z;
        ex.uu = uu; ex.uv = uv; ex.vv = vv; ex.D = D; ex.obj = i;
        bool ok = std::isfinite(n.x) && std::isfinite(n.y) && std::isfinite(n.z) && std::isfinite(uu) &&
                  std::isfinite(uv) && std::isfinite(vv) && std::isfinite(D) && D != 0.0f;
        if (ok) {
            // s = S.(x - v1), t = T.(x - v1) are the reference's barycentrics (raytracer.cpp:344, 348)
            // as linear functions, built from ITS rounded uu, uv, vv, D.
            double fuu = uu, fuv = uv, fvv = vv, fD = D;
            Vec S{(fuv * e2.x - fvv * e1.x) / fD, (fuv * e2.y - fvv * e1.y) / fD, (fuv * e2.z - fvv * e1.z) / fD};
            Vec T{(fuv * e1.x - fuu * e2.x) / fD, (fuv * e1.y - fuu * e2.y) / fD, (fuv * e1.z - fuu * e2.z) / fD};
            double le1 = std::sqrt(fuu), le2 = std::sqrt(fvv);
            double wmax = 1.5 * (le1 + le2);
            // rounding of the reference's own float formula
            double dBs = 2 * EPSF * wmax * (std::fabs(fuv) * le2 + fvv * le1) / std::fabs(fD) + 3 * EPSF;
            double dBt = 2 * EPSF * wmax * (std::fabs(fuv) * le1 + fuu * le2) / std::fabs(fD) + 3 * EPSF;
            ok = dBs < 0.125 && dBt < 0.125 && std::isfinite(dBs) && std::isfinite(dBt);
            // a point the reference accepts lies within (dBs, dBt) barycentric units of the triangle
            ex.box_pad = (float)(4.0 * (dBs + dBt) * (le1 + le2));
            if (ok) {
                // Each coordinate as a function of the two hit-point coordinates (u, v) other than
                // the normal's dominant axis k: on the plane x_k = (d0 - n_u x_u - n_v x_v) / n_k.
                double d0 = vdot(n, v1);
                double nn[3] = {n.x, n.y, n.z};
                int k = 0;
                if (std::fabs(nn[1]) > std::fabs(nn[k])) k = 1;
                if (std::fabs(nn[2]) > std::fabs(nn[k])) k = 2;
                const int iu = (k + 1) % 3, iv = (k + 2) % 3;
                t.axis = k;
                Vec U{-(S.x + T.x), -(S.y + T.y), -(S.z + T.z)};
                const double off3[3] = {-vdot(S, v1), -vdot(T, v1), 1.0 - vdot(U, v1)};
                const Vec* planes[3] = {&S, &T, &U};
                double coef[3][3];
                double norm2d[3];
                for (int q = 0; q < 3; q++) {
                    double pc[3] = {planes[q]->x, planes[q]->y, planes[q]->z};
                    coef[q][0] = pc[iu] - pc[k] * nn[iu] / nn[k];
                    coef[q][1] = pc[iv] - pc[k] * nn[iv] / nn[k];
                    coef[q][2] = off3[q] + pc[k] * d0 / nn[k];
                    norm2d[q] = std::sqrt(coef[q][0] * coef[q][0] + coef[q][1] * coef[q][1]);
                }
                const Mag& m0 = mag0[t.xf];
                const Mag& m1 = mag1[t.xf];
                for (int form = 0; form < 2; form++) {
                    float* f = form ? t.f_robust : t.f_static;
                    f[0] = (float)n.x; f[1] = (float)n.y; f[2] = (float)n.z; f[3] = (float)d0;
                    double ds, dt;
                    if (form == 0) {
                        // rounding of the FP32 filter itself (hit position and plane evaluation), worst
                        // case over both kinds of query and every ray
                        const double RA = std::max(m0.R * amp0[t.xf], m1.R * amp1[t.xf]);
                        ds = 4 * (EPSF * norm2d[0] * RA + dBs) + 1e-6;
                        dt = 4 * (EPSF * norm2d[1] * RA + dBt) + 1e-6;
                        f[13] = f[14] = 0.f;
                    } else {
                        // the same bound with the ray's own 1/|n.dir| in place of the worst case:
                        // 18 vm/|b| + 20 <= 40 vm/|b| because |b| <= vm; the factor is applied per test
                        ds = 4 * dBs + 1e-6;
                        dt = 4 * dBt + 1e-6;
                        const double nsum = norm2d[0] + norm2d[1];
                        f[13] = (float)(4 * EPSF * nsum * m0.R * 42.0 * m0.vm);
                        f[14] = (float)(4 * EPSF * nsum * m1.R * 42.0 * m1.vm);
                    }
                    const double du = ds + dt;
                    bool fok = std::isfinite(ds) && std::isfinite(dt) && du < 1e6;
                    f[4] = (float)coef[0][0]; f[5] = (float)coef[0][1]; f[6] = (float)(coef[0][2] + ds);
                    f[7] = (float)coef[1][0]; f[8] = (float)coef[1][1]; f[9] = (float)(coef[1][2] + dt);
                    f[10] = (float)coef[2][0]; f[11] = (float)coef[2][1]; f[12] = (float)(coef[2][2] + du);
                    f[15] = (float)(du * (1 + 1e-6));      // the largest margin folded into the offsets (LBVH pruning)
                    for (int q = 0; q < 16; q++) fok = fok && std::isfinite(f[q]);
                    (form ? t.ok_robust : t.ok_static) = fok;
                    if (form == 0) t.margin_static = (float)std::min(du, 1e30);
                }
            }
        }
        tris[k_tri] = t;
      }
    };
    parallel_ranges(tri_objects.size(), host_parts(tri_objects.size()),
                    [&](size_t, size_t k0, size_t k1) { build_records(k0, k1); });
    c->h_xf_ident = xf_ident;
    c->h_xf = xf;
    c->scene_bound = R_max;

    lap("triangle records");
    // ---- which form of the filter: the cheap static margins unless they are useless here ----
    {
        double sum = 0; size_t n_ok = 0, n_only_robust = 0;
        // plane -> number of triangles in it: open addressing over twice as many slots as triangles
        size_t n_slots = 16;
        while (n_slots < 2 * tris.size()) n_slots *= 2;
        std::vector<unsigned long long> slot_key(n_slots, 0);
        std::vector<int> slot_count(n_slots, 0);
        int largest_plane = 0;
        for (const HostTri& t : tris) {
            if (t.ok_robust && !t.ok_static) n_only_robust++;
            if (!t.ok_static) continue;
            sum += std::min((double)t.margin_static, 100.0); n_ok++;
            // triangles lying in one plane: a ray leaving that plane lists them all in the static form
            const float* f = t.f_static;
            unsigned long long key = (unsigned long long)(unsigned int)t.xf * 0x9E3779B97F4A7C15ull;
            for (int q = 0; q < 4; q++) {
                const double scale = q < 3 ? 2048.0 : 2048.0 / std::max(mag0[t.xf].R, 1e-30);
                key = (key ^ (unsigned long long)(long long)std::llround(f[q] * scale)) * 0x100000001B3ull;
            }
            key |= 1ull;                                    // 0 marks an empty slot
            size_t at = (size_t)(key >> 17) & (n_slots - 1);
            while (slot_key[at] != 0 && slot_key[at] != key) at = (at + 1) & (n_slots - 1);
            slot_key[at] = key;
            largest_plane = std::max(largest_plane, ++slot_count[at]);
        }
        const double mean_margin = n_ok ? sum / (double)n_ok : 0.0;
        c->auto_robust = (n_only_robust > 0 || mean_margin > 1.0 || largest_plane > 16) ? 1 : 0;
        // (2 % of a triangle's width per edge: the benchmark's teapot has 7 %, and its frame lists 21 M pairs with the
        // static margins, 14 M with the per-test ones, of which the reference accepts 11 M)
        c->auto_robust_bounded = (c->auto_robust || mean_margin > 0.02) ? 1 : 0;
        if (timing) fprintf(stderr, "  filter form: mean static margin %.4f of a triangle, %zu triangles only regular with per-test margins, largest plane %d triangles\n",
                            mean_margin, n_only_robust, largest_plane);
    }

    std::vector<DevMaterial> mats(d->n_materials);
    for (int i = 0; i < d->n_materials; i++) std::memcpy(&mats[i], &d->materials[i], sizeof(DevMaterial));
    std::vector<DevLight> lights(d->n_lights);
    for (int i = 0; i < d->n_lights; i++) std::memcpy(&lights[i], &d->lights[i], sizeof(DevLight));
    c->h_lights = lights;
    static_assert(sizeof(DevMaterial) == sizeof(rt_material), "material layout");
    static_assert(sizeof(DevLight) == sizeof(rt_light), "light layout");
    static_assert(sizeof(TriExact) == 128, "TriExact layout");

#define UP(buf, ...)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = c->buf.upload(__VA_ARGS__);                                                    \
        if (e_ != hipSuccess) return bail(fail(RT_ERR_DEVICE, std::string("upload " #buf ": ") + hipGetErrorString(e_))); \
    } while (0)
    // the per-object arrays go to the device straight from the caller's buffers
    const size_t n_obj = (size_t)N;
    UP(spheres, sph); UP(sphere_filt, sph_filt); UP(xforms, xf); UP(xf_identity, xf_ident);
    UP(obj_kind, d->obj_kind, n_obj); UP(obj_material, d->obj_material, n_obj);
    UP(obj_transform, d->obj_transform, n_obj); UP(obj_interp, d->obj_interp, n_obj);
    UP(obj_geom, d->obj_geom, 9 * n_obj); UP(obj_normals, d->obj_normals, 9 * n_obj);
    UP(materials, mats); UP(lights, lights);
#undef UP

    DevScene& s = c->hs;
    s.spheres = c->spheres.p; s.sphere_filt = c->sphere_filt.p;
    s.xforms = c->xforms.p; s.xf_identity = c->xf_identity.p; s.obj_kind = c->obj_kind.p;
    s.obj_material = c->obj_material.p; s.obj_transform = c->obj_transform.p; s.obj_interp = c->obj_interp.p;
    s.obj_geom = c->obj_geom.p; s.obj_normals = c->obj_normals.p; s.materials = c->materials.p; s.lights = c->lights.p;
    s.n_spheres = (int)sph.size();
    s.n_objects = N; s.n_lights = d->n_lights; s.n_transforms = NX;
    s.last_xf = N > 0 ? d->obj_transform[N - 1] : 0;
    s.width = d->width; s.height = d->height; s.max_depth = d->max_depth;
    s.eps = d->eps; s.shadow_eps = d->shadow_eps;
    std::memcpy(s.cam_eye, d->cam_eye, 24); std::memcpy(s.cam_ll, d->cam_ll, 24); std::memcpy(s.cam_lr, d->cam_lr, 24);
    std::memcpy(s.cam_ul, d->cam_ul, 24); std::memcpy(s.cam_ur, d->cam_ur, 24); std::memcpy(s.ambient, d->ambient, 24);
    // the shading ray of an empty transform table cannot occur: N > 0 implies NX > 0

    lap("form choice + small uploads");
    // triangle records, groups and the scene header, in the form chosen above
    if (int rc = install_filter(c, chosen_form(c))) return bail(rc);
    lap("install_filter");
    hipError_t e = c->own_frame.alloc((size_t)d->width * d->height);
    if (e != hipSuccess) return bail(fail(RT_ERR_DEVICE, std::string("frame: ") + hipGetErrorString(e)));
    c->frame = c->own_frame.p;
    e = hipMemset(c->frame, 0, (size_t)d->width * d->height * 4);
    if (e == hipSuccess) e = c->counters.alloc(1);
    if (e == hipSuccess) e = hipMemset(c->counters.p, 0, sizeof(DevCounters));
    if (e == hipSuccess) e = c->work.alloc(RT_WORK_KINDS * RT_WORK_FIELDS);
    if (e == hipSuccess) e = hipMemset(c->work.p, 0, RT_WORK_KINDS * RT_WORK_FIELDS * sizeof(unsigned long long));
    if (e != hipSuccess) return bail(fail(RT_ERR_DEVICE, std::string("init: ") + hipGetErrorString(e)));

    // launch geometry: as many resident blocks as the filter kernels admit
#define RT_GRID_OF(M, R, CN, RB, WK)                                                                               \
    do {                                                                                                            \
        int nb = 0;                                                                                                 \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_filter<M, R, RB, CN, WK>, RT_BLOCK, 0) != hipSuccess || nb < 1) nb = 2; \
        c->filter_grid[M][CN][RB ? 1 : 0][WK ? 1 : 0] = c->n_cu * std::min(nb, 8);                                  \
    } while (0)
#define RT_GRID_OF_ALL(M, R, CN) do { RT_GRID_OF(M, R, CN, false, false); RT_GRID_OF(M, R, CN, true, false); RT_GRID_OF(M, R, CN, false, true); RT_GRID_OF(M, R, CN, true, true); } while (0)
    RT_GRID_OF_ALL(0, RT_RPL, 0);
    RT_GRID_OF_ALL(1, RT_RPL, 0);
    RT_GRID_OF_ALL(0, RT_RPL_SHARED, 1);
    RT_GRID_OF_ALL(0, RT_RPL_ANY, 4);
    RT_GRID_OF_ALL(1, RT_RPL_SHARED, 2);
#undef RT_GRID_OF_ALL
#undef RT_GRID_OF
    c->aux_grid = c->n_cu * 8;
    if (const char* env = std::getenv("RAYTRACER_AMD_GRID_HINTS")) c->grid_hints = std::atoi(env) ? 1 : 0;
    if (const char* env = std::getenv("RAYTRACER_AMD_FOLD_BY_LEVEL")) c->fold_by_level = std::atoi(env) != 0;
    if (const char* env = std::getenv("RAYTRACER_AMD_LEAN_SCANS")) c->lean_scans = std::atoi(env) != 0;
    if (const char* env = std::getenv("RAYTRACER_AMD_CLEAR_EVERY_LAUNCH")) c->clear_every_launch = std::atoi(env) != 0;
    if (const char* env = std::getenv("RAYTRACER_AMD_SPHERE_CULL")) c->sphere_cull_min = std::max(0, std::atoi(env));
    if (const char* env = std::getenv("RAYTRACER_AMD_BESIDE_DIV")) c->beside_grid_div = std::max(1, std::min(16, std::atoi(env)));
    // this device's copy of the static workspace (for rt_render_tile_noalloc)
    if (int rc = resolve_scratch(c->device)) return bail(rc);
    *out = c;
    return 0;
}
