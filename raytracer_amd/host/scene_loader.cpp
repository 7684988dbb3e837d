// Host-side scene loader: `.scene` / `.obj` text -> flat rt_scene_desc.
//
// Accepts the grammar of the reference's parseSceneFile (raytracer.cpp:705-841)
// and parseObj (raytracer.cpp:624-703) and reproduces the arithmetic that
// decides the bits of what it emits:
//   * every number goes text -> atof -> float -> double (raytracer.cpp:585-596,
//     the parser's `float args[]`), so coordinates are float-representable;
//   * the current transform is the product of the stack in file order starting
//     from the identity (reduceTransforms, raytracer.cpp:616-622; product
//     algebra3.h:1181-1191);
//   * rotation is Rodrigues about the normalised axis by |axis| degrees with
//     the magnitude rounded to float (raytracer.cpp:40-42, 820; algebra3.h:1271);
//   * inverse is Gauss-Jordan with partial pivoting and reciprocal-multiply row
//     scaling (algebra3.h:1130-1161, 804-806), inverse-transpose its transpose.
// Built with -ffp-contract=off so no step is fused.
#include "rt_host.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <utility>
#include <exception>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

struct Mat4 { double a[4][4]; };

Mat4 mat_identity() {
    Mat4 m{};
    for (int i = 0; i < 4; i++) m.a[i][i] = 1.0;
    return m;
}
Mat4 mat_translate(double x, double y, double z) {
    Mat4 m = mat_identity();
    m.a[0][3] = x; m.a[1][3] = y; m.a[2][3] = z;
    return m;
}
Mat4 mat_scale(double x, double y, double z) {
    Mat4 m = mat_identity();
    m.a[0][0] = x; m.a[1][1] = y; m.a[2][2] = z;
    return m;
}
// Axis is normalised by multiplying with 1/length (algebra3.h:638-662).
Mat4 mat_rotate(double ax, double ay, double az, double angle_deg) {
    double rad = angle_deg * M_PI / 180.0;
    double c = std::cos(rad), s = std::sin(rad), t = 1.0 - c;
    double len = std::sqrt(ax * ax + ay * ay + az * az);
    double inv = 1. / len;
    ax *= inv; ay *= inv; az *= inv;
    Mat4 m{};
    m.a[0][0] = t * ax * ax + c;      m.a[0][1] = t * ax * ay - s * az; m.a[0][2] = t * ax * az + s * ay;
    m.a[1][0] = t * ax * ay + s * az; m.a[1][1] = t * ay * ay + c;      m.a[1][2] = t * ay * az - s * ax;
    m.a[2][0] = t * ax * az - s * ay; m.a[2][1] = t * ay * az + s * ax; m.a[2][2] = t * az * az + c;
    m.a[3][3] = 1.0;
    return m;
}
Mat4 mat_mul(const Mat4& p, const Mat4& q) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.a[i][j] = p.a[i][0] * q.a[0][j] + p.a[i][1] * q.a[1][j] + p.a[i][2] * q.a[2][j] + p.a[i][3] * q.a[3][j];
    return r;
}
Mat4 mat_transpose(const Mat4& p) {
    Mat4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.a[i][j] = p.a[j][i];
    return r;
}
// Returns false for a singular matrix (the reference asserts, algebra3.h:1148).
bool mat_inverse(const Mat4& src, Mat4* out) {
    Mat4 a = src, b = mat_identity();
    for (int j = 0; j < 4; j++) {
        int piv = j;
        for (int i = j + 1; i < 4; i++)
            if (std::fabs(a.a[i][j]) > std::fabs(a.a[piv][j])) piv = i;
        for (int k = 0; k < 4; k++) { std::swap(a.a[piv][k], a.a[j][k]); std::swap(b.a[piv][k], b.a[j][k]); }
        if (a.a[j][j] == 0.) return false;
        double binv = 1. / a.a[j][j];
        for (int k = 0; k < 4; k++) b.a[j][k] *= binv;
        double ainv = 1. / a.a[j][j];
        for (int k = 0; k < 4; k++) a.a[j][k] *= ainv;
        for (int i = 0; i < 4; i++) {
            if (i == j) continue;
            double fb = a.a[i][j];
            for (int k = 0; k < 4; k++) b.a[i][k] -= fb * b.a[j][k];
            double fa = a.a[i][j];
            for (int k = 0; k < 4; k++) a.a[i][k] -= fa * a.a[j][k];
        }
    }
    *out = b;
    return true;
}
bool same_bits(const void* p, const void* q, size_t n) { return std::memcmp(p, q, n) == 0; }

}  // namespace

// resize(n) of a vector with this allocator leaves new elements uninitialised (and their
// pages untouched), so the threads that fill a large array are the first to touch it.
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};

struct rth_scene {
    rt_scene_desc d{};
    std::vector<int32_t> kind, material, transform, interp;
    std::vector<double, NoInitAlloc<double>> geom, normals;
    std::vector<rt_material> materials;
    std::vector<rt_transform> transforms;
    std::vector<rt_light> lights;
    void seal() {
        d.struct_size = (uint32_t)sizeof(rt_scene_desc);
        d.n_objects = (int32_t)kind.size();
        d.n_materials = (int32_t)materials.size();
        d.n_transforms = (int32_t)transforms.size();
        d.n_lights = (int32_t)lights.size();
        d.obj_kind = kind.data(); d.obj_material = material.data();
        d.obj_transform = transform.data(); d.obj_interp = interp.data();
        d.obj_geom = geom.data(); d.obj_normals = normals.data();
        d.materials = materials.data(); d.transforms = transforms.data(); d.lights = lights.data();
    }
};

namespace {

// Reads up to `want` numbers the way the reference's parseArgs does
// (text -> atof -> float).  Missing values become 0 (the reference leaves
// them indeterminate) and surplus tokens are dropped (the reference overruns
// its array); both cases are reported on stderr like the reference does.
int read_floats(std::istringstream& iss, float* dst, int want, const char* what) {
    std::string tok;
    int n = 0;
    while (iss >> tok) {
        if (n < want) dst[n] = (float)atof(tok.c_str());
        n++;
    }
    for (int i = n; i < want; i++) dst[i] = 0.f;
    if (n != want) fprintf(stderr, "Incorrect number of arguments for %s\n", what);
    return n;
}

struct Builder {
    rth_scene* s;
    std::string dir;
    std::vector<Mat4> stack;
    rt_material cur_mat{};
    int cur_mat_id = -1, cur_xf_id = -1;

    int material_id() {
        if (cur_mat_id >= 0) return cur_mat_id;
        for (int k = (int)s->materials.size() - 1; k >= 0; k--)
            if (same_bits(&s->materials[k], &cur_mat, sizeof(rt_material))) return cur_mat_id = k;
        s->materials.push_back(cur_mat);
        return cur_mat_id = (int)s->materials.size() - 1;
    }
    int transform_id(bool* ok) {
        *ok = true;
        if (cur_xf_id >= 0) return cur_xf_id;
        Mat4 m = mat_identity();
        for (const Mat4& t : stack) m = mat_mul(m, t);
        for (int k = (int)s->transforms.size() - 1; k >= 0; k--)
            if (same_bits(s->transforms[k].m, m.a, sizeof m.a)) return cur_xf_id = k;
        Mat4 inv;
        if (!mat_inverse(m, &inv)) { *ok = false; return -1; }
        Mat4 invt = mat_transpose(inv);
        rt_transform t;
        std::memcpy(t.m, m.a, sizeof m.a);
        std::memcpy(t.inv, inv.a, sizeof inv.a);
        std::memcpy(t.inv_t, invt.a, sizeof invt.a);
        s->transforms.push_back(t);
        return cur_xf_id = (int)s->transforms.size() - 1;
    }
    void add_object(int kind, const double* g9, const double* n9, int mat, int xf) {
        s->kind.push_back(kind);
        s->material.push_back(mat);
        s->transform.push_back(xf);
        s->interp.push_back(n9 ? 1 : 0);
        for (int i = 0; i < 9; i++) s->geom.push_back(g9[i]);
        for (int i = 0; i < 9; i++) s->normals.push_back(n9 ? n9[i] : 0.0);
    }
    int load_obj(const std::string& name);
};

// ---- .obj files -------------------------------------------------------------------
// parseObj (raytracer.cpp:624-703) reads a line, splits it at white space and looks at
// the first token; the text of a mesh with a million faces is tens of megabytes, so the
// file is mapped, cut at line ends into chunks and the chunks are parsed concurrently.
// A face may only name vertices that stand before it in the file (anything else is out of
// range in the reference), so every face remembers how many vertices and normals its own
// chunk had seen; the chunks' totals give the rest once all of them are parsed, and the
// faces are then resolved straight into the scene's arrays, again chunk by chunk.

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// One white-space separated token of [p, end); false at the end of the line.
inline bool next_token(const char*& p, const char* end, const char*& tb, const char*& te) {
    while (p < end && is_blank(*p)) p++;
    if (p >= end) return false;
    tb = p;
    while (p < end && !is_blank(*p)) p++;
    te = p;
    return true;
}

// The token as a C string (atof / atol want a terminator; the mapped file has none).
struct TokenText {
    char small[64];
    std::string big;
    const char* get(const char* b, const char* e) {
        const size_t n = (size_t)(e - b);
        if (n < sizeof small) { std::memcpy(small, b, n); small[n] = 0; return small; }
        big.assign(b, e);
        return big.c_str();
    }
};

struct ObjFace {
    long vi[3], ni[3];
    long verts_before, norms_before;   // entries of this chunk's own arrays that precede the face
    bool with_normals;
};

struct ObjChunk {
    const char* begin = nullptr;
    const char* end = nullptr;
    std::vector<double> verts, norms;
    std::vector<ObjFace> faces;
    std::vector<unsigned char> messages;   // 0 unknown input, 1 vertex count, 2 normal count; file order
    long verts_base = 0, norms_base = 0;   // entries of earlier chunks
    size_t first_object = 0;
};

void parse_obj_chunk(ObjChunk& c) {
    TokenText text;
    const char* line = c.begin;
    while (line < c.end) {
        const char* eol = (const char*)std::memchr(line, '\n', (size_t)(c.end - line));
        if (!eol) eol = c.end;
        const char* p = line;
        const char *tb, *te;
        line = eol + 1;
        if (!next_token(p, eol, tb, te)) continue;                       // empty or blank line
        const size_t len = (size_t)(te - tb);
        if (len == 1 && *tb == '#') continue;
        const bool is_v = len == 1 && *tb == 'v';
        const bool is_vn = len == 2 && tb[0] == 'v' && tb[1] == 'n';
        if (is_v || is_vn) {
            float a[3] = {0.f, 0.f, 0.f};
            int n = 0;
            while (next_token(p, eol, tb, te)) {
                if (n < 3) a[n] = (float)atof(text.get(tb, te));
                n++;
            }
            if (n != 3) c.messages.push_back(is_v ? 1 : 2);
            std::vector<double>& dst = is_v ? c.verts : c.norms;
            for (int i = 0; i < 3; i++) dst.push_back((double)a[i]);
        } else if (len == 1 && *tb == 'f') {
            ObjFace f{};
            f.verts_before = (long)(c.verts.size() / 3);
            f.norms_before = (long)(c.norms.size() / 3);
            int n = 0;
            while (next_token(p, eol, tb, te)) {
                const char* sep = nullptr;                               // first "//" of the token
                for (const char* q = tb; q + 1 < te; q++)
                    if (q[0] == '/' && q[1] == '/') { sep = q; break; }
                f.with_normals = sep != nullptr;                         // the LAST token decides (raytracer.cpp:658)
                if (n < 3) {
                    if (sep) {
                        f.vi[n] = atol(text.get(tb, sep));
                        f.ni[n] = atol(text.get(sep + 2, te));
                    } else {
                        f.vi[n] = atol(text.get(tb, te));
                    }
                }
                n++;
            }
            c.faces.push_back(f);
        } else {
            c.messages.push_back(0);
        }
    }
}

struct MappedFile {
    const char* data = nullptr;
    size_t size = 0;
    bool ok = false;
    explicit MappedFile(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            size = (size_t)st.st_size;
            ok = true;
            if (size) {
                void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m == MAP_FAILED) { ok = false; size = 0; } else data = (const char*)m;
            }
        }
        ::close(fd);
    }
    ~MappedFile() { if (data) ::munmap((void*)data, size); }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
};

template <class F> void for_chunks(size_t n, int n_threads, F&& body) {
    if (n_threads <= 1 || n <= 1) { for (size_t k = 0; k < n; k++) body(k); return; }
    // An exception inside a worker (out of memory while a chunk's arrays grow) is carried back to
    // the caller; a thread that cannot be started leaves its share to the calling thread.
    std::vector<std::exception_ptr> failed((size_t)n_threads);
    auto work = [&](int t) {
        try { for (size_t k = (size_t)t; k < n; k += (size_t)n_threads) body(k); }
        catch (...) { failed[(size_t)t] = std::current_exception(); }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)n_threads);
    int started = 0;
    try { for (; started < n_threads; started++) pool.emplace_back(work, started); } catch (...) {}
    for (int t = started; t < n_threads; t++) work(t);
    for (auto& th : pool) th.join();
    for (auto& e : failed) if (e) std::rethrow_exception(e);
}

int Builder::load_obj(const std::string& name) {
    std::string path = (name.empty() || name[0] == '/' || dir.empty()) ? name : dir + "/" + name;
    MappedFile file(path);
    if (!file.ok) {
        fprintf(stderr, "Error opening obj file\n");   // the reference carries on (raytracer.cpp:632-635)
        return 0;
    }
    bool ok;
    int xf = transform_id(&ok);
    if (!ok) return fail(RT_ERR_ARG, "singular transform in front of obj " + name);
    int mat = material_id();

    // chunks of about 1 MiB (RTH_OBJ_CHUNK_BYTES overrides, for tests), each ending after a line end
    const unsigned hw = std::thread::hardware_concurrency();
    size_t chunk_bytes = (size_t)1 << 20;
    if (const char* e = getenv("RTH_OBJ_CHUNK_BYTES")) chunk_bytes = std::max<size_t>(1, (size_t)atoll(e));
    const size_t n_chunks_want = std::max<size_t>(1, file.size / chunk_bytes);
    const int n_threads = (int)std::min<size_t>(std::min<size_t>(16, hw ? hw : 4), n_chunks_want);
    std::vector<ObjChunk> chunks;
    const char* at = file.data;
    const char* const file_end = file.data + file.size;
    for (size_t k = 0; k < n_chunks_want && at < file_end; k++) {
        const char* stop = file_end;
        if (k + 1 < n_chunks_want) {
            const char* want = file.data + file.size / n_chunks_want * (k + 1);
            if (want < at) want = at;
            const char* nl = (const char*)std::memchr(want, '\n', (size_t)(file_end - want));
            stop = nl ? nl + 1 : file_end;
        }
        ObjChunk c;
        c.begin = at; c.end = stop;
        chunks.push_back(std::move(c));
        at = stop;
    }
    for_chunks(chunks.size(), n_threads, [&](size_t k) { parse_obj_chunk(chunks[k]); });

    // messages in file order, as the reference prints them while it reads
    for (const ObjChunk& c : chunks)
        for (unsigned char m : c.messages) {
            if (m == 0) fprintf(stderr, "Unknown input in obj file: %s.\n", path.c_str());
            else fprintf(stderr, "Incorrect number of arguments for %s\n", m == 1 ? "vertex in obj file." : "vertex normal in obj file.");
        }

    // index 0 of the vertex and normal arrays is a dummy: OBJ indices start at 1
    long n_verts = 1, n_norms = 1;
    size_t n_faces = 0;
    for (ObjChunk& c : chunks) {
        c.verts_base = n_verts; c.norms_base = n_norms; c.first_object = s->kind.size() + n_faces;
        n_verts += (long)(c.verts.size() / 3); n_norms += (long)(c.norms.size() / 3);
        n_faces += c.faces.size();
    }
    std::vector<double> verts((size_t)n_verts * 3, 0.0), norms((size_t)n_norms * 3, 0.0);
    const size_t total = s->kind.size() + n_faces;
    if (total > s->kind.capacity()) {                                   // room for what the scene file adds later
        const size_t cap = total + total / 16 + 1024;
        s->kind.reserve(cap); s->material.reserve(cap); s->transform.reserve(cap); s->interp.reserve(cap);
        s->geom.reserve(cap * 9); s->normals.reserve(cap * 9);
    }
    s->kind.resize(total, 0);
    s->material.resize(total, mat);
    s->transform.resize(total, xf);
    s->interp.resize(total, 0);
    s->geom.resize(total * 9);                                          // every element is written below
    s->normals.resize(total * 9);
    for_chunks(chunks.size(), n_threads, [&](size_t k) {
        const ObjChunk& c = chunks[k];
        if (!c.verts.empty()) std::memcpy(&verts[(size_t)c.verts_base * 3], c.verts.data(), c.verts.size() * sizeof(double));
        if (!c.norms.empty()) std::memcpy(&norms[(size_t)c.norms_base * 3], c.norms.data(), c.norms.size() * sizeof(double));
    });
    for_chunks(chunks.size(), n_threads, [&](size_t k) {
        const ObjChunk& c = chunks[k];
        for (size_t j = 0; j < c.faces.size(); j++) {
            const ObjFace& f = c.faces[j];
            const long nv = c.verts_base + f.verts_before, nn = c.norms_base + f.norms_before;
            double* g = &s->geom[(c.first_object + j) * 9];
            double* nrm = &s->normals[(c.first_object + j) * 9];
            for (int v = 0; v < 3; v++) {
                const bool in = f.vi[v] >= 0 && f.vi[v] < nv;           // out of range is undefined in the reference
                for (int x = 0; x < 3; x++) g[3 * v + x] = in ? verts[3 * (size_t)f.vi[v] + x] : 0.0;
                const bool inn = f.with_normals && f.ni[v] >= 0 && f.ni[v] < nn;
                for (int x = 0; x < 3; x++) nrm[3 * v + x] = inn ? norms[3 * (size_t)f.ni[v] + x] : 0.0;
            }
            s->interp[c.first_object + j] = f.with_normals ? 1 : 0;
        }
    });
    return 0;
}

int parse_scene(rth_scene* s, const char* scene_path, int width, int height, int max_depth, float eps) {
    std::ifstream f(scene_path);
    if (!f.is_open()) return fail(RT_ERR_ARG, std::string("No scene file found: ") + scene_path);
    Builder b;
    b.s = s;
    std::string sp(scene_path);
    size_t slash = sp.find_last_of('/');
    b.dir = slash == std::string::npos ? "" : sp.substr(0, slash);
    b.stack.push_back(mat_identity());
    s->d.width = width; s->d.height = height;
    s->d.max_depth = max_depth < 0 ? 6 : max_depth;
    s->d.eps = eps > 0.f ? eps : 0.05f;
    s->d.shadow_eps = 0.1f;
    bool have_cam = false;
    std::string line, cmd;
    while (std::getline(f, line)) {
        std::istringstream iss(line);
        // A whitespace-only line makes the reference repeat the previous command
        // without arguments (raytracer.cpp:723-733); it is skipped here.
        if (line.empty() || !(iss >> cmd)) continue;
        if (cmd == "cam") {
            float a[15];
            read_floats(iss, a, 15, "camera");
            for (int i = 0; i < 3; i++) {
                s->d.cam_eye[i] = a[i]; s->d.cam_ll[i] = a[3 + i]; s->d.cam_lr[i] = a[6 + i];
                s->d.cam_ul[i] = a[9 + i]; s->d.cam_ur[i] = a[12 + i];
            }
            // A new Scene starts with an ambient colour that is never set
            // (raytracer.cpp:473-474); it reads as 0 in practice.
            s->d.ambient[0] = s->d.ambient[1] = s->d.ambient[2] = 0.0;
            have_cam = true;
        } else if (cmd == "mat") {
            float a[13];
            read_floats(iss, a, 13, "material.");
            for (int i = 0; i < 3; i++) {
                b.cur_mat.ka[i] = a[i]; b.cur_mat.kd[i] = a[3 + i]; b.cur_mat.ks[i] = a[6 + i];
                b.cur_mat.kr[i] = a[10 + i];
            }
            b.cur_mat.sp = a[9];
            b.cur_mat_id = -1;
        } else if (cmd == "sph" || cmd == "tri") {
            bool sph = cmd == "sph";
            float a[9] = {0};
            read_floats(iss, a, sph ? 4 : 9, sph ? "sphere." : "triangle.");
            bool ok;
            int xf = b.transform_id(&ok);
            if (!ok) return fail(RT_ERR_ARG, "singular transform in front of " + cmd);
            double g[9];
            for (int i = 0; i < 9; i++) g[i] = a[i];
            b.add_object(sph ? 1 : 0, g, nullptr, b.material_id(), xf);
        } else if (cmd == "ltp" || cmd == "ltd") {
            bool point = cmd == "ltp";
            float a[6] = {0};
            int falloff = 0, n = 0;
            std::string tok;
            while (iss >> tok) {
                if (n < 6) a[n] = (float)atof(tok.c_str());
                else if (n == 6 && point) falloff = atoi(tok.c_str());
                n++;
            }
            if (n < 6 || n > (point ? 7 : 6))
                fprintf(stderr, "Incorrect number of arguments for %s light.\n", point ? "point" : "direct");
            rt_light l{};
            for (int i = 0; i < 3; i++) { l.pos[i] = a[i]; l.color[i] = a[3 + i]; }
            l.falloff = falloff;
            l.is_point = point ? 1 : 0;
            s->lights.push_back(l);
        } else if (cmd == "lta") {
            float a[3];
            read_floats(iss, a, 3, "ambient light.");
            if (!have_cam) return fail(RT_ERR_ARG, "lta before cam (the reference dereferences a null Scene here)");
            for (int i = 0; i < 3; i++) s->d.ambient[i] = a[i];
        } else if (cmd == "obj") {
            std::string name;
            iss >> name;
            int rc = b.load_obj(name);
            if (rc) return rc;
        } else if (cmd == "xfs" || cmd == "xft" || cmd == "xfr") {
            float a[3];
            read_floats(iss, a, 3, "transformation.");
            double x = a[0], y = a[1], z = a[2];
            if (cmd == "xfs") b.stack.push_back(mat_scale(x, y, z));
            else if (cmd == "xft") b.stack.push_back(mat_translate(x, y, z));
            else {
                float mag = (float)std::sqrt(x * x + y * y + z * z);   // magnitude() returns float
                b.stack.push_back(mat_rotate(x, y, z, (double)mag));
            }
            b.cur_xf_id = -1;
        } else if (cmd == "xfz") {
            b.stack.clear();
            b.stack.push_back(mat_identity());
            b.cur_xf_id = -1;
        }
        // anything else (comments included) is ignored, as in the reference
    }
    if (!have_cam) return fail(RT_ERR_ARG, "scene has no cam line");
    s->seal();
    return 0;
}

template <class T> void put(std::vector<unsigned char>& out, const T* p, size_t n) {
    const unsigned char* c = reinterpret_cast<const unsigned char*>(p);
    out.insert(out.end(), c, c + n * sizeof(T));
}

}  // namespace

extern "C" {

// No exception may cross the C boundary (an allocation that fails, a thread that cannot be
// started): every entry point turns one into an error code.
#define RTH_GUARD_BEGIN try {
#define RTH_GUARD_END(name)                                                                       \
    } catch (const std::bad_alloc&) { return fail(RT_ERR_ARG, name ": out of host memory"); }    \
      catch (const std::exception& e) { return fail(RT_ERR_ARG, std::string(name ": ") + e.what()); } \
      catch (...) { return fail(RT_ERR_ARG, name ": unknown failure"); }

int rth_load_scene(rth_scene** out, const char* scene_path, int width, int height, int max_depth, float eps) {
    if (!out || !scene_path || width <= 0 || height <= 0) return fail(RT_ERR_ARG, "rth_load_scene: bad argument");
    rth_scene* s = nullptr;
    try {
        s = new rth_scene();
        int rc = parse_scene(s, scene_path, width, height, max_depth, eps);
        if (rc) { delete s; return rc; }
    } catch (const std::exception& e) {
        delete s;
        return fail(RT_ERR_ARG, std::string("rth_load_scene: ") + e.what());
    } catch (...) {
        delete s;
        return fail(RT_ERR_ARG, "rth_load_scene: unknown failure");
    }
    *out = s;
    return 0;
}

int rth_save_blob(const rth_scene* s, const char* path) {
    if (!s || !path) return fail(RT_ERR_ARG, "rth_save_blob: bad argument");
    RTH_GUARD_BEGIN
    std::vector<unsigned char> b;
    const rt_scene_desc& d = s->d;
    put(b, "RTSD", 4);
    int32_t hdr[4] = {1, d.width, d.height, d.max_depth};
    put(b, hdr, 4);
    float e[2] = {d.eps, d.shadow_eps};
    put(b, e, 2);
    int32_t cnt[4] = {d.n_objects, d.n_materials, d.n_transforms, d.n_lights};
    put(b, cnt, 4);
    put(b, d.cam_eye, 3); put(b, d.cam_ll, 3); put(b, d.cam_lr, 3); put(b, d.cam_ul, 3); put(b, d.cam_ur, 3);
    put(b, d.ambient, 3);
    put(b, s->kind.data(), s->kind.size()); put(b, s->material.data(), s->material.size());
    put(b, s->transform.data(), s->transform.size()); put(b, s->interp.data(), s->interp.size());
    if (b.size() % 8) { int32_t z = 0; put(b, &z, 1); }
    put(b, s->geom.data(), s->geom.size()); put(b, s->normals.data(), s->normals.size());
    for (const rt_material& m : s->materials) put(b, &m, 1);
    for (const rt_transform& t : s->transforms) put(b, &t, 1);
    for (const rt_light& l : s->lights) put(b, &l, 1);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(RT_ERR_ARG, std::string("cannot write ") + path);
    size_t w = fwrite(b.data(), 1, b.size(), f);
    fclose(f);
    return w == b.size() ? 0 : fail(RT_ERR_ARG, "short write");
    RTH_GUARD_END("rth_save_blob")
}

int rth_load_blob(rth_scene** out, const char* path) {
    if (!out || !path) return fail(RT_ERR_ARG, "rth_load_blob: bad argument");
    rth_scene* s = nullptr;
    try {
    FILE* f = fopen(path, "rb");
    if (!f) return fail(RT_ERR_ARG, std::string("cannot read ") + path);
    std::vector<unsigned char> b;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    size_t off = 0;
    auto take = [&](void* dst, size_t bytes) {
        if (off + bytes > b.size()) return false;
        std::memcpy(dst, b.data() + off, bytes);
        off += bytes;
        return true;
    };
    char magic[4];
    int32_t hdr[4], cnt[4];
    float e[2];
    s = new rth_scene();
    rt_scene_desc& d = s->d;
    bool ok = take(magic, 4) && std::memcmp(magic, "RTSD", 4) == 0 && take(hdr, 16) && hdr[0] == 1 &&
              take(e, 8) && take(cnt, 16) && take(d.cam_eye, 24) && take(d.cam_ll, 24) && take(d.cam_lr, 24) &&
              take(d.cam_ul, 24) && take(d.cam_ur, 24) && take(d.ambient, 24);
    if (ok) ok = cnt[0] >= 0 && cnt[1] >= 0 && cnt[2] >= 0 && cnt[3] >= 0;
    if (ok) {
        // the counts come from the file: the arrays they announce must fit into what is left of it
        // before anything is sized from them
        const size_t left = b.size() - off;
        const size_t need = (size_t)cnt[0] * (16 + 144) + (size_t)cnt[1] * sizeof(rt_material) +
                            (size_t)cnt[2] * sizeof(rt_transform) + (size_t)cnt[3] * sizeof(rt_light);
        ok = need <= left;
    }
    if (ok) {
        d.width = hdr[1]; d.height = hdr[2]; d.max_depth = hdr[3];
        d.eps = e[0]; d.shadow_eps = e[1];
        size_t no = (size_t)cnt[0];
        s->kind.resize(no); s->material.resize(no); s->transform.resize(no); s->interp.resize(no);
        s->geom.resize(no * 9); s->normals.resize(no * 9);
        s->materials.resize(cnt[1]); s->transforms.resize(cnt[2]); s->lights.resize(cnt[3]);
        ok = take(s->kind.data(), no * 4) && take(s->material.data(), no * 4) &&
             take(s->transform.data(), no * 4) && take(s->interp.data(), no * 4);
        if (ok && off % 8) off += 4;
        ok = ok && take(s->geom.data(), no * 72) && take(s->normals.data(), no * 72) &&
             take(s->materials.data(), s->materials.size() * sizeof(rt_material)) &&
             take(s->transforms.data(), s->transforms.size() * sizeof(rt_transform)) &&
             take(s->lights.data(), s->lights.size() * sizeof(rt_light)) && off == b.size();
    }
    if (!ok) { delete s; return fail(RT_ERR_ARG, std::string("malformed RTSD blob ") + path); }
    s->seal();
    } catch (const std::exception& e) {
        delete s;
        return fail(RT_ERR_ARG, std::string("rth_load_blob: ") + e.what());
    } catch (...) {
        delete s;
        return fail(RT_ERR_ARG, "rth_load_blob: unknown failure");
    }
    *out = s;
    return 0;
}

int rth_tile_span(int width, int height, int tile_rows, int n_ranks, int k, size_t* offset, size_t* bytes) {
    if (width <= 0 || height <= 0 || tile_rows <= 0 || n_ranks <= 0 || k < 0 || !offset || !bytes)
        return fail(RT_ERR_ARG, "rth_tile_span: bad argument");
    const long long y0 = (long long)k * tile_rows;
    if (y0 >= height) return fail(RT_ERR_ARG, "rth_tile_span: no such tile");
    const long long y1 = std::min<long long>(y0 + tile_rows, height);
    // film row y is frame row height - 1 - y: the tile's top frame row is that of its last film row
    *offset = (size_t)(height - y1) * (size_t)width * 4;
    *bytes = (size_t)(y1 - y0) * (size_t)width * 4;
    return k % n_ranks;
}

int rth_packed_share(int width, int height, int tile_rows, int n_ranks, int rank, int* n_full, size_t* first_offset,
                     size_t* tile_bytes, size_t* stride_bytes, size_t* clip_offset, size_t* clip_bytes) {
    if (width <= 0 || height <= 0 || tile_rows <= 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks || !n_full || !first_offset ||
        !tile_bytes || !stride_bytes || !clip_offset || !clip_bytes)
        return fail(RT_ERR_ARG, "rth_packed_share: bad argument");
    const int n_tiles = (height + tile_rows - 1) / tile_rows;
    const size_t tb = (size_t)tile_rows * (size_t)width * 4;
    *n_full = 0; *first_offset = 0; *tile_bytes = tb; *stride_bytes = (size_t)n_ranks * tb; *clip_offset = 0; *clip_bytes = 0;
    for (int k = rank; k < n_tiles; k += n_ranks) {
        size_t off = 0, bytes = 0;
        if (rth_tile_span(width, height, tile_rows, n_ranks, k, &off, &bytes) != rank) return fail(RT_ERR_ARG, "rth_packed_share: span");
        if (bytes == tb) { (*n_full)++; *first_offset = off; }      // (offsets fall as k rises: the last full tile lies lowest)
        else { *clip_offset = off; *clip_bytes = bytes; }
    }
    return 0;
}

const rt_scene_desc* rth_desc(const rth_scene* s) { return s ? &s->d : nullptr; }
void rth_free_scene(rth_scene* s) { delete s; }
const char* rth_last_error(void) { return g_err.c_str(); }

}  // extern "C"
