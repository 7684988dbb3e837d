// Host-side helpers of the C ABI's translation unit: error reporting, small vector algebra for the
// bounds, host threads over ranges, an owning device buffer.
#pragma once

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct Vec { double x, y, z; };
Vec vsub(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
double vdot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
Vec vcross(Vec a, Vec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
double vlen(Vec a) { return std::sqrt(vdot(a, a)); }

// M * (p, 1) without the perspective divide: only used for bounds.
Vec affine_point(const double* m, Vec p) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
Vec linear_dir(const double* m, Vec v) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z};
}

struct Box {
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    void add(Vec p) {
        double c[3] = {p.x, p.y, p.z};
        for (int i = 0; i < 3; i++) {
            if (c[i] < lo[i]) lo[i] = c[i];
            if (c[i] > hi[i]) hi[i] = c[i];
        }
    }
    Vec corner(int k) const { return {(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]}; }
};

double max_abs(Vec p) { return std::max(std::fabs(p.x), std::max(std::fabs(p.y), std::fabs(p.z))); }

// body(part, begin, end) over [0, n) cut into equal parts, one host thread per part; a single
// inline call when n is small.  Returns the number of parts (indices 0 .. parts-1 were used).
inline size_t host_parts(size_t n) {
    if (n < 65536) return 1;
    const unsigned hw = std::thread::hardware_concurrency();
    return std::min<size_t>(16, std::max(1u, hw));
}
template <class F> void parallel_ranges(size_t n, size_t parts, F&& body) {
    if (parts <= 1) { body((size_t)0, (size_t)0, n); return; }
    std::vector<std::thread> pool;
    const size_t per = (n + parts - 1) / parts;
    for (size_t k = 0; k < parts; k++)
        pool.emplace_back([&body, k, per, n]() { body(k, std::min(n, k * per), std::min(n, (k + 1) * per)); });
    for (auto& th : pool) th.join();
}

template <class T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }          // (a buffer that goes out of scope on an error path is not leaked)
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    hipError_t upload(const T* src, size_t count) {
        hipError_t e = alloc(src ? count : 0);
        if (e != hipSuccess || !src || count == 0) return e;
        return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    }
    hipError_t upload(const std::vector<T>& v) { return upload(v.data(), v.size()); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
};

}  // namespace
