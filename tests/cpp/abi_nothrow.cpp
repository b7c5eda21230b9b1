// abi_nothrow.cpp -- the C ABI of libuvo_hip.so under a failing `operator new` ("never throws across the ABI"; the
// callers of VO_utility.h:96-117 expect a status, not std::terminate).  This program REPLACES the global allocation functions, so the
// library's own std::vector / std::string allocations come here; an armed allocation of a chosen size throws std::bad_alloc.  Only
// sizes the test itself causes are armed (a staging vector of n * sizeof(element) bytes for an n it picks), never the HIP runtime's
// internal allocations, whose exception safety is not ours to test.
//   abi_nothrow host                 host-only entries (no GPU needed)
//   abi_nothrow gpu <img0> <img1> <w> <h>   + entries that stage through std::vector, and a lane WORKER's allocation (mono pose stage)
// exit 0 = every armed call came back with a status (and the next, unarmed, call worked); anything else prints what failed.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "uvo_hip.h"

#include <pthread.h>
static std::atomic<size_t> g_fail_size{0};       // allocations of exactly this many bytes throw while armed (0 = off)
static std::atomic<int> g_fail_other_threads{0}; // 1: only on threads other than main (a lane worker's allocation is the target)
static std::atomic<long> g_thrown{0};
static pthread_t g_main_thread;

static void* alloc_or_throw(size_t n)
{
    const size_t f = g_fail_size.load(std::memory_order_relaxed);
    if (f != 0 && n == f && (!g_fail_other_threads.load(std::memory_order_relaxed) || !pthread_equal(pthread_self(), g_main_thread))) { g_thrown.fetch_add(1); throw std::bad_alloc(); }
    void* p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new(size_t n) { return alloc_or_throw(n); }
void* operator new[](size_t n) { return alloc_or_throw(n); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { try { return alloc_or_throw(n); } catch (...) { return nullptr; } }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { try { return alloc_or_throw(n); } catch (...) { return nullptr; } }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, size_t) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

static int host_only()
{
    // uvo_select_estimation_method: a std::vector<double>(n) for the median (VO_utility.cpp:725-748 -> math_utility.cpp:65-86)
    const int n = 1237;
    std::vector<uvo_point2f> a(n), b(n);
    for (int i = 0; i < n; i++) { a[i] = uvo_point2f{(float)i, 0.f}; b[i] = uvo_point2f{(float)i + 3.f, 4.f}; }     // every distance 5
    CHECK(uvo_select_estimation_method(a.data(), b.data(), n, 10) == 0, "median 5 < 10 is the homography branch");
    CHECK(uvo_select_estimation_method(a.data(), b.data(), n, 3) == 1, "median 5 >= 3 is the essential branch");
    const long before = g_thrown.load();
    g_fail_size = sizeof(double) * n;
    const int r = uvo_select_estimation_method(a.data(), b.data(), n, 10);
    g_fail_size = 0;
    CHECK(g_thrown.load() == before + 1, "the armed allocation was not reached (%ld)", g_thrown.load() - before);
    CHECK(r == -1, "bad_alloc inside the entry must come back as -1, got %d", r);
    CHECK(uvo_select_estimation_method(a.data(), b.data(), n, 10) == 0, "the entry works again afterwards");
    CHECK(uvo_select_estimation_method(nullptr, b.data(), n, 10) == -1, "null input");
    // entries without allocations stay what they were
    double rv[3] = {0.1, -0.2, 0.3}, R[9], back[3];
    CHECK(uvo_rodrigues(rv, 3, R) == UVO_OK && uvo_rodrigues(R, 9, back) == UVO_OK, "rodrigues");
    CHECK(fabs(back[0] - rv[0]) < 1e-12 && fabs(back[1] - rv[1]) < 1e-12 && fabs(back[2] - rv[2]) < 1e-12, "rodrigues round trip");
    CHECK(uvo_rodrigues(rv, 4, R) == UVO_INVALID_ARG, "rodrigues size");
    // without a GPU a context cannot be made, and says so with a status
    uvo_params p; uvo_params_default_stereo(&p);
    uvo_ctx* c = nullptr;
    const uvo_status st = uvo_ctx_create(&p, 0, 640, 360, 4096, &c);
    if (st == UVO_OK) uvo_ctx_destroy(c);
    else CHECK(c == nullptr && (st == UVO_NO_DEVICE || st == UVO_HIP_ERROR), "context creation without a device: status %d", (int)st);
    return 0;
}

static bool read_file(const char* path, std::vector<uint8_t>* out, size_t n)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    out->resize(n);
    const bool ok = fread(out->data(), 1, n, f) == n;
    fclose(f);
    return ok;
}

static int with_gpu(const char* p0, const char* p1, int w, int h)
{
    std::vector<uint8_t> img0, img1;
    CHECK(read_file(p0, &img0, (size_t)w * h) && read_file(p1, &img1, (size_t)w * h), "cannot read the images");
    uvo_params p; uvo_params_default_mono(&p);
    p.SURF_MIN_HESSIAN = 400; p.ESSENTIAL_OUTLIER_METHOD = 8; p.HOMOGRAPHY_OUTLIER_METHOD = 8; p.ESSENTIAL_THRESHOLD = 1.0; p.HOMOGRAPHY_THRESHOLD = 1.0;
    uvo_ctx* c = nullptr;
    CHECK(uvo_ctx_create(&p, 0, w, h, 4096, &c) == UVO_OK && c, "uvo_ctx_create");
    // ---- an operator that stages through a std::vector on the calling thread: uvo_triangulate_points (std::vector<float4>(n)) ----
    const int n = 1237;
    std::vector<uvo_point2f> x1(n), x2(n);
    for (int i = 0; i < n; i++) { x1[i] = uvo_point2f{100.f + (i % 40) * 9.f, 50.f + (i / 40) * 7.f}; x2[i] = uvo_point2f{x1[i].x - 12.f, x1[i].y}; }
    const double P1[12] = {500, 0, 320, 0, 0, 500, 180, 0, 0, 0, 1, 0}, P2[12] = {500, 0, 320, -165, 0, 500, 180, 0, 0, 0, 1, 0};
    std::vector<float> out4(4 * (size_t)n), ref4(4 * (size_t)n);
    CHECK(uvo_triangulate_points(c, P1, P2, x1.data(), x2.data(), n, ref4.data()) == UVO_OK, "%s", uvo_last_error(c));
    long before = g_thrown.load();
    g_fail_size = 16 * (size_t)n;                                   // the float4 staging vector
    uvo_status st = uvo_triangulate_points(c, P1, P2, x1.data(), x2.data(), n, out4.data());
    g_fail_size = 0;
    CHECK(g_thrown.load() == before + 1, "the armed allocation was not reached");
    CHECK(st == UVO_CAPACITY, "bad_alloc must come back as UVO_CAPACITY, got %d", (int)st);
    CHECK(strstr(uvo_last_error(c), "out of host memory") != nullptr, "uvo_last_error: '%s'", uvo_last_error(c));
    CHECK(uvo_triangulate_points(c, P1, P2, x1.data(), x2.data(), n, out4.data()) == UVO_OK, "%s", uvo_last_error(c));
    CHECK(memcmp(out4.data(), ref4.data(), sizeof(float) * out4.size()) == 0, "the operator's result changed after the failed call");
    // ---- a lane WORKER's allocation: the pose stage of a pipelined mono frame sizes its mask (std::vector<uint8_t>, M bytes) on the lane's
    // worker thread (ctx.hip: run_mono_stage_b); armed for threads other than this one only ----
    const double K[9] = {500, 0, w * 0.5, 0, 500, h * 0.5, 0, 0, 1};
    CHECK(uvo_mono_set_camera(c, K) == UVO_OK, "%s", uvo_last_error(c));
    uvo_mono_result r0, r1;
    CHECK(uvo_mono_step(c, img0.data(), w, h, w, UVO_MEM_HOST, 4.0, 0.2, &r0) == UVO_OK, "%s", uvo_last_error(c));
    CHECK(uvo_mono_step(c, img1.data(), w, h, w, UVO_MEM_HOST, 4.0, 0.2, &r1) == UVO_OK, "%s", uvo_last_error(c));
    CHECK(r1.n_matches >= 20, "the test images give %d matches", r1.n_matches);
    CHECK(uvo_mono_reset(c) == UVO_OK, "%s", uvo_last_error(c));
    CHECK(uvo_mono_submit(c, img0.data(), w, h, w, UVO_MEM_HOST, 4.0) == UVO_OK, "%s", uvo_last_error(c));      // init frame, synchronous
    uvo_mono_result q;
    CHECK(uvo_mono_collect(c, 0.2, &q) == UVO_OK, "%s", uvo_last_error(c));
    before = g_thrown.load();
    g_fail_other_threads = 1;
    g_fail_size = (size_t)r1.n_matches;                             // the worker's L->mono_mask.assign(M, 0)
    st = uvo_mono_submit(c, img1.data(), w, h, w, UVO_MEM_HOST, 4.0);
    uvo_status st2 = st == UVO_OK ? uvo_mono_collect(c, 0.2, &q) : st;
    g_fail_size = 0; g_fail_other_threads = 0;
    CHECK(st == UVO_OK, "submit: %s", uvo_last_error(c));
    CHECK(g_thrown.load() >= before + 1, "the worker's armed allocation was not reached");
    CHECK(st2 == UVO_CAPACITY, "a worker's bad_alloc must fail the frame with UVO_CAPACITY, got %d (%s)", (int)st2, uvo_last_error(c));
    // the context survives: the same two frames again give the first run's result
    CHECK(uvo_mono_reset(c) == UVO_OK, "%s", uvo_last_error(c));
    CHECK(uvo_mono_step(c, img0.data(), w, h, w, UVO_MEM_HOST, 4.0, 0.2, &q) == UVO_OK, "%s", uvo_last_error(c));
    CHECK(uvo_mono_step(c, img1.data(), w, h, w, UVO_MEM_HOST, 4.0, 0.2, &q) == UVO_OK, "%s", uvo_last_error(c));
    CHECK(memcmp(&q, &r1, sizeof(q)) == 0, "the mono loop's result changed after the failed frame");
    uvo_ctx_destroy(c);
    return 0;
}

int main(int argc, char** argv)
{
    g_main_thread = pthread_self();
    if (argc >= 2 && !strcmp(argv[1], "host")) return host_only();
    if (argc == 6 && !strcmp(argv[1], "gpu")) { const int rc = host_only(); return rc ? rc : with_gpu(argv[2], argv[3], atoi(argv[4]), atoi(argv[5])); }
    fprintf(stderr, "usage: abi_nothrow host | gpu <img0.raw> <img1.raw> <w> <h>\n");
    return 2;
}
