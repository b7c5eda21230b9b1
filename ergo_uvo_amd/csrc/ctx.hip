// ctx.hip -- the C ABI of libuvo_hip.so (include/uvo_hip.h): context, standalone operators and the
// fused stereo step (visual_odometry_node::stereo_VO, visual_odometry.h:474-739) with every
// intermediate kept in HBM.
#include "uvo_ctx.h"
#include <pthread.h>
#include "uvo_epnp.h"
#include <string.h>
#include <sched.h>
#include <atomic>
#include <math.h>
#include <float.h>
#include <new>
#include <algorithm>
#include <atomic>
namespace uvo { extern bool g_bdbg; extern std::atomic<double> g_bstat[16]; double now_us(); void operator+=(std::atomic<double>& a, double v); }

using namespace uvo;

struct uvo_ctx : public uvo::Ctx {};

// Nothing leaves the C ABI as a C++ exception ("never throws across the ABI"; the reference's callers, VO_utility.h:96-117,
// expect a status, and an exception through a C frame is undefined behaviour).  The host side allocates (std::vector staging buffers,
// std::string messages, std::deque / std::thread at context creation): every extern "C" body is a function-try-block whose handler
// lands here, and the lane workers catch around their job (lane_worker) -- std::bad_alloc becomes UVO_CAPACITY, anything else
// UVO_HIP_ERROR, with uvo_last_error set when a context is at hand.
namespace {
uvo_status abi_caught(uvo_ctx* c) noexcept
{
    uvo_status st = UVO_HIP_ERROR;
    const char* what = "unexpected C++ exception inside libuvo_hip";
    static thread_local char text[160];
    try { throw; }
    catch (const std::bad_alloc&) { st = UVO_CAPACITY; what = "out of host memory (std::bad_alloc) inside libuvo_hip"; }
    catch (const std::exception& e) { snprintf(text, sizeof(text), "C++ exception inside libuvo_hip: %s", e.what()); what = text; }
    catch (...) {}
    if (c) { try { c->err.assign(what); } catch (...) { c->err.clear(); } }     // (create_one reserves the message's room)
    return st;
}
}  // namespace
#define UVO_ABI_CATCH(c) catch (...) { return abi_caught(c); }
#define UVO_ABI_CATCH_RET(c, v) catch (...) { (void)abi_caught(c); return v; }
#define UVO_ABI_CATCH_VOID(c) catch (...) { (void)abi_caught(c); }

// Every pipeline lane owns two HIP streams; with ROCm's default of four hardware queues the lanes share queues and one lane's
// detection waits behind another lane's PnP kernels (2010 pairs/s instead of 2830 at depth 6).  The
// variable is read when the HIP runtime initialises, so it is set when this library is loaded -- before main() when the node
// links it -- unless the process already chose a value.
__attribute__((constructor)) static void uvo_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }


extern "C" void uvo_params_default_stereo(uvo_params* p)
{   // uvo/config/stereo_VO_parameters.yaml:20-47 (keys absent from that file stay zero, as the globals do)
    memset(p, 0, sizeof(*p));
    p->LOWE_RATIO_THRESHOLD = 0.8; p->REPROJECTION_TOLERANCE = 3.0;
    p->MIN_NUM_FEATURES = 5; p->MIN_NUM_3DPOINTS = 5; p->MIN_NUM_INLIERS = 5;
    p->ITERATIONS_COUNT = 1000; p->REPROJECTION_ERROR_THRESHOLD = 1.0; p->CONFIDENCE = 0.99;
    p->USE_EXTRINSIC_GUESS = 0; p->PNP_METHOD_FLAG = 1;
    p->SURF_MIN_HESSIAN = 1500; p->SURF_OCTAVES_NUMBER = 4; p->SURF_OCTAVES_LAYERS = 3; p->SURF_EXTENDED = 0; p->SURF_UPRIGHT = 1;
}
extern "C" void uvo_params_default_mono(uvo_params* p)
{   // uvo/config/mono_VO_parameters.yaml:13-49
    memset(p, 0, sizeof(*p));
    p->DISTANCE = 10; p->LOWE_RATIO_THRESHOLD = 0.7;
    p->ESSENTIAL_OUTLIER_METHOD = 4; p->ESSENTIAL_MAX_ITERS = 2000; p->ESSENTIAL_CONFIDENCE = 0.99; p->ESSENTIAL_THRESHOLD = 0.1;
    p->HOMOGRAPHY_OUTLIER_METHOD = 4; p->HOMOGRAPHY_MAX_ITERS = 2000; p->HOMOGRAPHY_CONFIDENCE = 0.99; p->HOMOGRAPHY_THRESHOLD = 0.1;
    p->HOMOGRAPHY_DISTANCE = 50.0; p->VPF_THRESHOLD = 0.4; p->REPROJECTION_TOLERANCE = 0.1;
    p->MIN_NUM_FEATURES = 20; p->MIN_NUM_INLIERS = 10; p->MIN_NUM_3DPOINTS = 5;
    p->SURF_MIN_HESSIAN = 50; p->SURF_OCTAVES_NUMBER = 4; p->SURF_OCTAVES_LAYERS = 3; p->SURF_EXTENDED = 0; p->SURF_UPRIGHT = 1;
}

template <class T>
static hipError_t dalloc(T** p, size_t count) { return hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (count ? count : 1)); }

// getGaussianKernel(20, 3.3, CV_32F) outer product (surf.cpp SURFInvoker ctor: DW)
static void make_desc_weights(float* DW)
{
    double t[20], sum = 0; float G[20];
    const double sigma = 3.3f;
    double scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < 20; i++) { double x = i - 19 * 0.5; t[i] = exp(scale2X * x * x); sum += t[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 20; i++) G[i] = (float)(t[i] * sum);
    for (int i = 0; i < 20; i++) for (int j = 0; j < 20; j++) DW[i * 20 + j] = G[i] * G[j];
}

static void lane_worker(uvo_ctx* L);
__global__ void k_prime(int* sink, int n);
static void run_stage_b(uvo_ctx* L, bool stage_a_ok);
static uvo_status prime_lanes(uvo_ctx* c, int w, int h);
extern "C" uvo_status uvo_mono_collect(uvo_ctx* c, double dt, uvo_mono_result* out);
static void run_mono_stage_b(uvo_ctx* L, bool stage_a_ok);
static void destroy_one(uvo_ctx* c);
static uvo_status queue_stage_a(uvo_ctx* c, uvo_ctx* A, uvo_ctx* B);
struct GatherPair;
// the ring of timing events behind UVO_TRACE / uvo_trace_enable (a lane's, created once)
static hipError_t trace_alloc(Ctx* c)
{
    if (c->trace.empty()) {
        c->trace.resize(Ctx::kTraceRing);
        for (auto& r : c->trace) for (int k = 0; k < 8; k++) { hipError_t e = hipEventCreate(&r.ev[k]); if (e != hipSuccess) return e; }
    }
    c->trace_on = true;
    return hipSuccess;
}

// HIP streams are parked, not destroyed, and a parked stream goes back to the ROLE it had (stage-A or PnP stream of lane i).
// Measured in round 3: a context created after another one of the same process had been destroyed ran ~10 % below its rate.  Round 4
// (tools/probe/ctx_reuse.py, five contexts in a row, pairs/s): create / destroy 4440 4047 4052 4094 4160; streams parked in one
// free list 4435 3880 4417 3859 4406 -- the same twelve streams, handed out in the reverse order by every second context, are 13 %
// slower: the runtime binds a stream to a hardware queue (and the driver that queue to a pipe of the command processor) when the
// stream is first used and never rebalances, so WHICH stream serves which role decides which roles share a pipe.  Parked per role,
// every context of a process gets the first one's binding.  UVO_STREAM_POOL=0 restores create / destroy (measurement).
namespace {
std::mutex g_pool_mu;
hipStream_t g_pool[64][2][uvo::Ctx::kMaxDepth];            // [device][role: 0 stage A, 1 PnP][lane]
bool pool_on() { static const bool on = !(getenv("UVO_STREAM_POOL") && atoi(getenv("UVO_STREAM_POOL")) == 0); return on; }
hipError_t stream_acquire(int device, int role, int lane, hipStream_t* out)
{
    if (pool_on() && lane >= 0 && lane < uvo::Ctx::kMaxDepth) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        hipStream_t& slot = g_pool[device & 63][role][lane];
        if (slot) { *out = slot; slot = nullptr; return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void stream_release(int device, int role, int lane, hipStream_t s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s);
    if (pool_on() && lane >= 0 && lane < uvo::Ctx::kMaxDepth) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        hipStream_t& slot = g_pool[device & 63][role][lane];
        if (!slot) { slot = s; return; }                    // (two live contexts of one process: the second one's streams are its own)
    }
    (void)hipStreamDestroy(s);
}
}  // namespace

// one set of buffers, streams and a stage-B worker thread: the caller's context, or a further pipeline lane of it
// parameters the implementation cannot honour are refused, never silently replaced (the reference passes them to OpenCV)
static const char* unsupported_params(const uvo_params* p)
{
    if (p->ITERATIONS_COUNT > 0 && p->PNP_METHOD_FLAG != 1)
        return "PNP_METHOD_FLAG: only 1 (cv::SOLVEPNP_EPNP, the value shipped in stereo_VO_parameters.yaml) is implemented";
    if (p->USE_EXTRINSIC_GUESS != 0) return "USE_EXTRINSIC_GUESS: only false is implemented (EPnP ignores the guess)";
    return nullptr;
}

static uvo_status create_one(const uvo_params* p, int device, int max_w, int max_h, int max_kpts, uvo_ctx** out, int lane = 0)
{
    if (!out) return UVO_INVALID_ARG;
    *out = nullptr;
    if (!p || max_w < 16 || max_h < 16 || max_kpts < 16 || unsupported_params(p)) return UVO_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return UVO_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return UVO_HIP_ERROR;
    uvo_ctx* c = new (std::nothrow) uvo_ctx();
    if (!c) return UVO_HIP_ERROR;
    c->p = *p; c->device = device; c->max_w = max_w; c->max_h = max_h; c->cap = max_kpts; c->lane_id = lane;
    const size_t cap = (size_t)max_kpts;
    const size_t npx = (size_t)max_w * max_h, nsum = (size_t)(max_w + 1) * (max_h + 1);
    const int nstrip = (max_h + 7) / 8;                          // integral image: strips of 8 rows (surf.hip)
    c->colpart_stride = ((max_w + 1 + 2047) / 2048) * 2048;
    const size_t nchunks = (cap + 127) / 128;                   // matcher shortlist: (cap/128) x cap float4 (match.hip)
    hipError_t e = hipSuccess;
#define A(expr) do { if (e == hipSuccess) e = (expr); } while (0)
    A(stream_acquire(device, 0, lane, &c->stream));
    A(hipEventCreate(&c->ev0)); A(hipEventCreate(&c->ev1));
    c->plane_pw = ((max_w + 1 + 3) / 4 + 1 + 1) & ~1;            // even: the integral kernel stores pairs of plane entries
    c->plane_stride = c->plane_pw * ((max_h + 1 + 3) / 4 + 1);
    for (int i = 0; i < 2; i++) {
        A(dalloc(&c->d_planes[i], (size_t)16 * c->plane_stride));
        if (e == hipSuccess) e = hipMemset(c->d_planes[i], 0, sizeof(int32_t) * 16 * c->plane_stride);       // row 0 / padding stay zero
        A(dalloc(&c->d_img[i], npx)); A(dalloc(&c->d_sum_base[i], nsum + 2 * kSumPad));
        if (e == hipSuccess) { e = hipMemset(c->d_sum_base[i], 0, sizeof(int32_t) * (nsum + 2 * kSumPad)); c->d_sum[i] = c->d_sum_base[i] + kSumPad; }
        A(dalloc(&c->d_cand[i], cap)); A(dalloc(&c->det[i].kps, cap)); A(dalloc(&c->det[i].desc, cap * 128));      // 128: SURF_EXTENDED rows
        A(dalloc(&c->d_tmp_desc[i], cap * 128)); A(dalloc(&c->d_matches[i], cap));
        A(dalloc(&c->d_as_kpsL[i], cap)); A(dalloc(&c->d_as_kpsR[i], cap)); A(dalloc(&c->d_as_descL[i], cap * 128));
        A(dalloc(&c->d_as_pts4[i], cap)); A(dalloc(&c->d_as_cam1[i], cap * 3)); A(dalloc(&c->d_as_flag[i], cap));
    }
    c->surv_cap = 4 * (int)cap; A(dalloc(&c->d_surv, (size_t)c->surv_cap));
    A(dalloc(&c->d_colpart, (size_t)2 * nstrip * c->colpart_stride));
    A(dalloc(&c->d_DW, 400)); A(dalloc(&c->d_rank, cap * 2)); A(dalloc(&c->d_big_par, cap * 4)); A(dalloc(&c->d_big_patch, cap * 2 * 448));
    A(dalloc(&c->d_mpart, 2 * nchunks * cap)); A(dalloc(&c->d_mscratch, 2 * (nchunks + 4))); A(dalloc(&c->d_knn_idx, 2 * cap * 2)); A(dalloc(&c->d_knn_dist, 2 * cap * 2));
    A(dalloc(&c->d_x1, cap)); A(dalloc(&c->d_x2, cap)); A(dalloc(&c->d_xc, cap)); A(dalloc(&c->d_pts4, cap));
    A(dalloc(&c->d_cam1, cap * 3)); A(dalloc(&c->d_flag, cap)); A(dalloc(&c->d_tmp_idx, cap)); A(dalloc(&c->d_tmp_row, cap));
    if (const char* ww = getenv("UVO_WORKER_WAIT")) c->worker_wait = !strcmp(ww, "spin") ? 0 : (!strcmp(ww, "block-all") ? 2 : ((!strcmp(ww, "sleep") || !strcmp(ww, "block")) ? 3 : 1));
    c->wait_eff.store(c->worker_wait == 0 ? 0 : (c->worker_wait == 2 ? 2 : 1), std::memory_order_relaxed);      // auto: decided by set_depth (apply_wait_policy)
    A(hipEventCreateWithFlags(&c->evBlock, hipEventDisableTiming | hipEventBlockingSync));
    A(hipEventCreateWithFlags(&c->evPoll, hipEventDisableTiming));
    A(hipEventCreateWithFlags(&c->evB, hipEventDisableTiming | (c->worker_wait == 2 ? hipEventBlockingSync : 0)));
    for (int i = 0; i < 2; i++) {
        A(dalloc(&c->d_good_pts[i], cap * 3)); A(dalloc(&c->d_good_idx[i], cap));
        A(dalloc(&c->d_opts[i], cap * 3)); A(dalloc(&c->d_ipts[i], cap));
        // evA[0] is the event the lane's worker waits on (sleeping unless UVO_WORKER_WAIT=spin), evA[1] the one the submitting thread polls
        A(hipEventCreateWithFlags(&c->evA[i], hipEventDisableTiming | ((i == 0 && c->worker_wait == 2) ? hipEventBlockingSync : 0)));
        A(hipHostMalloc(reinterpret_cast<void**>(&c->h_countsA[i]), sizeof(int) * CN_TOTAL));
    }
    A(hipEventCreateWithFlags(&c->evAS, hipEventDisableTiming));
    A(hipEventCreateWithFlags(&c->evSync, hipEventDisableTiming));
    A(hipEventCreateWithFlags(&c->evProducer, hipEventDisableTiming));
    A(hipEventCreateWithFlags(&c->evDet, hipEventDisableTiming)); A(hipEventCreateWithFlags(&c->evPrevRead, hipEventDisableTiming));
    // (Round 3 tried the highest stream priority for the PnP stream, so that its 1..8-workgroup kernels would be dispatched ahead of
    // queued detection tiles: no effect on a fresh context -- 3792 / 4309 pairs/s with, 3774 / 4308 without, 20- / 600-step forms --
    // and contexts created later in the same process ran at 60 % of their rate, as if the priority queues were never handed back.
    // UVO_PNP_PRIORITY=1 brings it back for measurements.)
    {
        int least = 0, greatest = 0;
        const char* pe = getenv("UVO_PNP_PRIORITY");
        if (pe && atoi(pe) != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            A(hipStreamCreateWithPriority(&c->pnp_stream, hipStreamNonBlocking, greatest));
        else
            A(stream_acquire(device, 1, lane, &c->pnp_stream));
    }
    A(dalloc(&c->d_countsB, (size_t)4)); A(hipHostMalloc(reinterpret_cast<void**>(&c->h_countsB), sizeof(int) * 4));
    A(dalloc(&c->d_counts, (size_t)CN_TOTAL));
    A(dalloc(&c->d_subsets, (size_t)kMaxHyp * 5)); A(dalloc(&c->d_models, (size_t)kMaxHyp * 6)); A(dalloc(&c->d_hcount, (size_t)kMaxHyp));
    A(dalloc(&c->d_inliers, cap)); A(dalloc(&c->d_refit, cap * 21)); A(dalloc(&c->d_pose, (size_t)6));
    A(hipHostMalloc(reinterpret_cast<void**>(&c->h_counts), sizeof(int) * CN_TOTAL));
    A(hipHostMalloc(reinterpret_cast<void**>(&c->h_subsets), sizeof(int) * kMaxHyp * 5));
    A(hipHostMalloc(reinterpret_cast<void**>(&c->h_hcount), sizeof(int) * kMaxHyp));
    A(hipHostMalloc(reinterpret_cast<void**>(&c->h_pose), sizeof(double) * 6));
    A(dalloc(&c->d_rng_raw, (size_t)1024)); A(hipMalloc(&c->d_spec, 64)); A(hipHostMalloc(&c->h_spec, 64));
#undef A
    if (e != hipSuccess) { destroy_one(c); return UVO_HIP_ERROR; }
    c->d_cand_n = c->d_counts + CN_CAND0;
    c->det[0].n = c->d_counts + CN_NL; c->det[1].n = c->d_counts + CN_NR;
    c->d_nmatch = c->d_counts + CN_M;
    c->d_as_n = c->d_counts + CN_AS0;
    c->d_big_n = c->d_counts + CN_BIG0;
    if (surf_build_area_tables(c) != UVO_OK) { destroy_one(c); return UVO_HIP_ERROR; }
    {   // getSubset's generator is cv::RNG((uint64)-1) for every call: its raw outputs are a constant table
        std::vector<unsigned> raw(1024);
        uint64_t state = (uint64_t)-1;
        for (auto& v : raw) v = rng_next(state);
        if (hipMemcpy(c->d_rng_raw, raw.data(), sizeof(unsigned) * raw.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(c->d_spec, 0, 64) != hipSuccess) { destroy_one(c); return UVO_HIP_ERROR; }
        memset(c->h_spec, 0, 64);
    }
    make_desc_weights(c->h_DW);
    if (hipMemcpy(c->d_DW, c->h_DW, sizeof(float) * 400, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(c->d_rank, 0, sizeof(int) * cap * 2) != hipSuccess ||
        hipMemset(c->d_counts, 0, sizeof(int) * CN_TOTAL) != hipSuccess) { destroy_one(c); return UVO_HIP_ERROR; }
    if (getenv("UVO_TRACE") && trace_alloc(c) != hipSuccess) { destroy_one(c); return UVO_HIP_ERROR; }
    try { c->err.reserve(256); c->worker = std::thread(lane_worker, c); }
    catch (...) { destroy_one(c); return UVO_HIP_ERROR; }
    *out = c;
    return UVO_OK;
}

// UVO_TRACE: one CSV row per traced pair, times in ms relative to the earliest "A begin"
static void write_trace(uvo_ctx* c)
{
    const char* path = getenv("UVO_TRACE");
    if (!path || !c->trace_on) return;
    (void)hipDeviceSynchronize();
    struct Row { long long pair; int lane; float t[6]; };
    std::vector<Row> rows;
    hipEvent_t ref = nullptr; long long ref_pair = -1;
    for (Ctx* l : c->lanes) for (auto& r : l->trace) if (r.pair >= 0 && (ref_pair < 0 || r.pair < ref_pair)) { ref = r.ev[0]; ref_pair = r.pair; }
    if (!ref) return;
    for (Ctx* l : c->lanes) for (auto& r : l->trace) {
        if (r.pair < 0) continue;
        Row w; w.pair = r.pair; w.lane = l->lane_id;
        for (int k = 0; k < 6; k++) { w.t[k] = -1.f; if (k < 3 || r.b_used) (void)hipEventElapsedTime(&w.t[k], ref, r.ev[k]); }
        rows.push_back(w);
    }
    std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.pair < b.pair; });
    FILE* f = fopen(path, "w");
    if (!f) return;
    fprintf(f, "pair,lane,a_begin_ms,detect_end_ms,a_end_ms,b_begin_ms,b_scored_ms,b_end_ms\n");
    for (const Row& w : rows) fprintf(f, "%lld,%d,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f\n", w.pair, w.lane, w.t[0], w.t[1], w.t[2], w.t[3], w.t[4], w.t[5]);
    fclose(f);
}

// Logical CPUs this process may keep busy: the smallest of its affinity mask, the container's CPU quota (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us; a process group that runs more busy threads than the quota is throttled -- every thread
// stopped until the period ends) and UVO_CPU_BUDGET, with which a launcher that knows how many ranks share the quota hands each
// rank its share (bench.py does).
static double host_cpu_budget()
{
    double b = 1e9;
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) b = CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; double per = 0;
        if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) b = std::min(b, atof(q) / per);
        fclose(f);
    } else {
        double q = -1, per = 0;
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &q) != 1) q = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &per) != 1) per = 0; fclose(g); }
        if (q > 0 && per > 0) b = std::min(b, q / per);
    }
    if (const char* e = getenv("UVO_CPU_BUDGET")) { const double v = atof(e); if (v > 0) b = std::min(b, v); }
    return b;
}
// The context's waiting policy for its current depth (uvo_ctx.h: worker_wait, stage_b_mode), stored where the lanes' own threads read
// it: an atomic per lane -- never the master's lane list, which set_depth rewrites while earlier lanes' workers are running.
static void apply_wait_policy(uvo_ctx* c)
{
    const int depth = (int)c->lanes.size();
    c->cpu_budget = host_cpu_budget();
    int eff = c->worker_wait == 0 ? 0 : (c->worker_wait == 2 ? 2 : 1);
    if (c->worker_wait == 1 && c->cpu_budget >= 2.0 * (depth + 2)) eff = 0;      // (logical CPUs: a core per thread -- two ranks of eight busy threads on eight CPUs each ran a 16-CPU container into its throttle: 343 pairs/s)
    for (Ctx* l : c->lanes) l->wait_eff.store(eff, std::memory_order_release);
    if (const char* e = getenv("UVO_STAGE_B")) c->stage_b_mode = !strcmp(e, "device") ? 1 : (!strcmp(e, "worker") ? 0 : -1);
}
// does a pipelined pair of this context take the device-driven PnP round (uvo_ctx.h: stage_b_mode)?
static bool stage_b_on_device(const uvo_ctx* c)
{
    if (c->stage_b_mode >= 0) return c->stage_b_mode == 1;
    return c->cpu_budget < (double)c->lanes.size() + 3.0;                          // no CPU for a worker per lane beside the submitter
}

static uvo_status set_depth(uvo_ctx* c, int depth)
{
    if (depth < 1 || depth > Ctx::kMaxDepth) { c->err = "pipeline depth must be 1..16"; return UVO_INVALID_ARG; }
    if (c->n_pending != 0) { c->err = "pipeline depth cannot change while pairs are in flight"; return UVO_INVALID_ARG; }
    while ((int)c->lanes.size() > depth) { destroy_one(static_cast<uvo_ctx*>(c->lanes.back())); c->lanes.pop_back(); }
    while ((int)c->lanes.size() < depth) {
        uvo_ctx* l = nullptr;
        uvo_status st = create_one(&c->p, c->device, c->max_w, c->max_h, c->cap, &l, (int)c->lanes.size());
        if (st != UVO_OK) { c->err = "could not allocate a further pipeline lane"; return st; }
        l->master = c; l->lane_id = (int)c->lanes.size(); l->timing = c->timing;
        if (c->trace_on && !l->trace_on && trace_alloc(l) != hipSuccess) { destroy_one(l); c->err = "could not allocate a further pipeline lane"; return UVO_HIP_ERROR; }
        c->lanes.push_back(l);
    }
    apply_wait_policy(c);
    // the previous pair's "after stereo match" set may live in a lane that no longer exists: restart the sequence
    c->warning.clear();
    {
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        const int nq = q ? atoi(q) : 4;
        if (depth > 2 && nq < 2 * depth)
            c->warning = "GPU_MAX_HW_QUEUES = " + std::to_string(nq) + " with " + std::to_string(depth) + " pipeline lanes (two HIP streams each): lanes share "
                         "hardware queues and the pipeline runs well below its rate; set GPU_MAX_HW_QUEUES >= " + std::to_string(2 * depth) + " before the HIP runtime starts";
    }
    if (c->primed_w > 0) UVO_TRY(prime_lanes(c, c->primed_w, c->primed_h));                 // lanes added to a running sequence
    if (c->prev_lane >= depth || c->next_lane >= depth) { c->vo_initialized = false; c->init_matches.clear(); c->prev_lane = 0; c->next_lane = 0; c->prev_sync = true; }
    return UVO_OK;
}

// Measurement (UVO_STREAM_ORDER = 1 | 2, with UVO_PIPELINE_DEPTH lanes): create and first-use the pool's streams in another order
// than lane by lane -- 1: every stage-A stream, then every PnP stream; 2: the PnP streams first -- to see what the binding of
// streams to hardware queues is worth.
static void precreate_streams(int device)
{
    static const int order = getenv("UVO_STREAM_ORDER") ? atoi(getenv("UVO_STREAM_ORDER")) : 0;
    if (order < 1 || order > 2 || !pool_on()) return;
    const int depth = getenv("UVO_PIPELINE_DEPTH") ? std::min((int)uvo::Ctx::kMaxDepth, std::max(1, atoi(getenv("UVO_PIPELINE_DEPTH")))) : 6;
    if (hipSetDevice(device) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int* sink = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&sink), 64) != hipSuccess) return;
    for (int pass = 0; pass < 2; pass++) {
        const int role = order == 1 ? pass : 1 - pass;
        for (int l = 0; l < depth; l++) {
            hipStream_t& slot = g_pool[device & 63][role][l];
            if (slot) continue;
            if (hipStreamCreateWithFlags(&slot, hipStreamNonBlocking) != hipSuccess) { slot = nullptr; continue; }
            hipLaunchKernelGGL(k_prime, dim3(1), dim3(64), 0, slot, sink, 1);
            (void)hipStreamSynchronize(slot);
        }
    }
    (void)hipFree(sink);
}

extern "C" uvo_status uvo_ctx_create(const uvo_params* p, int device, int max_w, int max_h, int max_kpts, uvo_ctx** out)
try {
    precreate_streams(device);
    uvo_status st = create_one(p, device, max_w, max_h, max_kpts, out);
    if (st != UVO_OK) return st;
    (*out)->lanes.push_back(*out);
    if (getenv("UVO_MAX_B")) (*out)->max_b = std::min(16, std::max(1, atoi(getenv("UVO_MAX_B"))));
    if (getenv("UVO_A_OVERLAP")) (*out)->a_overlap = std::min(16, std::max(0, atoi(getenv("UVO_A_OVERLAP"))));
    if (getenv("UVO_MAX_B_MONO")) (*out)->max_b_mono = std::min(16, std::max(1, atoi(getenv("UVO_MAX_B_MONO"))));
    if (getenv("UVO_A_OVERLAP2")) (*out)->a_overlap2 = std::min(4, std::max(0, atoi(getenv("UVO_A_OVERLAP2"))));
    if (getenv("UVO_BATCH")) (*out)->batch = atoi(getenv("UVO_BATCH")) == 2 ? 2 : 1;
    if (getenv("UVO_A_OVERLAP_MONO")) (*out)->a_overlap_mono = std::min(8, std::max(0, atoi(getenv("UVO_A_OVERLAP_MONO"))));
    st = set_depth(*out, 2);                      // two pairs in flight by default (uvo_stereo_set_depth changes it)
    if (st != UVO_OK) { uvo_ctx_destroy(*out); *out = nullptr; }
    return st;
} UVO_ABI_CATCH(nullptr)

extern "C" uvo_status uvo_stereo_set_batch(uvo_ctx* c, int pairs)
try {
    if (!c) return UVO_INVALID_ARG;
    if (pairs != 1 && pairs != 2) { c->err = "uvo_stereo_set_batch: 1 or 2 pairs per launch set"; return UVO_INVALID_ARG; }
    if (c->n_pending != 0) { c->err = "the launch mode cannot change while pairs are in flight"; return UVO_INVALID_ARG; }
    c->batch = pairs;
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_stereo_set_depth(uvo_ctx* c, int depth)
try {
    if (!c) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    return set_depth(c, depth);
} UVO_ABI_CATCH(c)

extern "C" void uvo_ctx_destroy(uvo_ctx* c)
try {
    if (!c) return;
    if (uvo::g_bdbg && uvo::g_bstat[4].load() > 0) {
        const double n = uvo::g_bstat[4].load();
        fprintf(stderr, "[uvo] stage B over %.0f calls, host wall us per call: wait-for-A %.1f | hyp+score to sync %.1f | host scan + refit launch %.1f | refit to sync %.1f | semaphore wait %.1f\n",
                n, uvo::g_bstat[0].load() / n, uvo::g_bstat[1].load() / n, uvo::g_bstat[2].load() / n, uvo::g_bstat[3].load() / n, uvo::g_bstat[5].load() / n);
        fprintf(stderr, "[uvo] uvo_stereo_submit host wall: %.1f us per pair over %.0f pairs\n", uvo::g_bstat[6].load() / std::max(1.0, uvo::g_bstat[7].load()), uvo::g_bstat[7].load());
        const double np = std::max(1.0, uvo::g_bstat[7].load());
        fprintf(stderr, "[uvo]   of which: upload %.1f | a_overlap wait %.1f | detector launches %.1f | matcher .. extract_3Dpoints launches %.1f\n",
                uvo::g_bstat[8].load() / np, uvo::g_bstat[9].load() / np, uvo::g_bstat[10].load() / np, uvo::g_bstat[11].load() / np);
    }
    write_trace(c);
    for (size_t i = c->lanes.size(); i > 1; i--) destroy_one(static_cast<uvo_ctx*>(c->lanes[i - 1]));
    c->lanes.clear();
    destroy_one(c);
} UVO_ABI_CATCH_VOID(c)

static void destroy_one(uvo_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->worker.joinable()) {
        { std::lock_guard<std::mutex> lk(c->mu); c->quit = true; }
        c->cv.notify_all();
        c->worker.join();
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(c->d_img[i]); (void)hipFree(c->d_sum_base[i]); (void)hipFree(c->d_planes[i]); (void)hipFree(c->d_cand[i]); (void)hipFree(c->det[i].kps);
        (void)hipFree(c->det[i].desc); (void)hipFree(c->d_tmp_desc[i]); (void)hipFree(c->d_matches[i]);
        (void)hipFree(c->d_as_kpsL[i]); (void)hipFree(c->d_as_kpsR[i]); (void)hipFree(c->d_as_descL[i]);
        (void)hipFree(c->d_as_pts4[i]); (void)hipFree(c->d_as_cam1[i]); (void)hipFree(c->d_as_flag[i]);
    }
    if (c->pnp_stream) (void)hipStreamSynchronize(c->pnp_stream);
    mono_ws_free(c);
    pre_ws_free(c);
    codec_ws_free(c);
    sift_ws_free(c);
    akaze_ws_free(c);
    orb_ws_free(c);
    void* ptrs[] = { c->d_hess_order, c->d_surv, c->d_octpat, c->d_colpart, c->d_DW, c->d_rank, c->d_big_par, c->d_big_patch, c->d_area_tabs, c->d_area_iscale, c->d_ori_w, c->d_mpart, c->d_mscratch, c->d_knn_idx, c->d_knn_dist, c->d_x1, c->d_x2, c->d_xc, c->d_pts4, c->d_cam1,
                     c->d_flag, c->d_tmp_idx, c->d_tmp_row, c->d_good_pts[0], c->d_good_pts[1], c->d_good_idx[0], c->d_good_idx[1], c->d_opts[0], c->d_opts[1],
                     c->d_ipts[0], c->d_ipts[1], c->d_counts, c->d_countsB, c->d_subsets, c->d_models,
                     c->d_hcount, c->d_inliers, c->d_refit, c->d_pose };
    for (void* p : ptrs) (void)hipFree(p);
    (void)hipHostFree(c->h_counts); (void)hipHostFree(c->h_subsets); (void)hipHostFree(c->h_hcount); (void)hipHostFree(c->h_pose);
    (void)hipHostFree(c->h_countsB); (void)hipHostFree(c->h_spec); (void)hipFree(c->d_rng_raw); (void)hipFree(c->d_spec);
    for (int i = 0; i < 2; i++) { (void)hipHostFree(c->h_countsA[i]); if (c->evA[i]) (void)hipEventDestroy(c->evA[i]); }
    {
        const char* pe = getenv("UVO_PNP_PRIORITY");
        if (c->pnp_stream && pe && atoi(pe) != 0) (void)hipStreamDestroy(c->pnp_stream);      // a priority stream is not pooled
        else stream_release(c->device, 1, c->lane_id, c->pnp_stream);
    }
    if (c->evAS) (void)hipEventDestroy(c->evAS);
    if (c->evBlock) (void)hipEventDestroy(c->evBlock);
    if (c->evPoll) (void)hipEventDestroy(c->evPoll);
    if (c->evB) (void)hipEventDestroy(c->evB);
    if (c->evSync) (void)hipEventDestroy(c->evSync);
    if (c->evProducer) (void)hipEventDestroy(c->evProducer);
    for (auto& r : c->trace) for (int k = 0; k < 8; k++) if (r.ev[k]) (void)hipEventDestroy(r.ev[k]);
    if (c->evDet) (void)hipEventDestroy(c->evDet);
    if (c->evPrevRead) (void)hipEventDestroy(c->evPrevRead);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    stream_release(c->device, 0, c->lane_id, c->stream);
    delete c;
}

extern "C" const char* uvo_last_error(const uvo_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" void* uvo_ctx_stream(uvo_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" uvo_status uvo_ctx_set_params(uvo_ctx* c, const uvo_params* p)
try {
    if (!c || !p) return UVO_INVALID_ARG;
    if (c->n_pending != 0) { c->err = "parameters cannot change while pairs are in flight"; return UVO_INVALID_ARG; }
    if (const char* why = unsupported_params(p)) { c->err = why; return UVO_INVALID_ARG; }
    if ((p->SURF_EXTENDED != 0) != (c->p.SURF_EXTENDED != 0) && (c->vo_initialized || c->mono_initialized)) {
        c->err = "SURF_EXTENDED changes the descriptor width: the previous frame's descriptors held by the running VO loop would not match (uvo_stereo_reset / uvo_mono_reset first)";
        return UVO_INVALID_ARG;
    }
    for (Ctx* l : c->lanes) l->p = *p;
    return UVO_OK;
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_ctx_set_producer_stream(uvo_ctx* c, void* hip_stream, int enabled)
try {
    if (!c) return UVO_INVALID_ARG;
    c->producer_stream = static_cast<hipStream_t>(hip_stream); c->has_producer = enabled != 0;
    return UVO_OK;
} UVO_ABI_CATCH(c)
extern "C" const char* uvo_ctx_warning(const uvo_ctx* c) { return c ? c->warning.c_str() : ""; }
extern "C" int uvo_ctx_pending(const uvo_ctx* c) { return c ? c->n_pending : 0; }
extern "C" const char* uvo_ctx_host_policy(uvo_ctx* c)
try {
    if (!c) return "";
    static const char* const wn[3] = { "poll", "timed-sleep+poll", "interrupt" };
    char buf[160];
    snprintf(buf, sizeof(buf), "wait=%s stage_b=%s cpu_budget=%.2f depth=%d", wn[c->wait_eff.load() % 3], stage_b_on_device(c) ? "device" : "worker",
             c->cpu_budget > 1e8 ? -1.0 : c->cpu_budget, (int)c->lanes.size());
    c->policy_text = buf;
    return c->policy_text.c_str();
} UVO_ABI_CATCH_RET(c, "")

// UVO_MEM_DEVICE inputs: order lane L's next reads after the work queued so far on the declared producer stream
static uvo_status wait_for_producer(uvo_ctx* m, Ctx* L, int mem)
{
    if (mem != UVO_MEM_DEVICE || !m->has_producer) return UVO_OK;
    UVO_HIP_TRY(m, hipEventRecord(L->evProducer, m->producer_stream));
    UVO_HIP_TRY(m, hipStreamWaitEvent(L->stream, L->evProducer, 0));
    return UVO_OK;
}
// the standalone operators work on lane 0's buffers and stream: not while pipelined pairs or frames are in flight
static uvo_status need_idle(uvo_ctx* c, const char* who)
{
    if (c->n_pending == 0) return UVO_OK;
    c->err = std::string(who) + ": collect the pairs / frames in flight first (the operator uses lane 0's buffers)";
    return UVO_INVALID_ARG;
}

static uvo_status fail(uvo_ctx* c, uvo_status s, const char* msg) { c->err = msg; return s; }

namespace uvo {
// Host waits of the pipelines' hot paths poll with queries of their own.  hipStreamSynchronize / hipEventSynchronize spin for ~100 us and
// then sleep on the signal's interrupt, and on this stack about one such sleep in a hundred wakes 2-4 ms late (round 4: pipeline_trace
// showed 3.3 and 4.4 ms between the end of a stage A and its worker's first PnP launch; the driver's round-3 record of 1990 pairs/s
// was one of those inside a 10 ms window).  A query never sleeps.
static inline void poll_pause() { for (int i = 0; i < 40; i++) __builtin_ia32_pause(); }
hipError_t poll_event(hipEvent_t ev)
{
    hipError_t e;
    while ((e = hipEventQuery(ev)) == hipErrorNotReady) poll_pause();
    return e;
}
hipError_t host_sync(Ctx* c, hipStream_t st)
{
    // (an event of the lane's own and hipEventQuery -- a load of the signal; hipStreamQuery in a loop cost the 600-step form 6 % and put
    // millisecond gaps back: p99 of the collect gaps 0.95 ms against 0.33)
    if (c->wait_eff.load(std::memory_order_acquire) != 2) { const hipError_t e = hipEventRecord(c->evPoll, st); return e == hipSuccess ? poll_event(c->evPoll) : e; }     // every mode but block-all
    hipError_t e = hipEventRecord(c->evBlock, st);
    return e == hipSuccess ? hipEventSynchronize(c->evBlock) : e;
}
}

// ------------------------------------------------------------------------------------------ SURF
static uvo_status read_counts(uvo_ctx* c)
{
    UVO_HIP_TRY(c, hipMemcpyAsync(c->h_counts, c->d_counts, sizeof(int) * CN_TOTAL, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
}

static uvo_status check_cand_overflow(uvo_ctx* c, int nimg)
{
    if (c->h_counts[CN_ORI_DROP] != 0)
        return fail(c, UVO_INVALID_ARG, "SURF orientation: a keypoint had no gradient sample inside the image; OpenCV drops such keypoints, which this build does not do");
    for (int i = 0; i < nimg; i++)
        if (c->h_counts[CN_CAND0 + i] > c->cap)
            return fail(c, UVO_CAPACITY, "SURF found more keypoints than the context's max_kpts; results would be order-dependent");
    return UVO_OK;
}

// detect_features inside the fused steps: the SURF branch (VOU:114-119) or, when the context's feature detector is "SIFT", VOU:107-112
static uvo_status detect_dispatch(Ctx* c, int nimg, int gate_min_features = -1)
{
    return c->use_sift() ? sift_detect_lane(c, nimg, gate_min_features) : surf_detect(c, nimg, gate_min_features);
}

extern "C" uvo_status uvo_ctx_set_feature_detector(uvo_ctx* c, const char* name)
try {
    if (!c || !name) return UVO_INVALID_ARG;
    const bool sift = strcmp(name, "SIFT") == 0;
    if (!sift && strcmp(name, "SURF") != 0) return fail(c, UVO_INVALID_ARG, "uvo_ctx_set_feature_detector: the fused steps run on \"SURF\" or \"SIFT\" (AKAZE and ORB are the operators uvo_akaze_detect / uvo_orb_detect + the Hamming matcher)");
    if (c->n_pending != 0) return fail(c, UVO_INVALID_ARG, "the feature detector cannot change while pairs are in flight");
    if ((sift ? 1 : 0) != c->feature_sift && (c->vo_initialized || c->mono_initialized))
        return fail(c, UVO_INVALID_ARG, "the feature detector changes the descriptors: the previous frame's set held by the running VO loop would not match (uvo_stereo_reset / uvo_mono_reset first)");
    if (!sift && c->feature_sift) {                                     // back to SURF: the lanes' scale-space workspaces (0.5 GB per image slot) go
        (void)hipSetDevice(c->device);
        for (Ctx* l : c->lanes) { if (l->stream) (void)hipStreamSynchronize(l->stream); if (l->pnp_stream) (void)hipStreamSynchronize(l->pnp_stream); sift_ws_free(l); }
    }
    c->feature_sift = sift ? 1 : 0;
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_surf_detect(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem,
                                      uvo_keypoint* kps, float* desc, int cap, int* n)
try {
    if (!c || !n) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_surf_detect"));
    UVO_TRY(wait_for_producer(c, c, mem));
    UVO_TRY(surf_upload(c, 0, gray, w, h, stride, mem));
    UVO_TRY(surf_detect(c, 1));
    UVO_TRY(read_counts(c));
    UVO_TRY(check_cand_overflow(c, 1));
    int cnt = c->h_counts[CN_NL];
    *n = cnt;
    if ((kps || desc) && cnt > cap) return fail(c, UVO_CAPACITY, "uvo_surf_detect: output capacity too small");
    if (kps && cnt) UVO_HIP_TRY(c, hipMemcpyAsync(kps, c->det[0].kps, sizeof(uvo_keypoint) * cnt, hipMemcpyDeviceToHost, c->stream));
    const int dsize = c->p.SURF_EXTENDED ? 128 : 64;
    if (desc && cnt) UVO_HIP_TRY(c, hipMemcpyAsync(desc, c->det[0].desc, sizeof(float) * dsize * cnt, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_sift_detect(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, int nfeatures, int n_octave_layers,
                                      double contrast_threshold, double edge_threshold, double sigma, uvo_keypoint* kps, float* desc, int cap, int* n)
try {
    if (!c || !n || !gray) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_sift_detect"));
    UVO_TRY(wait_for_producer(c, c, mem));
    return sift_detect(c, gray, w, h, stride, mem, nfeatures, n_octave_layers, contrast_threshold, edge_threshold, sigma, kps, desc, cap, n);
} UVO_ABI_CATCH(c)

// detect_features' AKAZE branch (VO_utility.cpp:93-98): AKAZE::create()->detectAndCompute; 61-byte M-LDB rows for the Hamming matcher
extern "C" uvo_status uvo_akaze_detect(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, uvo_keypoint* kps, uint8_t* desc, int cap, int* n)
try {
    if (!c || !n || !gray) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_akaze_detect"));
    UVO_TRY(wait_for_producer(c, c, mem));
    return akaze_detect(c, gray, w, h, stride, mem, kps, desc, cap, n);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_akaze_plane(uvo_ctx* c, int level, int what, float* out, int cap_floats, int* w, int* h)
try {
    if (!c || !out || !w || !h) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_akaze_plane"));
    return akaze_plane(c, level, what, out, cap_floats, w, h);
} UVO_ABI_CATCH(c)

// detect_features' ORB branch (VO_utility.cpp:100-105): ORB::create(10000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, 10)->detectAndCompute; 32-byte rBRIEF
// rows for the Hamming matcher.  The sampling table (OpenCV's bit_pattern_31_) is the caller's: uvo_orb_set_pattern.
extern "C" uvo_status uvo_orb_configure(uvo_ctx* c, int nfeatures, float scale_factor, int nlevels, int edge_threshold, int patch_size, int fast_threshold)
try {
    if (!c) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_orb_configure"));
    return orb_configure(c, nfeatures, scale_factor, nlevels, edge_threshold, patch_size, fast_threshold);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_orb_set_pattern(uvo_ctx* c, const int* pattern)
try {
    if (!c) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_orb_set_pattern"));
    return orb_set_pattern(c, pattern);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_orb_detect(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, uvo_keypoint* kps, uint8_t* desc, int cap, int* n)
try {
    if (!c || !n || !gray) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_orb_detect"));
    UVO_TRY(wait_for_producer(c, c, mem));
    return orb_detect(c, gray, w, h, stride, mem, kps, desc, cap, n);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_orb_plane(uvo_ctx* c, int level, int what, uint8_t* out, int cap_bytes, int* w, int* h)
try {
    if (!c || !out || !w || !h) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_orb_plane"));
    return orb_level_plane(c, level, what, out, cap_bytes, w, h);
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_sift_layer(uvo_ctx* c, int octave, int layer, int dog, float* out, int cap_floats, int* w, int* h)
try {
    if (!c || !w || !h) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_sift_layer"));
    return sift_layer(c, octave, layer, dog, out, cap_floats, w, h);
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_integral(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, int32_t* sum)
try {
    if (!c || !sum) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_integral"));
    UVO_TRY(wait_for_producer(c, c, mem));
    UVO_TRY(surf_upload(c, 0, gray, w, h, stride, mem));
    UVO_TRY(surf_integral(c, 1));
    UVO_HIP_TRY(c, hipMemcpyAsync(sum, c->d_sum[0], sizeof(int32_t) * (size_t)(w + 1) * (h + 1), hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_hessian_layer(uvo_ctx* c, int octave, int layer, float* det, float* trace)
try {
    if (!c || !det || !trace) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    return surf_hessian_layer_debug(c, octave, layer, det, trace);
} UVO_ABI_CATCH(c)

// ------------------------------------------------------------------------------------------ matching
static uvo_status stage_desc(uvo_ctx* c, int slot, const float* d, int n, int mem, const float** out)
{
    if (n > c->cap) return fail(c, UVO_CAPACITY, "descriptor count exceeds the context's max_kpts");
    if (mem == UVO_MEM_DEVICE) { *out = d; return UVO_OK; }
    if (n) UVO_HIP_TRY(c, hipMemcpyAsync(c->d_tmp_desc[slot], d, sizeof(float) * c->desc_dim() * (size_t)n, hipMemcpyHostToDevice, c->stream));
    *out = c->d_tmp_desc[slot];
    return UVO_OK;
}

// Row width of the standalone matchers for the duration of one call.  The entries without a `dim` argument match THIS CONTEXT'S SURF
// rows -- SURF::descriptorSize() = 64, or 128 with SURF_EXTENDED -- whatever detector the fused steps are switched to
// (uvo_ctx_set_feature_detector): their callers hand over n x 64 buffers, and the width of the loops' detector is not theirs.
namespace { struct DimScope { uvo_ctx* c; int prev; DimScope(uvo_ctx* c_, int d) : c(c_), prev(c_->match_dim) { if (!prev) c->match_dim = d; } ~DimScope() { c->match_dim = prev; } };
            int surf_dim(const uvo_ctx* c) { return c->p.SURF_EXTENDED ? 128 : 64; } }
extern "C" uvo_status uvo_match_knn2(uvo_ctx* c, const float* d1, int n1, const float* d2, int n2, int mem, int* idx, float* dist)
try {
    if (!c || n1 < 0 || n2 < 0 || (n1 && !d1) || (n2 && !d2) || !idx || !dist) return UVO_INVALID_ARG;
    DimScope surf_rows(c, surf_dim(c));
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_match_knn2"));
    if (n1 == 0) return UVO_OK;
    if (n2 == 0) { for (int i = 0; i < 2 * n1; i++) { idx[i] = -1; dist[i] = FLT_MAX; } return UVO_OK; }
    UVO_TRY(wait_for_producer(c, c, mem));
    const float *q, *t;
    UVO_TRY(stage_desc(c, 0, d1, n1, mem, &q));
    UVO_TRY(stage_desc(c, 1, d2, n2, mem, &t));
    UVO_TRY(match_knn2(c, q, nullptr, n1, t, nullptr, n2));
    UVO_HIP_TRY(c, hipMemcpyAsync(idx, c->d_knn_idx, sizeof(int) * 2 * n1, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(dist, c->d_knn_dist, sizeof(float) * 2 * n1, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_match_knn2_ratio(uvo_ctx* c, const float* d1, int n1, const float* d2, int n2, int mem,
                                           float ratio, uvo_dmatch* out, int cap, int* m)
try {
    if (!c || n1 < 0 || n2 < 0 || (n1 && !d1) || (n2 && !d2) || !out || !m || *m < 0) return UVO_INVALID_ARG;
    DimScope surf_rows(c, surf_dim(c));
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_match_knn2_ratio"));
    if (n1 == 0 || n2 == 0) return UVO_OK;          // knnMatch on an empty query/train set yields no matches
    UVO_TRY(wait_for_producer(c, c, mem));
    const float *q, *t;
    UVO_TRY(stage_desc(c, 0, d1, n1, mem, &q));
    UVO_TRY(stage_desc(c, 1, d2, n2, mem, &t));
    UVO_TRY(match_knn2(c, q, nullptr, n1, t, nullptr, n2));
    UVO_TRY(match_ratio_compact(c, nullptr, n1, ratio, c->d_matches[0], c->d_nmatch, c->cap));
    UVO_TRY(read_counts(c));
    int cnt = c->h_counts[CN_M];
    if (*m + cnt > cap) return fail(c, UVO_CAPACITY, "uvo_match_knn2_ratio: output capacity too small");
    if (cnt) UVO_HIP_TRY(c, hipMemcpyAsync(out + *m, c->d_matches[0], sizeof(uvo_dmatch) * cnt, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *m += cnt;                                       // appended, as VOU:538
    return UVO_OK;
} UVO_ABI_CATCH(c)

// match_features' L2 arm for descriptors that are not this context's SURF rows (VO_utility.cpp:525-529 sends "SIFT" -- 128 floats
// per row whatever SURF_EXTENDED says -- to the same BFMatcher(NORM_L2)): the row width is given per call.
extern "C" uvo_status uvo_match_knn2_dim(uvo_ctx* c, const float* d1, int n1, const float* d2, int n2, int dim, int mem, int* idx, float* dist)
try {
    if (!c) return UVO_INVALID_ARG;
    if (dim != 64 && dim != 128) return fail(c, UVO_INVALID_ARG, "uvo_match_knn2_dim: rows of 64 or 128 floats");
    DimScope ds(c, dim);
    return uvo_match_knn2(c, d1, n1, d2, n2, mem, idx, dist);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_match_knn2_ratio_dim(uvo_ctx* c, const float* d1, int n1, const float* d2, int n2, int dim, int mem,
                                               float ratio, uvo_dmatch* out, int cap, int* m)
try {
    if (!c) return UVO_INVALID_ARG;
    if (dim != 64 && dim != 128) return fail(c, UVO_INVALID_ARG, "uvo_match_knn2_ratio_dim: rows of 64 or 128 floats");
    DimScope ds(c, dim);
    return uvo_match_knn2_ratio(c, d1, n1, d2, n2, mem, ratio, out, cap, m);
} UVO_ABI_CATCH(c)

// The AKAZE / ORB branch of match_features (VO_utility.cpp:520-524): binary descriptors, Hamming distance
static uvo_status stage_bytes(uvo_ctx* c, int slot, const uint8_t* d, int n, int bytes, int mem, const uint8_t** out)
{
    if (n > c->cap) return fail(c, UVO_CAPACITY, "descriptor count exceeds the context's max_kpts");
    if (bytes < 1 || bytes > 64) return fail(c, UVO_INVALID_ARG, "binary descriptor rows of 1..64 bytes");
    if (mem == UVO_MEM_DEVICE) { *out = d; return UVO_OK; }
    if (n) UVO_HIP_TRY(c, hipMemcpyAsync(c->d_tmp_desc[slot], d, (size_t)bytes * n, hipMemcpyHostToDevice, c->stream));     // cap x 512 bytes of staging
    *out = reinterpret_cast<const uint8_t*>(c->d_tmp_desc[slot]);
    return UVO_OK;
}
extern "C" uvo_status uvo_match_knn2_hamming(uvo_ctx* c, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int bytes, int mem, int* idx, float* dist)
try {
    if (!c || n1 < 0 || n2 < 0 || (n1 && !d1) || (n2 && !d2) || !idx || !dist) return UVO_INVALID_ARG;
    if (bytes < 1 || bytes > 64) return fail(c, UVO_INVALID_ARG, "uvo_match_knn2_hamming: descriptor rows of 1..64 bytes");    // before anything is staged
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_match_knn2_hamming"));
    if (n1 == 0) return UVO_OK;
    if (n2 == 0) { for (int i = 0; i < 2 * n1; i++) { idx[i] = -1; dist[i] = FLT_MAX; } return UVO_OK; }
    UVO_TRY(wait_for_producer(c, c, mem));
    const uint8_t *q, *t;
    UVO_TRY(stage_bytes(c, 0, d1, n1, bytes, mem, &q));
    UVO_TRY(stage_bytes(c, 1, d2, n2, bytes, mem, &t));
    UVO_TRY(match_knn2_hamming(c, q, n1, t, n2, bytes));
    UVO_HIP_TRY(c, hipMemcpyAsync(idx, c->d_knn_idx, sizeof(int) * 2 * n1, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(dist, c->d_knn_dist, sizeof(float) * 2 * n1, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_match_knn2_ratio_hamming(uvo_ctx* c, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int bytes, int mem,
                                                   float ratio, uvo_dmatch* out, int cap, int* m)
try {
    if (!c || n1 < 0 || n2 < 0 || (n1 && !d1) || (n2 && !d2) || !out || !m || *m < 0) return UVO_INVALID_ARG;
    if (bytes < 1 || bytes > 64) return fail(c, UVO_INVALID_ARG, "uvo_match_knn2_ratio_hamming: descriptor rows of 1..64 bytes");   // before anything is staged
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_match_knn2_ratio_hamming"));
    if (n1 == 0 || n2 == 0) return UVO_OK;
    UVO_TRY(wait_for_producer(c, c, mem));
    const uint8_t *q, *t;
    UVO_TRY(stage_bytes(c, 0, d1, n1, bytes, mem, &q));
    UVO_TRY(stage_bytes(c, 1, d2, n2, bytes, mem, &t));
    UVO_TRY(match_knn2_hamming(c, q, n1, t, n2, bytes));
    UVO_TRY(match_ratio_compact(c, nullptr, n1, ratio, c->d_matches[0], c->d_nmatch, c->cap));
    UVO_TRY(read_counts(c));
    int cnt = c->h_counts[CN_M];
    if (*m + cnt > cap) return fail(c, UVO_CAPACITY, "uvo_match_knn2_ratio_hamming: output capacity too small");
    if (cnt) UVO_HIP_TRY(c, hipMemcpyAsync(out + *m, c->d_matches[0], sizeof(uvo_dmatch) * cnt, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *m += cnt;                                       // appended, as VOU:538
    return UVO_OK;
} UVO_ABI_CATCH(c)

// ------------------------------------------------------------------------------------------ geometry operators
extern "C" uvo_status uvo_triangulate_points(uvo_ctx* c, const double* P1, const double* P2,
                                             const uvo_point2f* x1, const uvo_point2f* x2, int n, float* out4xn)
try {
    if (!c || !P1 || !P2 || n < 0 || (n && (!x1 || !x2 || !out4xn))) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_triangulate_points"));
    if (n == 0) return UVO_OK;
    if (n > c->cap) return fail(c, UVO_CAPACITY, "point count exceeds the context's max_kpts");
    UVO_HIP_TRY(c, hipMemcpyAsync(c->d_x1, x1, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(c->d_x2, x2, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, c->stream));
    UVO_TRY(pose_triangulate(c, P1, P2, nullptr, n));
    std::vector<float4> tmp(n);
    UVO_HIP_TRY(c, hipMemcpyAsync(tmp.data(), c->d_pts4, sizeof(float4) * n, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) { out4xn[i] = tmp[i].x; out4xn[n + i] = tmp[i].y; out4xn[2*n + i] = tmp[i].z; out4xn[3*n + i] = tmp[i].w; }
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_extract_3d_points(uvo_ctx* c, const uvo_point2f* k1, const uvo_point2f* k2, int n,
                                            const double* R1, const double* t1, const double* R2, const double* t2,
                                            const double* K1, const double* K2, const float* points4d,
                                            double* pts, int* idx, int* g)
try {
    if (!c || n < 0 || !g || !R1 || !t1 || !R2 || !t2 || !K1 || !K2 || (n && (!k1 || !k2 || !points4d || !pts || !idx))) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    *g = 0;
    UVO_TRY(need_idle(c, "uvo_extract_3d_points"));
    if (n == 0) return UVO_OK;
    if (n > c->cap) return fail(c, UVO_CAPACITY, "point count exceeds the context's max_kpts");
    std::vector<float4> tmp(n);
    for (int i = 0; i < n; i++) tmp[i] = make_float4(points4d[i], points4d[n + i], points4d[2*n + i], points4d[3*n + i]);
    UVO_HIP_TRY(c, hipMemcpyAsync(c->d_pts4, tmp.data(), sizeof(float4) * n, hipMemcpyHostToDevice, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(c->d_x1, k1, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(c->d_x2, k2, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(c->d_xc, k1, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, c->stream));
    UVO_TRY(pose_extract3d(c, 0, R1, t1, R2, t2, K1, K2, nullptr, n));
    UVO_TRY(read_counts(c));
    int G = c->h_counts[CN_G];
    if (G) {
        UVO_HIP_TRY(c, hipMemcpyAsync(pts, c->d_good_pts[0], sizeof(double) * 3 * G, hipMemcpyDeviceToHost, c->stream));
        UVO_HIP_TRY(c, hipMemcpyAsync(idx, c->d_good_idx[0], sizeof(int) * G, hipMemcpyDeviceToHost, c->stream));
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    *g = G;
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_reproject_errors(uvo_ctx* c, const double* world, int n, const double* R, const double* t,
                                           const double* K, const uvo_point2f* img, double* err)
try {
    if (!c || n < 0 || !R || !t || !K || (n && (!world || !img || !err))) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_reproject_errors"));
    if (n == 0) return UVO_OK;
    if (n > c->cap) return fail(c, UVO_CAPACITY, "point count exceeds the context's max_kpts");
    return pose_reproject_errors(c, world, n, R, t, K, img, err);
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_solve_pnp_ransac(uvo_ctx* c, const double* obj, const uvo_point2f* img, int n, const double* K,
                                           int iterations_count, float reprojection_error, double confidence,
                                           double* rvec, double* tvec, int* inliers, int* n_inliers, int* ok)
try {
    if (!c || !obj || !img || !K || !rvec || !tvec || !n_inliers || !ok || n < 0) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_solve_pnp_ransac"));
    if (n > c->cap) return fail(c, UVO_CAPACITY, "point count exceeds the context's max_kpts");
    std::vector<float> of((size_t)3 * n);
    for (int i = 0; i < 3 * n; i++) of[i] = (float)obj[i];                     // opoints0.convertTo(opoints, CV_32F)
    if (n) {
        UVO_HIP_TRY(c, hipMemcpyAsync(c->d_opts[0], of.data(), sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
        UVO_HIP_TRY(c, hipMemcpyAsync(c->d_ipts[0], img, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, c->stream));
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    int ni = 0;
    UVO_TRY(pose_pnp_ransac(c, 0, n, K, iterations_count, reprojection_error, confidence, rvec, tvec, &ni, ok));
    *n_inliers = ni;
    if (inliers && ni) {
        UVO_HIP_TRY(c, hipMemcpyAsync(inliers, c->d_inliers, sizeof(int) * ni, hipMemcpyDeviceToHost, c->pnp_stream));
        UVO_HIP_TRY(c, hipStreamSynchronize(c->pnp_stream));
    }
    return UVO_OK;
} UVO_ABI_CATCH(c)

// host fp64; the matrix -> vector direction needs the 3x3 Jacobi SVD
extern "C" uvo_status uvo_rodrigues(const double* in, int n_in, double* out)
try {
    if (!in || !out) return UVO_INVALID_ARG;
    if (n_in == 3) { rodrigues_vec2mat(in, out); return UVO_OK; }
    if (n_in == 9) {
        double Rm[9], sc[33];
        memcpy(Rm, in, sizeof(Rm));
        rodrigues_mat2vec(SArr<1>{Rm}, SArr<1>{sc}, out);
        return UVO_OK;
    }
    return UVO_INVALID_ARG;
} UVO_ABI_CATCH(nullptr)

// ------------------------------------------------------------------------------------------ stereo step
// VO:569-579: curr_{left,right}_{descr,keypoints}_after_stereo_match by the stereo matches' indices
__device__ __forceinline__ void gather_after_stereo(int bx, const uvo_dmatch* m, const int* cn, const uvo_keypoint* kL, const uvo_keypoint* kR,
                                                    const float* dL, uvo_keypoint* okL, uvo_keypoint* okR, float* odL, int dim)
{
    const int meff = cn[CN_MEFF];
    const int row = bx * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (row >= meff) return;
    const int q = m[row].queryIdx, t = m[row].trainIdx;
    for (int v = sub; v < dim / 4; v += 16) reinterpret_cast<float4*>(odL + (size_t)row * dim)[v] = reinterpret_cast<const float4*>(dL + (size_t)q * dim)[v];
    if (sub == 0) { okL[row] = kL[q]; okR[row] = kR[t]; }
}
// VO:601-617, 637-640: points of the triangular matches (prev left / prev right by queryIdx, curr left by trainIdx).  pmap (two-pair
// launch): the previous pair's set is not gathered yet -- its row q is keypoint pmap[q].queryIdx of that pair's left list and
// pmap[q].trainIdx of its right list (pL, pR are then those lists)
__device__ __forceinline__ void gather_triangular(int bx, const uvo_dmatch* m, const int* cn, const uvo_keypoint* pL, const uvo_keypoint* pR,
                                                  const uvo_keypoint* cL, uvo_point2f* x1, uvo_point2f* x2, uvo_point2f* xc, const uvo_dmatch* pmap)
{
    const int T = cn[CN_T];
    const int i = bx * 256 + threadIdx.x;
    if (i >= T) return;
    const int q = m[i].queryIdx, t = m[i].trainIdx;
    const int ql = pmap ? pmap[q].queryIdx : q, qr = pmap ? pmap[q].trainIdx : q;
    x1[i] = uvo_point2f{pL[ql].x, pL[ql].y};
    x2[i] = uvo_point2f{pR[qr].x, pR[qr].y};
    xc[i] = uvo_point2f{cL[t].x, cL[t].y};
}
// both gathers of a stereo step in one launch: blocks [0, n_as) build this pair's "after stereo match" set, the rest the point
// pairs of the triangular matches; blockIdx.y: the pair of a two-pair launch
struct GatherArgs { const uvo_dmatch* m_s; const uvo_dmatch* m_t; const int* cn; const uvo_keypoint* kL; const uvo_keypoint* kR; const float* dL;
                    uvo_keypoint* okL; uvo_keypoint* okR; float* odL; int dim; const uvo_keypoint* pL; const uvo_keypoint* pR;
                    uvo_point2f* x1; uvo_point2f* x2; uvo_point2f* xc; int n_as; const uvo_dmatch* pmap; };
struct GatherPair { GatherArgs g[2]; };
__global__ __launch_bounds__(256) void k_gather_stereo_step(GatherPair gp)
{
    const GatherArgs& a = gp.g[blockIdx.y];
    if ((int)blockIdx.x < a.n_as) gather_after_stereo(blockIdx.x, a.m_s, a.cn, a.kL, a.kR, a.dL, a.okL, a.okR, a.odL, a.dim);
    else gather_triangular(blockIdx.x - a.n_as, a.m_t, a.cn, a.pL, a.pR, a.kL, a.x1, a.x2, a.xc, a.pmap);
}
__global__ void k_gather_kps_idx(const uvo_keypoint* src, const int* idx, int n, uvo_keypoint* dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_gather_desc_idx(const float* src, int nsrc, const int* idx, int n, float* dst, int dim)
{
    int row = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (row >= n) return;
    int q = idx[row];
    for (int e = sub; e < dim / 4; e += 16) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);      // the reference leaves an out-of-range row uninitialised (VOU:690)
        if (q >= 0 && q < nsrc) v = reinterpret_cast<const float4*>(src + (size_t)q * dim)[e];
        reinterpret_cast<float4*>(dst + (size_t)row * dim)[e] = v;
    }
}

extern "C" uvo_status uvo_stereo_set_rig(uvo_ctx* c, const double* K_left, const double* K_right, const double* R_right, const double* t_right)
try {
    if (!c || !K_left || !K_right || !R_right || !t_right) return UVO_INVALID_ARG;
    memcpy(c->K_left, K_left, sizeof(double) * 9); memcpy(c->K_right, K_right, sizeof(double) * 9);
    memcpy(c->R_right, R_right, sizeof(double) * 9); memcpy(c->t_right, t_right, sizeof(double) * 3);
    const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
    projection_matrix(I, z, K_left, c->P_eye_left);               // VO:460
    projection_matrix(R_right, t_right, K_right, c->P_right);     // VO:462
    c->rig_set = true;
    return uvo_stereo_reset(c);
} UVO_ABI_CATCH(c)

// collect and drop every entry in flight, stereo pairs and mono frames alike (each collect dequeues its entry even when the pair failed)
static void drain_in_flight(uvo_ctx* c)
{
    while (c->n_pending > 0) {
        const int e = c->inflight[0];
        const bool mono = e == Ctx::kInflightMonoInit || (e >= 0 && c->lanes[e]->job.kind == 1);
        const int before = c->n_pending;
        if (mono) { uvo_mono_result r; (void)uvo_mono_collect(c, 1.0, &r); }
        else { uvo_stereo_result r; (void)uvo_stereo_collect(c, 1.0, &r); }
        if (c->n_pending == before) break;                 // cannot happen; never spin
    }
}

extern "C" uvo_status uvo_stereo_reset(uvo_ctx* c)
try {
    if (!c) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    drain_in_flight(c);                                    // results of whatever is still in flight are dropped
    for (Ctx* l : c->lanes) {
        if (l->pnp_stream) (void)hipStreamSynchronize(l->pnp_stream);
        UVO_HIP_TRY(c, hipMemsetAsync(l->d_counts, 0, sizeof(int) * CN_TOTAL, l->stream));
        UVO_HIP_TRY(c, hipStreamSynchronize(l->stream));
        l->as_w = 0; l->pending = Ctx::Pending();
    }
    c->vo_initialized = false; c->init_matches.clear(); c->stereo_init_results.clear();
    c->prev_lane = 0; c->prev_buf = 0; c->prev_sync = true; c->next_lane = 0;
    c->n_pending = 0; c->n_submitted = c->n_collected = 0;
    for (int i = 0; i < 3; i++) c->t_prev_curr[i] = c->rvec[i] = c->tvec[i] = 0;
    return UVO_OK;
} UVO_ABI_CATCH(c)

// init phase VO:474-520 (first pairs only; host-assisted because results_match_prev accumulates)
static uvo_status stereo_init_step(uvo_ctx* c, uvo_stereo_result* out)
{
    const uvo_params& p = c->p;
    UVO_TRY(read_counts(c));
    UVO_TRY(check_cand_overflow(c, 2));
    const int nL = c->h_counts[CN_NL], nR = c->h_counts[CN_NR];
    out->n_left = nL; out->n_right = nR;
    c->last_nL = nL; c->last_nR = nR; c->last_M = c->last_T = c->last_G = c->last_ninl = 0;
    c->kp_hint = nL > nR ? nL : nR;
    if (nL >= p.MIN_NUM_FEATURES && nR >= p.MIN_NUM_FEATURES) {                         // VO:489
        UVO_TRY(match_knn2(c, c->det[0].desc, nullptr, nL, c->det[1].desc, nullptr, nR));
        UVO_TRY(match_ratio_compact(c, nullptr, nL, (float)p.LOWE_RATIO_THRESHOLD, c->d_matches[0], c->d_nmatch, c->cap));
        UVO_TRY(read_counts(c));
        int m = c->h_counts[CN_M];
        if (m > c->cap) return fail(c, UVO_CAPACITY, "stereo matches exceed max_kpts");
        size_t old = c->init_matches.size();
        c->init_matches.resize(old + m);                                                  // VOU:538 appends
        if (m) UVO_HIP_TRY(c, hipMemcpyAsync(c->init_matches.data() + old, c->d_matches[0], sizeof(uvo_dmatch) * m, hipMemcpyDeviceToHost, c->stream));
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if ((int)c->init_matches.size() > p.MIN_NUM_FEATURES) c->vo_initialized = true;  // VO:500
    }
    const int total = (int)c->init_matches.size();
    out->n_stereo_matches = total; c->last_M = total;
    if (c->vo_initialized) {
        if (total > c->cap) return fail(c, UVO_CAPACITY, "accumulated init matches exceed max_kpts");
        // VO:508-520 with the reference's bounds checks (stale indices from failed attempts)
        std::vector<int> iL(total), iR(total), vL, vR;
        for (int i = 0; i < total; i++) { iL[i] = c->init_matches[i].queryIdx; iR[i] = c->init_matches[i].trainIdx; }
        for (int i = 0; i < total; i++) { if (iL[i] >= 0 && iL[i] < nL) vL.push_back(iL[i]); if (iR[i] >= 0 && iR[i] < nR) vR.push_back(iR[i]); }
        if ((int)vL.size() != total || (int)vR.size() != total)
            return fail(c, UVO_INVALID_ARG, "stale stereo-init matches index past the current keypoints; the reference's "
                                            "keypoint/descriptor sets would go out of step (OpenCV asserts downstream)");
        const int b = c->as_w;         // lane 0 writes the first "after stereo match" set, synchronously
        c->as_w ^= 1; c->prev_lane = 0; c->prev_buf = b; c->prev_sync = true;
        int* d_idx = c->d_tmp_idx;     // scratch
        UVO_HIP_TRY(c, hipMemcpyAsync(d_idx, iL.data(), sizeof(int) * total, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_desc_idx, dim3((total + 15) / 16), dim3(256), 0, c->stream, c->det[0].desc, nL, d_idx, total, c->d_as_descL[b], c->desc_dim());
        hipLaunchKernelGGL(k_gather_kps_idx, dim3((total + 255) / 256), dim3(256), 0, c->stream, c->det[0].kps, d_idx, total, c->d_as_kpsL[b]);
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        UVO_HIP_TRY(c, hipMemcpyAsync(d_idx, iR.data(), sizeof(int) * total, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_kps_idx, dim3((total + 255) / 256), dim3(256), 0, c->stream, c->det[1].kps, d_idx, total, c->d_as_kpsR[b]);
        UVO_HIP_TRY(c, hipMemcpyAsync(c->d_as_n + b, &total, sizeof(int), hipMemcpyHostToDevice, c->stream));
        UVO_HIP_TRY(c, hipMemcpyAsync(c->d_matches[0], c->init_matches.data(), sizeof(uvo_dmatch) * total, hipMemcpyHostToDevice, c->stream));
        {   // the set's rows, triangulated for the pair that follows (every later pair does this in the tail of its stage A)
            const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
            UVO_TRY(pose_as_triangulate(c, c->stream, b, nullptr, total, c->P_eye_left, c->P_right, I, z, c->R_right, c->t_right, c->K_left, c->K_right));
        }
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        UVO_HIP_TRY(c, hipGetLastError());
    }
    out->initialized = 0; out->valid = 0;
    return UVO_OK;
}

// First use of a pipeline lane costs what every first use costs -- the hardware queues behind its two HIP streams are created at
// their first submission, the per-image-size detector tables are built and uploaded (two allocations, four host syncs) -- and a
// caller that times its first pairs through a deep pipeline pays it once per lane inside that loop.  The synchronous init phase
// of a sequence is where the image size becomes known: every lane is prepared there, once per size.
// A kernel with 128 bytes of private (scratch) memory per lane -- at least what any kernel of the pipeline needs (k_pnp_refit_fast:
// 112, k_pnp_refit: 32, k_match_resolve: 20; hipcc -Rpass-analysis=kernel-resource-usage).  The runtime sets a hardware queue's scratch
// up at the first dispatch that needs it, with the host in the loop: measured as +130 us on a lane's first matcher tail and +220 us
// on its first PnP refit (UVO_TRACE of a 20-pair run whose warm-up had not reached the lane).
__global__ void k_prime(int* sink, int n)
{
    volatile int priv[32];
    for (int i = 0; i < 32; i++) priv[i] = i * n;
    int acc = 0;
    for (int i = 0; i < 32; i++) acc += priv[(i * 7 + n) & 31];
    if (n < 0) *sink = acc;                               // never taken: n >= 0
}
static uvo_status prime_lanes(uvo_ctx* c, int w, int h)
{
    for (Ctx* l : c->lanes) {
        if (l->primed_w == w && l->primed_h == h) continue;
        uvo_status st = surf_prepare(l, w, h);
        if (st != UVO_OK) { if (l != c) c->err = l->err; return st; }
        hipLaunchKernelGGL(k_prime, dim3(1), dim3(64), 0, l->stream, l->d_countsB, 1);
        UVO_HIP_TRY(c, hipStreamSynchronize(l->stream));
        // the PnP stream from the lane's worker thread, which is the thread that will use it: its first launch and its first wait
        // (event + stream) also set up the runtime's per-thread state
        UVO_HIP_TRY(c, hipEventRecord(l->evA[0], l->stream));
        l->t_handover_us = 0;
        { std::lock_guard<std::mutex> lk(l->mu); l->job.kind = 2; l->job.state = 1; l->job_state_a.store(1, std::memory_order_release); }
        l->cv.notify_all();
        { std::unique_lock<std::mutex> lk(l->mu); l->cv.wait(lk, [&] { return l->job.state == 2; }); l->job.state = 0; l->job_state_a.store(0, std::memory_order_release); }
        l->primed_w = w; l->primed_h = h;
    }
    return UVO_OK;
}

// Stage A of one pair (VO:548-632): detect, stereo match, triangular match, triangulation,
// extract_3Dpoints -- all enqueued on the lane's stream without a host sync; the counters are copied to the
// lane's pinned mirror, an event marks completion and the lane's worker thread takes over for stage B.
extern "C" uvo_status uvo_stereo_submit(uvo_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride, int mem)
try {
    if (!c || !left || !right) return UVO_INVALID_ARG;
    if (!c->rig_set) return fail(c, UVO_INVALID_ARG, "uvo_stereo_set_rig has not been called");
    const int depth = (int)c->lanes.size();
    if (c->n_pending >= depth) return fail(c, UVO_INVALID_ARG, "uvo_stereo_submit: the pipeline is full; collect a pair first (uvo_stereo_set_depth)");
    if (c->timing && c->n_pending > 0) return fail(c, UVO_INVALID_ARG, "timing mode measures one pair at a time: collect before submitting");
    (void)hipSetDevice(c->device);
    const uvo_params& p = c->p;
    if (!c->vo_initialized) {                                                               // VO:474-520, on lane 0, synchronous
        // (earlier init pairs may still await their collect: they are complete, only their results are queued)
        uvo_stereo_result res;
        memset(&res, 0, sizeof(res));
        UVO_TRY(wait_for_producer(c, c, mem));
        UVO_TRY(surf_upload(c, 0, left, w, h, stride, mem));
        UVO_TRY(surf_upload(c, 1, right, w, h, stride, mem));
        UVO_TRY(detect_dispatch(c, 2));
        UVO_TRY(stereo_init_step(c, &res));
        UVO_TRY(prime_lanes(c, w, h));
        if (c->use_sift()) for (Ctx* l : c->lanes) { uvo_status st_ = sift_prepare_lane(l, w, h, 2); if (st_ != UVO_OK) { if (l != c) c->err = l->err; return st_; } }
        c->stereo_init_results.push_back(res);
        c->inflight[c->n_pending++] = Ctx::kInflightStereoInit; c->n_submitted++;
        c->next_lane = 0;
        return UVO_OK;
    }
    // ---- plan: lane, buffers, what the pair reads from its predecessor; the sequential state moves on here, in submit order ----
    const int li = c->next_lane;
    Range r_submit("uvo:stereo_submit");
    uvo_ctx* L = static_cast<uvo_ctx*>(c->lanes[li]);
    L->pending = Ctx::Pending();
    L->pending.used = true;
    Ctx::StagePlan& pl = L->plan;
    pl.lane = li; pl.prev_lane = c->prev_lane; pl.prev_buf = c->prev_buf; pl.prev_sync = c->prev_sync; pl.curr = L->as_w;
    pl.pending_before = c->n_pending; pl.trace_slot = -1;
    if (L->trace_on) {
        L->trace_cur = (int)(L->trace_count++ % Ctx::kTraceRing);
        pl.trace_slot = L->trace_cur;
        Ctx::TraceRec* tr = &L->trace[L->trace_cur];
        tr->pair = c->n_submitted; tr->b_used = false; tr->det_marked = false;
        for (double& v : tr->host_us) v = 0;
        tr->host_us[0] = uvo::now_us();
    }
    // a pair joins a two-pair launch when the mode is on and the pair can take that path: upright SURF, not the synchronous step,
    // no per-stage timing, two lanes at least
    const bool may_batch = c->batch == 2 && depth >= 2 && !c->in_sync_step && !c->timing && !c->use_sift() && p.SURF_UPRIGHT && p.SURF_OCTAVES_NUMBER == 4;
    uvo_ctx* S = c->stashed_lane >= 0 ? static_cast<uvo_ctx*>(c->lanes[c->stashed_lane]) : nullptr;      // the pair before, waiting for this one
    // uploads: a pair's images go to its own lane's buffers (or are read in place); the copies of a two-pair launch are ordered on the
    // stream that will run the kernels -- the first lane's
    hipStream_t own = L->stream;
    if (S && may_batch) L->stream = S->stream;
    uvo_status up = wait_for_producer(c, L, mem);
    if (up == UVO_OK) up = surf_upload(L, 0, left, w, h, stride, mem);
    if (up == UVO_OK) up = surf_upload(L, 1, right, w, h, stride, mem);
    L->stream = own;
    if (up != UVO_OK) { if (L != c) c->err = L->err; L->pending.used = false; return up; }
    // state carry VO:727-733: this pair's set is the next pair's "prev"
    L->as_w = pl.curr ^ 1;
    c->prev_lane = li; c->prev_buf = pl.curr; c->prev_sync = false;
    c->next_lane = (li + 1) % depth;
    c->inflight[c->n_pending++] = li; c->n_submitted++;
    L->inline_b = c->in_sync_step;
    L->job.kind = 0;                                           // a stereo pair (the lane is free: its worker holds no job)
    if (S && (!may_batch || S->img_w != L->img_w || S->img_h != L->img_h)) { UVO_TRY(queue_stage_a(c, S, nullptr)); S = nullptr; }   // cannot pair up: the waiting pair goes alone
    if (S) return queue_stage_a(c, S, L);
    if (may_batch) { c->stashed_lane = li; return UVO_OK; }           // waits for its partner (or for the collect that needs it)
    return queue_stage_a(c, L, nullptr);
} UVO_ABI_CATCH(c)

// Stage A of one pair (VO:548-632) -- or of two consecutive pairs, lanes A and B, in one set of launches on A's stream: detect,
// stereo match, triangular match, gathers, triangulation, extract_3Dpoints -- queued without a host sync; the counters land in each
// lane's pinned mirror, events mark completion and each lane's worker thread takes over for stage B.
static uvo_status queue_stage_a(uvo_ctx* c, uvo_ctx* A, uvo_ctx* B)
{
    const uvo_params& p = c->p;
    const int depth = (int)c->lanes.size();
    const double t_sub = uvo::g_bdbg ? uvo::now_us() : 0;
    if (c->stashed_lane == A->plan.lane) c->stashed_lane = -1;
    const Ctx::StagePlan& pa = A->plan;
    Ctx* P = c->lanes[pa.prev_lane];
    hipStream_t st = A->stream;
    Ctx::TraceRec* trA = pa.trace_slot >= 0 ? &A->trace[pa.trace_slot] : nullptr;
    Ctx::TraceRec* trB = (B && B->plan.trace_slot >= 0) ? &B->trace[B->plan.trace_slot] : nullptr;
#define LANE_TRY(expr) do { uvo_status st_ = (expr); if (st_ != UVO_OK) { if (A != c) c->err = A->err; return st_; } } while (0)
    if (trA) UVO_HIP_TRY(c, hipEventRecord(trA->ev[0], st));
    if (trB) UVO_HIP_TRY(c, hipEventRecord(trB->ev[0], st));
    // Stage A is a run of chip-filling detection kernels followed by thin ones.  Two stage As side by side fill each other's
    // gaps; three or more interleave at kernel granularity, evict each other's LDS-sized workgroups and every one of them
    // slows down (measured with UVO_TRACE: detection 390 us with two, 1050 us with five lanes in stage A).
    // So this pair's kernels wait for the end of the stage A submitted a_overlap pairs ago; the lanes beyond that hold pairs
    // in their PnP stage.
    double t_seg = uvo::g_bdbg ? uvo::now_us() : 0;
    auto seg = [&](int slot) { if (uvo::g_bdbg) { const double t = uvo::now_us(); uvo::g_bstat[slot] += t - t_seg; t_seg = t; } };
    seg(8);                                                                                // upload (+ producer wait)
    // Pacing.  At most a_overlap stage As run side by side, and it is the submitting thread that waits (hipEventSynchronize spins) for
    // the stage A submitted a_overlap pairs ago before it queues this pair's kernels.  A device-side wait instead (hipStreamWaitEvent on
    // an event that is still pending, so that the kernels sit behind a barrier packet in their hardware queue) was measured at 1200-2200
    // pairs/s against 3800: a queue parked on a barrier slows the other queues down.  evA[1] is the twin of the event the lane's worker
    // blocks on -- the runtime holds an event's lock while a thread waits on it.
    // (Queueing the light integral kernels ahead of the wait was tried: no gain.)  A two-pair launch set counts as one stage A and
    // waits for the set a_overlap2 sets before it.
    {
        const int back = B ? 2 * c->a_overlap2 : c->a_overlap;
        if (back > 0 && depth > back && pa.pending_before >= back) {
            Ctx* H = c->lanes[(pa.lane + depth - back) % depth];
            (void)uvo::poll_event(H->evA[1]);
        }
    }
    seg(9);                                                                                // the pacing wait
    if (trA) trA->host_us[1] = uvo::now_us();
    if (trB) trB->host_us[1] = uvo::now_us();
    { Range r("uvo:detect_features x2");                                                   // VO:548-549, and the VO:556 gate
      if (B) LANE_TRY(surf_detect_lanes(A, B, 2, p.MIN_NUM_FEATURES)); else LANE_TRY(detect_dispatch(A, 2, p.MIN_NUM_FEATURES)); }
    seg(10);                                                                               // detector launches
    if (trA) UVO_HIP_TRY(c, hipEventRecord(trA->ev[1], st));
    if (trB) UVO_HIP_TRY(c, hipEventRecord(trB->ev[1], st));
    const int cap = c->cap, curr = pa.curr, prev = pa.prev_buf;
    int* cn = A->d_counts;
    const float ratio = (float)p.LOWE_RATIO_THRESHOLD;
    // Stereo matching L -> R (VO:558, gated on the device by VO:556) and triangular matching prev-left-after-stereo -> curr-left
    // (VO:592) share their launches: both only need this pair's descriptors and the previous pair's "after stereo match" set
    // (another lane's buffers, behind its event).  The triangular match is computed for every row of that set; whether it is used
    // is VO:567's decision, taken by the first compaction's gate and read by the second's (cn[CN_NQB]).  In a two-pair launch the
    // second pair's previous set is the first pair's, which does not exist yet: match_two_pairs goes through the first pair's stereo matches.
    { Range r_ms("uvo:match_features stereo + triangular, select");
    if (!pa.prev_sync && P != A) UVO_HIP_TRY(c, hipStreamWaitEvent(st, P->evAS, 0));
    GatherPair gp;
    gp.g[0] = GatherArgs{ A->d_matches[0], A->d_matches[1], cn, A->det[0].kps, A->det[1].kps, A->det[0].desc,
                          A->d_as_kpsL[curr], A->d_as_kpsR[curr], A->d_as_descL[curr], A->desc_dim(), P->d_as_kpsL[prev], P->d_as_kpsR[prev],
                          A->d_x1, A->d_x2, A->d_xc, (cap + 15) / 16, nullptr };
    if (!B) {
        LANE_TRY(match_knn2_two(A, A->det[0].desc, cn + CN_NQA, A->det[1].desc, cn + CN_NR,
                                P->d_as_descL[prev], P->d_as_n + prev, A->det[0].desc, cn + CN_NL, cap));
        const GateArgs gate_b = { 1, cn, p.MIN_NUM_FEATURES, cap, A->d_as_n + curr, P->d_as_n + prev };       // VO:567
        const GateArgs gate_c = { 2, cn, p.MIN_NUM_FEATURES, cap, nullptr, nullptr };                           // VO:626
        LANE_TRY(match_ratio_compact2(A, ratio, cn + CN_NQA, A->d_matches[0], cn + CN_M, gate_b,
                                      cn + CN_NQB, A->d_matches[1], cn + CN_TRAW, gate_c, cap, cap));
        gp.g[1] = gp.g[0];
    } else {
        const int currB = B->plan.curr;
        LANE_TRY(match_two_pairs(A, B, P->d_as_descL[prev], P->d_as_n + prev, P->d_as_n + prev, curr, currB, ratio, p.MIN_NUM_FEATURES));
        // the second pair's triangular points: rows of the first pair's set, through its stereo matches, from its keypoint lists
        gp.g[1] = GatherArgs{ B->d_matches[0], B->d_matches[1], B->d_counts, B->det[0].kps, B->det[1].kps, B->det[0].desc,
                              B->d_as_kpsL[currB], B->d_as_kpsR[currB], B->d_as_descL[currB], B->desc_dim(), A->det[0].kps, A->det[1].kps,
                              B->d_x1, B->d_x2, B->d_xc, (cap + 15) / 16, A->d_matches[0] };
    }
    const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
    if (B) {
        {
            StageTimer t(A, ST_GATHER);
            hipLaunchKernelGGL(k_gather_stereo_step, dim3((cap + 15) / 16 + (cap + 255) / 256, 2), dim3(256), 0, st, gp);
        }
        UVO_HIP_TRY(c, hipGetLastError());
        // triangulation + extract_3Dpoints (VO:631-632) of both pairs on the point pairs just gathered; the rows of the second pair's set
        // for a pair that follows alone (the first pair's set is read by the second only)
        Range r_tri("uvo:triangulatePoints + extract_3Dpoints");
        LANE_TRY(pose_triangulate_extract3d(A, 0, c->P_eye_left, c->P_right, I, z, c->R_right, c->t_right, c->K_left, c->K_right, cn + CN_T, cap,
                                            A->h_countsA[0], B, B->h_countsA[0]));              // the counters land in pinned memory, no copy queued
        LANE_TRY(pose_as_triangulate(B, st, B->plan.curr, B->d_as_n + B->plan.curr, cap, c->P_eye_left, c->P_right, I, z, c->R_right, c->t_right, c->K_left, c->K_right));
        UVO_HIP_TRY(c, hipEventRecord(A->evAS, st));
        UVO_HIP_TRY(c, hipEventRecord(B->evAS, st));
    } else {
        // one launch: this pair's set (VO:569-579) and its rows triangulated for the next pair; extract_3Dpoints (VO:632) on the rows of
        // the previous pair's set that the triangular matches select (VO:631's points, triangulated in that pair's tail)
        Range r_tri("uvo:select + triangulatePoints + extract_3Dpoints");
        LANE_TRY(pose_stereo_tail(A, P, prev, curr, 0, c->P_eye_left, c->P_right, I, z, c->R_right, c->t_right, c->K_left, c->K_right, A->h_countsA[0]));
        UVO_HIP_TRY(c, hipEventRecord(A->evAS, st));
    }
    }
    UVO_HIP_TRY(c, hipEventRecord(A->evA[1], st));                                          // end of stage A, for the pacing of later pairs (see uvo_ctx.h)
    if (B) UVO_HIP_TRY(c, hipEventRecord(B->evA[1], st));
    if (trA) UVO_HIP_TRY(c, hipEventRecord(trA->ev[2], st));
    if (trB) UVO_HIP_TRY(c, hipEventRecord(trB->ev[2], st));
    // The first RANSAC round without the host (pose.hip: k_pnp_*_spec), in two places:
    //  * the synchronous step (one pair in flight) queues it on this stream, right behind extract_3Dpoints, and the calling thread
    //    confirms it (0.80 -> 0.77 ms per pair at C3);
    //  * a pipelined pair in DEVICE-DRIVEN mode (uvo_ctx.h: stage_b_mode) queues it on the lane's PnP stream behind stage A's event:
    //    nobody is handed the pair, uvo_stereo_collect confirms the round (or redoes the stage itself).  Queued on the lane's
    //    stage-A stream instead, the same kernels cost the pipeline a quarter of its rate (round 3: 4370 -> 3200 pairs/s).
    // UVO_PNP_SPEC=0 switches the round off (the host-driven stage then runs on the calling / collecting thread).
    static const int spec_env = getenv("UVO_PNP_SPEC") ? atoi(getenv("UVO_PNP_SPEC")) : 1;
    const bool dev_b = !B && !A->inline_b && !c->timing && stage_b_on_device(c);
    A->dev_b = dev_b;
    if (B) B->dev_b = false;
    A->spec_queued = !B && spec_env != 0 && (A->inline_b || dev_b) && !c->timing && p.ITERATIONS_COUNT >= 1;
    if (B) B->spec_queued = false;
    hipStream_t sb = st;
    if (dev_b) { sb = A->pnp_stream; UVO_HIP_TRY(c, hipStreamWaitEvent(sb, A->evA[1], 0)); }
    if (A->spec_queued) LANE_TRY(pose_pnp_spec_launch(A, sb, c->K_left, p.ITERATIONS_COUNT, (float)p.REPROJECTION_ERROR_THRESHOLD, p.CONFIDENCE, p.MIN_NUM_3DPOINTS));
    if (dev_b) UVO_HIP_TRY(c, hipEventRecord(A->evB, sb));                                  // what uvo_stereo_collect waits for
    else {
        UVO_HIP_TRY(c, hipEventRecord(A->evA[0], sb));                                      // what the lane's worker waits for
        if (B) UVO_HIP_TRY(c, hipEventRecord(B->evA[0], sb));
    }
    if (trA) trA->host_us[2] = uvo::now_us();
    if (trB) trB->host_us[2] = uvo::now_us();
    uvo_ctx* both[2] = { A, B };
    for (uvo_ctx* L : both) {
        if (!L) continue;
        if (L->inline_b) {
            // the synchronous step waits for its own pair: the calling thread polls the stream's end itself and finishes stage B inline
            // (no worker wake-up, no condition variable: two thread hand-overs less on the pair's critical path)
            UVO_HIP_TRY(c, hipEventRecord(L->evSync, sb));
            L->job.kind = 0;
        } else if (L->dev_b) {
            L->job.kind = 0;                                   // nobody is woken: the round is on the device, collect finishes the pair
        } else {   // hand stage B to the lane's worker
            L->t_handover_us = uvo::now_us();
            { std::lock_guard<std::mutex> lk(L->mu); L->job.kind = 0; L->job.state = 1; L->job_state_a.store(1, std::memory_order_release); }
            L->cv.notify_all();
        }
    }
    seg(11);                                                                               // matcher .. extract_3Dpoints launches, hand-over
    if (uvo::g_bdbg) { uvo::g_bstat[6] += uvo::now_us() - t_sub; uvo::g_bstat[7] += B ? 2 : 1; }
    return UVO_OK;
#undef LANE_TRY
}

// Stage B of one lane's pair (VO:634-648), on the lane's worker thread: wait for stage A, then solvePnPRansac.
namespace uvo { extern bool g_bdbg; extern std::atomic<double> g_bstat[16]; double now_us(); void operator+=(std::atomic<double>& a, double v); }
static void run_stage_b(uvo_ctx* L, bool stage_a_ok)
{
    Ctx::BJob& j = L->job;
    const Ctx* m = L->master ? L->master : L;
    j.st = UVO_OK; j.err.clear(); j.ran = j.ninl = j.ok = j.wrote = 0;
    if (!stage_a_ok) { j.st = UVO_HIP_ERROR; j.err = "stage A of the pair failed"; return; }
    const int* hc = L->h_countsA[0];
    const int cap = L->cap;
    if (hc[CN_CAND0] > cap || hc[CN_CAND1] > cap || hc[CN_M] > cap || hc[CN_TRAW] > cap) return;    // reported by collect
    const int G = hc[CN_G];
    const uvo_params& p = L->p;
    if (G > p.MIN_NUM_3DPOINTS) {                                                          // VO:634
        j.ran = 1;
        Range rg("uvo:solvePnPRansac");
        PnpResult r;
        Ctx* one[1] = { L };
        if (L->spec_queued && pose_pnp_spec_accept(L, G, p.ITERATIONS_COUNT, p.CONFIDENCE, &r)) j.st = UVO_OK;      // the device's round, confirmed by the host's scan
        else
        j.st = pose_pnp_ransac_batch(L, 1, one, &G, m->K_left, p.ITERATIONS_COUNT, (float)p.REPROJECTION_ERROR_THRESHOLD, p.CONFIDENCE, &r);   // VO:647-648
        if (j.st == UVO_OK) j.st = r.st;
        if (j.st != UVO_OK) j.err = L->err;
        else { j.wrote = r.wrote; j.ok = r.ok; j.ninl = r.ninl; memcpy(j.rvec, r.rvec, sizeof(j.rvec)); memcpy(j.tvec, r.tvec, sizeof(j.tvec)); }
    }
}
// Does this lane's pipeline poll rather than sleep (uvo_ctx.h: worker_wait; decided per depth by apply_wait_policy)?
static bool pipeline_polls(const Ctx* L) { return L->wait_eff.load(std::memory_order_acquire) == 0; }
// A thread about to sleep on a lane's condition variable for job.state == want first polls the state's atomic twin for up to `spin_us`:
// in a running pipeline the hand-overs (submitter -> worker, worker -> collect) come within a few hundred microseconds, and a
// thread asleep on a futex was seen to be woken milliseconds late on a loaded host (the last of round 4's stalls: 4 ms holes with
// every device-side wait already polling).  An idle context's threads still end up asleep.
static void poll_job_state(uvo_ctx* L, int want, double spin_us)
{
    if (!pipeline_polls(L)) return;
    const double t0 = now_us();
    while (L->job_state_a.load(std::memory_order_acquire) != want) {          // (a context being destroyed: the spin times out and the condition variable sees `quit`)
        for (int i = 0; i < 40; i++) __builtin_ia32_pause();
        if (now_us() - t0 > spin_us) return;
    }
}
// The lane worker's long wait, for the end of its pair's stage A (uvo_ctx.h: worker_wait).
static bool wait_stage_a(uvo_ctx* L)
{
    const int eff = L->wait_eff.load(std::memory_order_acquire);
    if (eff == 2) return hipEventSynchronize(L->evA[0]) == hipSuccess;       // sleeps on the interrupt (hipEventBlockingSync)
    if (eff == 0) return uvo::poll_event(L->evA[0]) == hipSuccess;           // spin, or auto with a CPU per worker (+ submitter + one spare)
    const double t0 = L->t_handover_us, mean = L->stage_a_mean_us;
    for (;;) {
        const hipError_t e = hipEventQuery(L->evA[0]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) return false;
        const double left = t0 + 0.8 * mean - now_us();                 // sleep through the first four fifths of an average stage A ..
        if (left > 60.0) std::this_thread::sleep_for(std::chrono::microseconds((long long)(left < 400.0 ? left - 40.0 : 360.0)));
        else for (int i = 0; i < 40; i++) __builtin_ia32_pause();        // .. and poll over the rest (a query every microsecond or two)
    }
    if (t0 > 0) {
        const double d = now_us() - t0;
        L->stage_a_mean_us = mean == 0.0 ? d : mean + 0.125 * (d - mean);      // the lane worker's own figure: no other thread touches it
    }
    return true;
}
static void lane_worker(uvo_ctx* L)
{
    { char nm[16]; snprintf(nm, sizeof(nm), "uvo-lane%d", L->lane_id); (void)pthread_setname_np(pthread_self(), nm); }      // (top, /proc/<pid>/task/*/comm: tools/probe/thread_cpu.py)
    (void)hipSetDevice(L->device);
    std::unique_lock<std::mutex> lk(L->mu);
    for (;;) {
        if (L->job.state != 1 && !L->quit) { lk.unlock(); poll_job_state(L, 1, 2000.0); lk.lock(); }     // the next pair usually comes within the pipeline's period
        L->cv.wait(lk, [&] { return L->quit || L->job.state == 1; });
        if (L->quit) return;
        lk.unlock();
        // stage A's end is awaited before a PnP slot is taken, so a slot is never held idle
        const double t0 = g_bdbg ? now_us() : 0;
        const bool stage_a_ok = wait_stage_a(L);
        if (g_bdbg) g_bstat[0] += now_us() - t0;
        Ctx::TraceRec* wtr = (L->trace_on && L->trace_cur >= 0 && L->job.kind == 0) ? &L->trace[L->trace_cur] : nullptr;
        if (wtr) wtr->host_us[3] = now_us();
        {   // at most max_b PnP stages at a time over all lanes: their thin, latency-bound kernels slow down and are slowed by stage A's
            Ctx* m = L->master ? L->master : L;
            std::unique_lock<std::mutex> g(m->b_mu);
            const double tw = g_bdbg ? now_us() : 0;
            const int limit = L->job.kind == 1 ? m->max_b_mono : m->max_b;                  // the mono pose stage is far longer and thinner (see uvo_ctx.h)
            m->b_cv.wait(g, [&] { return m->b_running < limit; });
            if (g_bdbg) g_bstat[5] += now_us() - tw;
            m->b_running++;
            g.unlock();
            if (wtr) wtr->host_us[4] = now_us();
            try {
                if (L->job.kind == 1) { const double tb = g_bdbg ? now_us() : 0; run_mono_stage_b(L, stage_a_ok); if (g_bdbg) { g_bstat[1] += now_us() - tb; g_bstat[4] += 1; } }
                else if (L->job.kind == 2) { hipLaunchKernelGGL(k_prime, dim3(1), dim3(64), 0, L->pnp_stream, L->d_countsB, 1); (void)host_sync(L, L->pnp_stream); }   // prime_lanes
                else run_stage_b(L, stage_a_ok);
            } catch (...) {                                                // a worker has no caller to unwind into: the pair fails with a status, its collect reports it
                L->job.st = abi_caught(L); L->job.ran = 0;
                try { L->job.err = L->err; } catch (...) { L->job.err.clear(); }
            }
            if (wtr) wtr->host_us[5] = now_us();
            g.lock();
            m->b_running--;
            m->b_cv.notify_one();
        }
        lk.lock();
        L->job.state = 2;
        L->job_state_a.store(2, std::memory_order_release);
        L->cv.notify_all();
    }
}

// Result of the oldest submitted pair: gates, pose inversion and output (VO:634-717, VO:148-159), in order.
extern "C" uvo_status uvo_stereo_collect(uvo_ctx* c, double dt, uvo_stereo_result* out)
try {
    if (!c || !out) return UVO_INVALID_ARG;
    if (c->n_pending <= 0) return fail(c, UVO_INVALID_ARG, "uvo_stereo_collect: nothing submitted");
    if (c->inflight[0] == Ctx::kInflightMonoInit || (c->inflight[0] >= 0 && c->lanes[c->inflight[0]]->job.kind != 0))
        return fail(c, UVO_INVALID_ARG, "uvo_stereo_collect: the oldest entry in flight is a mono frame (uvo_mono_collect)");
    (void)hipSetDevice(c->device);
    Range r_collect("uvo:stereo_collect");
    const uvo_params& p = c->p;
    const int li = c->inflight[0];
    for (int i = 1; i < c->n_pending; i++) c->inflight[i - 1] = c->inflight[i];
    c->n_pending--; c->n_collected++;
    if (li == Ctx::kInflightStereoInit) {                       // a synchronous init pair (its state is already applied)
        *out = c->stereo_init_results.front();
        c->stereo_init_results.pop_front();
        c->last_lane = 0;
        return UVO_OK;
    }
    uvo_ctx* L = static_cast<uvo_ctx*>(c->lanes[li]);
    c->last_lane = li;
    L->pending.used = false;
    if (c->stashed_lane == li) {                               // the pair was waiting for a partner that has not come: it goes alone, now
        const uvo_status qs = queue_stage_a(c, L, nullptr);
        if (qs != UVO_OK) return qs;
    }
    if (L->inline_b) {
        L->inline_b = false;
        run_stage_b(L, uvo::poll_event(L->evSync) == hipSuccess);
    } else if (L->dev_b) {
        // device-driven pair: wait for its round on the lane's PnP stream (polling; on the interrupt in block-all mode), confirm it
        // with the host's own scan, or run the host-driven stage here when the round could not decide the pair
        L->dev_b = false;
        Ctx::TraceRec* wtr = (L->trace_on && L->plan.trace_slot >= 0) ? &L->trace[L->plan.trace_slot] : nullptr;
        if (wtr) wtr->host_us[3] = uvo::now_us();
        const bool ok = (L->wait_eff.load(std::memory_order_acquire) == 2 ? hipEventSynchronize(L->evB) : uvo::poll_event(L->evB)) == hipSuccess;
        if (wtr) wtr->host_us[4] = uvo::now_us();
        run_stage_b(L, ok);
        if (wtr) wtr->host_us[5] = uvo::now_us();
    } else {
        poll_job_state(L, 2, 5000.0);
        std::unique_lock<std::mutex> lk(L->mu);
        L->cv.wait(lk, [&] { return L->job.state == 2; });
        L->job.state = 0; L->job_state_a.store(0, std::memory_order_release);
    }
    const Ctx::BJob& j = L->job;
    memset(out, 0, sizeof(*out));
    out->initialized = 1;
    if (j.st != UVO_OK && !j.ran) return fail(c, j.st, j.err.c_str());
    const int* hc = L->h_countsA[0];
    memcpy(L->h_counts, hc, sizeof(int) * CN_TOTAL);
    if (check_cand_overflow(L, 2) != UVO_OK) return fail(c, UVO_CAPACITY, L->err.c_str());
    const int cap = c->cap;
    if (hc[CN_M] > cap || hc[CN_TRAW] > cap) return fail(c, UVO_CAPACITY, "match count exceeds max_kpts");
    const int nL = hc[CN_NL], nR = hc[CN_NR], M = hc[CN_M], T = hc[CN_TRAW], G = hc[CN_G];
    out->n_left = nL; out->n_right = nR; out->n_stereo_matches = M; out->n_tri_matches = T; out->n_good3d = G;
    L->last_nL = nL; L->last_nR = nR; L->last_M = M; L->last_T = hc[CN_T]; L->last_G = G; L->last_ninl = 0;
    c->kp_hint = nL > nR ? nL : nR;
    int valid = 0;
    if (j.ran) {                                                                           // VO:634
        if (j.st != UVO_OK) return fail(c, j.st, j.err.c_str());
        if (j.wrote) { memcpy(c->rvec, j.rvec, sizeof(c->rvec)); memcpy(c->tvec, j.tvec, sizeof(c->tvec)); }     // also the last hypothesis of a failed RANSAC, as OpenCV
        L->last_ninl = j.ninl; out->n_inliers = j.ninl;
        if (j.ninl >= p.MIN_NUM_INLIERS) {                                                 // VO:665
            double R[9];
            rodrigues_vec2mat(c->rvec, R);                                                 // VO:673
            for (int i = 0; i < 3; i++) {                                                  // VO:675: -R^T t
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += R[k*3 + i] * c->tvec[k];
                c->t_prev_curr[i] = acc * -1.0;
            }
            valid = 1;
        }
    }
    out->valid = valid;
    for (int i = 0; i < 3; i++) {
        out->rvec[i] = c->rvec[i]; out->tvec[i] = c->tvec[i]; out->t_prev_curr[i] = c->t_prev_curr[i];
        out->velocity[i] = c->t_prev_curr[i] / dt;                                         // VO:152
    }
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_stereo_step(uvo_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                                      int mem, double dt, uvo_stereo_result* out)
try {
    if (!c || !out) return UVO_INVALID_ARG;
    if (c->n_pending != 0) return fail(c, UVO_INVALID_ARG, "uvo_stereo_step: pairs submitted with uvo_stereo_submit are still in flight");
    c->in_sync_step = true;
    const uvo_status st = uvo_stereo_submit(c, left, right, w, h, stride, mem);
    c->in_sync_step = false;
    UVO_TRY(st);
    return uvo_stereo_collect(c, dt, out);
} UVO_ABI_CATCH(c)

extern "C" int uvo_stereo_get(uvo_ctx* m, const char* what, void* out, int cap_bytes)
try {
    if (!m || !what || !out) return 0;
    (void)hipSetDevice(m->device);
    const Ctx* c = m->lanes.empty() ? m : m->lanes[m->last_lane];    // the lane of the last collected pair
    const void* src = nullptr; int count = 0; size_t esz = 0;
    std::string w(what);
    if (w == "kps_left") { src = c->det[0].kps; count = c->last_nL; esz = sizeof(uvo_keypoint); }
    else if (w == "kps_right") { src = c->det[1].kps; count = c->last_nR; esz = sizeof(uvo_keypoint); }
    else if (w == "desc_left") { src = c->det[0].desc; count = c->last_nL; esz = c->desc_dim() * sizeof(float); }
    else if (w == "desc_right") { src = c->det[1].desc; count = c->last_nR; esz = c->desc_dim() * sizeof(float); }
    else if (w == "matches_stereo") { src = c->d_matches[0]; count = c->last_M; esz = sizeof(uvo_dmatch); }
    else if (w == "matches_tri") { src = c->d_matches[1]; count = c->h_counts[CN_TRAW]; esz = sizeof(uvo_dmatch); }
    else if (w == "points4d") { src = c->d_pts4; count = c->last_T; esz = sizeof(float4); }
    else if (w == "good_pts") { src = c->d_good_pts[0]; count = c->last_G; esz = 3 * sizeof(double); }
    else if (w == "good_idx") { src = c->d_good_idx[0]; count = c->last_G; esz = sizeof(int); }
    else if (w == "inliers") { src = c->d_inliers; count = c->last_ninl; esz = sizeof(int); }
    else return 0;
    if ((size_t)count * esz > (size_t)cap_bytes) return -count;
    if (count) {
        if (hipMemcpy(out, src, (size_t)count * esz, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    }
    return count;
} UVO_ABI_CATCH_RET(m, 0)


// ------------------------------------------------------------------------------------------ get_image
// cvUndistortPointsInternal for one point, no R, no P (normalised output): five fixed-point iterations of the inverse of the
// (k1, k2, p1, p2) model
static void undistort_point_normalised(double u, double v, const double* K, const double* d4, double* xo, double* yo)
{
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], ifx = 1. / fx, ify = 1. / fy;
    const double k1 = d4[0], k2 = d4[1], p1 = d4[2], p2 = d4[3];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((0 * r2 + k2) * r2 + k1) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *xo = x; *yo = y;
}

// VO_utility.cpp:658-675.  getOptimalNewCameraMatrix(alpha = 0, no validPixROI, centerPrincipalPoint = false): a 9 x 9 grid of
// image points is undistorted to normalised coordinates, the rectangle inscribed in the grid's border is mapped onto the
// viewport (calibration.cpp cvGetOptimalNewCameraMatrix / icvGetRectangles).  Host arithmetic, once per run.
extern "C" uvo_status uvo_resize_camera_matrix(int original_width, int original_height, int desired_width, double* K, const double* dist4,
                                               double* newK, int* desired_height_out)
try {
    if (!K || !dist4 || !newK || desired_width <= 0 || original_width <= 0 || original_height <= 0) return UVO_INVALID_ARG;
    const double ratio = (double)original_width / (double)desired_width;
    const int desired_height = (int)(original_height / ratio);
    if (desired_height_out) *desired_height_out = desired_height;
    const double skew = K[1];
    for (int i = 0; i < 9; i++) K[i] = K[i] / ratio;
    K[1] = skew; K[8] = 1;
    const int N = 9;
    double iX0 = -DBL_MAX, iX1 = DBL_MAX, iY0 = -DBL_MAX, iY1 = DBL_MAX;
    double oX0 = DBL_MAX, oX1 = -DBL_MAX, oY0 = DBL_MAX, oY1 = -DBL_MAX;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            double px, py;
            undistort_point_normalised((double)x * (desired_width - 1) / (N - 1), (double)y * (desired_height - 1) / (N - 1), K, dist4, &px, &py);
            oX0 = std::min(oX0, px); oX1 = std::max(oX1, px); oY0 = std::min(oY0, py); oY1 = std::max(oY1, py);
            if (x == 0) iX0 = std::max(iX0, px);
            if (x == N - 1) iX1 = std::min(iX1, px);
            if (y == 0) iY0 = std::max(iY0, py);
            if (y == N - 1) iY1 = std::min(iY1, py);
        }
    const double alpha = 0;
    const double fx0 = (desired_width - 1) / (iX1 - iX0), fy0 = (desired_height - 1) / (iY1 - iY0);
    const double cx0 = -fx0 * iX0, cy0 = -fy0 * iY0;
    const double fx1 = (desired_width - 1) / (oX1 - oX0), fy1 = (desired_height - 1) / (oY1 - oY0);
    const double cx1 = -fx1 * oX0, cy1 = -fy1 * oY0;
    for (int i = 0; i < 9; i++) newK[i] = K[i];
    newK[0] = fx0 * (1 - alpha) + fx1 * alpha;
    newK[4] = fy0 * (1 - alpha) + fy1 * alpha;
    newK[2] = cx0 * (1 - alpha) + cx1 * alpha;
    newK[5] = cy0 * (1 - alpha) + cy1 * alpha;
    return UVO_OK;
} UVO_ABI_CATCH(nullptr)

extern "C" uvo_status uvo_get_image(uvo_ctx* c, const uint8_t* rgb, int w, int h, int stride, int mem, const double* K, const double* dist4,
                                    const double* newK, int desired_width, int clahe, int clip_limit, uint8_t* out, int out_mem,
                                    int* out_w, int* out_h)
try {
    if (!c || !rgb || !K || !dist4 || !newK || !out || !out_w || !out_h) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    const uint8_t* d_res = nullptr;
    UVO_TRY(pre_get_image(c, rgb, w, h, stride, mem, K, dist4, newK, desired_width, clahe, clip_limit, &d_res, out_w, out_h));
    const size_t n = (size_t)*out_w * *out_h;
    UVO_HIP_TRY(c, hipMemcpyAsync(out, d_res, n, out_mem == UVO_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
} UVO_ABI_CATCH(c)

// ------------------------------------------------------------------------------------------ compressed-image ingest
extern "C" uvo_status uvo_decode_image(uvo_ctx* c, const uint8_t* data, size_t n, const char* format, uint8_t* out, size_t cap_bytes, int out_mem,
                                       int* w, int* h, int* channels)
try {
    if (!c || !data || !w || !h || !channels) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_decode_image"));
    const std::string fmt = format ? format : "";
    const int bayer = fmt.find("bayer") != std::string::npos ? 1 : 0;
    if (!out) return codec_peek(c, data, n, bayer, w, h, channels);      // size query: the headers only, nothing is decoded
    const uint8_t* d_res = nullptr;
    UVO_TRY(codec_decode(c, data, n, bayer, &d_res, w, h, channels));
    const size_t bytes = (size_t)*w * *h * *channels;
    if (bytes > cap_bytes) return fail(c, UVO_CAPACITY, "uvo_decode_image: output capacity too small");
    UVO_HIP_TRY(c, hipMemcpyAsync(out, d_res, bytes, out_mem == UVO_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_bayer_bggr2bgr(uvo_ctx* c, const uint8_t* bayer, int w, int h, int stride, int mem, uint8_t* out_bgr, int out_mem)
try {
    if (!c || !bayer || !out_bgr || w <= 0 || h <= 0 || stride < w) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    UVO_TRY(need_idle(c, "uvo_bayer_bggr2bgr"));
    UVO_TRY(wait_for_producer(c, c, mem));
    const uint8_t* d_res = nullptr;
    UVO_TRY(codec_bayer(c, bayer, w, h, stride, mem, &d_res));
    UVO_HIP_TRY(c, hipMemcpyAsync(out_bgr, d_res, (size_t)w * h * 3, out_mem == UVO_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
} UVO_ABI_CATCH(c)

// ------------------------------------------------------------------------------------------ mono path
extern "C" uvo_status uvo_find_essential_mat(uvo_ctx* c, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K, int method,
                                             double prob, double threshold, int max_iters, double* E, uint8_t* mask, int* ok)
try {
    if (!c || !p1 || !p2 || !K || !E || !mask || !ok || n < 0 || (method != 4 && method != 8)) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    return mono_find_essential(c, p1, p2, n, K, method, prob, threshold, max_iters, E, mask, ok);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_recover_pose(uvo_ctx* c, const double* E, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K,
                                       double* R, double* t, uint8_t* mask, int* good)
try {
    if (!c || !E || !p1 || !p2 || !K || !R || !t || !mask || !good || n < 0) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    return mono_recover_pose(c, E, p1, p2, n, K, R, t, mask, good);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_find_homography(uvo_ctx* c, const uvo_point2f* p1, const uvo_point2f* p2, int n, int method, double threshold,
                                          int max_iters, double confidence, double* H, uint8_t* mask, int* ok)
try {
    if (!c || !p1 || !p2 || !H || !mask || !ok || n < 0 || (method != 4 && method != 8)) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    return mono_find_homography(c, p1, p2, n, method, threshold, max_iters, confidence, H, mask, ok);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_decompose_homography_mat(const double* H, const double* K, double* Rs, double* ts, double* ns, int* n_solutions)
try {
    if (!H || !K || !Rs || !ts || !ns || !n_solutions) return UVO_INVALID_ARG;
    *n_solutions = decompose_homography_mat(H, K, Rs, ts, ns);
    return UVO_OK;
} UVO_ABI_CATCH(nullptr)
extern "C" uvo_status uvo_recover_pose_homography(uvo_ctx* c, const double* H, const uvo_point2f* p1, const uvo_point2f* p2, int n,
                                                  const double* K, double* R, double* t, int* max_good)
try {
    if (!c || !H || !p1 || !p2 || !K || !R || !t || !max_good || n < 0) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    return mono_recover_pose_homography(c, H, p1, p2, n, K, c->p.HOMOGRAPHY_DISTANCE, R, t, max_good);
} UVO_ABI_CATCH(c)

// MU:65-86 compute_median
static double compute_median(std::vector<double> v)
{
    size_t size = v.size();
    if (size == 0) return 0.0;
    std::sort(v.begin(), v.end());
    if (size % 2 == 0) { size_t mid = size / 2; return (v[mid - 1] + v[mid]) / 2.0; }
    return v[size / 2];
}
// select_estimation_method (VOU:725-748): 1 = essential, 0 = homography
static int select_estimation_method(const uvo_point2f* k1, const uvo_point2f* k2, int n, int distance)
{
    std::vector<double> d(n > 0 ? n : 0);
    for (int i = 0; i < n; i++) { double dx = k1[i].x - k2[i].x, dy = k1[i].y - k2[i].y; d[i] = sqrt(dx * dx + dy * dy); }
    return compute_median(d) < distance ? 0 : 1;
}
extern "C" int uvo_select_estimation_method(const uvo_point2f* k1, const uvo_point2f* k2, int n, int distance)
try {
    if (n < 0 || (n && (!k1 || !k2))) return -1;
    return select_estimation_method(k1, k2, n, distance);
} UVO_ABI_CATCH_RET(nullptr, -1)
// extract_inliers (VOU:306-329)
static int extract_inliers(const uvo_point2f* k1, const uvo_point2f* k2, const uint8_t* mask, int n, uvo_point2f* in1, uvo_point2f* in2)
{
    int k = 0;
    for (int i = 0; i < n; i++) if (mask[i] != 0) { in1[k] = k1[i]; in2[k] = k2[i]; k++; }
    return k;
}
// One attempt of estimate_relative_pose's loop (VOU:143-156) on host points: the essential or the homography branch; *valid_inliers =
// countNonZero(mask) after it (VOU:157).  R, t are written when the branch produced a pose.
static uvo_status estimate_once(uvo_ctx* c, const uvo_point2f* k1, const uvo_point2f* k2, int n, const double* K, int use_essential, double* R, double* t,
                                uvo_point2f* in1, uvo_point2f* in2, int* n_in, uint8_t* mask, int* valid_inliers)
{
    const uvo_params& p = c->p;
    int ok = 0;
    memset(mask, 0, n);
    if (use_essential) {
        double E[9];
        UVO_TRY(mono_find_essential(c, k1, k2, n, K, p.ESSENTIAL_OUTLIER_METHOD, p.ESSENTIAL_CONFIDENCE, p.ESSENTIAL_THRESHOLD,
                                    (int)p.ESSENTIAL_MAX_ITERS, E, mask, &ok));
        *n_in = extract_inliers(k1, k2, mask, n, in1, in2);
        int good = 0;
        if (ok) UVO_TRY(mono_recover_pose(c, E, k1, k2, n, K, R, t, mask, &good));
        else memset(mask, 0, n);          // OpenCV would throw on the empty E; the attempt simply fails here
    } else {
        double H[9];
        UVO_TRY(mono_find_homography(c, k1, k2, n, p.HOMOGRAPHY_OUTLIER_METHOD, p.HOMOGRAPHY_THRESHOLD, (int)p.HOMOGRAPHY_MAX_ITERS,
                                     p.HOMOGRAPHY_CONFIDENCE, H, mask, &ok));
        *n_in = extract_inliers(k1, k2, mask, n, in1, in2);
        int good = 0;
        if (ok) UVO_TRY(mono_recover_pose_homography(c, H, k1, k2, n, K, p.HOMOGRAPHY_DISTANCE, R, t, &good));
    }
    int v = 0;
    for (int i = 0; i < n; i++) v += mask[i] != 0;
    *valid_inliers = v;
    return UVO_OK;
}
// the acceptance test of VOU:157-164
static bool estimate_accepted(const uvo_params& p, int valid_inliers, int n)
{
    const double valid_point_fraction = (double)valid_inliers / n;
    return valid_point_fraction >= p.VPF_THRESHOLD && valid_inliers >= p.MIN_NUM_INLIERS;
}
static const char* bad_outlier_methods(const uvo_params& p)
{
    if ((p.ESSENTIAL_OUTLIER_METHOD != 4 && p.ESSENTIAL_OUTLIER_METHOD != 8) || (p.HOMOGRAPHY_OUTLIER_METHOD != 4 && p.HOMOGRAPHY_OUTLIER_METHOD != 8))
        return "outlier methods must be 4 (LMEDS) or 8 (RANSAC)";
    return nullptr;
}
// estimate_relative_pose (VOU:134-180).  use_essential is the reference's global (in/out); R, t are in/out.
extern "C" uvo_status uvo_estimate_relative_pose(uvo_ctx* c, const uvo_point2f* k1, const uvo_point2f* k2, int n, const double* K,
                                                 int* use_essential, double* R, double* t, uvo_point2f* in1, uvo_point2f* in2, int* n_in,
                                                 uint8_t* mask, int* success)
try {
    if (!c || !k1 || !k2 || !K || !use_essential || !R || !t || !in1 || !in2 || !n_in || !mask || !success || n <= 0) return UVO_INVALID_ARG;
    (void)hipSetDevice(c->device);
    Range r_erp("uvo:estimate_relative_pose");
    const uvo_params& p = c->p;
    if (const char* why = bad_outlier_methods(p)) return fail(c, UVO_INVALID_ARG, why);
    bool estimate_completed = false, switch_method = false;
    *success = 0;
    while (!estimate_completed) {
        int valid_inliers = 0;
        UVO_TRY(estimate_once(c, k1, k2, n, K, *use_essential, R, t, in1, in2, n_in, mask, &valid_inliers));
        if (estimate_accepted(p, valid_inliers, n)) { *success = 1; estimate_completed = true; }
        else {
            if (switch_method) break;
            switch_method = true;
            *use_essential = !*use_essential;
        }
    }
    return UVO_OK;
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_mono_set_camera(uvo_ctx* c, const double* K)
try {
    if (!c || !K) return UVO_INVALID_ARG;
    memcpy(c->mono_K, K, sizeof(c->mono_K));
    c->mono_cam_set = true;
    return uvo_mono_reset(c);
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_mono_reset(uvo_ctx* c)
try {
    if (!c) return UVO_INVALID_ARG;
    drain_in_flight(c);                                    // results of whatever is still in flight are dropped
    c->mono_init_results.clear();
    for (Ctx* l : c->lanes) { if (l->stream) (void)hipStreamSynchronize(l->stream); l->prev_read_pending = false; l->pending = Ctx::Pending(); }
    if (c->n_pending == 0) { c->prev_lane = 0; c->next_lane = 0; }
    c->mono_pipelined = false;
    c->mono_initialized = false; c->mono_use_essential = 1; c->mono_SF = 1.0; c->mono_n_prev = 0;
    const double I[9] = {1,0,0,0,1,0,0,0,1};
    memcpy(c->mono_R, I, sizeof(I)); c->mono_t[0] = c->mono_t[1] = c->mono_t[2] = 0;
    c->mono_mask.clear(); c->mono_good_pts.clear(); c->mono_dev_n = c->mono_dev_M = c->mono_dev_G = 0; c->mono_good_on_host = true; c->mono_matched = false;
    return UVO_OK;
} UVO_ABI_CATCH(c)

// ---------------------------------------------------------------------------------------------------------------------
// mono pipeline (uvo_mono_submit / uvo_mono_collect): the loop body of uvo_mono_step split at the point where the host
// takes over.  Stage A (this thread, the lane's stream, no host sync): upload, detect, match the previous frame's
// descriptors -- read straight from the previous lane's buffers behind its evDet -- against this frame's, ratio test,
// gather the matched point pairs.  Stage B (the lane's worker): the gates on the counts, method selection, the
// host-orchestrated estimators, triangulation and scale.  Every exit of the reference's loop body rolls the state, so the
// previous frame is always the frame before; what is sequential is R, t (kept when no estimator wrote them) and the
// scale factor: uvo_mono_collect applies them in order.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_mono_pairs(const uvo_dmatch* __restrict__ m, const int* __restrict__ count, int cap,
                                                           const uvo_keypoint* __restrict__ prev_kps, const uvo_keypoint* __restrict__ kps,
                                                           uvo_point2f* __restrict__ x1, uvo_point2f* __restrict__ x2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(*count, cap)) return;
    const uvo_keypoint a = prev_kps[m[i].queryIdx], b = kps[m[i].trainIdx];              // VOU:567-568
    x1[i] = uvo_point2f{a.x, a.y}; x2[i] = uvo_point2f{b.x, b.y};
}

// Stage B of one mono frame (VO:276-376 after the matching) on lane L's buffers and stream -- the lane's worker for a pipelined frame,
// the calling thread for uvo_mono_step: fills job.mres / pose_written / sf_written.  The matched points are on the device (d_x1 / d_x2,
// mirrored in pinned memory by k_mono_prep together with select_estimation_method's decision).  A frame whose method is the essential
// matrix takes the device-resident chain (mono.hip: mono_essential_resident, two host syncs); homography-first frames and second
// attempts (VOU:165-178) take the host-pointer operators on the pinned mirrors.  Keypoints, matches and good points are NOT copied to the
// host: uvo_mono_get reads them from the lane's buffers when asked.
static void run_mono_stage_b(uvo_ctx* L, bool stage_a_ok)
{
    Ctx::BJob& j = L->job;
    const Ctx* m = L->master ? L->master : L;
    const uvo_params& p = L->p;
    j.st = UVO_OK; j.err.clear(); j.pose_written = j.sf_written = false;
    memset(&j.mres, 0, sizeof(j.mres));
    L->mono_mask.clear(); L->mono_good_pts.clear();
    L->mono_dev_n = L->mono_dev_M = L->mono_dev_G = 0; L->mono_good_on_host = false;
    if (!stage_a_ok) { j.st = UVO_HIP_ERROR; j.err = "stage A of the frame failed"; return; }
    Range r_b("uvo:mono estimate_relative_pose + triangulation + scale");
    auto hip_ok = [&](hipError_t e) { if (e != hipSuccess && j.st == UVO_OK) { j.st = UVO_HIP_ERROR; j.err = hipGetErrorString(e); } return e == hipSuccess; };
    const int* hc = L->h_countsA[0];
    const int cap = L->cap;
    if (hc[CN_CAND0] > cap) { j.st = UVO_CAPACITY; j.err = "SURF found more keypoints than the context's max_kpts; results would be order-dependent"; return; }
    if (hc[CN_ORI_DROP] != 0) { j.st = UVO_INVALID_ARG; j.err = "SURF orientation: a keypoint had no gradient sample inside the image; OpenCV drops such keypoints, which this build does not do"; return; }
    const int n = hc[CN_NL], M = L->mono_matched ? hc[CN_M] : 0;
    j.mres.initialized = 1; j.mres.n_kps = n;
    L->mono_dev_n = n;
    hipStream_t st = L->stream;
    if (n < p.MIN_NUM_FEATURES) return;                                                    // VO:276-284
    if (M > cap) { j.st = UVO_CAPACITY; j.err = "match count exceeds max_kpts"; return; }
    j.mres.n_matches = M; L->mono_dev_M = M;
    if (M < p.MIN_NUM_FEATURES) return;                                                    // VO:299-307
    if (const char* why = bad_outlier_methods(p)) { j.st = UVO_INVALID_ARG; j.err = why; return; }
    const uvo_point2f *k1 = nullptr, *k2 = nullptr;
    int use_essential = mono_prep_method(L, M, &k1, &k2);                                  // VO:310-317 (select_estimation_method, on the device)
    if (use_essential < 0) { j.st = UVO_HIP_ERROR; j.err = "mono pose stage: select_estimation_method's kernel did not report"; return; }
    L->mono_mask.assign(M, 0);
    // R, t are in/out in the reference (kept when no estimator writes them): run on a sentinel and report whether they were written
    double R[9], t[3];
    const double kSentinel = -7.0e300;
    for (int i = 0; i < 9; i++) R[i] = kSentinel;
    for (int i = 0; i < 3; i++) t[i] = kSentinel;
    int n_in = 0, success = 0;
    bool resident = false;                       // the accepted estimate came from the device-resident chain: triangulation and scale are done
    MonoResident mr;
    std::vector<uvo_point2f> in1, in2;           // extract_inliers' output of a host-pointer attempt
    for (int attempt = 0; attempt < 2 && !success; attempt++) {                            // estimate_relative_pose (VOU:134-180)
        int valid_inliers = 0;
        if (use_essential && attempt == 0) {
            j.st = mono_essential_resident(L, M, m->mono_K, p.ESSENTIAL_OUTLIER_METHOD, p.ESSENTIAL_CONFIDENCE, p.ESSENTIAL_THRESHOLD, (int)p.ESSENTIAL_MAX_ITERS, &mr);
            if (j.st != UVO_OK) { j.err = L->err; return; }
            n_in = mr.n_in;
            if (mr.ok) { memcpy(R, mr.R, sizeof(R)); memcpy(t, mr.t, sizeof(t)); memcpy(L->mono_mask.data(), mr.mask, (size_t)M); valid_inliers = mr.valid_inliers; }
            else memset(L->mono_mask.data(), 0, (size_t)M);
            resident = true;
        } else {
            in1.resize(M); in2.resize(M);
            j.st = estimate_once(L, k1, k2, M, m->mono_K, use_essential, R, t, in1.data(), in2.data(), &n_in, L->mono_mask.data(), &valid_inliers);
            if (j.st != UVO_OK) { j.err = L->err; return; }
            resident = false;
        }
        if (estimate_accepted(p, valid_inliers, M)) success = 1;
        else if (attempt == 0) use_essential = !use_essential;                             // VOU:165-178: the other method, once
    }
    j.pose_written = R[0] != kSentinel;
    if (j.pose_written) { memcpy(j.R, R, sizeof(R)); memcpy(j.t, t, sizeof(t)); }
    j.mres.success = success; j.mres.used_essential = use_essential; j.mres.n_inliers = n_in;
    j.mres.published = 1;
    int valid = success ? 1 : 0;                                                           // VO:335-344
    if (success) {                                                                         // VO:351-376 (success implies the pose was written)
        int G = 0, n_front = 0;
        std::vector<double> zs;
        if (resident) {                                                                    // triangulated, filtered and projected on the device already
            G = mr.G; n_front = mr.n_front;
            L->mono_dev_G = G;
            zs.assign(mr.zs, mr.zs + n_front);
        } else {
            const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
            double P_prev[12], P_curr[12];
            projection_matrix(I, z, m->mono_K, P_prev);
            projection_matrix(R, t, m->mono_K, P_curr);
            if (n_in > 0) {
                if (!hip_ok(hipMemcpyAsync(L->d_x1, in1.data(), sizeof(uvo_point2f) * n_in, hipMemcpyHostToDevice, st))) return;
                if (!hip_ok(hipMemcpyAsync(L->d_x2, in2.data(), sizeof(uvo_point2f) * n_in, hipMemcpyHostToDevice, st))) return;
                if (!hip_ok(hipMemcpyAsync(L->d_xc, in2.data(), sizeof(uvo_point2f) * n_in, hipMemcpyHostToDevice, st))) return;
                if ((j.st = pose_triangulate_extract3d(L, 0, P_prev, P_curr, I, z, R, t, m->mono_K, m->mono_K, nullptr, n_in)) != UVO_OK) { j.err = L->err; return; }   // VO:355-356
                if ((j.st = read_counts(L)) != UVO_OK) { j.err = L->err; return; }
                G = L->h_counts[CN_G];
                L->mono_good_pts.resize((size_t)3 * G);
                if (G && !hip_ok(hipMemcpyAsync(L->mono_good_pts.data(), L->d_good_pts[0], sizeof(double) * 3 * G, hipMemcpyDeviceToHost, st))) return;
                if (!hip_ok(host_sync(L, st))) return;
            }
            L->mono_good_on_host = true;
            for (int i = 0; i < G; i++) {                                                  // convert_3Dpoints_camera (VOU:46-63)
                const double* q = &L->mono_good_pts[3 * (size_t)i];
                double zt = (R[6]*q[0] + R[7]*q[1] + R[8]*q[2]) * 1.0 + t[2] * 1.0;
                if (zt > 0) zs.push_back(q[2]);
            }
            n_front = (int)zs.size();
        }
        j.mres.n_good3d = G;
        if (G < p.MIN_NUM_3DPOINTS) valid = 0;                                             // VO:358
        else {
            j.mres.n_front = n_front;
            if (!zs.empty()) { j.SF = (float)j.range / compute_median(zs); j.sf_written = true; }   // compute_scale_factor (VOU:23-38)
            else valid = 0;
        }
    } else L->mono_good_on_host = true;                                                    // (no good points: the empty host vector is the answer)
    j.mres.valid = valid;
}
// What uvo_mono_collect / uvo_mono_step apply, in frame order, from a finished stage B: R, t (kept when no estimator wrote them), the
// scale factor, the reference's `use_essential`, and the published velocity (mono_output_computation, VO:126-140)
static void apply_mono_result(uvo_ctx* c, const Ctx::BJob& j, double dt, uvo_mono_result* out)
{
    *out = j.mres;
    if (j.mres.n_kps > 0) c->kp_hint = j.mres.n_kps;          // sizes the descriptor launch's small-window grid of the frames to come
    if (j.pose_written) { memcpy(c->mono_R, j.R, sizeof(c->mono_R)); memcpy(c->mono_t, j.t, sizeof(c->mono_t)); }
    if (j.sf_written) c->mono_SF = j.SF;
    c->mono_use_essential = j.mres.published ? j.mres.used_essential : c->mono_use_essential;
    if (j.mres.published) {
        // -SF * R^T * t / dt as one gemm with alpha = (-SF) * (1/dt)
        double alpha = (-c->mono_SF) * (1.0 / dt);
        for (int i = 0; i < 3; i++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += c->mono_R[k*3 + i] * c->mono_t[k];
            out->velocity[i] = acc * alpha;
        }
        out->SF = c->mono_SF;
        memcpy(out->R, c->mono_R, sizeof(out->R)); memcpy(out->t, c->mono_t, sizeof(out->t));
    }
}

// mono loop body, visual_odometry_node::mono_VO (visual_odometry.h:227-245 init, 247-397 main loop, 126-140 output), one frame at a
// time on lane 0: the previous frame's keypoints and descriptors are rolled into d_as_kpsL[0] / d_as_descL[0] (VO:279-282 / 392-395).
// Detection, matching, the point gather and select_estimation_method's kernel are queued without a host sync; the counts are read
// once, then the pose stage runs on the calling thread (run_mono_stage_b).
extern "C" uvo_status uvo_mono_step(uvo_ctx* c, const uint8_t* img, int w, int h, int stride, int mem, double range, double dt,
                                    uvo_mono_result* out)
try {
    if (!c || !img || !out) return UVO_INVALID_ARG;
    if (!c->mono_cam_set) return fail(c, UVO_INVALID_ARG, "uvo_mono_set_camera has not been called");
    if (c->mono_pipelined) return fail(c, UVO_INVALID_ARG, "uvo_mono_step after uvo_mono_submit: call uvo_mono_reset first (the previous frame is held by the pipeline)");
    (void)hipSetDevice(c->device);
    const uvo_params& p = c->p;
    Range r_step("uvo:mono_step");
    memset(out, 0, sizeof(*out));
    c->mono_mask.clear(); c->mono_good_pts.clear();
    c->mono_dev_n = c->mono_dev_M = c->mono_dev_G = 0; c->mono_good_on_host = true;
    UVO_TRY(wait_for_producer(c, c, mem));
    UVO_TRY(surf_upload(c, 0, img, w, h, stride, mem));
    UVO_TRY(detect_dispatch(c, 1));                                                        // VO:238 / VO:274
    // match_features 7-arg (VO:287 -> VOU:551-573) against the rolled previous frame: query = previous, train = this frame (its count
    // stays on the device), ratio test, the matched point pairs, select_estimation_method -- whether the gates let the frame use them
    // is decided below, from the counts
    c->mono_matched = c->mono_initialized && c->mono_n_prev > 0;
    if (c->mono_matched) {
        UVO_TRY(match_knn2(c, c->d_as_descL[0], nullptr, c->mono_n_prev, c->det[0].desc, c->det[0].n, c->cap));
        UVO_TRY(match_ratio_compact(c, nullptr, c->mono_n_prev, (float)p.LOWE_RATIO_THRESHOLD, c->d_matches[0], c->d_nmatch, c->cap));
        hipLaunchKernelGGL(k_gather_mono_pairs, dim3((c->cap + 255) / 256), dim3(256), 0, c->stream, c->d_matches[0], c->d_counts + CN_M, c->cap,
                           c->d_as_kpsL[0], c->det[0].kps, c->d_x1, c->d_x2);
        UVO_HIP_TRY(c, hipGetLastError());
        UVO_TRY(mono_prep_launch(c, c->stream, c->mono_K, p.DISTANCE));
    }
    UVO_HIP_TRY(c, hipMemcpyAsync(c->h_countsA[0], c->d_counts, sizeof(int) * CN_TOTAL, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, host_sync(c, c->stream));
    memcpy(c->h_counts, c->h_countsA[0], sizeof(int) * CN_TOTAL);
    UVO_TRY(check_cand_overflow(c, 1));
    const int n = c->h_counts[CN_NL];
    out->n_kps = n;
    c->kp_hint = n;
    c->mono_dev_n = n;
    auto roll_state = [&]() -> uvo_status {                                                // VO:279-282 / VO:392-395 (stream-ordered: the next frame's kernels follow)
        if (n) {
            UVO_HIP_TRY(c, hipMemcpyAsync(c->d_as_descL[0], c->det[0].desc, sizeof(float) * c->desc_dim() * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
            UVO_HIP_TRY(c, hipMemcpyAsync(c->d_as_kpsL[0], c->det[0].kps, sizeof(uvo_keypoint) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
        }
        c->mono_n_prev = n;
        return UVO_OK;
    };
    if (!c->mono_initialized) {                                                            // VO:227-245
        UVO_TRY(roll_state());
        if (n >= p.MIN_NUM_FEATURES) c->mono_initialized = true;
        return UVO_OK;
    }
    c->job.kind = 1; c->job.range = range;
    run_mono_stage_b(c, true);                                                             // VO:276-376
    if (c->job.st != UVO_OK) { const uvo_status st_ = c->job.st; c->err = c->job.err; return st_; }
    apply_mono_result(c, c->job, dt, out);
    return roll_state();
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_mono_submit(uvo_ctx* c, const uint8_t* img, int w, int h, int stride, int mem, double range)
try {
    if (!c || !img) return UVO_INVALID_ARG;
    if (!c->mono_cam_set) return fail(c, UVO_INVALID_ARG, "uvo_mono_set_camera has not been called");
    const int depth = (int)c->lanes.size();
    if (depth < 2) return fail(c, UVO_INVALID_ARG, "uvo_mono_submit needs at least two lanes (uvo_stereo_set_depth): a frame is matched against the previous lane's buffers");
    if (c->n_pending >= depth) return fail(c, UVO_INVALID_ARG, "uvo_mono_submit: the pipeline is full; collect a frame first (uvo_stereo_set_depth)");
    (void)hipSetDevice(c->device);
    Range r_submit("uvo:mono_submit");
    const uvo_params& p = c->p;
    if (!c->mono_initialized) {                                                            // VO:227-245 on lane 0, synchronous
        // (earlier init frames may still await their collect: they are complete, only their result is queued)
        uvo_mono_result r;
        memset(&r, 0, sizeof(r));
        UVO_TRY(uvo_mono_step(c, img, w, h, stride, mem, range, 1.0, &r));                 // nothing is published before initialisation
        c->mono_init_results.push_back(r);
        c->inflight[c->n_pending++] = Ctx::kInflightMonoInit; c->n_submitted++;                // a synchronous init frame
        c->prev_lane = 0; c->next_lane = 1 % depth;
        UVO_HIP_TRY(c, hipEventRecord(c->evDet, c->stream));                               // lane 0 holds the frame the next one matches against
        UVO_TRY(prime_lanes(c, w, h));
        if (c->use_sift()) for (Ctx* l : c->lanes) { uvo_status st_ = sift_prepare_lane(l, w, h, 1); if (st_ != UVO_OK) { if (l != c) c->err = l->err; return st_; } }
        return UVO_OK;
    }
    const int li = c->next_lane;
    uvo_ctx* L = static_cast<uvo_ctx*>(c->lanes[li]);
    Ctx* P = c->lanes[c->prev_lane];
    c->mono_pipelined = true;                  // the previous frame now lives in a lane's buffers, not in uvo_mono_step's rolled state
#define LANE_TRY(expr) do { uvo_status st_ = (expr); if (st_ != UVO_OK) { if (L != c) c->err = L->err; return st_; } } while (0)
    hipStream_t st = L->stream;
    L->pending = Ctx::Pending(); L->pending.used = true;
    if (L->prev_read_pending) { UVO_HIP_TRY(c, hipStreamWaitEvent(st, L->evPrevRead, 0)); L->prev_read_pending = false; }   // the frame after this lane's last one has read its buffers
    UVO_TRY(wait_for_producer(c, L, mem));
    LANE_TRY(surf_upload(L, 0, img, w, h, stride, mem));
    if (c->a_overlap_mono > 0 && depth > c->a_overlap_mono) {                              // as uvo_stereo_submit: paced by this thread; one image per frame, so more stage As side by side
        Ctx* H = c->lanes[(li + depth - c->a_overlap_mono) % depth];
        if (c->n_pending >= c->a_overlap_mono) (void)uvo::poll_event(H->evA[1]);
    }
    LANE_TRY(detect_dispatch(L, 1));                                                       // VO:274
    UVO_HIP_TRY(c, hipEventRecord(L->evDet, st));
    int* cn = L->d_counts;
    const int cap = c->cap;
    if (P != L) UVO_HIP_TRY(c, hipStreamWaitEvent(st, P->evDet, 0));
    // match_features 7-arg (VO:287 -> VOU:551-573): query = previous frame, train = this frame; the counts stay on the device
    LANE_TRY(match_knn2(L, P->det[0].desc, P->det[0].n, cap, L->det[0].desc, L->det[0].n, cap));
    LANE_TRY(match_ratio_compact(L, P->det[0].n, cap, (float)p.LOWE_RATIO_THRESHOLD, L->d_matches[0], cn + CN_M, cap));
    hipLaunchKernelGGL(k_gather_mono_pairs, dim3((cap + 255) / 256), dim3(256), 0, st, L->d_matches[0], cn + CN_M, cap,
                       P->det[0].kps, L->det[0].kps, L->d_x1, L->d_x2);
    UVO_HIP_TRY(c, hipGetLastError());
    LANE_TRY(mono_prep_launch(L, st, c->mono_K, p.DISTANCE));                              // normalised points, select_estimation_method, pinned mirrors
    L->mono_matched = true;
    if (P != L) { UVO_HIP_TRY(c, hipEventRecord(P->evPrevRead, st)); P->prev_read_pending = true; }
    UVO_HIP_TRY(c, hipMemcpyAsync(L->h_countsA[0], L->d_counts, sizeof(int) * CN_TOTAL, hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, hipEventRecord(L->evA[0], st));
    UVO_HIP_TRY(c, hipEventRecord(L->evA[1], st));                                          // the same point for other streams (see uvo_ctx.h)
    c->prev_lane = li; c->next_lane = (li + 1) % depth;
    c->inflight[c->n_pending++] = li; c->n_submitted++;
    L->t_handover_us = uvo::now_us();
    {
        std::lock_guard<std::mutex> lk(L->mu);
        L->job.kind = 1; L->job.range = range; L->job.state = 1; L->job_state_a.store(1, std::memory_order_release);
    }
    L->cv.notify_all();
    return UVO_OK;
#undef LANE_TRY
} UVO_ABI_CATCH(c)

extern "C" uvo_status uvo_mono_collect(uvo_ctx* c, double dt, uvo_mono_result* out)
try {
    if (!c || !out) return UVO_INVALID_ARG;
    if (c->n_pending <= 0) return fail(c, UVO_INVALID_ARG, "uvo_mono_collect: nothing submitted");
    if (c->inflight[0] == Ctx::kInflightStereoInit || (c->inflight[0] >= 0 && c->lanes[c->inflight[0]]->job.kind != 1))
        return fail(c, UVO_INVALID_ARG, "uvo_mono_collect: the oldest entry in flight is a stereo pair (uvo_stereo_collect)");
    (void)hipSetDevice(c->device);
    const int li = c->inflight[0];
    for (int i = 1; i < c->n_pending; i++) c->inflight[i - 1] = c->inflight[i];
    c->n_pending--; c->n_collected++;
    if (li < 0) {                                          // a synchronous init frame (its state is already applied)
        *out = c->mono_init_results.front();
        c->mono_init_results.pop_front();
        c->last_lane = 0;
        return UVO_OK;
    }
    uvo_ctx* L = static_cast<uvo_ctx*>(c->lanes[li]);
    c->last_lane = li;
    L->pending.used = false;
    {
        poll_job_state(L, 2, 5000.0);
        std::unique_lock<std::mutex> lk(L->mu);
        L->cv.wait(lk, [&] { return L->job.state == 2; });
        L->job.state = 0; L->job_state_a.store(0, std::memory_order_release);
    }
    const Ctx::BJob& j = L->job;
    if (j.st != UVO_OK) return fail(c, j.st, j.err.c_str());
    // (uvo_mono_get reads the frame's intermediates from this lane, c->last_lane, until the lane is submitted to again)
    apply_mono_result(c, j, dt, out);
    return UVO_OK;
} UVO_ABI_CATCH(c)

// last frame's intermediates.  Keypoints, matches and (after the device-resident pose chain) the good points live in the frame's lane
// until that lane is submitted to again: they are copied out here, when asked for, not once per frame.
extern "C" int uvo_mono_get(uvo_ctx* c, const char* what, void* out, int cap_bytes)
try {
    if (!c || !what || !out) return 0;
    (void)hipSetDevice(c->device);
    const void* src = nullptr; size_t nb = 0; int count = 0; bool on_device = false;
    std::string w(what);
    const Ctx* f = c->mono_pipelined ? c->lanes[c->last_lane] : c;       // the lane of the last collected frame
    if (w == "kps") { src = f->det[0].kps; count = f->mono_dev_n; nb = (size_t)count * sizeof(uvo_keypoint); on_device = true; }
    else if (w == "matches") { src = f->d_matches[0]; count = f->mono_dev_M; nb = (size_t)count * sizeof(uvo_dmatch); on_device = true; }
    else if (w == "mask") { src = f->mono_mask.data(); count = (int)f->mono_mask.size(); nb = (size_t)count; }
    else if (w == "good_pts") {
        if (f->mono_good_on_host) { src = f->mono_good_pts.data(); count = (int)f->mono_good_pts.size() / 3; }
        else { src = f->d_good_pts[0]; count = f->mono_dev_G; on_device = true; }
        nb = (size_t)count * 3 * sizeof(double);
    }
    else return 0;
    if (nb > (size_t)cap_bytes) return -count;
    if (nb) {
        if (on_device) { if (hipMemcpy(out, src, nb, hipMemcpyDeviceToHost) != hipSuccess) return 0; }
        else memcpy(out, src, nb);
    }
    return count;
} UVO_ABI_CATCH_RET(c, 0)

// ------------------------------------------------------------------------------------------ pipeline trace
// The UVO_TRACE machinery through the ABI: device timestamps (hipEvents on the lane's streams) and host timestamps (steady clock) of
// every pipelined pair's phases, kept in a ring of 256 pairs per lane.
extern "C" uvo_status uvo_trace_enable(uvo_ctx* c, int on)
try {
    if (!c) return UVO_INVALID_ARG;
    if (c->n_pending != 0) { c->err = "the pipeline trace cannot change while pairs are in flight"; return UVO_INVALID_ARG; }
    (void)hipSetDevice(c->device);
    for (Ctx* l : c->lanes) {
        if (on) { UVO_HIP_TRY(c, trace_alloc(l)); for (auto& r : l->trace) r.pair = -1; l->trace_count = 0; l->trace_cur = -1; }
        else l->trace_on = false;
    }
    return UVO_OK;
} UVO_ABI_CATCH(c)
extern "C" int uvo_trace_read(uvo_ctx* c, uvo_trace_row* rows, int cap)
try {
    if (!c || (cap > 0 && !rows) || c->n_pending != 0) return -1;
    (void)hipSetDevice(c->device);
    for (Ctx* l : c->lanes) { if (l->stream) (void)hipStreamSynchronize(l->stream); if (l->pnp_stream) (void)hipStreamSynchronize(l->pnp_stream); }
    hipEvent_t ref = nullptr; long long ref_pair = -1; double ref_host = 0;
    for (Ctx* l : c->lanes) for (auto& r : l->trace) if (r.pair >= 0 && (ref_pair < 0 || r.pair < ref_pair)) { ref = r.ev[0]; ref_pair = r.pair; ref_host = r.host_us[0]; }
    if (!ref) return 0;
    int n = 0;
    for (Ctx* l : c->lanes) for (auto& r : l->trace) {
        if (r.pair < 0) continue;
        if (n < cap) {
            uvo_trace_row& w = rows[n];
            w.pair = r.pair; w.lane = l->lane_id; w.b_used = r.b_used ? 1 : 0;
            for (int k = 0; k < 6; k++) {
                w.dev_ms[k] = -1.f;
                if (k < 3 || r.b_used) (void)hipEventElapsedTime(&w.dev_ms[k], ref, r.ev[k]);
                w.host_ms[k] = r.host_us[k] > 0 ? (r.host_us[k] - ref_host) * 1e-3 : -1.0;
            }
            for (int k = 6; k < 8; k++) { w.dev_ms[k] = -1.f; if (r.det_marked) (void)hipEventElapsedTime(&w.dev_ms[k], ref, r.ev[k]); }
        }
        n++;
    }
    std::sort(rows, rows + std::min(n, cap), [](const uvo_trace_row& a, const uvo_trace_row& b) { return a.pair < b.pair; });
    return n;
} UVO_ABI_CATCH_RET(c, -1)

// ------------------------------------------------------------------------------------------ timing
extern "C" uvo_status uvo_timing_enable(uvo_ctx* c, int on)
try {
    if (!c) return UVO_INVALID_ARG;
    if (c->n_pending != 0) { c->err = "timing mode cannot change while pairs are in flight"; return UVO_INVALID_ARG; }
    for (Ctx* l : c->lanes) l->timing = on != 0;
    return UVO_OK;
} UVO_ABI_CATCH(c)
extern "C" int uvo_timing_count(uvo_ctx*) { return ST_COUNT; }
extern "C" const char* uvo_timing_name(uvo_ctx*, int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }
extern "C" uvo_status uvo_timing_get(uvo_ctx* c, int i, double* ms, long long* launches)
try {
    if (!c || i < 0 || i >= ST_COUNT) return UVO_INVALID_ARG;
    double t = 0; long long n = 0;
    for (Ctx* l : c->lanes) { t += l->stage_ms[i]; n += l->stage_n[i]; }
    if (ms) *ms = t;
    if (launches) *launches = n;
    return UVO_OK;
} UVO_ABI_CATCH(c)
extern "C" uvo_status uvo_timing_reset(uvo_ctx* c)
try {
    if (!c) return UVO_INVALID_ARG;
    for (Ctx* l : c->lanes) for (int i = 0; i < ST_COUNT; i++) { l->stage_ms[i] = 0; l->stage_n[i] = 0; }
    return UVO_OK;
} UVO_ABI_CATCH(c)
