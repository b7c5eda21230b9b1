// uvo_ctx.h -- internal context of libuvo_hip: device workspaces, stream, stage timers.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <vector>
#include <deque>
#include <stdint.h>
#include "../../include/uvo_hip.h"
#include "uvo_experimental.h"

namespace uvo {

#define UVO_HIP_TRY(c, expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                     \
            return UVO_HIP_ERROR;                                                             \
        }                                                                                     \
    } while (0)

#define UVO_TRY(expr)                                                                         \
    do { uvo_status _s = (expr); if (_s != UVO_OK) return _s; } while (0)

enum Stage {
    ST_INTEGRAL = 0, ST_HESSIAN_O0, ST_HESSIAN_O1, ST_HESSIAN_O2, ST_HESSIAN_O3, ST_SORT, ST_DESCRIPTOR, ST_MATCH, ST_MATCH_MERGE, ST_GATHER,
    ST_TRIANGULATE, ST_EXTRACT3D, ST_PNP_HYP, ST_PNP_SCORE, ST_PNP_REFIT, ST_COUNT
};
static const char* const kStageNames[ST_COUNT] = {
    "integral", "hessian_nms_o0", "hessian_nms_o1", "hessian_nms_o2", "hessian_nms_o3", "kp_sort", "descriptor64", "match_top2", "match_merge", "gather",
    "triangulate", "extract3d", "pnp_epnp5", "pnp_score", "pnp_refit"
};

// fixed slots of Ctx::d_counts / h_counts
enum { CN_T = 0, CN_G = 1, CN_NINL = 2, CN_NQA = 3, CN_NQB = 4, CN_MEFF = 5, CN_NL = 6, CN_NR = 7,
       CN_CAND0 = 8, CN_CAND1 = 9, CN_M = 10, CN_TRAW = 11, CN_AS0 = 12, CN_AS1 = 13, CN_BIG0 = 14, CN_BIG1 = 15, CN_SURV = 16, CN_BIGL0 = 17, CN_BIGL1 = 18, CN_ORI_DROP = 19, CN_TOTAL = 20 };   // CN_BIG*: large-window keypoints per image, CN_BIGL*: those too wide for three-column tasks; CN_SURV: NMS survivors
// The gate ladder of the stereo loop, evaluated on the device by the last thread of the kernel that produces the count it
// tests (no launches of their own): mode 1 = VO:567 after the stereo matcher's compaction, mode 2 = VO:626 after the
// triangular matcher's.
struct GateArgs { int mode; int* cn; int min_features, cap; int* as_curr_n; const int* as_prev_n; };

static const int kSumPad = 256;      // ints of slack before and after each integral image (see Ctx::d_sum_base)
static const int kMaxHyp = 2048;      // RANSAC hypotheses evaluated per call (>= ITERATIONS_COUNT)

// A sample of a middle layer (1..3) that beat the Hessian threshold, its eight in-layer neighbours and every adjacent layer
// the detection kernel computes (surf.hip): n9 = the 3 x 3 x 3 neighbourhood, layer L-1 first; the row of an outer layer
// (0 or 4) is filled in by k_hessian_finish.
struct Survivor { int im, octave, L, i, j; float n9[27]; };

struct DetectSet {                    // one image's detector outputs (device)
    uvo_keypoint* kps;                // sorted, cap
    float* desc;                      // cap x 64
    int* n;                           // device count (clamped to cap)
};

struct Ctx {
    uvo_params p;
    int device = 0, max_w = 0, max_h = 0, cap = 0;
    hipStream_t stream = nullptr;
    std::string err, warning, policy_text;
    hipStream_t producer_stream = nullptr; bool has_producer = false;   // master: uvo_ctx_set_producer_stream
    hipEvent_t evProducer = nullptr;             // this lane's marker on the producer stream

    // ---- SURF ----
    uint8_t* d_img[2] = {nullptr, nullptr};
    const uint8_t* img[2] = {nullptr, nullptr};   // what the kernels read: d_img[i], or the caller's device image when it can be read in place
    int32_t* d_sum[2] = {nullptr, nullptr};      // (max_h+1) x (max_w+1); = d_sum_base + kSumPad
    int32_t* d_sum_base[2] = {nullptr, nullptr}; // allocation: kSumPad ints of slack on both sides, so the Hessian tile fill can
                                                 // read whole quads of a clamped row without per-element bounds checks
    int32_t* d_colpart = nullptr;                // [2][strips of 8 rows][colpart_stride]: column sums of the strips, then their scan
    int colpart_stride = 0;
    int32_t* d_planes[2] = {nullptr, nullptr};   // integral de-interleaved by (row & 3, col & 3): 16 planes of plane_ph x plane_pw
    int plane_pw = 0, plane_stride = 0;
    uvo_keypoint* d_cand[2] = {nullptr, nullptr};// unsorted candidates
    int* d_cand_n = nullptr;                     // [2] raw atomic counters
    Survivor* d_surv = nullptr; int surv_cap = 0; // NMS survivors of a frame (both images, every octave), 4 * cap
    void* d_octpat = nullptr;                    // the four OctavePat of the current image size, for k_hessian_finish
    std::vector<unsigned char> h_octpat;         // what d_octpat holds
    uint32_t* d_hess_order = nullptr; size_t d_hess_order_cap = 0;       // merged detection launch: octave and tile (column, row) of every block
    std::vector<uint32_t> h_hess_order; int hess_order_key[3] = {0, 0, 0};
    int4* d_big_par = nullptr; int* d_big_n = nullptr;   // [2][cap] (sorted index, win, start_x, start_y) of large-window keypoints in append order, then [2][cap] by descending win; [2] counts
    struct AreaTab* d_area_tabs = nullptr;       // [kMaxWin + 1][21] INTER_AREA resize tables of every descriptor window size (surf_build_area_tables)
    float* d_ori_w = nullptr;                    // 13 x 13 Gaussian weights of the orientation samples (SURF_UPRIGHT = false)
    int* d_area_iscale = nullptr;                // [kMaxWin + 1] integer scale of the sizes resizeAreaFast_ handles, else 0
    uint8_t* d_big_patch = nullptr;              // [2][cap][448] 21x21 patches of the large-window keypoints
    int* d_rank = nullptr;                       // [2][cap] sort ranks (zero between frames)
    DetectSet det[2];                            // current left/right
    int img_w = 0, img_h = 0;
    float h_DW[400];                             // descriptor Gaussian weights (host copy)
    float* d_DW = nullptr;

    // ---- matcher ----
    // two slots each (the stereo loop's two matches share their launches, match_knn2_two)
    float4* d_mpart = nullptr;                   // [2] shortlist per (train chunk, query): the four smallest keys
    float* d_mscratch = nullptr;                 // [2][train chunk] max |t|^2 of the chunk
    int* d_knn_idx = nullptr;  float* d_knn_dist = nullptr;   // [2][cap][2]
    float* d_tmp_desc[2] = {nullptr, nullptr};   // staging for the standalone match API
    uvo_dmatch* d_matches[2] = {nullptr, nullptr};            // [0] stereo (L-R), [1] triangular (prev-curr)
    int* d_nmatch = nullptr;                     // [2]

    // ---- stereo state: "after stereo match" sets, double-buffered (prev / curr) ----
    uvo_keypoint* d_as_kpsL[2] = {nullptr, nullptr};
    uvo_keypoint* d_as_kpsR[2] = {nullptr, nullptr};
    float* d_as_descL[2] = {nullptr, nullptr};
    int* d_as_n = nullptr;                       // [2] counts (device)
    // the rows of a set, triangulated (triangulatePoints + the per-point half of extract_3Dpoints are functions of a row's two keypoints
    // and the rig): filled by the pair that builds the set, read by the next pair through its triangular matches (pose.hip: k_stereo_tail)
    float4* d_as_pts4[2] = {nullptr, nullptr};  double* d_as_cam1[2] = {nullptr, nullptr};  int* d_as_flag[2] = {nullptr, nullptr};
    bool vo_initialized = false;
    std::vector<uvo_dmatch> init_matches;        // results_match_prev (VO:468): survives failed init attempts
    double K_left[9], K_right[9], R_right[9], t_right[3], P_eye_left[12], P_right[12];
    bool rig_set = false;
    double t_prev_curr[3] = {0, 0, 0}, rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};

    // ---- triangulation / extract_3Dpoints ----
    uvo_point2f *d_x1 = nullptr, *d_x2 = nullptr, *d_xc = nullptr;   // prevL, prevR, currL points of the T matches
    float4* d_pts4 = nullptr;                    // T homogeneous points
    double* d_cam1 = nullptr;                    // T x 3
    int* d_flag = nullptr;                       // T
    // outputs of extract_3Dpoints = inputs of PnP, double-buffered by pipeline slot (pair index & 1)
    double* d_good_pts[2] = {nullptr, nullptr};  int* d_good_idx[2] = {nullptr, nullptr};     // G x 3, G
    float* d_opts[2] = {nullptr, nullptr};  uvo_point2f* d_ipts[2] = {nullptr, nullptr};      // G x 3 f32 object points, G image points
    int* d_tmp_idx = nullptr;                    // extract_3Dpoints scratch
    int* d_tmp_row = nullptr;                    // the same, rows of the previous set (k_stereo_tail)
    int* d_counts = nullptr;                     // misc device counters: [0]=T, [1]=G, [2]=n_inliers, ...
    int* h_counts = nullptr;                     // pinned mirror

    // ---- PnP RANSAC ----
    int* d_subsets = nullptr;                    // kMaxHyp x 5
    int* h_subsets = nullptr;                    // pinned
    double* d_models = nullptr;                  // kMaxHyp x 6 (rvec | tvec)
    int* d_hcount = nullptr;  int* h_hcount = nullptr;               // inlier count per hypothesis
    int* d_inliers = nullptr;                    // cap
    double* d_refit = nullptr;                   // refit workspace: pws 3n, us 2n, alphas 4n, pcs 3n, tmp n, M 24n, small
    double* d_pose = nullptr;  double* h_pose = nullptr;             // rvec(3) tvec(3)
    unsigned* d_rng_raw = nullptr;               // the first raw outputs of cv::RNG((uint64)-1): getSubset's stream (pose.hip, speculative round)
    void* d_spec = nullptr; void* h_spec = nullptr;                  // PnpSpecState of the lane's pair, device / pinned
    bool spec_queued = false;                    // this lane's pair has a speculative PnP round queued on `stream`
    hipEvent_t evSync = nullptr; bool inline_b = false;              // synchronous step: end of the pair's device work, polled by the calling thread, which runs stage B itself
    bool in_sync_step = false;                   // master: the submit in progress is uvo_stereo_step's (one pair in flight)

    // ---- pipeline: stage A (detect .. extract_3Dpoints) on `stream`, stage B (PnP) on `pnp_stream` ----
    // A context is also one LANE of the stereo pipeline.  Lane 0 is the context the caller holds; uvo_stereo_set_depth
    // adds child contexts (own buffers, streams and stage-B worker thread) so that `depth` consecutive pairs are in
    // flight at once: pair k runs on lane k mod depth.  The only data one pair takes from the previous one is its
    // "after stereo match" set, read from the previous lane's buffers behind an event.
    hipStream_t pnp_stream = nullptr;
    int* d_countsB = nullptr;  int* h_countsB = nullptr;             // [0] = n_inliers
    hipEvent_t evA[2] = {nullptr, nullptr};      // stage A of this lane's pair finished (counts in h_countsA[0]): [0] is the one the lane's worker
                                                 // blocks on (hipEventSynchronize), [1] its twin for other lanes' hipStreamWaitEvent -- the runtime holds
                                                 // an event's lock while a host thread waits on it, so a stream wait on the SAME event blocked the
                                                 // submitting thread until the event completed (133 us per pair at C3)
    // How the host waits inside a running pipeline (env UVO_WORKER_WAIT = spin | sleep | block-all; default auto):
    //   1 auto (default) -- `spin` when the process's CPU budget (affinity mask, cgroup quota, UVO_CPU_BUDGET) carries a polling
    //                  thread per lane (>= 2 (depth + 2) logical CPUs), else `sleep`
    //   0 spin      -- always poll (hipEventQuery + pause): one busy host thread per waiting worker, up to `depth` per context
    //   3 sleep     -- the worker's long wait, for the end of its pair's stage A, sleeps ON A TIMER through the first four fifths of a
    //                  running mean of the stage's length, then polls.  Keeps a rank at one spinning submitter + at most max_b polling
    //                  workers; on a loaded host a timer sleep was seen to overrun by 8-10 ms about once in 3000 pairs
    //   2 block-all -- every wait, the PnP stage's two short ones included, sleeps on the GPU's interrupt (hipEventBlockingSync,
    //                  evBlock): for hosts with fewer cores than threads.  Until round 4 the stage-A wait did this by default, and about
    //                  one such wait in a hundred woke 3-4 ms late -- the stall behind round 3's 1990 pairs/s driver record
    // Inside the PnP stage the two short syncs (host_sync) poll in every mode but block-all.
    // Which thread runs the PnP stage of a pipelined pair is a separate choice: stage_b_mode below.
    int worker_wait = 1;
    // What the lane's threads read instead of walking the master's lane list (which set_depth rewrites while workers run): how this
    // lane's waits behave -- 0 poll, 1 timed sleep + poll, 2 sleep on the interrupt.  Stored by create_one / set_depth.
    std::atomic<int> wait_eff{1};
    // Who drives the PnP stage of a PIPELINED pair (env UVO_STAGE_B = worker | device; default auto):
    //   0 worker -- the lane's worker thread wakes at the end of stage A, draws the subsets, launches hypotheses + scoring, waits, replays
    //               the scan, launches mask + refit, waits (two host round trips; `depth` workers + the submitter = 7-9 busy host
    //               threads per GPU at depth 6 when they poll)
    //   1 device -- the first RANSAC round (k_pnp_*_spec, pose.hip) is queued on the lane's PnP stream behind stage A's event at submit
    //               time; nobody is handed the pair.  uvo_stereo_collect waits for the round's event, replays the scan over the counts
    //               in pinned memory with the host's libm (pose_pnp_spec_accept) and takes the device's pose when it arrives at the
    //               same winner; otherwise -- the scan needs more than the round's 64 hypotheses, too few inliers for the parallel
    //               refit, exactly five points -- the COLLECTING thread runs the host-driven stage.  Results are those of the worker
    //               path by construction; a rank then keeps ONE host thread busy (submit / collect), whatever the depth.
    //   auto     -- device when the CPU budget cannot carry a polling thread per lane, else worker
    int stage_b_mode = -1;                       // master: -1 auto, 0 worker, 1 device
    bool dev_b = false;                          // lane: this lane's pair was queued device-driven (uvo_stereo_collect finishes it)
    hipEvent_t evB = nullptr;                    // lane: the device-driven round of this lane's pair has finished (on pnp_stream)
    double cpu_budget = 0;                       // master: logical CPUs this process may keep busy (host_cpu_budget(), ctx.hip)
    double stage_a_mean_us = 0;                  // lane (its worker's own): hand-over -> end of stage A, running mean (the timed sleep)
    double t_handover_us = 0;                    // lane: when the pair's stage A was handed to the worker
    std::atomic<int> job_state_a{0};             // lane: job.state again, for the threads that poll for it before they sleep on cv (ctx.hip: wait_job_state)
    hipEvent_t evBlock = nullptr;                // hipEventBlockingSync marker for host_sync()
    hipEvent_t evPoll = nullptr;                 // host_sync()'s marker when it polls
    int* h_countsA[2] = {nullptr, nullptr};      // pinned copy of d_counts
    hipEvent_t evAS = nullptr;                   // this lane's "after stereo match" set is written
    int a_overlap = 2;                           // master: stage As (detect .. extract_3Dpoints) allowed side by side (env UVO_A_OVERLAP, 0 = no limit)
    int a_overlap_mono = 4;                      // the same for mono frames (one image each; env UVO_A_OVERLAP_MONO)
    // Two-pair launches (uvo_stereo_set_batch(c, 2); env UVO_BATCH): uvo_stereo_submit holds every first pair of two back until the
    // next one arrives and queues the stage As of both -- lanes i and i + 1 -- as ONE set of launches on lane i's stream.  A pair
    // waiting alone is queued by the first collect that needs it.  Results are those of single launches, pair for pair.
    int batch = 1;                               // master: pairs per launch set (1 or 2)
    int a_overlap2 = 2;                          // master: two-pair launch sets allowed side by side (env UVO_A_OVERLAP2)
    struct StagePlan { int lane = -1, prev_lane = 0, prev_buf = 0, curr = 0, pending_before = 0; bool prev_sync = true; int trace_slot = -1; };
    StagePlan plan;                              // lane: this lane's pair as uvo_stereo_submit planned it
    int stashed_lane = -1;                       // master: lane of the pair that waits for its partner (-1: none)
    std::vector<Ctx*> lanes;                     // master only: lanes[0] == this
    // keypoints per image of the last pair / frame whose counts reached the host (master; 0: none yet): sizes the descriptor launch's
    // small-window part -- a grid of max_kpts workgroups spends a sixth of the launch's wave-time on workgroups that find no keypoint
    int kp_hint = 0;
    Ctx* master = nullptr;                       // children only
    int lane_id = 0;
    int primed_w = 0, primed_h = 0;              // image size this lane's streams and detector tables are ready for (prime_lanes)
    int as_w = 0;                                // the as-buffer this lane writes next
    int prev_lane = 0, prev_buf = 0;             // master: where the previous pair's as-set lives
    bool prev_sync = true;                       // master: that set was written synchronously (init step), no event to wait for
    int next_lane = 0;                           // master: lane of the next submitted pair
    struct Pending { bool used = false; };
    static const int kInflightMonoInit = -1, kInflightStereoInit = -2;      // inflight[] entries of the synchronous init frames / pairs (else: the lane)
    Pending pending;                             // this lane's pair
    static const int kMaxDepth = 16;
    int inflight[kMaxDepth]; int n_pending = 0;          // master: lanes of the submitted, not yet collected pairs (FIFO)
    long long n_submitted = 0, n_collected = 0;
    int last_lane = 0;
    // stage-B worker of this lane
    struct BJob {
        int state = 0;                           // 0 idle, 1 queued, 2 done
        int kind = 0;                            // 0: stereo PnP stage, 1: mono pose stage
        uvo_status st = UVO_OK; std::string err;
        int ran = 0, ninl = 0, ok = 0, wrote = 0; double rvec[3], tvec[3];
        // mono (uvo_mono_submit): inputs and what uvo_mono_collect applies in order
        double range = 0;
        uvo_mono_result mres;                    // counters and flags of the frame (pose, SF, velocity are filled by collect)
        bool pose_written = false; double R[9], t[3];      // estimate_relative_pose wrote R, t (kept from the previous frame otherwise)
        bool sf_written = false; double SF = 0;
    } job;
    // mono pipeline: this lane's keypoints/descriptors are ready (evDet); the following frame has finished reading them (evPrevRead)
    hipEvent_t evDet = nullptr, evPrevRead = nullptr; bool prev_read_pending = false;
    std::thread worker; std::mutex mu; std::condition_variable cv; bool quit = false;
    std::mutex b_mu; std::condition_variable b_cv; int b_running = 0, max_b = 3;   // master: PnP stages running / allowed at once
    int max_b_mono = 10;                         // ... and mono pose stages (env UVO_MAX_B_MONO): 2 ms of host-orchestrated, latency-bound kernels each (the
                                                 // five-point solver keeps one wave per SIMD busy for 0.9 ms), so many of them side by side cost little

    // last-step bookkeeping for uvo_stereo_get
    int last_nL = 0, last_nR = 0, last_M = 0, last_T = 0, last_G = 0, last_ninl = 0;

    // ---- mono stage (mono.hip) ----
    void* mono_ws = nullptr;                     // MonoWs*, allocated on first use
    void* pre_ws = nullptr;                      // PreWs* (get_image), allocated on first use
    void* codec_ws = nullptr;                    // CodecWs* (uvo_decode_image), allocated on first use
    void* orb_ws = nullptr;                      // OrbWs* (uvo_orb_detect): parameters, sampling table, buffers of the last image size
    void* akaze_ws = nullptr;                    // AkazeWs* (uvo_akaze_detect), allocated on first use per image size
    void* sift_ws[2] = {nullptr, nullptr};       // SiftWs* per image slot (uvo_sift_detect uses slot 0), allocated on first use
    int feature_sift = 0;                        // the reference's global FEATURE_DETECTOR == "SIFT" (uvo_ctx_set_feature_detector); read from the master context
    double mono_K[9]; bool mono_cam_set = false, mono_initialized = false, mono_pipelined = false;
    int mono_use_essential = 1;                  // the reference's global `use_essential` (VOH:89)
    double mono_R[9] = {1,0,0,0,1,0,0,0,1}, mono_t[3] = {0,0,0}, mono_SF = 1.0;
    // the last frame's intermediates for uvo_mono_get: keypoints (det[0].kps), matches (d_matches[0]) and, after the device-resident pose
    // chain, the good points (d_good_pts[0]) stay in the lane's device buffers -- counts here -- until someone asks
    int mono_dev_n = 0, mono_dev_M = 0, mono_dev_G = 0; bool mono_good_on_host = true;
    bool mono_matched = false;                   // this lane's frame was matched against a previous one (CN_M is this frame's)
    int mono_n_prev = 0;                         // rows of the prev descriptors kept in d_as_descL[0]
    std::deque<uvo_mono_result> mono_init_results;   // uvo_mono_submit: results of the synchronous init frames awaiting their collect
    std::deque<uvo_stereo_result> stereo_init_results;   // uvo_stereo_submit: the same for the stereo init pairs (VO:474-520)
    std::vector<uint8_t> mono_mask; std::vector<double> mono_good_pts;

    // ---- UVO_TRACE=<file>: device timestamps of every pipelined pair's phases (hipEvents with timing), written as CSV by
    // uvo_ctx_destroy: pair, lane, A begin, detection end, A end, B begin, B hypotheses scored, B end (ms since the first) ----
    static const int kTraceRing = 256;
    struct TraceRec { long long pair = -1; hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; bool b_used = false, det_marked = false;   // [6], [7]: around the detection launch (k_hessian_nms_all), recorded when det_marked
                      double host_us[6] = {0, 0, 0, 0, 0, 0}; };     // steady clock: submit entered, pacing wait over, submit returned, worker past stage A's event,
                                                                    // worker holds a PnP slot, stage B done
    std::vector<TraceRec> trace; int trace_cur = -1; long long trace_count = 0;
    bool trace_on = false;                       // UVO_TRACE=<file> at creation, or uvo_trace_enable (the events are created at first use)

    int match_dim = 0;                           // uvo_match_knn2*_dim: row width of the standalone matcher for the duration of one call (0 = SURF's)
    bool use_sift() const { return (master ? master : this)->feature_sift != 0; }
    int desc_dim() const { return match_dim ? match_dim : (use_sift() || p.SURF_EXTENDED ? 128 : 64); }      // SURF::descriptorSize(): floats per descriptor row

    // ---- timing ----
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double stage_ms[ST_COUNT] = {0};
    long long stage_n[ST_COUNT] = {0};
};

// RAII-less stage timer: records events around a launch when timing is on (and synchronises, so
// timing mode serialises the stream -- bench.py uses it only in its roofline leg).
struct StageTimer {
    Ctx* c; int st; hipStream_t s;
    StageTimer(Ctx* c_, int st_, hipStream_t s_ = nullptr) : c(c_), st(st_), s(s_ ? s_ : c_->stream) { if (c->timing) (void)hipEventRecord(c->ev0, s); }
    ~StageTimer()
    {
        if (!c->timing) return;
        (void)hipEventRecord(c->ev1, s);
        (void)hipEventSynchronize(c->ev1);
        float ms = 0; (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->stage_ms[st] += ms; c->stage_n[st] += 1;
    }
};

// UVO_ROCTX=1: named ranges around the phases of a step (rocprofv3 --marker-trace shows them beside the kernels).  The marker
// library is looked up at run time (librocprofiler-sdk-roctx.so, then libroctx64.so); without the variable a range costs one load.
void range_push(const char* name);
void range_pop();
struct Range { explicit Range(const char* name) { range_push(name); } ~Range() { range_pop(); } Range(const Range&) = delete; Range& operator=(const Range&) = delete; };

// Wait on the host for everything queued on `st` so far, the way lane `c`'s worker_wait says (polling or sleeping)
hipError_t host_sync(Ctx* c, hipStream_t st);
// hot-path host waits by query loops: never the runtime's sleep on an interrupt (ctx.hip)
hipError_t poll_event(hipEvent_t ev);

// surf.hip
uvo_status surf_upload(Ctx* c, int slot, const uint8_t* gray, int w, int h, int stride, int mem);
uvo_status surf_integral(Ctx* c, int nimg);
uvo_status surf_build_area_tables(Ctx* c);
uvo_status surf_prepare(Ctx* c, int w, int h);                           // per-image-size tables of this lane (idempotent)
uvo_status surf_detect(Ctx* c, int nimg, int gate_min_features = -1);
uvo_status surf_detect_lanes(Ctx* c, Ctx* c2, int nimg, int gate_min_features);     // c2: a second lane's pair in the same launches (or nullptr)   // integral -> ... -> sorted kps + descriptors in c->det[];
                                                                        // gate_min_features >= 0: also evaluate VO:556 into d_counts[CN_NQA]
uvo_status surf_hessian_layer_debug(Ctx* c, int octave, int layer, float* det, float* trace);
// match.hip
uvo_status match_knn2(Ctx* c, const float* d_q, const int* d_nq, int nq_max, const float* d_t, const int* d_nt, int nt_max);
uvo_status match_knn2_hamming(Ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int bytes);
uvo_status match_knn2_two(Ctx* c, const float* d_q0, const int* d_nq0, const float* d_t0, const int* d_nt0,
                          const float* d_q1, const int* d_nq1, const float* d_t1, const int* d_nt1, int n_max);
uvo_status match_ratio_compact2(Ctx* c, float ratio, const int* d_nq0, uvo_dmatch* d_out0, int* d_nout0, const GateArgs& g0,
                                const int* d_nq1, uvo_dmatch* d_out1, int* d_nout1, const GateArgs& g1, int n_max, int out_cap);
uvo_status match_two_pairs(Ctx* a, Ctx* b, const float* d_prev_desc, const int* d_prev_n, const int* d_prev_as_n, int curr_a, int curr_b,
                           float ratio, int min_features);
uvo_status match_ratio_compact(Ctx* c, const int* d_nq, int nq_max, float ratio, uvo_dmatch* d_out, int* d_nout, int out_cap,
                               const GateArgs* gate = nullptr);
// pose.hip
uvo_status pose_triangulate(Ctx* c, const double* P1, const double* P2, const int* d_n, int n_max);
uvo_status pose_triangulate_extract3d(Ctx* c, int slot, const double* P1, const double* P2, const double* R1, const double* t1,
                                      const double* R2, const double* t2, const double* K1, const double* K2, const int* d_n, int n_max,
                                      int* counts_host = nullptr,      // counts_host: pinned mirror of d_counts written by the last kernel
                                      Ctx* c2 = nullptr, int* counts_host2 = nullptr);   // c2: a second lane's pair in the same launches
// The tail of the stereo loop's stage A in one launch: lane `a` gathers its "after stereo match" set (buffer curr) and triangulates
// its rows, and runs extract_3Dpoints on the rows of lane p's set (buffer prev) that its triangular matches select (VO:569-579, 631-632).
uvo_status pose_stereo_tail(Ctx* a, Ctx* p, int prev, int curr, int slot, const double* P1, const double* P2, const double* R1, const double* t1,
                            const double* R2, const double* t2, const double* K1, const double* K2, int* counts_host);
// the rows of lane c's set `buf`, triangulated by a launch of their own (a set gathered by other means)
uvo_status pose_as_triangulate(Ctx* c, hipStream_t st, int buf, const int* d_n, int n_max, const double* P1, const double* P2, const double* R1, const double* t1,
                               const double* R2, const double* t2, const double* K1, const double* K2);
uvo_status pose_triangulate_extract3d_pick(Ctx* c, const double* P1, const double* P2x4, const double* Rx4, const double* tx4, const double* K,
                                           const int* d_best, const uvo_point2f* d_in1, const uvo_point2f* d_in2, const int* d_n, int n_max);
uvo_status pose_extract3d(Ctx* c, int slot, const double* R1, const double* t1, const double* R2, const double* t2,
                          const double* K1, const double* K2, const int* d_n, int n_max);
uvo_status pose_reproject_errors(Ctx* c, const double* world, int n, const double* R, const double* t, const double* K,
                                 const uvo_point2f* img, double* err);
struct PnpResult { uvo_status st; int wrote, ok, ninl; double rvec[3], tvec[3]; };
uvo_status pose_pnp_spec_launch(Ctx* c, hipStream_t st, const double* K, int iters, float reprojectionError, double confidence, int min3d);
bool pose_pnp_spec_accept(Ctx* c, int G, int iters, double confidence, PnpResult* r);
uvo_status pose_pnp_ransac_batch(Ctx* m, int n, Ctx* const* lanes, const int* G, const double* K, int iters, float reproj, double conf,
                                 PnpResult* res);
uvo_status pose_pnp_ransac(Ctx* c, int slot, int G, const double* K, int iters, float reproj, double conf,
                           double* rvec, double* tvec, int* n_inliers, int* ok);
int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters);
// preproc.hip
void pre_ws_free(Ctx* c);
uvo_status pre_get_image(Ctx* c, const uint8_t* rgb, int w, int h, int stride, int mem, const double* K, const double* dist4, const double* newK,
                         int desired_width, int clahe_on, int clip_limit, const uint8_t** d_out, int* out_w, int* out_h);
// codec.hip
void codec_ws_free(Ctx* c);
uvo_status codec_decode(Ctx* c, const uint8_t* data, size_t n, int bayer, const uint8_t** d_out, int* w, int* h, int* channels);
uvo_status codec_peek(Ctx* c, const uint8_t* data, size_t n, int bayer, int* w, int* h, int* channels);
uvo_status codec_bayer(Ctx* c, const uint8_t* bayer, int w, int h, int stride, int mem, const uint8_t** d_out);
// sift.hip
void sift_ws_free(Ctx* c);
// orb.hip
void orb_ws_free(Ctx* c);
uvo_status orb_configure(Ctx* c, int nfeatures, float scaleFactor, int nlevels, int edgeThreshold, int patchSize, int fastThreshold);
uvo_status orb_set_pattern(Ctx* c, const int* pattern);
uvo_status orb_detect(Ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, uvo_keypoint* kps, uint8_t* desc, int cap, int* n_out);
uvo_status orb_level_plane(Ctx* c, int level, int what, uint8_t* out, int cap_bytes, int* ow, int* oh);
// akaze.hip
void akaze_ws_free(Ctx* c);
uvo_status akaze_detect(Ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, uvo_keypoint* kps, uint8_t* desc, int cap, int* n_out);
uvo_status akaze_plane(Ctx* c, int level, int what, float* out, int cap_floats, int* ow, int* oh);
uvo_status sift_detect(Ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, int nfeatures, int nL, double contrastThreshold,
                       double edgeThreshold, double sigma, uvo_keypoint* kps, float* desc, int cap, int* n_out);
uvo_status sift_layer(Ctx* c, int octave, int layer, int dog, float* out, int cap_floats, int* ow, int* oh);
uvo_status sift_prepare_lane(Ctx* c, int w, int h, int nimg);
uvo_status sift_detect_lane(Ctx* c, int nimg, int gate_min_features);      // SIFT in place of surf_detect inside the fused steps
// mono.hip
void mono_ws_free(Ctx* c);
uvo_status mono_find_essential(Ctx* c, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K, int method,
                               double prob, double threshold, int maxIters, double* E, uint8_t* mask, int* ok);
uvo_status mono_recover_pose(Ctx* c, const double* E, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K,
                             double* R, double* t, uint8_t* mask, int* good);
uvo_status mono_find_homography(Ctx* c, const uvo_point2f* p1, const uvo_point2f* p2, int n, int method, double thr, int maxIters,
                                double confidence, double* H, uint8_t* mask, int* ok);
int decompose_homography_mat(const double* H, const double* K, double* Rs, double* ts, double* ns);
// the mono loop's pose stage on the matched points left on the device (mono.hip, "RESIDENT ON THE DEVICE")
struct MonoResident { int ok, n_in, valid_inliers, good, G, n_front; double E[9], R[9], t[3]; const uint8_t* mask; const double* zs; };   // mask, zs: pinned, valid until the lane's next frame
uvo_status mono_prep_launch(Ctx* c, hipStream_t st, const double* K, int distance);
int mono_prep_method(Ctx* c, int M, const uvo_point2f** k1, const uvo_point2f** k2);      // select_estimation_method's result (1 essential, 0 homography, -1 not reported) + the pinned point mirrors
uvo_status mono_essential_resident(Ctx* c, int n, const double* K, int method, double prob, double threshold, int maxIters, MonoResident* out);
uvo_status mono_recover_pose_homography(Ctx* c, const double* H, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K,
                                        double HOMOGRAPHY_DISTANCE, double* R, double* t, int* max_good);
void projection_matrix(const double* R, const double* t, const double* K, double* P);

}  // namespace uvo
