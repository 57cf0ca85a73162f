// Host-side ICP driver for the MI355X backend: the control flow of
// t::pipelines::registration::MultiScaleICP (cpp/open3d/t/pipelines/
// registration/Registration.cpp:24-62 ComputeRegistrationResult, :221-273
// pyramid, :275-360 DoSingleScaleICPIterations, :362-444 MultiScaleICP) with
// TransformationEstimationPointToPlane (TransformationEstimation.cpp:196-227),
// re-cut for the GPU:
//
//   reference, per iteration          here, per iteration
//   ------------------------------    -----------------------------------------
//   HybridSearch kernel               one fused kernel: search + sum d2 + count
//   counts.Sum  -> D2H sync             + Jacobian/29-sum accumulation
//   dist.Sum    -> D2H sync           sums posted to a host mailbox by the final
//                                     reduction kernel (no copy / sync call)
//   29-sum kernel -> D2H sync         [optional cross-GPU all-reduce hook]
//   6x6 solve on host (F64)           6x6 solve on host (F64), same arithmetic
//   4x4 upload + transform kernel     the transform rides in the NEXT search
//                                     launch (matrix by value, points moved
//                                     in place before they are searched)
//
// The source cloud is transformed incrementally in its own dtype every
// iteration exactly like the reference (Registration.cpp:322), not re-derived
// from the cumulative transform, so float rounding accumulates the same way.
//
// Other estimators (o3dmi_registration_multiscale_icp_ex): the same fused
// search launch in its point-to-point form (correspondences + raw moments),
// then per estimator: point-to-point -> R, t from the moments on the host;
// symmetric / coloured -> a second launch that gathers by correspondence and
// accumulates their 29 sums (two mailbox waits per iteration). Also
// EvaluateRegistration and GetInformationMatrix (Registration.cpp:64-91,
// 446-486) at the end of this file.

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <string>
#include <utility>
#include <vector>

#include "../common.h"
#include "../mailbox.h"
#include "../collectives.h"
#include "../vds.h"
#include "o3d_mi355x_host.h"

extern "C" int o3dmi_icp_search_accumulate_post(
        const o3dmi_nns_t* nns, const void* src_dev,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream);

extern "C" int o3dmi_nns_set_normals(o3dmi_nns_t* nns, const void* normals_dev,
                                     o3dmi_stream_t stream);

extern "C" int o3dmi_icp_colored_accumulate_post(
        const void* src_dev, const void* src_colors_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const void* tgt_colors_dev,
        const void* tgt_color_gradients_dev, const int64_t* corr_dev, int64_t n,
        int dtype, double lambda_geometric, int robust_kernel,
        double scaling_parameter, double shape_parameter, double* sums29_dev,
        double* partials_dev, double* mail_data, int* mail_flag, int mail_seq,
        o3dmi_stream_t stream);

extern "C" int o3dmi_icp_symmetric_accumulate_post(
        const void* src_dev, const void* src_normals_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const int64_t* corr_dev, int64_t n,
        int dtype, const double* source_mean3, const double* target_mean3,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        double* sums29_dev, double* partials_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream);

extern "C" int o3dmi_internal_nns_destroy_completed(o3dmi_nns_t* nns);
extern "C" int o3dmi_internal_nns_create_with_normals(
        const void* points_dev, const void* normals_dev, int64_t n, int dtype,
        double radius, o3dmi_stream_t stream, o3dmi_nns_t** out);
extern "C" int o3dmi_internal_nns_create_small_deferred(
        const void* points_dev, const void* normals_dev, const int* n_dev,
        int dtype, double radius, o3dmi_stream_t stream, o3dmi_nns_t** out);
extern "C" int o3dmi_internal_nns_adopt_count(o3dmi_nns_t* nns, int64_t n);
extern "C" int o3dmi_internal_nns_create_many(
        int count, const void* const* points_dev,
        const void* const* normals_dev, const int64_t* n, int dtype,
        const double* radius, o3dmi_stream_t stream, o3dmi_nns_t** out);
extern "C" int o3dmi_internal_icp_transform_search_accumulate(
        const o3dmi_nns_t* nns, void* src_dev, const double* transformation,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream);
extern "C" int o3dmi_internal_sums_tail(double* sums32_dev, double t29,
                                        double t30, double t31,
                                        o3dmi_stream_t stream);
extern "C" int o3dmi_internal_sums_post(const double* sums32_dev,
                                        double* mail_data, int* mail_flag,
                                        int seq, o3dmi_stream_t stream);

using namespace o3dmi;

// Device-side all-reduce hook of the calling host thread (one rank = one
// process, or one thread per device): see o3dmi_set_device_allreduce.
static thread_local o3dmi_allreduce_device_t g_dev_allreduce = nullptr;
static thread_local void* g_dev_allreduce_user = nullptr;

extern "C" int o3dmi_set_device_allreduce(o3dmi_allreduce_device_t fn,
                                          void* user) {
    g_dev_allreduce = fn;
    g_dev_allreduce_user = user;
    return O3DMI_OK;
}

// Device-resident cloud sizes for the NEXT driver call of this host thread
// (o3dmi_registration_set_device_counts).
static thread_local const int32_t* g_ns_dev = nullptr;
static thread_local const int32_t* g_nt_dev = nullptr;

extern "C" int o3dmi_registration_set_device_counts(const int32_t* ns_dev,
                                                    const int32_t* nt_dev) {
    g_ns_dev = ns_dev;
    g_nt_dev = nt_dev;
    return O3DMI_OK;
}

// With a communicator installed (o3dmi_set_comm): who shards the source cloud.
// 0: the caller -- it passes ITS shard (the semantics of the two hooks);
// 1: the driver -- every rank passes the WHOLE source, the pyramid is built
//    from it on every rank (so it is the unsharded run's pyramid, level for
//    level), and each rank searches / accumulates its contiguous slice of
//    every level.
static thread_local int g_level_sharding = 0;

extern "C" int o3dmi_set_icp_level_sharding(int on) {
    g_level_sharding = on ? 1 : 0;
    return O3DMI_OK;
}

namespace {

void Eye4(double* T) {
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

// update.Matmul(transformation), Registration.cpp:319 (host F64).
void Matmul4(const double* A, const double* B, double* C) {
    double R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            R[i * 4 + j] = s;
        }
    std::memcpy(C, R, sizeof(R));
}

struct DeviceBuffer {
    void* p = nullptr;
    // The driver drains the stream before its buffers go out of scope
    // (SyncOnExit below).
    ~DeviceBuffer() { PoolFree(p); }
    int Alloc(size_t bytes) {
        PoolFree(p);
        p = nullptr;
        return PoolAlloc(&p, bytes ? bytes : 1);
    }
};

struct Level {
    // source: positions, normals (symmetric), colours (coloured);
    // target: positions, normals, colours, colour gradients
    DeviceBuffer src, srcn, srcc, tgt, nrm, tgtc, tgtg;
    int64_t ns = 0, nt = 0;
    const void* tgt_ptr = nullptr;  // may alias the caller's buffers
    const void* nrm_ptr = nullptr;
    const void* tgtc_ptr = nullptr;
    const void* tgtg_ptr = nullptr;
};

// One pyramid level without a host wait (vds.h). PointCloud::VoxelDownSample
// averages every attribute; the kernel takes positions + one attribute, so
// further attributes go through it again (the voxel order, first occurrence,
// is the same every time, and so are the positions and the count).
// next_voxel > 0: the chain's next call down-samples out_pos by that size;
// from_previous: pos is the out_pos of the chain's previous call (vds.h: a
// level with a single pass then carries the next level's hash insert).
int DownSampleAttrsAsync(const void* pos, int64_t n_max, const int* n_dev,
                         int dtype, double voxel, void* out_pos, int* m_dev,
                         int* err_dev, std::vector<void*>& scratch,
                         hipStream_t cs, int chain,
                         std::initializer_list<std::pair<const void*, void*>>
                                 attrs,
                         double next_voxel = 0, bool from_previous = false,
                         VdsLevelJob* defer = nullptr) {
    if (voxel <= 0) {
        SetLastError("voxel_size must be positive.");
        return O3DMI_ERR_INVALID_ARG;
    }
    int passes = 0;
    for (const auto& a : attrs) passes += a.first ? 1 : 0;
    const bool single = passes <= 1;
    if (defer && single) {
        // one pass: the caller launches it together with the other cloud's
        // level (VdsPairAsync)
        defer->pos = pos;
        defer->n_max = n_max;
        defer->n_dev = n_dev;
        defer->voxel_size = voxel;
        defer->out_pos = out_pos;
        defer->m_dev = m_dev;
        defer->err_dev = err_dev;
        defer->chain = chain;
        defer->next_voxel_size = next_voxel;
        defer->from_previous = from_previous;
        for (const auto& a : attrs)
            if (a.first) {
                defer->attr = a.first;
                defer->out_attr = a.second;
            }
        return O3DMI_OK;
    }
    bool done = false;
    for (const auto& a : attrs) {
        if (!a.first) continue;
        int st = VdsAsync(pos, a.first, n_max, n_dev, dtype, voxel, out_pos,
                          a.second, m_dev, err_dev, scratch, cs, chain,
                          single ? next_voxel : 0.0, single && from_previous);
        if (st) return st;
        done = true;
    }
    if (!done)
        return VdsAsync(pos, nullptr, n_max, n_dev, dtype, voxel, out_pos,
                        nullptr, m_dev, err_dev, scratch, cs, chain,
                        next_voxel, from_previous);
    return O3DMI_OK;
}

// Device-side level counts of one cloud's pyramid: [level] voxel counts, then
// one word of error flags, in a persistent buffer per host thread, device and
// chain (zero when allocated; the posting launch re-zeroes the error word).
// Read once, at the end of the chain, through the chain's host mailbox: no
// clearing fill, no copy, no stream synchronisation per call.
constexpr int kMaxDevices = 64;
int CurrentDevice() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return -1;
    return d;
}
constexpr int kMaxScales = 30;
static_assert(kMaxScales + 1 <= kCountsErr && kMaxScales + 1 <= 32,
              "level counts below the error word; a post carries <= 32 values");
struct ChainCounts {
    int* dev = nullptr;
    int levels = 0;
    int chain = 0;
    std::vector<void*> scratch;  // pooled scratch of the level launches
    // The counts, the error word and the VoxelDownSample workspaces of a
    // chain are cleaned by the chain's own last launches (the posting launch,
    // the levels' reduce launches). `open`: a chain was started on this
    // thread / device / chain slot and its counts were never waited for -- an
    // error return somewhere between its first launch and Wait. The next
    // chain then starts from re-zeroed counts and fresh workspaces instead of
    // inheriting a stale error bit or stale table entries.
    static bool& Open(int d, int chain_id) {
        static thread_local bool open[kMaxDevices][2] = {};
        return open[d][chain_id];
    }
    int device = -1;
    int Init(int n_levels, int chain_id, hipStream_t cs) {
        static thread_local int* bufs[kMaxDevices][2] = {};
        O3DMI_REQUIRE(n_levels >= 1 && n_levels <= kMaxScales,
                      "too many scales");
        const int d = CurrentDevice();
        O3DMI_REQUIRE(d >= 0 && (chain_id == 0 || chain_id == 1),
                      "bad device / chain");
        int*& b = bufs[d][chain_id];
        const bool fresh = !b;
        if (!b)
            O3DMI_HIP_CHECK(hipMalloc((void**)&b,
                                      sizeof(int) * 2 * kCountsKeep));
        if (fresh || Open(d, chain_id)) {
            if (!fresh) {
                O3DMI_HIP_CHECK(hipDeviceSynchronize());
                VdsChainInvalidate(chain_id);
            }
            O3DMI_HIP_CHECK(hipMemsetAsync(b, 0,
                                           sizeof(int) * 2 * kCountsKeep, cs));
        }
        Open(d, chain_id) = true;
        dev = b;
        levels = n_levels;
        chain = chain_id;
        device = d;
        return O3DMI_OK;
    }
    int* Count(int level) { return dev + level; }
    // (valid behind the posting launch, PostCountsPairAsync)
    const int* KeptCount(int level) const { return dev + kCountsKeep + level; }
    int* Err() { return dev + kCountsErr; }  // (a post's value `levels`)
    // Post: the counts leave for the chain's mailbox behind the chain's
    // launches; Wait: for that, and returns them. (Both chains post before
    // either is waited for: one host round trip, not two.)
    int posted_seq = 0;
    int Post(hipStream_t cs) {
        Mailbox* mb = ThreadMailbox(1 + chain);
        O3DMI_REQUIRE(mb != nullptr, "host mailbox allocation failed");
        posted_seq = ++mb->seq;
        return PostCountsAsync(dev, levels + 1, mb->data, mb->flag, posted_seq,
                               cs);
    }
    // The post carried by the chain's last level launch (vds.h VdsPost): the
    // request to hand to that level's job, then Posted() if it was taken up.
    bool sealed = false;
    int offered_seq = 0;
    VdsPost OfferPost() {
        VdsPost p;
        Mailbox* mb = ThreadMailbox(1 + chain);
        if (!mb) return p;
        offered_seq = ++mb->seq;
        p.counts = dev;
        p.n = levels + 1;
        p.mail_data = mb->data;
        p.mail_flag = mb->flag;
        p.mail_seq = offered_seq;
        return p;
    }
    void Posted() {
        posted_seq = offered_seq;
        sealed = true;
    }
    // both chains were built in the same launches: one posting launch
    static int PostPair(ChainCounts& a, ChainCounts& b, hipStream_t cs) {
        Mailbox* ma = ThreadMailbox(1 + a.chain);
        Mailbox* mb = ThreadMailbox(1 + b.chain);
        O3DMI_REQUIRE(ma != nullptr && mb != nullptr && a.levels == b.levels,
                      "host mailbox allocation failed");
        a.posted_seq = ++ma->seq;
        b.posted_seq = ++mb->seq;
        return PostCountsPairAsync(a.dev, ma->data, ma->flag, a.posted_seq,
                                   b.dev, mb->data, mb->flag, b.posted_seq,
                                   a.levels + 1, cs);
    }
    int Fetch(std::vector<int>& out, hipStream_t cs) {
        int st = Post(cs);
        if (st) return st;
        return Wait(out, cs);
    }
    int Wait(std::vector<int>& out, hipStream_t cs) {
        out.assign((size_t)levels + 1, 0);
        Mailbox* mb = ThreadMailbox(1 + chain);
        O3DMI_REQUIRE(mb != nullptr && posted_seq != 0, "counts not posted");
        const int seq = posted_seq;
        posted_seq = 0;
        double sealed32[32];
        hipError_t e = sealed ? MailboxWaitSealed(mb, seq, cs, sealed32)
                              : MailboxWait(mb, seq, cs);
        const bool was_sealed = sealed;
        sealed = false;
        // the posting launch was the chain's last: its stream has drained
        // (no hipStreamSynchronize, 16 us on an idle stream). (A sealed post
        // comes from the last level's launch while it runs: only chains of
        // the tiled form, which hold no pooled scratch.)
        if (e == hipSuccess && !was_sealed) {
            for (void* p : scratch) PoolFree(p);
            scratch.clear();
        }
        Release(cs);
        if (e != hipSuccess) {
            SetLastError(std::string("pyramid read-back: ") +
                         hipGetErrorString(e));
            return O3DMI_ERR_HIP;
        }
        // the posting launch has run: counts and error word are zero again,
        // every level's last launch has cleaned its workspace
        if (device >= 0) Open(device, chain) = false;
        for (int k = 0; k <= levels; ++k)
            out[(size_t)k] = (int)(was_sealed ? sealed32[k] : mb->data[k]);
        if (out[(size_t)levels] & kErrKeyRange) {
            SetLastError("VoxelDownSample: voxel coordinate outside +-2^20");
            return O3DMI_ERR_KEY_RANGE;
        }
        return O3DMI_OK;
    }
    void Release(hipStream_t cs) {
        if (scratch.empty()) return;
        (void)hipStreamSynchronize(cs);  // pooled blocks: stream drained
        for (void* p : scratch) PoolFree(p);
        scratch.clear();
    }
};

// One side stream and event per host thread AND device for the overlapped
// pyramid build (a thread may switch devices between calls: per-device pool,
// VoxelBlockGrid::To(device)).
hipStream_t SideStream() {
    static thread_local hipStream_t side[kMaxDevices] = {};
    const int d = CurrentDevice();
    if (d < 0) return nullptr;
    if (!side[d] &&
        hipStreamCreateWithFlags(&side[d], hipStreamNonBlocking) != hipSuccess)
        side[d] = nullptr;
    return side[d];
}

hipEvent_t SideEvent() {
    static thread_local hipEvent_t ev[kMaxDevices] = {};
    const int d = CurrentDevice();
    if (d < 0) return nullptr;
    if (!ev[d] &&
        hipEventCreateWithFlags(&ev[d], hipEventDisableTiming) != hipSuccess)
        ev[d] = nullptr;
    return ev[d];
}

// `completed`: set by the owner once every kernel that used the index is
// known to have finished (its results were read on the host); the destructor
// then skips the device-wide wait of the public o3dmi_nns_destroy.
struct NnsGuard {
    o3dmi_nns_t* nns = nullptr;
    bool completed = false;
    ~NnsGuard() {
        if (completed) o3dmi_internal_nns_destroy_completed(nns);
        else o3dmi_nns_destroy(nns);
    }
};

struct SearchResult {
    double fitness = 0, inlier_rmse = 0;
    double sums[32];
};

}  // namespace

extern "C" int o3dmi_registration_multiscale_icp(
        const void* source_dev, int64_t ns, const void* target_dev,
        const void* target_normals_dev, int64_t nt, int dtype, int num_scales,
        const double* voxel_sizes, const o3dmi_icp_criteria_t* criterias,
        const double* max_dists, const double* init, int robust_kernel,
        double scaling_parameter, double shape_parameter,
        o3dmi_icp_callback_t callback, void* callback_user,
        o3dmi_allreduce_sum_t allreduce, void* allreduce_user,
        int64_t* correspondences_dev, o3dmi_registration_result_t* result,
        o3dmi_stream_t stream) {
    return o3dmi_registration_multiscale_icp_ex(
            source_dev, ns, target_dev, target_normals_dev, nt, dtype,
            num_scales, voxel_sizes, criterias, max_dists, init,
            O3DMI_ICP_POINT_TO_PLANE, nullptr, robust_kernel, scaling_parameter,
            shape_parameter, callback, callback_user, allreduce, allreduce_user,
            correspondences_dev, result, stream);
}

extern "C" int o3dmi_registration_multiscale_icp_ex(
        const void* source_dev, int64_t ns, const void* target_dev,
        const void* target_normals_dev, int64_t nt, int dtype, int num_scales,
        const double* voxel_sizes, const o3dmi_icp_criteria_t* criterias,
        const double* max_dists, const double* init, int estimation,
        const o3dmi_icp_attributes_t* attrs, int robust_kernel,
        double scaling_parameter, double shape_parameter,
        o3dmi_icp_callback_t callback, void* callback_user,
        o3dmi_allreduce_sum_t allreduce, void* allreduce_user,
        int64_t* correspondences_dev, o3dmi_registration_result_t* result,
        o3dmi_stream_t stream) {
    // sizes that live on the device (consumed by this call, whatever it does)
    const int32_t* ns_dev = g_ns_dev;
    const int32_t* nt_dev = g_nt_dev;
    g_ns_dev = g_nt_dev = nullptr;
    // AssertInputMultiScaleICP, Registration.cpp:119-219.
    O3DMI_REQUIRE(result != nullptr, "result is null");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "Only Float32 and Float64 point clouds are supported.");
    O3DMI_REQUIRE(source_dev && target_dev && ns > 0 && nt > 0,
                  "Source and/or Target pointcloud is empty.");
    O3DMI_REQUIRE(estimation >= O3DMI_ICP_POINT_TO_PLANE &&
                          estimation <= O3DMI_ICP_COLORED,
                  "estimation must be point-to-plane, point-to-point, "
                  "symmetric or colored");
    const bool p2plane = estimation == O3DMI_ICP_POINT_TO_PLANE;
    const bool symmetric = estimation == O3DMI_ICP_SYMMETRIC;
    const bool colored = estimation == O3DMI_ICP_COLORED;
    const bool need_tn = p2plane || symmetric || colored;
    if (!need_tn) target_normals_dev = nullptr;
    const void* source_normals_dev =
            symmetric && attrs ? attrs->source_normals : nullptr;
    const void* source_colors_dev =
            colored && attrs ? attrs->source_colors : nullptr;
    const void* target_colors_dev =
            colored && attrs ? attrs->target_colors : nullptr;
    const void* target_gradients_dev =
            colored && attrs ? attrs->target_color_gradients : nullptr;
    double lambda_geometric = colored && attrs ? attrs->lambda_geometric : 0.968;
    // TransformationEstimationForColoredICP ctor, TransformationEstimation.h:
    // 293-299
    if (!(lambda_geometric >= 0 && lambda_geometric <= 1.0))
        lambda_geometric = 0.968;
    O3DMI_REQUIRE(!(p2plane || colored) || target_normals_dev != nullptr,
                  "Target pointcloud missing normals attribute.");
    O3DMI_REQUIRE(!symmetric || (source_normals_dev && target_normals_dev),
                  "SymmetricICP requires both source and target to have "
                  "normals.");
    O3DMI_REQUIRE(!colored || (source_colors_dev && target_colors_dev),
                  "Source and/or Target pointcloud missing colors attribute.");
    // the fused search kernel accumulates the point-to-plane terms itself;
    // the other estimators take the point-to-point moments from it
    const int search_mode = p2plane ? 0 : 1;
    O3DMI_REQUIRE(num_scales > 0 && voxel_sizes && criterias && max_dists,
                  "Size of criterias, voxel_size, max_correspondence_distances "
                  "vectors must be same.");
    for (int i = 0; i < num_scales; ++i) {
        O3DMI_REQUIRE(max_dists[i] > 0,
                      "max_correspondence_distance must be positive");
        if (i + 1 < num_scales)
            O3DMI_REQUIRE(voxel_sizes[i + 1] <= 0 ||
                                  voxel_sizes[i] > voxel_sizes[i + 1],
                          "Decreasing order of voxel_sizes is required.");
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == O3DMI_F64 ? 8 : 4;

    // O3DMI_ICP_TIMING=2: wall time of the whole call including the exit path
    // (declared first: destroyed last), without the per-phase waits of =1.
    struct ExitTimer {
        bool on = false;
        double t0 = 0;
        std::string marks;
        void Mark(const char* what) {
            if (!on) return;
            char buf[64];
            std::snprintf(buf, sizeof(buf), " %s@%.0f", what, Now() - t0);
            marks += buf;
        }
        static double Now() {
            return std::chrono::duration<double, std::micro>(
                           std::chrono::steady_clock::now().time_since_epoch())
                    .count();
        }
        ~ExitTimer() {
            if (on)
                std::fprintf(stderr, "[o3dmi] icp: whole call %.0f us;%s\n",
                             Now() - t0, marks.c_str());
        }
    } exit_timer;
    if (const char* e = std::getenv("O3DMI_ICP_TIMING")) {
        exit_timer.on = true;
        exit_timer.t0 = ExitTimer::Now();
        (void)e;
    }
    // InitializePointCloudPyramidForMultiScaleICP, Registration.cpp:221-273.
    std::vector<Level> pyr((size_t)num_scales);
    // Declared after the pyramid so that it runs first on every exit path:
    // pooled buffers may only be released once the streams have drained.
    // `drained`: set once the host has SEEN the last launch of the call finish
    // (the mailbox of the final evaluation): the caller's stream is in order,
    // and everything the side stream did was waited for by a later launch on
    // the caller's stream, so both are idle -- and hipStreamSynchronize costs
    // 16 us per stream even then (measured: 32 us of every tracked frame).
    struct SyncOnExit {
        hipStream_t s, side;
        bool drained;
        ~SyncOnExit() {
            if (drained) return;
            (void)hipStreamSynchronize(s);
            if (side && side != s) (void)hipStreamSynchronize(side);
        }
    } sync_on_exit{s, nullptr, false};
    const int last = num_scales - 1;
    int st;
    // One index per scale (target_nns.HybridIndex(max_correspondence_distance),
    // Registration.cpp:406-412), built right behind the pyramids.
    std::vector<NnsGuard> guards((size_t)num_scales);
    const double t_entry =
            std::chrono::duration<double, std::micro>(
                    std::chrono::steady_clock::now().time_since_epoch())
                    .count();
    // The source pyramid and the target pyramid are independent chains of
    // VoxelDownSample levels, each a string of small launches whose sizes stay
    // on the device (the voxel count of one level is the point count of the
    // next; level buffers are sized by the input cloud) -- latency, not
    // throughput. One host thread issues both, level by level, the source
    // chain on the caller's stream and the target chain on a side stream, so
    // the GPU works on the two chains at once; the counts of both are read
    // back once, at the end. (Round 2 first ran the target chain from a helper
    // thread: the thread start and its first HIP call cost more than issuing
    // the second chain's launches from here.)
    auto clone = [&](DeviceBuffer& dst, const void* src, int64_t n,
                     hipStream_t cs) -> int {
        int e = dst.Alloc((size_t)n * 3 * esz);
        if (e) return e;
        O3DMI_HIP_CHECK(hipMemcpyAsync(dst.p, src, (size_t)n * 3 * esz,
                                       hipMemcpyDeviceToDevice, cs));
        return O3DMI_OK;
    };
    const bool finest_is_input = voxel_sizes[last] <= 0;
    if (finest_is_input && (ns_dev || nt_dev)) {
        // without a down-sampled finest level the sizes size the searches:
        // they have to come to the host (one small copy and wait)
        int32_t host_n[2] = {(int32_t)ns, (int32_t)nt};
        if (ns_dev)
            O3DMI_HIP_CHECK(hipMemcpyAsync(&host_n[0], ns_dev, sizeof(int32_t),
                                           hipMemcpyDeviceToHost, s));
        if (nt_dev)
            O3DMI_HIP_CHECK(hipMemcpyAsync(&host_n[1], nt_dev, sizeof(int32_t),
                                           hipMemcpyDeviceToHost, s));
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        O3DMI_REQUIRE(host_n[0] > 0 && host_n[0] <= ns && host_n[1] > 0 &&
                              host_n[1] <= nt,
                      "Source and/or Target pointcloud is empty.");
        ns = host_n[0];
        nt = host_n[1];
        ns_dev = nt_dev = nullptr;
    }
    ChainCounts scc, tcc;
    struct ChainGuard {
        ChainCounts& c;
        hipStream_t s;
        ~ChainGuard() { c.Release(s); }
    };
    auto source_level = [&](int k, hipStream_t cs, VdsLevelJob* job) -> int {
        int e;
        Level& L = pyr[(size_t)k];
        if (k == last && finest_is_input) {
            L.ns = ns;
            // the source is moved in place every iteration: private copies
            if ((e = clone(L.src, source_dev, ns, cs))) return e;
            if (symmetric && (e = clone(L.srcn, source_normals_dev, ns, cs)))
                return e;
            if (colored && (e = clone(L.srcc, source_colors_dev, ns, cs)))
                return e;
            return O3DMI_OK;
        }
        if ((e = L.src.Alloc((size_t)ns * 3 * esz))) return e;
        if (symmetric && (e = L.srcn.Alloc((size_t)ns * 3 * esz))) return e;
        if (colored && (e = L.srcc.Alloc((size_t)ns * 3 * esz))) return e;
        // (the coarser level is built from this one's output: vds.h)
        const double next_vs = k > 0 ? voxel_sizes[k - 1] : 0.0;
        if (k == last)
            return DownSampleAttrsAsync(source_dev, ns, (const int*)ns_dev,
                                        dtype, voxel_sizes[k], L.src.p,
                                        scc.Count(k),
                                        scc.Err(), scc.scratch, cs, 0,
                                        {{source_normals_dev, L.srcn.p},
                                         {source_colors_dev, L.srcc.p}},
                                        next_vs, false, job);
        Level& F = pyr[(size_t)k + 1];
        const bool f_host = k + 1 == last && finest_is_input;
        return DownSampleAttrsAsync(F.src.p, ns,
                                    f_host ? nullptr : scc.Count(k + 1), dtype,
                                    voxel_sizes[k], L.src.p, scc.Count(k),
                                    scc.Err(), scc.scratch, cs, 0,
                                    {{F.srcn.p, L.srcn.p},
                                     {F.srcc.p, L.srcc.p}},
                                    next_vs, !f_host, job);
    };
    bool finest_on_host = finest_is_input;
    auto target_level = [&](int k, hipStream_t cs, VdsLevelJob* job) -> int {
        o3dmi_stream_t cstream = (o3dmi_stream_t)cs;
        int e;
        Level& L = pyr[(size_t)k];
        if (k == last) {
            if (finest_is_input) {
                L.nt = nt;
                L.tgt_ptr = target_dev;
                L.nrm_ptr = target_normals_dev;
                L.tgtc_ptr = target_colors_dev;
                L.tgtg_ptr = target_gradients_dev;
            } else {
                if ((e = L.tgt.Alloc((size_t)nt * 3 * esz))) return e;
                if (need_tn && (e = L.nrm.Alloc((size_t)nt * 3 * esz)))
                    return e;
                if (colored) {
                    if ((e = L.tgtc.Alloc((size_t)nt * 3 * esz))) return e;
                    if (target_gradients_dev &&
                        (e = L.tgtg.Alloc((size_t)nt * 3 * esz)))
                        return e;
                }
                e = DownSampleAttrsAsync(target_dev, nt, (const int*)nt_dev,
                                         dtype, voxel_sizes[k], L.tgt.p,
                                         tcc.Count(k), tcc.Err(), tcc.scratch,
                                         cs, 1,
                                         {{target_normals_dev, L.nrm.p},
                                          {target_colors_dev, L.tgtc.p},
                                          {target_gradients_dev, L.tgtg.p}},
                                         k > 0 ? voxel_sizes[k - 1] : 0.0,
                                         false, job);
                if (e) return e;
                L.tgt_ptr = L.tgt.p;
                L.nrm_ptr = L.nrm.p;  // stays NULL without normals
                L.tgtc_ptr = L.tgtc.p;
                L.tgtg_ptr = L.tgtg.p;
            }
            if (colored && !L.tgtg_ptr) {
                // Registration.cpp:243-262: EstimateColorGradients(30, radius)
                // on the finest level of the target pyramid. The operator
                // needs the level's size on the host: this (rare) path waits.
                if (!finest_is_input) {
                    int host_n = 0;
                    O3DMI_HIP_CHECK(hipMemcpyAsync(&host_n, tcc.Count(k),
                                                   sizeof(int),
                                                   hipMemcpyDeviceToHost, cs));
                    O3DMI_HIP_CHECK(hipStreamSynchronize(cs));
                    L.nt = host_n;
                    finest_on_host = true;
                }
                const double radius = voxel_sizes[k] <= 0 ? max_dists[k] * 2.0
                                                          : voxel_sizes[k] * 4.0;
                if ((e = L.tgtg.Alloc((size_t)nt * 3 * esz))) return e;
                e = o3dmi_pointcloud_estimate_color_gradients(
                        L.tgt_ptr, L.nrm_ptr, L.tgtc_ptr, L.nt, dtype, 30,
                        radius, L.tgtg.p, cstream);
                if (e) return e;
                L.tgtg_ptr = L.tgtg.p;
            }
            return O3DMI_OK;
        }
        Level& F = pyr[(size_t)k + 1];
        if ((e = L.tgt.Alloc((size_t)nt * 3 * esz))) return e;
        if (need_tn && (e = L.nrm.Alloc((size_t)nt * 3 * esz))) return e;
        if (colored) {
            if ((e = L.tgtc.Alloc((size_t)nt * 3 * esz))) return e;
            if ((e = L.tgtg.Alloc((size_t)nt * 3 * esz))) return e;
        }
        // the finest level's size is a host number when it is the input
        // itself (or was read back above): then n_max = that size
        const bool f_host = k + 1 == last && finest_on_host;
        e = DownSampleAttrsAsync(F.tgt_ptr, f_host ? F.nt : nt,
                                 f_host ? nullptr : tcc.Count(k + 1), dtype,
                                 voxel_sizes[k], L.tgt.p, tcc.Count(k),
                                 tcc.Err(), tcc.scratch, cs, 1,
                                 {{F.nrm_ptr, L.nrm.p},
                                  {F.tgtc_ptr, L.tgtc.p},
                                  {F.tgtg_ptr, L.tgtg.p}},
                                 k > 0 ? voxel_sizes[k - 1] : 0.0, !f_host,
                                 job);
        if (e) return e;
        L.tgt_ptr = L.tgt.p;
        L.nrm_ptr = L.nrm.p;
        L.tgtc_ptr = L.tgtc.p;
        L.tgtg_ptr = L.tgtg.p;
        return O3DMI_OK;
    };
    // Two streams unless there is nothing to overlap (a single level without
    // down-sampling is two copies).
    const bool overlap = num_scales > 1 || voxel_sizes[last] > 0;
    hipStream_t side = s;
    hipEvent_t ev = nullptr;
    if (overlap) {
        side = SideStream();
        ev = SideEvent();
        O3DMI_REQUIRE(side != nullptr && ev != nullptr,
                      "side stream creation failed");
        sync_on_exit.side = side;
        // the caller's clouds may still be in flight on its stream
        O3DMI_HIP_CHECK(hipEventRecord(ev, s));
        O3DMI_HIP_CHECK(hipStreamWaitEvent(side, ev, 0));
    }
    bool indices_on_side = false;
    int next_index = 1;  // first scale whose index is not issued yet
    {
        // Round 6: the two pyramids advance level by level in the SAME
        // launches on the caller's stream (VdsPairAsync: blockIdx.y = cloud)
        // -- 7 launches + one posting launch for a three-level pair of
        // pyramids instead of two chains of 8 on two streams. Coloured ICP
        // (three attribute passes per target level) keeps the two chains.
        static const bool unpaired = std::getenv("O3DMI_VDS_UNPAIRED") != nullptr;
        const bool paired = !colored && !unpaired;
        hipStream_t ts = paired ? s : side;
        static const bool no_folded_post =
                std::getenv("O3DMI_VDS_POST_LAUNCH") != nullptr;
        bool counts_posted = false;
        ChainGuard sg{scc, s}, tg{tcc, ts};
        if ((st = scc.Init(num_scales, 0, s))) return st;
        if ((st = tcc.Init(num_scales, 1, ts))) return st;
        for (int k = last; k >= 0; --k) {
            VdsLevelJob jobs[2];
            if ((st = target_level(k, ts, paired ? &jobs[1] : nullptr)))
                return st;
            if ((st = source_level(k, s, paired ? &jobs[0] : nullptr)))
                return st;
            if (!paired) continue;
            // (a level that is the input itself leaves its job empty)
            VdsLevelJob both[2];
            int n_jobs = 0;
            if (jobs[0].pos) both[n_jobs++] = jobs[0];
            if (jobs[1].pos) both[n_jobs++] = jobs[1];
            // the coarsest level's launch posts both chains' counts itself
            const bool offer = k == 0 && n_jobs == 2 && !no_folded_post;
            if (offer) {
                both[0].post = scc.OfferPost();
                both[1].post = tcc.OfferPost();
            }
            bool posted = false;
            if (n_jobs &&
                (st = VdsPairAsync(both, n_jobs, dtype, scc.scratch, s,
                                   &posted)))
                return st;
            if (posted) {
                scc.Posted();
                tcc.Posted();
                counts_posted = true;
            }
        }
        std::vector<int> counts;
        // The coarsest scale's index, queued behind the posting launch BEFORE
        // the sizes are read back (its one-workgroup build takes the target
        // level's size from the count the posting launch keeps): it runs while
        // the counts cross PCIe and the host gets ready to launch the first
        // search -- that search used to wait for count -> host -> allocation ->
        // build launch -> build (a 20 us hole in every tracked frame, r6a).
        NnsGuard early;  // (destroyed with a device-wide wait if not adopted)
        const bool early_build =
                paired && !(last == 0 && finest_is_input) && pyr[0].tgt_ptr;
        if (paired) {
            if (!counts_posted &&
                (st = ChainCounts::PostPair(scc, tcc, s)))
                return st;
            if (early_build &&
                (st = o3dmi_internal_nns_create_small_deferred(
                         pyr[0].tgt_ptr, p2plane ? pyr[0].nrm_ptr : nullptr,
                         tcc.KeptCount(0), dtype, max_dists[0],
                         (o3dmi_stream_t)s, &early.nns)))
                return st;
        } else {
            if ((st = tcc.Post(ts))) return st;
            if ((st = scc.Post(s))) return st;
        }
        if ((st = tcc.Wait(counts, ts))) return st;
        for (int k = 0; k < num_scales; ++k)
            if (!(k == last && finest_on_host))
                pyr[(size_t)k].nt = counts[(size_t)k];
        if ((st = scc.Wait(counts, s))) return st;
        for (int k = 0; k < num_scales; ++k)
            if (!(k == last && finest_is_input))
                pyr[(size_t)k].ns = counts[(size_t)k];
        // Indices: the first scale's on the caller's stream (its first search
        // follows at once). The others go to the side stream LATER, one per
        // search launch of the first scale, issued while the host would
        // otherwise spin on that launch's sums (build_next_index below):
        // issuing them here kept the host busy for ~60 us (eight launch
        // calls) before it got to the first search -- with the GPU idle
        // (tools/slam_timeline.py).
        if (early.nns &&
            o3dmi_internal_nns_adopt_count(early.nns, pyr[0].nt)) {
            guards[0].nns = early.nns;
            early.nns = nullptr;
        } else {
            const Level& L0 = pyr[0];
            if ((st = o3dmi_internal_nns_create_with_normals(
                         L0.tgt_ptr, p2plane ? L0.nrm_ptr : nullptr, L0.nt,
                         dtype, max_dists[0], (o3dmi_stream_t)s,
                         &guards[0].nns)))
                return st;
        }
    }

    // One more scale's index on the side stream (see above); the event the
    // second scale waits for is recorded behind the last one.
    int index_status = O3DMI_OK;
    // ALL the later scales' indices in the same four launches (round 6,
    // nns.hip BuildIndexMany; up to four per call), issued in the shadow of
    // the first search launch. One build per search launch (rounds 2-5) was a
    // fill + three launches + four pool allocations each: ~20 us of host time
    // behind an 11 us search -- two late hops per tracked frame.
    auto build_next_index = [&]() {
        while (next_index < num_scales && index_status == O3DMI_OK) {
            const void* pts[4];
            const void* nrm[4];
            int64_t nn[4];
            double rad[4];
            o3dmi_nns_t* made[4] = {};
            int cnt = 0;
            const int first = next_index;
            for (; cnt < 4 && next_index < num_scales; ++cnt, ++next_index) {
                const Level& Lk = pyr[(size_t)next_index];
                pts[cnt] = Lk.tgt_ptr;
                nrm[cnt] = p2plane ? Lk.nrm_ptr : nullptr;
                nn[cnt] = Lk.nt;
                rad[cnt] = max_dists[next_index];
            }
            index_status = o3dmi_internal_nns_create_many(
                    cnt, pts, nrm, nn, dtype, rad, (o3dmi_stream_t)side, made);
            for (int q = 0; q < cnt; ++q)
                guards[(size_t)(first + q)].nns = made[q];
        }
        if (index_status == O3DMI_OK && overlap && !indices_on_side &&
            num_scales > 1) {
            if (hipEventRecord(ev, side) != hipSuccess)
                index_status = O3DMI_ERR_HIP;
            else
                indices_on_side = true;
        }
    };
    auto ensure_indices = [&]() -> int {
        while (next_index < num_scales && index_status == O3DMI_OK)
            build_next_index();
        return index_status;
    };
    exit_timer.Mark("pyramid");
    const char* timing_env = std::getenv("O3DMI_ICP_TIMING");
    const bool timing = timing_env && std::atoi(timing_env) != 2;
    auto now = [] {
        return std::chrono::duration<double, std::micro>(
                       std::chrono::steady_clock::now().time_since_epoch())
                .count();
    };
    double t_mark = 0;
    if (timing) {
        (void)hipStreamSynchronize(s);
        t_mark = now();
        std::fprintf(stderr, "[o3dmi] icp: pyramid built in %.0f us\n",
                     t_mark - t_entry);
    }
    const double t_start = t_mark;

    // Per-iteration sums arrive through the thread's host mailbox: the final
    // reduction kernel writes them into host-mapped memory and bumps a
    // sequence word the host spins on (no copy / stream synchronise call).
    Mailbox* mb = ThreadMailbox();
    O3DMI_REQUIRE(mb != nullptr, "host mailbox allocation failed");
    const double* sums_host = mb->data;

    // The iteration's 32 sums, summed over the ranks when the cloud is
    // sharded. `launch(sums_dev, mail_data, mail_flag, seq)` issues the
    // accumulate + final-sum chain. t29..t31: values for out[29..31] (NaN =
    // keep what the kernels computed); they are set BEFORE the rank sum.
    //   no hook      final sum posts to the host mailbox
    //   device hook  final sum -> device buffer -> tail -> caller's collective
    //                on the launch stream (RCCL) -> post kernel -> mailbox:
    //                one host wait per iteration, nothing staged by the host
    //   host hook    as "no hook", then allreduce(host buffer)
    //   communicator (o3dmi_set_comm / o3dmi_set_rccl_comm): the device-hook
    //                route with the library's own collective (ncclAllReduce
    //                on the launch stream) in the hook's place
    o3dmi_comm* comm = ThreadComm();
    if (comm && comm->world <= 1) comm = nullptr;
    const bool dev_reduce = comm != nullptr || g_dev_allreduce != nullptr;
    DeviceBuffer dev_sums;
    if (dev_reduce && (st = dev_sums.Alloc(32 * sizeof(double)))) return st;
    // `sealed`: the launch posts through the search launch's final-sum tail
    // (mailbox.h MailboxWaitSealed), else through a final-sum kernel's fenced
    // post.
    auto fetch_sums = [&](auto&& launch, double* out32, double t29, double t30,
                          double t31, bool sealed = false) -> int {
        const int seq = ++mb->seq;
        if (dev_reduce) {
            double* d = (double*)dev_sums.p;
            int e = launch(d, (double*)nullptr, (int*)nullptr, 0);
            if (e) return e;
            if ((e = o3dmi_internal_sums_tail(d, t29, t30, t31, stream)))
                return e;
            if (comm) {
                if ((e = comm->AllreduceSumF64(d, 32, s))) return e;
            } else if (g_dev_allreduce(d, 32, stream, g_dev_allreduce_user) !=
                       0) {
                SetLastError("device all-reduce hook failed");
                return O3DMI_ERR_INVALID_ARG;
            }
            if ((e = o3dmi_internal_sums_post(d, mb->data, mb->flag, seq,
                                              stream)))
                return e;
            build_next_index();
            O3DMI_HIP_CHECK(MailboxWait(mb, seq, s));
            std::memcpy(out32, sums_host, sizeof(double) * 32);
            return O3DMI_OK;
        }
        int e = launch((double*)nullptr, mb->data, mb->flag, seq);
        if (e) return e;
        build_next_index();  // in the shadow of the launch just issued
        if (sealed) {
            O3DMI_HIP_CHECK(MailboxWaitSealed(mb, seq, s, out32));
        } else {
            O3DMI_HIP_CHECK(MailboxWait(mb, seq, s));
            std::memcpy(out32, sums_host, sizeof(double) * 32);
        }
        if (t29 == t29) out32[29] = t29;
        if (t30 == t30) out32[30] = t30;
        if (t31 == t31) out32[31] = t31;
        if (allreduce && allreduce(out32, 32, allreduce_user) != 0) {
            SetLastError("all-reduce hook failed");
            return O3DMI_ERR_INVALID_ARG;
        }
        return O3DMI_OK;
    };
    const double kKeep = std::nan("");

    double T[16];
    if (init) std::memcpy(T, init, sizeof(T));
    else Eye4(T);
    double fitness = 0, inlier_rmse = 0;
    bool converged = false;
    int iteration_count = 0;
    int status = O3DMI_OK;
    int64_t last_ns = 0;

    // ComputeRegistrationResult (+ the Jacobian sums of the same pass).
    // `pending`: the transformation the source still has to be moved by
    // (`source.Transform(...)`, Registration.cpp:322,404) -- applied by the
    // search launch itself, in place, before it searches.
    double pending[16];
    bool has_pending = false;
    // What this rank works on at a scale: the level's source cloud, or (level
    // sharding) its slice [first, first + ns) of it.
    struct SourceView {
        void* src = nullptr;
        void* srcn = nullptr;
        void* srcc = nullptr;
        int64_t ns = 0, first = 0;
    };
    const bool level_sharding = comm != nullptr && g_level_sharding != 0;
    auto view_of = [&](const Level& L) {
        SourceView v;
        int64_t b = 0, e = L.ns;
        if (level_sharding) {
            const int64_t base = L.ns / comm->world, rem = L.ns % comm->world;
            b = comm->rank * base + (comm->rank < rem ? comm->rank : rem);
            e = b + base + (comm->rank < rem ? 1 : 0);
        }
        const size_t off = (size_t)b * 3 * esz;
        v.src = (char*)L.src.p + off;
        v.srcn = L.srcn.p ? (char*)L.srcn.p + off : nullptr;
        v.srcc = L.srcc.p ? (char*)L.srcc.p + off : nullptr;
        v.ns = e - b;
        v.first = b;
        return v;
    };
    auto search = [&](o3dmi_nns_t* nns, const SourceView& L, int64_t* corr_out,
                      SearchResult& r) -> int {
        const double* xf = has_pending ? pending : nullptr;
        has_pending = false;
        int e = fetch_sums(
                [&](double* sums_dev, double* mail_data, int* mail_flag,
                    int seq) {
                    return o3dmi_internal_icp_transform_search_accumulate(
                            nns, L.src, xf, nullptr, L.ns, search_mode,
                            robust_kernel, scaling_parameter, shape_parameter,
                            corr_out, sums_dev, mail_data, mail_flag, seq,
                            stream);
                },
                r.sums, kKeep, kKeep, (double)L.ns, /*sealed=*/true);
        if (e) return e;
        const double num_correspondences = r.sums[30];
        if (num_correspondences != 0) {
            const double squared_error = r.sums[29];
            r.fitness = num_correspondences / r.sums[31];
            r.inlier_rmse = std::sqrt(squared_error / num_correspondences);
        } else {
            // "0 correspondence present between the pointclouds."
            r.fitness = 0;
            r.inlier_rmse = 0;
        }
        return O3DMI_OK;
    };

    // (Rounds 3-4 carried two more forms of the point-to-plane loop, both
    // bit-compatible and both measured slower -- the 6x6 solve in the search
    // launch's last workgroup with the host one launch ahead, and a next
    // launch queued ahead and gated on a host inbox: docs/rounds.md. What
    // stayed of them is the final sum in the search launch's last workgroup,
    // icp.hip SumTail.)
    for (int scale_idx = 0; scale_idx < num_scales; ++scale_idx) {
        Level& full_level = pyr[(size_t)scale_idx];
        struct ScaleView : SourceView {
            const void* tgt_ptr;
            const void* nrm_ptr;
            const void* tgtc_ptr;
            const void* tgtg_ptr;
            int64_t nt;
        } L;
        static_cast<SourceView&>(L) = view_of(full_level);
        L.tgt_ptr = full_level.tgt_ptr;
        L.nrm_ptr = full_level.nrm_ptr;
        L.tgtc_ptr = full_level.tgtc_ptr;
        L.tgtg_ptr = full_level.tgtg_ptr;
        L.nt = full_level.nt;
        last_ns = L.ns;
        // source_down_pyramid[scale].Transform(result.transformation_) :404
        // (positions and, when the estimator reads them, normals)
        std::memcpy(pending, T, sizeof(pending));
        has_pending = true;
        if (symmetric &&
            (st = o3dmi_transform_normals(T, L.srcn, L.ns, dtype, stream)))
            return st;
        DeviceBuffer corr_buf, sym_partials;
        if (symmetric || colored) {
            if ((st = corr_buf.Alloc(sizeof(int64_t) * (size_t)L.ns))) return st;
            if ((st = sym_partials.Alloc(sizeof(double) * 32 * 1024)))
                return st;
        }
        // target_nns.HybridIndex(max_correspondence_distance) :406-412:
        // built behind the target pyramid (above)
        NnsGuard& guard = guards[(size_t)scale_idx];
        if (scale_idx == 1) {
            if ((st = ensure_indices())) return st;
            if (indices_on_side) O3DMI_HIP_CHECK(hipStreamWaitEvent(s, ev, 0));
        }

        if (timing) {
            (void)hipStreamSynchronize(s);
            const double t = now();
            std::fprintf(stderr,
                         "[o3dmi] icp: scale %d (ns %lld nt %lld) index + "
                         "transform %.0f us\n",
                         scale_idx, (long long)L.ns, (long long)L.nt,
                         t - t_mark);
            t_mark = t;
        }
        // DoSingleScaleICPIterations :275-360
        double prev_fitness = fitness, prev_inlier_rmse = inlier_rmse;
        converged = false;
        int it = 0;
        bool no_corr = false;
        const o3dmi_icp_criteria_t& crit = criterias[scale_idx];
        for (it = 0; it < crit.max_iteration; ++it) {
            SearchResult r;
            guard.completed = false;
            if ((st = search(guard.nns, L,
                             symmetric || colored ? (int64_t*)corr_buf.p
                                                  : nullptr,
                             r)))
                return st;
            guard.completed = true;  // its sums were read: the search is done
            fitness = r.fitness;
            inlier_rmse = r.inlier_rmse;
            converged = false;
            if (r.sums[30] == 0) Eye4(T);  // Registration.cpp:56-58
            if (fitness <= std::numeric_limits<double>::min()) {
                no_corr = true;
                break;
            }
            double pose[6], update[16];
            if (p2plane) {
                float residual;
                int inlier_count;
                int e = o3dmi_decode_and_solve6x6(r.sums, pose, &residual,
                                                  &inlier_count);
                if (e) status = e;  // reference throws; report after the loop
                o3dmi_pose_to_transformation(pose, update);
            } else if (symmetric) {
                // ComputeTransformationSymmetric, kernel/Registration.cpp:
                // 80-135: means of the matched points (from the search pass'
                // moments), 29 sums about them, solve, half-angle pose ->
                // transformation.
                const double cnt = r.sums[15];
                double ms[3], mt[3];
                for (int k = 0; k < 3; ++k) {
                    ms[k] = r.sums[k] / cnt;
                    mt[k] = r.sums[3 + k] / cnt;
                }
                if (dtype == O3DMI_F32)
                    for (int k = 0; k < 3; ++k) {
                        ms[k] = (double)(float)ms[k];
                        mt[k] = (double)(float)mt[k];
                    }
                double sums29[32];
                int e = fetch_sums(
                        [&](double* sums_dev, double* mail_data,
                            int* mail_flag, int seq) {
                            return o3dmi_icp_symmetric_accumulate_post(
                                    L.src, L.srcn, L.tgt_ptr, L.nrm_ptr,
                                    (const int64_t*)corr_buf.p, L.ns, dtype,
                                    ms, mt, robust_kernel, scaling_parameter,
                                    shape_parameter, sums_dev,
                                    (double*)sym_partials.p, mail_data,
                                    mail_flag, seq, stream);
                        },
                        sums29, 0.0, 0.0, 0.0);
                if (e) return e;
                float residual;
                int inlier_count;
                e = o3dmi_decode_and_solve6x6(sums29, pose, &residual,
                                              &inlier_count);
                if (e) status = e;
                o3dmi_symmetric_pose_to_transformation(pose, ms, mt, update);
            } else if (colored) {
                // ComputePoseColoredICP + PoseToTransformation
                // (TransformationEstimation.cpp:420-432)
                double sums29[32];
                int e = fetch_sums(
                        [&](double* sums_dev, double* mail_data,
                            int* mail_flag, int seq) {
                            return o3dmi_icp_colored_accumulate_post(
                                    L.src, L.srcc, L.tgt_ptr, L.nrm_ptr,
                                    L.tgtc_ptr, L.tgtg_ptr,
                                    (const int64_t*)corr_buf.p, L.ns, dtype,
                                    lambda_geometric, robust_kernel,
                                    scaling_parameter, shape_parameter,
                                    sums_dev, (double*)sym_partials.p,
                                    mail_data, mail_flag, seq, stream);
                        },
                        sums29, 0.0, 0.0, 0.0);
                if (e) return e;
                float residual;
                int inlier_count;
                e = o3dmi_decode_and_solve6x6(sums29, pose, &residual,
                                              &inlier_count);
                if (e) status = e;
                o3dmi_pose_to_transformation(pose, update);
            } else {
                // ComputeRtPointToPoint + RtToTransformation
                // (TransformationEstimation.cpp:150-159)
                double R[9], t[3];
                int e = o3dmi_compute_rt_p2point(r.sums, R, t);
                if (e) return e;
                Eye4(update);
                for (int j = 0; j < 3; ++j) {
                    for (int k = 0; k < 3; ++k) update[j * 4 + k] = R[j * 3 + k];
                    update[j * 4 + 3] = t[j];
                }
            }
            Matmul4(update, T, T);
            // source.Transform(update): rides in the next search launch of
            // this scale (a scale that ends here has no further use for its
            // source cloud)
            std::memcpy(pending, update, sizeof(pending));
            has_pending = true;
            if (symmetric && (st = o3dmi_transform_normals(
                                      update, L.srcn, L.ns, dtype, stream)))
                return st;
            if (callback)
                callback(iteration_count + it, scale_idx, it, inlier_rmse,
                         fitness, T, callback_user);
            if (it != 0 &&
                std::abs(prev_fitness - fitness) < crit.relative_fitness &&
                std::abs(prev_inlier_rmse - inlier_rmse) < crit.relative_rmse) {
                converged = true;
                break;
            }
            prev_fitness = fitness;
            prev_inlier_rmse = inlier_rmse;
        }
        iteration_count += it;
        (void)no_corr;
        exit_timer.Mark("scale");
        if (timing) {
            (void)hipStreamSynchronize(s);
            const double t = now();
            std::fprintf(stderr,
                         "[o3dmi] icp: scale %d %d iterations %.0f us\n",
                         scale_idx, it, t - t_mark);
            t_mark = t;
        }

        if (scale_idx == num_scales - 1) {
            // Final fitness / rmse for the stored transformation :424-431
            bool preserved = converged;
            SearchResult r;
            // level sharding: this rank fills its rows of the level's
            // correspondence set, the others read -1 here
            if (level_sharding && correspondences_dev)
                O3DMI_HIP_CHECK(hipMemsetAsync(
                        correspondences_dev, 0xFF,
                        sizeof(int64_t) * (size_t)full_level.ns, s));
            if (level_sharding) last_ns = full_level.ns;
            if ((st = search(guard.nns, L,
                             correspondences_dev
                                     ? correspondences_dev + L.first
                                     : nullptr,
                             r)))
                return st;
            fitness = r.fitness;
            inlier_rmse = r.inlier_rmse;
            if (r.sums[30] == 0) Eye4(T);
            converged = preserved;
            // the search above waited for its sums: nothing of this call is
            // in flight any more (every scale's index was waited for by the
            // scale's first search)
            sync_on_exit.drained = true;
        }
        if (fitness <= std::numeric_limits<double>::min()) {
            converged = false;
            break;
        }
    }

    if (timing)
        std::fprintf(stderr, "[o3dmi] icp: after pyramid %.0f us in total\n",
                     now() - t_start);
    exit_timer.Mark("final");
    std::memcpy(result->transformation, T, sizeof(T));
    result->fitness = fitness;
    result->inlier_rmse = inlier_rmse;
    result->converged = converged ? 1 : 0;
    result->num_iterations = iteration_count;
    result->num_correspondences = correspondences_dev ? last_ns : 0;
    return status;
}

namespace {

// Shared front end of EvaluateRegistration / GetInformationMatrix: clone +
// transform the source, index the target, one fused search + sums pass.
int TransformSearch(const void* source_dev, int64_t ns, const void* target_dev,
                    int64_t nt, int dtype, double max_dist, const double* T,
                    int estimation, int64_t* corr_dev, double* sums32,
                    o3dmi_stream_t stream) {
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "Only Float32 and Float64 point clouds are supported.");
    O3DMI_REQUIRE(source_dev && target_dev && ns > 0 && nt > 0,
                  "Source and/or Target pointcloud is empty.");
    O3DMI_REQUIRE(max_dist > 0, "max_correspondence_distance must be positive");
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == O3DMI_F64 ? 8 : 4;
    DeviceBuffer src;
    struct SyncOnExit {
        hipStream_t s;
        ~SyncOnExit() { (void)hipStreamSynchronize(s); }
    } sync_on_exit{s};
    int st;
    if ((st = src.Alloc((size_t)ns * 3 * esz))) return st;
    O3DMI_HIP_CHECK(hipMemcpyAsync(src.p, source_dev, (size_t)ns * 3 * esz,
                                   hipMemcpyDeviceToDevice, s));
    if (T && (st = o3dmi_transform_points(T, src.p, ns, dtype, stream)))
        return st;
    NnsGuard guard;
    if ((st = o3dmi_nns_create(target_dev, nt, dtype, max_dist, stream,
                               &guard.nns)))
        return st;
    Mailbox* mb = ThreadMailbox();
    O3DMI_REQUIRE(mb != nullptr, "host mailbox allocation failed");
    const int seq = ++mb->seq;
    if ((st = o3dmi_icp_search_accumulate_post(
                 guard.nns, src.p, nullptr, ns, estimation, 0, 1.0, 1.0,
                 corr_dev, nullptr, mb->data, mb->flag, seq, stream)))
        return st;
    // (the search launch's tail posts a sealed block, mailbox.h)
    O3DMI_HIP_CHECK(MailboxWaitSealed(mb, seq, s, sums32));
    return O3DMI_OK;
}

}  // namespace

extern "C" int o3dmi_registration_evaluate(
        const void* source_dev, int64_t ns, const void* target_dev, int64_t nt,
        int dtype, double max_dist, const double* transformation,
        int64_t* correspondences_dev, o3dmi_registration_result_t* result,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(result != nullptr, "result is null");
    double sums[32];
    int st = TransformSearch(source_dev, ns, target_dev, nt, dtype, max_dist,
                             transformation, O3DMI_ICP_POINT_TO_POINT,
                             correspondences_dev, sums, stream);
    if (st) return st;
    // ComputeRegistrationResult, Registration.cpp:24-62.
    const double num = sums[30];
    if (transformation)
        std::memcpy(result->transformation, transformation, sizeof(double) * 16);
    else
        Eye4(result->transformation);
    if (num != 0) {
        result->fitness = num / (double)ns;
        result->inlier_rmse = std::sqrt(sums[29] / num);
    } else {
        result->fitness = 0;
        result->inlier_rmse = 0;
        Eye4(result->transformation);
    }
    result->converged = 0;
    result->num_iterations = 0;
    result->num_correspondences = correspondences_dev ? ns : 0;
    return O3DMI_OK;
}

extern "C" int o3dmi_registration_information_matrix(
        const void* source_dev, int64_t ns, const void* target_dev, int64_t nt,
        int dtype, double max_dist, const double* transformation,
        double* information36, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(information36 != nullptr, "information36 is null");
    double sums[32];
    int st = TransformSearch(source_dev, ns, target_dev, nt, dtype, max_dist,
                             transformation, 2, nullptr, sums, stream);
    if (st) return st;
    if (sums[30] == 0) {
        SetLastError(
                "0 correspondence present between the pointclouds. Try "
                "increasing the max_correspondence_distance parameter.");
        return O3DMI_ERR_NO_INLIERS;
    }
    // RegistrationCPU.cpp:727-733
    int i = 0;
    for (int j = 0; j < 6; j++)
        for (int k = 0; k <= j; k++) {
            information36[j * 6 + k] = information36[k * 6 + j] = sums[i];
            ++i;
        }
    return O3DMI_OK;
}

extern "C" int o3dmi_icp_residual_squares(const void* src_dev,
                                          const void* tgt_dev,
                                          const void* tgt_normals_dev,
                                          const int64_t* corr_dev, int64_t n,
                                          int dtype, double* sums2_dev,
                                          o3dmi_stream_t stream);

extern "C" int o3dmi_registration_compute_rmse(
        int estimation, const void* source_dev, int64_t ns,
        const void* target_dev, const void* target_normals_dev, int dtype,
        const o3dmi_icp_attributes_t* attrs, const int64_t* correspondences_dev,
        double* rmse_out, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(rmse_out != nullptr, "rmse_out is null");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "Only Float32 and Float64 point clouds are supported.");
    O3DMI_REQUIRE(source_dev && target_dev && ns > 0 && correspondences_dev,
                  "Source and/or Target pointcloud is empty.");
    O3DMI_REQUIRE(estimation >= O3DMI_ICP_POINT_TO_PLANE &&
                          estimation <= O3DMI_ICP_COLORED,
                  "unknown estimation");
    hipStream_t s = (hipStream_t)stream;
    DeviceBuffer sums;
    struct SyncOnExit {
        hipStream_t s;
        ~SyncOnExit() { (void)hipStreamSynchronize(s); }
    } sync_on_exit{s};
    int st = sums.Alloc(sizeof(double) * 32);
    if (st) return st;
    double h[32] = {0};
    const double zero3[3] = {0, 0, 0};
    switch (estimation) {
        case O3DMI_ICP_POINT_TO_PLANE:
            O3DMI_REQUIRE(target_normals_dev,
                          "Target pointcloud missing normals attribute.");
            st = o3dmi_icp_residual_squares(source_dev, target_dev,
                                            target_normals_dev,
                                            correspondences_dev, ns, dtype,
                                            (double*)sums.p, stream);
            break;
        case O3DMI_ICP_POINT_TO_POINT:
            st = o3dmi_icp_residual_squares(source_dev, target_dev, nullptr,
                                            correspondences_dev, ns, dtype,
                                            (double*)sums.p, stream);
            break;
        case O3DMI_ICP_SYMMETRIC:
            O3DMI_REQUIRE(attrs && attrs->source_normals && target_normals_dev,
                          "SymmetricICP requires both source and target to "
                          "have normals.");
            // the un-centred residual does not depend on the means
            st = o3dmi_icp_symmetric_accumulate(
                    source_dev, attrs->source_normals, target_dev,
                    target_normals_dev, correspondences_dev, ns, dtype, zero3,
                    zero3, 0, 1.0, 1.0, (double*)sums.p, stream);
            break;
        default: {
            O3DMI_REQUIRE(target_normals_dev,
                          "Target pointcloud missing normals attribute.");
            O3DMI_REQUIRE(attrs && attrs->source_colors && attrs->target_colors,
                          "Source and/or Target pointcloud missing colors "
                          "attribute.");
            O3DMI_REQUIRE(attrs->target_color_gradients,
                          "Target pointcloud missing color_gradients "
                          "attribute.");
            double lambda = attrs->lambda_geometric;
            if (!(lambda >= 0 && lambda <= 1.0)) lambda = 0.968;
            st = o3dmi_icp_colored_accumulate(
                    source_dev, attrs->source_colors, target_dev,
                    target_normals_dev, attrs->target_colors,
                    attrs->target_color_gradients, correspondences_dev, ns,
                    dtype, lambda, 0, 1.0, 1.0, (double*)sums.p, stream);
        }
    }
    if (st) return st;
    const int n_read = (estimation == O3DMI_ICP_POINT_TO_PLANE ||
                        estimation == O3DMI_ICP_POINT_TO_POINT)
                               ? 2
                               : 29;
    O3DMI_HIP_CHECK(hipMemcpyAsync(h, sums.p, sizeof(double) * n_read,
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (n_read == 2) {
        if (h[1] == 0) {
            SetLastError("No valid correspondence present.");
            return O3DMI_ERR_NO_INLIERS;
        }
        *rmse_out = std::sqrt(h[0] / h[1]);
    } else if (estimation == O3DMI_ICP_SYMMETRIC) {
        *rmse_out = h[28] == 0 ? 0.0 : std::sqrt(h[27] / h[28]);
    } else {
        *rmse_out = h[27];
    }
    return O3DMI_OK;
}
