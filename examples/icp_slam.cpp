// Frame-to-model tracking with point-cloud ICP, driven from plain C++ through
// the C ABI only (no Python, no torch): BASELINE configs[2], the loop
// tools/bench_slam.py --mode slam runs from Python.
//
// Per frame, with the calls a user of the reference's tensor API would make:
//   VoxelBlockGrid::GetUniqueBlockCoordinates(previous depth, previous pose)
//       -- here the same set, taken from the Integrate that just touched
//       those blocks: o3dmi_vbg_ray_cast_dev without block coordinates reads
//       the grid's own list (no second touch, no export launch, no host wait;
//       `examples/icp_slam ... 1` calls the reference's function instead)
//   VoxelBlockGrid::RayCast(depth + normal maps)            model frame
//   PointCloud::CreateFromDepthImage(ray-cast depth, stride 2; normals ride
//       along as the per-pixel attribute) + rotate the normals    model cloud
//   PointCloud::CreateFromDepthImage(new depth, stride 2)        frame cloud
//   MultiScaleICP(frame cloud -> model cloud, 5 / 2.5 / 1.25 cm voxels,
//       20 / 10 / 5 iterations, point-to-plane)
//   VoxelBlockGrid::Integrate(new frame at the estimated pose)
//
// The input is the analytic room of analytic_room.h; the trajectory is checked
// against the closed-form poses (exit code 0 = within 8 cm / 1 degree).
//
//   hipcc -O2 -std=c++17 examples/icp_slam.cpp -Iinclude \
//         -Lopen3d_amd/lib -lo3d_mi355x -Wl,-rpath,'$ORIGIN/../open3d_amd/lib' \
//         -o examples/icp_slam
//   examples/icp_slam [frames=60] [width=640] [height=480] [touch_again=0]
//                     [ranks=1] [transport=rccl|loopback]
//
// ranks > 1 (BASELINE configs[3], the tracking half): one host thread per
// rank, rank r on device r % device_count, every rank the same loop on the
// same frames with a replicated model (the ray cast needs every block); what
// is sharded is (1) the Gauss-Newton work of MultiScaleICP -- o3dmi_set_comm +
// o3dmi_set_icp_level_sharding(1): each rank searches / accumulates its slice
// of every pyramid level and the 32 float64 sums are all-reduced inside the
// iteration by the library itself (ncclAllReduce on the launch stream over
// xGMI; no Python anywhere) -- and (2) the model frame's ray cast --
// o3dmi_vbg_ray_cast_sharded: a band of pixel rows per rank, one all-gather
// per map, every rank ends with the single-rank maps bit for bit.
// `loopback` swaps RCCL for an in-process
// transport (host threads meeting at a barrier), which also runs when the
// ranks share one GPU -- RCCL refuses that -- and is how the mode is
// exercised on a single-GPU box. Every rank must end with the same poses.

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "analytic_room.h"
#include "o3d_mi355x_host.h"

namespace {

#define CHECK_HIP(expr)                                                      \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #expr, \
                         hipGetErrorString(e_));                             \
            std::exit(2);                                                    \
        }                                                                    \
    } while (0)

#define CHECK_O3D(expr)                                                       \
    do {                                                                      \
        int s_ = (expr);                                                      \
        if (s_ != O3DMI_OK) {                                                 \
            std::fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__,  \
                         #expr, s_, o3dmi_last_error());                      \
            std::exit(2);                                                     \
        }                                                                     \
    } while (0)

using analytic_room::Camera;
using analytic_room::RenderFrame;

template <typename T>
T* DeviceAlloc(size_t n) {
    void* p = nullptr;
    CHECK_HIP(hipMalloc(&p, sizeof(T) * (n ? n : 1)));
    return (T*)p;
}

void Matmul4(const double* A, const double* B, double* out) {
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            r[i * 4 + j] = s;
        }
    for (int i = 0; i < 16; ++i) out[i] = r[i];
}

// inverse of a rigid transformation [R t; 0 1]
void InvertRigid(const double* T, double* out) {
    double r[16] = {0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[i * 4 + j] = T[j * 4 + i];
    for (int i = 0; i < 3; ++i)
        r[i * 4 + 3] = -(r[i * 4 + 0] * T[3] + r[i * 4 + 1] * T[7] +
                         r[i * 4 + 2] * T[11]);
    r[15] = 1.0;
    for (int i = 0; i < 16; ++i) out[i] = r[i];
}

// ---- in-process transport for the loopback mode -------------------------------
struct Loopback {
    int world;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    long generation = 0;
    std::vector<double> rows;  // [world][32]
    std::vector<char> gathered;  // all-gather: [world][bytes per rank]
    explicit Loopback(int w) : world(w), rows((size_t)w * 32, 0.0) {}
    void Barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const long g = generation;
        if (++waiting == world) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != g; });
        }
    }
};
struct LoopbackRank {
    Loopback* shared;
    int rank;
};
// o3dmi_transport_t::allreduce_sum_f64: the stream is drained, the rows meet
// on the host, every rank adds them in rank order (identical sums everywhere)
int LoopbackAllreduce(void* user, double* dev, int64_t n, o3dmi_stream_t s) {
    auto* me = (LoopbackRank*)user;
    Loopback* lb = me->shared;
    if (n > 32) return 1;
    if (hipStreamSynchronize((hipStream_t)s) != hipSuccess) return 1;
    if (hipMemcpy(&lb->rows[(size_t)me->rank * 32], dev, sizeof(double) * n,
                  hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    lb->Barrier();
    double sum[32] = {0};
    for (int r = 0; r < lb->world; ++r)
        for (int64_t i = 0; i < n; ++i) sum[i] += lb->rows[(size_t)r * 32 + i];
    lb->Barrier();  // everybody has read the rows
    return hipMemcpy(dev, sum, sizeof(double) * n, hipMemcpyHostToDevice) ==
                           hipSuccess
                   ? 0
                   : 1;
}

// o3dmi_transport_t::allgather the same way (send may lie inside recv: the
// segment is on the host before anybody writes recv)
int LoopbackAllgather(void* user, const void* send, void* recv, int64_t bytes,
                      o3dmi_stream_t s) {
    auto* me = (LoopbackRank*)user;
    Loopback* lb = me->shared;
    if (hipStreamSynchronize((hipStream_t)s) != hipSuccess) return 1;
    if (me->rank == 0) lb->gathered.resize((size_t)bytes * lb->world);
    lb->Barrier();
    if (hipMemcpy(&lb->gathered[(size_t)bytes * me->rank], send, (size_t)bytes,
                  hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    lb->Barrier();
    const bool ok = hipMemcpy(recv, lb->gathered.data(),
                              (size_t)bytes * lb->world,
                              hipMemcpyHostToDevice) == hipSuccess;
    lb->Barrier();  // everybody has read the segments
    return ok ? 0 : 1;
}

struct RankResult {
    double seconds = 0, worst_translation = 0, worst_angle = 0;
    double phase[4] = {0, 0, 0, 0};
    long iterations = 0;
    std::vector<double> poses;  // 16 per tracked frame
};

int RunRank(int argc, char** argv, int rank, int world, o3dmi_comm_t* comm,
            RankResult* out);

}  // namespace

int main(int argc, char** argv) {
    const int world = argc > 5 ? std::atoi(argv[5]) : 1;
    const std::string transport = argc > 6 ? argv[6] : "rccl";
    if (world <= 1) {
        RankResult r;
        return RunRank(argc, argv, 0, 1, nullptr, &r);
    }
    int n_dev = 1;
    CHECK_HIP(hipGetDeviceCount(&n_dev));
    std::vector<RankResult> res((size_t)world);
    std::vector<int> rc((size_t)world, 0);
    std::vector<std::thread> threads;
    Loopback lb(world);
    char ident[128] = {0};
    if (transport == "rccl") {
        if (world > n_dev) {
            std::fprintf(stderr, "icp_slam: RCCL needs one GPU per rank (%d "
                                 "ranks, %d GPUs); use the loopback "
                                 "transport\n", world, n_dev);
            return 2;
        }
        CHECK_O3D(o3dmi_rccl_unique_id(ident));
    }
    std::vector<LoopbackRank> lranks((size_t)world);
    for (int r = 0; r < world; ++r) {
        lranks[(size_t)r] = {&lb, r};
        threads.emplace_back([&, r] {
            CHECK_HIP(hipSetDevice(r % n_dev));
            o3dmi_comm_t* comm = nullptr;
            if (transport == "rccl") {
                CHECK_O3D(o3dmi_comm_create_rccl(ident, r, world, &comm));
            } else {
                o3dmi_transport_t table = {};
                table.allreduce_sum_f64 = LoopbackAllreduce;
                table.allgather = LoopbackAllgather;
                CHECK_O3D(o3dmi_comm_create_custom(&table, &lranks[(size_t)r],
                                                   r, world, &comm));
            }
            CHECK_O3D(o3dmi_set_comm(comm));
            CHECK_O3D(o3dmi_set_icp_level_sharding(1));
            rc[(size_t)r] = RunRank(argc, argv, r, world, comm,
                                    &res[(size_t)r]);
            CHECK_O3D(o3dmi_set_comm(nullptr));
            CHECK_O3D(o3dmi_comm_destroy(comm));
        });
    }
    for (auto& t : threads) t.join();
    bool same = true;
    for (int r = 1; r < world; ++r)
        same = same && res[(size_t)r].poses == res[0].poses;
    int worst = 0;
    for (int r = 0; r < world; ++r) worst = rc[(size_t)r] ? rc[(size_t)r] : worst;
    std::printf("{\"example\": \"icp_slam.cpp\", \"ranks\": %d, "
                "\"transport\": \"%s\", \"devices\": %d, "
                "\"frames_per_s\": %.1f, \"icp_iterations_per_frame\": "
                "%.2f, \"poses_identical_on_all_ranks\": %s, "
                "\"max_translation_error_m\": %.3g, "
                "\"max_rotation_error_rad\": %.3g}\n",
                world, transport.c_str(), n_dev,
                (double)(res[0].poses.size() / 16) / res[0].seconds,
                (double)res[0].iterations /
                        (double)(res[0].poses.size() / 16),
                same ? "true" : "false", res[0].worst_translation,
                res[0].worst_angle);
    if (!same) std::fprintf(stderr, "icp_slam: ranks disagree on the poses\n");
    return worst ? worst : (same ? 0 : 1);
}

namespace {

int RunRank(int argc, char** argv, int rank, int world, o3dmi_comm_t* comm,
            RankResult* out) {
    (void)comm;
    const int n_frames = argc > 1 ? std::atoi(argv[1]) : 60;
    Camera cam;
    cam.width = argc > 2 ? std::atoi(argv[2]) : 640;
    cam.height = argc > 3 ? std::atoi(argv[3]) : 480;
    const bool touch_again = argc > 4 && std::atoi(argv[4]) != 0;
    cam.fx = 525.0 * cam.width / 640.0;
    cam.fy = 525.0 * cam.height / 480.0;
    cam.cx = 0.5 * cam.width - 0.5;
    cam.cy = 0.5 * cam.height - 0.5;
    const double K[9] = {cam.fx, 0, cam.cx, 0, cam.fy, cam.cy, 0, 0, 1};
    const float depth_scale = 1000.0f, depth_max = 3.0f, trunc = 8.0f;
    const int W = cam.width, H = cam.height;
    const size_t pixels = (size_t)W * H;
    const int64_t stride = 2;
    const size_t cloud_cap = (size_t)(W / stride) * (H / stride);

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    // start-up: the library's kernels are loaded now, not by the first frame
    CHECK_O3D(o3dmi_preload());

    std::vector<uint16_t*> depth_dev((size_t)n_frames);
    std::vector<uint8_t*> color_dev((size_t)n_frames);
    std::vector<double> eye((size_t)n_frames * 3);
    {
        std::vector<uint16_t> d;
        std::vector<uint8_t> c;
        for (int k = 0; k < n_frames; ++k) {
            double* e = &eye[(size_t)k * 3];
            e[0] = -0.2 + 0.005 * k;
            e[1] = 0.03 * std::sin(0.1 * k);
            e[2] = 0.0;
            RenderFrame(cam, e, d, c);
            depth_dev[(size_t)k] = DeviceAlloc<uint16_t>(pixels);
            color_dev[(size_t)k] = DeviceAlloc<uint8_t>(pixels * 3);
            CHECK_HIP(hipMemcpy(depth_dev[(size_t)k], d.data(),
                                sizeof(uint16_t) * pixels,
                                hipMemcpyHostToDevice));
            CHECK_HIP(hipMemcpy(color_dev[(size_t)k], c.data(), pixels * 3,
                                hipMemcpyHostToDevice));
        }
    }

    // VoxelBlockGrid({"tsdf", "weight", "color"}, {f32, u16, u16}, {1, 1, 3},
    //                8 mm, 16, 40000 blocks)
    const char* names[3] = {"tsdf", "weight", "color"};
    const int dtypes[3] = {O3DMI_F32, O3DMI_U16, O3DMI_U16};
    const int channels[3] = {1, 1, 3};
    o3dmi_vbg_t* grid = nullptr;
    CHECK_O3D(o3dmi_vbg_create(3, names, dtypes, channels, 0.008f, 16, 40000,
                               stream, &grid));

    const int64_t keys_cap = (int64_t)(H / 4) * (W / 4) * 4;
    int32_t* keys = DeviceAlloc<int32_t>((size_t)keys_cap * 3);
    int32_t* keys_count = DeviceAlloc<int32_t>(1);
    float* range_map = DeviceAlloc<float>((size_t)(H / 8) * (W / 8) * 2);
    float* rc_depth = DeviceAlloc<float>(pixels);
    float* rc_normal = DeviceAlloc<float>(pixels * 3);
    float* model_pts = DeviceAlloc<float>(cloud_cap * 3);
    float* model_nrm = DeviceAlloc<float>(cloud_cap * 3);
    float* frame_pts = DeviceAlloc<float>(cloud_cap * 3);
    int32_t* counts = DeviceAlloc<int32_t>(2);

    const double voxel_sizes[3] = {0.05, 0.025, 0.0125};
    const o3dmi_icp_criteria_t criteria[3] = {
            {1e-6, 1e-6, 20}, {1e-6, 1e-6, 10}, {1e-6, 1e-6, 5}};
    const double max_dist[3] = {0.15, 0.075, 0.0375};

    // extrinsic (world -> camera) of a camera at `eye` that is only translated
    auto extrinsic_at = [](const double* e, double* X) {
        const double t[16] = {1, 0, 0, -e[0], 0, 1, 0, -e[1],
                              0, 0, 1, -e[2], 0, 0, 0, 1};
        for (int i = 0; i < 16; ++i) X[i] = t[i];
    };
    double X[16];  // current extrinsic estimate
    extrinsic_at(&eye[0], X);
    CHECK_O3D(o3dmi_vbg_integrate_frame(grid, depth_dev[0], H, W, color_dev[0],
                                        H, W, O3DMI_U16, K, K, X, depth_scale,
                                        depth_max, trunc, stream));
    // (one rank: the ray cast reads the grid's own block list; the sharded
    // cast takes the coordinates as a buffer)
    if (world > 1)
        CHECK_O3D(o3dmi_vbg_last_frame_block_coordinates(grid, keys, keys_cap,
                                                         keys_count, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    double worst_translation = 0, worst_angle = 0;
    long iterations = 0;
    // host time per phase (the loop's host waits are inside the ICP call: its
    // pyramid counts and one per Gauss-Newton iteration; the clouds' sizes
    // stay on the device and Integrate is queued)
    double phase[5] = {0, 0, 0, 0, 0};
    auto now = [] {
        return std::chrono::duration<double, std::micro>(
                       std::chrono::steady_clock::now().time_since_epoch())
                .count();
    };
    // The FIRST tracked frame is the warm-up (as bench.py's untimed warm-up
    // steps): it allocates the driver's workspaces and pool blocks, creates
    // the side stream and the mailboxes and launches every kernel for the
    // first time -- 8.7 ms against 0.48 ms for every later frame (kernel
    // trace, profiles/r6k_first_frame.txt). The clock of `frames_per_s` starts
    // behind it; `frames_per_s_with_first_frame` is the whole run (what this
    // example reported until round 6).
    const auto t0 = std::chrono::steady_clock::now();
    auto t1 = t0;
    for (int k = 1; k < n_frames; ++k) {
        if (k == 2) {
            CHECK_HIP(hipStreamSynchronize(stream));
            t1 = std::chrono::steady_clock::now();
            phase[0] = phase[1] = phase[2] = phase[3] = 0;
        }
        const double p0 = now();
        // ---- model cloud at the previous pose ------------------------------
        int64_t m = keys_cap;
        if (touch_again)
            CHECK_O3D(o3dmi_vbg_get_unique_block_coordinates(
                    grid, depth_dev[(size_t)k - 1], O3DMI_U16, H, W, K, X,
                    depth_scale, depth_max, trunc, keys, &m, stream));
        const double p1 = now();
        if (world > 1) {
            // the model frame by all ranks together: a band of rows each (the
            // collective takes the number of keys from the host)
            if (!touch_again) {
                int32_t live = 0;
                CHECK_HIP(hipMemcpyAsync(&live, keys_count, sizeof(live),
                                         hipMemcpyDeviceToHost, stream));
                CHECK_HIP(hipStreamSynchronize(stream));
                m = live < keys_cap ? live : keys_cap;
            }
            CHECK_O3D(o3dmi_vbg_ray_cast_sharded(
                    grid, keys, m, K, X, W, H, range_map, rc_depth, nullptr,
                    nullptr, rc_normal, depth_scale, 0.1f, depth_max, 1.0f,
                    trunc, 8, stream));
        } else {
            // (without touch_again: the block list and the range map are
            // the grid's own -- no export launch, no clearing launch)
            CHECK_O3D(o3dmi_vbg_ray_cast_dev(
                    grid, touch_again ? keys : nullptr, m, nullptr, K, X, W, H,
                    touch_again ? range_map : nullptr, rc_depth, nullptr,
                    nullptr, rc_normal,
                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                    depth_scale, 0.1f, depth_max, 1.0f, trunc, 8, stream));
        }
        // Both clouds in the PREVIOUS CAMERA's frame, where the ray cast
        // leaves its vertex and normal maps (the frame the reference's
        // slam::Model tracks in, Model.cpp:69-89: the synthesised model frame
        // against the input frame): the model's normals are used as rendered
        // -- until round 6 this example moved both clouds to the world and
        // rotated the normals after them, one more launch per frame.
        static const double kEye[16] = {1, 0, 0, 0, 0, 1, 0, 0,
                                        0, 0, 1, 0, 0, 0, 0, 1};
        // ... and the frame cloud of the new depth image, in its own camera's
        // frame (the identity is ICP's initial guess for the motion between
        // the two), by the same launch.
        CHECK_O3D(o3dmi_unproject_pair(
                rc_depth, O3DMI_F32, rc_normal, model_pts, model_nrm, counts,
                kEye, depth_dev[(size_t)k], O3DMI_U16, nullptr, frame_pts,
                nullptr, counts + 1, kEye, H, W, K, depth_scale, depth_max,
                stride, stream));
        const double p2 = now();
        // The two cloud sizes stay on the device (no read-back, no stream
        // drain per frame): the ICP driver takes them from the device words
        // (the host arguments are then the buffer capacities).
        // ---- track ----------------------------------------------------------
        o3dmi_registration_result_t r;
        CHECK_O3D(o3dmi_registration_set_device_counts(counts + 1, counts));
        CHECK_O3D(o3dmi_registration_multiscale_icp(
                frame_pts, (int64_t)cloud_cap, model_pts, model_nrm,
                (int64_t)cloud_cap, O3DMI_F32, 3, voxel_sizes, criteria,
                max_dist, nullptr, /*L2Loss*/ 0, 1.0, 1.0, nullptr, nullptr,
                nullptr, nullptr, nullptr, &r, stream));
        iterations += r.num_iterations;
        const double p3 = now();
        // p_prev_cam = r.T p_cam, p_prev_cam = X_prev p_world
        //   =>  X_k = r.T^-1 X_prev
        double rinv[16];
        InvertRigid(r.transformation, rinv);
        Matmul4(rinv, X, X);
        out->poses.insert(out->poses.end(), X, X + 16);
        // ---- integrate at the estimated pose -------------------------------
        CHECK_O3D(o3dmi_vbg_integrate_frame(
                grid, depth_dev[(size_t)k], H, W, color_dev[(size_t)k], H, W,
                O3DMI_U16, K, K, X, depth_scale, depth_max, trunc, stream));
        if (!touch_again && world > 1)
            CHECK_O3D(o3dmi_vbg_last_frame_block_coordinates(
                    grid, keys, keys_cap, keys_count, stream));
        const double p4 = now();
        phase[0] += p1 - p0;
        phase[1] += p2 - p1;
        phase[2] += p3 - p2;
        phase[3] += p4 - p3;

        const double* e = &eye[(size_t)k * 3];
        double P[16];
        InvertRigid(X, P);  // camera pose in the world
        const double dt = std::sqrt((P[3] - e[0]) * (P[3] - e[0]) +
                                    (P[7] - e[1]) * (P[7] - e[1]) +
                                    (P[11] - e[2]) * (P[11] - e[2]));
        const double tr = (P[0] + P[5] + P[10] - 1.0) * 0.5;
        const double ang = std::acos(std::fmax(-1.0, std::fmin(1.0, tr)));
        const char* verbose = std::getenv("ICP_SLAM_VERBOSE");
        if (verbose && *verbose)
            std::fprintf(stderr, "frame %d error %.4f m, %.4f rad, %d it\n", k,
                         dt, ang, r.num_iterations);
        worst_translation = std::fmax(worst_translation, dt);
        worst_angle = std::fmax(worst_angle, ang);
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    const auto t_end = std::chrono::steady_clock::now();
    const double seconds_all = std::chrono::duration<double>(t_end - t0).count();
    const int timed = n_frames > 2 ? n_frames - 2 : n_frames - 1;
    const double seconds =
            n_frames > 2 ? std::chrono::duration<double>(t_end - t1).count()
                         : seconds_all;
    out->seconds = seconds_all;
    out->worst_translation = worst_translation;
    out->worst_angle = worst_angle;
    out->iterations = iterations;
    if (world == 1)
    std::printf(
            "{\"example\": \"icp_slam.cpp\", \"frames\": %d, \"width\": %d, "
            "\"height\": %d, \"frames_per_s\": %.1f, \"ms_per_frame\": %.4f, "
            "\"timed_frames\": %d, \"first_frame_ms\": %.3f, "
            "\"frames_per_s_with_first_frame\": %.1f, "
            "\"icp_iterations_per_frame\": %.2f, "
            "\"touch_again\": %d, "
            "\"host_us_block_touch_clouds_icp_integrate\": [%.0f, %.0f, %.0f, "
            "%.0f], "
            "\"max_translation_error_m\": %.3g, \"max_rotation_error_rad\": "
            "%.3g}\n",
            n_frames - 1, W, H, timed / seconds, seconds / timed * 1e3, timed,
            (seconds_all - seconds) * 1e3, (n_frames - 1) / seconds_all,
            (double)iterations / (n_frames - 1), (int)touch_again,
            phase[0] / timed, phase[1] / timed, phase[2] / timed,
            phase[3] / timed, worst_translation, worst_angle);

    (void)hipFree(keys);
    (void)hipFree(keys_count);
    (void)hipFree(range_map);
    (void)hipFree(rc_depth);
    (void)hipFree(rc_normal);
    (void)hipFree(model_pts);
    (void)hipFree(model_nrm);
    (void)hipFree(frame_pts);
    (void)hipFree(counts);
    CHECK_O3D(o3dmi_vbg_destroy(grid));
    for (int k = 0; k < n_frames; ++k) {
        (void)hipFree(depth_dev[(size_t)k]);
        (void)hipFree(color_dev[(size_t)k]);
    }
    (void)hipStreamDestroy(stream);
    const bool ok = worst_translation < 0.08 && worst_angle < 0.01745;
    if (!ok)
        std::fprintf(stderr, "icp_slam: self-check FAILED (rank %d)\n", rank);
    return ok ? 0 : 1;
}

}  // namespace
