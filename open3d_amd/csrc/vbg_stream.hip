// Frame-stream fast path of VoxelBlockGrid integration on MI355X (see
// stream_path.h for the contract and the reference lines it re-cuts).
//
//   front role      <- DepthTouchCPU (t/geometry/kernel/VoxelBlockGridCPU.cpp:117-201)
//                      + HashMap::Activate (core/hashmap/HashMap.cpp:166-181)
//                      + the per-pixel part of IntegrateCPU's lambda
//                        (t/geometry/kernel/VoxelBlockGridImpl.h:258-262,277-289)
//   integrate role  <- the per-voxel part of IntegrateCPU's lambda
//                        (VoxelBlockGridImpl.h:220-257,263-303), applied for
//                        each frame of a group in frame order
//
// FrameStepKernel runs the front roles of group g+1 and the integrate role of
// group g in ONE launch: the front role is a latency chain of hash atomics on
// ~75 workgroups per frame, the integrate role a bandwidth-bound sweep over
// ~4x10^3 work items; side by side they cost max() instead of sum().

#include <cmath>

#include "common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "stream_path.h"
#include "touch_device.h"

namespace o3dmi {
namespace {

template <typename T, int N, int A>
struct alignas(A) Vec {
    T v[N];
};

struct PrepParams {
    Camera color_cam;  // colour intrinsics, identity extrinsic, scale 1
    int color_rows, color_cols;
    bool with_color;
};

struct FrontParams {
    TouchParams p;
    PrepParams pp;
    const uint16_t* depth;
    const uint8_t* color;
    PixelRec* recs;
    FrameBlock* list;
    int64_t list_capacity;
    int* out_count;
    unsigned long long group_stamp;
    int group_bit;
    int n_touch_wg, n_prep_wg;
};

// `wg` = index of this workgroup within one frame's front role,
// [0, n_touch_wg + n_prep_wg). Workgroups [0, n_touch_wg) run the fused
// touch + activate, emitting {slot, key} entries for blocks not yet listed in
// this group; the rest run the per-pixel prepare pass.
__device__ __forceinline__ void FrontRole(const HashView& hv,
                                          const FrontParams& fp, int wg) {
    const TouchParams& p = fp.p;
    const PrepParams& pp = fp.pp;
    const uint16_t* __restrict__ depth = fp.depth;
    const uint8_t* __restrict__ color = fp.color;
    PixelRec* __restrict__ recs = fp.recs;
    FrameBlock* __restrict__ list = fp.list;
    const int n_touch_wg = fp.n_touch_wg;
    if (wg < n_touch_wg) {
        int n = p.rows_strided * p.cols_strided;
        int n_padded = ((n + 63) / 64) * 64;
        for (int w = wg * blockDim.x + threadIdx.x; w < n_padded;
             w += n_touch_wg * blockDim.x) {
            int xb[4], yb[4], zb[4];
            bool valid = (w < n) && RayCandidates(p, depth, w, xb, yb, zb);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bool ok = valid;
                if (ok && s > 0 && xb[s] == xb[s - 1] && yb[s] == yb[s - 1] &&
                    zb[s] == zb[s - 1])
                    ok = false;
                if (ok && !KeyInRange(xb[s], yb[s], zb[s])) {
                    atomicOr(&hv.counters[1], kErrKeyRange);
                    ok = false;
                }
                unsigned long long k = ok ? PackKey(xb[s], yb[s], zb[s]) : 0ull;
                if (ok && !hv.Owns(k)) ok = false;  // another rank's block
                if (WaveLeaderForKey(k, ok)) {
                    unsigned slot;
                    InsertKey<true>(hv, xb[s], yb[s], zb[s], slot);
                    if (TouchSlot(hv, slot, fp.group_stamp, fp.group_bit)) {
                        int o = atomicAdd(fp.out_count, 1);
                        if (o < fp.list_capacity) {
                            FrameBlock fb;
                            fb.slot = (int)slot;
                            fb.x = xb[s];
                            fb.y = yb[s];
                            fb.z = zb[s];
                            list[o] = fb;
                        } else {
                            atomicOr(&hv.counters[1], kErrCapacity);
                        }
                    }
                }
            }
        }
        return;
    }
    // Prepare pass: the voxel-independent sub-expressions of the integrate
    // lambda (VoxelBlockGridImpl.h:258-262 depth, :277-289 colour pixel).
    const int n_px = p.rows * p.cols;
    const int n_wg = fp.n_prep_wg;
    for (int i = (wg - n_touch_wg) * blockDim.x + threadIdx.x; i < n_px;
         i += n_wg * blockDim.x) {
        const int vi = i / p.cols;
        const int ui = i - vi * p.cols;
        PixelRec r;
        r.d = (float)depth[i] / p.depth_scale;
        r.rgba = 0u;
        if (pp.with_color) {
            float x, y, z, uf, vf;
            p.cam.Unproject((float)ui, (float)vi, 1.0f, x, y, z);
            pp.color_cam.Project(x, y, z, uf, vf);
            if (InBoundary2D(uf, vf, pp.color_rows, pp.color_cols)) {
                int uc = (int)roundf(uf);
                int vc = (int)roundf(vf);
                const uint8_t* in =
                        color + ((int64_t)vc * pp.color_cols + uc) * 3;
                r.rgba = (unsigned)in[0] | ((unsigned)in[1] << 8) |
                         ((unsigned)in[2] << 16) | (1u << 24);
            }
        }
        recs[i] = r;
    }
}

// ---- integrate role ---------------------------------------------------------
// Work item = (block of the group's list, 256-quad part of that block); the
// role strides over items so that a group's ~10^3 blocks spread as ~4x10^3
// workgroups over the 256 CUs. A lane owns 4 x-consecutive voxels: tsdf moves
// as one 16-byte access, u16 weight as 8 bytes, u16 colour as 24 bytes. Depth
// and colour come from the prepared PixelRec images (one 8-byte gather per
// voxel and frame).
struct IntegParams {
    Camera cam[kMaxGroup];  // depth intrinsics + extrinsic, scale = voxel_size
    const PixelRec* recs[kMaxGroup];
    int n_frames;
    int rows, cols, resolution;
    float sdf_trunc, depth_max;
    float inv_sdf_trunc;  // RN(1 / sdf_trunc), used by the kFastDiv variant
    const FrameBlock* list;
    const int* count;
    int64_t list_capacity;
    float* tsdf;
    void* weight;
    void* color;
    int* zero_counter;
    int* size_host;
    int status_stamp;
    int* prof_count;
    int* prof_frame_blocks;
};

// ---- exact division without the division sequence ---------------------------
// The per-voxel update has three correctly rounded float divisions; a full
// IEEE sequence is ~11 VALU instructions and the role is VALU-bound. Two of
// them have a special shape:
//   sdf / sdf_trunc   -- the divisor is a per-launch constant: with
//                        y = RN(1/b): q0 = RN(a y), r = fma(-b, q0, a) (exact),
//                        q = fma(r, y, q0)            (Markstein's correction);
//                        |a| < 1e-30 (zeros, the underflow range) keeps the
//                        IEEE sequence
//   1 / (w + 1)       -- w + 1 is an integer in [1, 65536] (uint16 weights):
//                        hardware reciprocal + one Newton step.
// Neither identity is taken on trust: before a kernel uses the short forms,
// VerifyFastDivision() compares them on the device against the IEEE division
// for EVERY float |a| <= b (the whole range the update can produce) and every
// integer 1..65536; any mismatch keeps the IEEE sequence. The check costs a
// few ms once per distinct truncation distance.
__device__ __forceinline__ float DivByConst(float a, float b, float y) {
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
}
__device__ __forceinline__ float RcpSmallInt(float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

// Lower bound of the magnitudes DivByConst handles itself; smaller inputs
// (zeros, denormals and their neighbourhood, where the exact-residual argument
// needs gradual underflow to cooperate) take the IEEE sequence.
constexpr float kDivTiny = 1.0e-30f;

__device__ __forceinline__ float DivByConstGuarded(float a, float b, float y) {
    // wave-uniform branch: the IEEE sequence only when some lane needs it
    if (__builtin_amdgcn_ballot_w64(fabsf(a) < kDivTiny) != 0ull) return a / b;
    return DivByConst(a, b, y);
}

__global__ void VerifyDivKernel(float b, float y, unsigned max_bits,
                                int* __restrict__ mismatch) {
    int bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x +
                                threadIdx.x;
         i <= max_bits; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __uint_as_float((unsigned)i);
        const float want_p = a / b, want_n = (-a) / b;
        const float got_p = DivByConstGuarded(a, b, y);
        const float got_n = DivByConstGuarded(-a, b, y);
        if (__float_as_uint(want_p) != __float_as_uint(got_p) ||
            __float_as_uint(want_n) != __float_as_uint(got_n))
            bad |= 1;
    }
    if (bad) atomicOr(mismatch, bad);
}
// 1 / z of the projection (z arbitrary): hardware reciprocal + kRcpSteps
// Newton steps inside [2^-60, 2^60]; IEEE outside. Verified for every float of
// that range (all 2^23 x 120 of them) before use.
template <int kSteps>
__device__ __forceinline__ float RcpGuarded(float z) {
    // One unsigned compare on the bit pattern covers sign, zero, denormals,
    // inf / NaN and both ends of the verified range; the branch is made
    // wave-uniform so that the common case is a straight scalar jump.
    const bool out = (__float_as_uint(z) - 0x21800000u) >= (0x5E000000u - 0x21800000u);
    if (__builtin_amdgcn_ballot_w64(out) != 0ull) return 1.0f / z;
    float r = __builtin_amdgcn_rcpf(z);
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        const float e = __builtin_fmaf(-z, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    return r;
}
template <int kSteps>
__global__ void VerifyRcpZKernel(int* __restrict__ mismatch, int bit) {
    // every positive float between 2^-61 and 2^61 (a margin around the range)
    const unsigned lo = 0x21000000u, hi = 0x5E800000u;
    bool bad = false;
    for (unsigned long long i = lo + blockIdx.x * (unsigned long long)blockDim.x +
                                threadIdx.x;
         i <= hi; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float z = __uint_as_float((unsigned)i);
        bad |= __float_as_uint(1.0f / z) !=
               __float_as_uint(RcpGuarded<kSteps>(z));
    }
    if (bad) atomicOr(mismatch, bit);
}

__global__ void VerifyRcpKernel(int* __restrict__ mismatch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i > 65536) return;
    const float b = (float)i;
    if (__float_as_uint(1.0f / b) != __float_as_uint(RcpSmallInt(b)))
        atomicOr(mismatch, 2);
}

// kDiv: 0 = IEEE divisions; 1 = short sdf / trunc and 1 / (w + 1); 2, 3 = also
// the short 1 / z with one / two Newton steps (whichever verified).
template <typename weight_t, typename color_t, bool kColor, int kDiv>
__device__ __forceinline__ void IntegrateRole(const HashView& hv,
                                              const IntegParams& ip, int wg,
                                              int n_wg) {
    using TVec = Vec<float, 4, 16>;
    using WVec = Vec<weight_t, 4, 4 * sizeof(weight_t)>;
    using CVec = Vec<color_t, 12, 4 * sizeof(color_t)>;
    float* __restrict__ tsdf_base = ip.tsdf;
    weight_t* __restrict__ weight_base = (weight_t*)ip.weight;
    color_t* __restrict__ color_base = (color_t*)ip.color;
    const FrameBlock* __restrict__ list = ip.list;
    int64_t n_blocks = *ip.count;
    if (n_blocks > ip.list_capacity) n_blocks = ip.list_capacity;

    if (wg == 0 && threadIdx.x == 0) {
        if (ip.zero_counter) *ip.zero_counter = 0;
        if (ip.prof_count) *ip.prof_count = (int)n_blocks;
        if (ip.size_host) {
            // The front roles of this group completed in an earlier launch,
            // so heap_top is at least the map size after this group's
            // activation (front roles of the next group may already be adding
            // to it; the host only needs an upper bound).
            ip.size_host[0] = hv.counters[0];
            ip.size_host[1] = hv.counters[1];
            ip.size_host[2] = (int)n_blocks;
            __hip_atomic_store(&ip.size_host[3], ip.status_stamp,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    const int res = ip.resolution;
    const int res3 = res * res * res;
    const int quads_per_row = res >> 2;
    const int n_quads = res3 >> 2;
    const int parts = (n_quads + 255) >> 8;
    const int64_t n_items = n_blocks * parts;
    int frame_blocks = 0;  // lane 0 of part 0 counts block-frames

    for (int64_t item = wg; item < n_items; item += n_wg) {
        const int64_t b = item / parts;
        const int part = (int)(item - b * parts);
        // Wave-uniform block header: {slot, key} -> buffer index, frame bits.
        const FrameBlock fb = list[b];
        const int slot = __builtin_amdgcn_readfirstlane(fb.slot);
        const int xb = __builtin_amdgcn_readfirstlane(fb.x);
        const int yb = __builtin_amdgcn_readfirstlane(fb.y);
        const int zb = __builtin_amdgcn_readfirstlane(fb.z);
        const int block_idx =
                __builtin_amdgcn_readfirstlane(hv.slot_vals[slot]);
        const unsigned bits = __builtin_amdgcn_readfirstlane(
                (unsigned)(hv.slot_touch[slot] & 0xffull));
        const int64_t block_base = (int64_t)block_idx * res3;
        if (part == 0 && threadIdx.x == 0 && ip.prof_frame_blocks)
            frame_blocks += __popc(bits);

        const int q = (part << 8) + threadIdx.x;
        if (q >= n_quads) continue;
        const int qx = q % quads_per_row;
        const int row = q / quads_per_row;
        const int yv = row % res;
        const int zv = row / res;
        const int x0 = xb * res + (qx << 2);
        const float fy = (float)(yb * res + yv);
        const float fz = (float)(zb * res + zv);
        const int64_t lin0 = block_base + ((int64_t)q << 2);

        TVec t4;
        WVec w4;
        CVec c12;
        bool loaded = false;

        for (int f = 0; f < ip.n_frames; ++f) {
            if (!((bits >> f) & 1u)) continue;  // wave-uniform
            const Camera& cam = ip.cam[f];
            const PixelRec* __restrict__ recs = ip.recs[f];
            // VoxelBlockGridImpl.h:244-267 with depth taken from the record.
            float sdf[4];
            unsigned rgba[4];
            bool ok[4];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float xc, yc, zc, u, v;
                cam.RigidTransform((float)(x0 + j), fy, fz, xc, yc, zc);
                if constexpr (kDiv >= 2) {
                    // Camera::Project with the verified short reciprocal
                    const float inv_z = RcpGuarded<kDiv - 1>(zc);
                    u = cam.fx * xc * inv_z + cam.cx;
                    v = cam.fy * yc * inv_z + cam.cy;
                } else {
                    cam.Project(xc, yc, zc, u, v);
                }
                ok[j] = InBoundary2D(u, v, ip.rows, ip.cols);
                sdf[j] = 0.f;
                rgba[j] = 0u;
                if (ok[j]) {
                    const int ui = (int)u;
                    const int vi = (int)v;
                    const PixelRec r = recs[(int64_t)vi * ip.cols + ui];
                    const float d = r.d;
                    float sd = d - zc;
                    if (d <= 0 || d > ip.depth_max || zc <= 0 ||
                        sd < -ip.sdf_trunc) {
                        ok[j] = false;
                    } else {
                        sd = sd < ip.sdf_trunc ? sd : ip.sdf_trunc;
                        sdf[j] = kDiv >= 1
                                         ? DivByConstGuarded(sd, ip.sdf_trunc,
                                                             ip.inv_sdf_trunc)
                                         : sd / ip.sdf_trunc;
                        rgba[j] = r.rgba;
                    }
                }
                any |= ok[j];
            }
            if (!any) continue;
            if (!loaded) {
                t4 = *reinterpret_cast<const TVec*>(tsdf_base + lin0);
                w4 = *reinterpret_cast<const WVec*>(weight_base + lin0);
                if constexpr (kColor)
                    c12 = *reinterpret_cast<const CVec*>(color_base + 3 * lin0);
                loaded = true;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!ok[j]) continue;
                // VoxelBlockGridImpl.h:269-302
                float inv_wsum;
                if constexpr (sizeof(weight_t) == 2)
                    inv_wsum = kDiv >= 1
                                       ? RcpSmallInt((float)((int)w4.v[j] + 1))
                                       : 1.0f / (float)((int)w4.v[j] + 1);
                else
                    inv_wsum = 1.0f / (w4.v[j] + 1);
                const float weight = (float)w4.v[j];
                t4.v[j] = (weight * t4.v[j] + sdf[j]) * inv_wsum;
                if constexpr (kColor) {
                    if (rgba[j] >> 24) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            // colour multiplier is 1 for uint8 input
                            const float in =
                                    (float)((rgba[j] >> (8 * i)) & 0xffu);
                            c12.v[3 * j + i] = (color_t)(
                                    (weight * (float)c12.v[3 * j + i] + in) *
                                    inv_wsum);
                        }
                    }
                }
                w4.v[j] = (weight_t)(weight + 1);
            }
        }
        if (loaded) {
            *reinterpret_cast<TVec*>(tsdf_base + lin0) = t4;
            *reinterpret_cast<WVec*>(weight_base + lin0) = w4;
            if constexpr (kColor)
                *reinterpret_cast<CVec*>(color_base + 3 * lin0) = c12;
        }
    }
    if (frame_blocks) atomicAdd(ip.prof_frame_blocks, frame_blocks);
}

struct StepParams {
    HashView hv;
    FrontParams front[kMaxGroup];
    IntegParams integ;
    int n_fronts;
    int front_wg;  // workgroups per front role
};

template <typename weight_t, typename color_t, bool kColor, int kDiv>
__global__ void __launch_bounds__(256) FrameStepKernel(StepParams sp) {
    const int b = (int)blockIdx.x;
    const int n_front_wg = sp.n_fronts * sp.front_wg;
    if (b < n_front_wg) {
        const int f = b / sp.front_wg;
        FrontRole(sp.hv, sp.front[f], b - f * sp.front_wg);
    } else {
        IntegrateRole<weight_t, color_t, kColor, kDiv>(
                sp.hv, sp.integ, b - n_front_wg, (int)gridDim.x - n_front_wg);
    }
}

}  // namespace

int64_t FrustumBlockBound(const double* K, int rows, int cols, float depth_max,
                          float block_size, int stride) {
    const int64_t rays = (int64_t)(rows / stride) * (cols / stride);
    const int64_t by_rays = rays * 4;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    if (!(fx > 0) || !(fy > 0) || !(block_size > 0) || !(depth_max > 0))
        return by_rays;
    // |x| <= tx*z, |y| <= ty*z, 0 <= z <= depth_max contains every sample
    // o + t*dir (dir.z = 1, t <= depth_max) in the camera frame.
    const double tx = std::fmax(std::fabs(cx), std::fabs(cols - 1 - cx)) / fx;
    const double ty = std::fmax(std::fabs(cy), std::fabs(rows - 1 - cy)) / fy;
    const double r = std::sqrt(3.0) * block_size;  // block diagonal
    const double sx = tx / std::sqrt(1 + tx * tx), sy = ty / std::sqrt(1 + ty * ty);
    if (!(sx > 0) || !(sy > 0)) return by_rays;
    // The r-dilation of the pyramid lies inside the pyramid with the same
    // side slopes, apex moved back by a and far plane at depth_max + r.
    const double a = r / std::fmin(sx, sy);
    const double h = (double)depth_max + r + a;
    const double vol = 4.0 / 3.0 * tx * ty * h * h * h;
    const double nb = vol / ((double)block_size * block_size * block_size);
    if (!(nb < 9e15)) return by_rays;
    const int64_t by_volume = (int64_t)std::ceil(nb) + 1;
    return by_volume < by_rays ? by_volume : by_rays;
}

// Exhaustive on-device proof that the short division forms equal the IEEE
// division for this truncation distance (see DivByConst); cached per value.
static int VerifyFastDivision(float b, float* y_out) {
    static std::mutex mu;
    static std::map<unsigned, int> cache;
    unsigned key;
    std::memcpy(&key, &b, sizeof(key));
    const float y = 1.0f / b;  // IEEE: correctly rounded reciprocal
    *y_out = y;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int ok = 0;
    static const bool disabled = std::getenv("O3DMI_EXACT_DIV") != nullptr;
    if (!disabled && b > 0.0f && std::isfinite(b) && std::isfinite(y)) {
        int* flag = nullptr;
        if (hipMalloc((void**)&flag, sizeof(int)) == hipSuccess) {
            // A private stream: the caller's stream may be mid-pipeline.
            hipStream_t vs = nullptr;
            if (hipStreamCreateWithFlags(&vs, hipStreamNonBlocking) ==
                hipSuccess) {
                int host = -1;
                (void)hipMemsetAsync(flag, 0, sizeof(int), vs);
                hipLaunchKernelGGL(VerifyDivKernel, dim3(kCUs * 16), dim3(256),
                                   0, vs, b, y, key, flag);
                hipLaunchKernelGGL(VerifyRcpKernel, dim3(256), dim3(256), 0, vs,
                                   flag);
                hipLaunchKernelGGL(VerifyRcpZKernel<1>, dim3(kCUs * 16),
                                   dim3(256), 0, vs, flag, 4);
                hipLaunchKernelGGL(VerifyRcpZKernel<2>, dim3(kCUs * 16),
                                   dim3(256), 0, vs, flag, 8);
                if (hipGetLastError() == hipSuccess &&
                    hipMemcpyAsync(&host, flag, sizeof(int),
                                   hipMemcpyDeviceToHost, vs) == hipSuccess &&
                    hipStreamSynchronize(vs) == hipSuccess)
                    // bits 1|2: sdf / w forms; 4: 1/z one step; 8: two steps
                    ok = (host & 3) ? 0 : (!(host & 4) ? 2 : (!(host & 8) ? 3 : 1));
                if (std::getenv("O3DMI_VERBOSE"))
                    std::fprintf(stderr, "[o3dmi] division check flags: %d\n",
                                 host);
                (void)hipStreamDestroy(vs);
            }
            (void)hipFree(flag);
        }
    }
    if (std::getenv("O3DMI_VERBOSE"))
        std::fprintf(stderr,
                     "[o3dmi] exact short division for sdf_trunc = %.9g: %s\n",
                     (double)b,
                     ok == 0 ? "not used"
                             : (ok == 1 ? "sdf, 1/(w+1)"
                                        : (ok == 2 ? "sdf, 1/(w+1), 1/z (1 step)"
                                                   : "sdf, 1/(w+1), 1/z (2 steps)")));
    cache[key] = ok;
    return ok;
}

int LaunchFrameStep(o3dmi_hash* bh, const FrameFrontArgs* fronts, int n_fronts,
                    const IntegrateStreamArgs* a, hipStream_t s) {
    O3DMI_REQUIRE((n_fronts > 0 && fronts) || a, "nothing to launch");
    O3DMI_REQUIRE(n_fronts >= 0 && n_fronts <= kMaxGroup, "bad group size");
    StepParams sp = {};
    sp.hv = bh->view;
    sp.n_fronts = n_fronts;
    int n_int_wg = 0;
    int grid_dtype = O3DMI_U16;
    bool col = false;
    int fast_div = 0;
    static const double eye4[16] = {1, 0, 0, 0, 0, 1, 0, 0,
                                    0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < n_fronts; ++i) {
        const FrameFrontArgs* f = &fronts[i];
        FrontParams& fp = sp.front[i];
        O3DMI_REQUIRE(f->group_bit >= 0 && f->group_bit < kMaxGroup &&
                              f->group_stamp > 0,
                      "bad group bit / stamp");
        fp.p = MakeTouchParams(f->depth_intrinsic, f->extrinsic, f->rows,
                               f->cols, f->stride, f->resolution,
                               f->voxel_size, f->sdf_trunc, f->depth_scale,
                               f->depth_max);
        fp.pp.color_cam = Camera::Make(f->color_intrinsic ? f->color_intrinsic
                                                          : f->depth_intrinsic,
                                       eye4, 1.0f);
        fp.pp.color_rows = f->color_rows;
        fp.pp.color_cols = f->color_cols;
        fp.pp.with_color = f->color != nullptr;
        fp.depth = f->depth;
        fp.color = f->color;
        fp.recs = f->recs;
        fp.list = f->list;
        fp.list_capacity = f->list_capacity;
        fp.out_count = f->count;
        fp.group_stamp = f->group_stamp;
        fp.group_bit = f->group_bit;
        const int n_rays = fp.p.rows_strided * fp.p.cols_strided;
        fp.n_touch_wg = (n_rays + kBlock - 1) / kBlock;
        // 4 pixels per prepare lane
        fp.n_prep_wg = (f->rows * f->cols + kBlock * 4 - 1) / (kBlock * 4);
        if (fp.n_prep_wg < 1) fp.n_prep_wg = 1;
        const int wg = fp.n_touch_wg + fp.n_prep_wg;
        // all frames of a launch share the image size
        O3DMI_REQUIRE(i == 0 || wg == sp.front_wg,
                      "frames of one launch must share the image size");
        sp.front_wg = wg;
    }
    if (a) {
        O3DMI_REQUIRE(a->resolution % 4 == 0,
                      "frame-stream path needs block_resolution % 4 == 0");
        O3DMI_REQUIRE(a->n_frames >= 1 && a->n_frames <= kMaxGroup,
                      "bad group size");
        IntegParams& ip = sp.integ;
        ip.n_frames = a->n_frames;
        for (int f = 0; f < a->n_frames; ++f) {
            ip.cam[f] = Camera::Make(a->depth_intrinsic, a->extrinsic[f],
                                     a->voxel_size);
            ip.recs[f] = a->recs[f];
        }
        ip.rows = a->rows;
        ip.cols = a->cols;
        ip.resolution = a->resolution;
        ip.sdf_trunc = a->sdf_trunc;
        ip.depth_max = a->depth_max;
        fast_div = VerifyFastDivision(a->sdf_trunc, &ip.inv_sdf_trunc);
        ip.list = a->list;
        ip.count = a->count;
        ip.list_capacity = a->list_capacity;
        ip.tsdf = a->tsdf;
        ip.weight = a->weight;
        ip.color = a->color;
        ip.zero_counter = a->zero_counter;
        ip.size_host = a->size_host;
        ip.status_stamp = a->status_stamp;
        ip.prof_count = a->prof_count;
        ip.prof_frame_blocks = a->prof_frame_blocks;
        const int n_quads =
                (a->resolution * a->resolution * a->resolution) >> 2;
        const int parts = (n_quads + 255) >> 8;
        // Grid from the expected block count (previous group + slack); the
        // role strides, so an under-estimate only costs balance.
        int64_t g = ((int64_t)a->grid_hint + (a->grid_hint >> 2) + 64) * parts;
        const int64_t g_max = (int64_t)kCUs * 32;
        if (g > g_max) g = g_max;
        if (g < kCUs) g = kCUs;
        n_int_wg = (int)g;
        grid_dtype = a->grid_dtype;
        col = a->with_color && a->color != nullptr;
    }
    dim3 grid((unsigned)(n_fronts * sp.front_wg + n_int_wg)), block(256);
#define O3DMI_LAUNCH_STEP_D(WT, VT, COLOR, D)                                 \
    hipLaunchKernelGGL((FrameStepKernel<WT, VT, COLOR, D>), grid, block, 0, s, \
                       sp)
#define O3DMI_LAUNCH_STEP(WT, VT, COLOR)                                      \
    do {                                                                      \
        switch (fast_div) {                                                   \
            case 3: O3DMI_LAUNCH_STEP_D(WT, VT, COLOR, 3); break;             \
            case 2: O3DMI_LAUNCH_STEP_D(WT, VT, COLOR, 2); break;             \
            case 1: O3DMI_LAUNCH_STEP_D(WT, VT, COLOR, 1); break;             \
            default: O3DMI_LAUNCH_STEP_D(WT, VT, COLOR, 0); break;            \
        }                                                                     \
    } while (0)
    if (grid_dtype == O3DMI_U16) {
        if (col) O3DMI_LAUNCH_STEP(uint16_t, uint16_t, true);
        else O3DMI_LAUNCH_STEP(uint16_t, uint16_t, false);
    } else {
        if (col) O3DMI_LAUNCH_STEP(float, float, true);
        else O3DMI_LAUNCH_STEP(float, float, false);
    }
#undef O3DMI_LAUNCH_STEP
#undef O3DMI_LAUNCH_STEP_D
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace o3dmi
