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
#include <string>

#include "stream_path.h"
#include "sliced_path.h"
#include "touch_device.h"

// Measurement builds only (tools/build_variant.sh -DO3DMI_ABLATE_GATHER=1):
// the integrate role's record gathers collapse to one cache line per
// instruction. Never defined in the product build.
#ifndef O3DMI_ABLATE_GATHER
#define O3DMI_ABLATE_GATHER 0
#endif

namespace o3dmi {
namespace {

template <typename T, int N, int A>
struct alignas(A) Vec {
    T v[N];
};

__device__ __forceinline__ float DivByConst(float a, float b, float y) {
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
}
struct PrepParams {
    Camera color_cam;  // colour intrinsics, identity extrinsic, scale 1
    int color_rows, color_cols;
    bool with_color;
};

// LDS key set of one 16 x 16 ray tile (<= 1024 candidates).
constexpr int kTileKeys = 2048;

// Front roles of one launch: what the frames of the group share, and what
// differs per frame (kernel arguments are limited to 4 KB; a group has up to
// 16 frames).
struct FrontShared {
    TouchParams p;  // intrinsics, sizes, truncation; p.cam.e comes per frame
    PrepParams pp;
    const int* col_lut;
    const int* row_lut;
    bool depth_div_short;
    bool prep_identity;     // tables are the identity, cols % 4 == 0
    float inv_depth_scale;  // RN(1 / depth_scale)
    FrameBlock* list;
    int64_t list_capacity;
    int* out_count;
    ReadyEntry* ready;
    int* tickets;
    int* touch_status;
    int n_touch_total;  // touch workgroups of the whole group (all frames)
    unsigned long long group_stamp;
    int touch_plane;
    int n_touch_wg, n_prep_wg;
};
// Field-by-field (the structs carry padding, so memcmp would compare
// indeterminate bytes): everything the frames of a launch must agree on; the
// pose (p.cam.e) travels per frame.
inline bool SameIntrinsics(const Camera& a, const Camera& b) {
    return a.fx == b.fx && a.fy == b.fy && a.cx == b.cx && a.cy == b.cy &&
           a.scale == b.scale;
}
inline bool SameGroup(const FrontShared& a, const FrontShared& b) {
    return SameIntrinsics(a.p.cam, b.p.cam) && a.p.rows == b.p.rows &&
           a.p.cols == b.p.cols && a.p.stride == b.p.stride &&
           a.p.rows_strided == b.p.rows_strided &&
           a.p.cols_strided == b.p.cols_strided &&
           a.p.block_size == b.p.block_size &&
           a.p.sdf_trunc == b.p.sdf_trunc &&
           a.p.depth_scale == b.p.depth_scale &&
           a.p.depth_max == b.p.depth_max &&
           SameIntrinsics(a.pp.color_cam, b.pp.color_cam) &&
           a.pp.color_rows == b.pp.color_rows &&
           a.pp.color_cols == b.pp.color_cols &&
           a.pp.with_color == b.pp.with_color && a.col_lut == b.col_lut &&
           a.row_lut == b.row_lut && a.depth_div_short == b.depth_div_short &&
           a.prep_identity == b.prep_identity &&
           a.inv_depth_scale == b.inv_depth_scale && a.list == b.list &&
           a.list_capacity == b.list_capacity && a.out_count == b.out_count &&
           a.ready == b.ready && a.tickets == b.tickets &&
           a.touch_status == b.touch_status &&
           a.n_touch_total == b.n_touch_total &&
           a.group_stamp == b.group_stamp && a.touch_plane == b.touch_plane &&
           a.n_touch_wg == b.n_touch_wg && a.n_prep_wg == b.n_prep_wg;
}
struct FrontFrame {
    float pose[3][4];  // inverse extrinsic (TouchParams::cam.e)
    const uint16_t* depth;
    const uint8_t* color;
    PixelRec* recs;
    int group_bit;
};

// A frame's front parameters assembled from the two (uniform: scalar loads).
struct FrontParams {
    TouchParams p;
    PrepParams pp;
    const uint16_t* depth;
    const uint8_t* color;
    const int* col_lut;
    const int* row_lut;
    bool depth_div_short;
    bool prep_identity;
    float inv_depth_scale;
    PixelRec* recs;
    FrameBlock* list;
    int64_t list_capacity;
    int* out_count;
    ReadyEntry* ready;
    int* tickets;
    int* touch_status;
    int n_touch_total;
    unsigned long long group_stamp;
    int group_bit;
    int touch_plane;
    int n_touch_wg, n_prep_wg;
    __device__ __forceinline__ FrontParams(const FrontShared& s,
                                           const FrontFrame& f)
        : p(s.p), pp(s.pp), depth(f.depth), color(f.color),
          col_lut(s.col_lut), row_lut(s.row_lut),
          depth_div_short(s.depth_div_short), prep_identity(s.prep_identity),
          inv_depth_scale(s.inv_depth_scale), recs(f.recs), list(s.list),
          list_capacity(s.list_capacity), out_count(s.out_count),
          ready(s.ready), tickets(s.tickets), touch_status(s.touch_status),
          n_touch_total(s.n_touch_total),
          group_stamp(s.group_stamp), group_bit(f.group_bit),
          touch_plane(s.touch_plane), n_touch_wg(s.n_touch_wg),
          n_prep_wg(s.n_prep_wg) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) p.cam.e[i][j] = f.pose[i][j];
    }
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
        // Block touch of a 16 x 16 tile of rays (DepthTouchCPU,
        // VoxelBlockGridCPU.cpp:144-180). The ~1000 candidate keys of a tile
        // are a few dozen distinct blocks: they are de-duplicated in an LDS
        // set first (LDS atomics, ~100 ns), and only the distinct keys go
        // through the chain of global atomics -- insert into the block hash,
        // frame bit in the touch word, list append -- one lane per key, all
        // keys of the tile in flight together. (One lane per ray walked that
        // chain four times in sequence, behind a wave-wide leader election
        // per candidate: ~20 us per wave, as long as the whole integrate
        // sweep it is supposed to hide behind.)
        __shared__ unsigned long long tile_keys[kTileKeys];
        for (int e = threadIdx.x; e < kTileKeys; e += blockDim.x)
            tile_keys[e] = kEmptyKey;
        __syncthreads();
        const int tiles_x = (p.cols_strided + 15) >> 4;
        const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
        const int rx = tx * 16 + (threadIdx.x & 15);
        const int ry = ty * 16 + (threadIdx.x >> 4);
        if (rx < p.cols_strided && ry < p.rows_strided) {
            int xb[4], yb[4], zb[4];
            if (RayCandidates(p, depth, ry * p.cols_strided + rx, xb, yb, zb)) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s > 0 && xb[s] == xb[s - 1] && yb[s] == yb[s - 1] &&
                        zb[s] == zb[s - 1])
                        continue;
                    if (!KeyInRange(xb[s], yb[s], zb[s])) {
                        atomicOr(&hv.counters[1], kErrKeyRange);
                        continue;
                    }
                    const unsigned long long k = PackKey(xb[s], yb[s], zb[s]);
                    if (!hv.Owns(k)) continue;  // another rank's block
                    unsigned h = HashKey(k) & (kTileKeys - 1);
                    while (true) {  // <= 1024 keys in 2048 slots: terminates
                        unsigned long long cur = tile_keys[h];
                        if (cur == k) break;
                        if (cur == kEmptyKey) {
                            cur = atomicCAS(&tile_keys[h], kEmptyKey, k);
                            if (cur == kEmptyKey || cur == k) break;
                        }
                        h = (h + 1) & (kTileKeys - 1);
                    }
                }
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < kTileKeys; e += blockDim.x) {
            const unsigned long long k = tile_keys[e];
            if (k == kEmptyKey) continue;
            const int x = (int)((k >> 42) & 0x1FFFFFull) - kKeyBias;
            const int y = (int)((k >> 21) & 0x1FFFFFull) - kKeyBias;
            const int z = (int)(k & 0x1FFFFFull) - kKeyBias;
            unsigned slot;
            InsertKey<true>(hv, x, y, z, slot, (int)fp.group_stamp);
            if (TouchSlot(hv, slot, fp.group_stamp, fp.group_bit,
                          fp.touch_plane)) {
                int o = atomicAdd(fp.out_count, 1);
                if (o < fp.list_capacity) {
                    // write-through (two 8-byte stores): the ready-list
                    // compaction below reads the entry from another
                    // workgroup of this launch
                    unsigned long long* e =
                            reinterpret_cast<unsigned long long*>(&list[o]);
                    __hip_atomic_store(
                            e, (unsigned long long)(unsigned)slot |
                                       ((unsigned long long)(unsigned)x << 32),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(
                            e + 1, (unsigned long long)(unsigned)y |
                                           ((unsigned long long)(unsigned)z << 32),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    atomicOr(&hv.counters[1], kErrCapacity);
                }
            }
        }
        // The ready list (stream_path.h ReadyEntry). The frame bits of a block
        // are final when EVERY touch workgroup of the group -- all its frames
        // -- has finished, so the workgroups take a ticket (their entries,
        // buffer indices and touch words are write-through stores / atomics,
        // drained first) and the last arrival turns the list into ready
        // entries: one lane per block, list entry -> buffer index + touch
        // word (the two dependent round trips an integrate work item would
        // otherwise spend on its header) -> one 16-byte entry. This happens
        // at the tail of the front roles, which finish early in a fused
        // launch; the entries are read in the NEXT launch.
        if (fp.ready &&
            LastArrival(fp.tickets, fp.group_bit * n_touch_wg + wg,
                        fp.n_touch_total)) {
            int n = __hip_atomic_load(fp.out_count, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
            if (n > fp.list_capacity) n = (int)fp.list_capacity;
            // A group that ran out of buffer indices -- or comes after one
            // that did -- is dropped as a whole: its integrate role finds an
            // empty list, the host reserves and replays from the first
            // dropped group (frames must be applied in order).
            const int overflow = __hip_atomic_load(
                    &hv.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (overflow != 0) {
                n = 0;
                if (threadIdx.x == 0)
                    __hip_atomic_store(fp.out_count, 0, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            if (threadIdx.x == 0 && fp.touch_status) {
                const int top = __hip_atomic_load(&hv.counters[0],
                                                  __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
                fp.touch_status[0] = top < hv.capacity ? top : hv.capacity;
                fp.touch_status[1] = overflow;
                fp.touch_status[2] = n;
                __hip_atomic_store(&fp.touch_status[3], (int)fp.group_stamp,
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            // One lane per block: list entry -> buffer index + touch word.
            const auto make_entry = [&](int i) -> ReadyEntry {
                const unsigned long long* e =
                        reinterpret_cast<const unsigned long long*>(&list[i]);
                const unsigned long long lo = __hip_atomic_load(
                        e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(
                        e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned slot = (unsigned)lo;
                const int x = (int)(unsigned)(lo >> 32);
                const int y = (int)(unsigned)hi, z = (int)(unsigned)(hi >> 32);
                const int idx = __hip_atomic_load(&hv.slot_vals[slot],
                                                  __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long word = __hip_atomic_load(
                        TouchWord(hv, slot, fp.touch_plane), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT);
                const bool own = (word >> kTouchBits) == fp.group_stamp;
                if (!own) atomicOr(&hv.counters[1], kErrTouchStamp);
                ReadyEntry re;
                re.key = PackKey(x, y, z);
                re.block_idx = idx;
                re.bits = own ? (unsigned)(word & ((1ull << kTouchBits) - 1ull))
                              : 0u;
                return re;
            };
            // LONGEST FIRST (round 5): the integrate role takes one workgroup
            // per (entry, part) in list order, and an entry's work is its
            // number of frames -- 1 to 12 of them. In arrival order the long
            // items start anywhere and the launch ends on the last of them;
            // a counting sort by frame count (the dispatcher's in-order issue
            // then is longest-processing-time-first scheduling, as for the
            // chunk launch of the sliced path) lets the short items fill the
            // tail. Up to kSortable entries (4 per lane, all their loads in
            // flight at once); a larger group keeps arrival order. Blocks are
            // independent: the order changes nothing but time.
            constexpr int kPerLane = 4;
            const int kSortable = kPerLane * (int)blockDim.x;
            __shared__ int s_bins[kTouchBits + 2];
            if (n <= kSortable) {
                if (threadIdx.x < kTouchBits + 2) s_bins[threadIdx.x] = 0;
                __syncthreads();
                ReadyEntry mine[kPerLane];
                int pc[kPerLane];
#pragma unroll
                for (int k = 0; k < kPerLane; ++k) {
                    const int i = (int)threadIdx.x + k * (int)blockDim.x;
                    pc[k] = -1;
                    if (i < n) {
                        mine[k] = make_entry(i);
                        pc[k] = __popc(mine[k].bits);
                    }
                }
#pragma unroll
                for (int k = 0; k < kPerLane; ++k)
                    if (pc[k] >= 0) atomicAdd(&s_bins[pc[k]], 1);
                __syncthreads();
                if (threadIdx.x == 0) {
                    // start of every bin, most frames first
                    int run = 0;
                    for (int b = kTouchBits; b >= 0; --b) {
                        const int c = s_bins[b];
                        s_bins[b] = run;
                        run += c;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < kPerLane; ++k)
                    if (pc[k] >= 0)
                        fp.ready[atomicAdd(&s_bins[pc[k]], 1)] = mine[k];
            } else {
                for (int i = threadIdx.x; i < n; i += blockDim.x)
                    fp.ready[i] = make_entry(i);
            }
        }
        return;
    }
    // Prepare pass: the voxel-independent sub-expressions of the integrate
    // lambda (VoxelBlockGridImpl.h:258-262 depth, :277-289 colour pixel).
    const int n_px = p.rows * p.cols;
    const int n_wg = fp.n_prep_wg;
    // sentinel behind the image: what a voxel outside the image reads
    if (wg == n_touch_wg && threadIdx.x == 0) {
        PixelRec r;
        r.d = 0.0f;
        r.rgba = 0u;
        recs[n_px] = r;
    }
    if (fp.col_lut && fp.prep_identity) {
        // Identity form (the tables map every depth pixel onto the colour
        // pixel of the same coordinates: same intrinsics and image size --
        // the usual RGB-D pair): 4 consecutive pixels per step as ONE 8-byte
        // depth load, three 4-byte colour loads and two 16-byte record stores,
        // 16 pixels per lane with all loads of the lane in flight together.
        // A prepare wave then lives for one memory round trip instead of
        // several dependent ones (it runs in a 128-register slot of the fused
        // launch, so its life is what it costs the integrate sweep).
        const int n_groups = n_px >> 2;  // cols % 4 == 0 (checked on the host)
        const int gstep = n_wg * blockDim.x;
        constexpr int kG = 4;
        for (int g0 = (wg - n_touch_wg) * blockDim.x + threadIdx.x;
             g0 < n_groups; g0 += kG * gstep) {
            uint2 dq[kG];
            unsigned cw[kG][3];
#pragma unroll
            for (int k = 0; k < kG; ++k) {
                const int g = g0 + k * gstep;
                const int gc = g < n_groups ? g : n_groups - 1;
                dq[k] = *reinterpret_cast<const uint2*>(depth + 4 * (int64_t)gc);
                if (pp.with_color) {
                    const unsigned* c = reinterpret_cast<const unsigned*>(
                            color + 12 * (int64_t)gc);
                    cw[k][0] = c[0];
                    cw[k][1] = c[1];
                    cw[k][2] = c[2];
                }
            }
#pragma unroll
            for (int k = 0; k < kG; ++k) {
                const int g = g0 + k * gstep;
                if (g >= n_groups) continue;
                const unsigned dv[4] = {dq[k].x & 0xffffu, dq[k].x >> 16,
                                        dq[k].y & 0xffffu, dq[k].y >> 16};
                unsigned rgba[4] = {0u, 0u, 0u, 0u};
                if (pp.with_color) {
                    // 12 bytes r0 g0 b0 r1 g1 b1 r2 g2 b2 r3 g3 b3
                    rgba[0] = (cw[k][0] & 0xffffffu) | (1u << 24);
                    rgba[1] = ((cw[k][0] >> 24) | ((cw[k][1] & 0xffffu) << 8)) |
                              (1u << 24);
                    rgba[2] = ((cw[k][1] >> 16) | ((cw[k][2] & 0xffu) << 16)) |
                              (1u << 24);
                    rgba[3] = (cw[k][2] >> 8) | (1u << 24);
                }
                uint4 o[2];
                unsigned w[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float df = (float)dv[j];
                    const float d =
                            fp.depth_div_short
                                    ? DivByConst(df, p.depth_scale,
                                                 fp.inv_depth_scale)
                                    : df / p.depth_scale;
                    w[2 * j] = __float_as_uint(d);
                    w[2 * j + 1] = rgba[j];
                }
                o[0] = make_uint4(w[0], w[1], w[2], w[3]);
                o[1] = make_uint4(w[4], w[5], w[6], w[7]);
                uint4* dst = reinterpret_cast<uint4*>(recs + 4 * (int64_t)g);
                dst[0] = o[0];
                dst[1] = o[1];
            }
        }
        return;
    }
    if (fp.col_lut) {
        // Table form: no division per pixel (the per-column / per-row parts
        // come from PrepTables, the depth division takes the short form when
        // the host verified it for every uint16 depth). Four pixels per lane
        // with every load of a stage requested before the first is used
        // (clamped indices instead of branches; only the store is
        // predicated).
        const int* __restrict__ col_lut = fp.col_lut;
        const int* __restrict__ row_lut = fp.row_lut;
        const int step = n_wg * blockDim.x;
        const int step_v = step / p.cols, step_u = step - step_v * p.cols;
        for (int first = (wg - n_touch_wg) * blockDim.x + threadIdx.x;
             first < n_px; first += 4 * step) {
            int idx[4], uc[4], vc[4];
            float df[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = first + k * step;
                idx[k] = i < n_px ? i : n_px - 1;
                df[k] = (float)depth[idx[k]];
            }
            if (pp.with_color) {
                int vi = first / p.cols;
                int ui = first - vi * p.cols;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // (a clamped tail pixel may read another pixel's table
                    // entries: its record is not stored)
                    uc[k] = col_lut[ui];
                    vc[k] = row_lut[vi < p.rows ? vi : p.rows - 1];
                    ui += step_u;
                    vi += step_v;
                    if (ui >= p.cols) {
                        ui -= p.cols;
                        ++vi;
                    }
                }
            }
            unsigned rgba[4] = {0u, 0u, 0u, 0u};
            if (pp.with_color) {
                unsigned c0[4], c1[4], c2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool has = (uc[k] | vc[k]) >= 0;
                    const uint8_t* in =
                            color + (has ? ((int64_t)vc[k] * pp.color_cols +
                                            uc[k]) * 3
                                         : 0);
                    c0[k] = in[0];
                    c1[k] = in[1];
                    c2[k] = in[2];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    rgba[k] = (uc[k] | vc[k]) >= 0
                                      ? (c0[k] | (c1[k] << 8) | (c2[k] << 16) |
                                         (1u << 24))
                                      : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                PixelRec r;
                r.d = fp.depth_div_short
                              ? DivByConst(df[k], p.depth_scale,
                                           fp.inv_depth_scale)
                              : df[k] / p.depth_scale;
                r.rgba = rgba[k];
                if (first + k * step < n_px) recs[idx[k]] = r;
            }
        }
        return;
    }
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
    // the group's frames share intrinsics and scale (cam0's, with frame 0's
    // extrinsic), only the extrinsics differ
    Camera cam0;  // depth intrinsics + extrinsic of frame 0, scale = voxel_size
    float ext[kMaxGroup][3][4];
    const PixelRec* recs[kMaxGroup];
    int n_frames;
    unsigned long long group_stamp;
    int touch_plane;
    int rows, cols, resolution;
    int res_shift;  // log2(resolution) when it is a power of two, else -1
    int cube;       // 1: a wave's lanes cover a compact cube of the block
    float sdf_trunc, depth_max;
    float inv_sdf_trunc;  // RN(1 / sdf_trunc), used by the kFastDiv variant
    const FrameBlock* list;
    const ReadyEntry* ready;  // null: header from list + hash (form 0, A / B)
    const int* count;
    int64_t list_capacity;
    float* tsdf;
    void* weight;
    void* color;
    int* zero_counter;
    int* size_host;
    int status_stamp;
    int* prof_count;
    unsigned long long* prof_items;  // kLong timeline: 4 words per (entry, part)
    int* prof_frame_blocks;
    int* prof_map_size;
    // RAW form (sliced block-ownership path, sliced_path.h): no prepared
    // records -- depth / colour are gathered from the frames' own images
    // (same size and intrinsics for both: the identity case of PrepTables)
    const uint16_t* raw_depth[kMaxGroup];
    const uint8_t* raw_color[kMaxGroup];
    float depth_scale, inv_depth_scale;
    bool depth_div_short;
    // LONG form (one launch per chunk of the sliced path): the work list
    // carries one bit per frame of the chunk, the frames' poses and images come
    // from a device table (n_frames <= kChunkFrames of them)
    const ChunkEntry* entries;
    const IntegFrame* frame_tab;
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
__device__ __forceinline__ bool RcpOutOfRange(float z) {
    return (__float_as_uint(z) - 0x21800000u) >= (0x5E000000u - 0x21800000u);
}
// RcpGuarded's short path alone (the caller has checked the range)
template <int kSteps>
__device__ __forceinline__ float RcpNewton(float z) {
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

// kDiv: 0 = IEEE divisions; 2 = the short sdf / trunc, 1 / (w + 1) and 1 / z
// (one Newton step) forms, used once the on-device proof of all three is in.
// ---- integrate role ---------------------------------------------------------
// Same arithmetic, different schedule and instruction selection. The role is
// bound by vector-ALU issue (profiles/r2a: ~55 % of the SIMD cycles issue VALU
// work, HBM traffic is a third of what the fabric can carry), so the form
// below is about instructions per voxel and about keeping the SIMDs fed:
//   1. every load of a work item is in flight at once: the voxel state (3
//      vector loads) is requested first, then for every frame of the group
//      the lane's 4 voxels are projected and their 8-byte records requested,
//      and only then the frames are applied, in frame order, from registers
//      (the form above walks gather -> lazy state load -> gather -> ...);
//   2. the float32 multiplies / adds / fused steps run as PACKED pairs
//      (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two voxels per instruction
//      at the same issue cost) -- lane-wise the very same IEEE operations in
//      the same order, so results do not change;
//   3. selects instead of exec-mask branches; a voxel that projects outside
//      the image reads a sentinel record (depth 0 = invalid) instead of
//      carrying a predicate; uint16 weights / colours are widened to float
//      once per work item and re-created in float form between the frames;
//   4. work items are dealt so that the parts of one block run on ONE XCD
//      (block b -> XCD b mod 8; observed dispatch: workgroup w runs on XCD
//      w mod 8): the parts share most of their projected pixel footprint,
//      which then sits in that XCD's L2 once instead of in four of them.
//      (Dealing whole neighbourhoods of blocks to one XCD -- per-class
//      sublists keyed by block position -- cut the fabric reads by 17 % and
//      cost 50 % in time: with ~760 blocks per group, classes that keep
//      neighbours together are too uneven. Measured and dropped, r2n.)
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 Splat(float a) { return f2{a, a}; }
__device__ __forceinline__ f2 PkFma(f2 a, f2 b, f2 c) {
    return __builtin_elementwise_fma(a, b, c);
}

// kP = voxel pairs per lane: 1 (2 x-consecutive voxels per lane, 72 registers,
// 7 waves per SIMD; 4 voxels per lane -- kP = 2, ~127 registers -- measured 9 %
// slower, profiles/r2q, and is no longer instantiated).
// kRaw: depth and colour come from the frames' raw uint16 / uint8 images
// instead of the prepared 8-byte records (IntegParams::raw_depth / raw_color): one
// 2-byte and one (unaligned) 4-byte gather per voxel and frame, the depth
// division float(depth) / depth_scale per voxel instead of per pixel -- the
// same float32 operations on the same operands, so the results are the
// records path's bit for bit. For ranks that integrate a fraction of the
// blocks (block-ownership sharding) this removes the per-pixel prepare pass,
// which every rank would otherwise run in full.
// kLong (with kRaw): the chunk launch of the sliced path -- a work item applies
// up to kChunkFrames frames (ChunkEntry::bits) to its register-resident voxels,
// kChunk at a time; per-frame constants come from IntegParams::frame_tab.
template <typename weight_t, typename color_t, bool kColor, int kDiv,
          int kChunk, int kP, bool kRaw = false, bool kLong = false>
__device__ __forceinline__ void IntegrateRoleWide(const HashView& hv,
                                                  const IntegParams& ip,
                                                  int wg, int n_wg,
                                                  int first_wg) {
    constexpr int kV = 2 * kP;             // voxels per lane
    constexpr int kVShift = kP == 2 ? 2 : 1;
    // kLong: the frame table is read through the CONSTANT address space and
    // the images through the GLOBAL one. A plain pointer makes the table's
    // loads vector loads (stores of earlier work items may alias it as far as
    // the compiler knows) -- one more dependent trip through the vector
    // memory queue per round, and in order with the gathers -- and the image
    // pointers, coming out of memory, generic: flat loads.
    using FrameTabC = const IntegFrame __attribute__((address_space(4)));
    using ByteG = const char __attribute__((address_space(1)));
    using U16G = const uint16_t __attribute__((address_space(1)));
    typedef unsigned U2v __attribute__((ext_vector_type(2)));
    using U2G = const U2v __attribute__((address_space(1)));
    FrameTabC* const ftab = (FrameTabC*)ip.frame_tab;
    using TVec = Vec<float, kV, 4 * kV>;
    using WVec = Vec<weight_t, kV, kV * sizeof(weight_t)>;
    using CVec = Vec<color_t, 3 * kV, kV * sizeof(color_t)>;
    constexpr bool kU16 = sizeof(weight_t) == 2;
    float* __restrict__ tsdf_base = ip.tsdf;
    weight_t* __restrict__ weight_base = (weight_t*)ip.weight;
    color_t* __restrict__ color_base = (color_t*)ip.color;
    const FrameBlock* __restrict__ list = ip.list;
    int64_t n_blocks = *ip.count;
    if (n_blocks > ip.list_capacity) n_blocks = ip.list_capacity;

    if (wg == 0 && threadIdx.x == 0) {
        if (ip.zero_counter) *ip.zero_counter = 0;
        if (ip.prof_count) *ip.prof_count = (int)n_blocks;
        if (ip.prof_map_size) *ip.prof_map_size = hv.counters[0];
        if (ip.size_host) {
            ip.size_host[0] = hv.counters[0];
            ip.size_host[1] = hv.counters[1];
            ip.size_host[2] = (int)n_blocks;
            __hip_atomic_store(&ip.size_host[3], ip.status_stamp,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    const int res = ip.resolution;
    const int res3 = res * res * res;
    const int quads_per_row = res >> kVShift;
    const int n_quads = res3 >> kVShift;
    const int parts = (n_quads + 255) >> 8;
    const int res_shift = ip.res_shift;  // log2(res) or -1
    int frame_blocks = 0;  // lane 0 of part 0 counts block-frames

    // XCD-aware deal: this workgroup's XCD, its rank among the role's
    // workgroups of that XCD and their number.
    const int xcd = (first_wg + wg) & 7;
    const int first_of_xcd = (xcd - first_wg) & 7;  // lowest wg on this XCD
    const int rank = (wg - first_of_xcd) >> 3;
    const int n_on_xcd =
            first_of_xcd < n_wg ? ((n_wg - 1 - first_of_xcd) >> 3) + 1 : 0;
    // An XCD takes every eighth block (b = xcd + 8k). (A CONTIGUOUS eighth of
    // the list -- roughly a band of the image per XCD, whose record rows then
    // sit in one L2 -- read 4 % fewer bytes and was 2 % slower: the eighths are
    // less even. profiles/r4m, dropped.)
    const int64_t first_b = xcd;
    const int64_t blocks_on_xcd = (n_blocks + 7 - xcd) >> 3;
    const int64_t n_items = blocks_on_xcd * parts;
    const unsigned sentinel_off =
            (unsigned)(ip.rows * ip.cols) * (unsigned)sizeof(PixelRec);
    const unsigned row_bytes = (unsigned)ip.cols * (unsigned)sizeof(PixelRec);
    const float vscale = ip.cam0.scale;
    const float fx = ip.cam0.fx, fyk = ip.cam0.fy;
    const float cx = ip.cam0.cx, cy = ip.cam0.cy;
    const unsigned u_max_bits = __float_as_uint(ip.cols - 1.0f);
    const unsigned v_max_bits = __float_as_uint(ip.rows - 1.0f);
    const unsigned last_col8 = (unsigned)(ip.rows * ip.cols) * 3u - 8u;

    // (Measured and dropped, profiles/r2l: persistent workgroups -- 1024 to
    // 1536 of them striding over the items, with the next item's block header
    // prefetched behind the current item's gathers -- are 8 % SLOWER than one
    // workgroup per item: the dispatcher's dynamic assignment balances the
    // uneven item times better than a static stride, and the prefetched
    // header costs registers the colour form does not have.)
    for (int64_t m = rank; m < n_items; m += n_on_xcd) {
        int64_t kb;
        int part;
        if (res_shift >= 0) {  // parts is a power of two as well
            const int ps0 = 3 * res_shift - kVShift - 8;
            const int ps = ps0 > 0 ? ps0 : 0;
            kb = m >> ps;
            part = (int)(m & ((1 << ps) - 1));
        } else {
            kb = m / parts;
            part = (int)(m - kb * parts);
        }
        const int64_t b = first_b + (kb << 3);
        unsigned long long item_t0 = 0ull;
        if constexpr (kLong)
            if (ip.prof_items && threadIdx.x == 0) item_t0 = wall_clock64();
        int xb, yb, zb, block_idx;
        unsigned bits;
        unsigned long_bits = 0u;  // kLong: lane l holds word l & 7
        if constexpr (kLong) {
            const ChunkEntry* __restrict__ ce = ip.entries + b;
            const unsigned klo =
                    __builtin_amdgcn_readfirstlane((unsigned)ce->key);
            const unsigned khi =
                    __builtin_amdgcn_readfirstlane((unsigned)(ce->key >> 32));
            const unsigned long long key =
                    ((unsigned long long)khi << 32) | klo;
            xb = (int)((key >> 42) & 0x1FFFFFull) - kKeyBias;
            yb = (int)((key >> 21) & 0x1FFFFFull) - kKeyBias;
            zb = (int)(key & 0x1FFFFFull) - kKeyBias;
            block_idx = __builtin_amdgcn_readfirstlane(ce->block_idx);
            // lane l keeps word l & 7 of the frame bits: a round takes its
            // word with one v_readlane instead of a memory round trip
            long_bits = ce->bits[threadIdx.x & (kChunkWords - 1)];
            unsigned any = 0u;
#pragma unroll
            for (int w = 0; w < kChunkWords; ++w) {
                const unsigned bw =
                        (unsigned)__builtin_amdgcn_readlane((int)long_bits, w);
                any |= bw;
                if (part == 0 && threadIdx.x == 0 && ip.prof_frame_blocks)
                    frame_blocks += __popc(bw);
            }
            bits = any;  // "any frame at all"
            // The launch lasts as long as its longest work item (a block seen
            // by all frames of the chunk applies them one round after the
            // other while its SIMD is shared with ~5 other waves): items with
            // many frames get the issue priority, items with few fill in.
            {
                int n_set = 0;
#pragma unroll
                for (int w = 0; w < kChunkWords; ++w)
                    n_set += __popc((unsigned)__builtin_amdgcn_readlane(
                            (int)long_bits, w));
                if (n_set >= 3 * ip.n_frames / 4) __builtin_amdgcn_s_setprio(3);
                else if (n_set >= ip.n_frames / 2) __builtin_amdgcn_s_setprio(2);
                else if (n_set >= ip.n_frames / 4) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
        } else if (ip.ready) {
            // ONE round trip: the ready entry the group's front roles left
            const ReadyEntry re = ip.ready[b];
            const unsigned klo = __builtin_amdgcn_readfirstlane((unsigned)re.key);
            const unsigned khi =
                    __builtin_amdgcn_readfirstlane((unsigned)(re.key >> 32));
            const unsigned long long key =
                    ((unsigned long long)khi << 32) | klo;
            xb = (int)((key >> 42) & 0x1FFFFFull) - kKeyBias;
            yb = (int)((key >> 21) & 0x1FFFFFull) - kKeyBias;
            zb = (int)(key & 0x1FFFFFull) - kKeyBias;
            block_idx = __builtin_amdgcn_readfirstlane(re.block_idx);
            bits = __builtin_amdgcn_readfirstlane(re.bits);
        } else {
            const FrameBlock fb = list[b];
            const int slot = __builtin_amdgcn_readfirstlane(fb.slot);
            xb = __builtin_amdgcn_readfirstlane(fb.x);
            yb = __builtin_amdgcn_readfirstlane(fb.y);
            zb = __builtin_amdgcn_readfirstlane(fb.z);
            block_idx = __builtin_amdgcn_readfirstlane(hv.slot_vals[slot]);
            const unsigned long long word =
                    *TouchWord(hv, slot, ip.touch_plane);
            const bool own = (word >> kTouchBits) == ip.group_stamp;
            if (!own && threadIdx.x == 0 && part == 0)
                atomicOr(&hv.counters[1], kErrTouchStamp);
            bits = __builtin_amdgcn_readfirstlane(
                    own ? (unsigned)(word & ((1ull << kTouchBits) - 1ull))
                        : 0u);
        }
        const int64_t block_base = (int64_t)block_idx * res3;
        if (!kLong && part == 0 && threadIdx.x == 0 && ip.prof_frame_blocks)
            frame_blocks += __popc(bits);
        int opaque = 0;  // a zero the optimiser cannot see through
        asm volatile("" : "+s"(opaque));

        const int q = (part << 8) + threadIdx.x;
        if (q >= n_quads || bits == 0u) continue;
        int qx, yv, zv;
        int lin_q = q;  // index of the lane's first voxel / kV in the block
        if (res_shift - kVShift >= 2 && ip.cube) {
            // Lane -> voxel by a permutation of q's bits: the 64 lanes of a
            // wave take a compact (4 << kVShift) x 4 x 4 cube of the block
            // instead of a (res x 8 x 1) slab, so that one gather instruction
            // touches the image rows of a 4-voxel-high patch, not those of a
            // slab res voxels long (the vector memory pipeline takes a cycle
            // per distinct cache line of an instruction). Any bijection gives
            // the same grid: voxels are independent.
            const int nx = res_shift - kVShift;  // bits of qx (>= 2: res >= 8)
            const unsigned uq = (unsigned)q;
            unsigned rest = uq >> 6;
            const unsigned qx_hi = rest & ((1u << (nx - 2)) - 1u);
            rest >>= (nx - 2);
            const unsigned y_hi = rest & ((1u << (res_shift - 2)) - 1u);
            rest >>= (res_shift - 2);
            qx = (int)((uq & 3u) | (qx_hi << 2));
            yv = (int)(((uq >> 2) & 3u) | (y_hi << 2));
            zv = (int)(((uq >> 4) & 3u) | (rest << 2));
            lin_q = (((zv << res_shift) | yv) << nx) | qx;
        } else if (res_shift >= 0) {
            qx = q & (quads_per_row - 1);
            const int row = q >> (res_shift - kVShift);
            yv = row & (res - 1);
            zv = row >> res_shift;
        } else {
            qx = q % quads_per_row;
            const int row = q / quads_per_row;
            yv = row % res;
            zv = row / res;
        }
        const int x0 = xb * res + (qx << kVShift);
        const float fy = (float)(yb * res + yv);
        const float fz = (float)(zb * res + zv);
        const int64_t lin0 = block_base + ((int64_t)lin_q << kVShift);

        // 1. voxel state, widened to float once per work item. A uint16
        // weight / colour is an exact float; between the frames of the group
        // the stored value is re-created in float form (truncation toward
        // zero = the float -> uint16 store conversion of a non-negative value
        // below 65536; the weight's wrap at 65536 = what the uint16 store
        // keeps of it). Voxel pair p = voxels 2p, 2p + 1 of the lane.
        const TVec t4 = *reinterpret_cast<const TVec*>(tsdf_base + lin0);
        float ts[kV];
        float wf[kV];
        float cf[kV][3];
#pragma unroll
        for (int v = 0; v < kV; ++v) ts[v] = t4.v[v];
        {
            const WVec w4 = *reinterpret_cast<const WVec*>(weight_base + lin0);
#pragma unroll
            for (int v = 0; v < kV; ++v) wf[v] = (float)w4.v[v];
            if constexpr (kColor) {
                const CVec c12 =
                        *reinterpret_cast<const CVec*>(color_base + 3 * lin0);
#pragma unroll
                for (int v = 0; v < kV; ++v)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        cf[v][i] = (float)c12.v[3 * v + i];
            }
        }

        // 2. projections + record requests of every frame.
        // Camera::RigidTransform (GeometryIndexer.h:62-78): x * scale, then
        // ((x e0 + y e1) + z e2) + e3 per row; the y and z products are the
        // same for the lane's 4 voxels.
        const float ys = fy * vscale, zs = fz * vscale;
        f2 xs[kP];
#pragma unroll
        for (int p = 0; p < kP; ++p)
            xs[p] = f2{(float)(x0 + 2 * p), (float)(x0 + 2 * p + 1)} * vscale;
        // (kChunk frames at a time: a group of up to kMaxGroup frames is
        // applied chunk after chunk to the same register-resident state; the
        // chunk loop is a real loop, so that the second chunk's gathers are
        // not hoisted above the first chunk's arithmetic -- they would double
        // the live registers)
        bool touched = false;
        // One ROUND = kChunk frames: `issue` projects the lane's voxels into
        // the round's frames and requests their records, `apply` applies the
        // frames in order from registers.
        struct Round {
            f2 zc[kChunk][kP];
            PixelRec rec[kChunk][kV];
            // kRaw: upper half of the 8 colour bytes, bit offset of the pixel
            unsigned chi[kRaw && kColor ? kChunk : 1][kV];
            unsigned csh[kRaw && kColor ? kChunk : 1][kV];
            unsigned in_mask;  // kRaw: voxel projects into the image
            unsigned cbits;    // frames of the round that touch the block
        };
        auto round_bits = [&](int c0) -> unsigned {
            // (kChunk divides 32)
            if constexpr (kLong)
                return ((unsigned)__builtin_amdgcn_readlane((int)long_bits,
                                                            c0 >> 5) >>
                        (c0 & 31)) &
                       ((1u << kChunk) - 1u);
            else
                return (bits >> c0) & ((1u << kChunk) - 1u);
        };
        auto issue = [&](int c0, Round& R) {
        R.in_mask = 0u;
#pragma unroll
        for (int fk = 0; fk < kChunk; ++fk) {
            const int f = c0 + fk;
            if (!((R.cbits >> fk) & 1u)) continue;  // wave-uniform
            // The frame's constants are fetched here, per work item (scalar
            // loads from the argument block / the frame table): hoisted out of
            // the item loop they would occupy ~60 scalar registers and spill.
            float e_tab[3][4];
            if constexpr (kLong) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) e_tab[i][j] = ftab[f].ext[i][j];
            }
            const float(&e)[3][4] =
                    kLong ? e_tab
                          : *reinterpret_cast<const float(*)[3][4]>(
                                    &ip.ext[kLong ? 0 : f][0][0] + opaque);
            const char* __restrict__ recs = reinterpret_cast<const char*>(
                    kRaw ? nullptr
                         : (kLong ? (const void*)ftab[f].recs
                                  : *(&ip.recs[kLong ? 0 : f] + opaque)));
            ByteG* __restrict__ dimg = (ByteG*)(
                    kLong ? (const void*)ftab[f].depth
                          : (kRaw ? *(&ip.raw_depth[kLong ? 0 : f] + opaque)
                                  : nullptr));
            ByteG* __restrict__ cimg = (ByteG*)(
                    kLong ? (const void*)ftab[f].color
                          : (kRaw ? *(&ip.raw_color[kLong ? 0 : f] + opaque)
                                  : nullptr));
            const float y0 = ys * e[0][1], z0 = zs * e[0][2];
            const float y1 = ys * e[1][1], z1 = zs * e[1][2];
            const float y2 = ys * e[2][1], z2 = zs * e[2][2];
            f2 xc[kP], yc[kP];
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                xc[p] = xs[p] * e[0][0] + y0 + z0 + e[0][3];
                yc[p] = xs[p] * e[1][0] + y1 + z1 + e[1][3];
                R.zc[fk][p] = xs[p] * e[2][0] + y2 + z2 + e[2][3];
            }
            // Camera::Project's 1 / z: the verified short reciprocal unless a
            // lane of the wave is outside its range. z is monotone along the
            // lane's 4 voxels, so the two end voxels decide.
            f2 inv_z[kP];
            const bool out = RcpOutOfRange(R.zc[fk][0].x) ||
                             RcpOutOfRange(R.zc[fk][kP - 1].y);
            if (kDiv < 2 || __builtin_amdgcn_ballot_w64(out) != 0ull) {
#pragma unroll
                for (int p = 0; p < kP; ++p)
                    inv_z[p] = f2{1.0f / R.zc[fk][p].x, 1.0f / R.zc[fk][p].y};
            } else {
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    f2 r = f2{__builtin_amdgcn_rcpf(R.zc[fk][p].x),
                              __builtin_amdgcn_rcpf(R.zc[fk][p].y)};
                    r = PkFma(PkFma(-R.zc[fk][p], r, Splat(1.0f)), r, r);
                    inv_z[p] = r;
                }
            }
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                // u = fx * x * inv_z + cx (GeometryIndexer.h:100-108)
                const f2 u = xc[p] * fx * inv_z[p] + cx;
                const f2 v = yc[p] * fyk * inv_z[p] + cy;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float uh = h ? u.y : u.x, vh = h ? v.y : v.x;
                    // InBoundary (0 <= u <= W - 1, 0 <= v <= H - 1 on the
                    // floats, GeometryIndexer.h) as ONE unsigned compare per
                    // coordinate: non-negative floats order like their bit
                    // patterns, every negative float (sign bit), NaN and inf
                    // compare above the pattern of the non-negative bound.
                    // The one value that would differ, u = -0.0f (>= 0 in
                    // IEEE), needs a principal point of -0.0f, which the host
                    // turns into +0.0f (same sums bit for bit).
                    const bool in = __float_as_uint(vh) <= v_max_bits &&
                                    __float_as_uint(uh) <= u_max_bits;
                    // 32-bit byte offset from a wave-uniform base; the
                    // sentinel record (depth 0) for voxels outside the image
                    // (24-bit multiply: rows and the row pitch are far below
                    // 2^24, and it issues at full rate)
                    if constexpr (kRaw) {
                        // pixel index; lanes outside the image read pixel 0
                        // and carry a cleared bit in R.in_mask
                        const unsigned pix =
                                in ? __umul24((unsigned)(int)vh,
                                              (unsigned)ip.cols) +
                                             (unsigned)(int)uh
                                   : 0u;
                        R.in_mask |= (in ? 1u : 0u) << (fk * kV + 2 * p + h);
                        PixelRec r;
                        // .d holds the raw uint16 depth (converted below)
                        r.d = __uint_as_float(
                                (unsigned)*(U16G*)(dimg + 2u * pix));
                        r.rgba = 0u;
                        if constexpr (kColor) {
                            // the 3 bytes at 3 * pix out of ONE aligned 8-byte
                            // load (an unaligned 4-byte load is split by the
                            // memory pipeline); the address is clamped so that
                            // no load passes the end of the image
                            const unsigned b3 = 3u * pix;
                            unsigned al = b3 & ~3u;
                            al = al < last_col8 ? al : last_col8;
                            const U2v q = *(U2G*)(cimg + al);
                            r.rgba = q.x;
                            R.chi[fk][2 * p + h] = q.y;
                            R.csh[fk][2 * p + h] = (b3 - al) * 8u;
                        }
                        R.rec[fk][2 * p + h] = r;
                    } else {
                    unsigned off =
                            __umul24((unsigned)(int)vh, row_bytes) +
                            (unsigned)(int)uh * (unsigned)sizeof(PixelRec);
#if O3DMI_ABLATE_GATHER
                    // MEASUREMENT BUILD ONLY (tools/build_variant.sh, wrong
                    // results): every lane of the wave reads the record of
                    // its first lane -- one cache line per gather instruction
                    // -- to bound what any record-staging scheme could gain
                    off = __builtin_amdgcn_readfirstlane(in ? off
                                                            : sentinel_off);
                    R.rec[fk][2 * p + h] =
                            *reinterpret_cast<const PixelRec*>(recs + off);
#else
                    R.rec[fk][2 * p + h] = *reinterpret_cast<const PixelRec*>(
                            recs + (in ? off : sentinel_off));
#endif
                    }
                }
            }
        }

        };
        // kRaw: float(depth) / depth_scale (VoxelBlockGridImpl.h:258-262),
        // what the prepare pass does per pixel: the short constant-divisor
        // form when the host verified it for all 65536 depths
        auto convert_depth = [&](int fk, Round& R) {
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                f2 a = f2{(float)__float_as_uint(R.rec[fk][2 * p].d),
                          (float)__float_as_uint(R.rec[fk][2 * p + 1].d)};
                f2 q;
                if (ip.depth_div_short) {
                    const f2 q0 = a * ip.inv_depth_scale;
                    const f2 r = PkFma(Splat(-ip.depth_scale), q0, a);
                    q = PkFma(r, Splat(ip.inv_depth_scale), q0);
                } else {
                    q = f2{a.x / ip.depth_scale, a.y / ip.depth_scale};
                }
                // outside the image: depth 0 = invalid (the records path's
                // sentinel)
                R.rec[fk][2 * p].d =
                        ((R.in_mask >> (fk * kV + 2 * p)) & 1u) ? q.x : 0.0f;
                R.rec[fk][2 * p + 1].d =
                        ((R.in_mask >> (fk * kV + 2 * p + 1)) & 1u) ? q.y : 0.0f;
            }
        };
        auto apply = [&](int c0, Round& R) {
        // 3. frames applied in order (VoxelBlockGridImpl.h:258-302). The
        // update of a voxel runs under the voxel's own predicate (an
        // exec-masked region per voxel of the lane) instead of being computed
        // for every lane and selected. Why: on gfx950 a v_cndmask, a v_cmp, a
        // v_trunc or a conversion occupies the SIMD's issue for 4.3 cycles, a
        // plain float32 multiply / add / fma for 2.4, and a PACKED float32
        // instruction for 4.3 -- no cheaper than the two plain ones it
        // replaces (profiles/r5b_valu_calibration.json). Rounds 2-4 ran this
        // block as packed pairs with two selects per state value; the selects
        // were a quarter of its cycles. A region whose predicate is false in
        // every lane of the wave is jumped over (s_cbranch_execz): the
        // wave-level "no voxel takes this frame" test of the packed form is
        // implied. Same IEEE operations, operands and order: same bits.
#pragma unroll
        for (int fk = 0; fk < kChunk; ++fk) {
            if (!((R.cbits >> fk) & 1u)) continue;  // wave-uniform
            if constexpr (kRaw) convert_depth(fk, R);
#pragma unroll
            for (int p = 0; p < kP; ++p) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float dh = R.rec[fk][2 * p + h].d;
                    const float zh = h ? R.zc[fk][p].y : R.zc[fk][p].x;
                    const float sh = dh - zh;
                    const bool ok = !(dh <= 0) && !(dh > ip.depth_max) &&
                                    !(zh <= 0) && !(sh < -ip.sdf_trunc);
                    if (!ok) continue;
                    touched = true;
                    // sh < trunc ? sh : trunc as ONE v_min_f32 (2.4 cycles
                    // against 11 for move + compare + select): the same
                    // value for every input -- sh comes out of a subtraction,
                    // so it is no signalling NaN, and for a quiet NaN both
                    // forms give trunc; -0.0 stays -0.0.
                    float cl;
                    asm("v_min_f32 %0, %1, %2"  // src0 may be a scalar register
                        : "=v"(cl)
                        : "s"(ip.sdf_trunc), "v"(sh));
                    // sdf / sdf_trunc: the verified short form, the IEEE
                    // sequence in the underflow range
                    float sd;
                    if (kDiv < 1 || fabsf(cl) < kDivTiny)
                        sd = cl / ip.sdf_trunc;
                    else
                        sd = DivByConst(cl, ip.sdf_trunc, ip.inv_sdf_trunc);
                    const int vx = 2 * p + h;
                    const float weight = wf[vx];
                    const float wsum = weight + 1.0f;  // exact (<= 65536)
                    float inv_wsum;
                    if constexpr (kU16 && kDiv >= 1) inv_wsum = RcpSmallInt(wsum);
                    else inv_wsum = 1.0f / wsum;
                    const float t_new = (weight * ts[vx] + sd) * inv_wsum;
                    ts[vx] = t_new;
                    if constexpr (kColor) {
                        unsigned rg = R.rec[fk][2 * p + h].rgba;
                        if constexpr (kRaw)
                            rg = (unsigned)((((unsigned long long)
                                                      R.chi[fk][2 * p + h]
                                              << 32) | rg) >>
                                            R.csh[fk][2 * p + h]);
                        // (raw form: the colour pixel IS the depth pixel,
                        // inside the image whenever the voxel is ok)
                        if (kRaw || (rg >> 24)) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                const float in =
                                        (float)((rg >> (8 * i)) & 0xffu);
                                float c_new =
                                        (weight * cf[vx][i] + in) * inv_wsum;
                                if constexpr (sizeof(color_t) == 2)
                                    c_new = truncf(c_new);
                                cf[vx][i] = c_new;
                            }
                        }
                    }
                    wf[vx] = wsum;
                    if constexpr (kU16) {
                        // what the uint16 store keeps of 65536 (a branch that
                        // is taken once in 65536 frames, not a select)
                        if (wsum >= 65536.0f) wf[vx] = 0.0f;
                    }
                }
            }
        }
        };
        if constexpr (kLong) {
#pragma nounroll
            for (int c0 = 0; c0 < ip.n_frames; c0 += kChunk) {
                Round r;
                r.cbits = round_bits(c0);
                if (r.cbits == 0u) continue;  // wave-uniform
                issue(c0, r);
                apply(c0, r);
            }
        } else {
            // (kChunk frames at a time: a group of up to kMaxGroup frames is
            // applied chunk after chunk to the same register-resident state;
            // the chunk loop is a real loop, so that the second chunk's
            // gathers are not hoisted above the first chunk's arithmetic --
            // they would double the live registers)
#pragma nounroll
            for (int c0 = 0; c0 < kMaxGroup; c0 += kChunk) {
                Round r;
                r.cbits = round_bits(c0);
                if (r.cbits == 0u) continue;  // wave-uniform
                issue(c0, r);
                apply(c0, r);
            }
        }
        if (touched) {
            TVec t_out;
            WVec w4;
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                t_out.v[2 * p] = ts[2 * p];
                t_out.v[2 * p + 1] = ts[2 * p + 1];
                w4.v[2 * p] = (weight_t)wf[2 * p];
                w4.v[2 * p + 1] = (weight_t)wf[2 * p + 1];
            }
            *reinterpret_cast<TVec*>(tsdf_base + lin0) = t_out;
            *reinterpret_cast<WVec*>(weight_base + lin0) = w4;
            if constexpr (kColor) {
                CVec c12;
#pragma unroll
                for (int p = 0; p < kP; ++p)
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        c12.v[6 * p + i] = (color_t)cf[2 * p][i];
                        c12.v[6 * p + 3 + i] = (color_t)cf[2 * p + 1][i];
                    }
                *reinterpret_cast<CVec*>(color_base + 3 * lin0) = c12;
            }
        }
        if constexpr (kLong) {
            // O3DMI_CHUNK_TIMELINE: {start, end} on the 100 MHz clock, the
            // entry's frame count and where the workgroup ran
            if (ip.prof_items && threadIdx.x == 0) {
                int n_set = 0;
#pragma unroll
                for (int w = 0; w < kChunkWords; ++w)
                    n_set += __popc((unsigned)__builtin_amdgcn_readlane(
                            (int)long_bits, w));
                unsigned long long* o = ip.prof_items + (b * parts + part) * 4;
                o[0] = item_t0;
                o[1] = wall_clock64();
                o[2] = ((unsigned long long)n_set << 32) |
                       (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
                o[3] = ((unsigned long long)(unsigned)xcd << 32) | (unsigned)wg;
            }
        }
    }
    if (frame_blocks) atomicAdd(ip.prof_frame_blocks, frame_blocks);
}

struct StepParams {
    HashView hv;
    FrontShared fshared;
    FrontFrame front[kMaxGroup];
    IntegParams integ;
    int n_fronts;
    int front_wg;  // workgroups per front role
};
static_assert(sizeof(StepParams) <= 4096, "kernel arguments are limited to 4 KB");


// The launch of a frame group: front roles of the NEXT group + the integrate
// role of this one (2 voxels per lane: 72 registers, 7 waves per SIMD), which
// applies the group's frames in chunks of kGroupChunk = 4 to the register-
// resident voxel state.
// Frames of a round of the raw / long form (the chunk launch of the sliced
// path): a rank's share of a chunk is a few hundred blocks, about one round of
// the chip, so a work item's chain of dependent memory round trips is what the
// launch lasts -- twice the frames in flight per round halve the rounds; the
// registers this costs (2 waves per SIMD allowed) are not needed for occupancy
// there.
#ifndef O3DMI_RAW_CHUNK
#define O3DMI_RAW_CHUNK 4
#endif
#ifndef O3DMI_RAW_WAVES
#define O3DMI_RAW_WAVES 5
#endif
constexpr int kRawChunk = O3DMI_RAW_CHUNK;
template <typename weight_t, typename color_t, bool kColor, int kDiv>
__global__ void __launch_bounds__(256, 7)
FrameStepKernel(StepParams sp) {
    const int b = (int)blockIdx.x;
    const int n_front_wg = sp.n_fronts * sp.front_wg;
    if (b < n_front_wg) {
        const int f = b / sp.front_wg;
        const FrontParams fp(sp.fshared, sp.front[f]);
        FrontRole(sp.hv, fp, b - f * sp.front_wg);
    } else {
        IntegrateRoleWide<weight_t, color_t, kColor, kDiv, kGroupChunk, 1>(
                sp.hv, sp.integ, b - n_front_wg, (int)gridDim.x - n_front_wg,
                n_front_wg);
    }
}

// The chunk launch of the sliced block-ownership path (sliced_path.h): the
// integrate role in its raw-image, long form -- nothing else in the launch.
struct ChunkParams {
    HashView hv;
    IntegParams integ;
};
// kRaw = false: the frames' prepared records (one 8-byte gather per voxel and
// frame, the per-group role's registers and occupancy); true: raw images.
// (Software-pipelined rounds -- the next round's gathers in flight during this
// round's arithmetic -- cost two waves of occupancy and were 0-10 % slower:
// profiles/r4p, dropped.)
template <typename weight_t, typename color_t, bool kColor, int kDiv,
          bool kRaw>
__global__ void __launch_bounds__(256, kRaw ? O3DMI_RAW_WAVES : 7)
ChunkIntegrateKernel(ChunkParams cp) {
    IntegrateRoleWide<weight_t, color_t, kColor, kDiv,
                      kRaw ? kRawChunk : kGroupChunk, 1, kRaw, true>(
            cp.hv, cp.integ, (int)blockIdx.x, (int)gridDim.x, 0);
}

}  // namespace

// The compact lane -> voxel map of the wide integrate role (default; needs a
// power-of-two resolution >= 8, else the slab map; against the slab map:
// profiles/r4zf -- chunk launch at 8 ranks 486 k -> 525 k frames/s, single-GPU
// stream 127.6 k -> 128.6 k).
static int LaneCube(int res_shift) { return res_shift >= 3 ? 1 : 0; }

bool PrepTables(const double* depth_intrinsic, const double* color_intrinsic,
                int rows, int cols, int color_rows, int color_cols,
                float depth_scale, int* col, int* row) {
    // TransformIndexer keeps float copies (GeometryIndexer.h:46-58); the
    // expressions below are Unproject(u, v, 1) -> Project -> InBoundary ->
    // round exactly as the per-pixel form above evaluates them (this file is
    // compiled without FMA contraction, x86-64 float arithmetic is IEEE).
    const float fx = (float)depth_intrinsic[0], fy = (float)depth_intrinsic[4];
    const float cx = (float)depth_intrinsic[2], cy = (float)depth_intrinsic[5];
    const double* ck = color_intrinsic ? color_intrinsic : depth_intrinsic;
    const float fx2 = (float)ck[0], fy2 = (float)ck[4];
    const float cx2 = (float)ck[2], cy2 = (float)ck[5];
    volatile float one = 1.0f;  // d = 1, inv_z = 1 / 1: keep the operations
    const float inv_z = 1.0f / one;
    for (int u = 0; u < cols; ++u) {
        const float x = ((float)u - cx) * one / fx;
        const float uf = fx2 * x * inv_z + cx2;
        col[u] = (uf >= 0 && uf <= color_cols - 1.0f) ? (int)roundf(uf) : -1;
    }
    for (int v = 0; v < rows; ++v) {
        const float y = ((float)v - cy) * one / fy;
        const float vf = fy2 * y * inv_z + cy2;
        row[v] = (vf >= 0 && vf <= color_rows - 1.0f) ? (int)roundf(vf) : -1;
    }
    if (!(depth_scale > 0.0f) || !std::isfinite(depth_scale)) return false;
    static const bool exact_div = std::getenv("O3DMI_EXACT_DIV") != nullptr;
    if (exact_div) return false;
    const float y = 1.0f / depth_scale;
    for (int dv = 0; dv < 65536; ++dv) {
        const float a = (float)dv;
        const float q0 = a * y;
        const float r = std::fmaf(-depth_scale, q0, a);
        const float q = std::fmaf(r, y, q0);
        const float want = a / depth_scale;
        if (std::memcmp(&q, &want, sizeof(float)) != 0) return false;
    }
    return true;
}

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
//
// The proof is ~3 x 10^9 divisions (a few ms of the whole chip) and runs
// ASYNCHRONOUSLY on a private stream: until it has finished, launches take the
// IEEE forms (kDiv = 0: the same results, ~4 % slower), so no caller ever
// waits for it. (Rounds 1-3 waited: the first integrate launch of a process
// with a new truncation distance stalled the host for 8-10 ms --
// profiles/r4a_c4_before.json shows it as a 9.6 ms gap in the kernel trace of
// the configs[4] leg, a quarter of that leg's timed pass.)
// o3dmi_vbg_create starts it for the API's default truncation multiplier, so
// that it is normally over before the first frame arrives.
struct DivProof {
    int result = -1;  // -1: still running
    hipEvent_t done = nullptr;
    hipStream_t stream = nullptr;
    int* flag_dev = nullptr;
    int* flag_host = nullptr;  // pinned
};
static std::mutex g_div_mu;
static std::map<std::pair<int, unsigned>, DivProof> g_div_proofs;

static int DivFormsFromFlags(int host) {
    // bits 1|2: sdf / w forms failed; 4: 1/z with one Newton step failed.
    // All three short forms or none (the parts this was measured on verify
    // all of them; IEEE forms are ~4 % slower, never wrong).
    return (host & 7) ? 0 : 2;
}

static void DivProofReport(float b, int ok, int flags) {
    if (!std::getenv("O3DMI_VERBOSE")) return;
    std::fprintf(stderr,
                 "[o3dmi] exact short division for sdf_trunc = %.9g: %s "
                 "(flags %d)\n",
                 (double)b,
                 ok == 0 ? "not used" : "sdf, 1/(w+1), 1/z (1 step)",
                 flags);
}

// Starts the proof for `b` on the current device if it has not been started;
// returns the forms that may be used NOW (0 while it runs). `wait`: block until
// it has finished (O3DMI_DIV_PROOF_WAIT=1, tests of the short forms).
static int VerifyFastDivision(float b, float* y_out, bool wait = false) {
    unsigned key;
    std::memcpy(&key, &b, sizeof(key));
    const float y = 1.0f / b;  // IEEE: correctly rounded reciprocal
    if (y_out) *y_out = y;
    static const bool disabled = std::getenv("O3DMI_EXACT_DIV") != nullptr;
    static const bool always_wait =
            std::getenv("O3DMI_DIV_PROOF_WAIT") != nullptr;
    if (disabled || !(b > 0.0f) || !std::isfinite(b) || !std::isfinite(y))
        return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(g_div_mu);
    DivProof& p = g_div_proofs[std::make_pair(dev, key)];
    if (p.result >= 0) return p.result;
    if (!p.done) {
        // not started yet
        bool ok = hipMalloc((void**)&p.flag_dev, sizeof(int)) == hipSuccess &&
                  hipHostMalloc((void**)&p.flag_host, sizeof(int)) ==
                          hipSuccess &&
                  hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) ==
                          hipSuccess &&
                  hipEventCreateWithFlags(&p.done, hipEventDisableTiming) ==
                          hipSuccess;
        if (ok) {
            *p.flag_host = -1;
            (void)hipMemsetAsync(p.flag_dev, 0, sizeof(int), p.stream);
            hipLaunchKernelGGL(VerifyDivKernel, dim3(kCUs * 16), dim3(256), 0,
                               p.stream, b, y, key, p.flag_dev);
            hipLaunchKernelGGL(VerifyRcpKernel, dim3(256), dim3(256), 0,
                               p.stream, p.flag_dev);
            hipLaunchKernelGGL(VerifyRcpZKernel<1>, dim3(kCUs * 16), dim3(256),
                               0, p.stream, p.flag_dev, 4);
            ok = hipGetLastError() == hipSuccess &&
                 hipMemcpyAsync(p.flag_host, p.flag_dev, sizeof(int),
                                hipMemcpyDeviceToHost, p.stream) ==
                         hipSuccess &&
                 hipEventRecord(p.done, p.stream) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            p.result = 0;  // could not run the proof: IEEE forms
            DivProofReport(b, 0, -1);
            return 0;
        }
    }
    if (wait || always_wait) (void)hipEventSynchronize(p.done);
    if (hipEventQuery(p.done) != hipSuccess) {
        (void)hipGetLastError();  // hipErrorNotReady is not an error here
        return 0;
    }
    const int flags = *p.flag_host;
    p.result = flags < 0 ? 0 : DivFormsFromFlags(flags);
    DivProofReport(b, p.result, flags);
    // The private stream goes back at once: a process has few hardware queues,
    // and the NEXT stream somebody creates (the ICP driver's side stream, in
    // the middle of a tracking loop) costs ~8 ms instead of ~2 ms when it
    // cannot reuse this one (profiles/r4j). The event and the two words stay
    // (freeing device memory synchronises the device; there are a handful of
    // distances per process).
    if (p.stream) {
        (void)hipStreamDestroy(p.stream);
        p.stream = nullptr;
    }
    return p.result;
}

int PrefetchFastDivision(float sdf_trunc, bool wait) {
    float y;
    return VerifyFastDivision(sdf_trunc, &y, wait);
}

// The integrate role tests InBoundary with unsigned compares on the float
// patterns (IntegrateRoleWide), which read u = -0.0f as outside; u = x + cx is
// -0.0f only if cx is: a principal point of -0.0f becomes +0.0f (x + -0.0f and
// x + 0.0f differ for x = -0.0f alone, where both are inside and truncate to
// pixel 0 -- results do not change).
static void CanonicalPrincipalPoint(Camera& c) {
    if (c.cx == 0.0f) c.cx = 0.0f;
    if (c.cy == 0.0f) c.cy = 0.0f;
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
        O3DMI_REQUIRE(f->group_bit >= 0 && f->group_bit < kMaxGroup &&
                              (f->group_stamp > 0 || f->prepare_only),
                      "bad group bit / stamp");
        const TouchParams tp = MakeTouchParams(
                f->depth_intrinsic, f->extrinsic, f->rows, f->cols, f->stride,
                f->resolution, f->voxel_size, f->sdf_trunc, f->depth_scale,
                f->depth_max);
        FrontFrame& ff = sp.front[i];
        std::memcpy(ff.pose, tp.cam.e, sizeof(ff.pose));
        ff.depth = f->depth;
        ff.color = f->color;
        ff.recs = f->recs;
        ff.group_bit = f->group_bit;
        FrontShared fs = {};
        fs.p = tp;
        fs.pp.color_cam = Camera::Make(f->color_intrinsic ? f->color_intrinsic
                                                          : f->depth_intrinsic,
                                       eye4, 1.0f);
        fs.pp.color_rows = f->color_rows;
        fs.pp.color_cols = f->color_cols;
        fs.pp.with_color = f->color != nullptr;
        fs.col_lut = f->col_lut;
        fs.row_lut = f->col_lut ? f->row_lut : nullptr;
        fs.depth_div_short = f->depth_div_short;
        fs.prep_identity = f->col_lut && f->prep_identity &&
                           (f->cols % 4) == 0 && f->rows == f->color_rows &&
                           f->cols == f->color_cols;
        fs.inv_depth_scale = 1.0f / f->depth_scale;
        fs.list = f->list;
        fs.list_capacity = f->list_capacity;
        fs.out_count = f->count;
        fs.ready = f->ready;
        fs.tickets = f->ready ? f->tickets : nullptr;
        fs.touch_status = f->ready ? f->touch_status : nullptr;
        fs.group_stamp = f->group_stamp;
        fs.touch_plane = f->touch_plane & 1;
        // one touch workgroup per 16 x 16 tile of rays
        fs.n_touch_wg = f->prepare_only
                                ? 0
                                : ((tp.cols_strided + 15) / 16) *
                                          ((tp.rows_strided + 15) / 16);
        // 16 pixels per prepare lane
        fs.n_prep_wg = (f->rows * f->cols + kBlock * 16 - 1) / (kBlock * 16);
        if (fs.n_prep_wg < 1) fs.n_prep_wg = 1;
        fs.n_touch_total = fs.n_touch_wg * n_fronts;
        if (i == 0) {
            sp.fshared = fs;
            sp.front_wg = fs.n_touch_wg + fs.n_prep_wg;
        } else {
            // the frames of a launch are one group: everything but the pose
            // and the image / record pointers is shared
            O3DMI_REQUIRE(SameGroup(sp.fshared, fs),
                          "frames of one launch must share image size, "
                          "intrinsics, scales and their group");
        }
    }
    if (a) {
        O3DMI_REQUIRE(a->resolution % 4 == 0,
                      "frame-stream path needs block_resolution % 4 == 0");
        O3DMI_REQUIRE(a->n_frames >= 1 && a->n_frames <= kMaxGroup,
                      "bad group size");
        IntegParams& ip = sp.integ;
        ip.n_frames = a->n_frames;
        ip.group_stamp = a->group_stamp;
        ip.touch_plane = a->touch_plane & 1;
        for (int f = 0; f < a->n_frames; ++f) {
            const Camera cf = Camera::Make(a->depth_intrinsic, a->extrinsic[f],
                                           a->voxel_size);
            if (f == 0) {
                ip.cam0 = cf;
                CanonicalPrincipalPoint(ip.cam0);
            }
            std::memcpy(ip.ext[f], cf.e, sizeof(ip.ext[f]));
            ip.recs[f] = a->recs[f];
            ip.raw_depth[f] = nullptr;
            ip.raw_color[f] = nullptr;
        }
        ip.rows = a->rows;
        ip.cols = a->cols;
        ip.resolution = a->resolution;
        ip.res_shift = -1;
        for (int sh = 2; sh < 12; ++sh)
            if ((1 << sh) == a->resolution) ip.res_shift = sh;
        ip.cube = LaneCube(ip.res_shift);
        ip.sdf_trunc = a->sdf_trunc;
        ip.depth_max = a->depth_max;
        fast_div = VerifyFastDivision(a->sdf_trunc, &ip.inv_sdf_trunc);
        ip.list = a->list;
        ip.ready = a->ready;
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
        ip.prof_map_size = a->prof_map_size;
        const int n_quads =
                (a->resolution * a->resolution * a->resolution) >> 1;
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
#define O3DMI_LAUNCH_STEP(WT, VT, COLOR)                                      \
    do {                                                                      \
        if (fast_div == 2)                                                    \
            hipLaunchKernelGGL((FrameStepKernel<WT, VT, COLOR, 2>), grid,     \
                               block, 0, s, sp);                              \
        else                                                                  \
            hipLaunchKernelGGL((FrameStepKernel<WT, VT, COLOR, 0>), grid,     \
                               block, 0, s, sp);                              \
    } while (0)
    if (grid_dtype == O3DMI_U16) {
        if (col) O3DMI_LAUNCH_STEP(uint16_t, uint16_t, true);
        else O3DMI_LAUNCH_STEP(uint16_t, uint16_t, false);
    } else {
        if (col) O3DMI_LAUNCH_STEP(float, float, true);
        else O3DMI_LAUNCH_STEP(float, float, false);
    }
#undef O3DMI_LAUNCH_STEP
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int LaunchChunkIntegrate(o3dmi_hash* bh, const ChunkIntegrateArgs& a,
                         hipStream_t s) {
    O3DMI_REQUIRE(a.resolution % 4 == 0 && a.n_frames >= 1 &&
                          a.n_frames <= kChunkFrames && a.frames &&
                          a.entries && a.count && a.depth_scale > 0,
                  "chunk integrate: bad arguments");
    ChunkParams cp = {};
    cp.hv = bh->view;
    IntegParams& ip = cp.integ;
    ip.n_frames = a.n_frames;
    static const double eye4[16] = {1, 0, 0, 0, 0, 1, 0, 0,
                                    0, 0, 1, 0, 0, 0, 0, 1};
    ip.cam0 = Camera::Make(a.depth_intrinsic, eye4, a.voxel_size);
    CanonicalPrincipalPoint(ip.cam0);
    ip.rows = a.rows;
    ip.cols = a.cols;
    ip.resolution = a.resolution;
    ip.res_shift = -1;
    for (int sh = 2; sh < 12; ++sh)
        if ((1 << sh) == a.resolution) ip.res_shift = sh;
    ip.cube = LaneCube(ip.res_shift);
    ip.sdf_trunc = a.sdf_trunc;
    ip.depth_max = a.depth_max;
    // The chunk launch exists in the proven short-division forms only (the
    // IEEE forms of it were half of this file's chunk kernels for a case that
    // does not occur on gfx950): the sliced path waits for the proof, and
    // where it fails the caller takes the replicated touch (same grids).
    const int fast_div =
            VerifyFastDivision(a.sdf_trunc, &ip.inv_sdf_trunc, /*wait=*/true);
    if (fast_div != 2) {
        SetLastError("chunk integrate: the short division forms are not "
                     "proven for this truncation distance (O3DMI_EXACT_DIV?) "
                     "-- use the replicated block touch");
        return O3DMI_ERR_UNSUPPORTED;
    }
    ip.list = nullptr;
    ip.ready = nullptr;
    ip.entries = a.entries;
    ip.frame_tab = a.frames;
    ip.count = a.count;
    ip.list_capacity = a.entries_cap;
    ip.tsdf = a.tsdf;
    ip.weight = a.weight;
    ip.color = a.color;
    ip.zero_counter = nullptr;
    ip.size_host = a.size_host;
    ip.status_stamp = a.status_stamp;
    ip.prof_count = a.prof_count;
    ip.prof_frame_blocks = a.prof_frame_blocks;
    ip.prof_map_size = a.prof_map_size;
    ip.depth_scale = a.depth_scale;
    ip.inv_depth_scale = 1.0f / a.depth_scale;
    ip.depth_div_short = a.depth_div_short;
    const int n_quads = (a.resolution * a.resolution * a.resolution) >> 1;
    const int parts = (n_quads + 255) >> 8;
    int64_t g = ((int64_t)a.grid_hint + (a.grid_hint >> 2) + 64) * parts;
    const int64_t g_max = (int64_t)kCUs * 32;
    if (g > g_max) g = g_max;
    if (g < kCUs) g = kCUs;
    const dim3 grid((unsigned)g), block(256);
    const bool col = a.with_color && a.color != nullptr;
    // O3DMI_CHUNK_TIMELINE=<file>[:<k>]: chunk launch number k (default 20)
    // of the process records a per-work-item timeline (analysis tool:
    // tools/chunk_timeline.py); the launch is synchronised, every other launch
    // is untouched.
    static std::string tl_file;
    static const int tl_launch = [] {
        const char* e = std::getenv("O3DMI_CHUNK_TIMELINE");
        if (!e) return -1;
        tl_file = e;
        int k = 20;
        const size_t colon = tl_file.rfind(':');
        if (colon != std::string::npos && colon + 1 < tl_file.size() &&
            tl_file.find_first_not_of("0123456789", colon + 1) ==
                    std::string::npos) {
            k = std::atoi(tl_file.c_str() + colon + 1);
            tl_file.resize(colon);
        }
        return k;
    }();
    const char* tl_path = tl_launch >= 0 ? tl_file.c_str() : nullptr;
    static int tl_seen = 0;
    unsigned long long* tl_dev = nullptr;
    const size_t tl_words = (size_t)a.entries_cap * parts * 4;
    if (tl_path && tl_seen++ == tl_launch) {
        O3DMI_HIP_CHECK(hipMalloc((void**)&tl_dev, tl_words * 8));
        O3DMI_HIP_CHECK(hipMemsetAsync(tl_dev, 0, tl_words * 8, s));
        ip.prof_items = tl_dev;
    }
#define O3DMI_LAUNCH_CHUNK_D(WT, VT, COLOR, D)                                \
    do {                                                                      \
        if (a.raw)                                                            \
            hipLaunchKernelGGL(                                               \
                    (ChunkIntegrateKernel<WT, VT, COLOR, D, true>), grid,     \
                    block, 0, s, cp);                                         \
        else                                                                  \
            hipLaunchKernelGGL(                                               \
                    (ChunkIntegrateKernel<WT, VT, COLOR, D, false>), grid,    \
                    block, 0, s, cp);                                         \
    } while (0)
#define O3DMI_LAUNCH_CHUNK(WT, VT, COLOR) O3DMI_LAUNCH_CHUNK_D(WT, VT, COLOR, 2)
    if (a.grid_dtype == O3DMI_U16) {
        if (col) O3DMI_LAUNCH_CHUNK(uint16_t, uint16_t, true);
        else O3DMI_LAUNCH_CHUNK(uint16_t, uint16_t, false);
    } else {
        if (col) O3DMI_LAUNCH_CHUNK(float, float, true);
        else O3DMI_LAUNCH_CHUNK(float, float, false);
    }
#undef O3DMI_LAUNCH_CHUNK
#undef O3DMI_LAUNCH_CHUNK_D
    O3DMI_HIP_CHECK(hipGetLastError());
    if (tl_dev) {
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(tl_words);
        O3DMI_HIP_CHECK(hipMemcpy(h.data(), tl_dev, tl_words * 8,
                                  hipMemcpyDeviceToHost));
        (void)hipFree(tl_dev);
        if (FILE* f = std::fopen(tl_path, "wb")) {
            const long long head[4] = {(long long)a.entries_cap, parts,
                                       a.n_frames, (long long)g};
            std::fwrite(head, sizeof(head), 1, f);
            std::fwrite(h.data(), 8, tl_words, f);
            std::fclose(f);
        }
    }
    return O3DMI_OK;
}

// o3dmi_preload: HIP loads this translation unit's code object at the first
// launch of one of its kernels; asking for a kernel's attributes does it now.
int PreloadStream() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(
                                               &VerifyRcpKernel)) == hipSuccess
                   ? 0
                   : 1;
}

}  // namespace o3dmi
