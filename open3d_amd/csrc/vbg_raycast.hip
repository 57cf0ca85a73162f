// Surface prediction for VoxelBlockGrid on MI355X.
//
//   o3dmi_vbg_estimate_range <- EstimateRangeCUDA (t/geometry/kernel/VoxelBlockGridImpl.h:310-555)
//   o3dmi_vbg_raycast        <- RayCastCUDA<tsdf_t,weight_t,color_t> (VoxelBlockGridImpl.h:578-1120)
//
// EstimateRange: the reference expands every block's projected rectangle into
// 16x16 "fragments" through an atomically grown buffer (3 passes, one host
// sync, overflow drops fragments for the current call). Min/max are
// order-independent, so here one wavefront per block rasterises its rectangle
// straight into the range map with integer atomics on the float bit patterns;
// no fragment buffer, no overflow case, same result.
//
// RayCast: one lane per pixel with the reference's per-ray arithmetic, laid out
// for the memory system instead of for the image: a wave owns an 8 x 8 pixel
// tile (a workgroup 32 x 8), so its 64 rays stay within one or two voxel
// blocks and their voxel loads fall into a handful of cache lines (a 1 x 64
// pixel strip spreads over ~20 cm of surface and three blocks); the 8 x 8 tile
// is also exactly one cell of the range map (down factor 8), so the whole wave
// marches the same depth interval. Block look-ups go through three levels: the
// 1-entry register cache the reference keeps, a workgroup-shared table in LDS
// (256 entries, one 64-bit word each, absent blocks included -- free space is
// where most look-ups happen), and only then the open-addressing probe of
// block_hash.hip in HBM. The trilinear neighbourhood is fetched as a batch
// (8 weights, then 8 tsdf and 24 colour values in flight together).

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace o3dmi {
namespace {

__device__ __forceinline__ void AtomicMinF(float* addr, float value) {
    // GeometryMacros.h:69-77
    if (value >= 0) atomicMin((int*)addr, __float_as_int(value));
    else atomicMax((unsigned int*)addr, __float_as_uint(value));
}
__device__ __forceinline__ void AtomicMaxF(float* addr, float value) {
    // GeometryMacros.h:79-87
    if (value >= 0) atomicMax((int*)addr, __float_as_int(value));
    else atomicMin((unsigned int*)addr, __float_as_uint(value));
}

__global__ void RangeFillKernel(float* __restrict__ range, int64_t n,
                                float depth_min, float depth_max) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        range[2 * i + 0] = depth_max;
        range[2 * i + 1] = depth_min;
    }
}

// One wave per block key (4 waves per workgroup).
// What the ray cast of the LAST frame measured per tile, turned into this
// frame's tile order, longest first (round 6). A 1280 x 720 image is 3600
// tiles for 1280 resident workgroups: the launch runs in ~3 rounds and lasted
// as long as whatever long-marching tiles (the floor's grazing rays) happened
// to start in the last one. Workgroups are dispatched in index order, so
// handing workgroup k the k-th longest tile of the previous frame (the camera
// moves a few millimetres between frames) is longest-processing-time-first
// scheduling. Costs are {frame number, 10 ns ticks of the tile's slowest
// wave}, written with atomicMax by the ray cast itself; a tile without a cost
// of the wanted frame sorts last. Which tile a workgroup renders changes no
// pixel. The sort is one extra workgroup of the range-estimate launch.
struct TileOrder {
    const unsigned long long* cost;  // [n_tiles] or NULL
    int* order;                      // [n_tiles]
    int n_tiles;
    unsigned want_seq;
};
__device__ __forceinline__ void SortTilesLongestFirst(const TileOrder& to) {
    __shared__ int hist[256];
    __shared__ int wave_sum[4];
    constexpr int kPerThread = 16;  // <= 4096 tiles, kBlock = 256 threads
    for (int c = threadIdx.x; c < 256; c += blockDim.x) hist[c] = 0;
    // every load of the thread in flight at once (a loop of load -> LDS
    // atomic is one memory round trip per tile: 14 of them for 3600 tiles)
    int cls[kPerThread];
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
        const int t = (int)threadIdx.x + u * 256;
        cls[u] = -1;
        if (t < to.n_tiles) {
            const unsigned long long v = to.cost[t];
            const unsigned ticks = (unsigned)v >> 5;  // 320 ns classes
            cls[u] = (unsigned)(v >> 32) != to.want_seq
                             ? 0
                             : (int)(ticks < 255u ? ticks : 255u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPerThread; ++u)
        if (cls[u] >= 0) atomicAdd(&hist[cls[u]], 1);
    __syncthreads();
    // class c starts behind every longer class: an exclusive scan over the
    // classes in descending order (thread t holds class 255 - t)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = hist[255 - (int)threadIdx.x];
    int incl = h;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = incl - h;
    for (int w = 0; w < wave; ++w) before += wave_sum[w];
    hist[255 - (int)threadIdx.x] = before;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPerThread; ++u)
        if (cls[u] >= 0)
            to.order[atomicAdd(&hist[cls[u]], 1)] =
                    (int)threadIdx.x + u * 256;
}

__global__ void EstimateRangeKernel(const int* __restrict__ block_keys,
                                    int key_stride, int64_t n_blocks,
                                    const int* __restrict__ n_blocks_dev,
                                    float* __restrict__ range,
                                    Camera cam, int h_down, int w_down,
                                    int down_factor, int64_t block_resolution,
                                    float voxel_size, float depth_min,
                                    float depth_max, TileOrder to) {
    if (to.cost && blockIdx.x == gridDim.x - 1) {
        // (the launch was given one workgroup more for this)
        SortTilesLongestFirst(to);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t wave_id =
            ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves =
            ((int64_t)(gridDim.x - (to.cost ? 1 : 0)) * blockDim.x) >> 6;
    // Device-resident count (frame-stream callers): the live length of the
    // key list, bounded by the host-side capacity n_blocks.
    if (n_blocks_dev) {
        const int64_t live = *n_blocks_dev;
        if (live < n_blocks) n_blocks = live;
    }
    for (int64_t b = wave_id; b < n_blocks; b += n_waves) {
        // ({n,3} keys, or the x, y, z of a frame stream's {slot, x, y, z}
        // block list: stride 4)
        const int* key = block_keys + (int64_t)key_stride * b;
        int u_min = w_down - 1, v_min = h_down - 1, u_max = 0, v_max = 0;
        float z_min = depth_max, z_max = depth_min;
        // VoxelBlockGridImpl.h:386-412 (all lanes compute the same rectangle)
        for (int i = 0; i < 8; ++i) {
            float xw = (key[0] + ((i & 1) > 0)) * block_resolution * voxel_size;
            float yw = (key[1] + ((i & 2) > 0)) * block_resolution * voxel_size;
            float zw = (key[2] + ((i & 4) > 0)) * block_resolution * voxel_size;
            float xc, yc, zc, u, v;
            cam.RigidTransform(xw, yw, zw, xc, yc, zc);
            if (zc <= 0) continue;
            cam.Project(xc, yc, zc, u, v);
            u /= down_factor;
            v /= down_factor;
            v_min = min((int)floorf(v), v_min);
            v_max = max((int)ceilf(v), v_max);
            u_min = min((int)floorf(u), u_min);
            u_max = max((int)ceilf(u), u_max);
            z_min = fminf(z_min, zc);
            z_max = fmaxf(z_max, zc);
        }
        v_min = max(0, v_min);
        v_max = min(h_down - 1, v_max);
        u_min = max(0, u_min);
        u_max = min(w_down - 1, u_max);
        if (v_min >= v_max || u_min >= u_max || z_min >= z_max) continue;

        const int rw = u_max - u_min + 1;
        const int area = rw * (v_max - v_min + 1);
        for (int k = lane; k < area; k += 64) {
            int v = v_min + k / rw;
            int u = u_min + k % rw;
            float* range_ptr = range + 2 * ((int64_t)v * w_down + u);
            AtomicMinF(range_ptr + 0, z_min);
            AtomicMaxF(range_ptr + 1, z_max);
        }
    }
}

struct RayCastParams {
    Camera c2w;  // intrinsics + inverse extrinsic
    Camera w2c;  // intrinsics + extrinsic
    int h, w;
    int block_resolution;
    float voxel_size, block_size;
    float depth_scale, weight_threshold, sdf_trunc;
    int range_down, w_down;
    float *depth, *vertex, *color, *normal;
    long long* index;
    uint8_t* mask;
    float *ratio, *ratio_dx, *ratio_dy, *ratio_dz;
    int* steps;  // diagnostics (O3DMI_RAYCAST_STEPS=1): march steps per pixel
    long long* clocks;  // ... and per workgroup: start, march done, end (100 MHz)
    int xcd_bands;  // tiles dealt to the XCDs in image bands (0: round-robin)
    int coop;       // bit 0: idle lanes sample ahead for crawling rays
                    // bit 1: a grid-owned range map is left CLEAN by the
                    // launch that consumes it -- with an 8-pixel down factor a
                    // wave's 8 x 8 pixels are exactly one cell, read by that
                    // wave alone: its lane 0 writes the cell back to the
                    // {depth_max, depth_min} EstimateRange starts from (kept
                    // in the two floats behind the map's last cell), so the
                    // next frame needs no clearing launch. (A flag bit and a
                    // load, not three more scalar arguments: the slim 16^3
                    // form sits at its register cap.)
    // longest-first tile order (TileOrder above): workgroup k renders tile
    // tile_order[k] and leaves its duration in tile_cost; NULL: tile k
    const int* tile_order;
    unsigned long long* tile_cost;
    unsigned cost_seq;
    int band_ty0, band_tys;  // the launch renders tile rows [ty0, ty0 + tys)
                             // only (o3dmi_vbg_raycast_rows: a rank's band of
                             // a pixel-sharded ray cast); tys = 0: the image
};

struct BlockCache {
    int x, y, z, block_idx;
    __device__ __forceinline__ int Check(int xi, int yi, int zi) const {
        return (xi == x && yi == y && zi == z) ? block_idx : -1;
    }
    __device__ __forceinline__ void Update(int xi, int yi, int zi, int b) {
        x = xi; y = yi; z = zi; block_idx = b;
    }
};

__device__ __forceinline__ int SignI(int x) {
    return (x > 0) ? 1 : ((x < 0) ? -1 : 0);
}

// Workgroup-shared block table in LDS. One 64-bit word per entry:
//   bit 63      valid
//   bits 56-32  block key relative to the table's origin, 3 x 8 bits biased
//               by 128 (|offset| < 128 blocks; farther blocks bypass the
//               table)
//   bits 31-0   buffer index + 1 (0 = the block does not exist)
// A word is written with one ds_write_b64, so an entry is never torn; a
// colliding key simply replaces the entry (it is a cache: a miss falls back
// to the hash map in HBM).
constexpr int kLdsBlocks = 256;
struct LdsBlockTable {
    unsigned long long* e;  // [kLdsBlocks] in LDS
    int ox, oy, oz;         // origin block (wave-uniform)

    __device__ __forceinline__ bool Encode(int x, int y, int z,
                                           unsigned& rel) const {
        const unsigned dx = (unsigned)(x - ox + 128);
        const unsigned dy = (unsigned)(y - oy + 128);
        const unsigned dz = (unsigned)(z - oz + 128);
        rel = (dx << 16) | (dy << 8) | dz;
        return (dx | dy | dz) < 256u;
    }
    __device__ __forceinline__ unsigned Slot(unsigned rel) const {
        return (rel * 0x9E3779B1u) >> 24;  // 8 bits
    }
    // buffer index, -1 = known absent, -2 = not in the table
    __device__ __forceinline__ int Lookup(unsigned rel) const {
        const unsigned long long w = e[Slot(rel)];
        if ((unsigned)(w >> 32) != (0x80000000u | rel)) return -2;
        return (int)(unsigned)w - 1;
    }
    __device__ __forceinline__ void Store(unsigned rel, int buf_idx) {
        e[Slot(rel)] = ((unsigned long long)(0x80000000u | rel) << 32) |
                       (unsigned)(buf_idx + 1);
    }
};

// Block look-up through register cache -> LDS table -> hash map.
__device__ __forceinline__ int FindBlock(const HashView& hv,
                                         LdsBlockTable& tab, BlockCache& cache,
                                         int x_b, int y_b, int z_b) {
    int idx = cache.Check(x_b, y_b, z_b);
    if (idx >= 0) return idx;
    unsigned rel;
    const bool in_table = tab.Encode(x_b, y_b, z_b, rel);
    if (in_table) idx = tab.Lookup(rel);
    else idx = -2;
    if (idx == -2) {
        idx = hv.Find(x_b, y_b, z_b);
        if (in_table) tab.Store(rel, idx);
    }
    if (idx >= 0) cache.Update(x_b, y_b, z_b, idx);
    return idx;
}

// FULL = the per-neighbour maps (index / mask / interp_ratio*) are requested.
// The common depth / vertex / colour / normal rendering (slam::Model) runs the
// slim instantiation: without the 8-entry output arrays it needs half the
// registers, so every ray of a 720p frame is resident at once.
template <typename weight_t, typename color_t, bool FULL, int RES, bool DIAG>
__global__ void __launch_bounds__(256, RES ? (FULL ? 4 : 5) : 0)
RayCastKernel(HashView hv, RayCastParams p, const float* __restrict__ tsdf_base,
              const weight_t* __restrict__ weight_base,
              const color_t* __restrict__ color_base,
              const float* range_map) {
    __shared__ unsigned long long lds_blocks[kLdsBlocks];
    // cooperative march (below): ray state of up to 32 rays per wave, and the
    // samples their helper lanes fetch
    __shared__ float coop_state[4][32][8];
    __shared__ float coop_result[4][32][6];
    // RES = 16 (the block resolution everything uses) folds the index
    // arithmetic into shifts; RES = 0 takes it from the call
    const int res = RES ? RES : p.block_resolution;
    const int res2 = res * res;
    const int res3 = res2 * res;
    const bool render_color = color_base != nullptr && p.color != nullptr;
    const bool visit_neighbors = render_color || p.normal || p.mask ||
                                 p.index || p.ratio || p.ratio_dx ||
                                 p.ratio_dy || p.ratio_dz;
    // Workgroup tile 32 x 8 pixels, wave tile 8 x 8.
    const int tiles_x = (p.w + 31) / 32;
    const int tiles_y = p.band_tys > 0 ? p.band_tys : (p.h + 7) / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LdsBlockTable tab;
    tab.e = lds_blocks;
    // Table origin: the block under the camera centre's first sample does not
    // matter, any block near the frustum does -- the camera's own block.
    {
        float x_o, y_o, z_o;
        p.c2w.RigidTransform(0, 0, 0, x_o, y_o, z_o);
        tab.ox = (int)floorf(x_o / p.block_size);
        tab.oy = (int)floorf(y_o / p.block_size);
        tab.oz = (int)floorf(z_o / p.block_size);
    }

    // XCD-aware deal (round 3): workgroup b runs on XCD b % 8 (observed; only
    // speed depends on it) and every XCD has its own 4 MB L2. Dealt
    // round-robin, the voxels a tile's rays meet are fetched into all eight
    // L2s (58 % of the launch's L2 requests miss, profiles/
    // r3n_raycast_counters.json); dealt in eight contiguous BANDS, one per
    // XCD, into one. The bands are vertical strips (tiles numbered column by
    // column): horizontal bands hand all the long-marching tiles of a frame --
    // the floor's grazing rays -- to one or two XCDs and the launch, which is
    // as long as its slowest tiles, gets slower (67.7 against 65.4 us at
    // VGA); strips have floor, walls and ceiling each: 60.3 against 67.5 us.
    // Only while every tile is resident at once (<= 5 workgroups per CU): at
    // 720p the launch runs in rounds and the strips cost 4 %.
    // gridDim.x is a multiple of 8 either way.
    const int n_tiles_all = tiles_x * tiles_y;
    const int per_band = (n_tiles_all + 7) >> 3;
    const int k_step = p.xcd_bands ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    for (int k = p.xcd_bands ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;;
         k += k_step) {
        int tile;
        if (p.xcd_bands) {
            if (k >= per_band) break;
            tile = (int)(blockIdx.x & 7) * per_band + k;
            if (tile >= n_tiles_all) break;
        } else {
            tile = k;
            if (tile >= n_tiles_all) break;
            if (p.tile_order) tile = p.tile_order[k];
        }
        const unsigned long long tile_t0 = p.tile_cost ? wall_clock64() : 0ull;
        if (DIAG && threadIdx.x == 0)
            p.clocks[3 * (int64_t)tile] = wall_clock64();
        __syncthreads();  // the previous tile's readers are done
        for (int k = threadIdx.x; k < kLdsBlocks; k += blockDim.x)
            lds_blocks[k] = 0ull;
        __syncthreads();
        // bands: tiles numbered column by column, so a band is a vertical
        // strip of the image (floor, walls and ceiling in every strip)
        const int ty = p.xcd_bands ? tile % tiles_y : tile / tiles_x;
        const int tx = p.xcd_bands ? tile / tiles_y : tile - ty * tiles_x;
        const int x = tx * 32 + wave * 8 + (lane & 7);
        const int y = (ty + p.band_ty0) * 8 + (lane >> 3);
        // a pixel outside the image keeps its lane in the wave (the march
        // below is wave-uniform), it just has no ray and stores nothing
        const bool inside = x < p.w && y < p.h;
        const int64_t workload_idx =
                inside ? (int64_t)y * p.w + x : (int64_t)0;
        const float* range =
                range_map +
                (inside ? 2 * ((int64_t)(y / p.range_down) * p.w_down +
                               (x / p.range_down))
                        : (int64_t)0);

        float* depth_ptr = p.depth ? p.depth + workload_idx : nullptr;
        float* vertex_ptr = p.vertex ? p.vertex + 3 * workload_idx : nullptr;
        float* color_ptr = p.color ? p.color + 3 * workload_idx : nullptr;
        float* normal_ptr = p.normal ? p.normal + 3 * workload_idx : nullptr;
        long long* index_ptr = p.index ? p.index + 8 * workload_idx : nullptr;
        uint8_t* mask_ptr = p.mask ? p.mask + 8 * workload_idx : nullptr;
        float* ratio_ptr = p.ratio ? p.ratio + 8 * workload_idx : nullptr;
        float* ratio_dx_ptr = p.ratio_dx ? p.ratio_dx + 8 * workload_idx : nullptr;
        float* ratio_dy_ptr = p.ratio_dy ? p.ratio_dy + 8 * workload_idx : nullptr;
        float* ratio_dz_ptr = p.ratio_dz ? p.ratio_dz + 8 * workload_idx : nullptr;

        // Outputs are accumulated in registers and written once.
        float out_depth = 0;
        float out_vertex[3] = {0, 0, 0};
        float out_color[3] = {0, 0, 0};
        float out_normal[3] = {0, 0, 0};
        constexpr int kNb = FULL ? 8 : 1;
        float o_ratio[kNb], o_dx[kNb], o_dy[kNb], o_dz[kNb];
        long long o_index[kNb];
        uint8_t o_mask[kNb];
#pragma unroll
        for (int k = 0; k < kNb; ++k) {
            o_ratio[k] = o_dx[k] = o_dy[k] = o_dz[k] = 0;
            o_index[k] = 0;
            o_mask[k] = 0;
        }

        float t = range[0];
        const float t_max = range[1];
        if (p.coop & 2) {
            // (the two loads above are this cell's only readers)
            asm volatile("" ::: "memory");
            if (lane == 0 && inside) {
                const float* clean =
                        range_map + 2 * (int64_t)(p.h / 8) * p.w_down;
                float* cell = const_cast<float*>(range);
                cell[0] = clean[0];
                cell[1] = clean[1];
            }
        }
        {
            float x_c, y_c, z_c, x_g, y_g, z_g, x_o, y_o, z_o;
            float t_prev = t;
            float tsdf_prev = -1.0f;
            float tsdf = 1.0f;
            float wgt = 0.0f;

            p.c2w.RigidTransform(0, 0, 0, x_o, y_o, z_o);
            p.c2w.Unproject((float)x, (float)y, 1.0f, x_c, y_c, z_c);
            p.c2w.RigidTransform(x_c, y_c, z_c, x_g, y_g, z_g);
            const float x_d = x_g - x_o, y_d = y_g - y_o, z_d = z_g - z_o;

            BlockCache cache{0, 0, 0, -1};
            bool surface_found = false;
            int n_crawl = 0;
            int n_steps = 0;
            // GetLinearIdxAtT, VoxelBlockGridImpl.h:784-823, for the sample
            // at parameter tj of the ray with direction (dx, dy, dz); -1 if
            // its block does not exist
            const auto locate = [&](float tj, float dx, float dy,
                                    float dz) -> int64_t {
                const float xg = x_o + tj * dx;
                const float yg = y_o + tj * dy;
                const float zg = z_o + tj * dz;
                const int x_b = (int)floorf(xg / p.block_size);
                const int y_b = (int)floorf(yg / p.block_size);
                const int z_b = (int)floorf(zg / p.block_size);
                const int block_buf_idx =
                        FindBlock(hv, tab, cache, x_b, y_b, z_b);
                if (block_buf_idx < 0) return -1;
                const int x_v =
                        (int)((xg - x_b * p.block_size) / p.voxel_size);
                const int y_v =
                        (int)((yg - y_b * p.block_size) / p.voxel_size);
                const int z_v =
                        (int)((zg - z_b * p.block_size) / p.voxel_size);
                return (int64_t)block_buf_idx * res3 +
                       (z_v * res2 + y_v * res + x_v);
            };
            // The reference's march (VoxelBlockGridImpl.h:825-870), sample
            // for sample. The launch lasts as long as its slowest wave, and
            // a wave as long as its longest ray: 6.7 samples on average in
            // the tracking scene, 67 for the worst, most of them CRAWL steps
            // -- samples with tsdf * sdf_trunc < voxel_size (unobserved
            // voxels of an allocated block, tsdf 0 weight 0; voxels near or
            // behind a surface the weight threshold hides), after which the
            // march moves exactly one voxel size. The slowest workgroup took
            // 59 us where the median takes 17 (O3DMI_RAYCAST_STEPS=1 prints
            // these distributions). A step is ~300 instructions plus a load,
            // and the instructions are the larger part: sampling ahead in
            // the ray's own lane costs what it saves, a warmed block table
            // or fewer dependent loads change nothing (all measured, round
            // 3). But while the long rays crawl, most lanes of their wave are
            // idle. So the march has two phases: the plain per-lane loop
            // while more than half of the wave is marching, then a
            // WAVE-UNIFORM loop in which a finished lane stays, idle -- and
            // whenever one of the rays left is crawling, every ray gets 2
            // lanes (up to 32 rays left), 4 (16), ... 64 (the last one),
            // which locate and load the samples at t, t + voxel, t + 2 voxel,
            // ... at the price of one step. The window is then resolved in
            // parallel: sample j + 1 is reached iff sample j is a crawl step
            // inside the range, the first lane whose sample is not such a
            // step holds the last sample consumed and works out the ray's
            // state after it (t, t_prev, tsdf, tsdf_prev as the reference's
            // loop would leave them). `t + voxel_size` is the same float
            // addition in a helper lane and in the march: the sample
            // positions, and so every output, are the reference's bit for
            // bit. 72 -> 57.5 us per VGA launch, 113 -> 98 us at 720p, the
            // slowest workgroup 59 -> 42 us (same box).
            bool mine = inside && t < t_max;
            bool crawling = false;
            int it_plain = 0, it_coop = 0;  // DIAG: loop passes of the wave
            // one sample of this lane's ray, the reference's loop body
            const auto step = [&]() {
                ++n_steps;
                const int64_t lin = locate(t, x_d, y_d, z_d);
                if (lin < 0) {
                    t_prev = t;
                    t += p.block_size;
                    crawling = false;
                } else {
                    tsdf_prev = tsdf;
                    tsdf = tsdf_base[lin];
                    wgt = (float)weight_base[lin];
                    if (tsdf_prev > 0 && wgt >= p.weight_threshold &&
                        tsdf <= 0) {
                        surface_found = true;
                    } else {
                        t_prev = t;
                        const float delta = tsdf * p.sdf_trunc;
                        crawling = delta < p.voxel_size;
                        if (DIAG) n_crawl += crawling ? 1 : 0;
                        t += crawling ? p.voxel_size : delta;
                    }
                }
                mine = !surface_found && t < t_max;
            };
            // while more than half of the wave marches: the plain loop
            while (mine) {
                if ((p.coop & 1) &&
                    __popcll(__builtin_amdgcn_ballot_w64(true)) <= 32)
                    break;
                if (DIAG) ++it_plain;
                step();
            }
            for (;;) {
                const unsigned long long act =
                        __builtin_amdgcn_ballot_w64(mine);
                if (act == 0) break;
                const int n_act = __popcll(act);
                // lanes per ray, as log2: 2 lanes while up to 32 rays are
                // left, 4 up to 16, ... 64 for the last ray
                const int shift =
                        n_act <= 1 ? 6 : __builtin_clz((unsigned)(n_act - 1)) - 26;
                const bool coop =
                        (p.coop & 1) && n_act <= 32 &&
                        __builtin_amdgcn_ballot_w64(mine && crawling) != 0;
                if (DIAG) {
                    it_plain += coop ? 0 : 1;
                    it_coop += coop ? 1 : 0;
                }
                if (!coop) {
                    if (mine) step();
                    continue;
                }
                // ---- cooperative step: the ray of rank r gets the lanes
                // r << shift .. ((r + 1) << shift) - 1, lane j of them the
                // sample at t + j voxel sizes
                const int rank = (int)__builtin_amdgcn_mbcnt_hi(
                        (unsigned)(act >> 32),
                        __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
                if (mine) {
                    float* st = coop_state[wave][rank];
                    st[0] = t;
                    st[1] = x_d;
                    st[2] = y_d;
                    st[3] = z_d;
                    st[4] = t_max;
                    st[5] = tsdf;
                    st[6] = tsdf_prev;
                    st[7] = t_prev;
                }
                __builtin_amdgcn_wave_barrier();
                {
                    const int width = 1 << shift;
                    const int slot = lane >> shift;
                    const int j = lane & (width - 1);
                    const bool helper = slot < n_act;
                    const float* st = coop_state[wave][helper ? slot : 0];
                    float tj = st[0];
                    const float hx = st[1], hy = st[2], hz = st[3];
                    const float h_max = st[4];
                    const float o_tsdf = st[5], o_tsdf_prev = st[6];
                    const float o_t_prev = st[7];
                    // j additions of the voxel size, one after the other as
                    // the march makes them
                    for (int k = 0; k < j; ++k) tj += p.voxel_size;
                    int64_t lin = -1;
                    const bool sampled = helper && (j == 0 || tj < h_max);
                    if (sampled) lin = locate(tj, hx, hy, hz);
                    // no branch around the loads (the compiler would wait
                    // for them inside it): a sample without a block reads
                    // voxel 0 of the buffers and is never looked at
                    const int64_t at = lin >= 0 ? lin : 0;
                    const float ts = tsdf_base[at];
                    const float wt = (float)weight_base[at];
                    // The march over the window, all samples at once. Lane j
                    // needs the tsdf of the one and two samples before it
                    // (the ray's own last two for j < 2) and the t of the
                    // sample before it; sample j + 1 is reached iff sample j
                    // is a crawl step that stays inside the range; the first
                    // sample that is not such a step is the last one
                    // consumed, and its lane works out the ray's new state.
                    float ts_1 = __shfl_up(ts, 1, width);
                    float ts_2 = __shfl_up(ts, 2, width);
                    float t_1 = __shfl_up(tj, 1, width);
                    if (j < 1) { ts_1 = o_tsdf; t_1 = o_t_prev; }
                    if (j < 2) ts_2 = j == 0 ? o_tsdf_prev : o_tsdf;
                    const bool valid = lin >= 0;
                    const bool surface = valid && ts_1 > 0 &&
                                         wt >= p.weight_threshold && ts <= 0;
                    const float delta = ts * p.sdf_trunc;
                    const bool crawl = delta < p.voxel_size;
                    const float t_next = tj + (crawl ? p.voxel_size : delta);
                    const bool goes_on = sampled && valid && !surface &&
                                         crawl && t_next < h_max;
                    const unsigned long long stops =
                            __builtin_amdgcn_ballot_w64(!goes_on) >>
                            (slot << shift);
                    // (a window always stops: its last lane at the latest)
                    const int j_last =
                            min(__builtin_ctzll(stops | (1ull << (width - 1))),
                                width - 1);
                    if (helper && j == j_last) {
                        float* rs = coop_result[wave][slot];
                        const bool all_on = goes_on;  // window used up
                        if (!valid) {
                            rs[0] = tj + p.block_size;  // t
                            rs[1] = tj;                 // t_prev
                            rs[2] = ts_1;               // tsdf
                            rs[3] = ts_2;               // tsdf_prev
                            rs[4] = 0.0f;               // 1 surface, 2 crawling
                        } else if (surface) {
                            rs[0] = tj;
                            rs[1] = t_1;
                            rs[2] = ts;
                            rs[3] = ts_1;
                            rs[4] = 1.0f;
                        } else {
                            rs[0] = t_next;
                            rs[1] = tj;
                            rs[2] = ts;
                            rs[3] = ts_1;
                            rs[4] = crawl ? 2.0f : 0.0f;
                        }
                        (void)all_on;
                        rs[5] = (float)(j_last + 1);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (mine) {
                    const float* rs = coop_result[wave][rank];
                    t = rs[0];
                    t_prev = rs[1];
                    tsdf = rs[2];
                    tsdf_prev = rs[3];
                    surface_found = rs[4] == 1.0f;
                    crawling = rs[4] == 2.0f;
                    const int consumed = (int)rs[5];
                    n_steps += consumed;
                    if (DIAG) n_crawl += crawling ? consumed : consumed - 1;
                    mine = !surface_found && t < t_max;
                }
                __builtin_amdgcn_wave_barrier();
            }

            if (DIAG && inside)
                p.steps[workload_idx] = (n_steps & 255) |
                                        ((min(it_plain, 255)) << 8) |
                                        ((min(it_coop, 255)) << 16) |
                                        ((n_crawl & 255) << 24);
            if (DIAG)
                atomicMax((unsigned long long*)&p.clocks[3 * (int64_t)tile + 1],
                          (unsigned long long)wall_clock64());
            if (surface_found) {
                float t_intersect =
                        (t * tsdf_prev - t_prev * tsdf) / (tsdf_prev - tsdf);
                x_g = x_o + t_intersect * x_d;
                y_g = y_o + t_intersect * y_d;
                z_g = z_o + t_intersect * z_d;

                out_depth = t_intersect * p.depth_scale;
                if (vertex_ptr)
                    p.w2c.RigidTransform(x_g, y_g, z_g, out_vertex[0],
                                         out_vertex[1], out_vertex[2]);

                bool go = visit_neighbors;
                int x_b = 0, y_b = 0, z_b = 0, block_buf_idx = -1;
                float x_v = 0, y_v = 0, z_v = 0;
                if (go) {
                    x_b = (int)floorf(x_g / p.block_size);
                    y_b = (int)floorf(y_g / p.block_size);
                    z_b = (int)floorf(z_g / p.block_size);
                    x_v = (x_g - (float)x_b * p.block_size) / p.voxel_size;
                    y_v = (y_g - (float)y_b * p.block_size) / p.voxel_size;
                    z_v = (z_g - (float)z_b * p.block_size) / p.voxel_size;
                    block_buf_idx = FindBlock(hv, tab, cache, x_b, y_b, z_b);
                    if (block_buf_idx < 0) go = false;
                }
                if (go) {
                    int x_v_floor = (int)floorf(x_v);
                    int y_v_floor = (int)floorf(y_v);
                    int z_v_floor = (int)floorf(z_v);
                    float ratio_x = x_v - (float)x_v_floor;
                    float ratio_y = y_v - (float)y_v_floor;
                    float ratio_z = z_v - (float)z_v_floor;

                    float sum_r = 0.0f;
                    // Pass 1: the 8 neighbour voxel indices (block look-ups
                    // only at block faces). Pass 2: all weight loads in
                    // flight together, then the per-neighbour terms in the
                    // reference's order (k = 0..7).
                    int64_t lin[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int dx_v = (k & 1) > 0 ? 1 : 0;
                        const int dy_v = (k & 2) > 0 ? 1 : 0;
                        const int dz_v = (k & 4) > 0 ? 1 : 0;
                        // GetLinearIdxAtP, VoxelBlockGridImpl.h:742-782
                        const int xv = x_v_floor + dx_v, yv = y_v_floor + dy_v,
                                  zv = z_v_floor + dz_v;
                        const int x_vn = (xv + res) % res;
                        const int y_vn = (yv + res) % res;
                        const int z_vn = (zv + res) % res;
                        const int dx_b = SignI(xv - x_vn);
                        const int dy_b = SignI(yv - y_vn);
                        const int dz_b = SignI(zv - z_vn);
                        if (dx_b == 0 && dy_b == 0 && dz_b == 0) {
                            lin[k] = (int64_t)block_buf_idx * res3 + zv * res2 +
                                     yv * res + xv;
                        } else {
                            const int kx = x_b + dx_b, ky = y_b + dy_b,
                                      kz = z_b + dz_b;
                            const int nb =
                                    FindBlock(hv, tab, cache, kx, ky, kz);
                            lin[k] = nb < 0 ? -1
                                            : (int64_t)nb * res3 + z_vn * res2 +
                                                      y_vn * res + x_vn;
                        }
                    }
                    weight_t wk[8];
                    float tk[8];
                    color_t ck[8][3];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        wk[k] = lin[k] >= 0 ? weight_base[lin[k]] : (weight_t)0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        tk[k] = (normal_ptr && lin[k] >= 0) ? tsdf_base[lin[k]]
                                                            : 0.0f;
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            ck[k][c] = (render_color && lin[k] >= 0)
                                               ? color_base[lin[k] * 3 + c]
                                               : (color_t)0;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int dx_v = (k & 1) > 0 ? 1 : 0;
                        const int dy_v = (k & 2) > 0 ? 1 : 0;
                        const int dz_v = (k & 4) > 0 ? 1 : 0;
                        const int64_t linear_idx_k = lin[k];

                        if (linear_idx_k >= 0 && wk[k] > 0) {
                            float rx = dx_v * (ratio_x) +
                                       (1 - dx_v) * (1 - ratio_x);
                            float ry = dy_v * (ratio_y) +
                                       (1 - dy_v) * (1 - ratio_y);
                            float rz = dz_v * (ratio_z) +
                                       (1 - dz_v) * (1 - ratio_z);
                            float r = rx * ry * rz;

                            if (FULL) {
                                o_ratio[k] = r;
                                o_mask[k] = 1;
                                o_index[k] = linear_idx_k;
                            }

                            float rdx = ry * rz * (2 * dx_v - 1);
                            float rdy = rx * rz * (2 * dy_v - 1);
                            float rdz = rx * ry * (2 * dz_v - 1);
                            if (FULL) {
                                o_dx[k] = rdx;
                                o_dy[k] = rdy;
                                o_dz[k] = rdz;
                            }

                            if (normal_ptr) {
                                float tsdf_k = tk[k];
                                out_normal[0] += rdx * tsdf_k;
                                out_normal[1] += rdy * tsdf_k;
                                out_normal[2] += rdz * tsdf_k;
                            }
                            if (render_color) {
                                out_color[0] += r * (float)ck[k][0];
                                out_color[1] += r * (float)ck[k][1];
                                out_color[2] += r * (float)ck[k][2];
                            }
                            sum_r += r;
                        }
                    }

                    if (sum_r > 0) {
                        // `sum_r *= 255.0` is a double multiply narrowed back
                        // to float (VoxelBlockGridImpl.h:1095).
                        sum_r = (float)((double)sum_r * 255.0);
                        if (render_color) {
                            out_color[0] /= sum_r;
                            out_color[1] /= sum_r;
                            out_color[2] /= sum_r;
                        }
                        if (normal_ptr) {
                            const float EPSILON = 1e-5f;
                            float norm = sqrtf(out_normal[0] * out_normal[0] +
                                               out_normal[1] * out_normal[1] +
                                               out_normal[2] * out_normal[2]);
                            norm = fmaxf(norm, EPSILON);
                            float nx, ny, nz;
                            p.w2c.Rotate(-out_normal[0] / norm,
                                         -out_normal[1] / norm,
                                         -out_normal[2] / norm, nx, ny, nz);
                            out_normal[0] = nx;
                            out_normal[1] = ny;
                            out_normal[2] = nz;
                        }
                    }
                }
            }
        }

        if (p.tile_cost && lane == 0) {
            // this wave's time on the tile; the slowest wave's counts
            const unsigned long long dt = wall_clock64() - tile_t0;
            atomicMax(&p.tile_cost[tile],
                      ((unsigned long long)p.cost_seq << 32) |
                              (dt < 0xFFFFFFFFull ? dt : 0xFFFFFFFFull));
        }
        if (!inside) continue;  // (no barrier and no wave-wide step below)
        if (depth_ptr) *depth_ptr = out_depth;
        if (vertex_ptr) {
            vertex_ptr[0] = out_vertex[0];
            vertex_ptr[1] = out_vertex[1];
            vertex_ptr[2] = out_vertex[2];
        }
        if (color_ptr) {
            color_ptr[0] = out_color[0];
            color_ptr[1] = out_color[1];
            color_ptr[2] = out_color[2];
        }
        if (normal_ptr) {
            normal_ptr[0] = out_normal[0];
            normal_ptr[1] = out_normal[1];
            normal_ptr[2] = out_normal[2];
        }
#pragma unroll
        for (int k = 0; k < (FULL ? 8 : 0); ++k) {
            if (ratio_ptr) ratio_ptr[k] = o_ratio[k];
            if (ratio_dx_ptr) ratio_dx_ptr[k] = o_dx[k];
            if (ratio_dy_ptr) ratio_dy_ptr[k] = o_dy[k];
            if (ratio_dz_ptr) ratio_dz_ptr[k] = o_dz[k];
            if (index_ptr) index_ptr[k] = o_index[k];
            if (mask_ptr) mask_ptr[k] = o_mask[k];
        }
        if (DIAG)
            atomicMax((unsigned long long*)&p.clocks[3 * (int64_t)tile + 2],
                      (unsigned long long)wall_clock64());
    }
}

}  // namespace
// o3dmi_preload: HIP loads this translation unit's code object at the first
// launch of one of its kernels; asking for a kernel's attributes does it now.
int PreloadRaycast() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(
                                               &RangeFillKernel)) == hipSuccess
                   ? 0
                   : 1;
}

}  // namespace o3dmi

using namespace o3dmi;

extern "C" {

int o3dmi_vbg_estimate_range(const int32_t* block_keys_dev, int64_t n_blocks,
                             float* range_minmax_map_dev,
                             const double* intrinsic, const double* extrinsic,
                             int h, int w, int down_factor,
                             int64_t block_resolution, float voxel_size,
                             float depth_min, float depth_max,
                             o3dmi_stream_t stream) {
    return o3dmi_vbg_estimate_range_dev(
            block_keys_dev, n_blocks, nullptr, range_minmax_map_dev, intrinsic,
            extrinsic, h, w, down_factor, block_resolution, voxel_size,
            depth_min, depth_max, stream);
}

// Internal (o3dmi_vbg_ray_cast_dev with a grid-owned range map / block list):
// the keys are every `key_stride`-th int triple from block_keys_dev; a map the
// last ray cast left clean skips the clearing launch.
int o3dmi_internal_estimate_range(const int32_t* block_keys_dev, int key_stride,
                                  int64_t max_blocks,
                                  const int32_t* n_blocks_dev,
                                  float* range_minmax_map_dev, int map_is_clean,
                                  const double* intrinsic,
                                  const double* extrinsic, int h, int w,
                                  int down_factor, int64_t block_resolution,
                                  float voxel_size, float depth_min,
                                  float depth_max, o3dmi_stream_t stream);

// The NEXT ray-cast launch of this host thread writes every range cell it
// reads back to the {lo, hi} stored in the two floats behind the map's last
// cell (consumed by that launch; whole-image launches with a down factor of 8
// only).
static thread_local int g_reset_range = 0;
int o3dmi_internal_raycast_reset_range(void) {
    g_reset_range = 1;
    return O3DMI_OK;
}
// The NEXT ray-cast launch of this host thread takes its tiles in `order`
// (NULL: index order) and writes their durations to `cost` tagged `seq`;
// the NEXT range-estimate launch sorts `cost` entries tagged `want_seq` into
// `order` first (n_tiles of them). Consumed by those launches.
static thread_local TileOrder g_sort_tiles = {};
static thread_local const int* g_tile_order = nullptr;
static thread_local unsigned long long* g_tile_cost = nullptr;
static thread_local unsigned g_cost_seq = 0;
int o3dmi_internal_raycast_tile_order(unsigned long long* cost, int* order,
                                      int n_tiles, unsigned want_seq,
                                      unsigned seq) {
    g_sort_tiles.cost = cost;
    g_sort_tiles.order = order;
    g_sort_tiles.n_tiles = n_tiles;
    g_sort_tiles.want_seq = want_seq;
    g_tile_order = order;
    g_tile_cost = cost;
    g_cost_seq = seq;
    return O3DMI_OK;
}

// Drops whatever the two calls above left for launches that were never made
// (an error return between the call and its launch): the caller's scope guard.
int o3dmi_internal_raycast_forget(void) {
    g_reset_range = 0;
    g_sort_tiles = TileOrder{};
    g_tile_order = nullptr;
    g_tile_cost = nullptr;
    g_cost_seq = 0;
    return O3DMI_OK;
}

int o3dmi_vbg_estimate_range_dev(const int32_t* block_keys_dev,
                                 int64_t max_blocks,
                                 const int32_t* n_blocks_dev,
                                 float* range_minmax_map_dev,
                                 const double* intrinsic,
                                 const double* extrinsic, int h, int w,
                                 int down_factor, int64_t block_resolution,
                                 float voxel_size, float depth_min,
                                 float depth_max, o3dmi_stream_t stream) {
    return o3dmi_internal_estimate_range(
            block_keys_dev, 3, max_blocks, n_blocks_dev, range_minmax_map_dev,
            0, intrinsic, extrinsic, h, w, down_factor, block_resolution,
            voxel_size, depth_min, depth_max, stream);
}

int o3dmi_internal_estimate_range(const int32_t* block_keys_dev, int key_stride,
                                  int64_t max_blocks,
                                  const int32_t* n_blocks_dev,
                                  float* range_minmax_map_dev, int map_is_clean,
                                  const double* intrinsic,
                                  const double* extrinsic, int h, int w,
                                  int down_factor, int64_t block_resolution,
                                  float voxel_size, float depth_min,
                                  float depth_max, o3dmi_stream_t stream) {
    // (the side channel is this call's whether or not it gets to its launch)
    const TileOrder to = g_sort_tiles;
    g_sort_tiles = TileOrder{};
    O3DMI_REQUIRE(range_minmax_map_dev && intrinsic && extrinsic,
                  "null argument");
    O3DMI_REQUIRE(down_factor > 0 && h >= down_factor && w >= down_factor,
                  "bad image size / down factor");
    O3DMI_REQUIRE(max_blocks == 0 || block_keys_dev != nullptr,
                  "block keys is null");
    hipStream_t s = (hipStream_t)stream;
    int h_down = h / down_factor, w_down = w / down_factor;
    int64_t n_px = (int64_t)h_down * w_down;
    if (!map_is_clean)
        hipLaunchKernelGGL(RangeFillKernel, dim3(GridFor(n_px, kBlock)),
                           dim3(kBlock), 0, s, range_minmax_map_dev, n_px,
                           depth_min, depth_max);
    if (max_blocks > 0) {
        Camera cam = Camera::Make(intrinsic, extrinsic, 1.0f);
        // With a device-resident count the list is usually far shorter than
        // its capacity: a fixed grid (2 workgroups per CU) strides over it.
        const int grid = (n_blocks_dev ? GridFor(max_blocks, 4, kCUs * 2)
                                       : GridFor(max_blocks, 4)) +
                         (to.cost ? 1 : 0);
        hipLaunchKernelGGL(EstimateRangeKernel, dim3(grid), dim3(kBlock), 0, s,
                           block_keys_dev, key_stride, max_blocks, n_blocks_dev,
                           range_minmax_map_dev, cam, h_down, w_down,
                           down_factor, block_resolution, voxel_size, depth_min,
                           depth_max, to);
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_vbg_raycast(o3dmi_hash_t* block_hash, const float* tsdf_dev,
                      const void* weight_dev, const void* color_buf_dev,
                      int grid_dtype, const float* range_map_dev,
                      float* out_depth, float* out_vertex, float* out_color,
                      float* out_normal, int64_t* out_index, uint8_t* out_mask,
                      float* out_ratio, float* out_ratio_dx,
                      float* out_ratio_dy, float* out_ratio_dz,
                      const double* intrinsic, const double* extrinsic, int h,
                      int w, int block_resolution, float voxel_size,
                      float depth_scale, float depth_min, float depth_max,
                      float weight_threshold, float trunc_voxel_multiplier,
                      int range_map_down_factor, o3dmi_stream_t stream) {
    return o3dmi_vbg_raycast_rows(
            block_hash, tsdf_dev, weight_dev, color_buf_dev, grid_dtype,
            range_map_dev, out_depth, out_vertex, out_color, out_normal,
            out_index, out_mask, out_ratio, out_ratio_dx, out_ratio_dy,
            out_ratio_dz, intrinsic, extrinsic, h, w, 0, h, block_resolution,
            voxel_size, depth_scale, depth_min, depth_max, weight_threshold,
            trunc_voxel_multiplier, range_map_down_factor, stream);
}

int o3dmi_vbg_raycast_rows(
        o3dmi_hash_t* block_hash, const float* tsdf_dev, const void* weight_dev,
        const void* color_buf_dev, int grid_dtype, const float* range_map_dev,
        float* out_depth, float* out_vertex, float* out_color,
        float* out_normal, int64_t* out_index, uint8_t* out_mask,
        float* out_ratio, float* out_ratio_dx, float* out_ratio_dy,
        float* out_ratio_dz, const double* intrinsic, const double* extrinsic,
        int h, int w, int row_begin, int row_end, int block_resolution,
        float voxel_size, float depth_scale, float depth_min, float depth_max,
        float weight_threshold, float trunc_voxel_multiplier,
        int range_map_down_factor, o3dmi_stream_t stream) {
    (void)depth_min;
    (void)depth_max;
    // (the side channels are this call's whether or not it gets to its launch)
    const int reset_range = g_reset_range;
    const int* const tile_order = g_tile_order;
    unsigned long long* const tile_cost = g_tile_cost;
    const unsigned cost_seq = g_cost_seq;
    g_reset_range = 0;
    g_tile_order = nullptr;
    g_tile_cost = nullptr;
    g_cost_seq = 0;
    O3DMI_REQUIRE(block_hash && tsdf_dev && weight_dev && range_map_dev &&
                          intrinsic && extrinsic,
                  "null argument");
    O3DMI_REQUIRE(h > 0 && w > 0 && range_map_down_factor > 0, "bad size");
    // The range map is {h / down, w / down, 2} (VoxelBlockGrid.cpp:357-360) and
    // a pixel reads cell (y / down, x / down): an image that is not a multiple
    // of the down factor reads PAST the map (upstream does too: undefined
    // behaviour there; here the march of such a pixel may never end). Refused.
    O3DMI_REQUIRE(h % range_map_down_factor == 0 &&
                          w % range_map_down_factor == 0,
                  "ray cast: height and width must be multiples of "
                  "range_map_down_factor (the range map has h / down x w / "
                  "down cells)");
    O3DMI_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= h &&
                          (row_begin % 8) == 0 &&
                          ((row_end % 8) == 0 || row_end == h),
                  "ray cast rows: a band of whole 8-row tiles of the image");
    if (row_begin == row_end) return O3DMI_OK;
    O3DMI_REQUIRE(grid_dtype == O3DMI_U16 || grid_dtype == O3DMI_F32,
                  "Unsupported value data type combination.");
    RayCastParams p;
    double pose[16];
    InverseTransformation(extrinsic, pose);
    p.c2w = Camera::Make(intrinsic, pose, 1.0f);
    p.w2c = Camera::Make(intrinsic, extrinsic, 1.0f);
    p.h = h;
    p.w = w;
    p.block_resolution = block_resolution;
    p.voxel_size = voxel_size;
    p.block_size = voxel_size * block_resolution;
    p.depth_scale = depth_scale;
    p.weight_threshold = weight_threshold;
    p.sdf_trunc = voxel_size * trunc_voxel_multiplier;
    p.range_down = range_map_down_factor;
    p.w_down = w / range_map_down_factor;
    p.depth = out_depth;
    p.vertex = out_vertex;
    p.color = out_color;
    p.normal = out_normal;
    p.index = (long long*)out_index;
    p.mask = out_mask;
    p.ratio = out_ratio;
    p.ratio_dx = out_ratio_dx;
    p.ratio_dy = out_ratio_dy;
    p.ratio_dz = out_ratio_dz;
    hipStream_t s = (hipStream_t)stream;
    p.steps = nullptr;
    p.clocks = nullptr;
    static const bool count_steps = std::getenv("O3DMI_RAYCAST_STEPS") != nullptr;
    if (count_steps) {
        O3DMI_HIP_CHECK(hipMalloc((void**)&p.steps, sizeof(int) * (size_t)h * w));
        O3DMI_HIP_CHECK(hipMemsetAsync(p.steps, 0, sizeof(int) * (size_t)h * w, s));
        const size_t nt = (size_t)((w + 31) / 32) * ((h + 7) / 8);
        O3DMI_HIP_CHECK(hipMalloc((void**)&p.clocks, sizeof(long long) * 3 * nt));
        O3DMI_HIP_CHECK(hipMemsetAsync(p.clocks, 0, sizeof(long long) * 3 * nt, s));
    }
    // one workgroup per 32 x 8 pixel tile (grid-strided beyond 16 per CU)
    const bool whole = row_begin == 0 && row_end == h;
    p.band_ty0 = whole ? 0 : row_begin / 8;
    p.band_tys = whole ? 0 : (row_end - row_begin + 7) / 8;
    const int64_t n_tiles = (int64_t)((w + 31) / 32) *
                            (whole ? (h + 7) / 8 : p.band_tys);
    p.xcd_bands = n_tiles <= kCUs * 5 ? 1 : 0;
    p.coop = 1;
    // (o3dmi_internal_raycast_reset_range: consumed by this launch)
    if (reset_range && whole && range_map_down_factor == 8) p.coop |= 2;
    // (o3dmi_internal_raycast_tile_order: consumed by this launch; only for
    // launches that run in rounds, whose grid is one workgroup per tile)
    p.tile_order = nullptr;
    p.tile_cost = nullptr;
    p.cost_seq = 0;
    if (tile_cost && whole && !p.xcd_bands && n_tiles <= kCUs * 16) {
        p.tile_order = tile_order;
        p.tile_cost = tile_cost;
        p.cost_seq = cost_seq;
    }
    // a multiple of 8 workgroups: every XCD gets the same number
    dim3 grid((unsigned)((GridFor(n_tiles, 1, kCUs * 16) + 7) & ~7)),
            block(kBlock);
    const bool full = out_index || out_mask || out_ratio || out_ratio_dx ||
                      out_ratio_dy || out_ratio_dz;
#define O3DMI_RAYCAST_R(WT, CT, FULL, RES, DIAG)                              \
    hipLaunchKernelGGL((RayCastKernel<WT, CT, FULL, RES, DIAG>), grid, block, \
                       0, s, block_hash->view, p, tsdf_dev,                   \
                       (const WT*)weight_dev, (const CT*)color_buf_dev,       \
                       range_map_dev)
    // (the diagnostics exist for the common case only: 16^3 blocks, slim maps)
#define O3DMI_RAYCAST(WT, CT, FULL)                                           \
    do {                                                                      \
        if (block_resolution != 16) O3DMI_RAYCAST_R(WT, CT, FULL, 0, false);  \
        else if (count_steps && !FULL)                                        \
            O3DMI_RAYCAST_R(WT, CT, false, 16, true);                         \
        else O3DMI_RAYCAST_R(WT, CT, FULL, 16, false);                        \
    } while (0)
    if (grid_dtype == O3DMI_F32) {
        if (full) O3DMI_RAYCAST(float, float, true);
        else O3DMI_RAYCAST(float, float, false);
    } else {
        if (full) O3DMI_RAYCAST(uint16_t, uint16_t, true);
        else O3DMI_RAYCAST(uint16_t, uint16_t, false);
    }
#undef O3DMI_RAYCAST
#undef O3DMI_RAYCAST_R
    O3DMI_HIP_CHECK(hipGetLastError());
    if (count_steps) {
        std::vector<int> hs((size_t)h * w);
        O3DMI_HIP_CHECK(hipMemcpyAsync(hs.data(), p.steps, sizeof(int) * hs.size(),
                                       hipMemcpyDeviceToHost, s));
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(p.steps);
        // per-pixel and per-8x8-tile (= per wave) statistics
        long long sum = 0, wsum = 0;
        int mx = 0, nw = 0, wmax_max = 0;
        const std::vector<int> packed = hs;  // steps | plain passes << 8 | cooperative passes << 16 | crawl steps << 24
        for (int& v : hs) v &= 255;
        for (int v : hs) { sum += v; mx = v > mx ? v : mx; }
        std::vector<int> wmaxes;
        for (int ty = 0; ty < h / 8; ++ty)
            for (int tx = 0; tx < w / 8; ++tx) {
                int m = 0;
                for (int dy = 0; dy < 8; ++dy)
                    for (int dx = 0; dx < 8; ++dx) {
                        const int v = hs[(size_t)(ty * 8 + dy) * w + tx * 8 + dx];
                        m = v > m ? v : m;
                    }
                wsum += m; ++nw; wmax_max = m > wmax_max ? m : wmax_max;
                wmaxes.push_back(m);
            }
        std::sort(wmaxes.begin(), wmaxes.end());
        std::fprintf(stderr,
                     "[o3dmi] raycast steps: mean per ray %.1f, max %d; per "
                     "wave tile: mean of max %.1f, p50 %d, p90 %d, p99 %d, max %d\n",
                     (double)sum / hs.size(), mx, (double)wsum / nw,
                     wmaxes[wmaxes.size() / 2], wmaxes[wmaxes.size() * 9 / 10],
                     wmaxes[wmaxes.size() * 99 / 100], wmax_max);
        const int th[6] = {8, 16, 24, 32, 48, 64};
        long long over[6] = {0, 0, 0, 0, 0, 0}, steps_over[6] = {0};
        for (int v : hs)
            for (int k = 0; k < 6; ++k)
                if (v > th[k]) { ++over[k]; steps_over[k] += v - th[k]; }
        std::fprintf(stderr, "[o3dmi] raycast rays with more than");
        for (int k = 0; k < 6; ++k)
            std::fprintf(stderr, " %d steps: %lld (%lld steps beyond);", th[k],
                         over[k], steps_over[k]);
        std::fprintf(stderr, "\n");
        // workgroup clocks (100 MHz): when the tiles start, how long their
        // march and their whole work lasts, when the last one ends
        const size_t nt = (size_t)((w + 31) / 32) * ((h + 7) / 8);
        std::vector<long long> ck(3 * nt);
        O3DMI_HIP_CHECK(hipMemcpy(ck.data(), p.clocks,
                                  sizeof(long long) * ck.size(),
                                  hipMemcpyDeviceToHost));
        (void)hipFree(p.clocks);
        long long t0 = ck[0];
        for (size_t k = 0; k < nt; ++k) t0 = ck[3 * k] < t0 ? ck[3 * k] : t0;
        std::vector<double> start, march, total, end;
        for (size_t k = 0; k < nt; ++k) {
            start.push_back((ck[3 * k] - t0) * 0.01);
            march.push_back((ck[3 * k + 1] - ck[3 * k]) * 0.01);
            total.push_back((ck[3 * k + 2] - ck[3 * k]) * 0.01);
            end.push_back((ck[3 * k + 2] - t0) * 0.01);
        }
        const auto q = [](std::vector<double> v, double f) {
            std::sort(v.begin(), v.end());
            return v[(size_t)((v.size() - 1) * f)];
        };
        // the slowest tiles, with their rays' look-up counts
        std::vector<size_t> order(nt);
        for (size_t k = 0; k < nt; ++k) order[k] = k;
        std::sort(order.begin(), order.end(),
                  [&](size_t a, size_t b) { return total[a] > total[b]; });
        const int tiles_x = (w + 31) / 32, tiles_y = (h + 7) / 8;
        for (int r = 0; r < 6 && r < (int)nt; ++r) {
            const size_t tile = order[r];
            const int ty = p.xcd_bands ? (int)(tile % tiles_y) : (int)(tile / tiles_x);
            const int tx = p.xcd_bands ? (int)(tile / tiles_y) : (int)(tile % tiles_x);
            int m[4] = {0, 0, 0, 0};
            long long sm[4] = {0, 0, 0, 0};
            for (int dy = 0; dy < 8; ++dy)
                for (int dx = 0; dx < 32; ++dx) {
                    const int y = ty * 8 + dy, x = tx * 32 + dx;
                    if (y >= h || x >= w) continue;
                    const unsigned v = (unsigned)packed[(size_t)y * w + x];
                    for (int c = 0; c < 4; ++c) {
                        const int f = (v >> (8 * c)) & 255;
                        m[c] = f > m[c] ? f : m[c];
                        sm[c] += f;
                    }
                }
            std::fprintf(stderr,
                         "[o3dmi] raycast slow tile (%d, %d): start %.1f march "
                         "%.1f total %.1f us; per ray max / mean: steps %d / "
                         "%.1f, plain passes %d / %.1f, cooperative passes %d / "
                         "%.1f, crawl steps %d / %.1f\n",
                         tx, ty, start[tile], march[tile], total[tile], m[0],
                         sm[0] / 256.0, m[1], sm[1] / 256.0, m[2],
                         sm[2] / 256.0, m[3], sm[3] / 256.0);
        }
        const char* names[4] = {"start", "march", "total", "end"};
        const std::vector<double>* vs[4] = {&start, &march, &total, &end};
        for (int k = 0; k < 4; ++k)
            std::fprintf(stderr,
                         "[o3dmi] raycast workgroup %s us: p10 %.1f p50 %.1f "
                         "p90 %.1f p99 %.1f max %.1f\n",
                         names[k], q(*vs[k], 0.1), q(*vs[k], 0.5),
                         q(*vs[k], 0.9), q(*vs[k], 0.99), q(*vs[k], 1.0));
    }
    return O3DMI_OK;
}

}  // extern "C"
