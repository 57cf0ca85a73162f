// Internal interface of the device-count VoxelDownSample (pointcloud.hip): one
// pyramid level as a string of launches with no host wait, so that the ICP
// driver can chain the levels of a cloud and read all counts back at once.
#pragma once

#include <vector>

#include "common.h"

namespace o3dmi {

// positions {n,3} (+ optional attribute {n,3}, averaged the same way) of
// dtype O3DMI_F32 / O3DMI_F64 -> out_pos / out_attr, sized for n_max rows.
//   n_dev   device int holding the live point count (<= n_max), or NULL to
//           use n_max itself
//   m_dev   device int receiving the voxel count
//   err_dev device int: kErrKeyRange is OR-ed in for out-of-range coordinates
// Clouds of up to 2^20 points take the tiled form (insert, then partition +
// reduce per level), whose buffers live in a persistent workspace per host
// thread, device and `chain` (0 or 1; the calls of one chain must be
// stream-ordered). Larger clouds take the seven-launch sort, with a
// persistent workspace of its own (`scratch` is unused nowadays and kept for
// the callers' release paths).
//   next_voxel_size  > 0: the caller's NEXT call on this chain will down-
//           sample out_pos (same n_max, dtype) by this voxel size -- a pyramid
//           built from its own output. The tiled form then inserts the
//           output into the next level's table in its last launch, and the
//           next call starts at its second one. A next call that turns out
//           different is still correct (the insert is discarded). Only when
//           that next call is the ONLY pass over out_pos (no attribute passes
//           in between).
//   from_previous    this call IS such a next call: pos is the out_pos of the
//           chain's previous call (anything else discards a pending insert).
int VdsAsync(const void* pos, const void* attr, int64_t n_max, const int* n_dev,
             int dtype, double voxel_size, void* out_pos, void* out_attr,
             int* m_dev, int* err_dev, std::vector<void*>& scratch,
             hipStream_t s, int chain = 0, double next_voxel_size = 0,
             bool from_previous = false);

// The same level for one or two clouds at once: the launches of the tiled form
// take two jobs (blockIdx.y), so the source and the target pyramid of
// MultiScaleICP advance level by level in the SAME launches on one stream
// (round 5: two chains of launches on two streams, 16 launches per frame
// pair of pyramids; now 7 + 1). The jobs of one call must name different
// chains; clouds beyond the tiled form fall back to one call each.
// The counts of a chain posted to its host mailbox by the chain's LAST level
// launch itself (the reduce launch knows the level's voxel count when it
// starts): counts[0..n - 1) -- the level written by this call taken from the
// launch, the others from memory -- and the error word counts[kCountsErr] as
// value n - 1 go out as a sealed block (mailbox.h), are copied to
// counts[kCountsKeep + i], and the error word is cleared. Only on a level
// without a fused next-level insert.
struct VdsPost {
    int* counts = nullptr;   // NULL: no post
    int n = 0;               // levels + 1
    double* mail_data = nullptr;
    int* mail_flag = nullptr;
    int mail_seq = 0;
};

struct VdsLevelJob {
    VdsPost post;
    const void* pos = nullptr;
    const void* attr = nullptr;
    int64_t n_max = 0;
    const int* n_dev = nullptr;
    double voxel_size = 0;
    void* out_pos = nullptr;
    void* out_attr = nullptr;
    int* m_dev = nullptr;
    int* err_dev = nullptr;
    int chain = 0;
    double next_voxel_size = 0;
    bool from_previous = false;
};
// *posted (optional): whether the jobs' VdsPost requests were carried out
// (tiled form, every job of the call asking); if not, the caller posts with
// PostCountsPairAsync as before.
int VdsPairAsync(const VdsLevelJob* jobs, int n_jobs, int dtype,
                 std::vector<void*>& scratch, hipStream_t s,
                 bool* posted = nullptr);

// The calling thread's workspaces of `chain` on the current device may have
// been left dirty by a chain that was abandoned mid-way (error return between
// its first launch and the wait for its counts): their next user discards
// them and starts from freshly initialised buffers.
void VdsChainInvalidate(int chain);

// counts_dev[0..n - 1) and the error word counts_dev[kCountsErr] (int) ->
// mail_data[0..n) (as float64) + sequence word `mail_seq` (mailbox.h), on
// stream s; the words are zeroed afterwards.
int PostCountsAsync(int* counts_dev, int n, double* mail_data, int* mail_flag,
                    int mail_seq, hipStream_t s);
// Two chains' counts (built in the same launches) posted by one launch; every
// count is also copied to counts[kCountsKeep + i], where it stays until the
// next posting launch (counts buffers hold 2 * kCountsKeep ints).
constexpr int kCountsKeep = 32;
// The chain's error word lives at a FIXED slot of the counts buffer (not
// behind the last level: chains of different depths share the buffer, and a
// post that leaves with the last level's launch cannot zero the count that
// launch is still reading); a post delivers it as value n - 1.
constexpr int kCountsErr = kCountsKeep - 1;
int PostCountsPairAsync(int* counts_a, double* mail_data_a, int* mail_flag_a,
                        int mail_seq_a, int* counts_b, double* mail_data_b,
                        int* mail_flag_b, int mail_seq_b, int n,
                        hipStream_t s);

}  // namespace o3dmi
