// rt_kernels.h — launch interface between the C ABI (rt_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_types.h"

typedef float rt_f32x4 __attribute__((ext_vector_type(4)));

// Margin of the phase-1 filter, relative to a * (|o|^2 + |c|^2 + r^2) ("phase-1 filter").
#define RT_P1_MARGIN 0x1p-16f

// Kernel arguments (by value, lands in SGPRs). Scene arrays are the device-side re-layout built
// by rt_set_scene: sph_p1[pair] = {cx0,cx1, cy0,cy1, cz0,cz1, K'0,K'1}, K' = (|c|^2 - r^2) - RT_P1_MARGIN *
// (|c|^2 + r^2) (phase-1 filter, scalar loads, two spheres per packed instruction), sph_test[i] = {cx, cy, cz, fl(r*r)} (exact test of the candidates),
// sph_shade[i] = {cx, cy, cz, r}, sph_col[i] = {material.color.rgb, +-fl(1/r)} (per-hit fetch).
struct rt_kargs {
  float p2w[16];  // CameraRaw.pixel_to_world, column-major
  float pos[3];   // CameraRaw.pos.xyz
  uint32_t width, height;   // textureDimensions(color_buffer)
  uint32_t cov_w, cov_h;    // pixels a pass covers (floor16 or full)
  uint32_t tiles_x, n_tiles;  // 8x8 tiles over the covered area
  uint32_t rank, world;     // tile t belongs to this launch iff t % world == rank
  uint32_t my_tiles;        // number of tiles of this rank: tile = j * world + rank, j < my_tiles
  uint32_t n_units;         // queue length: my_tiles, or my_tiles * n_groups in a staged launch
  uint32_t queue_chunk;     // units a wave takes per atomic on the queue head (1; more when units are many and short: rt_api.cpp); 0: no queue, wave w takes units w + k * waves
  uint32_t reject_tries;    // rejection-sampling rounds per loop iteration (>= 1)
  uint32_t spp;
  uint32_t first_frame, n_frames;
  uint32_t seed_t0;              // bits(time) * 87189 + 189273 of the first frame (single-frame launches; wgsl:260)
  float cw0;                     // f32(first_frame) / f32(first_frame + 1) (single-frame launches)
  const uint2* frame_table;      // n_frames x {seed_t, bits of cw} when n_frames > 1 (device), else null
  float inv_spp;                 // 1/spp when spp is a power of two (exact), else 0
  uint32_t batch_lanes;          // lanes that must be parked before frame ends / pixel hand-outs are serviced
  uint32_t n_spheres;
  uint32_t flags;
  uint32_t p1_tame;              // bit 0: every sphere is inside the magnitude range the phase-1 error bound assumes; bit 1: bit 0 and
                                 // the camera position too (|pos|^2 <= 2^40, the FMA form the kernel uses for every other ray origin)
  const float* sph_p1;           // n_spheres rounded up to a multiple of 4, x 4 floats
  const rt_f32x4* sph_test;
  const rt_f32x4* sph_shade;
  const rt_f32x4* sph_col;
  void* accum;                   // RGBA8: uint32 per pixel; F32: float4 per pixel
  rt_f32x4* staging;             // staged launch: n_frames x my_tiles x 64 frame colours (this rank's own tiles, pixel (v & 7) * 8 + (u & 7) within the tile), else null
  uint32_t* rng_out;             // optional
  unsigned long long* counters;  // 14 x u64, COUNT variants only
  uint32_t* queue;               // tile queue head, zeroed on the stream before every launch
  const uint32_t* tile_order;    // optional: permutation of [0, my_tiles): queue position -> tile
  uint32_t* tile_cost;           // optional: per-tile trace iterations of this launch (zeroed before it)
  const uint32_t* tile_masks;    // optional, per tile of this rank (rt_tile_masks_kernel): 0 = sky tile (no primary ray can hit a sphere),
                                 // else the spheres its primary rays can hit (<= 32 spheres: bit cnt_pad-1-i; else all ones)
  const uint16_t* tile_lists;    // optional, scenes of > 32 spheres, 16 x u16 per tile of this rank: [0] = number of spheres a
                                 // primary ray of the tile can hit (<= 15; 0xffff = more, or rays not provably tame), [1..] their
                                 // indices in ascending order
  // GRID (scenes of more than 64 spheres, optional: rt_set_scene builds it when it pays): a uniform grid over the spheres; a cell lists
  // the spheres whose INFLATED ball (radius^2 + 2^-18 S_max, plus the traversal's slack) touches it. One buffer, addressed by byte
  // offsets from its base: the cell table at grid_cell0 (per cell the offset of its first list entry, 0 = empty cell), entries of
  // 32 bytes {cx, cy, cz, K' (the filter's constants of sph_p1), sphere index | last-of-its-cell << 31, -, -, -} (rt_api.cpp, build_grid).
  // The general trace of a ray whose origin satisfies |o|^2 <= grid_o2max walks the cells along the ray and runs the phase-1 filter
  // on the listed spheres only (rt_kernels.hip, "grid"); other rays take the loop over all spheres.
  const uint32_t* grid_cells;      // the buffer; null: no grid
  uint32_t grid_cell0;
  float grid_org[3], grid_inv[3];  // cell coordinates of a point p: (p - org) * inv
  uint32_t grid_dim[3], grid_ncells;
  float grid_o2max;
  uint32_t frame_group, n_groups;  // staged launch: a unit of work is (tile, frame_group consecutive frames: a power of two), n_groups = ceil(n_frames / frame_group)
};

// n_waves persistent single-wave workgroups pull tiles from a.queue until a.my_tiles are handed out
// split: a SPLIT variant - primary rays traced against their tile's candidates: a.tile_masks for <= 32 spheres, a.tile_lists
// for > 64 (RT_SMALL_MAX); never for counting launches
hipError_t rt_launch_path_kernel(const rt_kargs& a, uint32_t n_waves, bool strict, bool count, bool split, hipStream_t stream);
// folds the staged frame colours into the accumulation image, in frame order, for the pixels of this rank's tiles
hipError_t rt_launch_combine_frames(const rt_kargs& a, hipStream_t stream);
// masks[j] = spheres a primary ray of this rank's j-th tile can hit, 0 for a sky tile (conservative, double precision);
// lists (optional, scenes of > 32 spheres): the same as a short list per tile (rt_kargs::tile_lists)
hipError_t rt_launch_tile_masks(const rt_kargs& a, uint32_t* masks, uint16_t* lists, hipStream_t stream);
// order = this rank's tiles, dearest first by `cost`, the sky tiles (masks[j] == 0, or without masks cost[j] == 0) spread evenly between them
hipError_t rt_launch_tile_order(const uint32_t* cost, const uint32_t* masks, uint32_t* order, uint32_t n, hipStream_t stream);
// table[i] = {time_bits_host[i] * 87189 + 189273, bits of f32(first_frame + i) / f32(first_frame + i + 1)}, i < n <= 256; the time
// bits travel as kernel arguments (copied at the call)
hipError_t rt_launch_frame_table(uint2* table, const uint32_t* time_bits_host, uint32_t first_frame, uint32_t n, hipStream_t stream);
hipError_t rt_launch_quantize(const void* src_f32x4, void* dst_rgba8, size_t n_pixels, hipStream_t stream);
