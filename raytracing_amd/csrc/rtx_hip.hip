// HIP (gfx950) implementation of include/rtx.h: kernels, scene upload, C ABI.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see raytracing_amd/build.py).
// -ffp-contract=off is part of the contract: the pixel arithmetic must not be fused (see
// rtx_device.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "lbvh_build.hpp"
#include "sah_build.hpp"
#include "rtx_device.hpp"
#include "rtx_trace.hpp"
#include "host_copy.hpp"

using namespace rtx;

namespace {

constexpr int kBlock = 256;     // 4 waves per block
constexpr int kStackDepth = 32; // LDS traversal stack entries per lane (checked against the tree's depth)
constexpr int kMaxSpheres = 1 << 24;

struct DevCamera {
    float ox, oy, oz;
    float m[9];
    float tan_half_fov;
    int add_origin;
};

struct DevFrame {
    int width, height, depth, row_first, row_step, rows;
    int spp;  // RTX_OPT_SPP: 1 or 4
    // k_render's LDS per lane: `stack_entries` dwords of traversal stack (as deep as the scene's trees), then the heads of
    // `lds_frames` frames of the recursion, six dwords each
    int stack_entries, lds_frames;
};

// src/Scene.cpp:217-220; (dx, dy) = offset of the sample from the pixel centre, 0 in the reference
RTX_DEV V3 primaryDir(const DevCamera &cam, int width, int height, int i, int j, float dx = 0.0f, float dy = 0.0f) {
    // (also in the tolerance build, which contracts elsewhere: every pixel's whole tree hangs on this direction)
#pragma clang fp contract(off)
    float x = (((float)j + 0.5f) + dx) - (float)width / 2.0f;
    float y = -(((float)i + 0.5f) + dy) + (float)height / 2.0f;
    float z = (float)(-height) / (2.0f * cam.tan_half_fov);
    // mat3 * vec3: include/glm/detail/type_mat3x3.inl:468-474
    V3 d = v3(cam.m[0] * x + cam.m[3] * y + cam.m[6] * z, cam.m[1] * x + cam.m[4] * y + cam.m[7] * z,
              cam.m[2] * x + cam.m[5] * y + cam.m[8] * z);
    if (cam.add_origin) d = d + v3(cam.ox, cam.oy, cam.oz);
    return normalize(d);
}

// getRGBColor, src/Scene.cpp:204-210: 255 * clamp(c, 0, 1), truncated
RTX_DEV unsigned char toByte(float c) { return (unsigned char)(255 * fmaxf(0.0f, fminf(1.0f, c))); }

template <bool STATS>
RTX_DEV void flushCounters(const LaneCounters<STATS> &lc, Counters *out) {
    if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < CNT_SLOTS; ++k) {
            unsigned long long x = lc.v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&out->v[k], x);
        }
    }
}

// Work distribution.  The frame is cut into 8x8-pixel tiles, listed in `order` (expensive ones
// first, see tileOrder) and dealt round-robin into kQueues queues: queue k owns entries k,
// k + kQueues, ...  Each queue has a head counter in a cache line of its own.  The grid is sized to
// fill the chip once; every WAVE then works on its own: it takes a tile from its queue (one atomic
// per 64 pixels, on one of kQueues addresses -- atomics on a single address serialise at about
// 10 ns each, far too slow for 130 000 tiles per 4K frame), and when that queue is empty it looks at
// all heads at once (one per lane) and moves to the next queue that still has entries.  No wave
// slot ever waits for the slowest wave of a block, and no block pays the LDS staging twice.
//
// Inside the wave each lane owns one pixel of the tile and advances its path by ONE RAY per trip of
// the loop (closest-hit or shadow ray, see pathConsume), so a lane whose ray left the scene goes on
// with its next ray while its neighbours cast shadow rays.  A tile stays with its wave until all 64
// paths are complete: handing finished lanes pixels of the next tile keeps more lanes busy, but
// mixes rays of different screen regions in one wave, and the loss of coherence (more divergent
// category tests, longer mesh walks) costs more than the idle lanes do.
constexpr unsigned kQueues = 64, kHeadStride = 32;
constexpr unsigned kQuadFlag = 0x80000000u, kTileMask = 0x0fffffffu;

// Quad mode is for the tiles that lasted longer in the previous frame than a share (kQuadShare) of what
// the whole frame would take if its work were spread evenly over all wave slots -- the ones that would
// hold the frame up: they are entered as four quarters, all others once.  Never tiles shorter than
// min_ticks, never more than `cap` tiles.
constexpr float kQuadShare = 0.8f;
constexpr unsigned kQuadCostScale = 3;
constexpr int kSideLaunchTiles = 16;   // tiles per resident wave up to which the quad-mode tiles get a launch of their own  // the quad-mode launch: time of a tile's first quarter -> stand-in for the whole tile's time

// The tile list of a frame from the costs of the previous one, without a sort: the tiles are dealt into
// bins by cost on a logarithmic scale of eight steps per octave, most expensive first.  The bins above
// the quad-mode threshold (to the nearest bin edge) are entered as quarters (out must hold n + 3 * cap entries),
// the position from which the bins lie below the average cost is reported for k_render's batches.
// Inside a bin the order is whatever the atomics make it; the bins are narrow enough (9 %) for the
// queues to balance as well as with the exact order.  Three small launches (histogram per workgroup,
// scan, scatter), a few tens of microseconds: cheap enough to run for every frame of a moving camera.
constexpr unsigned kCostBins = 256, kBinBlocks = 64, kBinThreads = 1024;
RTX_DEV unsigned costBin(unsigned cost) {  // descending: bin 0 holds the most expensive tiles
    const unsigned bits = __float_as_uint((float)cost) >> 20;  // exponent and three mantissa bits
    const unsigned b = bits > (127u << 3) ? bits - (127u << 3) : 0u;
    return kCostBins - 1u - (b < kCostBins ? b : kCostBins - 1u);
}
// atomicAdd(&counters[bin], 1) for every lane with `valid`; returns the lane's own old value.  Most
// tiles of a frame fall into a handful of bins (sky, water), and 64 atomics on one LDS address go one
// after the other: the lanes that share the bin of the first lane (two rounds of that) use one atomic
// together, whoever is left adds for itself.  Must be called by all lanes of the wave together.
RTX_DEV unsigned binSlot(unsigned *counters, unsigned bin, bool valid) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned slot = 0;
    unsigned long long todo = __ballot(valid);
    for (int round = 0; round < 2 && todo; ++round) {
        const unsigned leader = (unsigned)__builtin_ctzll(todo);
        const unsigned lb = __builtin_amdgcn_readlane(bin, leader);
        const unsigned long long same = __ballot(valid && bin == lb) & todo;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&counters[lb], (unsigned)__popcll(same));
        base = __builtin_amdgcn_readlane(base, leader);
        if ((same >> lane) & 1ull) slot = base + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) slot = atomicAdd(&counters[bin], 1u);
    return slot;
}
// workgroup g owns tiles [g * per, (g + 1) * per)
__global__ __launch_bounds__(kBinThreads) void k_bin_count(const unsigned *__restrict__ cost, unsigned n, unsigned per,
                                                           unsigned *__restrict__ block_hist, unsigned long long *__restrict__ block_sum) {
    __shared__ unsigned s_hist[kCostBins];
    __shared__ unsigned long long s_sum[kBinThreads / 64];
    const unsigned t = threadIdx.x, first = blockIdx.x * per, last = first + per < n ? first + per : n;
    for (unsigned b = t; b < kCostBins; b += kBinThreads) s_hist[b] = 0u;
    __syncthreads();
    unsigned long long sum = 0;
    for (unsigned base = first; base < last; base += kBinThreads) {
        const unsigned i = base + t;
        const bool valid = i < last;
        const unsigned cst = valid ? cost[i] : 0u;
        sum += cst;
        binSlot(s_hist, costBin(cst), valid);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((t & 63u) == 0u) s_sum[t >> 6] = sum;
    __syncthreads();
    for (unsigned b = t; b < kCostBins; b += kBinThreads) block_hist[blockIdx.x * kCostBins + b] = s_hist[b];
    if (t == 0) {
        unsigned long long total = 0;
        for (unsigned w = 0; w < kBinThreads / 64; ++w) total += s_sum[w];
        block_sum[blockIdx.x] = total;
    }
}
// one workgroup of kCostBins threads: block_hist[g][b] becomes the list position at which workgroup g
// starts to write its tiles of bin b; head = {entries, first cheap entry, quad-mode tiles, quad-mode bins}
// cheap_cost: tiles in the bins below this cost's bin count as cheap (k_render takes them in batches); 0 = below
// the average cost
__global__ __launch_bounds__(kCostBins) void k_bin_scan(unsigned *__restrict__ block_hist, const unsigned long long *__restrict__ block_sum,
                                                        unsigned n, unsigned wave_slots, float quad_share, unsigned min_ticks, unsigned cap,
                                                        unsigned cheap_cost, unsigned *__restrict__ n_out, unsigned *__restrict__ head,
                                                        unsigned *quads_for_host) {
    __shared__ unsigned s_total[kCostBins], s_start[kCostBins];
    const unsigned b = threadIdx.x;
    unsigned at = 0;
    for (unsigned g = 0; g < kBinBlocks; ++g) {
        const unsigned c = block_hist[g * kCostBins + b];
        block_hist[g * kCostBins + b] = at;
        at += c;
    }
    s_total[b] = at;
    __syncthreads();
    if (b == 0) {
        unsigned long long sum = 0;
        for (unsigned g = 0; g < kBinBlocks; ++g) sum += block_sum[g];
        const float even = (float)sum / (float)(wave_slots ? wave_slots : 1u);
        const float share = quad_share * even;
        const unsigned limit = share > (float)min_ticks ? (unsigned)share : min_ticks;
        const unsigned mean = cheap_cost ? cheap_cost : (unsigned)(sum / (n ? n : 1u));
        // quad mode for the bins at or above the limit's, as long as that stays within `cap` tiles
        const unsigned limit_bin = costBin(limit);
        unsigned heavy = 0, heavy_bins = 0;
        for (; heavy_bins <= limit_bin && heavy_bins < kCostBins; ++heavy_bins) {
            if (heavy + s_total[heavy_bins] > cap) break;
            heavy += s_total[heavy_bins];
        }
        unsigned pos = 0, cheap_from = n;
        const unsigned mean_bin = costBin(mean);
        for (unsigned k = 0; k < kCostBins; ++k) {
            if (k == mean_bin + 1u) cheap_from = pos;
            s_start[k] = pos, pos += s_total[k];
        }
        if (cheap_from < heavy) cheap_from = heavy;
        n_out[0] = n + 3u * heavy;
        n_out[1] = 3u * heavy + cheap_from;
        n_out[2] = 4u * heavy;  // the quarters come first
        if (quads_for_host) *quads_for_host = heavy;  // (page-locked host memory: sizes the grid of the next quad-mode launches)
        head[0] = heavy, head[1] = heavy_bins;
    }
    __syncthreads();
    for (unsigned g = 0; g < kBinBlocks; ++g) block_hist[g * kCostBins + b] += s_start[b];
}
__global__ __launch_bounds__(kBinThreads) void k_bin_scatter(const unsigned *__restrict__ cost, unsigned n, unsigned per,
                                                             const unsigned *__restrict__ block_base, const unsigned *__restrict__ head,
                                                             unsigned *__restrict__ out) {
    __shared__ unsigned s_cursor[kCostBins];
    const unsigned t = threadIdx.x, first = blockIdx.x * per, last = first + per < n ? first + per : n;
    for (unsigned b = t; b < kCostBins; b += kBinThreads) s_cursor[b] = block_base[blockIdx.x * kCostBins + b];
    __syncthreads();
    const unsigned heavy = head[0], heavy_bins = head[1];
    for (unsigned base = first; base < last; base += kBinThreads) {
        const unsigned i = base + t;
        const bool valid = i < last;
        const unsigned b = costBin(valid ? cost[i] : 0u);
        const unsigned pos = binSlot(s_cursor, b, valid);
        if (valid) {
            if (b < heavy_bins) {
                for (unsigned sub = 0; sub < 4u; ++sub) out[4u * pos + sub] = i | (sub << 28) | kQuadFlag;
            } else {
                out[3u * heavy + pos] = i;
            }
        }
    }
}

// The first frame of a view has no costs to go by: its tiles are put into three classes from bounding
// boxes alone -- those that can see the mesh, those that can see a mirror or glass sphere, the rest --
// a longest-job-first schedule (a wave that walks the mesh at every bounce runs for hundreds of
// microseconds; if it only starts halfway through the dispatch the chip idles waiting for it).  The
// classes are written as stand-in costs and go through the same k_bin_* launches as measured costs do.
constexpr int kMaxRects = kMaxSmallPrims + 1;
constexpr unsigned kClassCost0 = 4096u, kClassCost1 = 1024u, kClassCost2 = 1u;
constexpr unsigned kClassCheapCost = 512u;  // class 2 counts as cheap
struct TileRects {
    int n;
    float r[kMaxRects][4];  // column min, max, image row min, max
    int cls[kMaxRects];
};
__global__ __launch_bounds__(256) void k_tile_classes(TileRects rects, unsigned n, unsigned tiles_x, int row_first, int row_step,
                                                      unsigned *__restrict__ cost) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned bx = i % tiles_x, by = i / tiles_x;
    const float c0 = (float)(bx * 8u), c1 = (float)(bx * 8u + 7u);
    const float r0 = (float)(row_first + (long long)by * 8 * row_step);
    const float r1 = (float)(row_first + ((long long)by * 8 + 7) * row_step);
    // Inside a class, the tiles near the middle of their rectangle first (up to twice the class's stand-in cost, which
    // keeps the classes apart): an object fills the middle of its bounding box, the corners are mostly empty, and the
    // tiles that last longest -- the mesh mirrored in itself -- should be under way in the first round of the dispatch.
    unsigned best = kClassCost2;
    for (int k = 0; k < rects.n; ++k)
        if (c1 >= rects.r[k][0] - 1 && c0 <= rects.r[k][1] + 1 && r1 >= rects.r[k][2] - 1 && r0 <= rects.r[k][3] + 1) {
            const float half_w = 0.5f * (rects.r[k][1] - rects.r[k][0]) + 4.0f, half_h = 0.5f * (rects.r[k][3] - rects.r[k][2]) + 4.0f * (float)row_step;
            const float dx = fabsf(0.5f * (c0 + c1) - 0.5f * (rects.r[k][0] + rects.r[k][1])) / half_w;
            const float dy = fabsf(0.5f * (r0 + r1) - 0.5f * (rects.r[k][2] + rects.r[k][3])) / half_h;
            const float off = fminf(fmaxf(dx, dy), 1.0f);
            const unsigned base = rects.cls[k] == 0 ? kClassCost0 : kClassCost1;
            const unsigned cst = base + (unsigned)((float)(base - 1u) * (1.0f - off));
            best = cst > best ? cst : best;
        }
    cost[i] = best;
}

// The pixel a lane works on; with several samples per pixel (WIDE build only) the lane traces them
// one after the other.
struct PixelJob {
    int row, col;  // image row, column
};

// Advances the lane's path, one ray per trip, until it is complete; the pixel's B,G,R bytes
// (include/bitmap_image.hpp:207-209) go to `bgr` (the wave's tile buffer in LDS).
template <bool STATS, bool WIDE, bool LOOK = false>
__device__ __forceinline__ void tracePaths(const DevScene &s, const SmallScene &q, const DevCamera &cam, const DevFrame &fr, PixelJob job,
                                           PathState &st, FrameStack &frames, bool active, const int quad_j, unsigned char *bgr, int *stack,
                                           LaneCounters<STATS> &lc) {
    V3 sum = v3(0.0f, 0.0f, 0.0f);
    int sample = 0;
    while (active) {
        bool hit = false;
        Hit h;
        float md = 0.0f;
        if (pathCasts(st, quad_j)) {
            pathCountRay<STATS>(st, quad_j, lc);
            const bool shadow = st.light >= 0 && !(LOOK && pathRayIsAhead(st, quad_j));
            hit = sceneIntersect<STATS, WIDE>(s, q, st.orig, st.dir, shadow, st.t_exit, h, md, stack, kBlock, lc);
        }
        V3 c;
        if (pathConsume<STATS, WIDE, LOOK>(s, q, st, frames, hit, h, md, quad_j, c, lc)) {
            if constexpr (WIDE) {
                if (fr.spp > 1) {
                    // RTX_OPT_SPP = 4: (((c0 + c1) + c2) + c3) * 0.25 over the rotated grid of include/rtx.h
                    sum = sample == 0 ? c : sum + c;
                    if (++sample < 4) {
                        const float dx = sample == 1 ? -0.375f : (sample == 2 ? 0.375f : -0.125f);
                        const float dy = sample == 1 ? 0.125f : (sample == 2 ? -0.125f : -0.375f);
                        pathStart(st, v3(cam.ox, cam.oy, cam.oz), primaryDir(cam, fr.width, fr.height, job.row, job.col, dx, dy), fr.depth);
                        continue;
                    }
                    c = sum * 0.25f;
                }
            }
            if (quad_j <= 0) bgr[0] = toByte(c.z), bgr[1] = toByte(c.y), bgr[2] = toByte(c.x);
            active = false;
        }
    }
}

// offset of sample k of a pixel from its centre (RTX_OPT_SPP = 4, include/rtx.h)
RTX_DEV float sampleDx(int k) { return k == 0 ? 0.125f : (k == 1 ? -0.375f : (k == 2 ? 0.375f : -0.125f)); }
RTX_DEV float sampleDy(int k) { return k == 0 ? 0.375f : (k == 1 ? 0.125f : (k == 2 ? -0.125f : -0.375f)); }

// Several samples per pixel, one lane per pixel: the 64 x 4 paths of a tile form a pool.  Path p is
// sample p / 64 of pixel p % 64; every lane starts with sample 0 of its own pixel and takes the next
// path of the pool whenever it finishes one, so the lanes stay busy until the pool is empty instead of
// each waiting for the slowest lane's four samples.  A path's colour does not depend on the lane that
// traces it; the colours are parked in `samples` ([sample][pixel][rgb], LDS) and summed per pixel in
// sample order afterwards.
template <bool STATS>
__device__ __forceinline__ void tracePathPool(const DevScene &s, const SmallScene &q, const DevCamera &cam, const DevFrame &fr, int x0, int y0,
                                              PathState &st, FrameStack &frames, float *samples, unsigned char *tile_bgr, int *stack,
                                              LaneCounters<STATS> &lc) {
    const unsigned lane = threadIdx.x & 63u;
    const V3 eye = v3(cam.ox, cam.oy, cam.oz);
    unsigned next = 0;  // wave-uniform: first path not handed out yet
    int path = -1;      // the path this lane is tracing
    bool active = false;
    for (;;) {
        // ---- hand paths to the idle lanes (a path of a pixel outside the frame is dropped at once) ----
        while (next < 256u) {
            const unsigned long long idle = __ballot(!active);
            if (!idle) break;
            if (!active) {
                const unsigned mine = next + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
                if (mine < 256u) {
                    const int pixel = (int)(mine & 63u), k = (int)(mine >> 6);
                    const int px = x0 + (pixel & 7), pk = y0 + (pixel >> 3);
                    if (px < fr.width && pk < fr.rows) {
                        const int row = fr.row_first + pk * fr.row_step;
                        pathStart(st, eye, primaryDir(cam, fr.width, fr.height, row, px, sampleDx(k), sampleDy(k)), fr.depth);
                        path = (int)mine;
                        active = true;
                    }
                }
            }
            next += (unsigned)__popcll(idle);
        }
        if (!__ballot(active)) break;
        if (active) {
            bool hit = false;
            Hit h;
            float md = 0.0f;
            if (pathCasts(st, -1)) {
                pathCountRay<STATS>(st, -1, lc);
                hit = sceneIntersect<STATS, true>(s, q, st.orig, st.dir, st.light >= 0, st.t_exit, h, md, stack, kBlock, lc);
            }
            V3 c;
            if (pathConsume<STATS, true>(s, q, st, frames, hit, h, md, -1, c, lc)) {
                float *o = samples + 3 * path;  // [sample][pixel][rgb] = 3 * (64 * sample + pixel)
                o[0] = c.x, o[1] = c.y, o[2] = c.z;
                active = false;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (((c0 + c1) + c2) + c3) * 0.25 for this lane's own pixel
    const float *a = samples + 3 * lane;
    V3 sum = v3(a[0], a[1], a[2]);
#pragma unroll
    for (int k = 1; k < 4; ++k) sum = sum + v3(a[192 * k], a[192 * k + 1], a[192 * k + 2]);
    sum = sum * 0.25f;
    unsigned char *bgr = tile_bgr + 3 * lane;
    bgr[0] = toByte(sum.z), bgr[1] = toByte(sum.y), bgr[2] = toByte(sum.x);
}

// Device memory behind a pointer that has been through LDS (k_render's per-tile values): the compiler no longer knows
// the address space and would use flat instructions, whose stores a wave has to wait for at its next LDS fence.
using GlobalU32 = __attribute__((address_space(1))) unsigned;
using GlobalU8 = __attribute__((address_space(1))) unsigned char;
template <typename G, typename T> __device__ __forceinline__ G *inDeviceMemory(T *p) { return (G *)p; }

// k_render's per-lane LDS (sized by the launch): the traversal stack, then the frame heads; dword k of lane t at [k * kBlock + t]
extern __shared__ int s_lane_lds[];

// LOOK = the launch for the quad-mode tiles, beside the frame's main launch (on a stream of its own): its waves take
// the quarters at the head of the list (the first list_hdr[2] entries, through queue heads of their own) and trace them
// with the look-ahead ray of pathConsume, then join the main launch's queues.  The look-ahead keeps a second hit record
// alive through the cast; in a loop that shares its registers with the one-lane-per-pixel loop the latter pays for
// that, so the main launch (LOOK = false, `split`) is compiled without it and starts behind the quarters.  Together
// the two launches have as many workgroups as fit the GPU at once, so neither waits for the other whichever starts first.
template <bool STATS, bool WIDE, bool LOOK>
__global__ __launch_bounds__(kBlock, STATS || WIDE ? 2 : 3) void k_render(DevScene s, DevCamera cam_arg, DevFrame fr_arg, unsigned tiles_x_arg, unsigned n_tiles,
                                                   const unsigned *order_arg, const unsigned *__restrict__ n_entries_dev,
                                                   unsigned batch_from_arg, unsigned *heads_arg, unsigned *heads_next,
                                                   unsigned *tile_cost_arg, unsigned char *out_arg,
                                                   Counters *counters, unsigned split, unsigned *quad_heads, unsigned *quad_heads_next) {
    __shared__ SmallScene q;
    __shared__ unsigned char s_tile[kBlock / 64][192];  // the B,G,R bytes of a wave's tile: 8 rows of 24 bytes
    __shared__ float s_samples[WIDE ? kBlock / 64 : 1][WIDE ? 4 * 64 * 3 : 1];  // tracePathPool: colours per sample and pixel
    // What only the code between two tiles looks at (camera, frame description, list and output pointers) is read from
    // LDS there: as kernel arguments these forty-odd dwords would sit in scalar registers through the whole path loop,
    // which has too few of those (the overflow is parked in vector-register lanes and fetched back with a vector
    // instruction per use).
    struct TileArgs {
        DevCamera cam;
        DevFrame fr;
        unsigned tiles_x;
        const unsigned *order;
        unsigned *heads, *tile_cost;
        unsigned char *out;
    };
    __shared__ TileArgs ta;
    // `split`: the quarters at the head of the list are the quad-mode launch's, the main queues start behind them
    const unsigned skip = split && n_entries_dev ? n_entries_dev[2] : 0u;
    const unsigned *const order_behind = order_arg ? order_arg + skip : nullptr;
    if (threadIdx.x == 0) {
        ta.cam = cam_arg, ta.fr = fr_arg, ta.tiles_x = tiles_x_arg, ta.order = order_behind, ta.heads = heads_arg, ta.tile_cost = tile_cost_arg;
        ta.out = out_arg;
    }
    loadSmallScene(q, s);
    __syncthreads();
    // (not in the general build: it is short of vector registers rather than scalar ones, and spills more this way)
    const DevCamera &cam = *(WIDE ? &cam_arg : &ta.cam);
    const DevFrame &fr = *(WIDE ? &fr_arg : &ta.fr);
    const unsigned &tiles_x = *(WIDE ? &tiles_x_arg : &ta.tiles_x);
    const unsigned *const &order = *(WIDE ? &order_behind : &ta.order);
    unsigned *const &heads = *(WIDE ? &heads_arg : &ta.heads);
    unsigned *const &tile_cost = *(WIDE ? &tile_cost_arg : &ta.tile_cost);
    unsigned char *const &out = *(WIDE ? &out_arg : &ta.out);
    int *stack = s_lane_lds + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    // the queue heads come in two sets: this launch counts in one and clears the other for the next
    // launch of the context (launches of a context follow one another), which saves a fill per frame
    if (blockIdx.x == 0 && threadIdx.x < kQueues) (LOOK ? quad_heads_next : heads_next)[threadIdx.x * kHeadStride] = 0u;
    // rows of whole dwords: a finished tile goes out as 48 dword stores instead of 192 byte stores
    const bool dword_rows = (fr.width & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0;
    static_assert(kQueues == 64, "one queue head per lane in the scan below");
    // the list has one entry per tile, or four for a tile that is traced in quad mode (see kQuadFlag)
    const unsigned n_entries = (n_entries_dev ? n_entries_dev[0] : n_tiles) - skip;
    // entries from here on are cheap tiles (below the average cost of the previous frame, or outside every
    // projected bounding box on the first frame of a view)
    const unsigned batch_from_all = n_entries_dev ? n_entries_dev[1] : batch_from_arg;
    const unsigned batch_from = batch_from_all > skip ? batch_from_all - skip : 0u;
    const unsigned my_len = lane < n_entries ? (n_entries - lane + kQueues - 1u) / kQueues : 0u;  // entries of queue `lane`
    const unsigned my_quarters = LOOK && lane < skip ? (skip - lane + kQueues - 1u) / kQueues : 0u;  // ... of the quarters' queue `lane`
    bool quarters_left = LOOK;  // wave-uniform: the quarters' queues have not been seen empty yet
    const unsigned first_queue = (blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6)) % kQueues;
    unsigned qi = first_queue;  // wave-uniform: current queue
    LaneCounters<STATS> lc;
    PathState st;
    FrameSpill spill;  // the heads of the first frames in LDS, what is indexed dynamically beyond them in scratch memory
    FrameStack frames = frameStack(spill, reinterpret_cast<float *>(s_lane_lds + fr_arg.stack_entries * kBlock + threadIdx.x), fr_arg.lds_frames, kBlock);
    const V3 eye = v3(cam.ox, cam.oy, cam.oz);
    // Taking a tile is two dependent round trips to memory (the head counter, then the list entry), a
    // few microseconds during which the wave does nothing -- as long as a cheap tile takes to trace.  The
    // tiles of the expensive front part of the list are taken one by one, the cheap ones behind it in
    // batches: one atomic, the entries loaded side by side, one per lane.  A batch is at most kBatch
    // tiles and at most a quarter of a wave's even share of the list, so that small frames (or small
    // shares of one) keep their balance.
    constexpr unsigned kBatch = 8;
    const unsigned even_share = n_entries / (gridDim.x * (kBlock / 64u));
    const unsigned batch_max = even_share / 4u < 1u ? 1u : (even_share / 4u > kBatch ? kBatch : even_share / 4u);
    bool cheap_zone = batch_from == 0u;  // wave-uniform: the queue position last seen lies in the cheap part
    unsigned batch = 0xffffffffu;        // lane k: entry k of the batch in hand
    unsigned batch_n = 0, batch_k = 0;   // wave-uniform: its size, the next one to trace
    for (;;) {
        // ---- take a tile ----
        unsigned tile = 0xffffffffu;
        if constexpr (LOOK) {
            // the quarters first, one at a time
            while (quarters_left) {
                const unsigned len = __builtin_amdgcn_readlane(my_quarters, qi);
                unsigned i = len;
                if (lane == 0 && len)
                    i = __hip_atomic_fetch_add(inDeviceMemory<GlobalU32>(quad_heads + qi * kHeadStride), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                i = __builtin_amdgcn_readfirstlane(i);
                if (i < len) {
                    unsigned e = 0;
                    if (lane == 0) e = *inDeviceMemory<const GlobalU32>(order_arg + qi + i * kQueues);
                    tile = __builtin_amdgcn_readfirstlane(e);
                    break;
                }
                const unsigned head = __hip_atomic_load(inDeviceMemory<GlobalU32>(quad_heads + lane * kHeadStride), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long live = __ballot(head < my_quarters);
                live &= ~(1ull << qi);
                if (!live) {
                    quarters_left = false, qi = first_queue;  // on with the main launch's queues
                    break;
                }
                const unsigned long long after = qi == 63u ? 0ull : live & (~0ull << (qi + 1u));
                qi = (unsigned)__builtin_ctzll(after ? after : live);
            }
        }
        if (tile != 0xffffffffu) {
        } else if (batch_k < batch_n) tile = __builtin_amdgcn_readlane(batch, batch_k++);
        else for (;;) {
            const unsigned len = __builtin_amdgcn_readlane(my_len, qi);
            const unsigned want = cheap_zone ? batch_max : 1u;
            unsigned i = len;
            if (lane == 0 && len)
                i = __hip_atomic_fetch_add(inDeviceMemory<GlobalU32>(heads + qi * kHeadStride), want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            i = __builtin_amdgcn_readfirstlane(i);
            if (i < len) {
                const unsigned n = want < len - i ? want : len - i;
                const unsigned e = qi + (i + lane) * kQueues;
                batch = 0xffffffffu;
                if (lane < n) batch = order ? *inDeviceMemory<const GlobalU32>(order + e) : e;
                batch_n = n, batch_k = 1u;
                cheap_zone = qi + (i + n) * kQueues >= batch_from;
                tile = __builtin_amdgcn_readlane(batch, 0);
                break;
            }
            // this queue is empty: every lane looks at one head, the wave moves to the next queue
            // (cyclically after qi) that still has entries, or leaves when there is none
            const unsigned head = __hip_atomic_load(inDeviceMemory<GlobalU32>(heads + lane * kHeadStride), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long live = __ballot(head < my_len);
            live &= ~(1ull << qi);
            if (!live) break;
            const unsigned long long after = qi == 63u ? 0ull : live & (~0ull << (qi + 1u));
            qi = (unsigned)__builtin_ctzll(after ? after : live);
        }
        if (tile == 0xffffffffu) break;  // every queue is empty
        // An entry with kQuadFlag is a quarter (4x4 pixels, bits 28-29 say which) of a tile that lasted
        // long in the previous frame: four lanes share a pixel (quad mode of pathConsume).
        const bool quad = (tile & kQuadFlag) != 0u;
        const unsigned sub = (tile >> 28) & 3u;
        tile &= kTileMask;
        const int quad_j = quad ? (int)(lane & 3u) : -1;
        const unsigned pix = quad ? lane >> 2 : lane;  // pixel of the (sub-)tile this lane works on
        const int side = quad ? 4 : 8;
        const unsigned side_shift = quad ? 2u : 3u;
        const int x0 = (int)(tile % tiles_x) * 8 + (quad ? (int)(sub & 1u) * 4 : 0);
        const int y0 = (int)(tile / tiles_x) * 8 + (quad ? (int)(sub >> 1) * 4 : 0);
        bool active;
        PixelJob job{0, 0};
        {
            const int px = x0 + (int)(pix & (unsigned)(side - 1)), pk = y0 + (int)(pix >> side_shift);
            active = px < fr.width && pk < fr.rows;  // pixels of an edge tile outside the frame are skipped
            if (active) {
                const int row = fr.row_first + pk * fr.row_step;
                job.row = row, job.col = px;
                if (WIDE && fr.spp > 1) pathStart(st, eye, primaryDir(cam, fr.width, fr.height, row, px, 0.125f, 0.375f), fr.depth);
                else pathStart(st, eye, primaryDir(cam, fr.width, fr.height, row, px), fr.depth);
            }
        }
        // ---- trace its paths ----
        const long long t_tile = wall_clock64();
        // two copies of the loop, one per mode: in the common one quad_j is the constant -1 and the quad
        // logic folds away
        if (quad) tracePaths<STATS, WIDE, LOOK>(s, q, cam, fr, job, st, frames, active, quad_j, &s_tile[threadIdx.x >> 6][3 * pix], stack, lc);
        else if (WIDE && fr.spp > 1) tracePathPool<STATS>(s, q, cam, fr, x0, y0, st, frames, s_samples[WIDE ? threadIdx.x >> 6 : 0], s_tile[threadIdx.x >> 6], stack, lc);
        else tracePaths<STATS, WIDE>(s, q, cam, fr, job, st, frames, active, -1, &s_tile[threadIdx.x >> 6][3 * pix], stack, lc);
        // what the tile cost (100 MHz ticks): the next frame of the same view starts the expensive ones first
        // (a quarter in quad mode runs about half as long as the whole tile would: the first quarter stands for it)
        // (with the look-ahead about a third as long)
        if (lane == 0 && (!quad || sub == 0u)) *inDeviceMemory<GlobalU32>(tile_cost + tile) = (unsigned)(wall_clock64() - t_tile) * (quad ? (LOOK ? kQuadCostScale : 2u) : 1u);
        // ---- write the tile: all 64 lanes are here.  LDS operations of a wave complete in order; the
        // fences keep the compiler from moving the reads above the byte writes. ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (dword_rows && x0 + side <= fr.width && y0 + side <= fr.rows) {
            // rows of 24 (12) bytes: 48 (12) dword stores instead of 192 (48) byte stores
            const unsigned row_dwords = quad ? 3u : 6u;
            if (lane < row_dwords * (unsigned)side) {
                const unsigned r = lane / row_dwords, w = lane % row_dwords;
                GlobalU32 *dst = inDeviceMemory<GlobalU32>(reinterpret_cast<unsigned *>(out + ((size_t)(y0 + (int)r) * fr.width + x0) * 3) + w);
                *dst = *reinterpret_cast<const unsigned *>(&s_tile[threadIdx.x >> 6][4 * lane]);
            }
        } else if (quad_j <= 0) {
            const int px = x0 + (int)(pix & (unsigned)(side - 1)), pk = y0 + (int)(pix >> side_shift);
            if (px < fr.width && pk < fr.rows) {
                const unsigned char *src = &s_tile[threadIdx.x >> 6][3 * pix];
                GlobalU8 *p = inDeviceMemory<GlobalU8>(out + ((size_t)pk * fr.width + px) * 3);
                p[0] = src[0], p[1] = src[1], p[2] = src[2];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    flushCounters<STATS>(lc, counters);
}

#include "rtx_wavefront.hpp"

template <bool WIDE>
__global__ __launch_bounds__(kBlock) void k_trace_rays(DevScene s, const float *__restrict__ orig,
                                                       const float *__restrict__ dir, int n, int depth,
                                                       float *__restrict__ rgb) {
    __shared__ int s_stack[kStackDepth * kBlock];
    __shared__ SmallScene q;
    loadSmallScene(q, s);
    __syncthreads();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    LaneCounters<false> lc;
    const V3 c = traceRay<false, WIDE>(s, q, v3(orig[3 * i], orig[3 * i + 1], orig[3 * i + 2]),
                                 v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), depth, s_stack + threadIdx.x, kBlock, lc);
    rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
}

template <bool WIDE>
__global__ __launch_bounds__(kBlock) void k_intersect_rays(DevScene s, const float *__restrict__ orig,
                                                           const float *__restrict__ dir, int n,
                                                           float *__restrict__ out16) {
    __shared__ int s_stack[kStackDepth * kBlock];
    __shared__ SmallScene q;
    loadSmallScene(q, s);
    __syncthreads();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    LaneCounters<false> lc;
    Hit h;
    h.point = v3(0, 0, 0), h.N = v3(0, 0, 0);
    h.m = Material{1.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // Material(), include/Material.h:14
    float md;
    const V3 from = v3(orig[3 * i], orig[3 * i + 1], orig[3 * i + 2]);
    const bool ok = sceneIntersect<false, WIDE>(s, q, from, v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), false, -1.0f, h, md,
                                          s_stack + threadIdx.x, kBlock, lc);
    float *o = out16 + 16 * (size_t)i;
    o[0] = ok ? 1.0f : 0.0f;
    o[1] = h.point.x, o[2] = h.point.y, o[3] = h.point.z;
    o[4] = h.N.x, o[5] = h.N.y, o[6] = h.N.z;
    o[7] = h.m.refractive, o[8] = h.m.a0, o[9] = h.m.a1, o[10] = h.m.a2, o[11] = h.m.a3;
    o[12] = h.m.dr, o[13] = h.m.dg, o[14] = h.m.db, o[15] = h.m.spec;
}

// The first frame of a view, better than by boxes: one ray per tile.  Lane i casts the primary ray of the pixel in the
// middle of tile i through the real sceneIntersect and puts the tile into a class by what it hits and by where the
// children of that hit would go: the mesh seen in itself (the longest chains of the frame: a walk at every bounce),
// the mesh, glass (two children per node), a mirroring surface whose child passes the mesh's box, any other hit, a
// miss.  One cast per lane, no shadow rays: some ten microseconds for the 129 600 tiles of a 3840x2160 frame.
// (Tracing the whole path of the tile's middle pixel gives a better order still -- a repeated view without schedule
// feedback 0.81 ms against 0.95 with the projected boxes -- but the probe kernel then lasts as long as the longest
// chain of casts, 0.3 ms, and the first frame has to wait for it.)  k_tile_spread then lets a tile inherit half of its
// dearest neighbour's class cost (the middle pixel of a tile on an object's outline may see the sky).  The classes
// are stand-in costs and go through the same k_bin_* launches as measured costs do.
constexpr unsigned kProbeSelfMesh = 16384u, kProbeMesh = 4096u, kProbeGlass = 2048u, kProbeMirrorsMesh = 1024u, kProbeHit = 64u, kProbeMiss = 1u;
constexpr unsigned kProbeCheapCost = 512u;  // plain hits and misses count as cheap (taken in batches)
template <bool WIDE>
__global__ __launch_bounds__(kBlock) void k_tile_probe(DevScene s, DevCamera cam, DevFrame fr, unsigned tiles_x, unsigned n_tiles,
                                                       unsigned *__restrict__ cost) {
    __shared__ int s_stack[kStackDepth * kBlock];
    __shared__ SmallScene q;
    loadSmallScene(q, s);
    __syncthreads();
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_tiles) return;
    const int bx = (int)(i % tiles_x), by = (int)(i / tiles_x);
    const int px = min(bx * 8 + 4, fr.width - 1), pk = min(by * 8 + 4, fr.rows - 1);
    const V3 orig = v3(cam.ox, cam.oy, cam.oz), dir = primaryDir(cam, fr.width, fr.height, fr.row_first + pk * fr.row_step, px);
    LaneCounters<false> lc;
    CastState cs;
    int *stack = s_stack + threadIdx.x;
    if (castBegin<false, WIDE>(s, q, orig, dir, -1.0f, cs, stack, kBlock, lc)) castMesh<false>(s, orig, dir, -1.0f, cs, stack, kBlock, lc);
    Hit h;
    float md = 0.0f;
    const bool hit = fr.depth > 0 && castEnd<false, WIDE>(s, q, orig, dir, false, -1.0f, cs, h, md, lc);
    unsigned c = kProbeMiss;
    if (hit) {
        const bool on_mesh = cs.kind == 4 && md == cs.minDist;  // (the planes of castEnd did not come first)
        const bool refl = h.m.a2 != 0.0f, refr = h.m.a3 != 0.0f;
        bool child_at_mesh = false;  // does a child ray pass the mesh's box?
        if (s.n_mesh_tris > 0 && fr.depth > 1) {
            if (refl) {
                const V3 d = reflectDir(dir, h.N);
                const V3 o = h.point + h.N * (__builtin_signbit(dot(d, h.N)) ? -s.fprec : s.fprec);
                child_at_mesh = child_at_mesh || slabMaybe(s.mesh_lo, s.mesh_hi, o, safeInverse(d));
            }
            if (refr) {
                const V3 d = normalize(refractDir(dir, h.N, h.m.refractive));
                const V3 o = h.point + h.N * (__builtin_signbit(dot(d, h.N)) ? -s.fprec : s.fprec);
                child_at_mesh = child_at_mesh || slabMaybe(s.mesh_lo, s.mesh_hi, o, safeInverse(d));
            }
        }
        c = on_mesh ? (child_at_mesh ? kProbeSelfMesh : kProbeMesh)
                    : (refl && refr ? kProbeGlass + (child_at_mesh ? kProbeMirrorsMesh : 0u) : (child_at_mesh ? kProbeMirrorsMesh : kProbeHit));
    }
    cost[i] = c;
}
// ... and, like k_tile_classes, puts the tiles near the middle of an object's projected rectangle first within their class
// (up to twice the class's cost): an object fills the middle of its box, and the longest chains -- the mesh mirrored in
// itself -- are found there.
__global__ __launch_bounds__(256) void k_tile_spread(const unsigned *__restrict__ probed, TileRects rects, unsigned n, unsigned tiles_x, int row_first,
                                                     int row_step, unsigned *__restrict__ cost) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int bx = (int)(i % tiles_x), by = (int)(i / tiles_x), tiles_y = (int)((n + tiles_x - 1) / tiles_x);
    unsigned around = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int x = bx + dx, y = by + dy;
            if (x < 0 || y < 0 || x >= (int)tiles_x || y >= tiles_y || (unsigned)y * tiles_x + (unsigned)x >= n) continue;
            around = max(around, probed[(unsigned)y * tiles_x + (unsigned)x]);
        }
    const unsigned base = max(probed[i], around / 2u);
    const float c0 = (float)(bx * 8), c1 = (float)(bx * 8 + 7);
    const float r0 = (float)(row_first + (long long)by * 8 * row_step), r1 = (float)(row_first + ((long long)by * 8 + 7) * row_step);
    float nearness = 0.0f;  // 1 in the middle of a rectangle, 0 at its edge and outside
    unsigned by_box = 0u;   // the class the projected boxes alone would give (the middle pixel may see a fence post in front of the mesh)
    for (int k = 0; k < rects.n; ++k)
        if (c1 >= rects.r[k][0] - 1 && c0 <= rects.r[k][1] + 1 && r1 >= rects.r[k][2] - 1 && r0 <= rects.r[k][3] + 1) {
            const float half_w = 0.5f * (rects.r[k][1] - rects.r[k][0]) + 4.0f, half_h = 0.5f * (rects.r[k][3] - rects.r[k][2]) + 4.0f * (float)row_step;
            const float dx = fabsf(0.5f * (c0 + c1) - 0.5f * (rects.r[k][0] + rects.r[k][1])) / half_w;
            const float dy = fabsf(0.5f * (r0 + r1) - 0.5f * (rects.r[k][2] + rects.r[k][3])) / half_h;
            nearness = fmaxf(nearness, 1.0f - fminf(fmaxf(dx, dy), 1.0f));
            by_box = max(by_box, rects.cls[k] == 0 ? kProbeMesh : kProbeMirrorsMesh);
        }
    const unsigned cls = max(base, by_box);
    cost[i] = cls + (cls >= kProbeMirrorsMesh ? (unsigned)((float)(cls - 1u) * nearness) : 0u);
}

// gathered: n_ranks blocks of rows_per_rank rows; block r row k is image row r + k * n_ranks
__global__ void k_deinterleave(const unsigned char *__restrict__ gathered, int n_ranks, int rows_per_rank,
                               int row_bytes, int height, unsigned char *__restrict__ frame) {
    const size_t total = (size_t)height * row_bytes;
    if ((row_bytes & 3) == 0) {
        const size_t n4 = total >> 2;
        const int row4 = row_bytes >> 2;
        for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n4; t += (size_t)gridDim.x * blockDim.x) {
            const int i = (int)(t / row4), c = (int)(t % row4);
            const int r = i % n_ranks, k = i / n_ranks;
            reinterpret_cast<unsigned *>(frame)[t] =
                reinterpret_cast<const unsigned *>(gathered)[((size_t)r * rows_per_rank + k) * row4 + c];
        }
    } else {
        for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
            const int i = (int)(t / row_bytes), c = (int)(t % row_bytes);
            const int r = i % n_ranks, k = i / n_ranks;
            frame[t] = gathered[((size_t)r * rows_per_rank + k) * row_bytes + c];
        }
    }
}

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int failHip(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return RTX_E_DEVICE;
}

#define HIP_OR_FAIL(expr)                                  \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return failHip(e_, #expr);   \
    } while (0)

template <typename T>
int upload(const std::vector<T> &v, T **dptr, std::vector<void *> &owned) {
    *dptr = nullptr;
    if (v.empty()) return RTX_OK;
    void *p = nullptr;
    HIP_OR_FAIL(hipMalloc(&p, v.size() * sizeof(T)));
    owned.push_back(p);
    HIP_OR_FAIL(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dptr = static_cast<T *>(p);
    return RTX_OK;
}

}  // namespace

// Device memory with which a render schedules its tiles.  A context owns one set (hipMalloc, grown to the
// largest frame seen); the allocation-free entry point carves one out of a static device array.
struct SchedMem {
    unsigned *heads = nullptr;     // work-queue heads of k_render, two sets: a launch counts in one and clears the other
    unsigned *cost = nullptr;      // per tile: what it cost in the last frame (100 MHz ticks), or its class's stand-in cost
    unsigned *entries = nullptr;   // the tile list built from it: `quarters` x cap entries (the longest tiles as four quarters)
    unsigned *list_hdr = nullptr;  // {entries in the list, first cheap entry, quarters at the head of the list}
    unsigned *bins = nullptr;      // k_bin_*: per-workgroup histograms, then two words of header
    unsigned long long *bin_sums = nullptr;
    size_t cap = 0;                // tiles that cost and entries are sized for
    bool quarters = true;          // entries has room for every tile as four quarters (quad mode)
};
constexpr size_t kHeadWords = 2 * kQueues * kHeadStride, kBinWords = kBinBlocks * kCostBins + 2;

// A view = camera + frame description, compared bit for bit.
struct ViewKey {
    bool valid = false;
    rtx_camera cam{};
    rtx_frame frame{};
    bool is(const rtx_camera *c, const rtx_frame *f) const {
        return valid && memcmp(&cam, c, sizeof cam) == 0 && memcmp(&frame, f, sizeof frame) == 0;
    }
    void set(const rtx_camera *c, const rtx_frame *f) { valid = true, cam = *c, frame = *f; }
};
// What a SchedMem holds at the moment.
struct SchedState {
    bool heads_flip = false, quad_flip = false;  // which set of queue heads the next main / quad-mode launch counts in
    bool have_cost = false;  // `cost` holds the measured costs of a frame: of cost_view
    ViewKey cost_view;
    ViewKey by_cost_key;     // `entries` was built from measured costs for this view
    ViewKey static_key;      // `entries` was built from the tile classes for this view
    bool list_quads = false; // the list in `entries` may start with quarters (quad-mode tiles)
};

// ---------------------------------------------------------------------------------------------
// Scheduling memory for callers that must not allocate (rtx_render_device_noalloc): one static device
// array per GPU instead of the context's hipMalloc'ed buffers.  The array's name is only a host-side
// handle; its address on a device comes from hipGetSymbolAddress (scratchOnDevice, called by rtx_create
// with that device current) and is what every kernel argument and memset below uses.
// Layout (byte offsets, each a multiple of 256): queue heads, histogram bins, bin sums, list header,
// cost per tile, tile list.  No quarters: a list made from tile classes never uses quad mode.
// ---------------------------------------------------------------------------------------------
constexpr size_t kScratchTiles = 1u << 19;  // 524 288 tiles of 8x8 pixels, e.g. a 7680x4320 frame (518 400)
constexpr size_t align256(size_t n) { return (n + 255) / 256 * 256; }
constexpr size_t kScratchHeads = 0;
constexpr size_t kScratchBins = kScratchHeads + align256(kHeadWords * sizeof(unsigned));
constexpr size_t kScratchBinSums = kScratchBins + align256(kBinWords * sizeof(unsigned));
constexpr size_t kScratchListHdr = kScratchBinSums + align256(kBinBlocks * sizeof(unsigned long long));
constexpr size_t kScratchCost = kScratchListHdr + 256;
constexpr size_t kScratchEntries = kScratchCost + kScratchTiles * sizeof(unsigned);
constexpr size_t kScratchBytes = kScratchEntries + kScratchTiles * sizeof(unsigned);  // 4.08 MiB
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU, once looked up

// Looks up this GPU's copy of `scratch`; the device must be current.
static int scratchOnDevice(int device) {
    if (device < 0 || device >= kMaxDevices) return RTX_OK;  // such a device simply has no allocation-free path
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return RTX_OK;
    void *p = nullptr;
    HIP_OR_FAIL(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = p;
    return RTX_OK;
}

// Sub-list capacities of the pipeline's lists for strips of `strip` samples at recursion depth D, for the worst
// case of every call hitting and spawning both children (see WfList).  Returns the bytes of all arrays.
struct WfPlan {
    unsigned node_cap[kWfLevels + 1] = {}, shadow_cap[kWfLevels + 1] = {}, req_cap[kWfLevels + 1] = {};  // per sub-list
    size_t bytes = 0;
    bool fits = true;  // node indices fit the shadow queue's 27 bits, request indices the links' 28
};
// Device memory of the level-by-level pipeline (rtx_wavefront.hpp): one block, carved into the per-level
// request and record arrays for strips of `strip_samples` samples at recursion depth `depth`.
struct WfMem {
    void *block = nullptr;
    size_t block_bytes = 0;
    size_t strip_samples = 0;
    int depth = -1, n_lights = 0;
    WfPlan plan;
    WfCounters *counters = nullptr;
    float4 *ray_o[kWfLevels + 1] = {}, *ray_d[kWfLevels + 1] = {}, *thin[kWfLevels + 1] = {};
    float4 *pt = nullptr, *nrm = nullptr, *vw = nullptr, *dif = nullptr, *samples = nullptr;
    unsigned *hidden = nullptr;
    unsigned *shadow_q = nullptr;
};

struct rtx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // host staging (until commit)
    std::vector<Material> materials;
    std::vector<float4> lights;
    std::vector<float4> spheres;
    std::vector<int> sphere_mat;
    std::vector<float> tris;
    std::vector<int> tri_mat;
    std::vector<float> boxes;
    std::vector<int> box_mat;
    std::vector<float> mesh_verts;
    std::vector<int> mesh_idx;
    int mesh_mat = 0;
    std::vector<float4> islands;
    float island_y = -4.0f;
    int island_mat = 0;
    float light_green[3] = {0, 0, 0}, dark_green[3] = {0, 0, 0};
    std::vector<unsigned char> water;
    int water_w = 0, water_h = 0, water_mat = 0;
    float water_y = -4.5f, water_scale = 10.0f;
    std::vector<unsigned char> sky;
    int sky_w = 0, sky_h = 0, sky_shift = 100;
    // options
    bool stats = false;
    int prune = 1;
    int mesh_grid = 1;
    int bvh_builder = 0;  // RTX_OPT_BVH_BUILDER
    float far_clip = 700.0f, fprec = 0.001f;
    // device state
    bool committed = false;
    DevScene dev{};
    std::vector<void *> owned;
    int bvh_depth = 0;
    Counters *d_counters = nullptr;
    int resident_blocks[4] = {0, 0, 0, 0};  // grid size of k_render<STATS, WIDE>, index STATS + 2 * WIDE
    int spp = 1;                            // RTX_OPT_SPP
    bool big_spheres = false;               // more spheres than the LDS staging holds: they sit behind a BVH
    float bounded_objects[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};  // box around loose triangles, boxes, mesh and islands
    bool long_lists = false;                // another list is longer than its LDS staging: the general build of the kernels reads its tail from global memory
    int feedback = 1;                       // RTX_OPT_SCHEDULE_FEEDBACK
    int quad_mode = 1;                      // RTX_OPT_QUAD
    int tile_probe = 0;                     // RTX_OPT_TILE_PROBE
    int quad_kernel = 1;                    // RTX_OPT_QUAD_KERNEL
    hipStream_t quad_stream = nullptr;      // the quad-mode launch runs beside the main one: forked off the render's stream and joined again
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int resident_quad[4] = {0, 0, 0, 0};    // most workgroups of the quad-mode launch k_render<STATS, WIDE, true> that are resident at once
    unsigned *host_quads = nullptr;         // page-locked: quad-mode tiles of the list k_bin_scan built last
    SchedMem sched;                         // the context's own scheduling memory (grown on demand)
    SchedState sched_state;
    int pipeline = 2;                       // RTX_OPT_PIPELINE
    WfMem wf;
    int wf_stack_entries = kStackDepth;     // LDS traversal stack entries per lane of the pipeline's kernels
    int stack_entries = kStackDepth;        // ... and of k_render
    int lds_frame_slots[4] = {0, 0, 0, 0};  // frame heads of a lane that k_render keeps in LDS, per variant
    int sphere_bvh_depth = 0;
    int wf_blocks[3][4] = {};               // resident blocks of k_wf_trace / k_wf_shadow [STATS + 2 * WIDE], k_wf_shade [STATS]
    size_t wf_budget = (size_t)64 << 30;    // bytes the pipeline's arrays may take
    bool wf_pool = true;                    // RTX_OPT_RAY_POOL
    unsigned char *d_frame = nullptr;
    size_t d_frame_bytes = 0;
    HostStage stage;                        // rtx_render's way to the caller's host buffer (RTX_OPT_PIN_OUTPUT)
};

namespace {

int bind(rtx_ctx *c) {
    HIP_OR_FAIL(hipSetDevice(c->device));
    return RTX_OK;
}

// BoundingBox::init + initTriangles, src/SceneObjects.cpp:176-209: the 7x7x7 cells of the padded
// vertex bounds and, per triangle, the cells its vertices fall into.  step = edge lengths of a cell
// (the device forms the cell boxes from it the way the reference does: lo + step * i); keys = four
// ints per triangle: (cell << 22 | position in that cell's list) for each distinct cell, cell =
// (i*7 + j)*7 + k, -1 for unused slots.  Arithmetic as the reference writes it (binary32, no
// contraction), so that a vertex on a cell boundary lands in the same cell.
void gridCells(const std::vector<float> &verts, const std::vector<int> &idx, const float lo[3], const float hi[3],
               float step[3], std::vector<int> &keys) {
    const int size = 7;
    const float lx = hi[0] - lo[0], ly = hi[1] - lo[1], lz = hi[2] - lo[2];
    step[0] = lx / float(size), step[1] = ly / float(size), step[2] = lz / float(size);
    const size_t nt = idx.size() / 3;
    keys.assign(nt * 4, -1);
    std::vector<int> fill((size_t)size * size * size, 0);
    for (size_t ti = 0; ti < nt; ++ti) {
        int used = 0;
        for (int v = 0; v < 3; ++v) {
            const float *p = &verts[3 * (size_t)idx[3 * ti + v]];
            const float fx = (p[0] - lo[0]) * size / lx, fy = (p[1] - lo[1]) * size / ly, fz = (p[2] - lo[2]) * size / lz;
            const int i = int(fx), j = int(fy), k = int(fz);
            if (i < 0 || j < 0 || k < 0 || i >= size || j >= size || k >= size) continue;  // outside: in no cell
            const int cell = (i * size + j) * size + k;
            bool seen = false;
            for (int u = 0; u < used; ++u) seen |= (keys[4 * ti + u] >> 22) == cell;
            if (seen) continue;
            const int pos = fill[cell]++;
            keys[4 * ti + used++] = (cell << 22) | (pos < (1 << 22) ? pos : (1 << 22) - 1);
        }
    }
}

int checkMat(const rtx_ctx *c, const int *mat, int n) {
    for (int i = 0; i < n; ++i)
        if (mat[i] < 0 || (size_t)mat[i] >= c->materials.size()) return fail(RTX_E_INVALID, "material index out of range");
    return RTX_OK;
}

int checkFrame(const rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f) {
    if (!c || !cam || !f) return fail(RTX_E_INVALID, "null argument");
    if (!c->committed) return fail(RTX_E_STATE, "render before rtx_commit");
    if (f->width <= 0 || f->height <= 0 || f->row_step <= 0 || f->row_first < 0)
        return fail(RTX_E_INVALID, "bad frame description");
    if (f->depth < 0 || f->depth > RTX_MAX_DEPTH) return fail(RTX_E_LIMIT, "depth outside [0, RTX_MAX_DEPTH]");
    return RTX_OK;
}

DevCamera devCamera(const rtx_camera *cam) {
    DevCamera d;
    d.ox = cam->origin[0], d.oy = cam->origin[1], d.oz = cam->origin[2];
    for (int k = 0; k < 9; ++k) d.m[k] = cam->basis[k];
    d.tan_half_fov = cam->tan_half_fov;
    d.add_origin = cam->add_origin;
    return d;
}

// Screen-space rectangle (pixel columns, image rows) that contains the projection of a world-space
// box under the camera model of primaryDir: dir ~ M v + k c with v = (x, y, z_focal), k = add_origin.
// Inverting for a point P: M v = lambda (P - c) - k c, lambda fixed by v.z = z_focal.  Returns false
// (meaning "anywhere") when a corner is not in front of the image plane.
bool projectBox(const rtx_camera *cam, int W, int H, const float lo[3], const float hi[3], float rect[4]) {
    const float *M = cam->basis;
    const float zf = (float)(-H) / (2.0f * cam->tan_half_fov);
    float b[3] = {0, 0, 0};
    if (cam->add_origin)
        for (int r = 0; r < 3; ++r) b[r] = M[3 * r] * cam->origin[0] + M[3 * r + 1] * cam->origin[1] + M[3 * r + 2] * cam->origin[2];  // (M^T c)[r]
    rect[0] = rect[2] = 1e30f, rect[1] = rect[3] = -1e30f;
    for (int k = 0; k < 8; ++k) {
        const float P[3] = {(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]};
        const float d[3] = {P[0] - cam->origin[0], P[1] - cam->origin[1], P[2] - cam->origin[2]};
        float a[3];
        for (int r = 0; r < 3; ++r) a[r] = M[3 * r] * d[0] + M[3 * r + 1] * d[1] + M[3 * r + 2] * d[2];
        if (std::fabs(a[2]) < 1e-6f) return false;
        const float lambda = (zf + b[2]) / a[2];
        if (!(lambda > 0.0f)) return false;
        const float x = lambda * a[0] - b[0], y = lambda * a[1] - b[1];
        const float col = x + W / 2.0f - 0.5f, row = H / 2.0f - 0.5f - y;
        rect[0] = std::fmin(rect[0], col), rect[1] = std::fmax(rect[1], col);
        rect[2] = std::fmin(rect[2], row), rect[3] = std::fmax(rect[3], row);
    }
    return true;
}

// Screen rectangles of what makes tiles expensive, for k_tile_classes: class 0 = the mesh, class 1 = a
// mirror or glass sphere.  Returns false when the mesh may be anywhere on screen (a corner of its box is
// not in front of the image plane): no tile counts as cheap then.
bool tileRects(const rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f, TileRects &rects) {
    rects.n = 0;
    const DevScene &d = c->dev;
    if (d.n_mesh_tris > 0) {
        if (!projectBox(cam, f->width, f->height, d.mesh_lo, d.mesh_hi, rects.r[rects.n])) return false;
        rects.cls[rects.n++] = 0;
    }
    for (size_t i = 0; i < c->spheres.size() && !c->big_spheres; ++i) {  // (a large set is everywhere: no classes from it)
        const Material &m = c->materials[c->sphere_mat[i]];
        if (m.a2 == 0.0f && m.a3 == 0.0f) continue;
        const float4 sp = c->spheres[i];
        const float lo[3] = {sp.x - sp.w, sp.y - sp.w, sp.z - sp.w}, hi[3] = {sp.x + sp.w, sp.y + sp.w, sp.z + sp.w};
        if (rects.n < kMaxRects && projectBox(cam, f->width, f->height, lo, hi, rects.r[rects.n])) rects.cls[rects.n++] = 1;
    }
    return true;
}

// Grows the context's own scheduling memory to n tiles: cost per tile and the list built from it.
int scheduleBuffers(rtx_ctx *c, size_t n) {
    SchedMem &m = c->sched;
    if (n <= m.cap) return RTX_OK;
    HIP_OR_FAIL(hipDeviceSynchronize());  // an earlier launch may still use the old buffers
    (void)hipFree(m.cost);
    (void)hipFree(m.entries);
    m.cost = nullptr, m.entries = nullptr, m.cap = 0;
    c->sched_state.have_cost = false, c->sched_state.by_cost_key.valid = false, c->sched_state.static_key.valid = false;
    HIP_OR_FAIL(hipMalloc((void **)&m.cost, n * sizeof(unsigned)));
    HIP_OR_FAIL(hipMalloc((void **)&m.entries, 4 * n * sizeof(unsigned)));
    HIP_OR_FAIL(hipMemset(m.cost, 0, n * sizeof(unsigned)));
    m.cap = n;
    return RTX_OK;
}

// Is `cam` close enough to the camera `o` of the frame whose tile costs are in hand for those costs to say
// where this frame's expensive tiles are?  (Same projection, the basis turned by a few degrees at
// most, the origin moved by a few per cent of its distance to the scene.)
bool nearbyView(const rtx_ctx *c, const rtx_camera &o, const rtx_camera *cam) {
    if (cam->tan_half_fov != o.tan_half_fov || cam->add_origin != o.add_origin) return false;
    for (int k = 0; k < 9; ++k)
        if (!(std::fabs(cam->basis[k] - o.basis[k]) <= 0.05f)) return false;
    const float *b = c->dev.object_bounds;
    float dist2 = 0.0f, move2 = 0.0f, size2 = 0.0f;
    for (int k = 0; k < 3; ++k) {
        const bool bounded = b[k] <= b[3 + k];
        const float centre = bounded ? 0.5f * (b[k] + b[3 + k]) : 0.0f, half = bounded ? 0.5f * (b[3 + k] - b[k]) : 0.0f;
        dist2 += (cam->origin[k] - centre) * (cam->origin[k] - centre);
        move2 += (cam->origin[k] - o.origin[k]) * (cam->origin[k] - o.origin[k]);
        size2 += half * half;
    }
    return move2 <= 0.05f * 0.05f * (dist2 + size2);
}

// The two box filters in front of the sphere loop of a small sphere set (castBegin) must let through every ray
// that Sphere::rayIntersect can accept.  Its rounding lets a ray pass the centre at up to sqrt(r^2 + k^2 L^2)
// instead of r, L being the distance from the ray's origin to the sphere and k = kSphereSlack (see sphereWalk):
// a small sphere far from the origin is "hit" well outside its box.  So the boxes are sized per launch, for the
// largest L a ray of that launch can have.  Origins are the camera (or the caller's ray origins), points on the
// bounded objects, and points on the water / island planes, which lie within one far clip of either (a ray
// that leaves a horizontal plane does not come back to it unless an object sends it).
void sphereGateBounds(const rtx_ctx *c, DevScene &s, float origin_reach) {
    if (s.n_spheres <= 0 || s.sphere_bvh) return;
    const float *m = s.sphere_root_c;
    float object_reach = 0.0f;  // the farthest point of the bounded objects from the spheres' centre
    for (int k = 0; k < 8; ++k) {
        float d2 = 0.0f;
        for (int a = 0; a < 3; ++a) {
            const float lo = c->bounded_objects[a], hi = c->bounded_objects[3 + a];
            if (!(lo <= hi)) continue;
            const float p = ((k >> a) & 1) ? hi : lo;
            d2 += (p - m[a]) * (p - m[a]);
        }
        object_reach = std::fmax(object_reach, std::sqrt(d2));
    }
    const float L = (std::fmax(origin_reach, object_reach) + std::fabs(c->far_clip) + s.sphere_root_h) * 1.01f;
    float b[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};
    for (const float4 &sp : c->spheres) {
        const float r = std::fabs(sp.w);
        const float reach = std::sqrt(r * r + kSphereSlack * kSphereSlack * L * L) * 1.001f + 1e-6f;
        const float p[3] = {sp.x, sp.y, sp.z};
        for (int a = 0; a < 3; ++a) {
            const float e = 1e-5f * std::fabs(p[a]);
            b[a] = std::fmin(b[a], p[a] - reach - e), b[3 + a] = std::fmax(b[3 + a], p[a] + reach + e);
        }
    }
    for (int a = 0; a < 3; ++a) {
        s.sphere_bounds[a] = b[a], s.sphere_bounds[3 + a] = b[3 + a];
        s.object_bounds[a] = std::fmin(b[a], c->bounded_objects[a]), s.object_bounds[3 + a] = std::fmax(b[3 + a], c->bounded_objects[3 + a]);
    }
}
float distanceToSpheres(const rtx_ctx *c, const float o[3]) {
    const float *m = c->dev.sphere_root_c;
    return std::sqrt((o[0] - m[0]) * (o[0] - m[0]) + (o[1] - m[1]) * (o[1] - m[1]) + (o[2] - m[2]) * (o[2] - m[2]));
}
float farthestOrigin(const rtx_ctx *c, const float *orig, int n) {
    float far = 0.0f;
    for (int i = 0; i < n; ++i) far = std::fmax(far, distanceToSpheres(c, orig + 3 * (size_t)i));
    return far;
}

// the scene as one launch sees it: the options of the moment, the sphere filters for where its rays start
DevScene sceneForLaunch(const rtx_ctx *c, float origin_reach) {
    DevScene s = c->dev;
    s.prune = c->prune, s.far_clip = c->far_clip, s.fprec = c->fprec;
    sphereGateBounds(c, s, origin_reach);
    return s;
}

using RenderKernel = void (*)(DevScene, DevCamera, DevFrame, unsigned, unsigned, const unsigned *, const unsigned *, unsigned, unsigned *,
                              unsigned *, unsigned *, unsigned char *, Counters *, unsigned, unsigned *, unsigned *);
const RenderKernel kRenderKernels[4] = {k_render<false, false, false>, k_render<true, false, false>, k_render<false, true, false>, k_render<true, true, false>};
const RenderKernel kQuadKernels[4] = {k_render<false, false, true>, k_render<true, false, true>, k_render<false, true, true>, k_render<true, true, true>};

// bytes of k_render's per-lane LDS
constexpr size_t renderLds(int stack_entries, int frame_slots) { return (size_t)(stack_entries + 6 * frame_slots) * kBlock * sizeof(int); }

// Blocks of every k_render variant that are resident at once (the kernel is a loop over tiles; more
// blocks would only queue).  Worked out once, in rtx_commit.
int residentBlocks(rtx_ctx *c) {
    hipDeviceProp_t prop;
    HIP_OR_FAIL(hipGetDeviceProperties(&prop, c->device));
    // traversal stacks as deep as the scene's trees need (one entry more: the walk stores before it counts)
    c->stack_entries = std::max(1, std::min(kStackDepth, std::max(c->bvh_depth, c->sphere_bvh_depth) + 1));
    for (int variant = 0; variant < 4; ++variant) {
        int per_cu = 0;
        HIP_OR_FAIL(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kRenderKernels[variant], kBlock, renderLds(c->stack_entries, 0)));
        c->resident_blocks[variant] = (per_cu > 0 ? per_cu : 1) * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1);
        // frame heads go to LDS as far as that costs no resident block
        hipFuncAttributes attr;
        HIP_OR_FAIL(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kRenderKernels[variant])));
        int slots = 0;
        while (slots < kMaxDepth - 1) {
            int with = 0;
            const size_t lds = renderLds(c->stack_entries, slots + 1);
            if (attr.sharedSizeBytes + lds > 64u * 1024u) break;  // a workgroup's LDS without opting in to more
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&with, kRenderKernels[variant], kBlock, lds) != hipSuccess || with < per_cu) break;
            ++slots;
        }
        (void)hipGetLastError();
        c->lds_frame_slots[variant] = slots;
        // the quad-mode launch with the same per-lane LDS
        int quad_per_cu = 0;
        HIP_OR_FAIL(hipOccupancyMaxActiveBlocksPerMultiprocessor(&quad_per_cu, kQuadKernels[variant], kBlock, renderLds(c->stack_entries, slots)));
        c->resident_quad[variant] = (quad_per_cu > 0 ? quad_per_cu : 1) * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1);
    }
    return RTX_OK;
}

// Enqueues one render on `stream`; output rows * width * 3 bytes at d_out.  `m` (at least as many tiles as
// the frame has) and `ss` say where the tile lists live and what they hold; the counters and the timing
// events belong to the context: one render of a context at a time.  Nothing here allocates or waits.
int launchRender(rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f, unsigned char *d_out, hipStream_t stream, const SchedMem &m,
                 SchedState &ss, bool feedback) {
    const int rows = rtx_frame_rows(f);
    if (rows == 0) return RTX_OK;
    DevFrame fr{f->width, f->height, f->depth, f->row_first, f->row_step, rows, c->spp, c->stack_entries, 0};
    const int tiles_x = (f->width + 7) / 8, tiles_y = (rows + 7) / 8;
    const long long n_tiles = (long long)tiles_x * tiles_y;
    if (n_tiles > (long long)kTileMask || (size_t)n_tiles > m.cap) return fail(RTX_E_LIMIT, "frame too large for one launch");
    const DevScene s = sceneForLaunch(c, distanceToSpheres(c, cam->origin));
    // Longest tiles first.  Every frame records what each of its tiles cost.  A frame of the same frame
    // description whose camera is the previous frame's or close to it gets its list from those costs, and
    // a frame of exactly the view a list was built for reuses the list.  Anything else -- the first
    // frame, a cut to another view -- goes by classes from projected bounding boxes (k_tile_classes).
    // Either way the list is built by k_bin_*: three small launches in stream order, about 25 us.
    enum { kReuse, kFromCosts, kFromProbes, kFromClasses, kNatural } how = kNatural;
    TileRects rects;
    unsigned batch_from = 0;
    if (feedback && ss.by_cost_key.is(cam, f)) how = kReuse;
    else if (feedback && ss.have_cost && ss.cost_view.valid && memcmp(f, &ss.cost_view.frame, sizeof(rtx_frame)) == 0 &&
             nearbyView(c, ss.cost_view.cam, cam))
        how = kFromCosts;
    else if (ss.static_key.is(cam, f)) how = kReuse;
    else if (c->tile_probe && n_tiles >= 4096) how = kFromProbes;          // (a small frame is over before a probe pass would pay)
    else if (!tileRects(c, cam, f, rects)) batch_from = (unsigned)n_tiles;  // natural order, nothing is cheap
    else if (rects.n > 0) how = kFromClasses;                              // (no boxes at all: natural order, everything is cheap)
    const unsigned *order = how == kNatural ? nullptr : m.entries;
    const unsigned *list_hdr = how == kNatural ? nullptr : m.list_hdr;
    // the general build of the kernel only where it is needed: a large sphere set, several samples per pixel
    const bool wide = c->big_spheres || c->spp > 1 || c->long_lists;
    const int variant = (c->stats ? 1 : 0) + (wide ? 2 : 0);
    const int resident = c->resident_blocks[variant];
    fr.lds_frames = std::max(0, std::min(f->depth - 1, c->lds_frame_slots[variant]));  // a frame of depth d has at most d - 1 nodes waiting
    const long long want = (4 * n_tiles + kBlock / 64 - 1) / (kBlock / 64);  // room for the quarters of quad tiles
    const unsigned blocks = (unsigned)(want < resident ? want : resident);
    // (a stream that is being captured into a graph takes no timing events: rtx_get_stats could not wait for them)
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capture) != hipSuccess) (void)hipGetLastError(), capture = hipStreamCaptureStatusNone;
    const bool timed = capture == hipStreamCaptureStatusNone;
    if (timed) HIP_OR_FAIL(hipEventRecord(c->ev0, stream));
    if (c->stats) HIP_OR_FAIL(hipMemsetAsync(c->d_counters, 0, sizeof(Counters), stream));
    unsigned *const heads = m.heads + (ss.heads_flip ? kQueues * kHeadStride : 0u);
    unsigned *const heads_next = m.heads + (ss.heads_flip ? 0u : kQueues * kHeadStride);
    if (how == kFromCosts || how == kFromClasses || how == kFromProbes) {
        const bool by_cost = how == kFromCosts;
        // RTX_OPT_QUAD: 0 never, 1 the tiles that would hold the frame up (never one shorter than 0.2 ms,
        // never more than an eighth of all), 2 every tile; never on a list made from classes
        unsigned cap = !by_cost || !m.quarters || c->quad_mode == 0 ? 0u : (c->quad_mode == 2 ? (unsigned)n_tiles : (unsigned)(n_tiles / 8));
        float quad_share = c->quad_mode == 2 ? 0.0f : kQuadShare;
        unsigned quad_min = c->quad_mode == 2 ? 0u : 20000u;
        static const unsigned probe_quad = getenv("RTX_PROBE_QUAD") ? (unsigned)atoi(getenv("RTX_PROBE_QUAD")) : 0u;
        if (how == kFromProbes && probe_quad && m.quarters && c->quad_mode == 1) cap = (unsigned)(n_tiles / 8), quad_share = 0.0f, quad_min = probe_quad;
        const unsigned per = (unsigned)((n_tiles + kBinBlocks - 1) / kBinBlocks);
        unsigned *const d_head = m.bins + kBinBlocks * kCostBins;
        if (how == kFromProbes) {
            // estimates into the list's memory first (free until k_bin_scatter writes it), spread over neighbours into `cost`
            const dim3 grid((unsigned)((n_tiles + kBlock - 1) / kBlock));
            if (wide) hipLaunchKernelGGL(k_tile_probe<true>, grid, dim3(kBlock), 0, stream, s, devCamera(cam), fr, (unsigned)tiles_x, (unsigned)n_tiles, m.entries);
            else hipLaunchKernelGGL(k_tile_probe<false>, grid, dim3(kBlock), 0, stream, s, devCamera(cam), fr, (unsigned)tiles_x, (unsigned)n_tiles, m.entries);
            if (!tileRects(c, cam, f, rects)) rects.n = 0;
            hipLaunchKernelGGL(k_tile_spread, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, stream, m.entries, rects, (unsigned)n_tiles, (unsigned)tiles_x,
                               f->row_first, f->row_step, m.cost);
        }
        if (how == kFromClasses)
            hipLaunchKernelGGL(k_tile_classes, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, stream, rects, (unsigned)n_tiles,
                               (unsigned)tiles_x, f->row_first, f->row_step, m.cost);
        hipLaunchKernelGGL(k_bin_count, dim3(kBinBlocks), dim3(kBinThreads), 0, stream, m.cost, (unsigned)n_tiles, per, m.bins, m.bin_sums);
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kCostBins), 0, stream, m.bins, m.bin_sums, (unsigned)n_tiles,
                           (unsigned)resident * (kBlock / 64u), quad_share, quad_min, cap,
                           by_cost ? 0u : (how == kFromProbes ? kProbeCheapCost : kClassCheapCost), m.list_hdr, d_head,
                           m.quarters ? c->host_quads : nullptr);
        hipLaunchKernelGGL(k_bin_scatter, dim3(kBinBlocks), dim3(kBinThreads), 0, stream, m.cost, (unsigned)n_tiles, per, m.bins, d_head,
                           m.entries);
        HIP_OR_FAIL(hipGetLastError());
        ss.by_cost_key.valid = ss.static_key.valid = false;
        (how == kFromCosts ? ss.by_cost_key : ss.static_key).set(cam, f);
        ss.list_quads = cap > 0;
    }
    // The quad-mode tiles go to a launch of their own (k_render<.., LOOK = true>) on the context's side stream, between a
    // fork off `stream` and a join back into it, so that the two launches run side by side.  Not while a graph is being
    // captured (one stream, one node) and not from the static scheduling memory (no quarters there): the main launch
    // then traces them itself.  The side launch gets one workgroup per quad-mode tile of the list, as the host last saw
    // their number (k_bin_scan writes it into page-locked memory: exact for a kept list once the frame that built it
    // is through, the previous list's otherwise -- with too few, each takes more quarters; none seen yet, no side
    // launch), and the main launch that many fewer: together they fit the GPU at once.
    // It pays where the longest chains bound the frame, not the throughput (RTX_OPT_QUAD_KERNEL = 1: frames of at most
    // kSideLaunchTiles tiles per wave that is resident; measured on the default scene: 800x600 -9 %, 1920x1080 -10 %,
    // an eighth / a quarter of a 3840x2160 frame -12 %, half of one +11 %, all of one +8 %).
    const unsigned quads_seen = c->host_quads ? *(volatile unsigned *)c->host_quads : 0u;
    const bool chain_bound = n_tiles <= (long long)kSideLaunchTiles * resident * (kBlock / 64);
    const bool split = order && ss.list_quads && m.quarters && (c->quad_kernel == 2 || (c->quad_kernel == 1 && chain_bound)) && c->quad_stream && timed &&
                       quads_seen > 0 && blocks > 1;
    unsigned main_blocks = blocks;
    if (split) {
        const unsigned quad_blocks = std::min<unsigned>(std::min<unsigned>(quads_seen, blocks / 2u), (unsigned)c->resident_quad[variant]);
        main_blocks = blocks - quad_blocks;
        HIP_OR_FAIL(hipEventRecord(c->ev_fork, stream));
        HIP_OR_FAIL(hipStreamWaitEvent(c->quad_stream, c->ev_fork, 0));
        unsigned *const qheads = m.heads + kHeadWords + (ss.quad_flip ? kQueues * kHeadStride : 0u);
        unsigned *const qheads_next = m.heads + kHeadWords + (ss.quad_flip ? 0u : kQueues * kHeadStride);
        hipLaunchKernelGGL(kQuadKernels[variant], dim3(quad_blocks), dim3(kBlock), renderLds(fr.stack_entries, fr.lds_frames), c->quad_stream, s, devCamera(cam), fr,
                           (unsigned)tiles_x, (unsigned)n_tiles, order, list_hdr, batch_from, heads, heads_next, m.cost, d_out, c->d_counters, 1u, qheads, qheads_next);
        HIP_OR_FAIL(hipGetLastError());
        ss.quad_flip = !ss.quad_flip;  // (the side launch has cleared the other set for its next time, whenever that is)
        HIP_OR_FAIL(hipEventRecord(c->ev_join, c->quad_stream));
    }
    hipLaunchKernelGGL(kRenderKernels[variant], dim3(main_blocks), dim3(kBlock), renderLds(fr.stack_entries, fr.lds_frames), stream, s, devCamera(cam), fr, (unsigned)tiles_x, (unsigned)n_tiles,
                       order, list_hdr, batch_from, heads, heads_next, m.cost, d_out, c->d_counters, split ? 1u : 0u, (unsigned *)nullptr, (unsigned *)nullptr);
    HIP_OR_FAIL(hipGetLastError());
    if (split) HIP_OR_FAIL(hipStreamWaitEvent(stream, c->ev_join, 0));
    ss.heads_flip = !ss.heads_flip;  // only now: a launch that failed has cleared nothing
    ss.cost_view.set(cam, f), ss.have_cost = true;
    if (timed) HIP_OR_FAIL(hipEventRecord(c->ev1, stream));
    c->timed = timed;
    return RTX_OK;
}

// ---- the level-by-level pipeline (rtx_wavefront.hpp) ----
using WfTraceKernel = void (*)(DevScene, DevCamera, DevFrame, WfStrip, WfArgs, Counters *);
using WfShadowKernel = void (*)(DevScene, WfArgs, Counters *);
using WfShadeKernel = void (*)(DevScene, DevFrame, WfArgs, Counters *);
const WfTraceKernel kWfTrace[4] = {k_wf_trace<false, false>, k_wf_trace<true, false>, k_wf_trace<false, true>, k_wf_trace<true, true>};
const WfShadowKernel kWfShadow[4] = {k_wf_shadow<false, false>, k_wf_shadow<true, false>, k_wf_shadow<false, true>, k_wf_shadow<true, true>};
const WfShadowKernel kWfShadowPool[2] = {k_wf_shadow_pool<false>, k_wf_shadow_pool<true>};
const WfShadeKernel kWfShade[4] = {k_wf_shade<false, false>, k_wf_shade<true, false>, k_wf_shade<false, true>, k_wf_shade<true, true>};

int wavefrontBlocks(rtx_ctx *c) {
    hipDeviceProp_t prop;
    HIP_OR_FAIL(hipGetDeviceProperties(&prop, c->device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    // traversal stacks as deep as the scene's trees need (one entry more: the walk stores before it counts)
    c->wf_stack_entries = std::max(1, std::min(kStackDepth, std::max(c->bvh_depth, c->sphere_bvh_depth) + 1));
    const size_t lds = (size_t)c->wf_stack_entries * kBlock * sizeof(int);
    for (int v = 0; v < 4; ++v) {
        int per_cu = 0;
        HIP_OR_FAIL(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kWfTrace[v], kBlock, lds));
        c->wf_blocks[0][v] = (per_cu > 0 ? per_cu : 1) * cus;
        HIP_OR_FAIL(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kWfShadow[v], kBlock, lds));
        c->wf_blocks[1][v] = (per_cu > 0 ? per_cu : 1) * cus;
    }
    for (int v = 0; v < 4; ++v) {
        int per_cu = 0;
        HIP_OR_FAIL(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kWfShade[v], kBlock, 0));
        c->wf_blocks[2][v] = (per_cu > 0 ? per_cu : 1) * cus;
    }
    return RTX_OK;
}

constexpr size_t kWfThinBytes = 3 * sizeof(float4) + sizeof(uint2);  // a thin record: own, kid0, kid1, meta

// bits that hold a light's number in a shadow request
unsigned wfLightBits(int n_lights) {
    unsigned bits = 1;
    while ((1 << bits) < n_lights) ++bits;
    return bits;
}

WfPlan wavefrontPlan(int D, int n_lights, size_t strip) {
    WfPlan p;
    const size_t nl = (size_t)(n_lights > 0 ? n_lights : 1);
    size_t bytes = sizeof(WfCounters) + 256 + strip * sizeof(float4);  // counters, sample colours
    size_t req_chunks = (strip + kWfChunk - 1) / kWfChunk, fat_entries = 0;  // chunks of the level's requests
    for (int l = 0; l < D; ++l) {
        const size_t node_cap = (req_chunks / kWfSub + 1) * kWfChunk;  // chunks c with c % kWfSub == k, 64 nodes each
        const size_t node_chunks = (size_t)kWfSub * (node_cap / kWfChunk);
        const size_t req_cap = (node_chunks / kWfSub + 1) * kWfChunk * 2;  // two children per node
        p.node_cap[l] = (unsigned)node_cap, p.shadow_cap[l] = (unsigned)(node_cap * nl);
        // a node's index shares the 32 bits of a shadow request with the light's number and must fit a link's 28
        if (kWfSub * node_cap >= std::min<size_t>((size_t)1 << (32 - wfLightBits(n_lights)), (size_t)kWfMaxIndex) || kWfSub * node_cap * nl >= 0xffffffffull)
            p.fits = false;
        fat_entries = std::max(fat_entries, kWfSub * node_cap);
        if (l + 1 < D) {
            p.req_cap[l + 1] = (unsigned)req_cap;
            if (kWfSub * req_cap >= (size_t)kWfMaxIndex) p.fits = false;
            // requests (the two classes of a sub-list share its region), thin records
            bytes += kWfSub * req_cap * 2 * sizeof(float4) + kWfSub * node_cap * kWfThinBytes + 1024;
            req_chunks = (size_t)kWfSub * (req_cap / kWfChunk) + kWfMaxLists;  // all entries in whole chunks + a partly filled one per sub-list
        }
    }
    if (strip >= (size_t)kWfMaxIndex) p.fits = false;
    bytes += fat_entries * (4 * sizeof(float4) + sizeof(unsigned) + nl * sizeof(unsigned)) + 2048;  // fat records, shadow lists
    p.bytes = bytes;
    return p;
}

// Sizes the pipeline's arrays for this frame: the whole frame in one strip if that fits the budget -- the
// caller's (RTX_OPT_PIPELINE_BUDGET_MB) or three quarters of what the GPU has free, whichever is less -- otherwise
// strips of tile rows.  A block that is large enough is kept and carved anew (no allocation, no wait); a larger
// one is allocated after a device-wide wait, in halved strips if the GPU refuses.  RTX_E_NOMEM when not even the
// arrays of a single tile fit: the caller then renders with the one-kernel pipeline.
int wavefrontBuffers(rtx_ctx *c, int D, size_t frame_samples, int spp) {
    WfMem &w = c->wf;
    const int nl = c->dev.n_lights;
    const size_t unit = (size_t)64 * spp;  // samples of one tile
    const size_t slack = 256 * (4 * kWfLevels + 8);  // alignment of the carved arrays
    size_t budget = c->wf_budget;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, (free_b + w.block_bytes) / 4 * 3);
        else (void)hipGetLastError();
        budget = std::max(budget, w.block_bytes);  // what is already there may always be used
    }
    size_t strip = (frame_samples + unit - 1) / unit * unit;
    WfPlan plan = wavefrontPlan(D, nl, strip);
    for (;;) {
        while ((plan.bytes + slack > budget || !plan.fits) && strip > unit) {  // halve until it fits
            strip = std::max(unit, strip / 2 / unit * unit);
            plan = wavefrontPlan(D, nl, strip);
        }
        if (plan.bytes + slack > budget || !plan.fits) return fail(RTX_E_NOMEM, "the level-by-level arrays of a single tile do not fit the memory budget");
        if (w.block && w.block_bytes >= plan.bytes + slack) break;
        HIP_OR_FAIL(hipDeviceSynchronize());  // an earlier launch may still use the old block
        (void)hipFree(w.block);
        w = WfMem();
        const hipError_t e = hipMalloc(&w.block, plan.bytes + slack);
        if (e == hipSuccess) {
            w.block_bytes = plan.bytes + slack;
            break;
        }
        (void)hipGetLastError();
        w.block = nullptr;
        if (e != hipErrorOutOfMemory) return failHip(e, "level-by-level arrays");
        budget = (plan.bytes + slack) / 2;  // the GPU is fuller than it said: try half
        if (strip <= unit) return fail(RTX_E_NOMEM, "no device memory for the level-by-level arrays of a single tile");
    }
    if (w.depth == D && w.n_lights == nl && w.strip_samples == strip) return RTX_OK;  // carved for this plan already
    w.strip_samples = strip, w.depth = D, w.n_lights = nl;
    w.plan = plan;
    for (int l = 0; l <= kWfLevels; ++l) w.ray_o[l] = w.ray_d[l] = w.thin[l] = nullptr;
    w.pt = w.nrm = w.vw = w.dif = nullptr, w.hidden = nullptr, w.shadow_q = nullptr;
    unsigned char *p = static_cast<unsigned char *>(w.block);
    auto take = [&](size_t n) {
        unsigned char *q = p;
        p += (n + 255) / 256 * 256;
        return q;
    };
    w.counters = reinterpret_cast<WfCounters *>(take(sizeof(WfCounters)));
    w.samples = reinterpret_cast<float4 *>(take(strip * sizeof(float4)));
    size_t fat_entries = 0;
    for (int l = 0; l < D; ++l) {
        fat_entries = std::max(fat_entries, (size_t)kWfSub * plan.node_cap[l]);
        if (l + 1 < D) {
            w.ray_o[l + 1] = reinterpret_cast<float4 *>(take((size_t)kWfSub * plan.req_cap[l + 1] * sizeof(float4)));
            w.ray_d[l + 1] = reinterpret_cast<float4 *>(take((size_t)kWfSub * plan.req_cap[l + 1] * sizeof(float4)));
            w.thin[l] = reinterpret_cast<float4 *>(take((size_t)kWfSub * plan.node_cap[l] * kWfThinBytes));
        }
    }
    if (D > 0) {
        w.pt = reinterpret_cast<float4 *>(take(fat_entries * sizeof(float4)));
        w.nrm = reinterpret_cast<float4 *>(take(fat_entries * sizeof(float4)));
        w.vw = reinterpret_cast<float4 *>(take(fat_entries * sizeof(float4)));
        w.dif = reinterpret_cast<float4 *>(take(fat_entries * sizeof(float4)));
        w.hidden = reinterpret_cast<unsigned *>(take(fat_entries * sizeof(unsigned)));
        w.shadow_q = reinterpret_cast<unsigned *>(take(fat_entries * (size_t)(nl > 0 ? nl : 1) * sizeof(unsigned)));
    }
    if ((size_t)(p - static_cast<unsigned char *>(w.block)) > w.block_bytes) {
        w.depth = -1;
        return fail(RTX_E_NOMEM, "level-by-level arrays overrun their block");
    }
    return RTX_OK;
}

WfArgs wfArgs(const WfMem &w, int l, int launch) {
    WfArgs a;
    a.req = WfList{w.counters->n[0][l], w.plan.req_cap[l], 2 * kWfSub};
    a.nodes = WfList{w.counters->n[1][l], w.plan.node_cap[l], kWfSub};
    a.shadows = WfList{w.counters->n[2][l], w.plan.shadow_cap[l], 2 * kWfSub};
    a.next_req = WfList{w.counters->n[0][l + 1], w.plan.req_cap[l + 1], 2 * kWfSub};
    a.work = w.counters->work[launch][l];
    a.ray_o = w.ray_o[l], a.ray_d = w.ray_d[l], a.next_o = w.ray_o[l + 1], a.next_d = w.ray_d[l + 1];
    a.pt = w.pt, a.nrm = w.nrm, a.vw = w.vw, a.dif = w.dif, a.hidden = w.hidden, a.shadow_q = w.shadow_q, a.thin = w.thin[l], a.parent = l == 0 ? w.samples : w.thin[l - 1];
    a.thin_n = kWfSub * w.plan.node_cap[l], a.parent_n = l == 0 ? 0u : kWfSub * w.plan.node_cap[l - 1];
    a.level = l;
    a.light_bits = wfLightBits(w.n_lights);
    return a;
}

// Which pipeline renders this context's frames (RTX_OPT_PIPELINE).  Left to itself the library goes level by
// level when the spheres sit behind a BVH -- measured on the 10 000-sphere scene: 122 ms against 177 ms for a
// 4096 x 4096 frame with 4 samples per pixel -- and uses the one-kernel pipeline otherwise (0.92 ms against
// 2.0 ms for the default scene at 3840 x 2160).
bool byLevels(const rtx_ctx *c) { return c->pipeline == 1 || (c->pipeline == 2 && c->big_spheres && c->dev.n_lights <= kWfMaxLights); }

// Enqueues one render through the level-by-level pipeline.
int launchWavefront(rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f, unsigned char *d_out, hipStream_t stream) {
    const int rows = rtx_frame_rows(f);
    if (rows == 0) return RTX_OK;
    if (c->dev.n_lights > kWfMaxLights) return fail(RTX_E_LIMIT, "RTX_OPT_PIPELINE = 1 handles at most 32 lights");
    DevFrame fr{f->width, f->height, f->depth, f->row_first, f->row_step, rows, c->spp};
    const unsigned tiles_x = (unsigned)(f->width + 7) / 8, tiles_y = (unsigned)(rows + 7) / 8;
    const unsigned long long n_tiles = (unsigned long long)tiles_x * tiles_y;
    if (n_tiles > 0xffffffffull / (64u * (unsigned)c->spp)) return fail(RTX_E_LIMIT, "frame too large");
    const int D = f->depth;
    if (int e = wavefrontBuffers(c, D, (size_t)n_tiles * 64 * c->spp, c->spp)) return e;
    const WfMem &w = c->wf;
    const DevScene s = sceneForLaunch(c, distanceToSpheres(c, cam->origin));
    const DevCamera dcam = devCamera(cam);
    const bool wide = c->big_spheres || c->spp > 1 || c->long_lists;
    const int variant = (c->stats ? 1 : 0) + (wide ? 2 : 0);
    HIP_OR_FAIL(hipEventRecord(c->ev0, stream));
    if (c->stats) HIP_OR_FAIL(hipMemsetAsync(c->d_counters, 0, sizeof(Counters), stream));
    const unsigned strip_tiles = (unsigned)(w.strip_samples / (64u * (unsigned)c->spp));
    for (unsigned long long first = 0; first < n_tiles; first += strip_tiles) {
        WfStrip strip{(unsigned)first, (unsigned)std::min<unsigned long long>(strip_tiles, n_tiles - first), tiles_x};
        const unsigned long long samples = (unsigned long long)strip.n_tiles * 64u * (unsigned)c->spp;
        auto grid = [&](int resident, unsigned long long items) {
            const unsigned long long chunks = (items + kBlock - 1) / kBlock + kWfMaxLists;
            return dim3((unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(chunks, (unsigned long long)resident)));
        };
        const size_t lds = (size_t)c->wf_stack_entries * kBlock * sizeof(int);
        hipLaunchKernelGGL(k_wf_clear, dim3(4), dim3(kBlock), 0, stream, w.counters, std::min(D, kWfLevels));
        for (int l = 0; l < (D > 0 ? D : 1); ++l) {
            hipLaunchKernelGGL(kWfTrace[variant], grid(c->wf_blocks[0][variant], samples << l), dim3(kBlock), lds, stream, s, dcam, fr, strip,
                               wfArgs(w, l, 0), c->d_counters);
            if (D == 0) break;
            hipLaunchKernelGGL(s.sphere_bvh && c->wf_pool ? kWfShadowPool[variant & 1] : kWfShadow[variant], grid(c->wf_blocks[1][variant], (samples << l) * (unsigned)std::max(1, s.n_lights)), dim3(kBlock), lds,
                               stream, s, wfArgs(w, l, 1), c->d_counters);
            hipLaunchKernelGGL(kWfShade[variant], grid(c->wf_blocks[2][variant], samples << l), dim3(kBlock), 0, stream, s, fr,
                               wfArgs(w, l, 2), c->d_counters);
        }
        for (int l = D - 2; l >= 0; --l) {
            hipLaunchKernelGGL(k_wf_resolve, grid(c->wf_blocks[2][0] * 2, samples << l), dim3(kBlock), 0, stream, wfArgs(w, l, 3));
        }
        hipLaunchKernelGGL(k_wf_pixels, grid(c->wf_blocks[2][0] * 2, (unsigned long long)strip.n_tiles * 64u), dim3(kBlock), 0, stream, fr, strip,
                           w.samples, d_out);
        HIP_OR_FAIL(hipGetLastError());
    }
    HIP_OR_FAIL(hipEventRecord(c->ev1, stream));
    c->timed = true;
    return RTX_OK;
}

}  // namespace

extern "C" {

const char *rtx_last_error(void) { return g_err.c_str(); }

int rtx_create(rtx_ctx **out, int device_id) {
    if (!out) return fail(RTX_E_INVALID, "rtx_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(RTX_E_DEVICE, std::string("rtx_create: no HIP device available (") +
                                      (e != hipSuccess ? hipGetErrorString(e) : "0 devices") + ")");
    if (device_id < 0 || device_id >= n) return fail(RTX_E_INVALID, "rtx_create: device id out of range");
    rtx_ctx *c = new rtx_ctx();
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_counters, sizeof(Counters))) != hipSuccess ||
        (e = hipMalloc((void **)&c->sched.heads, 2 * kHeadWords * sizeof(unsigned))) != hipSuccess ||  // the main launch's and the quad-mode launch's
        (e = hipMemset(c->sched.heads, 0, 2 * kHeadWords * sizeof(unsigned))) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->quad_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->host_quads, sizeof(unsigned), hipHostMallocDefault)) != hipSuccess ||
        (e = hipMalloc((void **)&c->sched.bins, kBinWords * sizeof(unsigned))) != hipSuccess ||
        (e = hipMalloc((void **)&c->sched.list_hdr, 4 * sizeof(unsigned))) != hipSuccess ||
        (e = hipMalloc((void **)&c->sched.bin_sums, kBinBlocks * sizeof(unsigned long long))) != hipSuccess) {
        rtx_destroy(c);
        return failHip(e, "rtx_create");
    }
    *c->host_quads = 0u;
    if (int rc = scratchOnDevice(device_id)) {
        rtx_destroy(c);
        return rc;
    }
    *out = c;
    return RTX_OK;
}

void rtx_destroy(rtx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->quad_stream) (void)hipStreamSynchronize(c->quad_stream), (void)hipStreamDestroy(c->quad_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->host_quads) (void)hipHostFree(c->host_quads);
    for (void *p : c->owned) (void)hipFree(p);
    (void)hipFree(c->d_counters);
    (void)hipFree(c->sched.heads);
    (void)hipFree(c->sched.bins);
    (void)hipFree(c->sched.bin_sums);
    (void)hipFree(c->sched.list_hdr);
    (void)hipFree(c->sched.cost);
    (void)hipFree(c->sched.entries);
    (void)hipFree(c->wf.block);
    (void)hipFree(c->d_frame);
    c->stage.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rtx_set_option(rtx_ctx *c, int option, uint32_t value) {
    if (!c) return fail(RTX_E_INVALID, "null context");
    float f;
    memcpy(&f, &value, 4);
    switch (option) {
        case RTX_OPT_STATS: c->stats = value != 0; break;
        case RTX_OPT_PRUNE: c->prune = value != 0; break;
        case RTX_OPT_FAR_CLIP: c->far_clip = f; break;
        case RTX_OPT_SURFACE_BIAS: c->fprec = f; break;
        case RTX_OPT_SCHEDULE_FEEDBACK: c->feedback = value != 0; break;
        case RTX_OPT_QUAD:
            if (value > 2) return fail(RTX_E_INVALID, "RTX_OPT_QUAD: 0 never, 1 automatic, 2 every tile");
            c->quad_mode = (int)value;
            c->sched_state.by_cost_key.valid = false;  // the entry list is rebuilt with the next frame
            break;
        case RTX_OPT_SPP:
            if (value != 1 && value != 4) return fail(RTX_E_INVALID, "RTX_OPT_SPP: 1 or 4");
            c->spp = (int)value;
            c->sched_state.have_cost = false, c->sched_state.by_cost_key.valid = false;  // what a tile costs changes with it
            break;
        case RTX_OPT_PIPELINE:
            if (value > 2) return fail(RTX_E_INVALID, "RTX_OPT_PIPELINE: 0 one kernel per frame, 1 level by level, 2 automatic");
            c->pipeline = (int)value;
            break;
        case RTX_OPT_PIPELINE_BUDGET_MB:
            if (value == 0) return fail(RTX_E_INVALID, "RTX_OPT_PIPELINE_BUDGET_MB: at least 1");
            c->wf_budget = (size_t)value << 20;
            break;
        case RTX_OPT_BVH_BUILDER:
            if (c->committed) return fail(RTX_E_STATE, "RTX_OPT_BVH_BUILDER must be set before rtx_commit");
            if (value > 3) return fail(RTX_E_INVALID, "RTX_OPT_BVH_BUILDER: 0 surface-area heuristic on the GPU, 1 Morton order on the GPU, 2 surface-area heuristic on the host, 3 clustering on the GPU");
            c->bvh_builder = (int)value;
            break;
        case RTX_OPT_QUAD_KERNEL:
            if (value > 2) return fail(RTX_E_INVALID, "RTX_OPT_QUAD_KERNEL: 0 never, 1 automatic, 2 whenever there are quad-mode tiles");
            c->quad_kernel = (int)value;
            break;
        case RTX_OPT_RAY_POOL: c->wf_pool = value != 0; break;
        case RTX_OPT_TILE_PROBE:
            c->tile_probe = value != 0;
            c->sched_state.static_key.valid = false;
            break;
        case RTX_OPT_PIN_OUTPUT:
            c->stage.may_pin = value != 0;
            if (!c->stage.may_pin) c->stage.release();
            break;
        case RTX_OPT_MESH_GRID:
            if (c->committed) return fail(RTX_E_STATE, "RTX_OPT_MESH_GRID must be set before rtx_commit");
            c->mesh_grid = value != 0;
            break;
        default: return fail(RTX_E_INVALID, "unknown option");
    }
    return RTX_OK;
}

#define NEED(c)                                           \
    if (!(c)) return fail(RTX_E_INVALID, "null context"); \
    if ((c)->committed) return fail(RTX_E_STATE, "scene is frozen after rtx_commit")

int rtx_set_materials(rtx_ctx *c, const float *m9, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && !m9)) return fail(RTX_E_INVALID, "rtx_set_materials: bad arguments");
    c->materials.resize(n);
    static_assert(sizeof(Material) == 9 * sizeof(float), "Material must be 9 packed floats");
    if (n) memcpy(c->materials.data(), m9, sizeof(Material) * (size_t)n);
    return RTX_OK;
}

int rtx_set_lights(rtx_ctx *c, const float *xyzi, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && !xyzi)) return fail(RTX_E_INVALID, "rtx_set_lights: bad arguments");
    c->lights.resize(n);
    for (int i = 0; i < n; ++i) c->lights[i] = make_float4(xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3]);
    return RTX_OK;
}

int rtx_set_spheres(rtx_ctx *c, const float *cx, const float *cy, const float *cz, const float *r, const int *mat, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && (!cx || !cy || !cz || !r || !mat))) return fail(RTX_E_INVALID, "rtx_set_spheres: bad arguments");
    if (int e = checkMat(c, mat, n)) return e;
    c->spheres.resize(n);
    for (int i = 0; i < n; ++i) c->spheres[i] = make_float4(cx[i], cy[i], cz[i], r[i]);
    c->sphere_mat.assign(mat, mat + n);
    return RTX_OK;
}

int rtx_set_boxes(rtx_ctx *c, const float *a, const float *b, const int *mat, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && (!a || !b || !mat))) return fail(RTX_E_INVALID, "rtx_set_boxes: bad arguments");
    if (int e = checkMat(c, mat, n)) return e;
    c->boxes.resize(6 * (size_t)n);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {  // Cube::load, src/SceneObjects.cpp:75-89
            const float p = a[3 * i + k], q = b[3 * i + k];
            c->boxes[6 * i + k] = p < q ? p : q;      // std::min(minP, maxP)
            c->boxes[6 * i + 3 + k] = p < q ? q : p;  // std::max(minP, maxP)
        }
    c->box_mat.assign(mat, mat + n);
    return RTX_OK;
}

int rtx_set_triangles(rtx_ctx *c, const float *v9, const int *mat, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && (!v9 || !mat))) return fail(RTX_E_INVALID, "rtx_set_triangles: bad arguments");
    if (int e = checkMat(c, mat, n)) return e;
    c->tris.assign(v9, v9 + 9 * (size_t)n);
    c->tri_mat.assign(mat, mat + n);
    return RTX_OK;
}

int rtx_set_mesh(rtx_ctx *c, const float *verts, int nv, const int *idx, int nt, int mat) {
    NEED(c);
    if (nv < 0 || nt < 0 || (nv > 0 && !verts) || (nt > 0 && !idx)) return fail(RTX_E_INVALID, "rtx_set_mesh: bad arguments");
    if (nt > 0)
        if (int e = checkMat(c, &mat, 1)) return e;
    for (int i = 0; i < 3 * nt; ++i)
        if (idx[i] < 0 || idx[i] >= nv) return fail(RTX_E_INVALID, "rtx_set_mesh: vertex index out of range");
    c->mesh_verts.assign(verts, verts + 3 * (size_t)nv);
    c->mesh_idx.assign(idx, idx + 3 * (size_t)nt);
    c->mesh_mat = mat;
    return RTX_OK;
}

int rtx_set_islands(rtx_ctx *c, const float *p, int n, float plane_y, int mat, const float *light_rgb, const float *dark_rgb) {
    NEED(c);
    if (n < 0 || (n > 0 && (!p || !light_rgb || !dark_rgb))) return fail(RTX_E_INVALID, "rtx_set_islands: bad arguments");
    if (n > 0)
        if (int e = checkMat(c, &mat, 1)) return e;
    c->islands.resize(n);
    for (int i = 0; i < n; ++i) c->islands[i] = make_float4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
    c->island_y = plane_y, c->island_mat = mat;
    if (n > 0)
        for (int k = 0; k < 3; ++k) c->light_green[k] = light_rgb[k], c->dark_green[k] = dark_rgb[k];
    return RTX_OK;
}

int rtx_set_water(rtx_ctx *c, const unsigned char *rgb, int w, int h, float plane_y, float scale, int mat) {
    NEED(c);
    if (w < 0 || h < 0 || ((w > 0 || h > 0) && (!rgb || w == 0 || h == 0))) return fail(RTX_E_INVALID, "rtx_set_water: bad arguments");
    if (w > 0)
        if (int e = checkMat(c, &mat, 1)) return e;
    c->water_w = w, c->water_h = h, c->water_y = plane_y, c->water_scale = scale, c->water_mat = mat;
    c->water.assign(rgb, rgb + (size_t)w * h * 3);
    return RTX_OK;
}

int rtx_set_envmap(rtx_ctx *c, const unsigned char *rgb, int w, int h, int row_shift) {
    NEED(c);
    if (w <= 0 || h <= 0 || !rgb) return fail(RTX_E_INVALID, "rtx_set_envmap: bad arguments");
    c->sky_w = w, c->sky_h = h, c->sky_shift = row_shift;
    c->sky.assign(rgb, rgb + (size_t)w * h * 3);
    return RTX_OK;
}

int rtx_commit(rtx_ctx *c) {
    if (!c) return fail(RTX_E_INVALID, "null context");
    if (c->committed) return fail(RTX_E_STATE, "already committed");
    if (c->sky_w == 0) return fail(RTX_E_STATE, "rtx_commit: no environment map (the reference cannot render without one)");
    if (c->materials.empty()) return fail(RTX_E_STATE, "rtx_commit: no materials");
    c->long_lists = (int)c->materials.size() > kMaxMaterials || (int)c->lights.size() > kMaxLights || (int)c->tri_mat.size() > kMaxSmallPrims ||
                    (int)c->box_mat.size() > kMaxSmallPrims || (int)c->islands.size() > kMaxSmallPrims;
    if (c->spheres.size() > (size_t)kMaxSpheres) return fail(RTX_E_LIMIT, "rtx_commit: more than 16 777 216 spheres");
    if (int e = bind(c)) return e;
    for (void *p : c->owned) (void)hipFree(p);  // (what a commit that failed half-way left behind)
    c->owned.clear();
    DevScene &d = c->dev;
    memset(&d, 0, sizeof d);
    Material *dm;
    float4 *dl, *dsp, *dis;
    int *dsm, *dtm, *dbm;
    float *dt, *db;
    unsigned char *dw, *dsk;
    if (int e = upload(c->materials, &dm, c->owned)) return e;
    if (int e = upload(c->lights, &dl, c->owned)) return e;
    if (int e = upload(c->spheres, &dsp, c->owned)) return e;
    if (int e = upload(c->sphere_mat, &dsm, c->owned)) return e;
    if (int e = upload(c->tris, &dt, c->owned)) return e;
    if (int e = upload(c->tri_mat, &dtm, c->owned)) return e;
    if (int e = upload(c->boxes, &db, c->owned)) return e;
    if (int e = upload(c->box_mat, &dbm, c->owned)) return e;
    if (int e = upload(c->islands, &dis, c->owned)) return e;
    if (int e = upload(c->water, &dw, c->owned)) return e;
    if (int e = upload(c->sky, &dsk, c->owned)) return e;
    d.materials = dm, d.n_materials = (int)c->materials.size();
    d.lights = dl, d.n_lights = (int)c->lights.size();
    d.spheres = dsp, d.sphere_mat = dsm, d.n_spheres = (int)c->spheres.size();
    d.tris = dt, d.tri_mat = dtm, d.n_tris = (int)c->tri_mat.size();
    d.boxes = db, d.box_mat = dbm, d.n_boxes = (int)c->box_mat.size();
    d.islands = dis, d.n_islands = (int)c->islands.size(), d.island_mat = c->island_mat, d.island_y = c->island_y;
    for (int k = 0; k < 3; ++k) d.light_green[k] = c->light_green[k], d.dark_green[k] = c->dark_green[k];
    d.water = dw, d.water_w = c->water_w, d.water_h = c->water_h, d.water_mat = c->water_mat;
    d.water_y = c->water_y, d.water_scale = c->water_scale;
    d.sky = dsk, d.sky_w = c->sky_w, d.sky_h = c->sky_h, d.sky_shift = c->sky_shift;

    // Conservative bounds per category for the cheap ray filter (slabMaybe).  They only have to
    // contain everything the exact tests can report a hit on; the padding covers the rounding of
    // those tests (a sphere or triangle test can accept a ray that passes a few ulp outside).
    {
        auto reset = [](float b[6]) {
            for (int k = 0; k < 3; ++k) b[k] = 3.0e38f, b[3 + k] = -3.0e38f;
        };
        auto grow = [](float b[6], float x, float y, float z) {
            const float p[3] = {x, y, z};
            for (int k = 0; k < 3; ++k) b[k] = p[k] < b[k] ? p[k] : b[k], b[3 + k] = p[k] > b[3 + k] ? p[k] : b[3 + k];
        };
        auto pad = [](float b[6]) {
            for (int k = 0; k < 3; ++k) {
                const float m = std::fmax(std::fabs(b[k]), std::fabs(b[3 + k]));
                const float e = 1e-4f * (b[3 + k] - b[k]) + 1e-5f * m + 1e-6f;
                b[k] -= e, b[3 + k] += e;
            }
        };
        reset(d.sphere_bounds), reset(d.tri_bounds), reset(d.box_bounds);
        for (const float4 &sp : c->spheres) {
            const float r = std::fabs(sp.w) * 1.001f;
            grow(d.sphere_bounds, sp.x - r, sp.y - r, sp.z - r), grow(d.sphere_bounds, sp.x + r, sp.y + r, sp.z + r);
        }
        for (size_t i = 0; i + 2 < c->tris.size(); i += 3) grow(d.tri_bounds, c->tris[i], c->tris[i + 1], c->tris[i + 2]);
        for (size_t i = 0; i + 2 < c->boxes.size(); i += 3) grow(d.box_bounds, c->boxes[i], c->boxes[i + 1], c->boxes[i + 2]);
        pad(d.sphere_bounds), pad(d.tri_bounds), pad(d.box_bounds);
    }

    // More spheres than the LDS staging holds: a BVH over their geometric boxes (widened by a few ulp;
    // the slack for the rounding of the sphere test is added per ray by the walk).
    d.sphere_bvh = nullptr;
    c->big_spheres = (int)c->spheres.size() > kMaxSmallPrims;
    if (!c->spheres.empty() && !c->big_spheres) {  // what the per-ray slack of the sphere gates is worked out from
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f}, rmax = 0.0f, h2 = 0.0f;
        for (const float4 &sp : c->spheres) {
            const float c3[3] = {sp.x, sp.y, sp.z}, r = std::fabs(sp.w);
            rmax = std::fmax(rmax, r);
            for (int k = 0; k < 3; ++k) lo[k] = std::fmin(lo[k], c3[k] - r), hi[k] = std::fmax(hi[k], c3[k] + r);
        }
        for (int k = 0; k < 3; ++k) {
            d.sphere_root_c[k] = 0.5f * (lo[k] + hi[k]);
            const float h = 0.5f * (hi[k] - lo[k]);
            h2 += h * h;
        }
        d.sphere_root_h = std::sqrt(h2) * 1.001f;
        d.sphere_rmax = rmax * 1.001f;
    }
    if (c->big_spheres) {
        std::vector<SahBox> boxes(c->spheres.size());
        SahBox all;
        all.reset();
        float rmax = 0.0f;
        for (size_t i = 0; i < c->spheres.size(); ++i) {
            const float4 sp = c->spheres[i];
            if (!(std::fabs(sp.x) <= 1e18f && std::fabs(sp.y) <= 1e18f && std::fabs(sp.z) <= 1e18f && std::fabs(sp.w) <= 1e18f))
                return fail(RTX_E_INVALID, "rtx_commit: sphere coordinates must be finite (and below 1e18) when there are more than 16");
            const float c3[3] = {sp.x, sp.y, sp.z};
            const float r = std::fabs(sp.w);
            rmax = std::fmax(rmax, r);
            for (int k = 0; k < 3; ++k) {
                const float e = 1e-6f * (std::fabs(c3[k]) + r) + 1e-30f;
                boxes[i].lo[k] = c3[k] - r - e, boxes[i].hi[k] = c3[k] + r + e;
            }
            all.grow(boxes[i]);
        }
        std::vector<BvhNode> nodes;
        SahBuilder builder(boxes);
        const int depth = builder.build(nodes, kStackDepth);
        if (depth > kStackDepth) return fail(RTX_E_LIMIT, "sphere BVH deeper than the traversal stack");
        c->sphere_bvh_depth = depth;
        BvhNode *dn;
        if (int e = upload(nodes, &dn, c->owned)) return e;
        d.sphere_bvh = dn;
        float h2 = 0.0f;
        for (int k = 0; k < 3; ++k) {
            d.sphere_root_c[k] = 0.5f * (all.lo[k] + all.hi[k]);
            const float h = 0.5f * (all.hi[k] - all.lo[k]);
            h2 += h * h;
        }
        d.sphere_root_h = std::sqrt(h2) * 1.001f;
        d.sphere_rmax = rmax * 1.001f;
    }

    d.lights_plain = 1;
    for (const float4 &l : c->lights)
        if (!(l.w >= 0.0f && l.w <= 3.0e38f)) d.lights_plain = 0;
    const int nt = (int)(c->mesh_idx.size() / 3), nv = (int)(c->mesh_verts.size() / 3);
    d.n_mesh_tris = 0;
    if (nt > 0) {
        // gate box: min/max over all vertices -/+ 1e-4, src/SceneObjects.cpp:222-243
        float lo[3] = {1000, 1000, 1000}, hi[3] = {-1000, -1000, -1000};
        for (int i = 0; i < nv; ++i)
            for (int k = 0; k < 3; ++k) {
                const float v = c->mesh_verts[3 * i + k];
                lo[k] = v < lo[k] ? v : lo[k];  // std::min(min_x, v.x)
                hi[k] = hi[k] < v ? v : hi[k];  // std::max(max_x, v.x)
            }
        const float eps = 0.0001f;
        for (int k = 0; k < 3; ++k) {
            const float a = lo[k] - eps, b = hi[k] + eps;
            d.mesh_lo[k] = a < b ? a : b, d.mesh_hi[k] = a < b ? b : a;  // Cube::load
        }
        float *dv;
        int *di;
        if (int e = upload(c->mesh_verts, &dv, c->owned)) return e;
        if (int e = upload(c->mesh_idx, &di, c->owned)) return e;
        // RTX_OPT_BVH_BUILDER: the tree is built on the GPU -- top-down by the surface-area heuristic unless one of the
        // Morton-order builders is asked for -- or, on request, by the host's surface-area builder, which also
        // steps in when a Morton-order tree comes out deeper than the traversal stack (it bounds its depth).
        LbvhResult r;
        const GpuTree gpu_tree = c->bvh_builder == 2 ? GpuTree::kNone
                                                      : (c->bvh_builder == 1 ? GpuTree::kMorton : (c->bvh_builder == 3 ? GpuTree::kCluster : GpuTree::kSah));
        hipError_t he = lbvhBuild(dv, di, nt, gpu_tree, kStackDepth, c->stream, &r);
        if (he != hipSuccess) return failHip(he, "mesh tree build");
        c->owned.push_back(r.tris), c->owned.push_back(r.nodes);
        if (gpu_tree == GpuTree::kNone || r.max_depth > kStackDepth) {
            std::vector<BvhNode> nodes;
            SahBuilder builder(c->mesh_verts, c->mesh_idx, kBvhPad);
            r.max_depth = builder.build(nodes, kStackDepth);
            HIP_OR_FAIL(hipMemcpy(r.nodes, nodes.data(), nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
        }
        if (r.max_depth > kStackDepth) {
            char msg[160];
            snprintf(msg, sizeof msg, "LBVH depth %d exceeds the traversal stack (%d entries)", r.max_depth, kStackDepth);
            return fail(RTX_E_LIMIT, msg);
        }
        c->bvh_depth = r.max_depth;
        d.mesh_tris = r.tris, d.bvh = r.nodes, d.n_mesh_tris = nt, d.mesh_mat = c->mesh_mat;
        d.mesh_cells = nullptr;
        if (c->mesh_grid) {
            std::vector<int> keys;
            gridCells(c->mesh_verts, c->mesh_idx, d.mesh_lo, d.mesh_hi, d.grid_step, keys);
            int *dk;
            if (int e = upload(keys, &dk, c->owned)) return e;
            d.mesh_cells = reinterpret_cast<const int4 *>(dk);
            // the hit-point shortcut of the mesh walk needs its 1e-3-of-a-cell margin to stay far above
            // float rounding: cells not tiny against the coordinates, hits not too far away
            float min_step = d.grid_step[0], max_abs = 0.0f;
            for (int k = 0; k < 3; ++k) {
                min_step = std::fmin(min_step, d.grid_step[k]);
                max_abs = std::fmax(max_abs, std::fmax(std::fabs(d.mesh_lo[k]), std::fabs(d.mesh_hi[k])));
                d.grid_inv_step[k] = d.grid_step[k] > 0.0f ? 1.0f / d.grid_step[k] : 0.0f;
            }
            d.grid_fast_tmax = (min_step > 0.0f && max_abs < 500.0f * min_step) ? 500.0f * min_step : 0.0f;
        }
    }
    {  // union of everything that is tested against a box, and the rectangle around the islands
        float *u = d.object_bounds;
        for (int k = 0; k < 3; ++k) u[k] = 3.0e38f, u[3 + k] = -3.0e38f;
        auto join = [&](const float *lo, const float *hi) {
            for (int k = 0; k < 3; ++k) u[k] = std::fmin(u[k], lo[k]), u[3 + k] = std::fmax(u[3 + k], hi[k]);
        };
        if (!c->tris.empty()) join(d.tri_bounds, d.tri_bounds + 3);
        if (!c->boxes.empty()) join(d.box_bounds, d.box_bounds + 3);
        if (d.n_mesh_tris > 0) {
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) {
                const float e = 1e-4f * (d.mesh_hi[k] - d.mesh_lo[k]) + 1e-5f * std::fmax(std::fabs(d.mesh_lo[k]), std::fabs(d.mesh_hi[k])) + 1e-6f;
                lo[k] = d.mesh_lo[k] - e, hi[k] = d.mesh_hi[k] + e;
            }
            join(lo, hi);
        }
        for (int k = 0; k < 6; ++k) c->bounded_objects[k] = u[k];  // (what sphereGateBounds joins the spheres' box with per launch)
        if (!c->spheres.empty()) join(d.sphere_bounds, d.sphere_bounds + 3);
        float *r = d.island_rect;
        r[0] = r[2] = 3.0e38f, r[1] = r[3] = -3.0e38f;
        for (const float4 &is : c->islands) {  // cx, cz, a, b: the reference's own rectangle is cx -+ a, cz -+ b
            const float a = std::fabs(is.z), b = std::fabs(is.w);
            const float ex = 1e-4f * a + 1e-5f * std::fabs(is.x) + 1e-6f, ez = 1e-4f * b + 1e-5f * std::fabs(is.y) + 1e-6f;
            r[0] = std::fmin(r[0], is.x - a - ex), r[1] = std::fmax(r[1], is.x + a + ex);
            r[2] = std::fmin(r[2], is.y - b - ez), r[3] = std::fmax(r[3], is.y + b + ez);
        }
    }
    if (int e = residentBlocks(c)) return e;
    if (int e = wavefrontBlocks(c)) return e;
    c->committed = true;
    // host copies are no longer needed
    std::vector<unsigned char>().swap(c->sky);
    std::vector<unsigned char>().swap(c->water);
    return RTX_OK;
}

int rtx_frame_rows(const rtx_frame *f) {
    if (!f || f->row_step <= 0 || f->row_first < 0 || f->row_first >= f->height) return 0;
    return (f->height - f->row_first + f->row_step - 1) / f->row_step;
}

int rtx_render(rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f, unsigned char *bgr_out, rtx_stats *stats) {
    if (int e = checkFrame(c, cam, f)) return e;
    const int rows = rtx_frame_rows(f);
    if (rows > 0 && !bgr_out) return fail(RTX_E_INVALID, "null output buffer");
    if (int e = bind(c)) return e;
    const size_t bytes = (size_t)rows * f->width * 3;
    if (bytes > c->d_frame_bytes) {
        (void)hipFree(c->d_frame);
        c->d_frame = nullptr, c->d_frame_bytes = 0;
        HIP_OR_FAIL(hipMalloc((void **)&c->d_frame, bytes));
        c->d_frame_bytes = bytes;
    }
    int levels = byLevels(c) ? launchWavefront(c, cam, f, c->d_frame, c->stream) : RTX_E_NOMEM;
    if (levels != RTX_OK && (levels != RTX_E_NOMEM || c->pipeline == 1)) return levels;
    if (levels == RTX_E_NOMEM) {  // the one-kernel pipeline: chosen, or no memory for the level-by-level arrays
        if (int e = scheduleBuffers(c, (size_t)((f->width + 7) / 8) * (size_t)((rows + 7) / 8))) return e;
        if (int e = launchRender(c, cam, f, c->d_frame, c->stream, c->sched, c->sched_state, c->feedback != 0)) return e;
    }
    std::string why;
    if (!c->stage.toHost(bgr_out, c->d_frame, bytes, c->stream, &why)) return fail(RTX_E_DEVICE, why);
    if (stats) return rtx_get_stats(c, stats);
    return RTX_OK;
}

int rtx_render_device(rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f, void *d_bgr_out, void *hip_stream) {
    if (int e = checkFrame(c, cam, f)) return e;
    if (rtx_frame_rows(f) > 0 && !d_bgr_out) return fail(RTX_E_INVALID, "null output buffer");
    if (int e = bind(c)) return e;
    const int rows = rtx_frame_rows(f);
    hipStream_t stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    if (byLevels(c)) {
        const int e = launchWavefront(c, cam, f, static_cast<unsigned char *>(d_bgr_out), stream);
        if (e != RTX_E_NOMEM || c->pipeline == 1) return e;  // (no memory for the arrays of even one tile: the one-kernel pipeline)
    }
    if (int e = scheduleBuffers(c, (size_t)((f->width + 7) / 8) * (size_t)((rows + 7) / 8))) return e;
    return launchRender(c, cam, f, static_cast<unsigned char *>(d_bgr_out), stream, c->sched, c->sched_state, c->feedback != 0);
}

int rtx_render_device_noalloc(rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f, void *d_bgr_out, void *hip_stream) {
    if (int e = checkFrame(c, cam, f)) return e;
    const int rows = rtx_frame_rows(f);
    if (rows > 0 && !d_bgr_out) return fail(RTX_E_INVALID, "null output buffer");
    const size_t n_tiles = (size_t)((f->width + 7) / 8) * (size_t)((rows + 7) / 8);
    if (n_tiles > kScratchTiles) {
        char msg[200];
        snprintf(msg, sizeof msg, "rtx_render_device_noalloc: the frame has %zu tiles of 8x8 pixels, the static scheduling memory holds %zu "
                                  "(e.g. 7680x4320); use rtx_render_device", n_tiles, kScratchTiles);
        return fail(RTX_E_LIMIT, msg);
    }
    unsigned char *const base = c->device < kMaxDevices ? static_cast<unsigned char *>(g_scratch_ptr[c->device]) : nullptr;
    if (!base) return fail(RTX_E_STATE, "rtx_render_device_noalloc: no static scheduling memory on this device");
    if (int e = bind(c)) return e;
    hipStream_t stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    SchedMem m;
    m.heads = reinterpret_cast<unsigned *>(base + kScratchHeads);
    m.bins = reinterpret_cast<unsigned *>(base + kScratchBins);
    m.bin_sums = reinterpret_cast<unsigned long long *>(base + kScratchBinSums);
    m.list_hdr = reinterpret_cast<unsigned *>(base + kScratchListHdr);
    m.cost = reinterpret_cast<unsigned *>(base + kScratchCost);
    m.entries = reinterpret_cast<unsigned *>(base + kScratchEntries);
    m.cap = kScratchTiles, m.quarters = false;
    // The array keeps what the last call on this GPU (of any context) left in it.  The queue heads must
    // start at zero; bins, sums, header, costs and list are written before they are read by this call's
    // own launches (k_tile_classes, k_bin_*) -- the tile list is always made afresh from the tile classes.
    HIP_OR_FAIL(hipMemsetAsync(m.heads, 0, kHeadWords * sizeof(unsigned), stream));
    SchedState fresh;
    return launchRender(c, cam, f, static_cast<unsigned char *>(d_bgr_out), stream, m, fresh, false);
}

void *rtx_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void rtx_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int rtx_get_stats(rtx_ctx *c, rtx_stats *o) {
    if (!c || !o) return fail(RTX_E_INVALID, "null argument");
    memset(o, 0, sizeof *o);
    if (!c->timed) return RTX_OK;
    if (int e = bind(c)) return e;
    HIP_OR_FAIL(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_OR_FAIL(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    o->kernel_ms = ms;
    if (c->stats) {
        Counters h;
        HIP_OR_FAIL(hipMemcpy(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
        static_assert(offsetof(rtx_stats, kernel_ms) == sizeof(uint64_t) * CNT_SLOTS, "rtx_stats layout");
        memcpy(o, h.v, sizeof(uint64_t) * CNT_SLOTS);  // slots are in rtx_stats field order
    }
    return RTX_OK;
}

int rtx_deinterleave_rows_device(rtx_ctx *c, const void *d_gathered, int n_ranks, int rows_per_rank, int width,
                                 int height, void *d_frame, void *hip_stream) {
    if (!c || !d_gathered || !d_frame) return fail(RTX_E_INVALID, "null argument");
    if (n_ranks <= 0 || rows_per_rank <= 0 || width <= 0 || height <= 0) return fail(RTX_E_INVALID, "bad sizes");
    if ((long long)rows_per_rank * n_ranks < height) return fail(RTX_E_INVALID, "gathered buffer holds fewer rows than the frame");
    if (int e = bind(c)) return e;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    const size_t total = (size_t)height * width * 3;
    const unsigned grid = (unsigned)std::min<size_t>((total / 4 + 255) / 256 + 1, 256 * 8);
    hipLaunchKernelGGL(k_deinterleave, dim3(grid), dim3(256), 0, st, static_cast<const unsigned char *>(d_gathered), n_ranks,
                       rows_per_rank, width * 3, height, static_cast<unsigned char *>(d_frame));
    HIP_OR_FAIL(hipGetLastError());
    return RTX_OK;
}

int rtx_trace_rays(rtx_ctx *c, const float *orig, const float *dir, int n, int depth, float *rgb_out) {
    if (!c || n < 0 || (n > 0 && (!orig || !dir || !rgb_out))) return fail(RTX_E_INVALID, "rtx_trace_rays: bad arguments");
    if (!c->committed) return fail(RTX_E_STATE, "trace before rtx_commit");
    if (depth < 0 || depth > RTX_MAX_DEPTH) return fail(RTX_E_LIMIT, "depth outside [0, RTX_MAX_DEPTH]");
    if (n == 0) return RTX_OK;
    if (int e = bind(c)) return e;
    float *d_o = nullptr, *d_d = nullptr, *d_c = nullptr;
    const size_t bytes = sizeof(float) * 3 * (size_t)n;
    int rc = RTX_OK;
    hipError_t e;
    if ((e = hipMalloc((void **)&d_o, bytes)) != hipSuccess || (e = hipMalloc((void **)&d_d, bytes)) != hipSuccess ||
        (e = hipMalloc((void **)&d_c, bytes)) != hipSuccess || (e = hipMemcpy(d_o, orig, bytes, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_d, dir, bytes, hipMemcpyHostToDevice)) != hipSuccess) {
        rc = failHip(e, "rtx_trace_rays");
    } else {
        const DevScene s = sceneForLaunch(c, farthestOrigin(c, orig, n));
        if (c->big_spheres || c->long_lists) hipLaunchKernelGGL(k_trace_rays<true>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, s, d_o, d_d, n, depth, d_c);
        else hipLaunchKernelGGL(k_trace_rays<false>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, s, d_o, d_d, n, depth, d_c);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess ||
            (e = hipMemcpy(rgb_out, d_c, bytes, hipMemcpyDeviceToHost)) != hipSuccess)
            rc = failHip(e, "rtx_trace_rays");
    }
    (void)hipFree(d_o), (void)hipFree(d_d), (void)hipFree(d_c);
    return rc;
}

int rtx_intersect_rays(rtx_ctx *c, const float *orig, const float *dir, int n, float *out16) {
    if (!c || n < 0 || (n > 0 && (!orig || !dir || !out16))) return fail(RTX_E_INVALID, "rtx_intersect_rays: bad arguments");
    if (!c->committed) return fail(RTX_E_STATE, "intersect before rtx_commit");
    if (n == 0) return RTX_OK;
    if (int e = bind(c)) return e;
    float *d_o = nullptr, *d_d = nullptr, *d_c = nullptr;
    const size_t bytes = sizeof(float) * 3 * (size_t)n, obytes = sizeof(float) * 16 * (size_t)n;
    int rc = RTX_OK;
    hipError_t e;
    if ((e = hipMalloc((void **)&d_o, bytes)) != hipSuccess || (e = hipMalloc((void **)&d_d, bytes)) != hipSuccess ||
        (e = hipMalloc((void **)&d_c, obytes)) != hipSuccess || (e = hipMemcpy(d_o, orig, bytes, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_d, dir, bytes, hipMemcpyHostToDevice)) != hipSuccess) {
        rc = failHip(e, "rtx_intersect_rays");
    } else {
        const DevScene s = sceneForLaunch(c, farthestOrigin(c, orig, n));
        if (c->big_spheres || c->long_lists) hipLaunchKernelGGL(k_intersect_rays<true>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, s, d_o, d_d, n, d_c);
        else hipLaunchKernelGGL(k_intersect_rays<false>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, s, d_o, d_d, n, d_c);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess ||
            (e = hipMemcpy(out16, d_c, obytes, hipMemcpyDeviceToHost)) != hipSuccess)
            rc = failHip(e, "rtx_intersect_rays");
    }
    (void)hipFree(d_o), (void)hipFree(d_d), (void)hipFree(d_c);
    return rc;
}

}  // extern "C"
