// Device-side helpers shared by the kernels of liblsf_hip.so (gfx950 / CDNA4 only; wave = 64 lanes).
// All float arithmetic here is written so that, compiled with -ffp-contract=off, it follows the
// operation order of oracle/lsf_oracle.py bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "../../include/lsf_hip.h"

namespace lsf {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU
constexpr int kCuBlock = 1024;       // 16 waves: a CU's whole complement at <= 128 VGPRs (see wave_list_walk)
constexpr int kMaxBlockWaves = kCuBlock / kWave;

// n / d for n < 2^31 by multiply-high and shift (Granlund & Montgomery): the tile walk decomposes a tile number per
// tile and per wave, and the compiler's expansion of a 32-bit division by a run-time divisor is ~30 instructions each
struct FastDiv {
    unsigned d, m, l;
};

__host__ inline FastDiv make_fast_div(unsigned d) {
    FastDiv f;
    f.d = d;
    f.l = 0;
    while ((1ull << f.l) < d) ++f.l;
    f.m = (unsigned)(((1ull << 32) * ((1ull << f.l) - d)) / d + 1ull);
    return f;
}

__device__ inline unsigned fast_div(unsigned n, const FastDiv& f) { return (__umulhi(n, f.m) + n) >> f.l; }

struct Grid {
    int nz, ny, nx;
    int z_begin, z_end;
    int z_global_offset;
    unsigned chunk_tiles;  // tiles per scheduling chunk (see tile_walk)
    int tile_y;            // rows per tile = waves per block (1..4); blockDim.x = 64 * tile_y
    long long plane;       // nz*ny*nx: stride between the planes of a planar vector field
    FastDiv div_tiles_x, div_tiles_y, div_chunk, div_nx, div_ny;
    int fast_ok;           // 32-bit buffer addressing is possible: 3 planes of a vector field stay below 4 GiB
    int wide_ok;           // float4 state kernels: a wave's neighbourhoods may be addressed relative to its first lane
    int e_begin, e_end;    // slices whose energies count (lsf_grid::energy_z_begin / _end; default: all)
    unsigned list_group;   // wave-units per group of the fused kernel's list walk (wave_list_walk)
    int list_store_nt;     // list walk: non-temporal stores of the new state (lists too long for the Infinity Cache)
    int shift_x, shift_y;  // log2(nx), log2(ny) when both are powers of two (voxel index -> x, y, z by shifts), else -1
    unsigned index_offset; // linear_index(g, x, y, z) - vidx(g, x, y, z) = z_global_offset * ny * nx
    int gather_nz, gather_z_offset;  // slices / global z of slice 0 of the hierarchical kernel's GATHER operand (the packed
                                     // live field): the grid's own, or a wider / replicated copy (lsf_hier_params::packed_nz)
    int y_global_offset, ny_global;  // slabs cut along y (lsf_grid::y_global_offset / ny_global); y_cut = either differs
    int ey_begin, ey_end;            // rows whose energies count (lsf_grid::energy_y_begin / _end; default: all)
    int y_cut;
};

__host__ inline Grid make_grid(const lsf_grid* g, int tile_y = 4) {
    Grid r;
    r.tile_y = tile_y;
    r.nz = g->nz; r.ny = g->ny; r.nx = g->nx;
    r.z_begin = g->z_begin; r.z_end = g->z_end;
    r.z_global_offset = g->z_global_offset;
    r.plane = (long long)g->nz * g->ny * g->nx;
    // scheduling chunk: a few consecutive z-slices of tiles (3-D) or a few tile rows (2-D), so that the volume is
    // cut into >= ~32 chunks dealt round-robin to the 8 XCDs (load balance when the narrow band is localised)
    // while a chunk is still thick enough that most stencil / halo re-reads stay inside one XCD's L2
    const unsigned tiles_x = (unsigned)(g->nx + 63) / 64, tiles_y = (unsigned)(g->ny + tile_y - 1) / tile_y;
    const unsigned slices = (unsigned)(g->z_end - g->z_begin);
    if (g->dims == 3) {
        unsigned s = slices / 32;
        s = s < 1 ? 1 : (s > 8 ? 8 : s);
        r.chunk_tiles = tiles_x * tiles_y * s;
    } else {
        unsigned rows = tiles_y / 32;
        r.chunk_tiles = tiles_x * (rows < 1 ? 1 : rows);
    }
    r.div_tiles_x = make_fast_div(tiles_x);
    r.div_tiles_y = make_fast_div(tiles_y);
    r.div_chunk = make_fast_div(r.chunk_tiles);
    r.div_nx = make_fast_div((unsigned)g->nx);
    r.div_ny = make_fast_div((unsigned)g->ny);
    r.fast_ok = r.plane * 3 * 4 < 0xffffffffll;
    r.wide_ok = r.fast_ok;
    const bool limited = g->energy_z_end > g->energy_z_begin;
    r.e_begin = limited ? g->energy_z_begin : 0;
    r.e_end = limited ? g->energy_z_end : g->nz;
    r.list_group = 4u;
    r.list_store_nt = 0;
    auto log2_of = [](int v) {
        int k = 0;
        while ((1 << k) < v) ++k;
        return (1 << k) == v ? k : -1;
    };
    r.shift_x = log2_of(g->nx);
    r.shift_y = log2_of(g->ny);
    if (r.shift_x < 0 || r.shift_y < 0) r.shift_x = r.shift_y = -1;
    r.index_offset = (unsigned)((long long)g->z_global_offset * g->ny * g->nx);
    r.gather_nz = g->nz;
    r.gather_z_offset = g->z_global_offset;
    r.y_global_offset = g->y_global_offset;
    r.ny_global = g->ny_global > 0 ? g->ny_global : g->ny;
    const bool y_limited = g->energy_y_end > g->energy_y_begin;
    // (a limited energy row range counts as "cut along y" too: kernels specialised for y_cut == 0 ignore all four fields)
    r.y_cut = r.y_global_offset != 0 || r.ny_global != g->ny || y_limited;
    r.ey_begin = y_limited ? g->energy_y_begin : 0;
    r.ey_end = y_limited ? g->energy_y_end : g->ny;
    return r;
}

__host__ inline bool grid_is_y_cut(const lsf_grid* g) {
    return g->y_global_offset != 0 || (g->ny_global > 0 && g->ny_global != g->ny);
}

__host__ inline int check_grid(const lsf_grid* g, bool allow_y_cut = false) {
    // every entry point validates its grid first: also drop any stale (non-sticky) error an earlier, unrelated
    // runtime call of this thread left behind, so that launch_status() reports THIS launch only
    (void)hipGetLastError();
    if (!g) return LSF_ERR_BAD_ARGUMENT;
    if (g->dims != 2 && g->dims != 3) return LSF_ERR_BAD_DIMS;
    if (g->nx <= 0 || g->ny <= 0 || g->nz <= 0) return LSF_ERR_BAD_ARGUMENT;
    if (g->dims == 2 && g->nz != 1) return LSF_ERR_BAD_DIMS;
    if (g->z_begin < 0 || g->z_end > g->nz || g->z_begin > g->z_end) return LSF_ERR_BAD_ARGUMENT;
    if ((long long)g->nz * g->ny * g->nx > 0x7fffffffll) return LSF_ERR_BAD_DIMS;  // 32-bit voxel indices
    if (g->y_global_offset < 0 || g->ny_global < 0 || (g->ny_global > 0 && g->y_global_offset + g->ny > g->ny_global))
        return LSF_ERR_BAD_ARGUMENT;
    // slabs cut along y: only the entry points of the fused path on band lists know about them
    if (!allow_y_cut && (g->y_global_offset != 0 || (g->ny_global > 0 && g->ny_global != g->ny))) return LSF_ERR_BAD_ARGUMENT;
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// launch geometry: one thread per voxel, x fastest.  Blocks are (64 x 4) tiles in (x, y) of one z slice, so a
// wave reads 256 contiguous bytes per field row.  The grid is 1-D over tiles; consecutive block ids walk x
// tiles, then y tiles, then z -- neighbouring blocks (which share halo rows) therefore land on different XCDs
// under round-robin dispatch, so remap block ids so that each XCD owns a contiguous chunk of z-slices (T1).
// ------------------------------------------------------------------------------------------------------
constexpr int kTileX = 64;
constexpr int kTileY = 4;

struct Tiling {
    int tiles_x, tiles_y, tiles_z;
    unsigned total;
};

__host__ __device__ inline Tiling make_tiling(const Grid& g) {
    Tiling t;
    t.tiles_x = (g.nx + kTileX - 1) / kTileX;
    t.tiles_y = (g.ny + g.tile_y - 1) / g.tile_y;
    t.tiles_z = g.z_end - g.z_begin;
    t.total = (unsigned)t.tiles_x * t.tiles_y * t.tiles_z;
    return t;
}

// Persistent blocks: a launch uses at most kMaxBlocks workgroups (8 per CU); each walks its share of the tile
// sequence.  One reduction + one atomic per block and address instead of one per tile: 65536 same-address atomics
// (256^3 / 256-voxel tiles) serialise at ~12 ns each, which alone cost 0.8 ms per launch.
// XCD-aware order (T1): blocks b, b+8, b+16 ... share an XCD under round-robin dispatch, so XCD k owns the k-th
// contiguous chunk of the tile sequence (a contiguous z-range) and its blocks walk that chunk side by side;
// neighbouring tiles -- which share halo rows -- then hit the same L2.  Speed only, never correctness.
constexpr unsigned kMaxBlocks = 2048;
constexpr unsigned kXcds = 8;

// 251 blocks per XCD (prime): a block's stride through its XCD's tile sequence must not divide the number of tiles
// in a z-slice (256 at 256^2), or every block would revisit the same (x, y) tile position in every slice and the
// blocks whose position lies in the narrow band would do all the work (measured: 0.21 ms vs 0.13 ms at 256^3).
constexpr unsigned kBlocksPerXcd = 251;

// persistent-grid size of the tile walks (measured 64 ... 256 blocks per XCD in rounds 1-2: DESIGN.md section 7; a
// variant build -- tools/build_variant.sh NAME - -DLSF_BLOCKS_PER_XCD=n -- overrides it for measurements)
#ifndef LSF_BLOCKS_PER_XCD
#define LSF_BLOCKS_PER_XCD 0
#endif
__host__ inline unsigned blocks_per_xcd(unsigned fallback = kBlocksPerXcd) {
    return LSF_BLOCKS_PER_XCD > 0 ? (unsigned)LSF_BLOCKS_PER_XCD : fallback;
}

__host__ inline unsigned launch_blocks(unsigned total_tiles, unsigned per_xcd = kBlocksPerXcd) {
    const unsigned full = kXcds * per_xcd;
    if (total_tiles >= full) return full;
    unsigned b = total_tiles;
    if (b > kXcds) b -= b % kXcds;
    return b;
}

// Tiles are numbered x fastest, then y, then z.  The sequence is cut into chunks of g.chunk_tiles tiles; chunk c
// belongs to XCD (c mod 8) (blocks b, b+8, ... share an XCD under round-robin dispatch: MI355X_MICROARCH.md), and
// the blocks of one XCD walk that XCD's chunks side by side.
struct TileWalk {
    unsigned q, count, step, xcd, chunk;  // XCD-local sequence position / length / stride
    FastDiv div_chunk;
    bool plain;                            // tiny grids: plain grid-stride over [0, count)
};

__device__ inline TileWalk tile_walk(unsigned total, const Grid& g) {
    TileWalk w;
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    const unsigned chunk = g.chunk_tiles;
    w.chunk = chunk;
    w.div_chunk = g.div_chunk;
    if (nb % kXcds != 0) {
        w.plain = true; w.q = bid; w.count = total; w.step = nb; w.xcd = 0;
        return w;
    }
    w.plain = false;
    w.xcd = bid % kXcds;
    w.q = bid / kXcds;
    w.step = nb / kXcds;
    const unsigned cycle = kXcds * chunk;
    const unsigned rem = total % cycle;
    const unsigned lo = w.xcd * chunk;
    const unsigned extra = rem > lo ? (rem - lo < chunk ? rem - lo : chunk) : 0u;
    w.count = (total / cycle) * chunk + extra;
    return w;
}

__device__ inline unsigned tile_of(const TileWalk& w, unsigned q) {
    if (w.plain) return q;
    const unsigned m = fast_div(q, w.div_chunk), within = q - m * w.chunk;
    return (w.xcd + kXcds * m) * w.chunk + within;
}

// calls f(x, y, z) for every voxel this thread owns (z in [z_begin, z_end))
template <class F>
__device__ inline void for_each_voxel(const Grid& g, F&& f) {
    const Tiling t = make_tiling(g);
    const TileWalk w = tile_walk(t.total, g);
    const int lx = threadIdx.x & (kTileX - 1), ly = threadIdx.x / kTileX;
    for (unsigned q = w.q; q < w.count; q += w.step) {
        const unsigned tile = tile_of(w, q);
        const unsigned rest = fast_div(tile, g.div_tiles_x);
        const int tx = (int)(tile - rest * (unsigned)t.tiles_x);
        const int tz = (int)fast_div(rest, g.div_tiles_y);
        const int ty = (int)rest - tz * t.tiles_y;
        const int x = tx * kTileX + lx, y = ty * g.tile_y + ly, z = g.z_begin + tz;
        if (x < g.nx && y < g.ny) f(x, y, z);
    }
}

// Walk over a BAND LIST (lsf_band_list_fill): `count` voxel indices (z * ny + y) * nx + x in ascending order.  Work
// unit = 256 consecutive entries = one block step; entries are short x-runs of consecutive rows, so the four waves of
// a block share stencil rows through L1 much like a tile does.  XCD k owns the k-th eighth of the units (a contiguous
// z-range -> its own L2), and the blocks of an XCD interleave over that eighth.  The host sizes the grid so that all
// blocks get the same number of units (band_list_blocks).  Calls f(x, y, z) for every listed voxel.
// list == nullptr: the dense tile walk of for_each_voxel (one loop for both so that f is instantiated once).
// the units [first, end) with stride `step` this block owns in a list walk over `count` entries
struct ListWalk {
    unsigned first, step, end;
};

__device__ inline ListWalk list_walk(unsigned count) {
    ListWalk w;
    const unsigned units = (count + kBlock - 1) / kBlock;
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    if (nb % kXcds == 0) {
        const unsigned per_xcd = (units + kXcds - 1) / kXcds, xcd = bid % kXcds;
        w.first = xcd * per_xcd + bid / kXcds;
        w.step = nb / kXcds;
        w.end = (xcd + 1) * per_xcd < units ? (xcd + 1) * per_xcd : units;
    } else {
        w.first = bid; w.step = nb; w.end = units;
    }
    return w;
}

// The list walk of the fused state kernel: ONE 1024-thread workgroup per CU.  A workgroup of 256 threads at ~126 VGPRs
// leaves room for four per CU, and a grid of "as many as fit" (928 for the 256^3 sphere pair: every block the same
// seven 256-entry units) is dealt 4 to some CUs and 3 to others -- measured with in-kernel clocks
// (tools/state_trace.py): the waves of the full CUs took 32 us, the others 18-21 us, and the launch ends with the
// slowest.  A CU-sized workgroup cannot be co-scheduled with another, so each of the 256 CUs gets exactly one.
// Work is numbered per workgroup: sequence number n = 0, 1, 2 ... -> wave-unit (64 consecutive list entries)
//   x_begin + ((n / G) * cus + q) * G + n % G      (G = Grid::list_group, q = the workgroup's rank in its XCD)
// i.e. XCD k owns the k-th eighth of the list (a contiguous z-range, its own L2), and the `cus` workgroups of an XCD
// sweep through that eighth TOGETHER in groups of G wave-units dealt round-robin: what one CU reads as the z -/+ 1
// rows of its voxels, a neighbouring CU reads or writes as centre rows at about the same time, so those rows are
// served by the XCD's 4 MB L2 instead of the memory side (at 512^3 one slice of band voxels is ~0.5 MB of state; 32 CUs
// on 32 separate z-ranges cycle through 50 MB).  The G wave-units of a group are consecutive list entries (short
// x-runs of adjacent rows) whose stencils overlap in the CU's L1.  The waves of a workgroup take sequence numbers
// as they come free (see the kernel): the SIMD issues oldest-wave-first, so a static deal leaves the last waves behind.
struct WaveWalk {
    unsigned x_begin, x_end, q, cus, group;  // wave-units of 64 entries
    __device__ inline unsigned unit(unsigned n) const {
        const unsigned g = n / group;
        return x_begin + (g * cus + q) * group + (n - g * group);
    }
};

__device__ inline WaveWalk wave_list_walk(unsigned count, unsigned group) {
    const unsigned units = (count + kWave - 1) / kWave;
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    WaveWalk w;
    w.group = group;
    if (nb % kXcds == 0) {
        const unsigned per_xcd = (units + kXcds - 1) / kXcds, xcd = bid % kXcds;
        w.cus = nb / kXcds;
        w.q = bid / kXcds;
        w.x_begin = xcd * per_xcd;
        w.x_end = w.x_begin + per_xcd < units ? w.x_begin + per_xcd : units;
    } else {
        w.cus = nb;
        w.q = bid;
        w.x_begin = 0u;
        w.x_end = units;
    }
    return w;
}

// grid of a wave_list_walk: one kCuBlock workgroup per CU, fewer for short lists (a wave-unit per wave at least)
__host__ inline unsigned cu_list_blocks(unsigned count, unsigned cus) {
    const unsigned units = (count + kWave - 1) / kWave;
    unsigned blocks = (units + kMaxBlockWaves - 1) / kMaxBlockWaves;
    if (blocks < 1) blocks = 1;
    if (blocks > cus) blocks = cus;
    if (blocks > kXcds) blocks -= blocks % kXcds;
    return blocks;
}

template <class F>
__device__ inline void for_each_listed_voxel(const Grid& g, const int* __restrict__ list, unsigned count, F&& f) {
    const bool listed = list != nullptr;
    const Tiling t = make_tiling(g);
    const TileWalk w = tile_walk(t.total, g);
    const int lx = threadIdx.x & (kTileX - 1), ly = threadIdx.x / kTileX;
    unsigned first = w.q, step = w.step, end = w.count;
    if (listed) {
        const unsigned units = (count + kBlock - 1) / kBlock;
        const unsigned nb = gridDim.x, bid = blockIdx.x;
        if (nb % kXcds == 0) {
            const unsigned per_xcd = (units + kXcds - 1) / kXcds, xcd = bid % kXcds;
            first = xcd * per_xcd + bid / kXcds;
            step = nb / kXcds;
            end = (xcd + 1) * per_xcd < units ? (xcd + 1) * per_xcd : units;
        } else {
            first = bid; step = nb; end = units;
        }
    }
    for (unsigned u = first; u < end; u += step) {
        int x, y, z;
        bool inside;
        if (listed) {
            const unsigned k = u * kBlock + threadIdx.x;
            inside = k < count;
            const unsigned i = inside ? (unsigned)list[k] : 0u;
            const unsigned zy = fast_div(i, g.div_nx);
            x = (int)(i - zy * (unsigned)g.nx);
            z = (int)fast_div(zy, g.div_ny);
            y = (int)zy - z * g.ny;
        } else {
            const unsigned tile = tile_of(w, u);
            const unsigned rest = fast_div(tile, g.div_tiles_x);
            const int tx = (int)(tile - rest * (unsigned)t.tiles_x);
            const int tz = (int)fast_div(rest, g.div_tiles_y);
            const int ty = (int)rest - tz * t.tiles_y;
            x = tx * kTileX + lx; y = ty * g.tile_y + ly; z = g.z_begin + tz;
            inside = x < g.nx && y < g.ny;
        }
        if (inside) f(x, y, z);
    }
}

// grid size for a list walk: <= 8 * kBlocksPerXcd-ish persistent blocks, every block the same number of units
__host__ inline unsigned band_list_blocks(unsigned count, unsigned default_per_xcd = 256u) {
    const unsigned units = (count + kBlock - 1) / kBlock;
    if (units <= kXcds) return units < 1 ? 1 : units;
    const unsigned per_xcd = (units + kXcds - 1) / kXcds;
    const unsigned cap = default_per_xcd;
    const unsigned rounds = (per_xcd + cap - 1) / cap;
    return kXcds * ((per_xcd + rounds - 1) / rounds);
}

// (A dynamic variant -- per-XCD work queues with atomic heads and stealing -- was measured and rejected: device-scope
// returning atomics on a contended address serialise at ~140 ns here, 16 K grabs per 256^3 launch cost 0.29 ms.)

// voxel index -> (x, y, z): shifts and masks when nx and ny are powers of two (BASELINE's 256^3 / 512^3: four full-rate
// instructions), else two multiply-high divisions (quarter rate on CDNA: ~22 issue slots); wave-uniform choice
__device__ inline void decode_voxel(const Grid& g, unsigned i, int& x, int& y, int& z) {
    if (g.shift_x >= 0) {
        x = (int)(i & ((1u << g.shift_x) - 1u));
        const unsigned zy = i >> g.shift_x;
        y = (int)(zy & ((1u << g.shift_y) - 1u));
        z = (int)(zy >> g.shift_y);
    } else {
        const unsigned zy = fast_div(i, g.div_nx);
        x = (int)(i - zy * (unsigned)g.nx);
        z = (int)fast_div(zy, g.div_ny);
        y = (int)zy - z * g.ny;
    }
}

__device__ inline unsigned linear_index(const Grid& g, int x, int y, int z) {
    return (unsigned)(((long long)(z + g.z_global_offset) * g.ny_global + (y + g.y_global_offset)) * g.nx + x);
}

// voxel index inside one plane: 32-bit on purpose (check_grid caps a plane at 2^31-1 voxels) so that loads take the
// "scalar base + 32-bit offset" form instead of 64-bit VALU address arithmetic
__device__ inline int vidx(const Grid& g, int x, int y, int z) {
    return (z * g.ny + y) * g.nx + x;
}

__device__ inline bool inside(const Grid& g, int x, int y, int z) {
    return (unsigned)x < (unsigned)g.nx && (unsigned)y < (unsigned)g.ny && (unsigned)z < (unsigned)g.nz;
}

// scalar read with a constant for out-of-bounds taps (utils/sampling.py:35-55)
__device__ inline float read_oob(const float* __restrict__ f, const Grid& g, int x, int y, int z, float oob) {
    return inside(g, x, y, z) ? f[vidx(g, x, y, z)] : oob;
}

// clamp-to-edge read (edge replication)
__device__ inline float read_clamp(const float* __restrict__ f, const Grid& g, int x, int y, int z) {
    x = min(max(x, 0), g.nx - 1);
    y = min(max(y, 0), g.ny - 1);
    z = min(max(z, 0), g.nz - 1);
    return f[vidx(g, x, y, z)];
}

// ------------------------------------------------------------------------------------------------------
// D-linear interpolation of a scalar field (oracle.sample_linear): per-tap OOB constant, lerp z, then y, then x.
// pz is the GLOBAL z position (z + z_global_offset + w): float32 rounding of "coordinate + displacement" depends
// on the coordinate's magnitude, so a slab must form it from the global coordinate to match the whole volume.
// ------------------------------------------------------------------------------------------------------
// per-axis tap addressing for the D-linear gathers: both taps of an axis are read from CLAMPED (always valid)
// coordinates and an out-of-bounds tap is replaced by select afterwards -- no exec-mask branches around loads
struct AxisTaps {
    int c0, c1;    // clamped coordinates of the two taps
    bool v0, v1;   // tap inside the array?
    float r, i;    // ratio and 1 - ratio
};

__device__ inline AxisTaps axis_taps(float p, int n, int global_offset) {
    AxisTaps t;
    const float f = floorf(p);
    t.r = p - f;
    t.i = 1.0f - t.r;
    // clamp the base so the int conversion is defined for wild warps; anything beyond is OOB anyway
    const int b = (int)fminf(fmaxf(f - (float)global_offset, -2.0f), (float)n + 1.0f);
    t.v0 = (unsigned)b < (unsigned)n;
    t.v1 = (unsigned)(b + 1) < (unsigned)n;
    t.c0 = min(max(b, 0), n - 1);
    t.c1 = min(max(b + 1, 0), n - 1);
    return t;
}

template <int D>
__device__ inline float sample_linear_taps(const float* __restrict__ f, const Grid& g, const AxisTaps& ax,
                                           const AxisTaps& ay, const AxisTaps& az, float oob) {
    const int row0 = ay.c0 * g.nx, row1 = ay.c1 * g.nx;
    if (D == 2) {
        float v00 = f[row0 + ax.c0], v01 = f[row1 + ax.c0], v10 = f[row0 + ax.c1], v11 = f[row1 + ax.c1];
        v00 = (ax.v0 && ay.v0) ? v00 : oob;
        v01 = (ax.v0 && ay.v1) ? v01 : oob;
        v10 = (ax.v1 && ay.v0) ? v10 : oob;
        v11 = (ax.v1 && ay.v1) ? v11 : oob;
        const float i0 = v00 * ay.i + v01 * ay.r;
        const float i1 = v10 * ay.i + v11 * ay.r;
        return i0 * ax.i + i1 * ax.r;
    } else {
        const int slice = g.nx * g.ny;
        const int s0 = az.c0 * slice, s1 = az.c1 * slice;
        float c[2][2];  // [x offset][y offset] after the z lerp
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy) {
                const int xy = (oy ? row1 : row0) + (ox ? ax.c1 : ax.c0);
                const bool vxy = (ox ? ax.v1 : ax.v0) && (oy ? ay.v1 : ay.v0);
                float a = f[s0 + xy], b = f[s1 + xy];
                a = (vxy && az.v0) ? a : oob;
                b = (vxy && az.v1) ? b : oob;
                c[ox][oy] = a * az.i + b * az.r;
            }
        const float i0 = c[0][0] * ay.i + c[0][1] * ay.r;
        const float i1 = c[1][0] * ay.i + c[1][1] * ay.r;
        return i0 * ax.i + i1 * ax.r;
    }
}

template <int D>
__device__ inline float sample_linear(const float* __restrict__ f, const Grid& g, float px, float py, float pz,
                                      float oob) {
    const AxisTaps ax = axis_taps(px, g.nx, 0), ay = axis_taps(py, g.ny, 0);
    AxisTaps az = ax;
    if (D == 3) az = axis_taps(pz, g.nz, g.z_global_offset);
    return sample_linear_taps<D>(f, g, ax, ay, az, oob);
}

// ------------------------------------------------------------------------------------------------------
// block reductions: 64-lane shuffles -> LDS across the 4 waves -> one atomic per block
// ------------------------------------------------------------------------------------------------------
__device__ inline unsigned long long shfl_down_u64(unsigned long long v, int delta) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_down(lo, delta, kWave);
    hi = __shfl_down(hi, delta, kWave);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ inline double shfl_down_f64(double v, int delta) {
    return __longlong_as_double((long long)shfl_down_u64((unsigned long long)__double_as_longlong(v), delta));
}

// Wave-wide reductions by DPP (data-parallel primitives: a VALU move that reads another lane's register, no LDS round
// trip): xor-1, xor-2 inside quads, rotate by 4 and 8 inside rows of 16, then the last lane of a row into the next row and
// lane 31 into rows 2 and 3 -- six steps, the result in lane 63, handed to every lane by a readlane.  Lanes without a
// source (row 0 of row_bcast:15, rows 0-1 of row_bcast:31) receive the identity.  ~85 VALU instructions for one packed
// maximum + three float64 sums where six ds_bpermute steps each took ~1 us at the end of every wave of the fused kernel.
// ALL 64 lanes must be active.
template <int CTRL>
__device__ inline unsigned long long dpp_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ inline unsigned long long broadcast_lane63_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// maximum over the wave, in EVERY lane
__device__ inline unsigned long long wave_max_u64(unsigned long long v) {
#define LSF_STEP(CTRL)                                   \
    {                                                    \
        const unsigned long long o = dpp_u64<CTRL>(v);   \
        v = o > v ? o : v;                               \
    }
    LSF_STEP(0xb1)   // quad_perm:[1,0,3,2]
    LSF_STEP(0x4e)   // quad_perm:[2,3,0,1]
    LSF_STEP(0x124)  // row_ror:4
    LSF_STEP(0x128)  // row_ror:8
    LSF_STEP(0x142)  // row_bcast:15
    LSF_STEP(0x143)  // row_bcast:31
#undef LSF_STEP
    return broadcast_lane63_u64(v);
}

// sum over the wave, in EVERY lane (a fixed association, the same in every launch)
__device__ inline double wave_sum_f64(double v) {
#define LSF_STEP(CTRL)                                                                                               \
    v += __longlong_as_double((long long)dpp_u64<CTRL>((unsigned long long)__double_as_longlong(v)));
    LSF_STEP(0xb1)
    LSF_STEP(0x4e)
    LSF_STEP(0x124)
    LSF_STEP(0x128)
    LSF_STEP(0x142)
    LSF_STEP(0x143)
#undef LSF_STEP
    return __longlong_as_double((long long)broadcast_lane63_u64((unsigned long long)__double_as_longlong(v)));
}

__device__ inline unsigned long long pack_max(float length, unsigned linear_index) {
    return ((unsigned long long)__float_as_uint(length) << 32) | (unsigned long long)(~linear_index);
}

__device__ inline float unpack_max_value(unsigned long long packed) { return __uint_as_float((unsigned)(packed >> 32)); }

// Block-wide: max of `packed` -> atomicMax(dst_max); sums of up to NS doubles -> atomicAdd(dst_sum[i]).
// Must be called by every thread of the block.
template <int NS>
__device__ inline void block_reduce_commit(unsigned long long packed, const double (&sums)[NS > 0 ? NS : 1],
                                           unsigned long long* dst_max, double* const (&dst_sum)[NS > 0 ? NS : 1]) {
    __shared__ unsigned long long s_max[kMaxBlockWaves];
    __shared__ double s_sum[(NS > 0 ? NS : 1)][kMaxBlockWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int n_waves = blockDim.x / kWave;
    unsigned long long m = wave_max_u64(packed);
    double s[NS > 0 ? NS : 1];
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] = wave_sum_f64(sums[i]);
    if (lane == 0) {
        s_max[wave] = m;
#pragma unroll
        for (int i = 0; i < NS; ++i) s_sum[i][wave] = s[i];
    }
    __syncthreads();
    if (wave == 0) {
        // one wave's partial results per lane, reduced the same way (a serial loop over the LDS words on one lane took
        // ~2 us at the end of every CU-sized workgroup)
        const unsigned long long mm = wave_max_u64(lane < n_waves ? s_max[lane] : 0ull);
        double t[NS > 0 ? NS : 1];
#pragma unroll
        for (int i = 0; i < NS; ++i) t[i] = wave_sum_f64(lane < n_waves ? s_sum[i][lane] : 0.0);
        if (lane == 0) {
            if (dst_max && mm != 0ull) atomicMax(dst_max, mm);
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (dst_sum[i] && t[i] != 0.0) atomicAdd(dst_sum[i], t[i]);
        }
    }
}

// taps of one separable-filter pass, by value in the kernel arguments
template <int NT>
struct TapsN {
    double k[NT];
};

// acc + k * v in float64.  When every tap is a float32 value (the reference's Sobolev kernels are:
// generate_1d_sobolev_kernel(..., precision=np.float32)) the product of a tap and a float32 sample is EXACT in float64
// -- 24 + 24 significant bits -- so the fused multiply-add rounds once exactly where the separate add does: the same
// bits with half the float64 instructions (these filters are bound by float64 issue, not by memory).  The launchers
// pick FMA only then (taps_are_float32); arbitrary float64 taps keep the two-instruction form.
template <bool FMA>
__device__ inline double mac(double acc, double k, double v) {
    return FMA ? __builtin_fma(k, v, acc) : acc + k * v;
}

static inline bool taps_are_float32(const double* taps, int n) {
    for (int j = 0; j < n; ++j)
        if (!((double)(float)taps[j] == taps[j])) return false;  // also false for NaN
    return true;
}

// vector length as np.linalg.norm(axis=-1) evaluates it in float32: sqrt((a^2 + b^2) [+ c^2])
template <int D>
__device__ inline float vec_length(const float (&v)[3]) {
    float s = v[0] * v[0] + v[1] * v[1];
    if (D == 3) s = s + v[2] * v[2];
    return sqrtf(s);
}

// scipy.ndimage.laplace on one axis: float32( -2*a0 + (ap + am) ) evaluated in double.  -2 * a0 is exact, so the fused
// multiply-add rounds exactly where the separate add would: the same bits, one float64 instruction fewer.
__device__ inline float second_difference_f64(float am, float a0, float ap) {
    return (float)__builtin_fma(-2.0, (double)a0, (double)ap + (double)am);
}

// device-side convergence gate (see lsf_gate in include/lsf_hip.h) ------------------------------------------
__device__ inline bool gate_closed(const lsf_gate& gate) {
    if (!gate.prev_record || gate.mode == LSF_GATE_OPEN) return false;
    unsigned long long p = 0ull;
#pragma unroll
    for (int k = 0; k < LSF_RECORD_SLOTS; ++k) {
        const unsigned long long q = gate.prev_record->slot[k].max_packed;
        p = q > p ? q : p;
    }
    if (p == 0ull) return true;  // previous iteration was itself a no-op
    float m = unpack_max_value(p);
    if (gate.mode == LSF_GATE_HIERARCHICAL) return m < gate.a;
    return !(gate.a < m && m < gate.b);
}

// the partial record this block accumulates into (see lsf_iteration_record)
__device__ inline lsf_record_slot* record_slot(lsf_iteration_record* r) {
    return &r->slot[blockIdx.x % LSF_RECORD_SLOTS];
}

__device__ inline unsigned long long* record_max(lsf_iteration_record* r) {
    return reinterpret_cast<unsigned long long*>(&record_slot(r)->max_packed);
}

__host__ inline lsf_gate gate_or_open(const lsf_gate* gate) {
    return gate ? *gate : lsf_gate{nullptr, 0, 0.0f, 0.0f};
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace lsf
