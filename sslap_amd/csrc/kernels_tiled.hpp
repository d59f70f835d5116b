// kernels_tiled.hpp -- the full-scan engine: the bid kernel (and the eCE / objective pass) with object prices tiled in LDS.
//
// Why: in the wave-per-row kernel (k_bid) every edge gathers an 8-byte price from a table of M x 8 bytes; the table lives
// in L2 but every random 8-byte read moves a whole sector L2 -> L1, ~8x the bytes of the edge stream, and that traffic --
// not HBM -- bounds the kernel (measured: 259 us with the gather, 106 us without, C3).  Here a workgroup owns one CU's
// LDS, holds TWO tiles of kTileColsHalf = 10 112 prices (79 KB each, double-buffered: three loader wavefronts fill the
// next tile by LDS-DMA while thirteen compute wavefronts look prices up in the current one with ds_read_b64) and walks
// the column tiles of its persons once.
//
// Second layout of the edges in HBM ("tile-major", built once at ingest, next to the row-major CSR; HBM capacity is not
// a constraint): persons are cut into layout blocks of kTileRB = 128; inside a block the edges are ordered by (column
// tile, person, stored order).  The segment of (person i, tile t) is [S[idx], S[idx+1]) with idx = ((i / RB) * T + t) *
// RB + i % RB (a 4-byte table entry per segment: start | odd-length flag).  Edges are PACKED, two per record, as a
// 16-bit price slot (the index of the edge's price inside the kernel's LDS buffers, tile parity included: a look-up is
// `slot << 3`, no column arithmetic) plus the value -- the record formats (template parameter kFmt):
//     0   6 B/edge  {u16 slot x 2, f32 value x 2}                     fp32-exact values, column-sorted rows
//     1  10 B/edge  {u16 slot x 2, f64 value x 2}                     arbitrary doubles, column-sorted rows
//     2   8 B/edge  {u16 slot x 2, u16 stored index x 2, f32 x 2}     rows whose columns are NOT ascending
//     3  12 B/edge  {u16 slot x 2, u16 stored index x 2, f64 x 2}     ... with arbitrary doubles
// The in-row tie rule ("the last stored index attaining the maximum wins", auction_.pyx:351) needs no data in formats
// 0 / 1: with ascending columns the tile-major position of an edge is monotone in its stored index, so a lane meets its
// elements in stored order and ">=" is the rule.  Formats 2 / 3 carry every edge's stored index within its row (rows of
// at most 65 536 edges) and compare it where two values are equal; the running "position of the best" is then the
// stored index itself.
//
// Work split: a 1024-thread workgroup = 13 compute wavefronts cut into lane groups of kGL = 4, 8 or 16 lanes (chosen at
// create from the average (person, tile) segment length: C3 10 edges -> 4, C4 20 -> 8, C2 50 -> 16) + 3 loader
// wavefronts.  A group owns up to kTileRows = 4 persons and keeps, PER LANE, the running top-2 of the elements that
// lane has seen (registers); the lanes of a group are merged once, after the last tile, with DPP all-reduces inside
// the group.  The edges of a segment beyond what the software-pipelined loads of a step cover (2 x kGL x kTileDepth)
// sit on per-person overflow lists that are read once, after the last tile.
#pragma once
#include <type_traits>

#include "device_common.hpp"
#include "kernels_check.hpp"
#include "kernels_round.hpp"

// (non-temporal edge loads were measured twice, on the 8- and the 6-byte layout: 5-20 % slower, removed)

namespace misslap {

// prices per LDS tile (a multiple of 128 = one 1-KB LDS-DMA piece):
constexpr int kTileColsBig = 157 * 128;   // 20096 -> 160768 B, one workgroup per CU owns (almost) all of its LDS
constexpr int kTileColsHalf = 79 * 128;   // 10112 ->  80896 B, two workgroups per CU: one computes while the
                                          //                     other waits for its tile fill
constexpr int kTileRB = 128;      // persons per layout block
constexpr int kTilePadRecords = 64;  // zeroed two-edge records behind the tile-major copy (see host_create.hpp)
// LDS behind the price buffers: statistics scratch of the epilogue (16 x 12 + 16 bytes), then ONE 256-byte scratch region
// that every loader wavefront's L2 touches land in (an LDS-DMA load writes one dword slot per lane, LDS base + 4 * lane,
// whatever its element size: 256 bytes per wavefront; the data is junk and never read, so the loaders share the region)
constexpr int kTileStatBytes = 16 * 12 + 16;
constexpr int kTileTouchBytes = 64 * 4;
static_assert((2 * (79 * 128) + 128) * 8 + kTileStatBytes + kTileTouchBytes <= 160 * 1024, "k_bid_tiled: price buffers + scratch exceed a CU's LDS");
// Launch shapes (template parameters of k_bid_tiled): THREADS per workgroup (one workgroup per CU: the
// price tile takes 128 of the 160 KB of LDS), ROWS persons per 8-lane group (register-resident running
// top-2 per lane), BATCH persons whose segment loads are in flight together.

// Record formats of the tile-major copy (see the header): two edges per record.
//   byte 0: u16 slot x 2;  formats 2 / 3: byte 4: u16 stored index (within the row) x 2;  then the two values.
template <int kFmt>
struct TileFmt {
    static_assert(kFmt >= 0 && kFmt <= 3, "tile-major record formats 0..3");
    static constexpr bool kF64 = (kFmt & 1) != 0;  // values as fp64 (arbitrary doubles) instead of fp32 bits
    static constexpr bool kG = kFmt >= 2;          // stored index carried (rows whose columns are not ascending)
    static constexpr int kValOff = kG ? 8 : 4;     // byte offset of the first value
    static constexpr int kRec = kValOff + (kF64 ? 16 : 8);  // bytes per record: 12, 20, 16, 24
    typedef typename std::conditional<kF64, double, int>::type VT;  // a value as held in a register
    static __device__ __forceinline__ double val(VT v) {
        if constexpr (kF64) return v;
        else return (double)__int_as_float(v);  // exact widening
    }
};
__host__ __device__ constexpr int tile_rec_bytes(int fmt) { return (fmt >= 2 ? 8 : 4) + ((fmt & 1) ? 16 : 8); }
// entry q of a tile-major copy in format `fmt`: its value (as two dwords; the high one is 0 for fp32), its stored index
__device__ __forceinline__ void tile_entry(const unsigned *tpk, int fmt, int q, unsigned &vlo, unsigned &vhi, int &g) {
    const unsigned *rec = tpk + (size_t)(q >> 1) * (tile_rec_bytes(fmt) / 4);
    const int vo = fmt >= 2 ? 2 : 1;
    if (fmt & 1) {
        vlo = rec[vo + 2 * (q & 1)];
        vhi = rec[vo + 2 * (q & 1) + 1];
    } else {
        vlo = rec[vo + (q & 1)];
        vhi = 0u;
    }
    g = fmt >= 2 ? (int)((rec[1] >> (16 * (q & 1))) & 0xffffu) : q;
}

// index of segment (person, tile) in the pointer table; the host enables the tiled path only while the
// table has < 2^31 entries, so 32-bit arithmetic is enough (and saves address registers in the kernel)
__host__ __device__ __forceinline__ int tile_idx(int person, int t, int T, int rb = kTileRB) {
    return ((person / rb) * T + t) * rb + (person % rb);
}

// ---- ingest: tile-major copy of the edges ---------------------------------------------------------------
// pass 1, one wavefront per person: L(i, t) = number of edges of row i with column < t * kTileCols
// (binary search, one tile boundary per lane); cnt[idx(i,t)] = L(i,t+1) - L(i,t); also the
// column-order check (*unsorted: bit 0 = some row's columns do not ascend, bit 1 = some row repeats a column).
// cnt = the count rounded up to an even number (every segment starts 16-byte aligned, so that a lane can
// take two edges with one dwordx4 load); len = the real count.
// (the columns of the row-major CSR: `cols[cs * g]` -- cs = 2 for the interleaved 8 B/edge layout, 1 for the 12 B/edge one)
__global__ __launch_bounds__(256) void k_tile_count(const int *cols, int cs, const int *row_ptr, int n_rows, int T,
                                                    int kTileCols, int rb, int *cnt, int *len, int *lrel,
                                                    int *unsorted) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + wave; i < n_rows; i += gridDim.x * 4) {
        const int s = row_ptr[i], e = row_ptr[i + 1];
        int bad = 0;  // bit 0: a column below its predecessor; bit 1: a column equal to it (an entry stored more than once)
        for (int g = s + 1 + lane; g < e; g += kWave) {
            const int c = cols[(size_t)cs * g], cp = cols[(size_t)cs * (g - 1)];
            bad |= (c < cp ? 1 : 0) | (c == cp ? 2 : 0);
        }
        for (int off = 32; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off);
        if (bad && lane == 0) atomicOr(unsorted, bad);
        for (int t0 = 0; t0 <= T; t0 += kWave) {
            const int t = t0 + lane;
            int lo = 0;
            if (t <= T) {
                if (t == T) {
                    lo = e - s;
                } else {
                    const int bound = t * kTileCols;  // first position with col >= bound
                    int hi = e - s;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (cols[(size_t)cs * (s + mid)] < bound) lo = mid + 1;
                        else hi = mid;
                    }
                }
            }
            const int nxt = __shfl_down(lo, 1);
            int next_lo = nxt;
            if (lane == 63 && t < T) {  // boundary handled by the next 64-tile pass: recompute
                const int bound = (t + 1) * kTileCols;
                int l2 = 0, hi = e - s;
                if (t + 1 == T) l2 = e - s;
                else {
                    while (l2 < hi) {
                        const int mid = (l2 + hi) >> 1;
                        if (cols[(size_t)cs * (s + mid)] < bound) l2 = mid + 1;
                        else hi = mid;
                    }
                }
                next_lo = l2;
            }
            if (t < T) {
                const int idx = tile_idx(i, t, T, rb);
                cnt[idx] = (next_lo - lo + 1) & ~1;
                len[idx] = next_lo - lo;
                lrel[idx] = lo;
            }
        }
    }
}

// The same counts for rows whose columns are NOT ascending (formats 2 / 3): one thread per edge, an atomic per edge on
// the segment's length (`len` zeroed by the caller); k_tile_even then forms the padded counts.  No binary search, no
// order requirement: the tile-major copy only needs the edges GROUPED by tile, in stored order inside a segment.
__global__ __launch_bounds__(256) void k_tile_count_any(const int *cols, int cs, const int *row_ptr, int n_rows, int T,
                                                        int kTileCols, int rb, int *len) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + wave; i < n_rows; i += gridDim.x * 4) {
        const int s = row_ptr[i], e = row_ptr[i + 1];
        for (int g = s + lane; g < e; g += kWave) atomicAdd(&len[tile_idx(i, cols[(size_t)cs * g] / kTileCols, T, rb)], 1);
    }
}
__global__ __launch_bounds__(256) void k_tile_even(const int *len, long long n, int *cnt) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) cnt[k] = (len[k] + 1) & ~1;
}

// exclusive scan of a long int array in three launches (chunk sums, scan of the sums, rescan + offset)
constexpr int kScanChunk = 4096;  // elements per 1024-thread block
__global__ __launch_bounds__(1024) void k_scan_sums(const int *in, long long n, int *sums) {
    __shared__ int s_w[16];
    const long long base = (long long)blockIdx.x * kScanChunk;
    int v = 0;
    for (int q = 0; q < 4; ++q) {
        const long long k = base + q * 1024 + threadIdx.x;
        if (k < n) v += in[k];
    }
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += s_w[w];
        sums[blockIdx.x] = t;
    }
}
__device__ __forceinline__ void scan_of_sums_body(int *sums, int nblocks) {  // in place, exclusive; one 1024-thread workgroup
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + t;
        const int v = (i < nblocks) ? sums[i] : 0;
        int x = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        int wpre = 0;
        for (int w2 = 0; w2 < wave; ++w2) wpre += s_w[w2];
        const int carry = s_carry;
        if (i < nblocks) sums[i] = carry + wpre + x - v;
        __syncthreads();
        if (t == 1023) s_carry = carry + wpre + x;
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_scan_of_sums(int *sums, int nblocks) { scan_of_sums_body(sums, nblocks); }
__global__ __launch_bounds__(1024) void k_scan_apply(const int *in, long long n, const int *sums, int *out) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long base = (long long)blockIdx.x * kScanChunk;
    if (t == 0) s_carry = sums[blockIdx.x];
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const long long k = base + q * 1024 + t;
        const int v = (k < n) ? in[k] : 0;
        int x = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        int wpre = 0;
        for (int w2 = 0; w2 < wave; ++w2) wpre += s_w[w2];
        const int carry = s_carry;
        if (k < n) out[k] = carry + wpre + x - v;
        if (k == n - 1) out[n] = carry + wpre + x;  // one-past-the-end entry = grand total
        __syncthreads();
        if (t == 1023) s_carry = carry + wpre + x;
        __syncthreads();
    }
}

// k_bid_tiled's segment table, 4 B per segment: start | (real length is odd).  Segments are laid out in table
// order and padded to an even length, so the next entry's start gives the padded length; n + 1 entries.
__global__ __launch_bounds__(256) void k_pack_seg4(const int *start, const int *len, long long n, int *seg4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += stride)
        seg4[k] = start[k] | (k < n ? (len[k] & 1) : 0);
}

// Overflow lists.  The tile loop of k_bid_tiled covers the first `cap` = 2 x lanes x depth edges of a (person, tile)
// segment with its software-pipelined loads.  A longer segment (one in 37 at C3) used to be finished in a loop with a
// load inside -- one exposed memory latency, behind the prefetches of the next step, in 58 % of all steps of a
// wavefront and 40 % of its cycles (stamped build).  Now the edges beyond `cap` are listed per person -- their
// tile-major positions, in stored order -- and looked at once, after the last tile, with prices from memory.
__global__ __launch_bounds__(256) void k_ovf_count(const int *len, int n_rows, int T, int rb, int cap, int *ovf_cnt) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) {
        int c = 0;
        for (int t = 0; t < T; ++t) c += max(len[tile_idx(i, t, T, rb)] - cap, 0);
        ovf_cnt[i] = c;
    }
}
// An entry is self-contained -- {tile-major position, column, fp32 bits of the value, 0} -- so that the epilogue reads
// its list with ONE sequential 16-byte load per edge and goes to memory once more, for the price (two random 4-byte
// reads per edge into the tile-major arrays pulled a 128-byte line each: + 25 MB per full scan at C3).  Runs after
// k_tile_scatter (it copies from the tile-major arrays).
// (formats: .x = the tile-major position in formats 0 / 1, the stored index within the row in formats 2 / 3 -- whatever
// orders equal values in the kernel; .z / .w = the value's dwords, .w = 0 for fp32 values)
__global__ __launch_bounds__(256) void k_ovf_fill(const int *len, const int *start, int n_rows, int T, int rb, int cap,
                                                  const int *ovf_ptr, const unsigned *tpk, const int *tcol, int4 *ovf, int fmt) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) {
        int at = ovf_ptr[i];
        for (int t = 0; t < T; ++t) {
            const int idx = tile_idx(i, t, T, rb);
            const int n = len[idx], s0 = start[idx];
            for (int k = cap; k < n; ++k) {
                const int q = s0 + k;
                unsigned vlo, vhi;
                int g;
                tile_entry(tpk, fmt, q, vlo, vhi, g);
                ovf[at++] = make_int4(g, tcol[q], (int)vlo, (int)vhi);
            }
        }
    }
}

// pass 3: copy every edge to its tile-major position, PACKED two edges per record in format kFmt (TileFmt above): the
// 16-bit slot = (col - tile * cols) + (tile & 1) * buf_stride is the index of the edge's price inside the kernel's LDS
// buffers (a look-up is `slot << 3`, no column arithmetic); the real column goes to the parallel array `tcol` (read by
// the ingest of the overflow lists and by the column-split shape).
// Formats 0 / 1 (ascending columns): the position inside the segment is the stored index minus the first stored index
// of the tile (lrel, from the binary search of k_tile_count).  Formats 2 / 3 (any column order): the wavefront walks the
// row in stored order, 64 edges at a time, and ranks the edges of each tile present in the chunk behind the running
// fill of that segment (`fill`, zeroed by the caller; one wavefront owns a row and walks it in order).
template <class E, int kFmt>
__global__ __launch_bounds__(256) void k_tile_scatter(E ed, const int *row_ptr, int n_rows, int T, int kTileCols, int rb,
                                                      const int *start, const int *lrel, int *fill, unsigned *tpk,
                                                      int *tcol, int buf_stride) {
    typedef TileFmt<kFmt> F;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + wave; i < n_rows; i += gridDim.x * 4) {
        const int s = row_ptr[i], e = row_ptr[i + 1];
        for (int g0 = s; g0 < e; g0 += kWave) {
            const int g = g0 + lane;
            const bool ok = g < e;
            int col = 0;
            double val = 0.0;
            ed.load(ok ? g : e - 1, col, val);
            const int t = col / kTileCols;
            const int idx = tile_idx(i, t, T, rb);
            int pos;
            if (!F::kG) {
                pos = start[idx] + (g - s - lrel[idx]);
            } else {
                pos = 0;
                unsigned long long todo = __ballot(ok);
                while (todo) {  // wave-uniform: one pass per tile present among the chunk's edges
                    const int src = __ffsll((long long)todo) - 1;
                    const int t0 = __builtin_amdgcn_readlane(t, src);
                    const int idx0 = __builtin_amdgcn_readlane(idx, src);
                    const unsigned long long m = __ballot(ok && t == t0);
                    int base = 0;  // (an atomic: performed at L2, so the next chunk's read of the same counter sees it)
                    if (lane == src) base = atomicAdd(&fill[idx0], __popcll(m));
                    base = __builtin_amdgcn_readlane(base, src);
                    if (ok && t == t0) pos = start[idx0] + base + __popcll(m & lanemask_lt());
                    todo &= ~m;
                }
            }
            if (ok) {
                tcol[pos] = col;
                const int slot = (col - t * kTileCols) + (t & 1) * buf_stride;
                unsigned *rec = tpk + (size_t)(pos >> 1) * (F::kRec / 4);
                reinterpret_cast<unsigned short *>(rec)[pos & 1] = (unsigned short)slot;
                if (F::kG) reinterpret_cast<unsigned short *>(rec)[2 + (pos & 1)] = (unsigned short)(g - s);
                if (F::kF64) {
                    rec[F::kValOff / 4 + 2 * (pos & 1)] = (unsigned)__double2loint(val);
                    rec[F::kValOff / 4 + 2 * (pos & 1) + 1] = (unsigned)__double2hiint(val);
                } else {
                    rec[F::kValOff / 4 + (pos & 1)] = (unsigned)__float_as_int((float)val);  // exact: the layout is chosen for fp32-exact values
                }
            }
        }
    }
}

// ---- bidders in person order (partial rounds) -----------------------------------------------------------------
// The unassigned list U is in the reference's order, which every round shuffles (evicted owners inherit the slots of
// the winners, push_all_left moves persons from the end into the holes).  In a round with K < N adjacent lane groups
// would then read segments scattered all over the tile-major copy, every 60-byte segment pulling its own 128-byte
// lines: PMC showed up to 521 MB of reads for such a round at C3 -- more than the 337 MB of a FULL scan.  So the
// bidders of a partial round are taken in person order: rank of person i = number of unassigned persons below i
// (p2o[i] == -1), and the bid is stored at the person's true list position (the tie rule of :379 is about list
// positions).  Two small launches ahead of the bid kernel (four until round 4): k_order_prepare -- the inverse of U and
// the chunk counts, which do not depend on each other -- and k_order_scatter, whose workgroups add up the counts of the
// chunks in front of theirs themselves (at most a few hundred) instead of waiting for a scan launch.
// (thr / min_K: the scan these kernels prepare runs only in a live round with K >= min_K -- otherwise nothing to do)
__device__ __forceinline__ bool order_needed(const Ctl *ctl, int thr, int min_K) { return round_live(ctl, thr) && ctl->K >= min_K; }
__device__ __forceinline__ void k_order_prepare_body(const Ctl *ctl, const int *U, int *pos_of, const int *p2o, int n_rows,
                                                        int nchunks, int *sums, int thr, int min_K) {
    if (!order_needed(ctl, thr, min_K)) return;
    const int K = ctl->K;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < K; n += gridDim.x * blockDim.x) pos_of[U[n]] = n;
    if ((int)blockIdx.x >= nchunks) return;  // (uniform over the workgroup; the grid covers the list AND the chunks)
    __shared__ int s_w[16];
    const int base = blockIdx.x * kScanChunk;
    int v = 0;
    for (int q = 0; q < 4; ++q) {
        const int k = base + q * 1024 + threadIdx.x;
        v += (k < n_rows && p2o[k] == -1);
    }
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += s_w[w];
        sums[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_order_prepare(const Ctl *ctl, const int *U, int *pos_of, const int *p2o, int n_rows,
                                                        int nchunks, int *sums, int thr, int min_K) { k_order_prepare_body(ctl, U, pos_of, p2o, n_rows, nchunks, sums, thr, min_K); }
struct F_k_order_prepare {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(const Ctl *ctl, const int *U, int *pos_of, const int *p2o, int n_rows, int nchunks, int *sums, int thr, int min_K) { k_order_prepare_body(ctl, U, pos_of, p2o, n_rows, nchunks, sums, thr, min_K); }
};

__device__ __forceinline__ void k_order_scatter_body(const Ctl *ctl, const int *p2o, int n_rows, const int *sums, const int *pos_of,
                                                        int *order_person, int *order_pos, int thr, int min_K) {
    if (!order_needed(ctl, thr, min_K)) return;
    __shared__ int s_w[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // unassigned persons in the chunks before this one: the chunk counts of k_order_prepare, added up here
    int carry = 0;
    {
        int v = 0;
        for (int b = t; b < (int)blockIdx.x; b += 1024) v += sums[b];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) s_w[wave] = v;
        __syncthreads();
        for (int w = 0; w < 16; ++w) carry += s_w[w];
    }
    for (int q = 0; q < 4; ++q) {
        const int i = blockIdx.x * kScanChunk + q * 1024 + t;
        const bool un = i < n_rows && p2o[i] == -1;
        const unsigned long long b = __ballot(un);
        __syncthreads();  // s_w of the previous pass is no longer read
        if (lane == 0) s_w[wave] = __popcll(b);
        __syncthreads();
        int wpre = 0, tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) {
            if (w2 < wave) wpre += s_w[w2];
            tot += s_w[w2];
        }
        if (un) {
            const int r = carry + wpre + __popcll(b & lanemask_lt());
            order_person[r] = i;
            order_pos[r] = pos_of[i];
        }
        carry += tot;
    }
}
__global__ __launch_bounds__(1024) void k_order_scatter(const Ctl *ctl, const int *p2o, int n_rows, const int *sums, const int *pos_of,
                                                        int *order_person, int *order_pos, int thr, int min_K) { k_order_scatter_body(ctl, p2o, n_rows, sums, pos_of, order_person, order_pos, thr, min_K); }
struct F_k_order_scatter {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(const Ctl *ctl, const int *p2o, int n_rows, const int *sums, const int *pos_of, int *order_person, int *order_pos, int thr, int min_K) { k_order_scatter_body(ctl, p2o, n_rows, sums, pos_of, order_person, order_pos, thr, min_K); }
};


// ---- the kernel -------------------------------------------------------------------------------------------
__device__ __forceinline__ double group8_max_f64(double v) {
    {
        const double o = dpp_f64<kDppXor1>(v);
        v = o > v ? o : v;
    }
    {
        const double o = dpp_f64<kDppXor2>(v);
        v = o > v ? o : v;
    }
    {
        const double o = dpp_f64<kDppHalfMirror>(v);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int group8_max_i32(int v) {
    {
        const int o = dpp_i32<kDppXor1>(v);
        v = o > v ? o : v;
    }
    {
        const int o = dpp_i32<kDppXor2>(v);
        v = o > v ? o : v;
    }
    {
        const int o = dpp_i32<kDppHalfMirror>(v);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double group4_max_f64(double v) {
    v = __builtin_fmax(v, dpp_f64<kDppXor1>(v));
    v = __builtin_fmax(v, dpp_f64<kDppXor2>(v));
    return v;
}
__device__ __forceinline__ int group4_max_i32(int v) {
    {
        const int o = dpp_i32<kDppXor1>(v);
        v = o > v ? o : v;
    }
    {
        const int o = dpp_i32<kDppXor2>(v);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double group16_max_f64(double v) {
    v = __builtin_fmax(v, dppf_f64<kDppXor1>(v));
    v = __builtin_fmax(v, dppf_f64<kDppXor2>(v));
    v = __builtin_fmax(v, dppf_f64<kDppHalfMirror>(v));
    v = __builtin_fmax(v, dppf_f64<kDppMirror>(v));
    return v;
}
__device__ __forceinline__ int group16_max_i32(int v) {
    v = max(v, dppf_i32<kDppXor1>(v));
    v = max(v, dppf_i32<kDppXor2>(v));
    v = max(v, dppf_i32<kDppHalfMirror>(v));
    v = max(v, dppf_i32<kDppMirror>(v));
    return v;
}
template <int kGL>
__device__ __forceinline__ int group_sum_i32(int v) {  // sum over the kGL lanes of a group, in every lane
    v += dpp_i32<kDppXor1>(v);
    v += dpp_i32<kDppXor2>(v);
    if (kGL >= 8) v += dpp_i32<kDppHalfMirror>(v);
    if (kGL >= 16) v += dpp_i32<kDppMirror>(v);
    return v;
}
template <int kGL>
__device__ __forceinline__ double group_max_f64(double v) {
    return kGL == 4 ? group4_max_f64(v) : kGL == 8 ? group8_max_f64(v) : group16_max_f64(v);
}
template <int kGL>
__device__ __forceinline__ int group_max_i32(int v) {
    return kGL == 4 ? group4_max_i32(v) : kGL == 8 ? group8_max_i32(v) : group16_max_i32(v);
}

struct TiledArgs {
    const unsigned *tpk; // tile-major edges, packed 6 B/edge (see k_tile_scatter); every segment starts at an even
                         // position, i.e. on a 12-byte record
    const int *tcol;     // real column of every tile-major entry
    const int *seg4;     // start | odd-length flag per (person block, tile, person), n_blocks * T * RB + 1 entries
    int T;               // number of column tiles
    int min_K;           // the kernel runs only for K >= min_K (k_bid takes the smaller rounds)
    int nnz;             // entries of the tile-major copy incl. padding (leftover loads are clamped below it)
    const int *order_person;  // bidders in person order and their list positions (partial rounds, see k_order_*);
    const int *order_pos;     // nullptr: list order (full scans: U is the identity)
    const int *ovf_ptr;       // [n_rows + 1] overflow edges of a person: ovf_q[ovf_ptr[i] .. ovf_ptr[i + 1])
    const int4 *ovf;          // ... {tile-major position, column, value bits, 0} (k_ovf_fill); built for ovf_cap edges per segment
    int ovf_cap;              // must equal 2 * lanes per person * loads per segment of the launch shape
    // column-split shapes (kCS > 1): per-(share of the tiles, bidder slot) partial top-2, merged by the workgroup of a
    // slice that finishes last
    double2 *part_vw;         // [kCS][part_stride] {best value, second-best value}
    int *part_g;              // [kCS][part_stride] tile-major position of the best edge (-1: no edge in that share)
    int part_stride;
    int *split_cnt;           // [slices] workgroups of the slice that have published their partial result (0 between launches)
    FinalOut fo;              // MODE 1 (the eCE / objective / validity pass, kernels_check.hpp): where its results go
};

// hand-over of a column-split shape's partial results between workgroups (see the epilogue of k_bid_tiled)
template <class T>
__device__ __forceinline__ void split_store(T *p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ T split_load(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All global loads of the tile loop are UNCONDITIONAL (masked-off lanes read a clamped, valid address and
// their value is neutralised): a load inside an `if` makes hipcc wait for it (vmcnt(0)) before the branch
// re-converges, which serialises every load of the kernel.  The price array is padded to a whole number
// of tiles so that the LDS fill needs no bounds test either.
//
// LDS: two price tiles (double buffer).  While the workgroup looks prices up in tile t, LDS-DMA
// (global_load_lds_dwordx4, no VGPRs) fills tile t+1 into the other buffer; one barrier per tile.  Slot
// kTileCols of each buffer holds +inf: a masked-off element reads it, gets value -inf and changes nothing.
//
// Per element (auction_.pyx:350-358 without branches):  v = cost - price;
//   w = max(w, min(v, best));  best' = max(best, v);  g = (v >= best) ? position : g
// which is the reference's update (">=": a later equal value becomes the best and the old best the second).
//
// Template: THREADS per workgroup, ROWS persons per 8-lane group, BATCH persons whose loads are in flight
// together, DEPTH unconditional 8-lane loads per segment (longer segments finish in a short loop),
// TILE_COLS prices per LDS tile.  ABL (diagnostics only, results wrong): 1 = no LDS fill, 2 = no
// per-element arithmetic, 3 = no edge loads, 4 = no barriers in the tile loop,
// 5 = price look-ups without the arithmetic, 6 = arithmetic without the look-ups.
// kCS = 2 (column split): two workgroups share a slice of bidders and each sees HALF of the column tiles, i.e. half of
// the price table -- per CU the tile fills (the whole table once per workgroup: 1.6 MB at C3, more than the 1.0 MB of
// edges a CU streams) halve, and so do the barriers; a lane group then owns twice the persons (kTileRows = 8).  The
// partial top-2s of a bidder go to memory (20 bytes each); the workgroup of the slice that arrives last forms the bids.
//
// MODE 1: the same engine as the pass over ALL rows that eCE_satisfied (auction_.pyx:443-485), get_obj (:489-523) and
// the validity flags need (kernels_check.hpp): instead of a running top-2 a lane keeps the running maximum of
// val - price and the last stored edge whose column is the person's object -- the column is known before the scan, so
// inside the tile that holds it the edge is recognised by its 16-bit price slot, with no column arithmetic per
// element.  No bids, no statistics; the result of a person is handled by final_person().  a.eps = the eps of the test.
// kFmt: the record format of the tile-major copy (TileFmt; the host launches the instance of the handle's format).
#ifndef MISSLAP_TILED_KEYCOL
#define MISSLAP_TILED_KEYCOL 1  // (0: positions in every format -- A/B builds)
#endif
#define MISSLAP_TILED_KEYCOL_ON (MISSLAP_TILED_KEYCOL != 0)
template <int kTileThreads, int kTileRows, int kTileBatch, int kTileDepth, int kTileCols, int kLoaders, int ABL = 0, int kGL = 4, int kCS = 1, int MODE = 0, int kFmt = 0, int kRev = 0>
__global__ __launch_bounds__(kTileThreads) void k_bid_tiled(RoundArgs a, TiledArgs ta) {
    typedef TileFmt<kFmt> F;
    typedef typename F::VT VT;
    // kRev = 1: the column tiles are walked from the last to the first.  Two passes over the tile-major copy in the SAME
    // direction meet the Infinity Cache in LRU order (a scan's 257 MB against 256 MB of cache: what the first pass touched
    // first is what its own tail has just evicted); a pass in the opposite direction reads first what the previous one
    // touched last (back to back, C3: 73 -> 64 us).  The partial rounds of an eps-phase follow its full scan directly, so
    // the host alternates the direction from one engine launch of a phase to the next.  A lane of a format-0 / 1 scan
    // then meets its tiles in DESCENDING column order while ">=" orders equal values in ascending order: a step whose
    // best merely TIES the best of the tiles walked before it (larger columns: later stored) is undone (`tie_only`).
    static_assert(kRev == 0 || (MODE == 0 && kCS == 1 && ABL == 0 && !F::kG && kTileCols != kTileColsBig && kLoaders > 0 && MISSLAP_TILED_KEYCOL_ON),
                  "the backward walk: column-keyed bid scans (formats 0 / 1) of the unsplit double-buffered shapes");
    static_assert(kFmt == 0 || (kCS == 1 && ABL == 0), "column split and ablations exist for the 6 B/edge format only");
    static_assert(kCS == 1 || kCS == 2 || kCS == 4, "column split: none, halves or quarters of the tiles");
    static_assert(MODE == 0 || (MODE == 1 && kCS == 1 && ABL == 0), "the check pass runs on the unsplit shapes");
    static_assert(kGL == 4 || kGL == 8 || kGL == 16, "lanes per person: 4, 8 or 16 (one DPP row at most)");
    static_assert(kGL * kTileDepth <= kTilePadRecords, "masked-off lanes read at most that far behind the copy");
    constexpr int kTileGroups = (kTileThreads - kLoaders * kWave) / kGL;  // kGL-lane groups; loader wavefronts own none
    // LDS (in doubles): buffer 0 at [0, kTileCols), the +inf slot at kTileCols, buffer 1 at [kBufDoubles, ...).
    // A price slot must fit the 16-bit field of a packed edge: 2 * kTileCols + 128 < 65536.
    constexpr int kBufDoubles = kTileCols + 128;
    // kTileColsBig: ONE buffer (fill, barrier, look up, barrier); kTileColsHalf: two buffers, fill overlapped
    constexpr bool kDouble = kTileCols != kTileColsBig;
    static_assert(kDouble || kLoaders == 0, "the single-buffer variant has no loader wavefronts");
    static_assert(kTileRows % kTileBatch == 0 && kTileRows / kTileBatch >= 2, "ROWS = BATCH * (>= 2 steps)");
    extern __shared__ __attribute__((aligned(16))) double s_price[];  // 2 * kBufDoubles
    const Ctl *ctl = a.ctl;
    int lo, hi;
    if (MODE == 0) {
        if (!round_live(ctl, a.thr) || ctl->K < ta.min_K) return;
        shard_range(ctl->K, a.rank, a.world, a.shard_min_K, lo, hi);
    } else {  // every person; an eCE-only pass has nothing left to find once any row has failed (the sample pass)
        if (!ta.fo.fin && ctl->ece_fail) return;
        lo = 0;
        hi = a.n_rows;
    }
    // this workgroup's slice of list positions (and, with a column split, its half of the tiles)
    const int n_slices = (int)gridDim.x / kCS, slice = (int)blockIdx.x / kCS, half = kCS > 1 ? (int)blockIdx.x % kCS : 0;
    const int per_wg = (hi - lo + n_slices - 1) / n_slices;
    const int p0 = lo + slice * per_wg;
    const int p1 = min(hi, p0 + per_wg);
    if (p0 >= p1) return;  // uniform over the workgroup
    const int t = threadIdx.x, lane = t & 63, gl = lane & (kGL - 1);
    const int group = t / kGL;
    const double eps = (double)a.eps;
    const double ninf = -__builtin_huge_val();
    const int T = ta.T;
    const int t_lo = kCS > 1 ? min(T, half * ((T + kCS - 1) / kCS)) : 0;
    const int t_hi = kCS > 1 ? min(T, t_lo + (T + kCS - 1) / kCS) : T;
    // The LDS-DMA pieces of a tile and a wavefront's own loads share one in-order counter (vmcnt): data of a
    // load issued AFTER a piece cannot be consumed before the piece has landed.  Two ways around it:
    //   kLoaders > 0: the last kLoaders wavefronts do nothing but the fills (wavefront specialisation);
    //   kLoaders = 0: every wavefront issues its share of the pieces right after the prefetch loads of the
    //                 tile's first step, so that only loads of the tile's later steps queue behind them.
    constexpr int kWaves = kTileThreads / kWave;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool loader = kLoaders > 0 && wave_u >= kWaves - kLoaders;
    const bool full = MODE == 1 || (ctl->K == a.n_rows && !ta.order_person);  // uniform
    int person[kTileRows];
    double sv[kTileRows], sw[kTileRows];
    // Formats 0 / 1 hold rows whose columns ascend STRICTLY (the ingest sends rows with a repeated column to the
    // stored-index formats): the column of an edge is then as good a key for the tie rule as its position -- unique
    // within the row and monotone in the stored index -- and it travels with the best element anyway (scol).  The tile
    // loop of a bid scan therefore does not track positions at all (two instructions per element less); sg takes the
    // columns over behind the loop, and the overflow entries and the merges order equal values by it as before.
    constexpr bool kKeyCol = MISSLAP_TILED_KEYCOL && MODE == 0 && kCS == 1 && !F::kG && ABL == 0;  // (the ablations carry no column)
    int sg[kTileRows];  // position of the lane's best element (kKeyCol: its column, from behind the tile loop on) ...
    // ... and its column and cost.  They are NOT updated per element (two more selects in the inner loop): after a step
    // that moved sg the winner is picked from the step's registers (see `note_best`).  Re-reading them at the end through
    // the position -- two random 4-byte reads per bidder into the tile-major arrays -- pulled 51 MB of lines per full
    // scan at C3 and put a memory latency in front of the bids.
    int scol[kTileRows];
    VT scost[kTileRows];
    // MODE 1 keeps in the same registers: sv = running maximum, sg = position of the last match (-1: none), scost = its
    // value bits; and per person the tile of the wanted column, its price slot there as an LDS byte offset, the number
    // of matches
    int wtile[kTileRows], wslot8[kTileRows], mcnt[kTileRows];
#pragma unroll
    for (int j = 0; j < kTileRows; ++j) {
        const int pos = p0 + j * kTileGroups + group;
        // (a full scan opens an eps-phase: K = N and the list is the identity, k_reset_phase -- one memory latency less
        // in front of the first segment loads)
        const int u = full ? pos : (ta.order_person ? ta.order_person : a.U)[min(pos, p1 - 1)];  // unconditional load (see the note above), masked afterwards
        person[j] = (pos < p1 && !loader) ? u : -1;
        sv[j] = ninf;
        sw[j] = ninf;
        sg[j] = -1;
        scol[j] = kKeyCol ? -1 : 0;  // (kKeyCol: "no element yet", like sg)
        scost[j] = VT(0);
        wtile[j] = -1;
        wslot8[j] = 0;
        mcnt[j] = 0;
    }
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) {
            const int c = wanted_column(a.p2o[max(person[j], 0)], a.n_cols);
            const int cc = max(c, 0), tj = cc / kTileCols;
            wtile[j] = (c >= 0 && person[j] >= 0) ? tj : -1;
            wslot8[j] = ((cc - tj * kTileCols) + (kDouble ? (tj & 1) * kBufDoubles : 0)) << 3;  // as k_tile_scatter forms it
        }
    }
    if (t == 0) s_price[kTileCols] = __builtin_huge_val();
    // an edge carries the slot of its price inside s_price (tile parity included, see k_tile_scatter)
    constexpr int kInfOff = kTileCols * 8;  // the +inf slot (behind buffer 0)
    constexpr int kSlotShift = 3;           // slot -> LDS byte address
    // ... as an ABSOLUTE LDS address: s_price is the kernel's only LDS object and therefore starts at LDS address 0
    // (checked below); going through the s_price symbol would cost a v_add of its link-time address per look-up
    typedef const __attribute__((address_space(3))) double *lds_cdp;
    if (t == 0 && (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void *)s_price != 0u)
        atomicOr(&a.ctl->err, kErrLdsBase);
    auto lds_price = [&](int off) { return *(lds_cdp)(__UINTPTR_TYPE__)(unsigned)off; };

    // pieces first, first + stride, ... of `tile` -> buffer tile & 1; one piece = 64 lanes x 16 B = 128 prices
    auto dma_fill = [&](int tile, int first, int stride) {
        constexpr int kPieces = kTileCols / 128;
        const double *gsrc = a.price + (size_t)tile * kTileCols + 2 * lane;
        double *dst = s_price + (kDouble ? (tile & 1) : 0) * kBufDoubles;
#pragma unroll 4
        for (int piece = first; piece < kPieces; piece += stride)
            if (ABL != 1 && !(ABL == 7 && (piece & 1)))  // (7: every other piece -- what half the fill bytes would be worth)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + piece * 128),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 128), 16, 0,
                                                 0);
    };
    auto rt = [&](int u) {  // step u of the walk -> its tile (steps past the end repeat the last one: prefetches only)
        const int uc = min(u, t_hi - 1);
        return kRev ? t_lo + t_hi - 1 - uc : uc;
    };
    if (loader) {
        const int me = wave_u - (kWaves - kLoaders);
#ifndef MISSLAP_TILED_TOUCH
#define MISSLAP_TILED_TOUCH 1  // tiles of lead of the L2 prefetch below (0 = off; C3: all launches of a solve 1.350 -> 1.324 ms)
#endif
        // L2 prefetch of the price tiles (speed only, never correctness).  All workgroups of an XCD fill the same tile
        // at about the same time, prices are rewritten between the launches, so every tile fill begins with first-touch
        // misses of that XCD's L2 -- one memory latency per tile that the double buffer (one tile of lead) does not
        // cover when a tile's compute is short (partial rounds).  A loader wavefront therefore touches its share of
        // the lines of tile t + MISSLAP_TILED_TOUCH (one 4-byte load per 128-byte line, 1 / 32 of the tile per
        // workgroup of the XCD under round-robin dispatch) behind the fill of tile t + 1; the fill waits with
        // vmcnt(1) -- loads return in order, the touch is the youngest and may stay in flight.
        constexpr int kTileLines = kTileCols * 8 / 128;
        const int nx = max(1, (int)gridDim.x / 8), xr = ((int)blockIdx.x / 8) % nx;
        const int lines_per_wg = (kTileLines + nx - 1) / nx, lines_per_wave = (lines_per_wg + kLoaders - 1) / (kLoaders > 0 ? kLoaders : 1);
        // (always ONE load per call, all lanes active, addresses clamped into the table: the vmcnt(1) below relies on
        // exactly one load behind the pieces of a fill.  The touch is an LDS-DMA of one byte per lane into the 256 bytes
        // of scratch behind the statistics words (kTileTouchBytes, shared by all loaders): a load WITHOUT a register
        // destination -- an asynchronous inline-asm load into a VGPR would land whenever it lands, in a register the
        // compiler may have re-used by then)
        char *touch_dst = reinterpret_cast<char *>(s_price + (kDouble ? kBufDoubles + kTileCols : kTileCols + 2)) + kTileStatBytes;
        auto touch = [&](int tile) {
            const int line = min(xr * lines_per_wg + me * lines_per_wave + min(lane, lines_per_wave - 1), kTileLines - 1);
            const char *src = reinterpret_cast<const char *>(a.price + (size_t)min(tile, T - 1) * kTileCols) + (size_t)line * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)touch_dst, 1, 0, 0);
        };
        if (t_lo < t_hi) dma_fill(rt(t_lo), me, kLoaders);
        if (MISSLAP_TILED_TOUCH > 0) touch(rt(t_lo + 1));
        for (int tile = t_lo; tile < t_hi; ++tile) {
            // my pieces of tile `tile` have landed (a touch issued behind them may still be in flight)
            if (MISSLAP_TILED_TOUCH > 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ABL != 4) __syncthreads();                    // ... and tile - 1 is no longer read
            if (tile + 1 < t_hi) dma_fill(rt(tile + 1), me, kLoaders);
            if (MISSLAP_TILED_TOUCH > 0) touch(rt(tile + 1 + MISSLAP_TILED_TOUCH));
        }
        if (MISSLAP_TILED_TOUCH > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // Software pipeline over steps = (tile, batch of kTileBatch persons): while step s is consumed, the edges
    // of step s+1 and the segment pointers of step s+2 are in flight; they do not depend on LDS.
    constexpr int kNB = kTileRows / kTileBatch;
#ifndef MISSLAP_TILED_PF
#define MISSLAP_TILED_PF 1  // steps of edge loads in flight ahead of the one being consumed
#endif
    // The two raw table entries of a (person, tile) segment.  They are decoded (seg_s0 / seg_s1) only where a step
    // USES them, steps after the load: decoded at the load site, the `& ~1` made every step wait, in its middle, for
    // loads it had issued at its top (ISA: s_waitcnt vmcnt(3) / (2) behind the look-ups).
    struct Seg {
        int x[kTileBatch], y[kTileBatch];
    };
    auto seg_s0 = [](const Seg &g, int jj) { return g.x[jj] & ~1; };
    auto seg_s1 = [](const Seg &g, int jj) { return (g.y[jj] & ~1) - (g.x[jj] & 1); };  // start + padded length - pad
    struct Edges {
        unsigned c[kTileBatch][kTileDepth];   // two 16-bit price slots
        unsigned gg[kTileBatch][F::kG ? kTileDepth : 1];  // (formats 2 / 3) two 16-bit stored indices
        VT v0[kTileBatch][kTileDepth], v1[kTileBatch][kTileDepth];  // two values (fp32 bits or fp64)
    };
    // Addresses are base (SGPR pair) + 32-bit unsigned BYTE offset (VGPR): the global_load "saddr" form, no
    // 64-bit address arithmetic per load (the host enables this kernel only while both tables are < 4 GiB).
    // `tiled` is followed by kTilePadRecords zeroed records, so the unconditional loads of masked-off lanes (at most
    // kGL x kTileDepth records past a segment start) need no clamp and never see a non-finite value.
    auto load_seg = [&](int tile, int b, Seg &sg_) {
        const int tl = min(tile, T - 1);
#pragma unroll
        for (int jj = 0; jj < kTileBatch; ++jj) {
            const int pj = person[b * kTileBatch + jj];
            const unsigned boff = (unsigned)tile_idx(pj >= 0 ? pj : 0, tl, T) << 2;
            typedef int v2i_t __attribute__((ext_vector_type(2), aligned(4)));  // this entry and the next one
            const v2i_t sp = *reinterpret_cast<const v2i_t *>(reinterpret_cast<const char *>(ta.seg4) + boff);
            sg_.x[jj] = sp.x;
            sg_.y[jj] = sp.y;
        }
    };
    auto load_edges = [&](const Seg &sg_, Edges &e) {
#pragma unroll
        for (int jj = 0; jj < kTileBatch; ++jj)
#pragma unroll
            for (int d = 0; d < kTileDepth; ++d) {
                if (ABL == 3) {
                    e.c[jj][d] = (unsigned)(seg_s0(sg_, jj) & 1023) * 0x10001u;
                    e.v0[jj][d] = e.v1[jj][d] = gl;
                } else {  // s0 is even: record s0 / 2; a group of kGL lanes covers 2 * kGL consecutive edges per load
                    typedef unsigned v2u_t __attribute__((ext_vector_type(2), aligned(4)));
                    typedef unsigned v3u_t __attribute__((ext_vector_type(3)));
                    typedef unsigned v4u_t __attribute__((ext_vector_type(4), aligned(4)));
                    constexpr unsigned kRec = (unsigned)F::kRec;
                    const unsigned boff = ((unsigned)seg_s0(sg_, jj) >> 1) * kRec + kRec * (unsigned)gl;
                    const char *src =  // + kRec * kGL * d bytes goes into the instruction's immediate offset
                        reinterpret_cast<const char *>(ta.tpk) + boff + kRec * kGL * d;
                    if constexpr (kFmt == 0) {         // 12 B: {slots, f32, f32}
                        const v3u_t y = *reinterpret_cast<const v3u_t *>(src);
                        e.c[jj][d] = y.x;
                        e.v0[jj][d] = (int)y.y;
                        e.v1[jj][d] = (int)y.z;
                    } else if constexpr (kFmt == 1) {  // 20 B: {slots, f64, f64}: one dword + four
                        e.c[jj][d] = *reinterpret_cast<const unsigned *>(src);
                        const v4u_t y = *reinterpret_cast<const v4u_t *>(src + 4);
                        e.v0[jj][d] = __hiloint2double((int)y.y, (int)y.x);
                        e.v1[jj][d] = __hiloint2double((int)y.w, (int)y.z);
                    } else if constexpr (kFmt == 2) {  // 16 B: {slots, stored indices, f32, f32}
                        const v4u_t y = *reinterpret_cast<const v4u_t *>(src);
                        e.c[jj][d] = y.x;
                        e.gg[jj][d] = y.y;
                        e.v0[jj][d] = (int)y.z;
                        e.v1[jj][d] = (int)y.w;
                    } else {                           // 24 B: {slots, stored indices, f64, f64}: two dwords + four
                        const v2u_t x = *reinterpret_cast<const v2u_t *>(src);
                        const v4u_t y = *reinterpret_cast<const v4u_t *>(src + 8);
                        e.c[jj][d] = x.x;
                        e.gg[jj][d] = x.y;
                        e.v0[jj][d] = __hiloint2double((int)y.y, (int)y.x);
                        e.v1[jj][d] = __hiloint2double((int)y.w, (int)y.z);
                    }
                }
            }
    };
    // Person rows beyond the workgroup's share are absent in every lane group (rows fill in order), so a round
    // with few bidders (K well below N: half of the launches of a solve) needs fewer steps per tile: the tile loop
    // is instantiated per number of person batches actually used.
#ifdef MISSLAP_TILED_STAMP
    // Diagnostic build (never the product): where a compute wavefront's cycles go -> Ctl::dbg[6..11], summed over all
    // compute wavefronts of the launch: [6] total, [7] waiting at the tile barrier, [8] waiting for the step's edges
    // (vmcnt), [9] look-ups + arithmetic of the step, [10] leftover loop, [11] number of wavefronts.
    unsigned long long sacc[6] = {0, 0, 0, 0, 0, 0};
    auto now = [&]() {
        unsigned long long tt;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
        return tt;
    };
    const unsigned long long t_begin = now();
#define MISSLAP_STAMP(K, EXPR)                      \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        const unsigned long long t_a = now();       \
        __builtin_amdgcn_sched_barrier(0);          \
        EXPR;                                       \
        __builtin_amdgcn_sched_barrier(0);          \
        sacc[K] += now() - t_a;                     \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)
#else
#define MISSLAP_STAMP(K, EXPR) \
    do {                       \
        EXPR;                  \
    } while (0)
#endif
    auto run_tiles = [&](auto nbe_) {
        constexpr int NBE = decltype(nbe_)::value;  // steps (person batches) per tile
#ifndef MISSLAP_TILED_SEGPF
#define MISSLAP_TILED_SEGPF 0  // extra steps of lead of the segment-table loads over the edge loads that need them
#endif
        // Queues of the pipeline: a step issues the edge loads of step s + kE -- their addresses come from the segment
        // entries of that step, which must have landed at the top of step s -- and the segment loads of step s + kQ.
        // kQ = kE + 1 makes those entries the youngest loads of the previous step; one more step of lead
        // (MISSLAP_TILED_SEGPF = 1, + 16 VGPRs) and / or a second step of edges in flight (MISSLAP_TILED_PF = 2) were
        // measured on the round-3 kernel: 91.4 / 90.8 us against 90.5 (C3 full scan in a solve) -- the wavefronts do not
        // wait for their own loads (stamped build: 1-2 % of their cycles), they wait at the tile barrier and in the
        // issue queue of the CU's memory path, which the fills and the edge loads share.
        constexpr int kE = MISSLAP_TILED_PF, kQ = kE + 1 + MISSLAP_TILED_SEGPF;
        Seg sq[kQ + 1];
        Edges eq[kE + 1];
#pragma unroll
        for (int k = 0; k < kQ; ++k) load_seg(rt(t_lo + k / NBE), k % NBE, sq[k]);
#pragma unroll
        for (int k = 0; k < kE; ++k) load_edges(sq[k], eq[k]);
        Seg &seg_cur = sq[0];
        Edges &e_cur = eq[0];
        if (kLoaders == 0 && kDouble) dma_fill(rt(t_lo), wave_u, kWaves);
        for (int tile_u = t_lo; tile_u < (loader ? t_lo : t_hi); ++tile_u) {
            const int tile = rt(tile_u);  // (the tile of this step of the walk)
            if (!kDouble) {
                __syncthreads();  // every lookup of the previous tile is done
                dma_fill(tile, wave_u, kWaves);
            }
            if (kLoaders == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of this tile
            MISSLAP_STAMP(1, if (ABL != 4) __syncthreads());  // every piece of this tile has landed; the other buffer may be refilled
#pragma unroll
            for (int b = 0; b < NBE; ++b) {
                // issue: edges of step s + kE, segment entries of step s + kQ
                load_edges(sq[kE], eq[kE]);
                load_seg(rt(tile_u + (b + kQ) / NBE), (b + kQ) % NBE, sq[kQ]);
                if (kDouble && kLoaders == 0 && b == 0 && tile_u + 1 < t_hi) dma_fill(rt(tile_u + 1), wave_u, kWaves);
#ifdef MISSLAP_TILED_STAMP
                // this step's edges: everything but the loads just issued (edges of the next step + two segment entries
                // per person of the step after it)
                MISSLAP_STAMP(2, asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kTileBatch * kTileDepth * kE + kTileBatch * (kE + 1)) : "memory"));
                const unsigned long long t_c = now();
#endif
                // consume step (tile, b): first ALL price look-ups of a depth range (independent ds_reads, one wait),
                // then the arithmetic -- a look-up followed by its use costs one LDS latency per element
                int rem[kTileBatch];  // real elements of the segment from this lane's first one on
#pragma unroll
                for (int jj = 0; jj < kTileBatch; ++jj)
                    rem[jj] = (person[b * kTileBatch + jj] >= 0 ? seg_s1(seg_cur, jj) - seg_s0(seg_cur, jj) : 0) - 2 * gl;
                const int col0 = tile * kTileCols - (kDouble ? (tile & 1) * kBufDoubles : 0);  // column of slot 0 of this tile's buffer
                auto consume = [&](const int dlo, const int dhi) {
                    double prs[kTileBatch][kTileDepth][2];
#pragma unroll
                    for (int jj = 0; jj < kTileBatch; ++jj) {
#pragma unroll
                        for (int d = dlo; d < dhi; ++d) {
                            const unsigned c = e_cur.c[jj][d];
                            const int a0 = (int)((c & 0xffffu) << kSlotShift), a1 = (int)((c >> 16) << kSlotShift);  // slot -> LDS byte address
                            if (ABL == 6) {  // no LDS look-ups: the "price" is made from the slot bits
                                prs[jj][d][0] = (double)a0;
                                prs[jj][d][1] = (double)a1;
                                continue;
                            }
                            prs[jj][d][0] = lds_price((2 * kGL * d < rem[jj]) ? a0 : kInfOff);  // masked-off: +inf
                            prs[jj][d][1] = lds_price((2 * kGL * d + 1 < rem[jj]) ? a1 : kInfOff);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the look-ups together, ahead of the arithmetic
#pragma unroll
                    for (int jj = 0; jj < kTileBatch; ++jj) {
                        const int j = b * kTileBatch + jj;
                        const int q0 = seg_s0(seg_cur, jj) + 2 * gl;
                        int sslot = -1;  // slot of the best element if it moved in this step
                        const double sv0 = sv[j];  // (kRev) the best value and its cost before this step
                        const VT sc0 = scost[j];
                        const int want8 = (MODE == 1 && tile == wtile[j]) ? wslot8[j] : -1;  // (MODE 1) no slot offset is negative
#pragma unroll
                        for (int d = dlo; d < dhi; ++d) {
                            if (ABL == 2) {
                                asm volatile("" ::"v"(e_cur.c[jj][d]), "v"(e_cur.v0[jj][d]), "v"(e_cur.v1[jj][d]));  // keep the loads alive
                                continue;
                            }
                            if (ABL == 5) {  // look-ups but no arithmetic
                                asm volatile("" ::"v"(prs[jj][d][0]), "v"(prs[jj][d][1]), "v"(e_cur.v0[jj][d]), "v"(e_cur.v1[jj][d]));
                                continue;
                            }
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const VT vb = h ? e_cur.v1[jj][d] : e_cur.v0[jj][d];
                                // a masked-off element has v = -inf and changes neither sv nor sw; `ok` keeps it from
                                // taking sg when sv is still -inf (rows whose objects all have an infinite price)
                                const double v = F::val(vb) - prs[jj][d][h];  // vi = cost - p[j]   (:350)
                                // what orders equal values: the tile-major position (ascending columns: monotone in
                                // the stored index) or the stored index itself (formats 2 / 3)
                                const int q = F::kG ? (int)((h ? e_cur.gg[jj][F::kG ? d : 0] >> 16 : e_cur.gg[jj][F::kG ? d : 0] & 0xffffu))
                                                    : q0 + 2 * kGL * d + h;
                                if (MODE == 1) {  // (:467-471, :480-482) a lane meets its elements in stored order: the last match stays
                                    const int a8 = (int)((h ? e_cur.c[jj][d] >> 16 : e_cur.c[jj][d] & 0xffffu) << 3);
                                    const bool m = (2 * kGL * d + h < rem[jj]) & (a8 == want8);
                                    const bool last = F::kG ? (m & (q > sg[j])) : m;  // (formats 2 / 3: by stored index)
                                    sv[j] = __builtin_fmax(sv[j], v);  // a masked-off element has v = -inf
                                    sg[j] = last ? q : sg[j];
                                    scost[j] = last ? vb : scost[j];
                                    mcnt[j] += m;
                                    continue;
                                }
                                // a lane meets its elements in stored order: ">=" is the reference's tie rule (formats
                                // 2 / 3: in ANY order -- among equal values the larger stored index wins, spelled out)
                                const bool ge = (2 * kGL * d + h < rem[jj]) &
                                                (F::kG ? ((v > sv[j]) | ((v == sv[j]) & (q > sg[j]))) : (v >= sv[j]));  // :351
                                sw[j] = __builtin_fmax(sw[j], __builtin_fmin(v, sv[j]));      // :353 / :357-358
                                sv[j] = __builtin_fmax(sv[j], v);
                                if (!kKeyCol) sg[j] = ge ? q : sg[j];
                                if (kCS == 1 && ABL == 0) {
                                    scost[j] = ge ? vb : scost[j];
                                    sslot = ge ? (int)(h ? e_cur.c[jj][d] >> 16 : e_cur.c[jj][d] & 0xffffu) : sslot;
                                }
                            }
                        }
                        // note_best: a person whose best element moved takes its column and cost from the registers just
                        // consumed -- element rel = 2 * kGL * d + h of my share of the segment; the slot of an edge is its
                        // column relative to the tile (+ the buffer of the tile's parity, see k_tile_scatter)
                        if constexpr (kRev != 0) {  // a best that only TIES the one of the tiles walked before (larger columns) stays with them
                            const bool tie_only = (sslot >= 0) & (sv[j] == sv0) & (scol[j] >= 0);
                            scost[j] = tie_only ? sc0 : scost[j];
                            sslot = tie_only ? -1 : sslot;
                        }
                        if (kCS == 1 && ABL == 0) scol[j] = sslot >= 0 ? col0 + sslot : scol[j];
                    }
                };
                // The first kFastDepth loads of a segment cover most segments and are consumed unconditionally; the
                // loads beyond them are issued with the others (software-pipelined, no stall) but consumed only in the
                // steps where some segment of the wavefront is that long (a wave-uniform branch with no load inside).
                constexpr int kFastDepth = kTileDepth < 2 ? kTileDepth : 2;
                consume(0, kFastDepth);
                if (kTileDepth > kFastDepth) {
                    bool more_d = false;
#pragma unroll
                    for (int jj = 0; jj < kTileBatch; ++jj) more_d |= rem[jj] > 2 * kGL * kFastDepth;
                    if (__any(more_d)) consume(kFastDepth, kTileDepth);
                }
#ifdef MISSLAP_TILED_STAMP
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t_l = now();
                sacc[3] += t_l - t_c;
#endif
                // (segments longer than 2 * kGL * kTileDepth edges: their tail is on the person's overflow list, looked
                // at after the last tile -- no load inside the tile loop)
#ifdef MISSLAP_TILED_STAMP
                __builtin_amdgcn_sched_barrier(0);
                sacc[4] += now() - t_l;
#endif
#pragma unroll
                for (int k = 0; k < kQ; ++k) sq[k] = sq[k + 1];
#pragma unroll
                for (int k = 0; k < kE; ++k) eq[k] = eq[k + 1];
            }
        }
    };
    {
        const int rows_used = (p1 - p0 + kTileGroups - 1) / kTileGroups;
        const int nb_used = (rows_used + kTileBatch - 1) / kTileBatch;  // uniform over the workgroup
        if (nb_used <= 1)
            run_tiles(std::integral_constant<int, 1>{});
        else if (kNB > 2 && nb_used <= 2)
            run_tiles(std::integral_constant<int, (kNB > 2 ? 2 : 1)>{});
        else
            run_tiles(std::integral_constant<int, kNB>{});
    }
#ifdef MISSLAP_TILED_STAMP
    // (MISSLAP_TILED_STAMP == 2: the epilogue instead -- [6] tile loop, [7] overflow lists, [8] column split: publish +
    // count in, [9] the last arriver's loads + merge, [10] bids + statistics, [11] number of wavefronts)
    unsigned long long t_e = now(), eacc[5] = {0, 0, 0, 0, 0};
    if (!loader && lane == 0 && MISSLAP_TILED_STAMP + 0 < 2) {
        atomicAdd(&a.ctl->dbg[6], t_e - t_begin);
        for (int k = 1; k <= 4; ++k) atomicAdd(&a.ctl->dbg[6 + k], sacc[k]);
        atomicAdd(&a.ctl->dbg[11], 1ull);
    }
    eacc[0] = t_e - t_begin;
#define MISSLAP_ESTAMP(K)                                    \
    do {                                                     \
        if (MISSLAP_TILED_STAMP + 0 >= 2) {                  \
            __builtin_amdgcn_sched_barrier(0);               \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
            const unsigned long long t_n = now();            \
            eacc[K] += t_n - t_e;                            \
            t_e = t_n;                                       \
            __builtin_amdgcn_sched_barrier(0);               \
        }                                                    \
    } while (0)
#define MISSLAP_EFLUSH()                                                          \
    do {                                                                          \
        if (MISSLAP_TILED_STAMP + 0 >= 2 && t == 0 && (MISSLAP_TILED_STAMP + 0 == 2 || (hi - lo < ta.part_stride) == (MISSLAP_TILED_STAMP + 0 == 3))) {   /* wavefront 0 only; 3: partial rounds, 4: full scans */     \
            for (int k = 0; k < 5; ++k) atomicAdd(&a.ctl->dbg[6 + k], eacc[k]);   \
            atomicAdd(&a.ctl->dbg[11], 1ull);                                     \
        }                                                                         \
    } while (0)
#else
#define MISSLAP_ESTAMP(K) \
    do {                  \
    } while (0)
#define MISSLAP_EFLUSH() \
    do {                 \
    } while (0)
#endif
    // overflow edges (see k_ovf_count): lane gl of a person's group takes entries gl, gl + kGL, ... of its list; the price
    // comes from memory (the table the tiles were filled from), the update is the same with the tie rule spelled out
    // -- these edges are met out of stored order, so among equal values the later stored POSITION must win (:351)
    if (kKeyCol) {
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) sg[j] = scol[j];  // from here on the key of the tie rule is the column
    }
    if (ABL == 0 && !loader) {
        if (t == 0 && ta.ovf_cap != 2 * kGL * kTileDepth) atomicOr(&a.ctl->err, kErrLdsBase);
        // (four persons at a time: with eight persons per lane group the lists of all of them together would not fit
        // the registers next to the running top-2s)
        constexpr int kOB = kTileRows < 4 ? kTileRows : 4;
#pragma unroll
        for (int j0 = 0; j0 < kTileRows; j0 += kOB) {
            int oi[kOB], oe[kOB];
            int ql[kOB], qh[kOB];  // (column split) the tile-major positions of my tiles inside the person's block
            bool more = false;
#pragma unroll
            for (int jj = 0; jj < kOB; ++jj) {
                const int j = j0 + jj;
                const int pj = max(person[j], 0);
                typedef int v2i_t __attribute__((ext_vector_type(2), aligned(4)));
                const v2i_t pp = *reinterpret_cast<const v2i_t *>(ta.ovf_ptr + pj);
                oi[jj] = pp.x + gl;
                oe[jj] = person[j] >= 0 ? pp.y : pp.x;
                more |= oi[jj] < oe[jj];
                ql[jj] = qh[jj] = 0;
                if (kCS > 1) {  // positions grow with (block, tile, person): tile t of a block starts at its first person's entry
                    const int blk = pj / kTileRB;
                    ql[jj] = ta.seg4[(blk * T + t_lo) * kTileRB] & ~1;
                    qh[jj] = ta.seg4[(blk * T + t_hi) * kTileRB] & ~1;
                }
            }
            while (__any(more)) {
                int q[kOB];
                VT vb[kOB];
                double pr[kOB];
                int4 en[kOB];
#pragma unroll
                for (int jj = 0; jj < kOB; ++jj) en[jj] = ta.ovf[oi[jj] < oe[jj] ? oi[jj] : 0];
#pragma unroll
                for (int jj = 0; jj < kOB; ++jj) {
                    q[jj] = kKeyCol ? en[jj].y : en[jj].x;  // the key that orders equal values: position (kKeyCol: column)
                    if constexpr (F::kF64) vb[jj] = __hiloint2double(en[jj].w, en[jj].z);
                    else vb[jj] = en[jj].z;
                    pr[jj] = a.price[en[jj].y];
                }
                more = false;
#pragma unroll
                for (int jj = 0; jj < kOB; ++jj) {
                    const int j = j0 + jj;
                    const bool ok = (oi[jj] < oe[jj]) & (kCS == 1 || ((q[jj] >= ql[jj]) & (q[jj] < qh[jj])));
                    const double v = ok ? F::val(vb[jj]) - pr[jj] : ninf;  // vi = cost - p[j]   (:350)
                    if (MODE == 1) {  // these edges are met out of stored order: the later stored POSITION is the last match
                        const int wcol = wtile[j] * kTileCols + (wslot8[j] >> 3) - (kDouble ? (wtile[j] & 1) * kBufDoubles : 0);
                        const bool m = ok & (wtile[j] >= 0) & (en[jj].y == wcol);
                        const bool later = m & (q[jj] > sg[j]);
                        sv[j] = __builtin_fmax(sv[j], v);
                        sg[j] = later ? q[jj] : sg[j];
                        scost[j] = later ? vb[jj] : scost[j];
                        mcnt[j] += m;
                        oi[jj] += kGL;
                        more |= oi[jj] < oe[jj];
                        continue;
                    }
                    const bool ge = ok & ((v > sv[j]) | ((v == sv[j]) & (q[jj] > sg[j])));    // :351, by stored position
                    sw[j] = __builtin_fmax(sw[j], __builtin_fmin(v, sv[j]));                // :353 / :357-358
                    sv[j] = __builtin_fmax(sv[j], v);
                    sg[j] = ge ? q[jj] : sg[j];
                    scol[j] = ge ? en[jj].y : scol[j];
                    scost[j] = ge ? vb[jj] : scost[j];
                    oi[jj] += kGL;
                    more |= oi[jj] < oe[jj];
                }
            }
        }
    }
    MISSLAP_ESTAMP(1);
    if (MODE == 1) {
        // one lane per person -- the one that holds the last match, lane 0 of the group when there is none -- finishes
        // the person (kernels_check.hpp); the price of the wanted column comes from memory, requested for all persons
        // of the lane group before the first is used
        int pj_[kTileRows];
        double wp[kTileRows];
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) {
            pj_[j] = a.p2o[max(person[j], 0)];
            wp[j] = a.price[max(wanted_column(pj_[j], a.n_cols), 0)];
        }
        FinalAcc acc;
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) {
            const double V = group_max_f64<kGL>(sv[j]);
            const int Q = group_max_i32<kGL>(sg[j]);
            const int n = group_sum_i32<kGL>(mcnt[j]);
            const bool me = person[j] >= 0 && (Q >= 0 ? sg[j] == Q : gl == 0);
            if (me) final_person(acc, ta.fo, person[j], pj_[j], Q >= 0, n, F::val(scost[j]), V, wp[j], eps);
        }
        flag_ece_failure(a.ctl, acc.bad);
        if (ta.fo.fin) {  // uniform over the launch
            __syncthreads();  // every look-up is done: the price buffers are free
            flush_final_wg(ta.fo, acc, s_price);
        }
        return;
    }
    // merge the kGL lanes of each group, once per person (same three all-reduces as top2_wave_reduce)
    unsigned long long edges = 0;
    int nb = 0, err = 0;
    int bcol[kTileRows];
    VT bcost[kTileRows];
    int rlen[kTileRows];
    double W[kTileRows];
    int G[kTileRows];
    bool mine[kTileRows];  // exactly one lane of the group
    // statistics / hand-over scratch.  It lives BEHIND the price buffers in the dynamic allocation: a static __shared__
    // array would be placed first and shift s_price off LDS address 0, which costs one v_add per price look-up (the
    // edges carry absolute LDS offsets)
    unsigned long long *s_e = reinterpret_cast<unsigned long long *>(s_price + (kDouble ? kBufDoubles + kTileCols : kTileCols + 2));
    int *s_n = reinterpret_cast<int *>(s_e + kTileThreads / kWave);
    if (kCS == 1) {
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) {
            const double V = group_max_f64<kGL>(sv[j]);
            G[j] = group_max_i32<kGL>(sv[j] == V ? sg[j] : -1);
            W[j] = group_max_f64<kGL>(sg[j] == G[j] ? sw[j] : sv[j]);
            mine[j] = person[j] >= 0 && sg[j] == G[j] && G[j] >= 0;  // the lane that holds the best edge
        }
    } else {
        // Column split: my top-2 covers my share of the tiles only.  Every workgroup of the slice publishes its partial
        // results and counts itself in; the LAST one to arrive finds all of them complete, merges them (the merge
        // operator of the reduction; positions of a person grow with the tile, so the later stored position wins a tie,
        // :351) and forms the bids of the slice.  Nobody waits for anybody.
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) {
            const double V = group_max_f64<kGL>(sv[j]);
            const int Gj = group_max_i32<kGL>(sv[j] == V ? sg[j] : -1);
            const double Wj = group_max_f64<kGL>(sg[j] == Gj ? sw[j] : sv[j]);
            mine[j] = person[j] >= 0 && gl == 0;
            if (mine[j]) {
                const size_t at = (size_t)half * ta.part_stride + (size_t)(p0 + j * kTileGroups + group);
                split_store(reinterpret_cast<unsigned long long *>(ta.part_vw + at), (unsigned long long)__double_as_longlong(V));
                split_store(reinterpret_cast<unsigned long long *>(ta.part_vw + at) + 1,
                            (unsigned long long)__double_as_longlong(Gj >= 0 ? Wj : ninf));
                split_store(ta.part_g + at, Gj);
            }
        }
        // The XCDs' L2s are not coherent with each other.  A device-scope fence (__threadfence) would write back and
        // invalidate the whole L2 of the XCD, per wavefront, under the other workgroups' price tiles (measured: the
        // scan three times as long).  The partial results travel as device-scope atomic stores / loads instead (sc1:
        // performed at the level all XCDs share), the arrival count is a device-scope atomic as well, and the order is
        // made by hand: my stores have completed (vmcnt) before the barrier behind which thread 0 counts us in; the
        // last arriver's loads are issued behind the barrier that hands it the count.
        int *s_arrived = s_n + kTileThreads / kWave;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) *s_arrived = atomicAdd(&ta.split_cnt[slice], 1);
        __syncthreads();
        MISSLAP_ESTAMP(2);
        if (*s_arrived != kCS - 1) {  // uniform over the workgroup
            MISSLAP_EFLUSH();
            return;
        }
        if (t == 0) ta.split_cnt[slice] = 0;  // for the next launch
        double2 P[kCS][kTileRows];
        int pg[kCS][kTileRows];
#pragma unroll
        for (int o = 0; o < kCS; ++o)
#pragma unroll
            for (int j = 0; j < kTileRows; ++j) {
                const size_t at = (size_t)o * ta.part_stride + (size_t)min(p0 + j * kTileGroups + group, p1 - 1);
                P[o][j].x = __longlong_as_double((long long)split_load(reinterpret_cast<const unsigned long long *>(ta.part_vw + at)));
                P[o][j].y = __longlong_as_double((long long)split_load(reinterpret_cast<const unsigned long long *>(ta.part_vw + at) + 1));
                pg[o][j] = split_load(ta.part_g + at);
            }
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) {
            double V = P[0][j].x;
            W[j] = P[0][j].y;
            G[j] = pg[0][j];
#pragma unroll
            for (int o = 1; o < kCS; ++o) {
                const double2 B = P[o][j];
                const int gb = pg[o][j];
                const bool take = (gb >= 0) & ((G[j] < 0) | (B.x > V) | ((B.x == V) & (gb > G[j])));
                W[j] = take ? __builtin_fmax(B.y, G[j] >= 0 ? V : ninf) : __builtin_fmax(W[j], gb >= 0 ? B.x : ninf);
                V = take ? B.x : V;
                G[j] = take ? gb : G[j];
            }
            if (mine[j] && G[j] < 0) err |= kErrRowGap;  // a bidder without a single edge: the ingest excludes it
            mine[j] = mine[j] && G[j] >= 0;
        }
    }
    if (kCS > 1) MISSLAP_ESTAMP(3);
#pragma unroll
    for (int j = 0; j < kTileRows; ++j) {
        const int pj = max(person[j], 0);
        if constexpr (kCS == 1) {  // `mine` is the lane that met the best edge
            bcol[j] = ABL == 0 ? scol[j] : pj % a.n_cols;  // (ablations carry no column: spread the atomics like real bids)
            bcost[j] = scost[j];
        } else {         // any workgroup of the slice may hold it: through its position (unconditional loads, used under `mine`)
            bcol[j] = ta.tcol[max(G[j], 0)];
            bcost[j] = (int)ta.tpk[(max(G[j], 0) >> 1) * 3 + 1 + (max(G[j], 0) & 1)];
        }
        rlen[j] = a.row_ptr[pj + 1] - a.row_ptr[pj];
    }
    if (kCS == 1) MISSLAP_ESTAMP(2);  // (unsplit shapes: [8] = the row lengths have landed, [9] = the bids' stores and atomics have retired)
#pragma unroll
    for (int j = 0; j < kTileRows; ++j) {
        if (mine[j]) {
            const double cost = F::val(bcost[j]);
            const double bid = (cost - W[j]) + eps;  // :360
            if (bid_is_bad(bid)) err |= kErrNegativeBid;
            const unsigned long long key = bid_to_key(bid);
            const int slot = p0 + j * kTileGroups + group;
            const int pos = ta.order_pos ? ta.order_pos[slot] : slot;
            a.bid_key[pos] = key;
            a.bid_obj[pos] = bcol[j];
            atomicMax(&a.best_key[bcol[j]], key);
            edges += (unsigned long long)rlen[j];
            nb += 1;
        }
    }
    if (kCS == 1) MISSLAP_ESTAMP(3);
    // statistics: one atomic per workgroup
    for (int off = 32; off >= 1; off >>= 1) {
        edges += ((unsigned long long)__shfl_xor((unsigned)(edges >> 32), off) << 32) |
                 (unsigned long long)__shfl_xor((unsigned)(edges & 0xffffffffull), off);
        nb += __shfl_xor(nb, off);
        err |= __shfl_xor(err, off);
    }
    if (lane == 0) {
        s_e[t >> 6] = edges;
        s_n[t >> 6] = nb;
        if (err && ABL == 0) atomicOr(&a.ctl->err, err);  // (an ablation's bids are garbage: 3 328 atomics on one word would be its timing)
    }
    __syncthreads();
    if (t == 0) {
        unsigned long long te = 0;
        int tb = 0;
        for (int w = 0; w < kTileThreads / kWave; ++w) {
            te += s_e[w];
            tb += s_n[w];
        }
        if (tb) {  // (no other workgroup touches my slot, see RoundArgs::wg_stats)
            unsigned long long *st = a.wg_stats + (size_t)kStatWords * blockIdx.x;
            st[kStatEdges] += te;
            st[kStatBids] += (unsigned long long)tb;
            if (a.world > 1 && a.ctl->K >= a.shard_min_K) st[kStatShardEdges] += te;
            if (a.launch_edges) st[kStatLaunchEdges] += te;
        }
    }
    MISSLAP_ESTAMP(4);
    MISSLAP_EFLUSH();
}

}  // namespace misslap
