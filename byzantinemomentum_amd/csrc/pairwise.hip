// pairwise.hip — the n x n squared-distance entry point, the DIRECT-DIFFERENCE kernel, the
// cross-workgroup reduction and the on-device score/rank step of Krum and Bulyan.
//
// Replaces the reference's per-pair loop
//     dist = gradients[x].sub(gradients[y]).norm().item()
// (aggregators/krum.py:44-48, bulyan.py:49-54, brute.py:43-45): n(n-1)/2 x {alloc d, write d,
// read d, host sync} ~ 100x the algorithmic traffic.  Here every coordinate is read from HBM
// once (4*d*n bytes).
//
// bm_pairwise_sqdist dispatches on BM_PAIR_MODE:
//   0 (default)  centred Gram contraction on the bf16 matrix cores with an error-free split, gram_bf16.hip,
//                followed by an accuracy gate that hands nearly coincident rows to the kernel of this file;
//   1            the kernel in this file for every pair: the direct form sum_k (a_k - b_k)^2, no cancellation
//                at all, but two VALU lane-ops per (pair, coordinate) make it VALU-bound on gfx950
//                (1.05 ms at n=51, d=11.2 M; VALU ~90 % busy).
// (The fp32-MFMA Gram of round 1 is no longer part of the library: scripts/probes/gram_fp32_mfma.hip.)
// Both give an exact 0 between bitwise-equal rows and bitwise-equal distances from them to any
// third row (exact score ties, broken by index like the reference's stable sort, krum.py:62).
//
// Direct kernel shape (gfx950):
//   * a workgroup stages a [rows][4*slots*S coords] tile in LDS with the LDS-DMA engine
//     (global_load_lds_dwordx4, 1 KiB per wave instruction, no staging VGPRs, no ds_write), two
//     tile buffers, the DMA of tile t+1 in flight under the arithmetic of tile t;
//   * the n rows are cut in groups of 4; a lane owns ONE 4x4 pair tile (I <= J) and one strip of
//     coordinates, keeps its 16 (x2, packed even/odd) fp32 accumulators in VGPRs for the whole
//     kernel, and per 4-coordinate slot reads 8 x ds_read_b128 (software pipelined, ping-pong
//     register sets) and issues 32 v_pk_add_f32 (neg) + 32 v_pk_fma_f32;
//   * LDS rows live in 1 KiB DMA blocks padded by one 16-B slot, block(I, a) = I + NG*(a/rb), and
//     lanes are mapped so that each hardware ds_read_b128 service group of 16 lanes sees one
//     strip and 16 distinct pair tiles: every read is bank-conflict free (0.2 % measured), and every
//     pair accumulates its coordinates in the same canonical order (needed for the exact ties);
//   * per-workgroup partial sums go to a workspace as fp64 and are reduced in a fixed order by
//     a second tiny kernel — deterministic, no float atomics.
#include "bm_common.h"
#include "rank_body.h"

namespace bm {

constexpr int kPairMaxThreads = 512;
constexpr int kTileR = 4;          // rows per group; pair tile = kTileR x kTileR
constexpr int kDmaBlock = 1024;    // bytes moved by one wave-wide global_load_lds_dwordx4
constexpr int kDmaPitch = 1024 + 16;  // LDS pitch of a DMA block: one 16-B slot of padding

struct PairGeom {
  int n;          // rows
  int ng;         // row groups = ceil(n/4)
  int tiles;      // ng*(ng+1)/2 pair tiles (I <= J)
  int ut;         // 16-lane units per strip = ceil(tiles/16)
  int strips;     // S: coordinate strips per LDS tile, one strip per 16-lane unit
  int slots;      // 16-byte slots (4 coordinates) a lane walks per tile: 8 or 16
  int threads;    // 16*ut*S rounded up to 64
  int width;      // coordinates per LDS tile = 4*slots*S
  int row_bytes;  // 16*slots*S in {256, 512, 1024}
  int rb;         // rows per 1 KiB DMA block = 1024/row_bytes in {4, 2, 1}
  int nb;         // DMA blocks per tile = ng * (4/rb)
};

__host__ __device__ inline PairGeom pair_geometry(int n, int forced) {
  PairGeom g;
  g.n = n;
  g.ng = (n + kTileR - 1) / kTileR;
  g.tiles = g.ng * (g.ng + 1) / 2;
  g.ut = (g.tiles + 15) / 16;
  // Candidates (strips, slots) with row_bytes = 16*slots*strips in {256, 512, 1024}.  Pick the best
  // lane utilisation; among equals the largest tile whose two buffers stay within 32 KB (so that
  // four or five workgroups fit in the 160 KB of LDS of a CU).  forced = strips*100+slots (experiments).
  const int cand[6][2] = {{2, 8}, {4, 8}, {8, 8}, {1, 16}, {2, 16}, {4, 16}};
  int best = -1, best_bytes = 0, best_num = -1, best_den = 1;  // utilisation as the fraction num/den
  for (int c = 0; c < 6; ++c) {
    const int st = cand[c][0], sl = cand[c][1];
    const int lanes = 16 * g.ut * st;
    const int thr = ((lanes + 63) / 64) * 64;
    const int rowb = 16 * sl * st;
    const int tile_bytes = g.ng * (4 * rowb / kDmaBlock) * kDmaPitch;
    if (thr > kPairMaxThreads) continue;
    if (forced > 0) {
      if (forced == st * 100 + sl) {
        best = c;
        break;
      }
      continue;
    }
    if (2 * tile_bytes > 32 * 1024) continue;
    // lanes/thr > best_num/best_den, or equal with a larger tile (exact integer comparison)
    const long lhs = (long)lanes * best_den, rhs = (long)best_num * thr;
    if (best < 0 || lhs > rhs || (lhs == rhs && tile_bytes > best_bytes)) {
      best = c;
      best_num = lanes;
      best_den = thr;
      best_bytes = tile_bytes;
    }
  }
  if (best < 0) best = 0;
  g.strips = cand[best][0];
  g.slots = cand[best][1];
  g.threads = ((16 * g.ut * g.strips + 63) / 64) * 64;
  g.width = 4 * g.slots * g.strips;
  g.row_bytes = 16 * g.slots * g.strips;
  g.rb = kDmaBlock / g.row_bytes;
  g.nb = g.ng * (kTileR / g.rb);
  return g;
}

// lane -> (unit within wave 0..3, position within unit 0..15), following the ds_read_b128
// service groups of gfx950: {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59}
// {36-43,48-51,60-63}.
__device__ __forceinline__ void lane_unit(int lane, int& unit, int& pos) {
  const int half = lane >> 5;
  const int l = lane & 31;
  int u, p;
  if (l < 4) {
    u = 0; p = l;
  } else if (l < 12) {
    u = 1; p = l - 4;
  } else if (l < 16) {
    u = 0; p = l - 8;      // 12..15 -> 4..7
  } else if (l < 20) {
    u = 1; p = l - 8;      // 16..19 -> 8..11
  } else if (l < 28) {
    u = 0; p = l - 12;     // 20..27 -> 8..15
  } else {
    u = 1; p = l - 16;     // 28..31 -> 12..15
  }
  unit = half * 2 + u;
  pos = p;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Byte offset, inside an LDS tile, of coordinate 0 of logical row 4*I + a.
// DMA block = I + ng*(a / rb); a block holds rb rows of row_bytes.  Its bank-slot class is
// (I + ng*(a/rb)) mod 16 (row_bytes and the 128-byte strip offset are multiples of 8 slots, and
// a ds_read_b128 service group always sits in one strip): for a fixed `a`, rows of different
// groups I never share a 16-byte bank slot, so a read whose 16 lanes touch 16 different groups is
// conflict free (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.2 % measured).
__device__ __forceinline__ int row_offset_bytes(const PairGeom& g, int I, int a) {
  const int blk = I + g.ng * (a / g.rb);
  return blk * kDmaPitch + (a % g.rb) * g.row_bytes;
}

constexpr int kRedWaves = 8;
constexpr int kDirectArrivalSlot = 97;  // int slot of the gate's 512-byte row-list area: arrival counter of the gated call

// entry e = tile*16 + a*4 + b of a reduced partial -> sq[i][j] and sq[j][i] (sub: sub-stack position -> row)
__device__ __forceinline__ void pair_scatter(int e, double tot, const PairGeom& g, int n_full, double* __restrict__ sq,
                                             const int* __restrict__ sub) {
  const int tile = e >> 4, a = (e >> 2) & 3, b = e & 3;
  int I = 0, t = tile, row_len = g.ng;
  while (t >= row_len) {
    t -= row_len;
    --row_len;
    ++I;
  }
  const int J = I + t;
  int i = I * kTileR + a, j = J * kTileR + b;
  if (i >= g.n || j >= g.n) return;
  const bool keep = (i != j) && (I != J || a < b);
  if (sub != nullptr) {  // sub-stack position -> row of the full stack
    i = sub[1 + i];
    j = sub[1 + j];
  }
  const int n = n_full;
  if (i == j) {
    sq[i * n + j] = 0.0;
  } else if (keep) {
    // off-diagonal tiles hold each unordered pair once; diagonal tiles hold (a,b) and (b,a)
    // with bitwise-equal sums, keep the a<b copy
    sq[i * n + j] = tot;
    sq[j * n + i] = tot;
  }
}

// ALIGNED: every row pointer is 16-byte aligned -> full tiles are brought in by the LDS-DMA
// engine (global_load_lds_dwordx4: no staging VGPRs, no ds_write) into the OTHER of two tile
// buffers while the workgroup computes on the current one.  Ragged last tiles and unaligned inputs
// use a plain load + ds_write loop (same layout, not overlapped).
template <bool ALIGNED, int ABLATE = 0>
__global__ __launch_bounds__(kPairMaxThreads) void pairwise_partial_kernel(
    RowTable rows, PairGeom g, int64_t d, double* __restrict__ partial, int* __restrict__ sub, int n_full,
    double* __restrict__ sq, RankArgs rk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // sub (device, may be NULL): the rows the accuracy gate of the Gram path asks to recompute
  // (gram_to_sqdist_kernel): sub[0] = how many (0: this launch has nothing to do), sub[1..] = their
  // indices.  The kernel then works on that sub-stack with the geometry of ITS row count.
  // rk.on (gated launches of bm_pairwise_rank): when rows WERE listed, the workgroup that arrives last ranks the rows
  // after it has written the corrected distances; when nothing was listed (the common case) the last workgroup of the
  // Gram reduction has ranked them already (gram_reduce_sqdist_kernel: 16 waves, one row each, where this launch has
  // 2-3) and this launch is empty.
  if (sub != nullptr) {
    if (sub[0] == 0) return;
    g = pair_geometry(sub[0], 0);
  }
  const float** row_ptr = reinterpret_cast<const float**>(smem);  // 512 B pointer table
  char* tiles = smem + BM_MAX_ROWS * sizeof(float*);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int nwaves = blockDim.x >> 6;
  const int tile_bytes = g.nb * kDmaPitch;

  // one-time: pointer table, zeroed tiles (rows >= n and block padding are never written again)
  // row pointers: per-lane loads from the kernarg segment (see gram.hip)
  if (tid < BM_MAX_ROWS) {
    typedef const float* __attribute__((address_space(4))) const* KargTable;
    KargTable karg = (KargTable)__builtin_amdgcn_kernarg_segment_ptr();
    row_ptr[tid] = tid < g.n ? (const float*)karg[sub != nullptr ? sub[1 + tid] : tid] : nullptr;
  }
  for (int o = tid * 16; o < 2 * tile_bytes; o += blockDim.x * 16)
    *reinterpret_cast<f32x4*>(tiles + o) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  __syncthreads();

  int unit_in_wave, pos;
  lane_unit(lane, unit_in_wave, pos);
  const int unit = wave * 4 + unit_in_wave;  // global 16-lane unit
  const int strip = unit / g.ut;
  const int ptile = (unit % g.ut) * 16 + pos;
  const bool active = (strip < g.strips) && (ptile < g.tiles);
  // pair tile -> (I, J), I <= J, enumerated row-major over the upper triangle
  int ti = 0, tj = 0;
  {
    int t = active ? ptile : 0, row_len = g.ng;
    while (t >= row_len) {
      t -= row_len;
      --row_len;
      ++ti;
    }
    tj = ti + t;
  }
  int oi[kTileR], oj[kTileR];  // byte offsets of this lane's 4+4 rows inside a tile buffer
#pragma unroll
  for (int a = 0; a < kTileR; ++a) {
    oi[a] = row_offset_bytes(g, ti, a) + strip * g.slots * 16;
    oj[a] = row_offset_bytes(g, tj, a) + strip * g.slots * 16;
  }

  f32x2 acc[kTileR][kTileR];
#pragma unroll
  for (int a = 0; a < kTileR; ++a)
#pragma unroll
    for (int b = 0; b < kTileR; ++b) acc[a][b] = f32x2{0.0f, 0.0f};

  const int width = g.width;
  // DMA lane geometry: a 1 KiB block = rb rows x (row_bytes/16 lanes x 16 B)
  const int lanes_per_row = g.row_bytes / 16;
  const int dma_w = lane / lanes_per_row;                  // row within the block
  const int dma_col = (lane - dma_w * lanes_per_row) * 4;  // first coordinate of this lane

  // Bring chunk `base` into tile buffer `buf`.  Returns without waiting when the DMA path is used.
  auto stage = [&](int64_t base, char* buf) {
    if (ABLATE == 2) return;  // experiment: no staging at all
    if (ALIGNED && base + width <= d) {
      for (int blk = wave; blk < g.nb; blk += nwaves) {
        const int I = blk % g.ng;
        const int a = (blk / g.ng) * g.rb + dma_w;
        const int r = I * kTileR + a;
        if (r < g.n) {
          const float* src = row_ptr[r] + base + dma_col;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)src,
              (__attribute__((address_space(3))) void*)(buf + blk * kDmaPitch), 16, 0, 0);
        }
      }
    } else {
      // ragged tail / unaligned rows: plain loads with zero fill, same LDS layout
      const int vpr = width / 4;
      for (int idx = tid; idx < g.n * vpr; idx += blockDim.x) {
        const int r = idx / vpr;
        const int col = (idx - r * vpr) * 4;
        const float* src = row_ptr[r] + base + col;
        const int64_t left = d - (base + col);
        f32x4 val = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ALIGNED && left >= 4) {
          val = *reinterpret_cast<const f32x4*>(src);
        } else {
          if (left > 0) val.x = src[0];
          if (left > 1) val.y = src[1];
          if (left > 2) val.z = src[2];
          if (left > 3) val.w = src[3];
        }
        *reinterpret_cast<f32x4*>(buf + row_offset_bytes(g, r >> 2, r & 3) + col * 4) = val;
      }
    }
  };

  struct Slot {
    f32x2 il[kTileR], ih[kTileR], jl[kTileR], jh[kTileR];
  };

  // chunk c of this workgroup = blockIdx.x + c*gridDim.x (neighbouring workgroups stream
  // neighbouring addresses); the per-pair accumulation order is the same for every pair.
  int64_t chunk = blockIdx.x;
  if (chunk * width < d) stage(chunk * width, tiles);
  for (int it = 0;; ++it) {
    const int64_t base = chunk * width;
    if (base >= d) break;
    char* cur = tiles + (it & 1) * tile_bytes;
    char* nxt = tiles + ((it + 1) & 1) * tile_bytes;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of `cur` have landed
    __syncthreads();  // ... everyone's have, and everyone is done reading `nxt`
    chunk += gridDim.x;
    if (chunk * width < d) stage(chunk * width, nxt);  // overlaps the compute below

    // ---- `slots` steps of 4 coordinates, canonical order for every pair ----
    if (active && ABLATE != 1) {
      // Software pipeline: the 8 ds_read_b128 of the next slot are issued before the 64 packed ops
      // of the current one (two register sets in ping-pong, no copies).  Within a slot all 16 "lo"
      // updates precede all 16 "hi" updates, so dependent v_pk_fma are 30+ instructions apart.
      auto load_slot = [&](Slot& v, int slot) {
#pragma unroll
        for (int a = 0; a < kTileR; ++a) {
          const f32x4 vi = *reinterpret_cast<const f32x4*>(cur + oi[a] + slot * 16);
          const f32x4 vj = *reinterpret_cast<const f32x4*>(cur + oj[a] + slot * 16);
          v.il[a] = f32x2{vi.x, vi.y};
          v.ih[a] = f32x2{vi.z, vi.w};
          v.jl[a] = f32x2{vj.x, vj.y};
          v.jh[a] = f32x2{vj.z, vj.w};
        }
      };
      auto accumulate = [&](const Slot& v) {
#pragma unroll
        for (int a = 0; a < kTileR; ++a)
#pragma unroll
          for (int b = 0; b < kTileR; ++b) {
            const f32x2 lo = v.il[a] - v.jl[b];  // v_pk_add_f32 with neg modifier
            acc[a][b] = __builtin_elementwise_fma(lo, lo, acc[a][b]);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < kTileR; ++a)
#pragma unroll
          for (int b = 0; b < kTileR; ++b) {
            const f32x2 hi = v.ih[a] - v.jh[b];
            acc[a][b] = __builtin_elementwise_fma(hi, hi, acc[a][b]);
          }
        __builtin_amdgcn_sched_barrier(0);
      };
      Slot sa, sb;
      load_slot(sa, 0);
      const int last = g.slots - 1;
#pragma unroll 1
      for (int k = 0; k < g.slots; k += 2) {
        load_slot(sb, k + 1);
        accumulate(sa);
        load_slot(sa, (k + 2) & last);  // the wrap-around read of slot 0 is harmless
        accumulate(sb);
      }
    }
  }
  // ---- combine strips in a fixed order, emit this workgroup's partial (fp64) ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(tiles);  // reuse: [strip][tile][16]
  if (active) {
#pragma unroll
    for (int a = 0; a < kTileR; ++a)
#pragma unroll
      for (int b = 0; b < kTileR; ++b)
        red[(strip * g.tiles + ptile) * 16 + a * kTileR + b] = acc[a][b].x + acc[a][b].y;
  }
  __syncthreads();
  const int per_block = g.tiles * 16;
  for (int e = tid; e < per_block; e += blockDim.x) {
    double s = 0.0;
    for (int st = 0; st < g.strips; ++st) s += (double)red[st * per_block + e];
    partial[(int64_t)blockIdx.x * per_block + e] = s;
  }
  if (sub == nullptr) return;  // whole-stack call: pairwise_reduce_kernel follows on the stream
  // Gated call: the workgroup that arrives LAST adds the partials itself, in the order of pairwise_reduce_kernel
  // (same bits), so that the common case — an empty row list — costs one empty launch instead of two.
  __shared__ double wsum[kRedWaves][64];
  __shared__ int last;
  if (!arrive_last(sub + kDirectArrivalSlot, (int)gridDim.x, &last)) return;
  for (int e0 = 0; e0 < per_block; e0 += 64) {
    for (int q = tid; q < 64 * kRedWaves; q += blockDim.x) {
      const int w = q >> 6, e = e0 + (q & 63);
      double s = 0.0;
      if (e < per_block) {
#pragma unroll 16
        for (int blk = w; blk < (int)gridDim.x; blk += kRedWaves) s += partial[(int64_t)blk * per_block + e];
      }
      wsum[w][q & 63] = s;
    }
    __syncthreads();
    if (tid < 64 && e0 + tid < per_block) {
      double tot = wsum[0][tid];
#pragma unroll
      for (int w = 1; w < kRedWaves; ++w) tot += wsum[w][tid];
      pair_scatter(e0 + tid, tot, g, n_full, sq, sub);
    }
    __syncthreads();
  }
  if (tid == 0) sub[kDirectArrivalSlot] = 0;
  if (rk.on) {
    __threadfence();
    __syncthreads();  // the corrected distances of this workgroup's own stores are visible to all its lanes
    krum_rank_body(sq, n_full, rk.f, rk.m, rk.mode, rk.order, rk.scores, reinterpret_cast<double*>(smem), rk.bitonic != 0);
  }
}

// Cross-workgroup reduction in a fixed order.  A workgroup of 8 waves owns 64 consecutive
// partial entries e = tile*16 + a*4 + b (coalesced 512-byte reads); wave w adds the partials of
// workgroups w, w+8, ... in increasing order, then wave 0 adds the 8 wave sums in order and
// scatters the value to sq[i][j] and sq[j][i].  (Whole-stack calls; the gated call reduces inside
// pairwise_partial_kernel, in the same order.)
__global__ __launch_bounds__(64 * kRedWaves) void pairwise_reduce_kernel(
    const double* __restrict__ partial, int nblocks, PairGeom g, int n_full, double* __restrict__ sq) {
  __shared__ double wsum[kRedWaves][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per_block = g.tiles * 16;
  const int e = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (e < per_block) {
#pragma unroll 8
    for (int blk = wave; blk < nblocks; blk += kRedWaves) s += partial[(int64_t)blk * per_block + e];
  }
  wsum[wave][lane] = s;
  __syncthreads();
  if (wave != 0 || e >= per_block) return;
  double tot = wsum[0][lane];
#pragma unroll
  for (int w = 1; w < kRedWaves; ++w) tot += wsum[w][lane];
  pair_scatter(e, tot, g, n_full, sq, nullptr);
}

static int pair_grid_blocks(const PairGeom& g, int64_t d) {
  const int64_t chunks = (d + g.width - 1) / g.width;
  int blocks = 256 * 4;
  if (blocks > chunks) blocks = (int)(chunks > 0 ? chunks : 1);
  return blocks;
}

// ---------------------------------------------------------------------------
// Score + stable rank in one workgroup.  Distances of a row are ranked by counting (all n^2
// elements in parallel), then lane i adds the `take` smallest of row i in ascending order in
// fp64 — the same sequence of additions as the reference's `sum(sorted(...)[:take])`.
// ---------------------------------------------------------------------------
constexpr int kRankThreads = 1024;
__global__ __launch_bounds__(kRankThreads) void krum_rank_kernel(const double* __restrict__ sq, int n,
                                                                 int f, int m, int mode,
                                                                 int32_t* __restrict__ order,
                                                                 double* __restrict__ scores_out, int bitonic) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // rank_lds_bytes(n)
  krum_rank_body(sq, n, f, m, mode, order, scores_out, reinterpret_cast<double*>(smem), bitonic != 0);
}

}  // namespace bm

namespace bm {
int gram_finish(const double* partial, int blocks, int n, int n_full, double* gram, double* sq_nxn, int* sub, double tau,
                hipStream_t s, const RankArgs* rank = nullptr);
int gram_arrival_slot();
int gram3_partials(const float* const* rows, int n, int64_t d, int64_t d_total, double* partial, int* sub,
                   int* blocks_out, hipStream_t s);
int64_t gram3_partial_doubles(int n);

// Workspace layout of bm_pairwise_sqdist: [row list of the gate: 512 B][Gram partials][Gram n(n+1)/2][direct partials]
static int64_t pair_gram_doubles(int n) { return gram3_partial_doubles(n) + (int64_t)n * (n + 1) / 2; }

// Direct-difference kernel + its reduction.  sub == nullptr: the whole stack, unconditionally.
// Otherwise sub is the DEVICE row list written by gram_to_sqdist_kernel: both launches return
// immediately when it is empty, else recompute exactly the pairs among the listed rows (device-side
// decision, no host synchronisation; the launch is shaped for the worst case, all n rows).
static int pairwise_direct(const float* const* rows, int n, int64_t d, double* sq_nxn, double* partial,
                           int* sub, hipStream_t s, const RankArgs* rank = nullptr) {
  const PairGeom g = pair_geometry(n, 0);
  RowTable tab{};
  for (int i = 0; i < n; ++i) tab.p[i] = rows[i];
  int threads = g.threads, per_block = g.tiles * 16;
  size_t lds_bytes = 0;
  int64_t min_width = g.width;
  // worst case over the sub-stack sizes the device may pick
  for (int k = (sub == nullptr ? n : 2); k <= n; ++k) {
    const PairGeom gk = (k == n) ? g : pair_geometry(k, 0);
    size_t need = (size_t)2 * gk.nb * kDmaPitch;  // two tile buffers
    const size_t red_bytes = (size_t)gk.strips * gk.tiles * 16 * sizeof(float);
    if (red_bytes > need) need = red_bytes;
    if (need > lds_bytes) lds_bytes = need;
    if (gk.threads > threads) threads = gk.threads;
    if (gk.width < min_width) min_width = gk.width;
  }
  lds_bytes += BM_MAX_ROWS * sizeof(float*);  // row pointer table in front
  RankArgs rk{};
  if (rank != nullptr && sub != nullptr) {
    rk = *rank;
    rk.on = 1;
    if (lds_bytes < (size_t)rank_lds_bytes(n)) lds_bytes = rank_lds_bytes(n);  // (the ranking's arrays alias the tiles)
  }
  PairGeom gw = g;
  gw.width = (int)min_width;
  const int blocks = pair_grid_blocks(gw, d);
  const bool aligned =
      common_vec_width(reinterpret_cast<const void* const*>(rows), n, nullptr) == 4;
  auto kern = aligned ? pairwise_partial_kernel<true> : pairwise_partial_kernel<false>;
  // (static: the wave sums of the in-kernel reduction, 4 KB)
  if (const int rc = lds_opt_in(reinterpret_cast<const void*>(kern), lds_bytes, kRedWaves * 64 * sizeof(double))) return rc;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds_bytes, s, tab, g, d, partial, sub, n, sq_nxn, rk);
  BM_LAUNCH_CHECK();
  if (sub == nullptr) {
    hipLaunchKernelGGL(pairwise_reduce_kernel, dim3((per_block + 63) / 64), dim3(64 * kRedWaves), 0, s,
                       partial, blocks, g, n, sq_nxn);
    BM_LAUNCH_CHECK();
  }
  return 0;
}
}  // namespace bm

namespace bm {
// The default distance pass (BM_PAIR_MODE 0): the Gram kernel, the reduction of its partial matrices with the
// squared distances and the accuracy gate's row list (last-arriving workgroup), the gated direct kernel — which
// returns at once unless rows were listed — and, with `rank`, the ranking of the rows inside that third launch.
static int pairwise_gram_path(const float* const* rows, int n, int64_t d, int64_t d_total, double* sq_nxn, void* ws,
                              const RankArgs* rank, hipStream_t s) {
  int* flag = static_cast<int*>(ws);  // flag[0] = rows to recompute, flag[1..] = their indices; counters behind them
  double* gram_partial = reinterpret_cast<double*>(static_cast<char*>(ws) + 512);
  double* direct_partial = gram_partial + pair_gram_doubles(n);
  const double tau = tuning().pair_tau;
  int blocks = 0;
  int rc = gram3_partials(rows, n, d, d_total, gram_partial, flag, &blocks, s);
  if (rc != 0) return rc;
  double* gram = gram_partial + gram3_partial_doubles(n);
  rc = gram_finish(gram_partial, blocks, n, n, gram, sq_nxn, flag, tau, s, rank);  // (ranks when nothing is listed)
  if (rc != 0 || tau <= 0.0) return rc;  // (no gate: nothing is ever listed, no third launch)
  return pairwise_direct(rows, n, d, sq_nxn, direct_partial, flag, s, rank);
}
}  // namespace bm

extern "C" int bm_pairwise_sqdist(const float* const* rows, int n, int64_t d, double* sq_nxn,
                                  void* ws, void* stream) {
  return bm_pairwise_sqdist_shard(rows, n, d, d, sq_nxn, ws, stream);
}

extern "C" int bm_pairwise_sqdist_shard(const float* const* rows, int n, int64_t d, int64_t d_total, double* sq_nxn,
                                        void* ws, void* stream) {
  using namespace bm;
  if (rows == nullptr || sq_nxn == nullptr || ws == nullptr || n < 1 || n > BM_MAX_ROWS || d < 0 || d_total < d)
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* gram_partial = reinterpret_cast<double*>(static_cast<char*>(ws) + 512);
  double* direct_partial = gram_partial + pair_gram_doubles(n);
  // BM_PAIR_MODE: 0 (default) centred Gram on the bf16 matrix cores, error-free split (gram_bf16.hip), ending with the
  //               accuracy check of gram_to_sqdist_kernel: the rows of the pairs it flags are recomputed among
  //               themselves by the direct kernel (decided on the device, no host round trip);
  //               1 direct differences on the VALU (this file) for every pair, no cancellation at all.
  const int mode = tuning().pair_mode;
  if (mode == 1) return pairwise_direct(rows, n, d, sq_nxn, direct_partial, nullptr, s);
  return pairwise_gram_path(rows, n, d, d_total, sq_nxn, ws, nullptr, s);
}

namespace bm {
// The tail of the distance pass for partial Gram matrices some other kernel has left in the workspace (step.hip: the
// first pass of a step contracts its rows itself): `blocks` partials of nc(nc+1)/2 doubles at pairwise_gram_area(ws);
// rows = the n_full rows of the stack, rows nc-1 .. n_full-1 being one row of G.  Same reduction, gate and exact pass
// as bm_pairwise_sqdist.
double* pairwise_gram_area(void* ws) { return reinterpret_cast<double*>(static_cast<char*>(ws) + 512); }
int* pairwise_arrival_counter(void* ws) { return static_cast<int*>(ws) + gram_arrival_slot(); }
int pairwise_from_gram_partials(const float* const* rows, int n_full, int nc, int blocks, int64_t d, double* sq_nxn,
                                void* ws, hipStream_t s) {
  int* flag = static_cast<int*>(ws);
  double* gram_partial = pairwise_gram_area(ws);
  double* direct_partial = gram_partial + pair_gram_doubles(n_full);
  double* gram = gram_partial + gram3_partial_doubles(n_full);
  const double tau = tuning().pair_tau;
  int rc = gram_finish(gram_partial, blocks, nc, n_full, gram, sq_nxn, flag, tau, s);
  if (rc != 0 || tau <= 0.0) return rc;
  return pairwise_direct(rows, n_full, d, sq_nxn, direct_partial, flag, s);
}
}  // namespace bm

extern "C" int bm_krum_rank(const double* sq_nxn, int n, int f, int m, int mode,
                            int32_t* order_out, double* scores_out, void* stream) {
  using namespace bm;
  if (sq_nxn == nullptr || order_out == nullptr || n < 1 || n > BM_MAX_ROWS || f < 0 ||
      (mode != BM_RANK_KRUM && mode != BM_RANK_BULYAN))
    return BM_EINVAL;
  if (const int rc = lds_opt_in(reinterpret_cast<const void*>(krum_rank_kernel), (size_t)rank_lds_bytes(n), 0)) return rc;
  hipLaunchKernelGGL(krum_rank_kernel, dim3(1), dim3(kRankThreads), rank_lds_bytes(n), static_cast<hipStream_t>(stream),
                     sq_nxn, n, f, m, mode, order_out, scores_out, rank_bitonic(n) ? 1 : 0);
  BM_LAUNCH_CHECK();
  return 0;
}

extern "C" int bm_pairwise_rank(const float* const* rows, int n, int64_t d, int64_t d_total, int f, int m, int mode,
                                double* sq_nxn, int32_t* order_out, double* scores_out, void* ws, void* stream) {
  using namespace bm;
  if (rows == nullptr || sq_nxn == nullptr || order_out == nullptr || ws == nullptr || n < 1 || n > BM_MAX_ROWS ||
      d < 0 || d_total < d || f < 0 || (mode != BM_RANK_KRUM && mode != BM_RANK_BULYAN))
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (tuning().pair_mode == 1) {
    const int rc = bm_pairwise_sqdist_shard(rows, n, d, d_total, sq_nxn, ws, stream);
    return rc != 0 ? rc : bm_krum_rank(sq_nxn, n, f, m, mode, order_out, scores_out, stream);
  }
  // The rows are ranked inside the launches of the distance pass: by the last workgroup of the Gram reduction when the
  // accuracy gate lists nothing, by the last workgroup of the gated direct kernel when it does (round 4 ranked in the
  // gated launch in both cases — 2-3 waves, one row at a time: 16.5 us of a 21 us launch at n = 25 — and kept a rank
  // launch of its own beyond 32 rows: 13.4 us at n = 51; profiles/r05_b_full_kernel_trace.csv).
  const RankArgs req{1, f, m, mode, rank_bitonic(n) ? 1 : 0, order_out, scores_out};
  return pairwise_gram_path(rows, n, d, d_total, sq_nxn, ws, &req, s);
}

namespace bm {
int64_t pairwise_workspace_bytes(int n, int64_t d) {
  const PairGeom g = pair_geometry(n, 0);
  const int blocks = 4096;  // upper bound on the grid of the direct kernel
  (void)d;
  const int64_t direct = (int64_t)blocks * g.tiles * 16 * (int64_t)sizeof(double);
  return 512 + pair_gram_doubles(n) * (int64_t)sizeof(double) + direct;
}
}  // namespace bm
