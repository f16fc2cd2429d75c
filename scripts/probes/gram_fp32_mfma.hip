// gram.hip — pairwise squared distances as a dense Gram contraction on the fp32 matrix cores.
//
// Why: the direct form sum_k (a_k-b_k)^2 (pairwise.hip) costs two VALU lane-ops per
// (pair, coordinate).  Measured on MI355X it is VALU-bound: 1.05 ms for n=51 x d=11.2 M
// (27 % of the HBM roofline, VALU ~90 % busy; profiles/r01_b_*).  Cast as G = X X^T with
//     d2(i,j) = G_ii + G_jj - 2 G_ij
// the contraction runs on v_mfma_f32_16x16x4_f32 (exact fp32 products, k-ordered fp32 fma chain,
// 256 FLOP/clk/CU on the matrix pipe) while the VALU stays idle.
//
// Data path (gfx950):
//   * HBM -> LDS by the LDS-DMA engine (global_load_lds_dwordx4, 1 KiB contiguous per wave
//     instruction and row: long DRAM bursts), two tile buffers per workgroup, the DMA of tile
//     t+1 in flight under the MFMAs of tile t.  (Loading the MFMA fragments straight from global
//     memory, 64 B per row and instruction, reached only 2.4-2.9 TB/s.)
//   * LDS -> VGPR: per step of 16 coordinates and per 16-row block ONE ds_read_b128 per lane
//     (lane l = (i = l & 15, q = l >> 4): row 16R+i, coordinates 16s+4q .. +3).  Component t of that
//     register is the A operand (and, for the column block, the B operand) of the t-th MFMA:
//     A[i][k=q] = X[16R+i][16s+4q+t]; a dot product does not care in which order the 16 coordinates
//     are visited as long as rows and columns agree.  RB fragment reads feed RB(RB+1)/2 * 4 MFMAs,
//     so LDS traffic (and its bank conflicts) is negligible.
//   * the waves of a workgroup split the 16-coordinate steps of a tile; a 4-wave fixed-order fp64
//     reduction at the end emits the workgroup's partial Gram (upper triangle only).
//
// Numerics:
//   * two-level fp32 accumulation (16..64-coordinate chains inside the MFMA accumulator, then a
//     per-wave fp32 sum of those chains), fp64 across waves, workgroups and GPUs:
//     error ~3e-9 * (G_ii + G_jj) measured, i.e. ~1e-6 relative on a distance even for rows whose
//     distance is 30x smaller than their norms;
//   * bitwise-equal rows give bitwise-equal G entries (same instruction sequence, same k order),
//     hence d2 = 0 exactly between aliased Byzantine rows and bitwise-equal distances from them
//     to any third row: the exact score ties of the reference survive (krum.py:62 stable sort);
//   * deterministic: fixed coordinate ownership, fixed reduction trees, no atomics.
#include "bm_common.h"

namespace bm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGramWaves = 4;         // waves per workgroup
constexpr int kGramDmaBlock = 1024;   // bytes per wave-wide global_load_lds_dwordx4
constexpr int kGramDmaPitch = 1024 + 16;

// number of 16x16 blocks on and above the diagonal
__host__ __device__ constexpr int gram_pairs(int rb) { return rb * (rb + 1) / 2; }

// compact index of (i, j), i <= j, in an n x n upper triangle stored row-major
__host__ __device__ inline int tri_index(int i, int j, int n) { return i * n - (i * (i - 1)) / 2 + (j - i); }

struct GramGeom {
  int n;
  int row_bytes;   // bytes of one row inside a tile: 256, 512 or 1024
  int width;       // coordinates per tile = row_bytes / 4
  int rows_per_dma;  // rows in one 1 KiB DMA block
  int nb;          // DMA blocks per tile
};

__device__ __forceinline__ int gram_row_offset(const GramGeom& g, int r) {
  return (r / g.rows_per_dma) * kGramDmaPitch + (r % g.rows_per_dma) * g.row_bytes;
}

// s_waitcnt vmcnt(k) with a run-time (wave-uniform) k: "all but the k most recent VMEM ops are done"
__device__ __forceinline__ void wait_vmem_all_but(int k) {
  switch (k) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // conservative
  }
}

// NBUF tile buffers per workgroup: the DMA of tile t+NBUF-1 is issued while tile t is contracted.
template <int RB, int NBUF, bool ALIGNED, int W = kGramWaves>
__global__ __launch_bounds__(64 * W) void gram_partial_kernel(RowTable rows, GramGeom g, int64_t d,
                                                              double* __restrict__ partial) {
  constexpr int NP = gram_pairs(RB);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const float** row_ptr = reinterpret_cast<const float**>(smem);  // 512 B pointer table
  double* red = reinterpret_cast<double*>(smem + BM_MAX_ROWS * sizeof(float*));  // [waves][256]
  char* tiles = smem + BM_MAX_ROWS * sizeof(float*) + W * 256 * sizeof(double);
  const int n = g.n;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lq = lane >> 4;
  const int tile_bytes = (g.nb + 1) * kGramDmaPitch;  // + one block that stays zero

  // Row pointers: one per-lane load from the kernarg segment (the table is the first kernel argument,
  // passed by value).  Round 1 filled the table from ONE thread (~7 us per workgroup) after a dynamic
  // index into the by-value struct (`rows.p[tid]`) had faulted: that form makes the compiler spill
  // the 512-byte struct to scratch; reading the kernarg segment itself needs no copy.
  if (tid < BM_MAX_ROWS) {
    typedef const float* __attribute__((address_space(4))) const* KargTable;
    KargTable karg = (KargTable)__builtin_amdgcn_kernarg_segment_ptr();
    row_ptr[tid] = tid < n ? (const float*)karg[tid] : nullptr;
  }
  for (int o = tid * 16; o < NBUF * tile_bytes; o += blockDim.x * 16)
    *reinterpret_cast<f32x4*>(tiles + o) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  __syncthreads();

  // byte offset of this lane's row in a tile buffer plus its 16-byte column slot; lanes whose
  // row does not exist (r >= n) read the all-zero DMA block appended to every buffer, so the
  // fragment loads need no predication
  int frag_off[RB];
#pragma unroll
  for (int R = 0; R < RB; ++R) {
    const int r = 16 * R + li;
    frag_off[R] = (r < n ? gram_row_offset(g, r) : g.nb * kGramDmaPitch) + 16 * lq;
  }

  f32x4 acc[NP], outer[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) outer[p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  const int width = g.width;
  const int lanes_per_row = g.row_bytes / 16;
  const int dma_w = lane / lanes_per_row;
  const int dma_col = (lane - dma_w * lanes_per_row) * 4;

  // Per-lane source pointers of this wave's DMA pieces (piece k = DMA block wave + k*W), resolved
  // once: inside the loop a piece costs one 64-bit add, no LDS pointer lookup and no wait.
  constexpr int kMaxDma = 8;
  const float* dma_src[kMaxDma];
#pragma unroll
  for (int k = 0; k < kMaxDma; ++k) {
    const int blk = wave + k * W;
    const int r = blk * g.rows_per_dma + dma_w;
    dma_src[k] = (blk < g.nb && r < n) ? row_ptr[r] + dma_col : nullptr;
  }
  auto stage = [&](int64_t base, char* buf) {
    if (ALIGNED && base + width <= d) {
#pragma unroll
      for (int k = 0; k < kMaxDma; ++k) {
        if (wave + k * W < g.nb) {  // wave-uniform
          if (dma_src[k] != nullptr)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(dma_src[k] + base),
                (__attribute__((address_space(3))) void*)(buf + (wave + k * W) * kGramDmaPitch), 16, 0, 0);
        }
      }
    } else {
      // ragged last tile / unaligned rows: plain loads with zero fill, same layout
      const int vpr = width / 4;
      for (int idx = tid; idx < n * vpr; idx += blockDim.x) {
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
        *reinterpret_cast<f32x4*>(buf + gram_row_offset(g, r) + col * 4) = val;
      }
    }
  };

  const int steps = width / 16;  // 16-coordinate steps per tile: 4, 8 or 16
  // DMA instructions this wave issues per tile (wave-uniform): the counted wait below leaves the
  // NBUF-2 most recent tiles in flight.
  int dma_per_tile = 0;
  for (int blk = wave; blk < g.nb; blk += W) ++dma_per_tile;
  int64_t chunk = blockIdx.x;                                 // chunk being contracted
  int64_t ahead = chunk;                                      // next chunk to stage
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b) {
    if (ahead * width < d) stage(ahead * width, tiles + b * tile_bytes);
    ahead += gridDim.x;
  }
  for (int it = 0;; ++it) {
    if (chunk * width >= d) break;
    const char* cur = tiles + (it % NBUF) * tile_bytes;
    char* nxt = tiles + ((it + NBUF - 1) % NBUF) * tile_bytes;
    // tiles still allowed in flight behind `cur`: those staged for chunks it+1 .. it+NBUF-2
    int later = 0;
#pragma unroll
    for (int b = 1; b <= NBUF - 2; ++b)
      if ((chunk + (int64_t)b * gridDim.x) * width < d) ++later;
    wait_vmem_all_but(ALIGNED ? later * dma_per_tile : 0);  // this wave's pieces of `cur` have landed
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): LDS reads of the previous tile
    __builtin_amdgcn_s_barrier();  // ... everyone's have landed, and everyone is done reading `nxt`
    if (ahead * width < d) stage(ahead * width, nxt);  // in flight under the MFMAs below
    ahead += gridDim.x;
    chunk += gridDim.x;

    // This wave's steps of the tile: s = wave, wave+4, ...  The first one starts every accumulator
    // from the inline constant 0 (no zeroing moves), the fragments of step s+4 are read before the
    // MFMAs of step s, and at the end of the tile the chains (<= 64 coordinates) are added to the
    // per-wave fp32 sums unconditionally — no phi copies of 2 x 4*NP registers around a branch.
    f32x4 x[RB], xn[RB];
#pragma unroll
    for (int R = 0; R < RB; ++R) x[R] = *reinterpret_cast<const f32x4*>(cur + frag_off[R] + wave * 64);
    {
#pragma unroll
      for (int R = 0; R < RB; ++R)
        xn[R] = *reinterpret_cast<const f32x4*>(cur + frag_off[R] + ((wave + W) % steps) * 64);
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int p = 0;
#pragma unroll
        for (int I = 0; I < RB; ++I)
#pragma unroll
          for (int J = I; J < RB; ++J) {
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[I][t], x[J][t], t == 0 ? zero : acc[p], 0, 0, 0);
            ++p;
          }
      }
    }
#pragma unroll 1
    for (int s = wave + W; s < steps; s += W) {
#pragma unroll
      for (int R = 0; R < RB; ++R) x[R] = xn[R];
#pragma unroll
      for (int R = 0; R < RB; ++R)
        xn[R] = *reinterpret_cast<const f32x4*>(cur + frag_off[R] + ((s + W) % steps) * 64);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int p = 0;
#pragma unroll
        for (int I = 0; I < RB; ++I)
#pragma unroll
          for (int J = I; J < RB; ++J) {
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[I][t], x[J][t], acc[p], 0, 0, 0);
            ++p;
          }
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) outer[p] += acc[p];
  }

  // ---- workgroup reduction, one 16x16 block at a time, fixed order; compact upper triangle ----
  // C/D layout of the 16x16 MFMA: lane l, register v -> row 4*(l>>4)+v, column l&15.
  const int per_block = n * (n + 1) / 2;
  __syncthreads();
  int p = 0;
#pragma unroll
  for (int I = 0; I < RB; ++I)
#pragma unroll
    for (int J = I; J < RB; ++J) {
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave * 256 + (4 * lq + v) * 16 + li] = (double)outer[p][v];
      __syncthreads();
      {
        if (tid < 256) {
          const int rr = tid >> 4, cc = tid & 15;  // 256 threads <-> 16x16 entries
          double s = red[tid];
#pragma unroll
          for (int w = 1; w < W; ++w) s += red[w * 256 + tid];
          const int gi = 16 * I + rr, gj = 16 * J + cc;
          if (gi <= gj && gj < n) partial[(int64_t)blockIdx.x * per_block + tri_index(gi, gj, n)] = s;
        }
      }
      __syncthreads();
      ++p;
    }
}

// G = sum over workgroups (fixed order), then sq[i][j] = G_ii + G_jj - 2 G_ij in fp64.
constexpr int kGramRedWaves = 16;  // 16 waves x 16 loads in flight: the sum is a latency chain over L2/HBM
__global__ __launch_bounds__(64 * kGramRedWaves) void gram_reduce_kernel(const double* __restrict__ partial,
                                                                         int nblocks, int n,
                                                                         double* __restrict__ gram) {
  __shared__ double wsum[kGramRedWaves][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per_block = n * (n + 1) / 2;
  const int e = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (e < per_block) {
#pragma unroll 16
    for (int blk = wave; blk < nblocks; blk += kGramRedWaves) s += partial[(int64_t)blk * per_block + e];
  }
  wsum[wave][lane] = s;
  __syncthreads();
  if (wave != 0 || e >= per_block) return;
  double tot = wsum[0][lane];
#pragma unroll
  for (int w = 1; w < kGramRedWaves; ++w) tot += wsum[w][lane];
  gram[e] = tot;
}

// sq[i][j] = G_ii + G_jj - 2 G_ij in fp64, one workgroup.  Also decides whether the Gram form was
// accurate enough: its absolute error is ~eps_G * (G_ii + G_jj) (eps_G ~ 6e-9, measured), so a pair
// whose squared distance is below tau * (G_ii + G_jj) — two rows that nearly coincide relative to
// their (centred) norms — has lost relative accuracy eps_G / tau.  The rows of such pairs are listed in
// `sub` (sub[0] = count, sub[1..] = indices, ascending); the caller recomputes the distances among them
// with the direct-difference kernel (pairwise.hip), which has no cancellation: near-duplicate rows
// lie close to EACH OTHER, so that sub-stack is exactly where the Gram form cannot be trusted.
// Bitwise-equal rows (G_ii == G_jj == G_ij) are exact (d2 = 0) and never listed.
constexpr int kSqThreads = 1024;
__global__ __launch_bounds__(kSqThreads) void gram_to_sqdist_kernel(const double* __restrict__ gram, int n,
                                                                    double tau, double* __restrict__ sq,
                                                                    int* __restrict__ sub) {
  __shared__ int listed[BM_MAX_ROWS];
  if (threadIdx.x < BM_MAX_ROWS) listed[threadIdx.x] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < n * n; e += kSqThreads) {
    const int i = e / n, j = e - i * n;
    if (i == j) {
      // a row with a non-finite coordinate is at non-finite distance of everything, itself
      // included in the reference (x - x = nan); keep 0 on the diagonal, it is never read
      sq[e] = 0.0;
      continue;
    }
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const double gii = gram[tri_index(lo, lo, n)], gjj = gram[tri_index(hi, hi, n)];
    const double gij = gram[tri_index(lo, hi, n)];
    double v = (gii + gjj) - 2.0 * gij;
    const bool same = (gii == gjj) && (gij == gii);
    if (!same && v < tau * (gii + gjj)) {  // NaN compares false: non-finite rows are never listed
      listed[i] = 1;                       // benign race: every writer stores 1
      listed[j] = 1;
    }
    if (v < 0.0) v = 0.0;  // rounding of nearly identical rows; NaN stays NaN
    sq[e] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0 && sub != nullptr) {
    int count = 0;
    for (int r = 0; r < n; ++r)
      if (listed[r]) sub[1 + count++] = r;
    sub[0] = count;
  }
}

constexpr int kGramMaxBlocks = 2048;

int64_t gram_partial_doubles(int n) { return (int64_t)kGramMaxBlocks * ((int64_t)n * (n + 1) / 2); }

static GramGeom gram_geometry(int n) {
  GramGeom g;
  g.n = n;
  // 256 coordinates per row and tile for few rows, 128 otherwise (measured best of 64/128/256 at
  // n = 25 and n = 51; BM_PAIR_STRIPS = 256|512|1024 forces the row bytes for experiments)
  const int forced = tuning().pair_strips;
  g.row_bytes = (forced == 256 || forced == 512 || forced == 1024) ? forced : (n <= 24 ? 1024 : 512);
  g.width = g.row_bytes / 4;
  g.rows_per_dma = kGramDmaBlock / g.row_bytes;
  g.nb = (n + g.rows_per_dma - 1) / g.rows_per_dma;
  return g;
}

template <int RB>
static int launch_gram(const RowTable& tab, const GramGeom& g, int64_t d, bool aligned, double* partial,
                       int blocks, hipStream_t s) {
  // Ring depth and workgroup width.  Deep ring (BM_PAIR_NBUF = 4 or 5 with 8-wave workgroups): the
  // LDS-DMA of tile t+NBUF-1 is issued while tile t is contracted, so several tiles (tens of KB per
  // CU) stay in flight — everything in flight through the DMA engine needs LDS behind it.
  int nbuf = tuning().pair_nbuf;
  if (nbuf < 2 || nbuf > 5) nbuf = 2;
  const bool wide = (nbuf >= 4 || tuning().pair_ablate == 8) && g.width / 16 >= 8;
  if (nbuf >= 4 && !wide) nbuf = 3;
  const int waves = wide ? 8 : kGramWaves;
  if ((g.nb + waves - 1) / waves > 8) return BM_EINVAL;  // kMaxDma pieces per wave and tile
  const size_t lds = BM_MAX_ROWS * sizeof(float*) + waves * 256 * sizeof(double) +
                     (size_t)nbuf * (g.nb + 1) * kGramDmaPitch;
  if (lds > 160 * 1024) return BM_EINVAL;
  void (*kern)(RowTable, GramGeom, int64_t, double*);
  if (!aligned) {
    kern = gram_partial_kernel<RB, 2, false>;
    nbuf = 2;
  } else if (wide) {
    kern = nbuf == 5 ? gram_partial_kernel<RB, 5, true, 8>
                     : (nbuf == 4 ? gram_partial_kernel<RB, 4, true, 8>
                                  : (nbuf == 3 ? gram_partial_kernel<RB, 3, true, 8>
                                               : gram_partial_kernel<RB, 2, true, 8>));
  } else {
    kern = nbuf == 3 ? gram_partial_kernel<RB, 3, true> : gram_partial_kernel<RB, 2, true>;
  }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_code(e);
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), lds, s, tab, g, d, partial);
  BM_LAUNCH_CHECK();
  return 0;
}

// Fixed-order sum of the per-workgroup partial Gram matrices, then squared distances + accuracy flag.
int gram_finish(const double* partial, int blocks, int n, double* gram, double* sq_nxn, int* sub, double tau,
                hipStream_t s) {
  const int64_t per_block = (int64_t)n * (n + 1) / 2;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((int)((per_block + 63) / 64)), dim3(64 * kGramRedWaves), 0, s,
                     partial, blocks, n, gram);
  BM_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_to_sqdist_kernel, dim3(1), dim3(kSqThreads), 0, s, gram, n, tau, sq_nxn, sub);
  BM_LAUNCH_CHECK();
  return 0;
}

int gram_sqdist(const float* const* rows, int n, int64_t d, double* sq_nxn, double* partial, double* gram,
                int* sub, double tau, hipStream_t s) {
  RowTable tab{};
  for (int i = 0; i < n; ++i) tab.p[i] = rows[i];
  const bool aligned = common_vec_width(reinterpret_cast<const void* const*>(rows), n, nullptr) == 4;
  const GramGeom g = gram_geometry(n);
  const int64_t chunks = (d + g.width - 1) / g.width;
  int blocks = tuning().pair_blocks > 0 ? tuning().pair_blocks : 256 * 4;
  if (blocks > kGramMaxBlocks) blocks = kGramMaxBlocks;
  if (blocks > chunks) blocks = (int)(chunks > 0 ? chunks : 1);
  const int rb = (n + 15) / 16;
  int rc;
  switch (rb) {
    case 1: rc = launch_gram<1>(tab, g, d, aligned, partial, blocks, s); break;
    case 2: rc = launch_gram<2>(tab, g, d, aligned, partial, blocks, s); break;
    case 3: rc = launch_gram<3>(tab, g, d, aligned, partial, blocks, s); break;
    default: rc = launch_gram<4>(tab, g, d, aligned, partial, blocks, s); break;
  }
  if (rc != 0) return rc;
  return gram_finish(partial, blocks, n, gram, sq_nxn, sub, tau, s);
}

}  // namespace bm
