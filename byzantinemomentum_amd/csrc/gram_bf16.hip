// gram_bf16.hip — pairwise squared distances as a CENTRED Gram contraction on the bf16 matrix
// cores with a three-way split of every fp32 value (default path of bm_pairwise_sqdist).
//
// Replaces the per-pair loop `gradients[x].sub(gradients[y]).norm().item()` of
// aggregators/krum.py:41-48, bulyan.py:48-54, brute.py:43-45.
//
// Why this shape (measured history in DESIGN.md §4.2):
//   * the direct form (pairwise.hip) is VALU-bound, 1.05 ms at n=51 x d=11.2 M;
//   * the fp32-MFMA Gram (gram.hip) is matrix-pipe bound at n=51 (fp32 MFMA = the fp32 vector
//     rate: 364 us floor > the 285 us HBM floor) and, staged through LDS-DMA double buffers, keeps
//     only ~50 KB per CU in flight (4.6 TB/s at n=25);
//   * here every fp32 value x is split EXACTLY into three bf16 numbers x = h + m + l
//     (h = rne_bf16(x), m = rne_bf16(x - h), l = x - h - m, which has <= 8 significant bits) and
//     x_i*x_j is taken as hh + mm + (hm + hl) + (mh + lh) on v_mfma_f32_16x16x32_bf16: 6 MFMAs at
//     16x the fp32-MFMA rate.  The dropped terms (ml, lm, ll) are <= 2^-25 relative to |x_i x_j|
//     each with random signs.  The matrix pipe is then ~45 % busy at n=51 and the kernel is
//     bound by the HBM stream.
//
// Data path (gfx950), one WAVE = one independent stream processor (no workgroup barrier in the
// main loop):
//   * HBM -> VGPR: global_load_dwordx4, lane (rho = l>>4, x = l&15) reads coordinates 4x..4x+3 of
//     row 4k+rho for k = 0..K-1 (K = ceil(n/4)): 256 contiguous bytes per row and instruction.  The
//     bytes in flight live in the VGPR file (512 KB per CU) exactly as in the column kernels —
//     LDS-DMA needs LDS behind every byte in flight and LDS is what ran out;
//   * centring: distances are translation invariant, so the per-coordinate mean over the n rows
//     (sum over k in registers + two cross-lane adds) is subtracted before the split.  G then holds
//     inner products of DEVIATIONS: the cancellation in G_ii + G_jj - 2 G_ij is bounded by
//     (|x_i-c|^2 + |x_j-c|^2) / |x_i-x_j|^2 instead of (|x_i|^2+|x_j|^2)/|x_i-x_j|^2, which is what
//     made the uncentred Gram lose small distances between rows with a large common component
//     (worker momentum late in training).  What centring cannot fix (far outliers next to a tight
//     cluster) is detected by gram_to_sqdist_kernel and recomputed by the direct kernel;
//   * split in registers (5.5 VALU ops per value, every lane busy), three bf16 planes written to a
//     WAVE-PRIVATE LDS region with ds_write_b64 ([plane][row][64 coords], 128-byte rows, 16-byte
//     slots XOR-swizzled by (row>>1)&7: writes and fragment reads are bank-conflict free);
//   * LDS -> MFMA operands: one ds_read_b128 per (plane, 16-row block, 32-coordinate step):
//     lane (i = l&15, g = l>>4) gets 8 consecutive bf16 of row 16R+i = the A operand of block R
//     and, for the column block, the B operand;
//   * per step and block pair three independent accumulators S0 = hh+mm, S1 = hm+hl, S2 = mh+lh
//     start from zero and are flushed as outer += S0 + (S1 + S2): exchanging the roles of the
//     two rows exchanges S1 and S2, so G_ij is bitwise symmetric in (i, j), and bitwise-equal rows
//     produce bitwise-equal G entries: d2 = 0 exactly between aliased Byzantine rows and bitwise
//     equal distances from them to any third row (the exact score ties of the reference survive,
//     krum.py:62 stable sort);
//   * fp32 chains cover 32 coordinates, per-wave fp32 sums ~100 chunks, everything wider is fp64
//     (workgroup, grid, GPUs) in a fixed order: deterministic, no atomics.
#include "gram_split.h"
#include "rank_body.h"

namespace bm {

constexpr int kB3Waves = 4;      // waves per workgroup (they only meet in the final reduction)
constexpr int kB3Chunk = 64;     // coordinates per wave and chunk: 256 B per row
constexpr int kB3RowBytes = 128; // one row of one bf16 plane in LDS

__host__ __device__ constexpr int b3_pairs(int rb) { return rb * (rb + 1) / 2; }

// workgroups per CU by shape (3 leave 168 VGPRs, 2 leave 256); the host sizes its grid with the same function
__host__ __device__ constexpr int b3_workgroups_per_cu(int K, int NPL) {
  return (K <= 6 || (K == 7 && NPL == 2) || (K == 8 && NPL == 3)) ? 3 : 2;
}

// K = ceil(n/4) load instructions per chunk; RB = ceil(K/4) 16-row blocks; NPL bf16 planes (3 = the
// exact split, 2 = h + rne_bf16(x - h): 16 significant bits, see gram3_partials).
template <int K, int NPL>
struct B3Shape {
  static constexpr int RB = (K + 3) / 4;
  static constexpr int NP = b3_pairs(RB);
  static constexpr int N4 = 4 * K;                      // LDS rows per plane
  static constexpr int PS = N4 * kB3RowBytes;           // plane stride
  static constexpr int WS = NPL * PS;                   // wave region
  static constexpr int NSETS = (K <= 8 && (NPL == 2 || K * NPL <= 21)) ? 2 : 1;  // register sets of loads in flight
  // workgroups per CU aimed at (3 leave 168 VGPRs: K = 8 with two planes and K = 7 with three no longer fit them
  // since the m m products have an accumulator of their own)
  static constexpr int MINW = b3_workgroups_per_cu(K, NPL);
  // fold the accumulators of a block pair behind the MFMAs of the next one (needs 16 more VGPRs)
  static constexpr bool PIPE = !(NPL == 3 && K >= 15);
  static constexpr int kPtrBytes = BM_MAX_ROWS * 8;
  static constexpr int kRedBytes = kB3Waves * 256 * 8;
  static constexpr int kLds = kPtrBytes + (kB3Waves * WS > kRedBytes ? kB3Waves * WS : kRedBytes);
};

// NT: the 16-byte row loads carry the non-temporal hint (the default: every byte is read once by this kernel).  The
// NT = false instances (BM_PAIR_LOAD_NT=0, aligned rows only, experiments) load with the default policy: the second
// pass of Krum / Bulyan reads the same rows again, starting where this kernel finished, and whether the Infinity Cache
// still holds that tail may depend on the hint — unmeasured.  The NT = true code is the code it always was.
template <int K, int NPL, bool ALIGNED, bool NT = true>
__global__ __launch_bounds__(64 * kB3Waves, (B3Shape<K, NPL>::MINW)) void gram3_partial_kernel(
    RowTable rows, int n, int64_t d, float inv_n, int centre, unsigned dither_seed, double* __restrict__ partial,
    int* __restrict__ arrival, int steady) {
  using S = B3Shape<K, NPL>;
  constexpr int RB = S::RB, NP = S::NP, NSETS = S::NSETS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const float** row_ptr = reinterpret_cast<const float**>(smem);
  const int tid = threadIdx.x;
  // (the wave index as a SCALAR: the chunk counters gw / c / nw and every loop condition then live in SGPRs — 20 VALU
  //  instructions fewer per pair of chunks at K = 7, 8-24 VGPRs fewer, and the K = 7 two-plane instance no longer
  //  parks `wave` in scratch across its main loop at the 168 VGPRs of three workgroups per CU)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  char* wbase = smem + S::kPtrBytes + wave * S::WS;

  // Row pointers: one per-lane load from the kernarg segment (the table is the first kernel
  // argument, passed by value) instead of a serial fill by one thread.
  if (tid < BM_MAX_ROWS) {
    typedef const float* __attribute__((address_space(4))) const* KargTable;
    KargTable karg = (KargTable)__builtin_amdgcn_kernarg_segment_ptr();
    // (slots past the stack point at its LAST row: the one load instruction that can reach a padded row — k = K - 1,
    //  rows n .. 4K-1 — then needs no clamp of its row index, a loop invariant the K = 6 three-plane instance parked
    //  in scratch; what a padded row holds is discarded at the end: gj < n)
    row_ptr[tid] = (const float*)karg[tid < n ? tid : n - 1];
  }
  // the arrival counter of this call's reduction (gram_reduce_sqdist_kernel, next on the stream) starts from zero
  if (blockIdx.x == 0 && tid == 0 && arrival != nullptr) *arrival = 0;
  __syncthreads();

  // ---- load side: lane (rho, x) owns coordinates 4x..4x+3 of rows 4k+rho ----
  const int rho = lane >> 4, x = lane & 15;
  const bool last_valid = (4 * (K - 1) + rho) < n;  // only the last instruction can hold a padded row
  // LDS write address of instruction k: row 4k+rho, 16-byte slot (x>>1) ^ ((row>>1)&7), half x&1.
  // (row>>1)&7 = ((2k)&7) | (rho>>1): the k part is a compile-time XOR on bits 4..6.
  const int wr_lane = rho * kB3RowBytes + ((((x >> 1) ^ (rho >> 1)) & 7) << 4) + ((x & 1) << 3);

  // ---- MFMA side: lane (i, g) reads row 16R+i, slot (4s+g) ^ ((row>>1)&7) ----
  const int li = lane & 15, lg = lane >> 4;
  const int rd0 = li * kB3RowBytes + (((lg ^ (li >> 1)) & 7) << 4);  // step 0; step 1 flips bit 6
  // last block: rows >= N4 do not exist in LDS, read row 0 instead (their products are discarded)
  const bool last_ok = (16 * (RB - 1) + li) < S::N4;
  const int rd_last0 = last_ok ? rd0 + (RB - 1) * 16 * kB3RowBytes : (lg << 4);

  // Per-wave running sums of the chunk results, fp32 over 85-280 chunks, DITHERED: acc <- acc + fma(acc, rc, t) with
  // rc a zero-mean pseudo-random number of the order of one ulp, a function of the chunk index alone.  Why: on rows
  // of few distinct values every chunk adds (nearly) the same t, the rounding of acc + t has the same sign step after
  // step and in all 2 048 waves at once, and the error grows like 2^-25 * chunks instead of averaging out —
  // measured in round 3 on exactly constant rows: 1.2e-4 on a squared distance that cancels 200-fold at d = 11.2 M,
  // 1.7e-5 at 2.5 M, nothing at 2^20.  The dither (triangular, two uniform terms of 1..2 ulp each: the bias that
  // is left is below 5 % of the undithered one whatever the binade position) makes the roundings of a wave
  // independent of each other and of the other waves', at the price of one v_fma per accumulator and chunk; every
  // entry of a chunk sees the same rc, so bitwise-equal rows keep bitwise-equal sums.
  using Acc = float;
  Acc outer[NP][4];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int v = 0; v < 4; ++v) outer[p][v] = (Acc)0;

  const int64_t nchunks = (d + kB3Chunk - 1) / kB3Chunk;
  const int64_t gw = (int64_t)blockIdx.x * kB3Waves + wave;
  const int64_t nw = (int64_t)gridDim.x * kB3Waves;

  f32x4 xs[NSETS][K];

  auto issue = [&](int64_t c, f32x4 (&v)[K]) {
    const int64_t coord = c * kB3Chunk + 4 * x;
    if (ALIGNED && (c + 1) * kB3Chunk <= d) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int r = 4 * k + rho;  // (rows >= n: the table repeats row n - 1)
        if constexpr (NT)
          v[k] = __builtin_nontemporal_load(reinterpret_cast<const GlobalF4*>((GlobalF)row_ptr[r] + coord));
        else
          v[k] = *reinterpret_cast<const GlobalF4*>((GlobalF)row_ptr[r] + coord);
      }
    } else {
      // ragged last chunk / rows that are not 16-byte aligned: guarded scalar loads, zero fill
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int r = 4 * k + rho;  // (rows >= n: the table repeats row n - 1)
        GlobalF src = (GlobalF)row_ptr[r] + coord;
        const int64_t left = d - coord;
        f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
        if (left > 0) t.x = src[0];
        if (left > 1) t.y = src[1];
        if (left > 2) t.z = src[2];
        if (left > 3) t.w = src[3];
        v[k] = t;
      }
    }
  };

  // Probe rows of the median-of-three centre: three rows spread over the stack, all held by the
  // rho = 0 lanes (rows 0, 4*(K/3), 4*(2K/3)); tiny stacks take rows 0, 1, 2 / 0, 2, 4.
  constexpr int kPa = 0, kPb = (K >= 3) ? K / 3 : 0, kPc = (K >= 3) ? (2 * K) / 3 : (K == 2 ? 1 : 0);
  const int src_a = x;
  const int src_b = (K >= 3) ? x : (K == 2 ? 32 + x : (n >= 3 ? 16 + x : x));
  const int src_c = (K >= 3) ? x : (K == 2 ? x : (n >= 3 ? 32 + x : x));

  auto contract = [&](f32x4 (&v)[K], int64_t chunk) {
    // -- per-coordinate centre (distances are translation invariant; any finite vector is legal) --
    f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
    if (centre == 2) {
      // median of three rows: stays inside the honest cluster as long as at most one of the three
      // probes is an outlier, whatever the outliers' magnitude (the mean does not)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = __shfl(v[kPa][e], src_a, 64);
        const float b = __shfl(v[kPb][e], src_b, 64);
        const float cc = __shfl(v[kPc][e], src_c, 64);
        const float s = __builtin_amdgcn_fmed3f(a, b, cc);
        c[e] = (__builtin_fabsf(s) < __builtin_inff()) ? s : 0.0f;
      }
    } else if (centre == 1) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (k == K - 1) {
          const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
          c += last_valid ? v[k] : z;
        } else {
          c += v[k];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = c[e];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        s *= inv_n;
        // a non-finite coordinate in ONE row must not poison the other rows' distances
        c[e] = (__builtin_fabsf(s) < __builtin_inff()) ? s : 0.0f;
      }
    }
    // -- split and store the planes --
    unsigned d01 = 0, d23 = 0;
    if constexpr (NPL == 2) {  // the dither of this lane's four coordinates, shared by all K rows
      const unsigned coord = (unsigned)chunk * (unsigned)kB3Chunk + 4u * (unsigned)x;
      // dither_seed == ~0u (BM_PAIR_DITHER=-1, experiments): half an ulp for everyone = round to nearest
      d01 = dither_seed == ~0u ? 0x80008000u : dither_pair(coord + dither_seed);
      d23 = dither_seed == ~0u ? 0x80008000u : dither_pair(coord + 2u + dither_seed);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      u32x2 h, m, l;
      if constexpr (NPL == 2)
        split2_dithered(v[k] - c, d01, d23, h, m);
      else
        split3(v[k] - c, h, m, l);
      char* dst = wbase + (wr_lane ^ (((2 * k) & 7) << 4)) + k * 4 * kB3RowBytes;
      *reinterpret_cast<u32x2*>(dst) = h;
      *reinterpret_cast<u32x2*>(dst + S::PS) = m;
      if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(dst + 2 * S::PS) = l;
    }
  };

  // Fragments of both 32-coordinate steps are read first; every block pair then runs its MFMAs of
  // both steps into the same three accumulators, which are folded into the per-wave fp32 sums once
  // per chunk (64 coordinates).
  // The matrix core does not round its sum to nearest: addends far below the accumulator lose their low bits when they
  // are aligned (scripts/probes/mfma_round_probe.hip), a bias with the sign of the addend.  With products of random
  // sign it averages out; with rows of few distinct values (all products of a chunk equal) it is systematic, 1e-7
  // of a Gram entry, i.e. 1e-4 of a squared distance that cancels 500-fold just above the accuracy gate (measured,
  // round 3).  So every accumulator only ever sums products of ONE magnitude class — S0 = h h, S3 = m m, S1 = h m
  // (+ h l), S2 = m h (+ l h) — sums that are exact for such rows, and the classes meet in round-to-nearest VALU
  // additions: t = S0 + ((S1 + S2) + S3), symmetric under the exchange of the two rows like S0 + (S1 + S2) was.
  auto fold = [](Acc (&acc)[4], const f32x4 a0, const f32x4 a1, const f32x4 a2, const f32x4 a3, const float rc) {
    const f32x4 t = a0 + ((a1 + a2) + a3);
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = acc[v] + __builtin_fmaf(acc[v], rc, t[v]);
  };
  auto multiply = [&](int64_t chunk) {
    // rc = (u1 + u2 - 1) * 2^-23 with u1, u2 uniform on [0, 1): 16 bits each from one mix of the chunk index
    const unsigned z = dither_pair((unsigned)chunk * 2u + 0x3C6EF372u);
    const float rc = (float)((int)(z & 0xffffu) + (int)(z >> 16) - 65536) * 0x1.0p-39f;
    u32x4 fh[2][RB], fm[2][RB], fl[2][RB];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int R = 0; R < RB; ++R) {
        const int off = ((R == RB - 1) ? rd_last0 : rd0 + R * 16 * kB3RowBytes) ^ (s << 6);
        fh[s][R] = *reinterpret_cast<const u32x4*>(wbase + off);
        fm[s][R] = *reinterpret_cast<const u32x4*>(wbase + off + S::PS);
        if constexpr (NPL == 3) fl[s][R] = *reinterpret_cast<const u32x4*>(wbase + off + 2 * S::PS);
      }
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    // Software pipeline over the block pairs: the MFMAs of pair p are issued before the VALU folds of pair p-1, so
    // the folds never wait for the matrix pipe; inside a pair the four chains alternate, dependent MFMAs are at
    // least two issue slots apart.
    f32x4 q0 = zero, q1 = zero, q2 = zero, q3 = zero;  // accumulators of the previous pair, not folded yet
    int p = 0;
#pragma unroll
    for (int I = 0; I < RB; ++I)
#pragma unroll
      for (int J = I; J < RB; ++J) {
        f32x4 s0 = mfma_bf16(fh[0][I], fh[0][J], zero);
        f32x4 s1 = mfma_bf16(fh[0][I], fm[0][J], zero);
        f32x4 s3 = mfma_bf16(fm[0][I], fm[0][J], zero);
        f32x4 s2 = mfma_bf16(fm[0][I], fh[0][J], zero);
        if constexpr (NPL == 3) {
          s1 = mfma_bf16(fh[0][I], fl[0][J], s1);
          s2 = mfma_bf16(fl[0][I], fh[0][J], s2);
        }
        s0 = mfma_bf16(fh[1][I], fh[1][J], s0);
        s1 = mfma_bf16(fh[1][I], fm[1][J], s1);
        s3 = mfma_bf16(fm[1][I], fm[1][J], s3);
        s2 = mfma_bf16(fm[1][I], fh[1][J], s2);
        if constexpr (NPL == 3) {
          s1 = mfma_bf16(fh[1][I], fl[1][J], s1);
          s2 = mfma_bf16(fl[1][I], fh[1][J], s2);
        }
        if constexpr (S::PIPE) {
          if (p > 0) fold(outer[p - 1], q0, q1, q2, q3, rc);
          q0 = s0;
          q1 = s1;
          q2 = s2;
          q3 = s3;
        } else {
          fold(outer[p], s0, s1, s2, s3, rc);
        }
        ++p;
      }
    if constexpr (S::PIPE) fold(outer[NP - 1], q0, q1, q2, q3, rc);
  };

  // ---- main loop: loads of the next chunk(s) stay in flight under the MFMAs of this one ----
  int64_t c = gw;
  if constexpr (ALIGNED) {
    // Steady state, free of conditions.  The compiler places `s_waitcnt vmcnt(N)` from what it can PROVE is in flight:
    // with the refill behind `if (nxt < nchunks)` (the generic loop below) it has to assume the path that issued
    // nothing, so the first use of set b waited for everything issued after it as well — the refill of the other set
    // had one `multiply` (24 MFMAs at K = 7) to arrive instead of a whole iteration, and the two register sets bought
    // nothing (round 3: K = 7 at 0.73 of the HBM peak, K = 13 with ONE set at 0.84).  Here every chunk index is a full
    // chunk by the loop bound, every issue is unconditional, and the counts come out exact: set b is consumed with
    // the NSETS - 1 younger sets still in flight.  Same chunks, same order per wave as the generic loop: same bits.
    const int64_t full = d / kB3Chunk;  // chunks that lie entirely inside the row
    auto issue_fast = [&](int64_t cc, f32x4 (&v)[K]) {
      const int64_t coord = cc * kB3Chunk + 4 * x;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int r = 4 * k + rho;  // (rows >= n: the table repeats row n - 1)
        if constexpr (NT)
          v[k] = __builtin_nontemporal_load(reinterpret_cast<const GlobalF4*>((GlobalF)row_ptr[r] + coord));
        else
          v[k] = *reinterpret_cast<const GlobalF4*>((GlobalF)row_ptr[r] + coord);
      }
    };
    if (steady != 0 && c + (2 * NSETS - 1) * nw < full) {  // (steady == 0: BM_GRAM_STEADY=0, the A/B against the generic loop)
#pragma unroll
      for (int b = 0; b < NSETS; ++b) issue_fast(c + b * nw, xs[b]);
      do {
#pragma unroll
        for (int b = 0; b < NSETS; ++b) {
          contract(xs[b], c + b * nw);
          issue_fast(c + (NSETS + b) * nw, xs[b]);
          multiply(c + b * nw);
        }
        c += NSETS * nw;
      } while (c + (2 * NSETS - 1) * nw < full);
      // drain: the chunks c + b nw were loaded by the last iteration; what is left after them is the generic loop's
#pragma unroll
      for (int b = 0; b < NSETS; ++b) {
        contract(xs[b], c + b * nw);
        multiply(c + b * nw);
      }
      c += NSETS * nw;
    }
  }
#pragma unroll
  for (int b = 0; b < NSETS; ++b)
    if (c + b * nw < nchunks) issue(c + b * nw, xs[b]);
  while (c < nchunks) {
#pragma unroll
    for (int b = 0; b < NSETS; ++b) {
      if (c < nchunks) {  // wave-uniform
        contract(xs[b], c);
        const int64_t nxt = c + NSETS * nw;
        if (nxt < nchunks) issue(nxt, xs[b]);
        multiply(c);
      }
      c += nw;
    }
  }

  // ---- workgroup reduction, one 16x16 block at a time, fixed order; compact upper triangle ----
  // C/D layout of the 16x16 MFMA: lane l, register v -> row 4*(l>>4)+v, column l&15.
  double* red = reinterpret_cast<double*>(smem + S::kPtrBytes);  // [waves][256], aliases the planes
  const int per_block = n * (n + 1) / 2;
  __syncthreads();
  int p = 0;
#pragma unroll
  for (int I = 0; I < RB; ++I)
#pragma unroll
    for (int J = I; J < RB; ++J) {
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave * 256 + (4 * lg + v) * 16 + li] = (double)outer[p][v];
      __syncthreads();
      if (tid < 256) {
        const int rr = tid >> 4, cc = tid & 15;
        double s = red[tid];
#pragma unroll
        for (int w = 1; w < kB3Waves; ++w) s += red[w * 256 + tid];
        const int gi = 16 * I + rr, gj = 16 * J + cc;
        if (gi <= gj && gj < n) partial[(int64_t)blockIdx.x * per_block + b3_tri_index(gi, gj, n)] = s;
      }
      __syncthreads();
      ++p;
    }
}

// sq[i][j] = G_ii + G_jj - 2 G_ij in fp64, by one workgroup of kSqThreads lanes.  Also decides whether the Gram form
// was accurate enough: its absolute error is ~eps_G * (G_ii + G_jj) (eps_G ~ 6e-9, measured), so a pair
// whose squared distance is below tau * (G_ii + G_jj) — two rows that nearly coincide relative to
// their (centred) norms — has lost relative accuracy eps_G / tau.  The rows of such pairs are listed in
// `sub` (sub[0] = count, sub[1..] = indices, ascending); the caller recomputes the distances among them
// with the direct-difference kernel (pairwise.hip), which has no cancellation: near-duplicate rows
// lie close to EACH OTHER, so that sub-stack is exactly where the Gram form cannot be trusted.
// Bitwise-equal rows (G_ii == G_jj == G_ij) are exact (d2 = 0) and never listed.
constexpr int kSqThreads = 1024;
constexpr int kArrivalSlot = 96;  // int slot of the 512-byte row-list area that counts the workgroups of the reduction
// n rows in G (compact), n_full >= n rows in sq: rows n-1 .. n_full-1 of the full stack are ONE row of G (the aliased
// Byzantine copies of a step: the Gram kernel contracted the row once); they are at distance exactly 0 of each other
// and share every other distance, and if the gate lists one of them it lists them all.
// dist (LDS, may be NULL): also receives the n_full x n_full DISTANCES as the ranking reads them (rank_body.h).
__device__ __forceinline__ void gram_to_sqdist(const double* __restrict__ gram, int n, double tau,
                                               double* __restrict__ sq, int* __restrict__ sub, int* listed, int n_full,
                                               double* dist = nullptr) {
  if (threadIdx.x < BM_MAX_ROWS) listed[threadIdx.x] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < n_full * n_full; e += kSqThreads) {
    const int i = e / n_full, j = e - i * n_full;
    const int ci = i < n ? i : n - 1, cj = j < n ? j : n - 1;
    if (ci == cj) {
      // the diagonal (never read: 0) and pairs of aliased copies: exactly 0 when the row is finite; a row with a
      // non-finite coordinate is at non-finite distance of everything, its own copies included (x - x = nan in
      // krum.py:44-47, which the rules turn into +inf; bm_pairwise_sqdist on the expanded stack says NaN too)
      const double gdd = gram[b3_tri_index(ci, ci, n)];
      const double same_row = (i != j && !(fabs(gdd) < __builtin_inf())) ? __builtin_nan("") : 0.0;
      sq[e] = same_row;
      if (dist != nullptr) dist[e] = rank_distance(same_row);
      continue;
    }
    const int lo = ci < cj ? ci : cj, hi = ci < cj ? cj : ci;
    const double gii = gram[b3_tri_index(lo, lo, n)], gjj = gram[b3_tri_index(hi, hi, n)];
    const double gij = gram[b3_tri_index(lo, hi, n)];
    double v = (gii + gjj) - 2.0 * gij;
    const bool same = (gii == gjj) && (gij == gii);
    // (tau <= 0 disables the gate: nothing may be listed then — rounding can leave v slightly negative — because the
    //  caller launches no exact pass and this launch must rank, pairwise.hip)
    if (tau > 0.0 && !same && v < tau * (gii + gjj)) {  // NaN compares false: non-finite rows are never listed
      listed[i] = 1;                       // benign race: every writer stores 1
      listed[j] = 1;
    }
    if (v < 0.0) v = 0.0;  // rounding of nearly identical rows; NaN stays NaN
    sq[e] = v;
    if (dist != nullptr) dist[e] = rank_distance(v);
  }
  __syncthreads();
  if (threadIdx.x == 0 && sub != nullptr) {
    bool alias_listed = false;
    for (int r = n - 1; r < n_full; ++r) alias_listed |= listed[r] != 0;
    int count = 0;
    for (int r = 0; r < n_full; ++r)
      if (listed[r] || (alias_listed && r >= n - 1 && n_full > n)) sub[1 + count++] = r;
    sub[0] = count;
  }
}

// G = sum over workgroups (fixed order); the workgroup that finishes LAST (arrival counter in the row-list area,
// zeroed by the Gram kernel of the same call) then forms the squared distances and the gate's row list: one launch
// instead of two (the second one was 5-7 us of a 76 us per-rank aggregation at 8 GPUs).  Which workgroup comes
// last does not matter for the result: every entry of G is summed in a fixed order.
//
// The grid is (entries / 64) x slices: a workgroup owns 64 consecutive entries of G (coalesced 512-byte reads) and ONE
// slice of the partial blocks; its wave w adds the blocks lo + w, lo + w + 16, ... of the slice (independent loads in
// flight), wave 0 adds the 16 wave sums in order, and the last workgroup adds the slice sums in slice order before it
// goes on.  Rounds 4-5 ran this with one slice: 6 workgroups (n = 25) pulling 2 MB of partials through 6 CUs took most
// of the launch's 21-25 us; with 8 slices 48 CUs share it.  (With one slice — short vectors — the order of the
// additions is the round-5 one.)
constexpr int kGramRedWaves = 16;  // 16 waves x 16 loads in flight: the sum is a latency chain over L2/HBM
constexpr int kGramSlicesMax = 8;
static_assert(64 * kGramRedWaves == kSqThreads, "the last workgroup of the reduction runs gram_to_sqdist");
// rk.on (bm_pairwise_rank): that last workgroup also RANKS the rows when the gate listed nothing — 16 waves, one row
// each, right where the distances were formed; when rows were listed the gated direct kernel, next on the stream,
// ranks after it has corrected them (pairwise.hip).
__global__ __launch_bounds__(64 * kGramRedWaves) void gram_reduce_sqdist_kernel(
    const double* __restrict__ partial, int nblocks, int n, double* __restrict__ gram, double tau,
    double* __restrict__ sq, int* __restrict__ sub, int n_full, RankArgs rk, double* __restrict__ slice_sums) {
  // dynamic LDS: the wave sums of the reduction, then (last workgroup, rk.on) the ranking's arrays in the same place
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);
  double(*wsum)[64] = reinterpret_cast<double(*)[64]>(smem);
  __shared__ int listed[BM_MAX_ROWS];
  __shared__ int last;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per_block = n * (n + 1) / 2;
  const int slices = (int)gridDim.y, slice = (int)blockIdx.y;
  const int per_slice = (nblocks + slices - 1) / slices;
  const int lo = slice * per_slice, hi = (lo + per_slice < nblocks) ? lo + per_slice : nblocks;
  const int e = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (e < per_block) {
#pragma unroll 16
    for (int blk = lo + wave; blk < hi; blk += kGramRedWaves) s += partial[(int64_t)blk * per_block + e];
  }
  wsum[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && e < per_block) {
    double tot = wsum[0][lane];
#pragma unroll
    for (int w = 1; w < kGramRedWaves; ++w) tot += wsum[w][lane];
    if (slices == 1)
      gram[e] = tot;
    else
      slice_sums[(int64_t)slice * per_block + e] = tot;
  }
  // arrival: release this workgroup's sums, take a ticket; the last ticket acquires everybody's
  if (!arrive_last(sub + kArrivalSlot, (int)(gridDim.x * gridDim.y), &last)) return;
  if (slices > 1) {
    for (int q = threadIdx.x; q < per_block; q += 64 * kGramRedWaves) {
      double tot = slice_sums[q];
      for (int sl = 1; sl < slices; ++sl) tot += slice_sums[(int64_t)sl * per_block + q];
      gram[q] = tot;
    }
    __syncthreads();  // (this workgroup's own stores, read back below)
  }
  gram_to_sqdist(gram, n, tau, sq, sub, listed, n_full, rk.on ? lds : nullptr);  // (the wave sums are done with)
  if (threadIdx.x == 0) {
    sub[kArrivalSlot] = 0;
    sub[kArrivalSlot + 1] = 0;  // arrival counter of the gated direct kernel (pairwise.hip), next on the stream
    last = sub[0];              // (the count this very lane has just stored)
  }
  if (rk.on) {
    __syncthreads();  // the distances in LDS and the count
    if (last == 0) krum_rank_from_distances(lds, n_full, rk.f, rk.m, rk.mode, rk.order, rk.scores, rk.bitonic != 0);
  }
}

constexpr int kB3MaxBlocks = 1024;  // partial blocks the workspace holds (gram3_partial_doubles); the slice sums sit behind the used ones

// Fixed-order sum of the per-workgroup partial Gram matrices (n rows), then the squared distances of the n_full >= n
// rows of the stack (rows n-1 .. n_full-1 alias the last row of G) + accuracy flag.
int gram_finish(const double* partial, int blocks, int n, int n_full, double* gram, double* sq_nxn, int* sub, double tau,
                hipStream_t s, const RankArgs* rank) {
  const int64_t per_block = (int64_t)n * (n + 1) / 2;
  RankArgs rk{};
  if (rank != nullptr) {
    rk = *rank;
    rk.on = 1;
  }
  int lds_bytes = kGramRedWaves * 64 * (int)sizeof(double);
  if (rk.on && rank_lds_bytes(n_full) > lds_bytes) lds_bytes = rank_lds_bytes(n_full);
  // (static: listed[] + last; beyond 48 KB from n = 55 with a ranking)
  if (const int rc = lds_opt_in(reinterpret_cast<const void*>(gram_reduce_sqdist_kernel), (size_t)lds_bytes,
                                (BM_MAX_ROWS + 1) * sizeof(int)))
    return rc;
  const int chunks = (int)((per_block + 63) / 64);
  // ~48-64 workgroups in all, at least 32 partial blocks per slice, and room for the slice sums behind the partials
  int slices = 64 / chunks;
  if (slices > kGramSlicesMax) slices = kGramSlicesMax;
  if (slices > blocks / 32) slices = blocks / 32;
  if (slices < 1 || blocks + slices > kB3MaxBlocks) slices = 1;
  double* slice_sums = const_cast<double*>(partial) + (int64_t)blocks * per_block;
  hipLaunchKernelGGL(gram_reduce_sqdist_kernel, dim3(chunks, slices), dim3(64 * kGramRedWaves), lds_bytes, s,
                     partial, blocks, n, gram, tau, sq_nxn, sub, n_full, rk, slice_sums);
  BM_LAUNCH_CHECK();
  return 0;
}

int gram_arrival_slot() { return kArrivalSlot; }

template <int K, int NPL>
static int launch_gram3_planes(const RowTable& tab, int n, int64_t d, bool aligned, int centre, double* partial,
                               int* arrival, int blocks, hipStream_t s) {
  using S = B3Shape<K, NPL>;
  auto kern = aligned ? (tuning().pair_load_nt != 0 ? gram3_partial_kernel<K, NPL, true, true>
                                                    : gram3_partial_kernel<K, NPL, true, false>)
                      : gram3_partial_kernel<K, NPL, false, true>;
  if (const int rc = lds_opt_in(reinterpret_cast<const void*>(kern), (size_t)S::kLds, 0)) return rc;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * kB3Waves), S::kLds, s, tab, n, d, 1.0f / (float)n, centre,
                     (unsigned)tuning().pair_dither, partial, arrival, tuning().gram_steady);
  BM_LAUNCH_CHECK();
  return 0;
}

template <int K>
static int launch_gram3(const RowTable& tab, int n, int64_t d, bool aligned, int centre, int planes,
                        double* partial, int* arrival, int blocks, hipStream_t s) {
  return planes == 2 ? launch_gram3_planes<K, 2>(tab, n, d, aligned, centre, partial, arrival, blocks, s)
                     : launch_gram3_planes<K, 3>(tab, n, d, aligned, centre, partial, arrival, blocks, s);
}

int64_t gram3_partial_doubles(int n) { return (int64_t)kB3MaxBlocks * ((int64_t)n * (n + 1) / 2); }

// Partial Gram matrices of the centred rows; returns the number of workgroups (= partial blocks)
// through *blocks_out.  `partial` holds gram3_partial_doubles(n) doubles; `sub` is the 512-byte row-list area of the
// accuracy gate, whose arrival counter (for gram_finish, next on the stream) this launch resets.
int gram3_partials(const float* const* rows, int n, int64_t d, int64_t d_total, double* partial, int* sub,
                   int* blocks_out, hipStream_t s) {
  RowTable tab{};
  for (int i = 0; i < n; ++i) tab.p[i] = rows[i];
  const bool aligned = common_vec_width(reinterpret_cast<const void* const*>(rows), n, nullptr) == 4;
  const int K = (n + 3) / 4;
  // Planes: the exact three-way split below 2^20 coordinates; above, two planes (x ~ h + m, 16 significant
  // bits, the remainder rounded with the coordinate dither of split2_dithered: what is dropped has zero mean and
  // is independent across coordinates by construction), whose error on a squared distance is a random walk
  // over the coordinates: relative 2.8 * 2^-16 * (|x| / |x_i - x_j|) / sqrt(d) <= 1e-6 for a pair just above the
  // accuracy gate at d = 2^20, for a third fewer MFMAs and conversion ops.  The length that counts is the
  // TOTAL one (d_total: all shards of a dim-sharded job), so that the choice does not depend on the world
  // size.  BM_PAIR_PLANES forces 2 or 3.
  int planes = tuning().pair_planes;
  if (planes != 2 && planes != 3) planes = (d_total >= ((int64_t)1 << 20)) ? 2 : 3;
  const int64_t chunks = (d + kB3Chunk - 1) / kB3Chunk;
  int blocks = compute_units() * b3_workgroups_per_cu(K, planes);
  if (blocks > kB3MaxBlocks) blocks = kB3MaxBlocks;
  const int64_t need = (chunks + kB3Waves - 1) / kB3Waves;
  if (blocks > need) blocks = (int)(need > 0 ? need : 1);
  const int centre = 2;  // median of three rows (1 = row mean, 0 = none: measured alternatives)
  int rc;
  switch (K) {
#define BM_B3_CASE(KK) \
  case KK: rc = launch_gram3<KK>(tab, n, d, aligned, centre, planes, partial, sub + kArrivalSlot, blocks, s); break;
    BM_B3_CASE(1) BM_B3_CASE(2) BM_B3_CASE(3) BM_B3_CASE(4) BM_B3_CASE(5) BM_B3_CASE(6) BM_B3_CASE(7)
    BM_B3_CASE(8) BM_B3_CASE(9) BM_B3_CASE(10) BM_B3_CASE(11) BM_B3_CASE(12) BM_B3_CASE(13) BM_B3_CASE(14)
    BM_B3_CASE(15) BM_B3_CASE(16)
#undef BM_B3_CASE
    default: rc = BM_EINVAL;
  }
  *blocks_out = blocks;
  return rc;
}

}  // namespace bm
