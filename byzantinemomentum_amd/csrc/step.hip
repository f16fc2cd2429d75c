// step.hip — the first pass of a simulation step fused into one kernel, plus the momentum /
// clipping primitives of the other momentum placements.
//
// Replaces (reference, PyTorch, one launch and one host sync per line):
//   grad.mul_(clip / grad.norm().item())  per sampled gradient                 attack.py:776-779,791-794
//   gmtm.mul_(mu).add_(grad, alpha=1-damp)  per honest worker                   attack.py:800-804
//   tools.compute_avg_dev_max(grad_sampleds), (grad_honests)                    attack.py:846-847
//   grad_avg / grad_att / byz_grad of the "identical" attacks                   attacks/identical.py:63-86,129-141
// bm_momentum_stats reads every sampled gradient and every momentum buffer ONCE and writes every
// buffer once: 4*d*(ks + 2h + 3) bytes instead of the 4*d*(3h + (h+2) + (ks+1)) of separate
// momentum / honest-statistics / sampled-statistics passes (63 against 103 row passes at ks = h = 20).
//
// Layout: lane <-> VEC consecutive coordinates, the ks + h values of a column group live in
// VGPRs (two-pass deviations without re-reading), row base pointers come from the kernarg
// table; reductions leave the kernel as fp64 per-workgroup partials and are finished in a fixed
// order by a one-wave kernel: deterministic, no float atomics, no host synchronisation.
#include "colwise_kernels.h"  // column_rule: the coordinate-wise rules on register-resident values
#include "gram_split.h"       // bf16 planes + MFMA: the distance pass riding along with the first pass

namespace bm {

constexpr int kStepBlock = 256;
constexpr int kStepMaxBlocks = 16384;  // per-workgroup partials: bm_workspace_bytes(BM_WS_STEP) holds kStepMaxBlocks + 1 sets

struct StepTable {
  const float* g[BM_MAX_ROWS];  // sampled gradients (ks)
  float* b[BM_MAX_ROWS];        // momentum buffers (h), updated in place
};

// NaN-propagating max of |.| partials across a workgroup; result valid on thread 0
template <int BLOCK>
__device__ __forceinline__ float block_reduce_absmax(float m, float* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_down(m, off, 64);
    m = (m != m || o != o) ? __builtin_nanf("") : fmaxf(m, o);
  }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
  __syncthreads();
  float r = 0.0f;
  if (threadIdx.x == 0) {
    r = lds[0];
    for (int w = 1; w < BLOCK / 64; ++w) {
      const float o = lds[w];
      r = (r != r || o != o) ? __builtin_nanf("") : fmaxf(r, o);
    }
  }
  __syncthreads();
  return r;
}

// Register-resident form (max(ks, h) <= T <= 20): the ks + h values of a column group stay in VGPRs, so the deviations
// are two-pass (no cancellation) without re-reading anything.
//   EXACT  ks == h == T: no row predicate is left in the code (the C5 shape: 20 sampled gradients, 20 buffers)
//   CLIP   clipping factors present
//   BURST  one workgroup of kStepBurstBlock lanes per CU, column groups interleaved across the CUs, and a workgroup
//          barrier between the loads + arithmetic and the 23 stores of an iteration: the chip alternates between
//          reading and writing instead of trickling writes between reads (scripts/probes/momentum_burst_probe.hip:
//          1 740 -> 1 685 us on the bare stream of this shape; the results wait in registers, they do not fit the LDS)
// Addressing: 32-bit byte offsets (the host cuts d into pieces below 2^32 bytes) on wave-uniform row pointers
// (saddr form, no 64-bit VALU arithmetic).  The 2T row pointers are NOT kept in SGPRs across the loop (80 of them
// at T = 20: round 2's kernel spilled 322 SGPRs to VGPR lanes): each use fetches its pointer from the kernarg
// segment with a scalar load behind an opaque copy of the segment address, so nothing is hoisted.
//   RULE   -1, or BM_OP_MEDIAN / TRMEAN / PHOCAS / MEAMED: the coordinate-wise rule over the T updated buffers and NB copies of the
//          Byzantine vector (attack.py:821 with a coordinate-wise GAR) is applied to the values this pass already
//          holds in registers and written to defense_out: the rule's own pass over the n rows (n + 1 of the 97 row
//          passes of a C5 step with the median) disappears.  Same network, same operations, same bits as bm_colwise.
typedef const float* __attribute__((address_space(4))) const* KargRowPtrs;
constexpr int kStepBurstBlock = 512;

//   NOMOM  no momentum buffers: the honest rows ARE the sampled rows (momentum at the update, attack.py:809-810): nothing
//          is loaded from or stored to the buffer table, both triples of statistics are those of the one stack
template <int T, int VEC, bool EXACT, bool CLIP, bool BURST, int RULE = -1, int NB = 0, bool NOMOM = false>
__global__ __launch_bounds__(BURST ? kStepBurstBlock : kStepBlock) void momentum_stats_kernel(
    StepTable tab, int ks_rt, int h_rt, uint32_t nvec, float mu, float omd, const float* __restrict__ clipf,
    float* __restrict__ s_avg_out, float* __restrict__ h_avg_out, float* __restrict__ byz_out, float scale,
    int attack_kind, double* __restrict__ partial, int rule_f, float rule_inv_keep, float* __restrict__ defense_out) {
  static_assert(RULE < 0 || (EXACT && NB >= 1), "the fused rule needs the row count at compile time");
  static_assert(!NOMOM || (EXACT && !CLIP), "without buffers: same stack, clipped in place beforehand");
  constexpr int BLOCK = BURST ? kStepBurstBlock : kStepBlock;
  __shared__ double red[BLOCK / 64];
  __shared__ float mred[BLOCK / 64];
  const float fks = (float)(EXACT ? T : ks_rt), fh = (float)(EXACT ? T : h_rt);
  float n2s = 0.0f, dvs = 0.0f, mxs = 0.0f, n2h = 0.0f, dvh = 0.0f, mxh = 0.0f;
  bool nan_s = false, nan_h = false;
  // the row tables are the first kernel argument: g[i] at byte 8 i, b[i] at byte 8 (BM_MAX_ROWS + i)
  uint64_t kbase = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  (void)tab;
  const uint32_t span = gridDim.x * BLOCK;
  const uint32_t iters = (nvec + span - 1) / span;
  const uint32_t first = blockIdx.x * BLOCK + threadIdx.x;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t v = it * span + first;
    const bool live = v < nvec;
    const uint32_t off = v * (uint32_t)(VEC * sizeof(float));
    float g[T][VEC], b[T][VEC];
    float sa[VEC], ha[VEC], bz[VEC], df[VEC];
    asm volatile("" : "+s"(kbase));  // (outside the divergent region: the segment address stays wave-uniform)
    // row counts: compile-time when EXACT, else per-iteration opaque copies (the 2T predicates are recomputed where
    // they are used instead of living in SGPR pairs across the loop)
    int ks = T, h = T;
    if constexpr (!EXACT) {
      ks = ks_rt;
      h = h_rt;
      asm volatile("" : "+s"(ks), "+s"(h));
    }
    if (live) {
      KargRowPtrs karg = (KargRowPtrs)kbase;
#pragma unroll
      for (int i = 0; i < T; ++i) {
        if (i < ks) load_stream_off<VEC>(karg[i], off, g[i]);
        if constexpr (!NOMOM) {
          if (i < h) load_stream_off<VEC>(karg[BM_MAX_ROWS + i], off, b[i]);
        }
      }
      // clip, then momentum: gmtm.mul_(mu).add_(grad, alpha=1-damp) = fma(1-damp, grad, round(mu*gmtm))
#pragma unroll
      for (int i = 0; i < T; ++i) {
        if constexpr (CLIP) {
          if (i < ks) {
            const float cf = clipf[i];  // wave-uniform scalar load (attack.py:791-794)
#pragma unroll
            for (int c = 0; c < VEC; ++c) g[i][c] *= cf;
          }
        }
        if (i < h) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) b[i][c] = NOMOM ? g[i][c] : __builtin_fmaf(omd, g[i][c], mu * b[i][c]);
        }
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        // sampled stack: sequential mean, ||avg||^2, max|avg|, sum_i ||s_i - avg||^2 (tools/pytorch.py:105-125)
        float s = g[0][c];
#pragma unroll
        for (int i = 1; i < T; ++i)
          if (i < ks) s += g[i][c];
        s = s / fks;
        sa[c] = s;
        n2s = __builtin_fmaf(s, s, n2s);
        mxs = fmaxf(mxs, __builtin_fabsf(s));
        nan_s |= (s != s);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < T; ++i)
          if (i < ks) {
            const float df = g[i][c] - s;
            q = __builtin_fmaf(df, df, q);
          }
        dvs += q;
        // honest stack = the updated momentum buffers
        float t = b[0][c];
#pragma unroll
        for (int i = 1; i < T; ++i)
          if (i < h) t += b[i][c];
        t = t / fh;
        ha[c] = t;
        n2h = __builtin_fmaf(t, t, n2h);
        mxh = fmaxf(mxh, __builtin_fabsf(t));
        nan_h |= (t != t);
        float qh = 0.0f;
#pragma unroll
        for (int i = 0; i < T; ++i)
          if (i < h) {
            const float df = b[i][c] - t;
            qh = __builtin_fmaf(df, df, qh);
          }
        dvh += qh;
        // empire: grad_att = grad_avg.neg();  little: grad_att = grad_stck.var(dim=0).sqrt_()
        const float dir = ((attack_kind & 15) == BM_ATTACK_LITTLE) ? __builtin_sqrtf(qh / (fh - 1.0f)) : -t;
        const float att = dir * scale;  // grad_att.mul_(factor)
        bz[c] = (attack_kind & BM_ATTACK_DIRECTION) ? att : t + att;  // byz_grad = grad_avg.add_(grad_att)
        if constexpr (RULE >= 0) {  // defense = GAR(honests + [byz] * NB) for this column (attack.py:821)
          float x[T + NB];
#pragma unroll
          for (int i = 0; i < T; ++i) x[i] = b[i][c];
#pragma unroll
          for (int i = 0; i < NB; ++i) x[T + i] = bz[c];
          df[c] = column_rule<T + NB, RULE, BLOCK>(x, rule_f, rule_inv_keep, nullptr);
        }
      }
    }
    if constexpr (BURST) __syncthreads();  // not for the data: it is what turns the stores of a CU into one burst
    asm volatile("" : "+s"(kbase));
    if (live) {
      KargRowPtrs karg = (KargRowPtrs)kbase;
      if constexpr (!NOMOM) {
#pragma unroll
        for (int i = 0; i < T; ++i)
          if (i < h) store_stream_off<VEC>(const_cast<float*>(karg[BM_MAX_ROWS + i]), off, b[i]);
      }
      if (s_avg_out != nullptr) store_stream_off<VEC>(s_avg_out, off, sa);
      if (h_avg_out != nullptr) store_stream_off<VEC>(h_avg_out, off, ha);
      if (byz_out != nullptr) store_stream_off<VEC>(byz_out, off, bz);
      if constexpr (RULE >= 0) store_stream_off<VEC>(defense_out, off, df);
    }
  }
  if (nan_s) mxs = __builtin_nanf("");  // torch's abs().max() propagates NaN; fmaxf does not
  if (nan_h) mxh = __builtin_nanf("");
  const double r0 = block_reduce_sum<BLOCK>((double)n2s, red);
  const double r1 = block_reduce_sum<BLOCK>((double)dvs, red);
  const double r3 = block_reduce_sum<BLOCK>((double)n2h, red);
  const double r4 = block_reduce_sum<BLOCK>((double)dvh, red);
  const float r2 = block_reduce_absmax<BLOCK>(mxs, mred);
  const float r5 = block_reduce_absmax<BLOCK>(mxh, mred);
  if (threadIdx.x == 0) {
    double* p = partial + (int64_t)blockIdx.x * 6;
    p[0] = r0;
    p[1] = r1;
    p[2] = (double)r2;
    p[3] = r3;
    p[4] = r4;
    p[5] = (double)r5;
  }
}

// ---------------------------------------------------------------------------------------------------
// First pass + squared distances (Krum / Bulyan steps).  The distance pass of those rules reads the h updated buffers
// and the Byzantine vector — exactly what the first pass has just formed in registers.  This kernel contracts the
// centred rows on the bf16 matrix cores where they are (gram_bf16.hip's scheme: two bf16 planes with a coordinate
// dither, one accumulator per magnitude class, dithered fp32 running sums), so the 25 row passes of the stand-alone
// distance kernel (of the 115 of a C5 step with Krum) are not made at all.
//   * shape: ks = h = 20, NB = 1..6 Byzantine copies, 16-byte columns, burst form (one workgroup of 512 lanes per CU);
//   * centre of the Gram: the honest average of the column (the first pass has it; it lies inside the honest stack);
//   * rows: 20 buffers + ONE Byzantine row (the NB copies are identical: the host expands the 21 x 21 matrix);
//   * a wave's iteration covers 256 coordinates, contracted in two halves of 128 through a wave-private LDS region
//     [plane][32 rows][128 coordinates] (rows padded to 272 B: the ds_read_b128 fragment reads of 16 rows are
//     conflict-free); rows 21..31 stay zero.
// Everything else (momentum, statistics, averages, Byzantine vector, stores) is momentum_stats_kernel<20, 4, true, CLIP, true>
// operation for operation: same bits.
// ---------------------------------------------------------------------------------------------------
constexpr int kSgRowBytes = 272;             // 128 coordinates x 2 B + 16 B of padding
constexpr int kSgWaves = kStepBurstBlock / 64;
template <int TT>
struct SgShape {                             // TT honest rows (20: n = 25, f = 5; 14: n = 25, f = 11)
  static constexpr int N = TT + 1;           // rows of the compact Gram: the buffers and ONE Byzantine row
  static constexpr int RB = (N + 15) / 16;   // 16-row blocks
  static constexpr int NP = RB * (RB + 1) / 2;
  static constexpr int kRows = N + 1;         // LDS rows per plane: the N rows and one row of zeros that stands for
                                             // rows N .. 16 RB - 1 of the last block (never written, read as zeros)
  static constexpr int kPlaneBytes = kRows * kSgRowBytes;
  static constexpr int kWaveBytes = 2 * kPlaneBytes;
  static constexpr int kAccBytes = NP * 4 * kStepBurstBlock * 4;  // the running sums of the Gram, [NP * 4][512 lanes] floats
  static constexpr int kRedBytes = kSgWaves * 256 * 8;            // the final reduction aliases the planes
  static constexpr int kPlanes = kSgWaves * kWaveBytes > kRedBytes ? kSgWaves * kWaveBytes : kRedBytes;
  static constexpr int kLds = kPlanes + kAccBytes;                // 95 744 + 24 576 B at TT = 20
};

template <int TT, bool CLIP, bool NOMOM = false>
__global__ __launch_bounds__(kStepBurstBlock) void momentum_gram_kernel(
    StepTable tab, uint32_t nvec, float mu, float omd, const float* __restrict__ clipf, float* __restrict__ s_avg_out,
    float* __restrict__ h_avg_out, float* __restrict__ byz_out, float scale, int attack_kind, unsigned dither_seed,
    double* __restrict__ partial, double* __restrict__ gram_partial, int* __restrict__ arrival, int stagger_ticks) {
  using SG = SgShape<TT>;
  constexpr int T = TT, VEC = 4, BLOCK = kStepBurstBlock;
  constexpr int kSgPlaneBytes = SG::kPlaneBytes, kSgWaveBytes = SG::kWaveBytes, kSgN = SG::N, RB = SG::RB, NP = SG::NP;
  extern __shared__ __attribute__((aligned(16))) char sg_smem[];
  __shared__ double red[BLOCK / 64];
  __shared__ float mred[BLOCK / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  char* wbase = sg_smem + wave * kSgWaveBytes;
  // zero this wave's planes once (the rows past the Byzantine one are never written again)
  for (int o = lane * 16; o < kSgWaveBytes; o += 64 * 16) *reinterpret_cast<f32x4*>(wbase + o) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if (blockIdx.x == 0 && tid == 0 && arrival != nullptr) *arrival = 0;  // see gram_reduce_sqdist_kernel
  // Stagger (BM_STEP_STAGGER_US, experiments): every workgroup walks load -> arithmetic -> barrier -> store in lockstep
  // with the 255 others (one per CU, started together), so the HBM system idles while the chip computes.  Every
  // other workgroup of an XCD (workgroup i runs on XCD i % 8) starts late by about half an iteration: one half of
  // the chip then computes while the other half loads or stores.
  if (stagger_ticks > 0 && ((blockIdx.x >> 3) & 1)) {
    const uint64_t t0 = wall_clock64();  // 100 MHz
    while ((uint64_t)wall_clock64() - t0 < (uint64_t)stagger_ticks) __builtin_amdgcn_s_sleep(16);
  }
  const float fks = (float)T, fh = (float)T;
  float n2s = 0.0f, dvs = 0.0f, mxs = 0.0f, n2h = 0.0f, dvh = 0.0f, mxh = 0.0f;
  bool nan_s = false, nan_h = false;
  // the running sums of the Gram live in LDS ([NP * 4][512] floats, one column per lane: conflict-free), not in 4 NP
  // registers that would be carried through the first half of the loop body, where the 40 loads are in flight
  float* acc_lds = reinterpret_cast<float*>(sg_smem + SG::kPlanes) + tid;
#pragma unroll
  for (int p = 0; p < NP * 4; ++p) acc_lds[p * BLOCK] = 0.0f;
  uint64_t kbase = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  (void)tab;
  const uint32_t span = gridDim.x * BLOCK;
  const uint32_t iters = (nvec + span - 1) / span;
  const uint32_t first = blockIdx.x * BLOCK + threadIdx.x;
  // fragment read addresses: lane (i = l & 15, g = l >> 4) reads row 16 R + i, 8 consecutive coordinates 32 s + 8 g
  const int li = lane & 15, lg = lane >> 4;
  const int rd0 = li * kSgRowBytes + lg * 16;
  // last block: rows N .. 16 RB - 1 are the one row of zeros at index N
  const int rd_last = ((16 * (RB - 1) + li < kSgN) ? (16 * (RB - 1) + li) : kSgN) * kSgRowBytes + lg * 16;
  const int wr0 = (lane & 31) * 8;  // this lane's 4 coordinates of its half: 8 B per plane and row
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t v = it * span + first;
    const bool live = v < nvec;
    const uint32_t off = v * (uint32_t)(VEC * sizeof(float));
    float b[T][VEC];
    float sa[VEC], ha[VEC], bz[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) sa[c] = ha[c] = bz[c] = 0.0f;
    asm volatile("" : "+s"(kbase));
    if (live) {
      float g[T][VEC];  // dead once the statistics of the sampled stack are formed
      KargRowPtrs karg = (KargRowPtrs)kbase;
#pragma unroll
      for (int i = 0; i < T; ++i) {
        load_stream_off<VEC>(karg[i], off, g[i]);
        if constexpr (!NOMOM) load_stream_off<VEC>(karg[BM_MAX_ROWS + i], off, b[i]);
      }
#pragma unroll
      for (int i = 0; i < T; ++i) {
        if constexpr (CLIP) {
          const float cf = clipf[i];
#pragma unroll
          for (int c = 0; c < VEC; ++c) g[i][c] *= cf;
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) b[i][c] = NOMOM ? g[i][c] : __builtin_fmaf(omd, g[i][c], mu * b[i][c]);
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        float s = g[0][c];
#pragma unroll
        for (int i = 1; i < T; ++i) s += g[i][c];
        s = s / fks;
        sa[c] = s;
        n2s = __builtin_fmaf(s, s, n2s);
        mxs = fmaxf(mxs, __builtin_fabsf(s));
        nan_s |= (s != s);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
          const float df = g[i][c] - s;
          q = __builtin_fmaf(df, df, q);
        }
        dvs += q;
        float t = b[0][c];
#pragma unroll
        for (int i = 1; i < T; ++i) t += b[i][c];
        t = t / fh;
        ha[c] = t;
        n2h = __builtin_fmaf(t, t, n2h);
        mxh = fmaxf(mxh, __builtin_fabsf(t));
        nan_h |= (t != t);
        float qh = 0.0f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
          const float df = b[i][c] - t;
          qh = __builtin_fmaf(df, df, qh);
        }
        dvh += qh;
        const float dir = ((attack_kind & 15) == BM_ATTACK_LITTLE) ? __builtin_sqrtf(qh / (fh - 1.0f)) : -t;
        const float att = dir * scale;
        bz[c] = (attack_kind & BM_ATTACK_DIRECTION) ? att : t + att;
      }
    }
    // ---- the Gram of the centred rows (20 updated buffers + the Byzantine row), two halves of 128 coordinates ----
    __builtin_amdgcn_sched_barrier(0);  // keep the two phases apart: interleaved, they do not fit the register file
    {
      const unsigned coord = v * 4u;
      const unsigned d01 = dither_seed == ~0u ? 0x80008000u : dither_pair(coord + dither_seed);
      const unsigned d23 = dither_seed == ~0u ? 0x80008000u : dither_pair(coord + 2u + dither_seed);
      // a non-finite centre must not poison the other rows (a NaN row then shows up as NaN distances of that row only)
      f32x4 ctr;
#pragma unroll
      for (int c = 0; c < VEC; ++c) ctr[c] = (__builtin_fabsf(ha[c]) < __builtin_inff()) ? ha[c] : 0.0f;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        // dither of the running sums: one zero-mean number of 1..2 ulps per (wave iteration, half), see gram_bf16.hip
        const unsigned z = dither_pair((v / 64u * 2u + (unsigned)half) * 2u + 0x3C6EF372u);
        const float rc = (float)((int)(z & 0xffffu) + (int)(z >> 16) - 65536) * 0x1.0p-39f;
        if ((lane >> 5) == half) {
#pragma unroll
          for (int r = 0; r <= T; ++r) {
            f32x4 x;
#pragma unroll
            for (int c = 0; c < VEC; ++c) x[c] = live ? ((r < T ? b[r < T ? r : 0][c] : bz[c]) - ctr[c]) : 0.0f;
            u32x2 hp, mp;
            split2_dithered(x, d01, d23, hp, mp);
            *reinterpret_cast<u32x2*>(wbase + r * kSgRowBytes + wr0) = hp;
            *reinterpret_cast<u32x2*>(wbase + kSgPlaneBytes + r * kSgRowBytes + wr0) = mp;
          }
        }
        __builtin_amdgcn_wave_barrier();
        // one block pair at a time (16 accumulator registers in flight instead of 48; the fragments are read again
        // for every pair they belong to: the LDS has the bandwidth, the register file has no room)
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        int p = 0;
#pragma unroll
        for (int I = 0; I < RB; ++I)
#pragma unroll
          for (int J = I; J < RB; ++J) {
            f32x4 s0 = zero, s1 = zero, s2 = zero, s3 = zero;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
              const int oi = (I == RB - 1 ? rd_last : rd0 + I * 16 * kSgRowBytes) + sl * 64;
              const int oj = (J == RB - 1 ? rd_last : rd0 + J * 16 * kSgRowBytes) + sl * 64;
              const u32x4 fhi = *reinterpret_cast<const u32x4*>(wbase + oi);
              const u32x4 fmi = *reinterpret_cast<const u32x4*>(wbase + kSgPlaneBytes + oi);
              const u32x4 fhj = (I == J) ? fhi : *reinterpret_cast<const u32x4*>(wbase + oj);
              const u32x4 fmj = (I == J) ? fmi : *reinterpret_cast<const u32x4*>(wbase + kSgPlaneBytes + oj);
              s0 = mfma_bf16(fhi, fhj, s0);
              s1 = mfma_bf16(fhi, fmj, s1);
              s3 = mfma_bf16(fmi, fmj, s3);
              s2 = mfma_bf16(fmi, fhj, s2);
            }
            const f32x4 t4 = s0 + ((s1 + s2) + s3);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float o = acc_lds[(p * 4 + q) * BLOCK];
              acc_lds[(p * 4 + q) * BLOCK] = o + __builtin_fmaf(o, rc, t4[q]);
            }
            ++p;
          }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // burst form: loads + arithmetic, then the stores of the whole workgroup
    asm volatile("" : "+s"(kbase));
    if (live) {
      KargRowPtrs karg = (KargRowPtrs)kbase;
      if constexpr (!NOMOM) {
#pragma unroll
        for (int i = 0; i < T; ++i) store_stream_off<VEC>(const_cast<float*>(karg[BM_MAX_ROWS + i]), off, b[i]);
      }
      if (s_avg_out != nullptr) store_stream_off<VEC>(s_avg_out, off, sa);
      if (h_avg_out != nullptr) store_stream_off<VEC>(h_avg_out, off, ha);
      store_stream_off<VEC>(byz_out, off, bz);
    }
  }
  if (nan_s) mxs = __builtin_nanf("");
  if (nan_h) mxh = __builtin_nanf("");
  const double r0 = block_reduce_sum<BLOCK>((double)n2s, red);
  const double r1 = block_reduce_sum<BLOCK>((double)dvs, red);
  const double r3 = block_reduce_sum<BLOCK>((double)n2h, red);
  const double r4 = block_reduce_sum<BLOCK>((double)dvh, red);
  const float r2 = block_reduce_absmax<BLOCK>(mxs, mred);
  const float r5 = block_reduce_absmax<BLOCK>(mxh, mred);
  if (threadIdx.x == 0) {
    double* p = partial + (int64_t)blockIdx.x * 6;
    p[0] = r0;
    p[1] = r1;
    p[2] = (double)r2;
    p[3] = r3;
    p[4] = r4;
    p[5] = (double)r5;
  }
  // ---- workgroup reduction of the partial Gram, fixed order; compact upper triangle of the 21 x 21 matrix ----
  // C/D layout of the 16x16 MFMA: lane l, register v -> row 4*(l>>4)+v, column l&15.
  double* gred = reinterpret_cast<double*>(sg_smem);  // [waves][256], aliases the planes
  constexpr int per_block = kSgN * (kSgN + 1) / 2;
  float outer[NP][4];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)
#pragma unroll
    for (int q = 0; q < 4; ++q) outer[pp][q] = acc_lds[(pp * 4 + q) * BLOCK];
  __syncthreads();
  int p = 0;
#pragma unroll
  for (int I = 0; I < RB; ++I)
#pragma unroll
    for (int J = I; J < RB; ++J) {
#pragma unroll
      for (int q = 0; q < 4; ++q) gred[wave * 256 + (4 * lg + q) * 16 + li] = (double)outer[p][q];
      __syncthreads();
      if (tid < 256) {
        const int rr = tid >> 4, cc = tid & 15;
        double sum = gred[tid];
#pragma unroll
        for (int w = 1; w < kSgWaves; ++w) sum += gred[w * 256 + tid];
        const int gi = 16 * I + rr, gj = 16 * J + cc;
        if (gi <= gj && gj < kSgN) gram_partial[(int64_t)blockIdx.x * per_block + b3_tri_index(gi, gj, kSgN)] = sum;
      }
      __syncthreads();
      ++p;
    }
}

// Streaming form of the same pass, used above 20 rows (where the register-resident form would have to drop
// to 8- or 4-byte loads, or spill): the rows of a coordinate group are consumed in batches of four and only
// running sums are kept, so the register footprint does not grow with the number of rows (108 VGPRs at
// any row count, VEC = 4).  At ks = h = 20 both forms take the same time (1.55-1.6 ms at d = 36.5 M,
// profiles/r02_b_*): the kernel is bound by its 40:23 read:write mix, not by occupancy.
// Deviations use the first row p = x_0 as a pivot:
//   sum_i (x_i - a)^2 = sum_i d_i^2 - 2 (a - p) sum_i d_i + k (a - p)^2,   d_i = x_i - p,
// with a the (rounded, sequential) mean.  The pivot's own deviation (a - p)^2 is part of the result, hence
// sum d_i^2 <= (k + 1) * result: the subtraction cancels at most a factor k + 1, never catastrophically.
template <int T, int VEC, bool CLIP>
__global__ __launch_bounds__(kStepBlock, 4) void momentum_stats_stream_kernel(
    StepTable tab, int ks, int h, uint32_t nvec, float mu, float omd, const float* __restrict__ clipf,
    float* __restrict__ s_avg_out, float* __restrict__ h_avg_out, float* __restrict__ byz_out, float scale,
    int attack_kind, double* __restrict__ partial) {
  __shared__ double red[kStepBlock / 64];
  __shared__ float mred[kStepBlock / 64];
  const float fks = (float)ks, fh = (float)h;
  float n2s = 0.0f, dvs = 0.0f, mxs = 0.0f, n2h = 0.0f, dvh = 0.0f, mxh = 0.0f;
  bool nan_s = false, nan_h = false;
  // row pointers and clipping factors are fetched where they are used (scalar loads from the kernarg segment /
  // the factor array, see momentum_stats_kernel): kept in SGPRs across the loop they were 278-1142 spilled SGPRs
  uint64_t kbase = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  (void)tab;
  const uint32_t stride = gridDim.x * kStepBlock;
  constexpr int kBatch = 4;
  for (uint32_t v = blockIdx.x * kStepBlock + threadIdx.x; v < nvec; v += stride) {
    const uint32_t off = v * (uint32_t)(VEC * sizeof(float));
    // per-iteration copies the compiler cannot see through: the 2T row predicates are recomputed where they are used
    // instead of living in SGPR pairs across the loop
    int ksl = ks, hl = h;
    asm volatile("" : "+s"(ksl), "+s"(hl));
    float ps[VEC], ss[VEC], qs[VEC], ts[VEC];  // sampled: pivot, sequential sum, sum d^2, sum d
    float ph[VEC], sh[VEC], qh[VEC], th[VEC];  // honest
#pragma unroll
    for (int base = 0; base < T; base += kBatch) {
      if (base < ksl) {  // wave-uniform
        float g[kBatch][VEC], b[kBatch][VEC];
        asm volatile("" : "+s"(kbase));
        KargRowPtrs karg = (KargRowPtrs)kbase;
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const int i = base + j;
          load_stream_off<VEC>(karg[i], off, g[j]);  // entries >= ks repeat the last row (host-side padding)
          if (base < hl) load_stream_off<VEC>(karg[BM_MAX_ROWS + i], off, b[j]);
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const int i = base + j;
          // branch-free arithmetic (rows that do not exist are masked with wave-uniform selects): a load
          // whose only use sits behind a branch would be sunk into it and lose its place in the batch
          const bool on_s = i < ksl, on_h = i < hl;
          float cfi = 1.0f;
          if constexpr (CLIP) cfi = clipf[i < ksl ? i : ksl - 1];  // wave-uniform scalar load
#pragma unroll
          for (int c = 0; c < VEC; ++c) {
            const float gv = CLIP ? g[j][c] * cfi : g[j][c];
            if (i == 0) {
              ps[c] = gv;
              ss[c] = gv;
              qs[c] = 0.0f;
              ts[c] = 0.0f;
            } else {
              const float dd = on_s ? gv - ps[c] : 0.0f;
              ss[c] += on_s ? gv : 0.0f;
              qs[c] = __builtin_fmaf(dd, dd, qs[c]);
              ts[c] += dd;
            }
            g[j][c] = gv;
          }
          if (base < hl) {  // wave-uniform, the loads of b sit in the same region
#pragma unroll
            for (int c = 0; c < VEC; ++c) b[j][c] = __builtin_fmaf(omd, g[j][c], mu * b[j][c]);
            if (on_h) store_stream_off<VEC>(const_cast<float*>(karg[BM_MAX_ROWS + i]), off, b[j]);
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
              const float bv = b[j][c];
              if (i == 0) {
                ph[c] = bv;
                sh[c] = bv;
                qh[c] = 0.0f;
                th[c] = 0.0f;
              } else {
                const float dd = on_h ? bv - ph[c] : 0.0f;
                sh[c] += on_h ? bv : 0.0f;
                qh[c] = __builtin_fmaf(dd, dd, qh[c]);
                th[c] += dd;
              }
            }
          }
        }
      }
    }
    float r_sa[VEC], r_ha[VEC], r_bz[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      const float s = ss[c] / fks;
      r_sa[c] = s;
      n2s = __builtin_fmaf(s, s, n2s);
      mxs = fmaxf(mxs, __builtin_fabsf(s));
      nan_s |= (s != s);
      const float es = s - ps[c];
      dvs += __builtin_fmaf(es, __builtin_fmaf(fks, es, -2.0f * ts[c]), qs[c]);
      const float t = sh[c] / fh;
      r_ha[c] = t;
      n2h = __builtin_fmaf(t, t, n2h);
      mxh = fmaxf(mxh, __builtin_fabsf(t));
      nan_h |= (t != t);
      const float eh = t - ph[c];
      float colq = __builtin_fmaf(eh, __builtin_fmaf(fh, eh, -2.0f * th[c]), qh[c]);
      colq = colq < 0.0f ? 0.0f : colq;  // rounding of a column whose rows coincide; NaN stays NaN
      dvh += colq;
      const float dir = ((attack_kind & 15) == BM_ATTACK_LITTLE) ? __builtin_sqrtf(colq / (fh - 1.0f)) : -t;
      const float att = dir * scale;
      r_bz[c] = (attack_kind & BM_ATTACK_DIRECTION) ? att : t + att;
    }
    if (s_avg_out != nullptr) store_stream_off<VEC>(s_avg_out, off, r_sa);
    if (h_avg_out != nullptr) store_stream_off<VEC>(h_avg_out, off, r_ha);
    if (byz_out != nullptr) store_stream_off<VEC>(byz_out, off, r_bz);
  }
  if (nan_s) mxs = __builtin_nanf("");
  if (nan_h) mxh = __builtin_nanf("");
  const double r0 = block_reduce_sum<kStepBlock>((double)n2s, red);
  const double r1 = block_reduce_sum<kStepBlock>((double)dvs, red);
  const double r3 = block_reduce_sum<kStepBlock>((double)n2h, red);
  const double r4 = block_reduce_sum<kStepBlock>((double)dvh, red);
  const float r2 = block_reduce_absmax<kStepBlock>(mxs, mred);
  const float r5 = block_reduce_absmax<kStepBlock>(mxh, mred);
  if (threadIdx.x == 0) {
    double* p = partial + (int64_t)blockIdx.x * 6;
    p[0] = r0;
    p[1] = r1 < 0.0 ? 0.0 : r1;
    p[2] = (double)r2;
    p[3] = r3;
    p[4] = r4;
    p[5] = (double)r5;
  }
}

// Fixed-order reduction of [nparts][6] partials: slots 0,1,3,4 are sums, 2 and 5 NaN-propagating maxima.
constexpr int kFinishThreads = 256;
__global__ __launch_bounds__(kFinishThreads) void step_finish_kernel(const double* __restrict__ partial, int nparts,
                                                                     double* __restrict__ out6) {
  __shared__ double red[kFinishThreads / 64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  bool nan2 = false, nan5 = false;
  for (int b = threadIdx.x; b < nparts; b += kFinishThreads) {
    const double* p = partial + (int64_t)b * 6;
    s[0] += p[0];
    s[1] += p[1];
    s[3] += p[3];
    s[4] += p[4];
    nan2 |= (p[2] != p[2]);
    nan5 |= (p[5] != p[5]);
    s[2] = p[2] > s[2] ? p[2] : s[2];
    s[5] = p[5] > s[5] ? p[5] : s[5];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s[0] += __shfl_down(s[0], off, 64);
    s[1] += __shfl_down(s[1], off, 64);
    s[3] += __shfl_down(s[3], off, 64);
    s[4] += __shfl_down(s[4], off, 64);
    const double o2 = __shfl_down(s[2], off, 64), o5 = __shfl_down(s[5], off, 64);
    s[2] = o2 > s[2] ? o2 : s[2];
    s[5] = o5 > s[5] ? o5 : s[5];
    nan2 |= (bool)__shfl_down((int)nan2, off, 64);
    nan5 |= (bool)__shfl_down((int)nan5, off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[wave][k] = s[k];
    red[wave][6] = nan2 ? 1.0 : 0.0;
    red[wave][7] = nan5 ? 1.0 : 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kFinishThreads / 64; ++w) {  // waves in index order: the result does not depend on timing
      s[0] += red[w][0];
      s[1] += red[w][1];
      s[3] += red[w][3];
      s[4] += red[w][4];
      s[2] = red[w][2] > s[2] ? red[w][2] : s[2];
      s[5] = red[w][5] > s[5] ? red[w][5] : s[5];
      nan2 |= red[w][6] != 0.0;
      nan5 |= red[w][7] != 0.0;
    }
  }
  if (threadIdx.x == 0) {  // thread 0 alone holds the totals (lane 0 of the other waves holds partial sums)
    out6[0] = s[0];
    out6[1] = s[1];
    out6[2] = nan2 ? __builtin_nan("") : s[2];
    out6[3] = s[3];
    out6[4] = s[4];
    out6[5] = nan5 ? __builtin_nan("") : s[5];
  }
}

template <int T, int VEC, bool EXACT, bool CLIP>
static int launch_momentum_stats_form(const StepTable& tab, int ks, int h, int64_t nvec, float mu, float omd,
                                      const float* clipf, float* s_avg, float* h_avg, float* byz, float scale,
                                      int kind, double* partial, int* grid_io, hipStream_t s) {
  // burst form: one workgroup per CU, once every CU has BM_STEP_BURST (default 8) iterations to alternate over
  const int cus = compute_units();
  const int64_t burst_iters = nvec / ((int64_t)cus * kStepBurstBlock);
  if (tuning().step_burst > 0 && burst_iters >= tuning().step_burst && cus < *grid_io) {
    *grid_io = cus;
    hipLaunchKernelGGL((momentum_stats_kernel<T, VEC, EXACT, CLIP, true>), dim3(cus), dim3(kStepBurstBlock), 0, s, tab,
                       ks, h, (uint32_t)nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial, 0, 0.0f, nullptr);
  } else {
    hipLaunchKernelGGL((momentum_stats_kernel<T, VEC, EXACT, CLIP, false>), dim3(*grid_io), dim3(kStepBlock), 0, s, tab,
                       ks, h, (uint32_t)nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial, 0, 0.0f, nullptr);
  }
  BM_LAUNCH_CHECK();
  return 0;
}

// *grid_io: in = the grid of the plain form, out = the number of workgroups launched (= partial sets written)
template <int T, int VEC>
static int launch_momentum_stats(const StepTable& tab, int ks, int h, int64_t nvec, float mu, float omd,
                                 const float* clipf, float* s_avg, float* h_avg, float* byz, float scale, int kind,
                                 double* partial, int* grid_io, hipStream_t s) {
#define BM_STEP_FORM(EX, CL) \
  launch_momentum_stats_form<T, VEC, EX, CL>(tab, ks, h, nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial, grid_io, s)
  const bool exact = (ks == T && h == T);
  if constexpr (T > 12) {  // with row predicates the 2 x 20 x VEC values no longer fit: the dispatcher sends those shapes elsewhere
    if (!exact) return BM_EINVAL;
    return clipf != nullptr ? BM_STEP_FORM(true, true) : BM_STEP_FORM(true, false);
  } else {
    if (clipf != nullptr) return exact ? BM_STEP_FORM(true, true) : BM_STEP_FORM(false, true);
    return exact ? BM_STEP_FORM(true, false) : BM_STEP_FORM(false, false);
  }
#undef BM_STEP_FORM
}

template <int T, int VEC>
static int launch_momentum_stats_stream(const StepTable& tab, int ks, int h, int64_t nvec, float mu, float omd,
                                        const float* clipf, float* s_avg, float* h_avg, float* byz, float scale,
                                        int kind, double* partial, int* grid_io, hipStream_t s) {
  if (clipf != nullptr)
    hipLaunchKernelGGL((momentum_stats_stream_kernel<T, VEC, true>), dim3(*grid_io), dim3(kStepBlock), 0, s, tab, ks, h,
                       (uint32_t)nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial);
  else
    hipLaunchKernelGGL((momentum_stats_stream_kernel<T, VEC, false>), dim3(*grid_io), dim3(kStepBlock), 0, s, tab, ks, h,
                       (uint32_t)nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial);
  BM_LAUNCH_CHECK();
  return 0;
}

// Forms by (ks, h): the register-resident kernel holds 2 * T * VEC values per lane; without row predicates (ks == h == T,
// T = 8, 12, 14, 20) it fits two waves per SIMD up to T = 20, with them up to T = 12; every other shape takes the streaming
// form, whose footprint does not depend on the number of rows.
template <int VEC>
static int dispatch_momentum_stats(const StepTable& tab, int ks, int h, int64_t nvec, float mu, float omd,
                                   const float* clipf, float* s_avg, float* h_avg, float* byz, float scale, int kind,
                                   double* partial, int* grid_io, hipStream_t s) {
  const int t = ks > h ? ks : h;
#define BM_STEP_ARGS tab, ks, h, nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial, grid_io, s
  // BM_STEP_STREAM=1: the streaming form at every size (tests, experiments)
  if (tuning().step_stream != 1) {
    if (t <= 8) return launch_momentum_stats<8, VEC>(BM_STEP_ARGS);
    if (t <= 12) return launch_momentum_stats<12, VEC>(BM_STEP_ARGS);
    if (ks == 14 && h == 14) return launch_momentum_stats<14, VEC>(BM_STEP_ARGS);  // n = 25, f = 11 (reproduce.py:181)
    if (ks == 20 && h == 20) return launch_momentum_stats<20, VEC>(BM_STEP_ARGS);  // n = 25, f = 5
  }
  if (t <= 20) return launch_momentum_stats_stream<20, VEC>(BM_STEP_ARGS);
  if (t <= 40) return launch_momentum_stats_stream<40, VEC>(BM_STEP_ARGS);
  return launch_momentum_stats_stream<64, VEC>(BM_STEP_ARGS);
#undef BM_STEP_ARGS
}

// ---------------------------------------------------------------------------
// out_i = fma(b, q_i, a * (c_i * p_i)) for k vectors: every momentum placement of attack.py
//   worker : out = p = buffer, q = gradient, a = mu, b = 1-damp                     attack.py:800-804
//   server : out = new,  p = gradient, a = 1-damp, q = server momentum (shared), b = mu   attack.py:805-808
//   update : out = p = server momentum, q = defense gradient, a = mu, b = 1-damp (k = 1)   attack.py:838-839
//   nesterov look-ahead: out = p = parameters, q = momentum, a = 1, b = -mu*lr             attack.py:762,767
// c_i: optional per-row device scalars (the clipping factors), applied to p_i.
// ---------------------------------------------------------------------------
struct Fma3Table {
  float* out[BM_MAX_ROWS];
  const float* p[BM_MAX_ROWS];
  const float* q[BM_MAX_ROWS];
};

template <int VEC>
__global__ __launch_bounds__(kStepBlock) void multi_fma3_kernel(Fma3Table tab, int64_t nvec, float a, float b_host,
                                                                const float* __restrict__ pscale,
                                                                const double* __restrict__ b_dev) {
  // b from device memory (bm_multi_fma3_bdev: the factor bm_attack_line_search_device left there), rounded to fp32 as
  // the host's double -> float conversion of the same number would be
  const float b = b_dev != nullptr ? (float)b_dev[0] : b_host;
  float* out = tab.out[blockIdx.y];
  const float* p = tab.p[blockIdx.y];
  const float* q = tab.q[blockIdx.y];
  const float c = pscale != nullptr ? pscale[blockIdx.y] : 1.0f;
  const int64_t stride = (int64_t)gridDim.x * kStepBlock;
  for (int64_t v = (int64_t)blockIdx.x * kStepBlock + threadIdx.x; v < nvec; v += stride) {
    float pp[VEC], qq[VEC];
    load_stream<VEC>(p + v * VEC, pp);
    load_stream<VEC>(q + v * VEC, qq);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float x = pscale != nullptr ? pp[e] * c : pp[e];
      pp[e] = __builtin_fmaf(b, qq[e], a * x);
    }
    store_stream<VEC>(out + v * VEC, pp);
  }
}

// y_i *= c_i in place, rows with c_i == 1 untouched (no traffic): the in-place gradient clipping of
// attack.py:776-779,791-794 once the factors are known.
struct ScaleTable {
  float* y[BM_MAX_ROWS];
};
template <int VEC>
__global__ __launch_bounds__(kStepBlock) void multi_scale_kernel(ScaleTable tab, int64_t nvec,
                                                                 const float* __restrict__ factors) {
  const float c = factors[blockIdx.y];
  if (c == 1.0f) return;
  float* y = tab.y[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * kStepBlock;
  for (int64_t v = (int64_t)blockIdx.x * kStepBlock + threadIdx.x; v < nvec; v += stride) {
    float yy[VEC];
    load_stream<VEC>(y + v * VEC, yy);
#pragma unroll
    for (int e = 0; e < VEC; ++e) yy[e] *= c;
    store_stream<VEC>(y + v * VEC, yy);
  }
}

// factors_i = clip / ||g_i|| if ||g_i|| > clip else 1 (attack.py:791-794), from squared norms on the device
__global__ __launch_bounds__(64) void clip_factors_kernel(const double* __restrict__ row_sq, int k, float clip,
                                                          float* __restrict__ factors) {
  const int i = threadIdx.x;
  if (i >= k) return;
  const double norm = sqrt(row_sq[i]);
  factors[i] = (norm > (double)clip) ? (float)((double)clip / norm) : 1.0f;
}

// ---------------------------------------------------------------------------
// First pass + coordinate-wise rule in one kernel (median / trimmed mean over the h = 20 updated buffers and 1..6
// copies of the Byzantine vector: the C5 shape and its neighbours).  Returns false when no instance fits.
// ---------------------------------------------------------------------------
template <int T, int RULE, int NB, bool CLIP, bool NOMOM = false>
static void launch_fused_rule(const StepTable& tab, int64_t nvec, float mu, float omd, const float* clipf, float* s_avg,
                              float* h_avg, float* byz, float scale, int kind, double* partial, int rule_f,
                              float* defense, int* grid_io, hipStream_t s) {
  constexpr int N = T + NB;
  const int keep = (RULE == BM_OP_TRMEAN) ? (N - 2 * rule_f) : ((RULE == BM_OP_PHOCAS || RULE == BM_OP_MEAMED) ? (N - rule_f) : N);
  const float inv_keep = 1.0f / (float)(keep > 0 ? keep : 1);
  const int cus = compute_units();
  const int64_t burst_iters = nvec / ((int64_t)cus * kStepBurstBlock);
  if (tuning().step_burst > 0 && burst_iters >= tuning().step_burst && cus < *grid_io) {
    *grid_io = cus;
    hipLaunchKernelGGL((momentum_stats_kernel<T, 4, true, CLIP, true, RULE, NB, NOMOM>), dim3(cus), dim3(kStepBurstBlock), 0,
                       s, tab, T, T, (uint32_t)nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial,
                       rule_f, inv_keep, defense);
  } else {
    hipLaunchKernelGGL((momentum_stats_kernel<T, 4, true, CLIP, false, RULE, NB, NOMOM>), dim3(*grid_io), dim3(kStepBlock), 0,
                       s, tab, T, T, (uint32_t)nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial,
                       rule_f, inv_keep, defense);
  }
}

template <int T, int RULE, int NB>
static void launch_fused_rule_clip(const StepTable& tab, int64_t nvec, float mu, float omd, const float* clipf,
                                   float* s_avg, float* h_avg, float* byz, float scale, int kind, double* partial,
                                   int rule_f, float* defense, int* grid_io, hipStream_t s, bool nomom) {
  if (nomom)
    launch_fused_rule<T, RULE, NB, false, true>(tab, nvec, mu, omd, nullptr, s_avg, h_avg, byz, scale, kind, partial, rule_f, defense, grid_io, s);
  else if (clipf != nullptr)
    launch_fused_rule<T, RULE, NB, true>(tab, nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial, rule_f, defense, grid_io, s);
  else
    launch_fused_rule<T, RULE, NB, false>(tab, nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial, rule_f, defense, grid_io, s);
}

// instances: the two shapes of the reference's n = 25 runs (reproduce.py:165-209: f = 5 -> 20 honest rows + 5 Byzantine
// copies, f = 11 -> 14 + 11); every other shape runs the first pass and the rule as two kernels (same results).  Round 3
// also instantiated 20 + 1..4 and 20 + 6 copies, which no configuration of the reference reaches: 120 of the 168
// instances of this kernel, two of the three minutes of the build.
static bool fused_rule_shape(int h, int nb) { return (h == 20 && nb == 5) || (h == 14 && nb == 11); }
static bool fused_rule_instance(int ks, int h, int nb, int op) {
  if ((op != BM_OP_MEDIAN && op != BM_OP_TRMEAN && op != BM_OP_PHOCAS && op != BM_OP_MEAMED) || tuning().step_stream == 1 ||
      ks != h)
    return false;
  return fused_rule_shape(h, nb);
}

static int launch_fused_rule_any(int op, int h, int nb, const StepTable& tab, int64_t nvec, float mu, float omd,
                                 const float* clipf, float* s_avg, float* h_avg, float* byz, float scale, int kind,
                                 double* partial, int rule_f, float* defense, int* grid_io, hipStream_t s,
                                 bool nomom = false) {
#define BM_FUSED_ARGS tab, nvec, mu, omd, clipf, s_avg, h_avg, byz, scale, kind, partial, rule_f, defense, grid_io, s, nomom
#define BM_FUSED_CASE(TV, NBV)                                                  \
  if (h == TV && nb == NBV) {                                                   \
    if (op == BM_OP_MEDIAN)                                                     \
      launch_fused_rule_clip<TV, BM_OP_MEDIAN, NBV>(BM_FUSED_ARGS);             \
    else if (op == BM_OP_TRMEAN)                                                \
      launch_fused_rule_clip<TV, BM_OP_TRMEAN, NBV>(BM_FUSED_ARGS);             \
    else if (op == BM_OP_PHOCAS)                                                \
      launch_fused_rule_clip<TV, BM_OP_PHOCAS, NBV>(BM_FUSED_ARGS);             \
    else                                                                        \
      launch_fused_rule_clip<TV, BM_OP_MEAMED, NBV>(BM_FUSED_ARGS);             \
    BM_LAUNCH_CHECK();                                                          \
    return 0;                                                                   \
  }
  BM_FUSED_CASE(20, 5) BM_FUSED_CASE(14, 11)
#undef BM_FUSED_CASE
#undef BM_FUSED_ARGS
  return BM_EINVAL;
}

// Gram contribution of the d mod 4 trailing columns (at most 3) of the fused distance pass: one more partial block,
// in fp64, centred on the honest average like the body.
__global__ __launch_bounds__(256) void tail_gram_kernel(RowTable rows /* the buffers + byz, offset to the tail */, int nrows,
                                                        const float* __restrict__ h_avg_tail, int cols,
                                                        double* __restrict__ block) {
  const int t = threadIdx.x;
  if (t >= nrows * (nrows + 1) / 2) return;
  int i = 0, rem = t, len = nrows;  // t -> (i, j), i <= j, row-major over the upper triangle
  while (rem >= len) {
    rem -= len;
    --len;
    ++i;
  }
  const int j = i + rem;
  double acc = 0.0;
  for (int c = 0; c < cols; ++c) {
    const float ctr = (__builtin_fabsf(h_avg_tail[c]) < __builtin_inff()) ? h_avg_tail[c] : 0.0f;
    acc += (double)(rows.p[i][c] - ctr) * (double)(rows.p[j][c] - ctr);
  }
  block[b3_tri_index(i, j, nrows)] = acc;
}

static inline int vec_of(uintptr_t bits) { return (bits & 15u) == 0 ? 4 : ((bits & 7u) == 0 ? 2 : 1); }

}  // namespace bm

namespace bm {
// rule_op < 0: the first pass alone (bm_momentum_stats); else also defense = rule(buffers + [byz] * nb) (bm_momentum_stats_colwise)
static int momentum_stats_impl(const float* const* sampled, int ks, float* const* buffers, int h, int64_t d, float mu,
                               float one_minus_damp, const float* clip_factors, float* sampled_avg, float* honest_avg,
                               float* byz_out, float scale, int attack_kind, double* out6, void* ws, void* stream,
                               int rule_op, int rule_f, int nb, float* defense_out) {
  if (sampled == nullptr || buffers == nullptr || out6 == nullptr || ws == nullptr || h < 1 || ks < h ||
      ks > BM_MAX_ROWS || d < 0 || ((attack_kind & ~BM_ATTACK_DIRECTION) != BM_ATTACK_EMPIRE && (attack_kind & ~BM_ATTACK_DIRECTION) != BM_ATTACK_LITTLE))
    return BM_EINVAL;
  if (rule_op >= 0 && ((rule_op != BM_OP_MEDIAN && rule_op != BM_OP_TRMEAN && rule_op != BM_OP_PHOCAS && rule_op != BM_OP_MEAMED) ||
                       nb < 1 || h + nb > BM_MAX_ROWS || (attack_kind & BM_ATTACK_DIRECTION) != 0 ||
                       (d > 0 && (byz_out == nullptr || defense_out == nullptr)) ||
                       (rule_op != BM_OP_MEDIAN && (rule_f < 0 || h + nb < 2 * rule_f + 1))))
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  StepTable tab{};
  uintptr_t bits = reinterpret_cast<uintptr_t>(sampled_avg) | reinterpret_cast<uintptr_t>(honest_avg) |
                   reinterpret_cast<uintptr_t>(byz_out) | reinterpret_cast<uintptr_t>(rule_op >= 0 ? defense_out : nullptr);
  for (int i = 0; i < ks; ++i) {
    tab.g[i] = sampled[i];
    bits |= reinterpret_cast<uintptr_t>(sampled[i]);
  }
  for (int i = 0; i < h; ++i) {
    tab.b[i] = buffers[i];
    bits |= reinterpret_cast<uintptr_t>(buffers[i]);
  }
  // pad the tables with their last row: the streaming kernel loads whole batches unconditionally
  for (int i = ks; i < BM_MAX_ROWS; ++i) tab.g[i] = sampled[ks - 1];
  for (int i = h; i < BM_MAX_ROWS; ++i) tab.b[i] = buffers[h - 1];
  double* partial = static_cast<double*>(ws);
  const int vec = vec_of(bits);
  // pieces of at most 2^29 coordinates: byte offsets fit 32 bits inside the register-resident kernel
  const int64_t pieces = d > 0 ? (d + kMaxColsPerLaunch - 1) / kMaxColsPerLaunch : 0;
  int cap = pieces > 1 ? (int)((kStepMaxBlocks - 1) / pieces) - 1 : 2047;
  if (cap > 2047) cap = 2047;
  if (cap < 1) return BM_EINVAL;  // d >= 2^42: not a gradient
  int nparts = 0;
  int rc = 0;
  for (int64_t lo = 0; lo < d; lo += kMaxColsPerLaunch) {
    const int64_t dp = (d - lo < kMaxColsPerLaunch) ? (d - lo) : kMaxColsPerLaunch;
    StepTable piece = tab;
    for (int i = 0; i < BM_MAX_ROWS; ++i) {
      piece.g[i] += lo;
      piece.b[i] += lo;
    }
    float* sa = sampled_avg ? sampled_avg + lo : nullptr;
    float* ha = honest_avg ? honest_avg + lo : nullptr;
    float* bz = byz_out ? byz_out + lo : nullptr;
    int64_t body = 0;
    // the rule inside the first pass where an instance exists (16-byte columns, ks = h = 20, 1..6 Byzantine copies)
    const bool fused = rule_op >= 0 && vec == 4 && dp / 4 > 0 && fused_rule_instance(ks, h, nb, rule_op);
    if (fused) {
      const int64_t nvec = dp / 4;
      int grid = stream_grid(nvec, kStepBlock, cap);
      rc = launch_fused_rule_any(rule_op, h, nb, piece, nvec, mu, one_minus_damp, clip_factors, sa, ha, bz, scale,
                                 attack_kind, partial + (int64_t)nparts * 6, rule_f, defense_out + lo, &grid, s);
      if (rc != 0) return rc;
      nparts += grid;
      body = nvec * 4;
    } else if (vec >= 2 && dp / vec > 0) {
      const int64_t nvec = dp / vec;
      int grid = stream_grid(nvec, kStepBlock, cap);
      rc = (vec == 4) ? dispatch_momentum_stats<4>(piece, ks, h, nvec, mu, one_minus_damp, clip_factors, sa, ha, bz,
                                                    scale, attack_kind, partial + (int64_t)nparts * 6, &grid, s)
                      : dispatch_momentum_stats<2>(piece, ks, h, nvec, mu, one_minus_damp, clip_factors, sa, ha, bz,
                                                    scale, attack_kind, partial + (int64_t)nparts * 6, &grid, s);
      if (rc != 0) return rc;
      nparts += grid;
      body = nvec * vec;
    }
    if (body < dp) {
      StepTable tail = piece;
      for (int i = 0; i < BM_MAX_ROWS; ++i) {
        tail.g[i] += body;
        tail.b[i] += body;
      }
      const int64_t rest = dp - body;
      int grid = (body == 0) ? stream_grid(rest, kStepBlock, cap) : 1;
      rc = dispatch_momentum_stats<1>(tail, ks, h, rest, mu, one_minus_damp, clip_factors, sa ? sa + body : nullptr,
                                      ha ? ha + body : nullptr, bz ? bz + body : nullptr, scale, attack_kind,
                                      partial + (int64_t)nparts * 6, &grid, s);
      if (rc != 0) return rc;
      nparts += grid;
    }
    if (rule_op >= 0) {  // the columns of this piece the fused kernel did not cover: the rule as its own launch
      const int64_t from = fused ? body : 0;
      if (from < dp) {
        const float* rows[BM_MAX_ROWS];
        for (int i = 0; i < h; ++i) rows[i] = buffers[i] + lo + from;
        for (int i = 0; i < nb; ++i) rows[h + i] = byz_out + lo + from;
        rc = bm_colwise(rule_op, rows, h + nb, dp - from, rule_f, defense_out + lo + from, stream);
        if (rc != 0) return rc;
      }
    }
  }
  // d == 0: nparts == 0 and the finish kernel writes zeros — every rank of a sharded job reaches its collective
  hipLaunchKernelGGL(step_finish_kernel, dim3(1), dim3(kFinishThreads), 0, s, partial, nparts, out6);
  BM_LAUNCH_CHECK();
  return 0;
}
}  // namespace bm

extern "C" int bm_momentum_stats(const float* const* sampled, int ks, float* const* buffers, int h, int64_t d,
                                 float mu, float one_minus_damp, const float* clip_factors, float* sampled_avg,
                                 float* honest_avg, float* byz_out, float scale, int attack_kind, double* out6,
                                 void* ws, void* stream) {
  return bm::momentum_stats_impl(sampled, ks, buffers, h, d, mu, one_minus_damp, clip_factors, sampled_avg, honest_avg,
                                 byz_out, scale, attack_kind, out6, ws, stream, -1, 0, 0, nullptr);
}

extern "C" int bm_momentum_stats_colwise(const float* const* sampled, int ks, float* const* buffers, int h, int64_t d,
                                         float mu, float one_minus_damp, const float* clip_factors, float* sampled_avg,
                                         float* honest_avg, float* byz_out, float scale, int attack_kind, int rule_op,
                                         int rule_f, int n_byz, float* defense_out, double* out6, void* ws,
                                         void* stream) {
  if (rule_op < 0) return BM_EINVAL;
  return bm::momentum_stats_impl(sampled, ks, buffers, h, d, mu, one_minus_damp, clip_factors, sampled_avg, honest_avg,
                                 byz_out, scale, attack_kind, out6, ws, stream, rule_op, rule_f, n_byz, defense_out);
}

namespace bm {
double* pairwise_gram_area(void* ws);       // pairwise.hip
int* pairwise_arrival_counter(void* ws);
int pairwise_from_gram_partials(const float* const* rows, int n_full, int nc, int blocks, int64_t d, double* sq_nxn,
                                void* ws, hipStream_t s);
}

extern "C" int bm_momentum_stats_sqdist(const float* const* sampled, int ks, float* const* buffers, int h, int64_t d,
                                        int64_t d_total, float mu, float one_minus_damp, const float* clip_factors,
                                        float* sampled_avg, float* honest_avg, float* byz_out, float scale,
                                        int attack_kind, int n_byz, double* sq_nxn, double* out6, void* ws,
                                        void* ws_pair, void* stream) {
  using namespace bm;
  const int n = h + n_byz;
  if (sampled == nullptr || buffers == nullptr || out6 == nullptr || ws == nullptr || ws_pair == nullptr ||
      sq_nxn == nullptr || h < 1 || ks < h || ks > BM_MAX_ROWS || n_byz < 1 || n > BM_MAX_ROWS || d < 0 || d_total < d ||
      (attack_kind != BM_ATTACK_EMPIRE && attack_kind != BM_ATTACK_LITTLE) || (d > 0 && byz_out == nullptr))
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* rows[BM_MAX_ROWS];
  for (int i = 0; i < h; ++i) rows[i] = buffers[i];
  for (int i = h; i < n; ++i) rows[i] = byz_out;
  uintptr_t bits = reinterpret_cast<uintptr_t>(sampled_avg) | reinterpret_cast<uintptr_t>(honest_avg) |
                   reinterpret_cast<uintptr_t>(byz_out);
  for (int i = 0; i < ks; ++i) bits |= reinterpret_cast<uintptr_t>(sampled[i]);
  for (int i = 0; i < h; ++i) bits |= reinterpret_cast<uintptr_t>(buffers[i]);
  const int cus = compute_units();
  const int64_t nvec = d / 4;
  const bool shape_ok = ks == h && ((h == 20 && n_byz <= 6) || (h == 14 && n_byz == 11));  // n = 25 with f = 5 / 11, and neighbours
  const bool fused = shape_ok && vec_of(bits) == 4 && honest_avg != nullptr &&
                     d <= kMaxColsPerLaunch && tuning().step_stream != 1 && tuning().pair_mode == 0 &&
                     tuning().pair_planes != 3 && tuning().step_burst > 0 &&
                     nvec / ((int64_t)cus * kStepBurstBlock) >= tuning().step_burst;
  if (!fused) {  // the two passes one after the other: same results as the fused kernel up to the distances' rounding
    int rc = bm_momentum_stats(sampled, ks, buffers, h, d, mu, one_minus_damp, clip_factors, sampled_avg, honest_avg,
                               byz_out, scale, attack_kind, out6, ws, stream);
    if (rc != 0) return rc;
    return bm_pairwise_sqdist_shard(rows, n, d, d_total, sq_nxn, ws_pair, stream);
  }
  StepTable tab{};
  for (int i = 0; i < ks; ++i) tab.g[i] = sampled[i];
  for (int i = 0; i < h; ++i) tab.b[i] = buffers[i];
  for (int i = ks; i < BM_MAX_ROWS; ++i) tab.g[i] = sampled[ks - 1];
  for (int i = h; i < BM_MAX_ROWS; ++i) tab.b[i] = buffers[h - 1];
  double* partial = static_cast<double*>(ws);
  double* gram_partial = pairwise_gram_area(ws_pair);
  const int nc = h + 1;  // rows of the compact Gram
  const int per_block = nc * (nc + 1) / 2;
  void (*kern)(StepTable, uint32_t, float, float, const float*, float*, float*, float*, float, int, unsigned, double*,
               double*, int*, int);
  int lds;
  if (h == 20) {
    kern = clip_factors != nullptr ? momentum_gram_kernel<20, true> : momentum_gram_kernel<20, false>;
    lds = SgShape<20>::kLds;
  } else {
    kern = clip_factors != nullptr ? momentum_gram_kernel<14, true> : momentum_gram_kernel<14, false>;
    lds = SgShape<14>::kLds;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return hip_code(e);
  hipLaunchKernelGGL(kern, dim3(cus), dim3(kStepBurstBlock), lds, s, tab, (uint32_t)nvec, mu, one_minus_damp,
                     clip_factors, sampled_avg, honest_avg, byz_out, scale, attack_kind, (unsigned)tuning().pair_dither,
                     partial, gram_partial, pairwise_arrival_counter(ws_pair), tuning().step_stagger_us * 100);
  BM_LAUNCH_CHECK();
  int nparts = cus, blocks = cus;
  const int64_t body = nvec * 4;
  if (body < d) {  // at most 3 trailing columns: the scalar form of the first pass, and their Gram as one more block
    StepTable tail = tab;
    for (int i = 0; i < BM_MAX_ROWS; ++i) {
      tail.g[i] += body;
      tail.b[i] += body;
    }
    int grid = 1;
    int rc = dispatch_momentum_stats<1>(tail, ks, h, d - body, mu, one_minus_damp, clip_factors,
                                        sampled_avg ? sampled_avg + body : nullptr, honest_avg + body, byz_out + body,
                                        scale, attack_kind, partial + (int64_t)nparts * 6, &grid, s);
    if (rc != 0) return rc;
    nparts += grid;
    RowTable trows{};
    for (int i = 0; i < h; ++i) trows.p[i] = buffers[i] + body;
    trows.p[h] = byz_out + body;
    hipLaunchKernelGGL(tail_gram_kernel, dim3(1), dim3(256), 0, s, trows, nc, honest_avg + body, (int)(d - body),
                       gram_partial + (int64_t)blocks * per_block);
    BM_LAUNCH_CHECK();
    blocks += 1;
  }
  hipLaunchKernelGGL(step_finish_kernel, dim3(1), dim3(kFinishThreads), 0, s, partial, nparts, out6);
  BM_LAUNCH_CHECK();
  return pairwise_from_gram_partials(rows, n, nc, blocks, d, sq_nxn, ws_pair, s);
}

// ---------------------------------------------------------------------------
// The same two fusions without momentum buffers: the honest rows are the sampled rows themselves (momentum at the
// update — the reference's default placement — attack.py:809-810,837-839).  out6[0..2] = out6[3..5] = the statistics
// of the one stack.
// ---------------------------------------------------------------------------
namespace bm {
static int stack_stats_plain(const float* const* rows, int k, int64_t d, float* avg_out, float* byz_out, float scale,
                             int attack_kind, double* out6, void* ws, void* stream) {
  int rc = bm_stack_stats(rows, k, d, avg_out, byz_out, scale, attack_kind, out6 + 3, ws, stream);
  if (rc != 0) return rc;
  return hip_code(hipMemcpyAsync(out6, out6 + 3, 3 * sizeof(double), hipMemcpyDeviceToDevice,
                                 static_cast<hipStream_t>(stream)));
}
static bool nomom_shape(int k, int nb) { return (k == 20 && nb >= 1 && nb <= 6) || (k == 14 && nb == 11); }
}  // namespace bm

extern "C" int bm_stack_stats_colwise(const float* const* rows, int k, int64_t d, float* avg_out, float* byz_out,
                                      float scale, int attack_kind, int rule_op, int rule_f, int n_byz,
                                      float* defense_out, double* out6, void* ws, void* stream) {
  using namespace bm;
  const int n = k + n_byz;
  if (rows == nullptr || out6 == nullptr || ws == nullptr || k < 1 || n_byz < 1 || n > BM_MAX_ROWS || d < 0 ||
      (attack_kind != BM_ATTACK_EMPIRE && attack_kind != BM_ATTACK_LITTLE) ||
      (rule_op != BM_OP_MEDIAN && rule_op != BM_OP_TRMEAN && rule_op != BM_OP_PHOCAS && rule_op != BM_OP_MEAMED) ||
      (rule_op != BM_OP_MEDIAN && (rule_f < 0 || n < 2 * rule_f + 1)) ||
      (d > 0 && (avg_out == nullptr || byz_out == nullptr || defense_out == nullptr)))
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  uintptr_t bits = reinterpret_cast<uintptr_t>(avg_out) | reinterpret_cast<uintptr_t>(byz_out) |
                   reinterpret_cast<uintptr_t>(defense_out);
  for (int i = 0; i < k; ++i) bits |= reinterpret_cast<uintptr_t>(rows[i]);
  const bool fused = fused_rule_shape(k, n_byz) && vec_of(bits) == 4 && d % 4 == 0 && d > 0 && d <= kMaxColsPerLaunch &&
                     tuning().step_stream != 1;
  if (fused) {
    StepTable tab{};
    for (int i = 0; i < BM_MAX_ROWS; ++i) {
      tab.g[i] = rows[i < k ? i : k - 1];
      tab.b[i] = nullptr;  // never dereferenced without momentum
    }
    double* partial = static_cast<double*>(ws);
    const int64_t nvec = d / 4;
    int grid = stream_grid(nvec, kStepBlock, 2047);
    int rc = launch_fused_rule_any(rule_op, k, n_byz, tab, nvec, 0.0f, 1.0f, nullptr, nullptr, avg_out, byz_out, scale,
                                   attack_kind, partial, rule_f, defense_out, &grid, s, true);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(step_finish_kernel, dim3(1), dim3(kFinishThreads), 0, s, partial, grid, out6);
    BM_LAUNCH_CHECK();
    return 0;
  }
  int rc = stack_stats_plain(rows, k, d, avg_out, byz_out, scale, attack_kind, out6, ws, stream);
  if (rc != 0 || d == 0) return rc;
  const float* all[BM_MAX_ROWS];
  for (int i = 0; i < k; ++i) all[i] = rows[i];
  for (int i = k; i < n; ++i) all[i] = byz_out;
  return bm_colwise(rule_op, all, n, d, rule_f, defense_out, stream);
}

extern "C" int bm_stack_stats_sqdist(const float* const* rows, int k, int64_t d, int64_t d_total, float* avg_out,
                                     float* byz_out, float scale, int attack_kind, int n_byz, double* sq_nxn,
                                     double* out6, void* ws, void* ws_pair, void* stream) {
  using namespace bm;
  const int n = k + n_byz;
  if (rows == nullptr || out6 == nullptr || ws == nullptr || ws_pair == nullptr || sq_nxn == nullptr || k < 1 ||
      n_byz < 1 || n > BM_MAX_ROWS || d < 0 || d_total < d ||
      (attack_kind != BM_ATTACK_EMPIRE && attack_kind != BM_ATTACK_LITTLE) ||
      (d > 0 && (avg_out == nullptr || byz_out == nullptr)))
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* all[BM_MAX_ROWS];
  for (int i = 0; i < k; ++i) all[i] = rows[i];
  for (int i = k; i < n; ++i) all[i] = byz_out;
  uintptr_t bits = reinterpret_cast<uintptr_t>(avg_out) | reinterpret_cast<uintptr_t>(byz_out);
  for (int i = 0; i < k; ++i) bits |= reinterpret_cast<uintptr_t>(rows[i]);
  const int cus = compute_units();
  const int64_t nvec = d / 4;
  const bool fused = nomom_shape(k, n_byz) && vec_of(bits) == 4 && d % 4 == 0 && d <= kMaxColsPerLaunch &&
                     tuning().step_stream != 1 && tuning().pair_mode == 0 && tuning().pair_planes != 3 &&
                     tuning().step_burst > 0 && nvec / ((int64_t)cus * kStepBurstBlock) >= tuning().step_burst;
  if (!fused) {
    int rc = stack_stats_plain(rows, k, d, avg_out, byz_out, scale, attack_kind, out6, ws, stream);
    if (rc != 0) return rc;
    return bm_pairwise_sqdist_shard(all, n, d, d_total, sq_nxn, ws_pair, stream);
  }
  StepTable tab{};
  for (int i = 0; i < BM_MAX_ROWS; ++i) {
    tab.g[i] = rows[i < k ? i : k - 1];
    tab.b[i] = nullptr;
  }
  double* partial = static_cast<double*>(ws);
  double* gram_partial = pairwise_gram_area(ws_pair);
  void (*kern)(StepTable, uint32_t, float, float, const float*, float*, float*, float*, float, int, unsigned, double*,
               double*, int*, int);
  int lds;
  if (k == 20) {
    kern = momentum_gram_kernel<20, false, true>;
    lds = SgShape<20>::kLds;
  } else {
    kern = momentum_gram_kernel<14, false, true>;
    lds = SgShape<14>::kLds;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return hip_code(e);
  hipLaunchKernelGGL(kern, dim3(cus), dim3(kStepBurstBlock), lds, s, tab, (uint32_t)nvec, 0.0f, 1.0f, nullptr, nullptr,
                     avg_out, byz_out, scale, attack_kind, (unsigned)tuning().pair_dither, partial, gram_partial,
                     pairwise_arrival_counter(ws_pair), tuning().step_stagger_us * 100);
  BM_LAUNCH_CHECK();
  hipLaunchKernelGGL(step_finish_kernel, dim3(1), dim3(kFinishThreads), 0, s, partial, cus, out6);
  BM_LAUNCH_CHECK();
  return pairwise_from_gram_partials(all, n, k + 1, cus, d, sq_nxn, ws_pair, s);
}

static int multi_fma3_launch(float* const* out, const float* const* p, const float* const* q, int k, int64_t d, float a,
                             float b, const float* p_scale, const double* b_dev, void* stream) {
  using namespace bm;
  if (out == nullptr || p == nullptr || q == nullptr || k < 1 || k > BM_MAX_ROWS || d < 0) return BM_EINVAL;
  if (d == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Fma3Table tab{};
  uintptr_t bits = 0;
  for (int i = 0; i < k; ++i) {
    tab.out[i] = out[i];
    tab.p[i] = p[i];
    tab.q[i] = q[i];
    bits |= reinterpret_cast<uintptr_t>(out[i]) | reinterpret_cast<uintptr_t>(p[i]) | reinterpret_cast<uintptr_t>(q[i]);
  }
  const int vec = vec_of(bits);
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    const int grid = stream_grid(nvec, kStepBlock, 2048);
    if (vec == 4)
      hipLaunchKernelGGL(multi_fma3_kernel<4>, dim3(grid, k), dim3(kStepBlock), 0, s, tab, nvec, a, b, p_scale, b_dev);
    else
      hipLaunchKernelGGL(multi_fma3_kernel<2>, dim3(grid, k), dim3(kStepBlock), 0, s, tab, nvec, a, b, p_scale, b_dev);
    BM_LAUNCH_CHECK();
    body = nvec * vec;
  }
  if (body < d) {
    Fma3Table tail = tab;
    for (int i = 0; i < k; ++i) {
      tail.out[i] += body;
      tail.p[i] += body;
      tail.q[i] += body;
    }
    const int64_t rest = d - body;
    hipLaunchKernelGGL(multi_fma3_kernel<1>, dim3(stream_grid(rest, kStepBlock, 2048), k), dim3(kStepBlock), 0, s,
                       tail, rest, a, b, p_scale, b_dev);
    BM_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int bm_multi_fma3(float* const* out, const float* const* p, const float* const* q, int k, int64_t d,
                             float a, float b, const float* p_scale, void* stream) {
  return multi_fma3_launch(out, p, q, k, d, a, b, p_scale, nullptr, stream);
}

extern "C" int bm_multi_fma3_bdev(float* const* out, const float* const* p, const float* const* q, int k, int64_t d,
                                  float a, const double* b_dev, const float* p_scale, void* stream) {
  if (b_dev == nullptr) return BM_EINVAL;
  return multi_fma3_launch(out, p, q, k, d, a, 0.0f, p_scale, b_dev, stream);
}

extern "C" int bm_multi_scale(float* const* y, int k, int64_t d, const float* factors, void* stream) {
  using namespace bm;
  if (y == nullptr || factors == nullptr || k < 1 || k > BM_MAX_ROWS || d < 0) return BM_EINVAL;
  if (d == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ScaleTable tab{};
  uintptr_t bits = 0;
  for (int i = 0; i < k; ++i) {
    tab.y[i] = y[i];
    bits |= reinterpret_cast<uintptr_t>(y[i]);
  }
  const int vec = vec_of(bits);
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    const int grid = stream_grid(nvec, kStepBlock, 2048);
    if (vec == 4)
      hipLaunchKernelGGL(multi_scale_kernel<4>, dim3(grid, k), dim3(kStepBlock), 0, s, tab, nvec, factors);
    else
      hipLaunchKernelGGL(multi_scale_kernel<2>, dim3(grid, k), dim3(kStepBlock), 0, s, tab, nvec, factors);
    BM_LAUNCH_CHECK();
    body = nvec * vec;
  }
  if (body < d) {
    ScaleTable tail = tab;
    for (int i = 0; i < k; ++i) tail.y[i] += body;
    const int64_t rest = d - body;
    hipLaunchKernelGGL(multi_scale_kernel<1>, dim3(stream_grid(rest, kStepBlock, 2048), k), dim3(kStepBlock), 0, s,
                       tail, rest, factors);
    BM_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int bm_clip_factors(const double* row_sq, int k, float clip, float* factors_out, void* stream) {
  using namespace bm;
  if (row_sq == nullptr || factors_out == nullptr || k < 1 || k > BM_MAX_ROWS || !(clip > 0.0f)) return BM_EINVAL;
  hipLaunchKernelGGL(clip_factors_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), row_sq, k, clip,
                     factors_out);
  BM_LAUNCH_CHECK();
  return 0;
}
