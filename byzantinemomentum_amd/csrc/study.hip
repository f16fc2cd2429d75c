// study.hip — the study block of a simulation step in ONE pass (attack.py:848-868).
//
// Replaces, per step (reference: one PyTorch launch and one host synchronisation per line):
//   tools.compute_avg_dev_max(grad_attacks)                        attack.py:848   tools/pytorch.py:97-125
//   grad_defense.norm().item(), grad_defense.abs().max().item()    attack.py:851-852
//   the six cosines among sampled avg / honest avg / defense / attack avg          attack.py:854-859
//   cosine with the previous sampled average, curvature sum over the past ones     attack.py:861-866
//   (params - origin).norm().item()                                                attack.py:830
// and the update of this implementation's curvature combination C = sum_i mu^i past_i (step.py).
// With the momentum at the update (the reference's DEFAULT placement, attack.py:832-839) the kernel also carries
//   grad_momentum_server.mul_(mu).add_(grad_defense, alpha=1 - dampening)          attack.py:836-838
// on the defense vector it reads anyway: M read and written here (2 row units) instead of a pass of its own that
// re-read the defense vector (3 units and a launch: 74 us of a 0.95 ms step at d = 36.5 M, round 5).
//
// Round 2 ran these as five to seven d-sized passes (statistics of the attack stack, statistics of the defense
// vector, the Gram of four vectors plus two dots, two passes over C, the two-row distance kernel): 16 + 2 row passes.
// Every one of them is an elementwise function or a dot product of the same few vectors, so ONE kernel reads
// s, h, defense, byz, the newest past average, C, the leaving past average (and params, origin) once and writes C
// once: 8 (+ 2) row passes.  The attack average is never materialised: the attack stack is f copies of byz, so its
// sequential mean a = (byz + ... + byz) / f and its deviations are rebuilt per element from byz with exactly the
// operations of tools/pytorch.py:105-125 (stack_stats_kernel on f aliased rows).
//
// Reductions: per-lane fp32 partials over at most 64 elements (the plain form folds them into fp64 every 16
// iterations, the burst form at every burst), fp64 per lane, workgroup and grid, finished in a fixed order by a
// one-workgroup kernel: deterministic, no atomics, no host synchronisation.
#include "bm_common.h"

namespace bm {

constexpr int kStudyBlock = 256;
constexpr int kStudyMaxBlocks = 2048;  // bm_workspace_bytes(BM_WS_STUDY) holds kStudyMaxBlocks + 1 partial sets
constexpr int kStudySums = 14;         // 10 Gram entries (upper triangle of 4 x 4), <s, past>, <s, C>, attack deviations, l2
constexpr int kStudyPartial = 16;      // + 2 maxima (|attack avg|, |defense|), NaN encoded as NaN

struct StudyArgs {
  const float* s;       // sampled average
  const float* h;       // honest average
  const float* def;     // aggregated gradient
  const float* byz;     // the Byzantine vector (ATT)
  const float* past;    // newest past sampled average (CM >= 2)
  const float* oldest;  // the past average that leaves the deque (CM == 3)
  const float* params;  // (L2)
  const float* origin;  // (L2)
  float* curv;          // C, read (CM >= 2) and written (CM >= 1)
  float* a_out;         // attack average, optional
  float* mom;           // momentum of the update (attack.py:836-838, --momentum-at update), or NULL: M <- fma(mom_b, defense, mom_a * M)
  float mom_a, mom_b;   // mu, 1 - dampening
};

// CM: 0 no curvature term kept (nb_past = 0); 1 first step: C <- s, no dot with the past;
//     2 C <- fma(1, s, mu * C); 3 C <- fma(1, s, mu * fma(w, oldest, C)) with w = -(mu^(P-1))
// (the arithmetic of the two bm_multi_fma3 passes round 2 used: identical bits).
template <bool ATT, int CM, bool L2, int VEC>
__global__ __launch_bounds__(kStudyBlock) void study_stats_kernel(StudyArgs a, int f_real, float mu, float w_oldest,
                                                                  int64_t nvec, double* __restrict__ partial) {
  __shared__ double red[kStudyBlock / 64];
  __shared__ float mred[kStudyBlock / 64];
  float acc[kStudySums];
  double acc64[kStudySums];  // the fp32 chains are folded into fp64 every 16 iterations (<= 64 elements each)
#pragma unroll
  for (int i = 0; i < kStudySums; ++i) {
    acc[i] = 0.0f;
    acc64[i] = 0.0;
  }
  int since = 0;
  float amax = 0.0f, dmax = 0.0f;
  bool a_nan = false, d_nan = false;
  const float ff = (float)f_real;
  const int64_t stride = (int64_t)gridDim.x * kStudyBlock;
  for (int64_t v = (int64_t)blockIdx.x * kStudyBlock + threadIdx.x; v < nvec; v += stride) {
    const int64_t j = v * VEC;
    float s[VEC], h[VEC], df[VEC], bz[VEC], pa[VEC], cv[VEC], ol[VEC], pp[VEC], oo[VEC], mm[VEC];
    // every load of the iteration is issued before the first use
    load_stream<VEC>(a.s + j, s);
    load_stream<VEC>(a.h + j, h);
    load_stream<VEC>(a.def + j, df);
    if (a.mom != nullptr) load_stream<VEC>(a.mom + j, mm);  // (wave-uniform)
    if constexpr (ATT) load_stream<VEC>(a.byz + j, bz);
    if constexpr (CM >= 2) {
      load_stream<VEC>(a.past + j, pa);
      load_stream<VEC>(a.curv + j, cv);
    }
    if constexpr (CM == 3) load_stream<VEC>(a.oldest + j, ol);
    if constexpr (L2) {
      load_stream<VEC>(a.params + j, pp);
      load_stream<VEC>(a.origin + j, oo);
    }
    float av[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      av[c] = 0.0f;
      if constexpr (ATT) {
        // grad_avg = samples[0].clone(); add_(...) f - 1 times; div_(f)   (tools/pytorch.py:108-111)
        float t = bz[c];
        for (int i = 1; i < f_real; ++i) t += bz[c];
        t = t / ff;
        av[c] = t;
        amax = fmaxf(amax, __builtin_fabsf(t));
        a_nan |= (t != t);
        // sum_i |a_i - avg|^2: f identical terms, accumulated like the reference's loop (tools/pytorch.py:117-121)
        const float dd = bz[c] - t;
        float q = 0.0f;
        for (int i = 0; i < f_real; ++i) q = __builtin_fmaf(dd, dd, q);
        acc[12] += q;
      }
      dmax = fmaxf(dmax, __builtin_fabsf(df[c]));
      d_nan |= (df[c] != df[c]);
      // Gram of (s, h, defense, attack avg), upper triangle in row order
      acc[0] = __builtin_fmaf(s[c], s[c], acc[0]);
      acc[1] = __builtin_fmaf(s[c], h[c], acc[1]);
      acc[2] = __builtin_fmaf(s[c], df[c], acc[2]);
      acc[4] = __builtin_fmaf(h[c], h[c], acc[4]);
      acc[5] = __builtin_fmaf(h[c], df[c], acc[5]);
      acc[7] = __builtin_fmaf(df[c], df[c], acc[7]);
      if constexpr (ATT) {
        acc[3] = __builtin_fmaf(s[c], av[c], acc[3]);
        acc[6] = __builtin_fmaf(h[c], av[c], acc[6]);
        acc[8] = __builtin_fmaf(df[c], av[c], acc[8]);
        acc[9] = __builtin_fmaf(av[c], av[c], acc[9]);
      }
      if constexpr (CM >= 2) {
        acc[10] = __builtin_fmaf(s[c], pa[c], acc[10]);
        acc[11] = __builtin_fmaf(s[c], cv[c], acc[11]);
      }
      if constexpr (L2) {
        const float e = pp[c] - oo[c];
        acc[13] = __builtin_fmaf(e, e, acc[13]);
      }
      // curvature combination for the NEXT step (after the dot with the old one)
      if constexpr (CM == 1) cv[c] = s[c];
      if constexpr (CM == 3) cv[c] = __builtin_fmaf(w_oldest, ol[c], 1.0f * cv[c]);
      if constexpr (CM >= 2) cv[c] = __builtin_fmaf(1.0f, s[c], mu * cv[c]);
    }
    if constexpr (CM >= 1) store_stream<VEC>(a.curv + j, cv);
    if (a.mom != nullptr) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) mm[c] = __builtin_fmaf(a.mom_b, df[c], a.mom_a * mm[c]);  // bm_multi_fma3's bits
      store_stream<VEC>(a.mom + j, mm);
    }
    if constexpr (ATT) {
      if (a.a_out != nullptr) store_stream<VEC>(a.a_out + j, av);
    }
    if (++since == 16) {
#pragma unroll
      for (int i = 0; i < kStudySums; ++i) {
        acc64[i] += (double)acc[i];
        acc[i] = 0.0f;
      }
      since = 0;
    }
  }
  if (a_nan) amax = __builtin_nanf("");  // torch's abs().max() propagates NaN; fmaxf does not
  if (d_nan) dmax = __builtin_nanf("");
  // partial layout: [slot][kStudyMaxBlocks + 1 workgroups] (the finish kernel then reads contiguous doubles)
  constexpr int64_t kSlotStride = kStudyMaxBlocks + 1;
  double* p = partial + blockIdx.x;
#pragma unroll
  for (int i = 0; i < kStudySums; ++i) {
    const double r = block_reduce_sum<kStudyBlock>(acc64[i] + (double)acc[i], red);
    if (threadIdx.x == 0) p[i * kSlotStride] = r;
  }
  // NaN-propagating maxima
  float m2[2] = {amax, dmax};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float m = m2[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_down(m, off, 64);
      m = (m != m || o != o) ? __builtin_nanf("") : fmaxf(m, o);
    }
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      float mm = mred[0];
      for (int wv = 1; wv < kStudyBlock / 64; ++wv) {
        const float o = mred[wv];
        mm = (mm != mm || o != o) ? __builtin_nanf("") : fmaxf(mm, o);
      }
      p[(kStudySums + k) * kSlotStride] = (double)mm;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// Burst form (CM >= 1, 16-byte columns, long vectors, no attack-average output): the same loads and the same
// arithmetic per element, but the ONE written stream — the curvature combination C — leaves the CU in bursts
// that coincide across the chip, like the results of colwise_burst_kernel (colwise_kernels.h): a write stream that
// trickles out between the reads of 256 CUs costs about twice its bytes on this HBM system, and here it is one
// stream in eight.  One workgroup of 1024 lanes per CU walks the column groups interleaved with the other CUs,
// two groups per lane and iteration (14 loads of 16 bytes in flight per lane instead of 7), stages kStudySlots
// iterations of C in LDS, meets at a barrier and writes them back to back.
// The per-lane fp32 partial sums are folded into fp64 at every burst (32 elements per sum), so the length of an fp32
// chain no longer grows with d (the plain form: one chain per lane over the whole grid-stride loop).
// ---------------------------------------------------------------------------------------------------
constexpr int kStudyBurstThreads = 1024;
constexpr int kStudySlotBudget = 8;  // staged iterations per burst over all written streams: 8 x 1024 lanes x 16 B = 128 KB of LDS

// MOM: the momentum of the update rides along (StudyArgs::mom), a second written stream: both are staged, four
// iterations each per burst instead of eight.
template <bool ATT, int CM, bool L2, bool MOM = false>
__global__ __launch_bounds__(kStudyBurstThreads) void study_stats_burst_kernel(StudyArgs a, int f_real, float mu,
                                                                               float w_oldest, uint32_t nvec,
                                                                               double* __restrict__ partial) {
  static_assert(CM >= 1, "without a written stream there is nothing to burst");
  constexpr int VEC = 4;
  constexpr int kStudySlots = MOM ? kStudySlotBudget / 2 : kStudySlotBudget;
  // column groups per lane and iteration (with params / origin, or with the momentum, two do not fit 128 VGPRs)
  constexpr int U = ((L2 && CM >= 2) || MOM) ? 1 : 2;
  using V = typename VecLoad<VEC>::T;
  __shared__ V stage[kStudySlots * kStudyBurstThreads];
  __shared__ V stage_mom[MOM ? kStudySlots * kStudyBurstThreads : 1];
  __shared__ double red[kStudyBurstThreads / 64];
  __shared__ float mred[kStudyBurstThreads / 64];
  float acc[kStudySums];
  double acc64[kStudySums];
#pragma unroll
  for (int i = 0; i < kStudySums; ++i) {
    acc[i] = 0.0f;
    acc64[i] = 0.0;
  }
  float amax = 0.0f, dmax = 0.0f;
  bool a_nan = false, d_nan = false;
  const float ff = (float)f_real;
  const uint32_t tid = threadIdx.x;
  const uint32_t span = gridDim.x * kStudyBurstThreads;  // column groups per iteration of the whole grid
  const uint32_t iters = (nvec + span - 1) / span;
  const uint32_t first = blockIdx.x * kStudyBurstThreads + tid;
  for (uint32_t p0 = 0; p0 < iters; p0 += kStudySlots) {
    const uint32_t p1 = (p0 + kStudySlots < iters) ? p0 + kStudySlots : iters;
    for (uint32_t it = p0; it < p1; it += U) {
      float s[U][VEC], h[U][VEC], df[U][VEC], bz[U][VEC], pa[U][VEC], cv[U][VEC], ol[U][VEC], pp[U][VEC], oo[U][VEC];
      float mm[U][VEC];
      bool live[U];
      // every load of the iteration is issued before the first use
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t v = (it + u) * span + first;
        live[u] = (it + u) < p1 && v < nvec;
        if (live[u]) {
          const int64_t j = (int64_t)v * VEC;
          load_stream<VEC>(a.s + j, s[u]);
          load_stream<VEC>(a.h + j, h[u]);
          load_stream<VEC>(a.def + j, df[u]);
          if constexpr (MOM) load_stream<VEC>(a.mom + j, mm[u]);
          if constexpr (ATT) load_stream<VEC>(a.byz + j, bz[u]);
          if constexpr (CM >= 2) {
            load_stream<VEC>(a.past + j, pa[u]);
            load_stream<VEC>(a.curv + j, cv[u]);
          }
          if constexpr (CM == 3) load_stream<VEC>(a.oldest + j, ol[u]);
          if constexpr (L2) {
            load_stream<VEC>(a.params + j, pp[u]);
            load_stream<VEC>(a.origin + j, oo[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          float av = 0.0f;
          if constexpr (ATT) {
            // grad_avg = samples[0].clone(); add_(...) f - 1 times; div_(f)   (tools/pytorch.py:108-111)
            float t = bz[u][c];
            for (int i = 1; i < f_real; ++i) t += bz[u][c];
            t = t / ff;
            av = t;
            amax = fmaxf(amax, __builtin_fabsf(t));
            a_nan |= (t != t);
            const float dd = bz[u][c] - t;
            float q = 0.0f;
            for (int i = 0; i < f_real; ++i) q = __builtin_fmaf(dd, dd, q);
            acc[12] += q;
          }
          dmax = fmaxf(dmax, __builtin_fabsf(df[u][c]));
          d_nan |= (df[u][c] != df[u][c]);
          acc[0] = __builtin_fmaf(s[u][c], s[u][c], acc[0]);
          acc[1] = __builtin_fmaf(s[u][c], h[u][c], acc[1]);
          acc[2] = __builtin_fmaf(s[u][c], df[u][c], acc[2]);
          acc[4] = __builtin_fmaf(h[u][c], h[u][c], acc[4]);
          acc[5] = __builtin_fmaf(h[u][c], df[u][c], acc[5]);
          acc[7] = __builtin_fmaf(df[u][c], df[u][c], acc[7]);
          if constexpr (ATT) {
            acc[3] = __builtin_fmaf(s[u][c], av, acc[3]);
            acc[6] = __builtin_fmaf(h[u][c], av, acc[6]);
            acc[8] = __builtin_fmaf(df[u][c], av, acc[8]);
            acc[9] = __builtin_fmaf(av, av, acc[9]);
          }
          if constexpr (CM >= 2) {
            acc[10] = __builtin_fmaf(s[u][c], pa[u][c], acc[10]);
            acc[11] = __builtin_fmaf(s[u][c], cv[u][c], acc[11]);
          }
          if constexpr (L2) {
            const float e = pp[u][c] - oo[u][c];
            acc[13] = __builtin_fmaf(e, e, acc[13]);
          }
          // curvature combination for the NEXT step: the arithmetic of study_stats_kernel, same bits per element
          if constexpr (CM == 1) cv[u][c] = s[u][c];
          if constexpr (CM == 3) cv[u][c] = __builtin_fmaf(w_oldest, ol[u][c], 1.0f * cv[u][c]);
          if constexpr (CM >= 2) cv[u][c] = __builtin_fmaf(1.0f, s[u][c], mu * cv[u][c]);
        }
        V packed;
#pragma unroll
        for (int c = 0; c < VEC; ++c) packed[c] = cv[u][c];
        stage[(it + u - p0) * kStudyBurstThreads + tid] = packed;
        if constexpr (MOM) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) packed[c] = __builtin_fmaf(a.mom_b, df[u][c], a.mom_a * mm[u][c]);  // bm_multi_fma3's bits
          stage_mom[(it + u - p0) * kStudyBurstThreads + tid] = packed;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kStudySums; ++i) {  // fp32 chains end here: 32 elements each
      acc64[i] += (double)acc[i];
      acc[i] = 0.0f;
    }
    __syncthreads();  // not for the data (a lane reads back its own slots): it is what makes the stores a burst
    for (uint32_t it = p0; it < p1; ++it) {
      const uint32_t v = it * span + first;
      if (v < nvec) {
        __builtin_nontemporal_store(stage[(it - p0) * kStudyBurstThreads + tid], reinterpret_cast<V*>(a.curv + (int64_t)v * VEC));
        if constexpr (MOM)
          __builtin_nontemporal_store(stage_mom[(it - p0) * kStudyBurstThreads + tid], reinterpret_cast<V*>(a.mom + (int64_t)v * VEC));
      }
    }
  }
  if (a_nan) amax = __builtin_nanf("");
  if (d_nan) dmax = __builtin_nanf("");
  constexpr int64_t kSlotStride = kStudyMaxBlocks + 1;
  double* p = partial + blockIdx.x;
#pragma unroll
  for (int i = 0; i < kStudySums; ++i) {
    const double r = block_reduce_sum<kStudyBurstThreads>(acc64[i], red);
    if (threadIdx.x == 0) p[i * kSlotStride] = r;
  }
  float m2[2] = {amax, dmax};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float m = m2[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_down(m, off, 64);
      m = (m != m || o != o) ? __builtin_nanf("") : fmaxf(m, o);
    }
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      float mm = mred[0];
      for (int wv = 1; wv < kStudyBurstThreads / 64; ++wv) {
        const float o = mred[wv];
        mm = (mm != mm || o != o) ? __builtin_nanf("") : fmaxf(mm, o);
      }
      p[(kStudySums + k) * kSlotStride] = (double)mm;
    }
    __syncthreads();
  }
}

// out (BM_STUDY_SLOTS doubles, layout in include/bm_gar.h) from the per-workgroup partials, fixed order:
// one wave per slot, lane l adds the partials of workgroups l, l + 64, ..., then a fixed shuffle tree.
__global__ __launch_bounds__(64) void study_finish_kernel(const double* __restrict__ partial, int nparts,
                                                          double* __restrict__ out) {
  const int slot = blockIdx.x, lane = threadIdx.x;  // slot < kStudyPartial
  // slots 0..22 are written below on every call (zeros where a term does not apply); the spare ones here — the
  // hipMemsetAsync this replaces was a launch of its own (5 us of a step)
  if (slot == 0 && lane >= 23 && lane < BM_STUDY_SLOTS) out[lane] = 0.0;
  double tot = 0.0;
  bool nan = false;
  const bool is_max = slot >= kStudySums;
  for (int b = lane; b < nparts; b += 64) {
    const double v = partial[(int64_t)slot * (kStudyMaxBlocks + 1) + b];
    if (is_max) {
      nan |= (v != v);
      tot = v > tot ? v : tot;
    } else {
      tot += v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(tot, off, 64);
    if (is_max) {
      tot = o > tot ? o : tot;
      nan |= (bool)__shfl_down((int)nan, off, 64);
    } else {
      tot += o;
    }
  }
  if (lane != 0) return;
  if (is_max) {
    out[20 + (slot - kStudySums)] = nan ? __builtin_nan("") : tot;  // [20] max|attack avg|, [21] max|defense|
    return;
  }
  if (slot < 10) {  // slot -> (r, c) of the upper triangle of the 4 x 4 Gram
    int k = 0;
    for (int r = 0; r < 4; ++r)
      for (int c = r; c < 4; ++c) {
        if (k == slot) {
          out[r * 4 + c] = tot;
          out[c * 4 + r] = tot;
        }
        ++k;
      }
    if (slot == 9) out[18] = tot;  // sum avg_a^2
  } else if (slot == 10) {
    out[16] = tot;
  } else if (slot == 11) {
    out[17] = tot;
  } else if (slot == 12) {
    out[19] = tot;
  } else {
    out[22] = tot;
  }
}

template <bool ATT, int CM, bool L2>
static int launch_study_vec(const StudyArgs& a, int vec, int f_real, float mu, float w, int64_t n, int grid,
                            double* partial, hipStream_t s) {
  if (vec == 4)
    hipLaunchKernelGGL((study_stats_kernel<ATT, CM, L2, 4>), dim3(grid), dim3(kStudyBlock), 0, s, a, f_real, mu, w, n, partial);
  else if (vec == 2)
    hipLaunchKernelGGL((study_stats_kernel<ATT, CM, L2, 2>), dim3(grid), dim3(kStudyBlock), 0, s, a, f_real, mu, w, n, partial);
  else
    hipLaunchKernelGGL((study_stats_kernel<ATT, CM, L2, 1>), dim3(grid), dim3(kStudyBlock), 0, s, a, f_real, mu, w, n, partial);
  BM_LAUNCH_CHECK();
  return 0;
}

template <bool ATT, bool L2>
static int launch_study_cm(const StudyArgs& a, int cm, int vec, int f_real, float mu, float w, int64_t n, int grid,
                           double* partial, hipStream_t s) {
  switch (cm) {
    case 0: return launch_study_vec<ATT, 0, L2>(a, vec, f_real, mu, w, n, grid, partial, s);
    case 1: return launch_study_vec<ATT, 1, L2>(a, vec, f_real, mu, w, n, grid, partial, s);
    case 2: return launch_study_vec<ATT, 2, L2>(a, vec, f_real, mu, w, n, grid, partial, s);
    default: return launch_study_vec<ATT, 3, L2>(a, vec, f_real, mu, w, n, grid, partial, s);
  }
}

template <bool ATT, bool L2, bool MOM>
static int launch_study_burst_mom(const StudyArgs& a, int cm, int f_real, float mu, float w, int64_t n, int grid,
                                  double* partial, hipStream_t s) {
  const uint32_t nv = (uint32_t)n;
  if (cm == 1)
    hipLaunchKernelGGL((study_stats_burst_kernel<ATT, 1, L2, MOM>), dim3(grid), dim3(kStudyBurstThreads), 0, s, a, f_real, mu, w, nv, partial);
  else if (cm == 2)
    hipLaunchKernelGGL((study_stats_burst_kernel<ATT, 2, L2, MOM>), dim3(grid), dim3(kStudyBurstThreads), 0, s, a, f_real, mu, w, nv, partial);
  else
    hipLaunchKernelGGL((study_stats_burst_kernel<ATT, 3, L2, MOM>), dim3(grid), dim3(kStudyBurstThreads), 0, s, a, f_real, mu, w, nv, partial);
  BM_LAUNCH_CHECK();
  return 0;
}
template <bool ATT, bool L2>
static int launch_study_burst(const StudyArgs& a, int cm, int f_real, float mu, float w, int64_t n, int grid,
                              double* partial, hipStream_t s) {
  return a.mom != nullptr ? launch_study_burst_mom<ATT, L2, true>(a, cm, f_real, mu, w, n, grid, partial, s)
                          : launch_study_burst_mom<ATT, L2, false>(a, cm, f_real, mu, w, n, grid, partial, s);
}

// The burst form pays from a few iterations per CU on (BM_STUDY_BURST, default 8; 0 = never, 1 = always: tests).
static bool study_burst_eligible(const StudyArgs& a, int cm, int vec, int64_t nvec) {
  const int threshold = tuning().study_burst;
  if (threshold <= 0 || cm < 1 || vec != 4 || a.a_out != nullptr || nvec >= ((int64_t)1 << 30)) return false;
  return nvec >= (int64_t)threshold * compute_units() * kStudyBurstThreads;
}

static int launch_study(const StudyArgs& a, bool att, bool l2, int cm, int vec, int f_real, float mu, float w, int64_t n,
                        int grid, double* partial, hipStream_t s) {
  if (att) return l2 ? launch_study_cm<true, true>(a, cm, vec, f_real, mu, w, n, grid, partial, s)
                     : launch_study_cm<true, false>(a, cm, vec, f_real, mu, w, n, grid, partial, s);
  return l2 ? launch_study_cm<false, true>(a, cm, vec, f_real, mu, w, n, grid, partial, s)
            : launch_study_cm<false, false>(a, cm, vec, f_real, mu, w, n, grid, partial, s);
}

int64_t study_workspace_bytes() { return (int64_t)(kStudyMaxBlocks + 1) * kStudyPartial * (int64_t)sizeof(double); }

}  // namespace bm

extern "C" int bm_study_stats(const float* sampled_avg, const float* honest_avg, const float* defense, const float* byz,
                              int f_real, float* attack_avg_out, const float* past_newest, float* curv,
                              const float* past_oldest, int curv_mode, float mu, float oldest_weight,
                              const float* params, const float* origin, int64_t d, double* out, void* ws, void* stream) {
  return bm_study_stats_update(sampled_avg, honest_avg, defense, byz, f_real, attack_avg_out, past_newest, curv, past_oldest,
                               curv_mode, mu, oldest_weight, params, origin, nullptr, 0.0f, 0.0f, d, out, ws, stream);
}

extern "C" int bm_study_stats_update(const float* sampled_avg, const float* honest_avg, const float* defense,
                                     const float* byz, int f_real, float* attack_avg_out, const float* past_newest,
                                     float* curv, const float* past_oldest, int curv_mode, float mu, float oldest_weight,
                                     const float* params, const float* origin, float* update_momentum, float momentum_mu,
                                     float one_minus_damp, int64_t d, double* out, void* ws, void* stream) {
  using namespace bm;
  const bool att = f_real > 0, l2 = params != nullptr && origin != nullptr;
  if (out == nullptr || ws == nullptr || d < 0 || f_real < 0 || f_real > BM_MAX_ROWS || curv_mode < 0 || curv_mode > 3 ||
      (d > 0 && (sampled_avg == nullptr || honest_avg == nullptr || defense == nullptr || (att && byz == nullptr) ||
                 (curv_mode >= 1 && curv == nullptr) || (curv_mode >= 2 && past_newest == nullptr) ||
                 (curv_mode == 3 && past_oldest == nullptr))))
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  StudyArgs a{sampled_avg, honest_avg, defense, att ? byz : nullptr, curv_mode >= 2 ? past_newest : nullptr,
              curv_mode == 3 ? past_oldest : nullptr, l2 ? params : nullptr, l2 ? origin : nullptr,
              curv_mode >= 1 ? curv : nullptr, att ? attack_avg_out : nullptr, update_momentum, momentum_mu, one_minus_damp};
  const void* ptrs[11] = {a.s, a.h, a.def, a.byz, a.past, a.oldest, a.params, a.origin, a.curv, a.a_out, a.mom};
  const int vec = common_vec_width(ptrs, 11, nullptr);  // null pointers do not constrain the width
  double* partial = static_cast<double*>(ws);
  int nparts = 0;
  int64_t body = 0;
  int rc = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    int grid;
    if (study_burst_eligible(a, curv_mode, vec, nvec)) {
      grid = compute_units();  // one workgroup of 1024 lanes per CU (<= kStudyMaxBlocks partial sets)
      if (grid > kStudyMaxBlocks) grid = kStudyMaxBlocks;
      if (att)
        rc = l2 ? launch_study_burst<true, true>(a, curv_mode, f_real, mu, oldest_weight, nvec, grid, partial, s)
                : launch_study_burst<true, false>(a, curv_mode, f_real, mu, oldest_weight, nvec, grid, partial, s);
      else
        rc = l2 ? launch_study_burst<false, true>(a, curv_mode, f_real, mu, oldest_weight, nvec, grid, partial, s)
                : launch_study_burst<false, false>(a, curv_mode, f_real, mu, oldest_weight, nvec, grid, partial, s);
    } else {
      grid = stream_grid(nvec, kStudyBlock, kStudyMaxBlocks);
      rc = launch_study(a, att, l2, curv_mode, vec, f_real, mu, oldest_weight, nvec, grid, partial, s);
    }
    if (rc != 0) return rc;
    nparts = grid;
    body = nvec * vec;
  }
  if (body < d) {
    StudyArgs t = a;
    auto adv = [body](const float* p) { return p != nullptr ? p + body : nullptr; };
    t.s = adv(a.s);
    t.h = adv(a.h);
    t.def = adv(a.def);
    t.byz = adv(a.byz);
    t.past = adv(a.past);
    t.oldest = adv(a.oldest);
    t.params = adv(a.params);
    t.origin = adv(a.origin);
    t.curv = a.curv != nullptr ? a.curv + body : nullptr;
    t.a_out = a.a_out != nullptr ? a.a_out + body : nullptr;
    t.mom = a.mom != nullptr ? a.mom + body : nullptr;
    const int64_t rest = d - body;
    const int grid = (body == 0) ? stream_grid(rest, kStudyBlock, kStudyMaxBlocks) : 1;
    rc = launch_study(t, att, l2, curv_mode, 1, f_real, mu, oldest_weight, rest, grid,
                      partial + nparts, s);
    if (rc != 0) return rc;
    nparts += grid;
  }
  // d == 0: no partial, the finish kernel writes zeros (every rank of a sharded job reaches its exchange)
  hipLaunchKernelGGL(study_finish_kernel, dim3(kStudyPartial), dim3(64), 0, s, partial, nparts, out);
  BM_LAUNCH_CHECK();
  return 0;
}
