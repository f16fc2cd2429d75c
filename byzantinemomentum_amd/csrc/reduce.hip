// reduce.hip — row averaging by a device index table, fused stack statistics, fused dot
// products and the multi-vector momentum update.
//
// Replaces (reference, PyTorch):
//   sum(grad for _, grad in scores[:m]).div_(m)          krum.py:80, brute.py:80, aksel.py:64
//   tools.compute_avg_dev_max                             tools/pytorch.py:97-125
//   torch.dot(...)/norm() chains of the study block       attack.py:851-868
//   gmtm.mul_(mu).add_(grad, alpha=1-damp) per worker     attack.py:800-804
// Each is ONE pass over its inputs with every reduction deterministic (fixed trees, fp64
// across lanes/workgroups, no float atomics) and no host synchronisation.
#include "bm_common.h"

namespace bm {

constexpr int kRedBlock = 256;
constexpr int kMaxPartialBlocks = 2048;

// ---------------------------------------------------------------------------
// selected_mean: out = (((0 + R[idx0]) + R[idx1]) + ... ) / m, sequential fp32.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kRedBlock) void selected_mean_kernel(RowTable rows,
                                                                  const int32_t* __restrict__ idx,
                                                                  int m, int64_t nvec, float fm, int nt_result,
                                                                  float* __restrict__ out, int tail) {
  __shared__ const float* sel[BM_MAX_ROWS];
  __shared__ int poisoned;
  if (threadIdx.x == 0) poisoned = 0;
  __syncthreads();
  if (threadIdx.x < m) {
    // (a NEGATIVE index is the Brute search saying "no selection, do not use this", brute.hip: the mean is then NaN
    //  everywhere instead of the average of some rows)
    const int i = load_index_coherent(idx + threadIdx.x);
    if (i < 0) poisoned = 1;
    sel[threadIdx.x] = rows.p[i < 0 ? 0 : i];
  }
  __syncthreads();
  if (poisoned) fm = __builtin_nanf("");
  const int64_t nblk = (nvec + kRedBlock - 1) / kRedBlock;
  for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int64_t v = b * kRedBlock + threadIdx.x;
    if (v >= nvec) continue;
    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.0f;
#pragma unroll 8
    for (int k = 0; k < m; ++k) {
      float t[VEC];
      load_stream<VEC>(sel[k] + v * VEC, t);
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[c] += t[c];
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = acc[c] / fm;
    store_result_policy<VEC>(out + v * VEC, acc, nt_result);
  }
  // the d % VEC trailing columns: one lane each, in the last workgroup (no second launch)
  if (VEC > 1 && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < tail) {
    const int64_t j = nvec * VEC + threadIdx.x;
    float acc = 0.0f;
    for (int k = 0; k < m; ++k) acc += sel[k][j];
    out[j] = acc / fm;
  }
}

// Burst form of the same kernel for long gradients (16-byte columns): one workgroup of 1024 lanes per CU,
// column groups interleaved across the CUs, the results of 9 iterations staged in LDS, a barrier, then written
// back to back — the recipe of colwise_burst_kernel (colwise_kernels.h), where the why is written down.  The
// additions per column are the same, in the same order: the result is bit-identical to the plain form.
constexpr int kMeanBurstThreads = 1024;
constexpr int kMeanBurstSlots = 9;  // 9 x 1024 x 16 B = 144 KB of results next to the pointer table
__global__ __launch_bounds__(kMeanBurstThreads) void selected_mean_burst_kernel(RowTable rows,
                                                                                const int32_t* __restrict__ idx, int m,
                                                                                int64_t nvec, float fm,
                                                                                float* __restrict__ out, int tail) {
  using V = typename VecLoad<4>::T;
  __shared__ V stage[kMeanBurstSlots * kMeanBurstThreads];
  __shared__ const float* sel[BM_MAX_ROWS];
  __shared__ int poisoned;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) poisoned = 0;
  __syncthreads();
  if ((int)tid < m) {
    const int i = load_index_coherent(idx + tid);  // (negative: no selection — the mean is NaN, see selected_mean_kernel)
    if (i < 0) poisoned = 1;
    sel[tid] = rows.p[i < 0 ? 0 : i];
  }
  __syncthreads();
  if (poisoned) fm = __builtin_nanf("");
  const uint32_t nv = (uint32_t)nvec;
  const uint32_t span = gridDim.x * kMeanBurstThreads;
  const uint32_t iters = (nv + span - 1) / span;
  const uint32_t first = blockIdx.x * kMeanBurstThreads + tid;
  for (uint32_t p0 = 0; p0 < iters; p0 += kMeanBurstSlots) {
    const uint32_t p1 = (p0 + kMeanBurstSlots < iters) ? p0 + kMeanBurstSlots : iters;
    for (uint32_t it = p0; it < p1; ++it) {
      const uint32_t v = it * span + first;
      if (v < nv) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
        for (int k = 0; k < m; ++k) {
          float t[4];
          load_stream<4>(sel[k] + (int64_t)v * 4, t);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] += t[c];
        }
        V packed;
#pragma unroll
        for (int c = 0; c < 4; ++c) packed[c] = acc[c] / fm;
        stage[(it - p0) * kMeanBurstThreads + tid] = packed;
      }
    }
    __syncthreads();  // what makes the stores below a burst
    for (uint32_t it = p0; it < p1; ++it) {
      const uint32_t v = it * span + first;
      if (v < nv) __builtin_nontemporal_store(stage[(it - p0) * kMeanBurstThreads + tid], reinterpret_cast<V*>(out) + v);
    }
  }
  // the d % 4 trailing columns: one lane each, in the last workgroup (no second launch: 4.4 us of a C3 aggregation)
  if (blockIdx.x == gridDim.x - 1 && (int)tid < tail) {
    const int64_t j = nvec * 4 + tid;
    float acc = 0.0f;
    for (int k = 0; k < m; ++k) acc += sel[k][j];
    out[j] = acc / fm;
  }
}

template <int VEC>
static int launch_selected_mean(const RowTable& tab, const int32_t* idx, int m, int64_t nvec,
                                float* out, hipStream_t s, int tail = 0) {
  if (nvec <= 0) return 0;
  if constexpr (VEC == 4) {
    const int cus = compute_units();
    // measured (profiles/r02_i_selected_mean_burst.txt): m = 37 at 11.2 M 306.7 -> 284.4 us, m = 18 at 36.5 M
    // 510.7 -> 471.9 us, m = 7 at 9 M 56.7 -> 58.0 us: from 12 rows and 8 iterations per CU on
    if (tuning().mean_burst > 0 && m >= 12 && nvec < ((int64_t)1 << 30) &&
        nvec / ((int64_t)cus * kMeanBurstThreads) >= tuning().mean_burst) {
      hipLaunchKernelGGL(selected_mean_burst_kernel, dim3(cus), dim3(kMeanBurstThreads), 0, s, tab, idx, m, nvec,
                         (float)m, out, tail);
      BM_LAUNCH_CHECK();
      return 0;
    }
  }
  const int grid = stream_grid(nvec, kRedBlock, 256 * 32);
  hipLaunchKernelGGL(selected_mean_kernel<VEC>, dim3(grid), dim3(kRedBlock), 0, s, tab, idx, m, nvec,
                     (float)m, 1, out, tail);
  BM_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// stack_stats: avg (sequential mean of k rows), ||avg||^2, sum_i ||x_i-avg||^2, max|avg|.
// Per-lane fp32 partials cover only a few dozen columns, everything wider is fp64.
// ---------------------------------------------------------------------------
template <int KMAX, int VEC>
__global__ __launch_bounds__(kRedBlock) void stack_stats_kernel(RowTable rows, int k, int64_t nvec,
                                                                float* __restrict__ avg_out,
                                                                float* __restrict__ scaled_out,
                                                                float scale, int attack_kind, int nt_result,
                                                                double* __restrict__ partial) {
  __shared__ double red[kRedBlock / 64];
  const float fk = (float)k;
  float norm2 = 0.0f, dev2 = 0.0f, amax = 0.0f;
  bool seen_nan = false;
  const int64_t stride = (int64_t)gridDim.x * kRedBlock;
  for (int64_t v = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; v < nvec; v += stride) {
    float x[KMAX][VEC];
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
      if (i < k) load_stream<VEC>(rows.p[i] + v * VEC, x[i]);
    float avg[VEC], colq[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      float s = x[0][c];  // grad_avg = samples[0].clone(); add_(...)  (tools/pytorch.py:108-110)
#pragma unroll
      for (int i = 1; i < KMAX; ++i)
        if (i < k) s += x[i][c];
      s = s / fk;
      avg[c] = s;
      norm2 = __builtin_fmaf(s, s, norm2);
      const float as = __builtin_fabsf(s);
      amax = fmaxf(amax, as);
      seen_nan |= (s != s);
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < KMAX; ++i)
        if (i < k) {
          const float df = x[i][c] - s;
          q = __builtin_fmaf(df, df, q);
        }
      dev2 += q;
      colq[c] = q;
    }
    if (avg_out != nullptr) store_result_policy<VEC>(avg_out + v * VEC, avg, nt_result);
    if (scaled_out != nullptr) {
      float sc[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        // empire: grad_att = grad_avg.neg();            little: grad_att = grad_stck.var(dim=0).sqrt_()
        const float dir = ((attack_kind & 15) == BM_ATTACK_LITTLE) ? __builtin_sqrtf(colq[c] / (fk - 1.0f)) : -avg[c];
        const float att = dir * scale;  // grad_att.mul_(factor)
        // byz_grad = grad_avg.add_(grad_att); BM_ATTACK_DIRECTION: grad_att alone
        sc[c] = (attack_kind & BM_ATTACK_DIRECTION) ? att : avg[c] + att;
      }
      store_result_policy<VEC>(scaled_out + v * VEC, sc, nt_result);
    }
  }
  // torch's abs().max() propagates NaN; fmaxf does not
  if (seen_nan) amax = __builtin_nanf("");
  const double n2 = block_reduce_sum<kRedBlock>((double)norm2, red);
  const double d2 = block_reduce_sum<kRedBlock>((double)dev2, red);
  // max: NaN-propagating tree
  float m = amax;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_down(m, off, 64);
    m = (m != m || o != o) ? __builtin_nanf("") : fmaxf(m, o);
  }
  __shared__ float mred[kRedBlock / 64];
  if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float mm = mred[0];
    for (int w = 1; w < kRedBlock / 64; ++w) {
      const float o = mred[w];
      mm = (mm != mm || o != o) ? __builtin_nanf("") : fmaxf(mm, o);
    }
    partial[blockIdx.x * 3 + 0] = n2;
    partial[blockIdx.x * 3 + 1] = d2;
    partial[blockIdx.x * 3 + 2] = (double)mm;
  }
}

// Scalar tail (d % VEC columns) folded into the same partial array as one more "workgroup".
__global__ void stats_finish_kernel(const double* __restrict__ partial, int nparts,
                                    double* __restrict__ out3) {
  // one wave; fixed-order reduction of the per-workgroup partials
  const int lane = threadIdx.x;
  double n2 = 0.0, d2 = 0.0, mx = 0.0;
  bool nan = false;
  for (int b = lane; b < nparts; b += 64) {
    n2 += partial[b * 3 + 0];
    d2 += partial[b * 3 + 1];
    const double m = partial[b * 3 + 2];
    nan |= (m != m);
    mx = m > mx ? m : mx;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n2 += __shfl_down(n2, off, 64);
    d2 += __shfl_down(d2, off, 64);
    const double o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
    nan |= (bool)__shfl_down((int)nan, off, 64);
  }
  if (lane == 0) {
    out3[0] = n2;
    out3[1] = d2;
    out3[2] = nan ? __builtin_nan("") : mx;
  }
}

template <int KMAX, int VEC>
static int launch_stack_stats(const RowTable& tab, int k, int64_t nvec, float* avg, float* scaled,
                              float scale, int kind, double* partial, int grid, hipStream_t s) {
  hipLaunchKernelGGL((stack_stats_kernel<KMAX, VEC>), dim3(grid), dim3(kRedBlock), 0, s, tab, k, nvec,
                     avg, scaled, scale, kind, 1, partial);
  BM_LAUNCH_CHECK();
  return 0;
}

template <int VEC>
static int dispatch_stack_stats(const RowTable& tab, int k, int64_t nvec, float* avg, float* scaled,
                                float scale, int kind, double* partial, int grid, hipStream_t s) {
  if (k <= 8) return launch_stack_stats<8, VEC>(tab, k, nvec, avg, scaled, scale, kind, partial, grid, s);
  if (k <= 16) return launch_stack_stats<16, VEC>(tab, k, nvec, avg, scaled, scale, kind, partial, grid, s);
  if (k <= 24) return launch_stack_stats<24, VEC>(tab, k, nvec, avg, scaled, scale, kind, partial, grid, s);
  if (k <= 32)
    return launch_stack_stats<32, (VEC > 2 ? 2 : VEC)>(tab, k, nvec * (VEC > 2 ? VEC / 2 : 1), avg, scaled,
                                                        scale, kind, partial, grid, s);
  return launch_stack_stats<64, 1>(tab, k, nvec * VEC, avg, scaled, scale, kind, partial, grid, s);
}

// ---------------------------------------------------------------------------
// multi_dot: Gram matrix of <= 4 "core" vectors + dot(core[0], extra[e]) for <= 32 extras.
// ---------------------------------------------------------------------------
constexpr int kMaxCore = 4;
constexpr int kMaxExtra = 32;
constexpr int kDotSlots = kMaxCore * (kMaxCore + 1) / 2 + kMaxExtra;  // 42

struct DotTable {
  const float* core[kMaxCore];
  const float* extra[kMaxExtra];
};

template <int VEC>
__global__ __launch_bounds__(kRedBlock) void multi_dot_kernel(DotTable tab, int nc, int ne,
                                                              int64_t nvec,
                                                              double* __restrict__ partial) {
  __shared__ double red[kRedBlock / 64];
  float acc[kDotSlots];
#pragma unroll
  for (int i = 0; i < kDotSlots; ++i) acc[i] = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kRedBlock;
  for (int64_t v = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; v < nvec; v += stride) {
    float c[kMaxCore][VEC];
#pragma unroll
    for (int a = 0; a < kMaxCore; ++a)
      if (a < nc) load_stream<VEC>(tab.core[a] + v * VEC, c[a]);
    int slot = 0;
#pragma unroll
    for (int a = 0; a < kMaxCore; ++a)
#pragma unroll
      for (int b = a; b < kMaxCore; ++b) {
        if (b < nc) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[slot] = __builtin_fmaf(c[a][e], c[b][e], acc[slot]);
        }
        ++slot;
      }
#pragma unroll
    for (int x = 0; x < kMaxExtra; ++x)
      if (x < ne) {
        float t[VEC];
        load_stream<VEC>(tab.extra[x] + v * VEC, t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[slot + x] = __builtin_fmaf(c[0][e], t[e], acc[slot + x]);
      }
  }
#pragma unroll
  for (int i = 0; i < kDotSlots; ++i) {
    const double r = block_reduce_sum<kRedBlock>((double)acc[i], red);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * kDotSlots + i] = r;
  }
}

__global__ __launch_bounds__(64) void dot_finish_kernel(const double* __restrict__ partial, int nparts,
                                                        int nc, int ne, double* __restrict__ out) {
  // one wave per slot: lane l adds the partials of workgroups l, l+64, ... in order, then a fixed
  // shuffle tree (a single thread walking 1024 partials was 250 us of dependent loads)
  const int s = blockIdx.x, lane = threadIdx.x;
  double tot = 0.0;
  for (int b = lane; b < nparts; b += 64) tot += partial[(int64_t)b * kDotSlots + s];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
  if (lane != 0) return;
  // slot -> (a, b) of the upper triangle of a kMaxCore x kMaxCore matrix, or extra index
  int slot = 0;
  for (int a = 0; a < kMaxCore; ++a)
    for (int b = a; b < kMaxCore; ++b) {
      if (slot == s && b < nc) {
        out[a * nc + b] = tot;
        out[b * nc + a] = tot;
      }
      ++slot;
    }
  const int x = s - kMaxCore * (kMaxCore + 1) / 2;
  if (x >= 0 && x < ne) out[nc * nc + x] = tot;
}

// ---------------------------------------------------------------------------
// row_sqnorms: sq[i] = sum_j rows[i][j]^2 for k rows, every row read once by its own slice of the grid
// (blockIdx.y = row).  Behind cge.py:28-38 (`gradient.norm().item()` per gradient) and the clipping of
// attack.py:776-779,791-794; round 2 took the norms from 4 x 4 Gram blocks (bm_multi_dot, ceil(k/4) launches).
// ---------------------------------------------------------------------------
constexpr int kNormBlocks = 128;  // workgroups per row: k * 128 * 8 bytes of partials fit the BM_WS_DOT workspace
template <int VEC>
__global__ __launch_bounds__(kRedBlock) void row_sqnorms_kernel(RowTable rows, int64_t nvec, double* __restrict__ partial) {
  __shared__ double red[kRedBlock / 64];
  const float* row = rows.p[blockIdx.y];
  float acc[2] = {0.0f, 0.0f};
  double wide = 0.0;  // the fp32 chains are folded into fp64 every 16 iterations: their length does not grow with d
  const int64_t stride = (int64_t)gridDim.x * kRedBlock;
  int64_t v = (int64_t)blockIdx.x * kRedBlock + threadIdx.x;
  int since = 0;
  for (; v + stride < nvec; v += 2 * stride) {  // two loads in flight per lane
    float a[VEC], b[VEC];
    load_stream<VEC>(row + v * VEC, a);
    load_stream<VEC>(row + (v + stride) * VEC, b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      acc[0] = __builtin_fmaf(a[e], a[e], acc[0]);
      acc[1] = __builtin_fmaf(b[e], b[e], acc[1]);
    }
    if (++since == 16) {
      wide += (double)acc[0] + (double)acc[1];
      acc[0] = acc[1] = 0.0f;
      since = 0;
    }
  }
  if (v < nvec) {
    float a[VEC];
    load_stream<VEC>(row + v * VEC, a);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0] = __builtin_fmaf(a[e], a[e], acc[0]);
  }
  const double r = block_reduce_sum<kRedBlock>(wide + ((double)acc[0] + (double)acc[1]), red);
  if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// one wave per row: fixed-order sum of the row's body partials and of its (scalar) tail partial
__global__ __launch_bounds__(64) void row_sqnorms_finish_kernel(const double* __restrict__ body, int nbody,
                                                               const double* __restrict__ tail, int ntail,
                                                               double* __restrict__ out) {
  const int row = blockIdx.x, lane = threadIdx.x;
  double tot = 0.0;
  for (int b = lane; b < nbody; b += 64) tot += body[(int64_t)row * nbody + b];
  for (int b = lane; b < ntail; b += 64) tot += tail[(int64_t)row * ntail + b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
  if (lane == 0) out[row] = tot;
}

// ---------------------------------------------------------------------------
// multi_axpby: y_i = fma(b, x_i, a*y_i) for k vectors; blockIdx.y selects the vector.
// (torch's vectorised `add_(x, alpha=b)` after `mul_(a)` is a*y rounded, then one fused
//  multiply-add.)
// ---------------------------------------------------------------------------
struct AxpbyTable {
  float* y[BM_MAX_ROWS];
  const float* x[BM_MAX_ROWS];
};

// (An unrolled variant with 4 vectors in flight per lane and non-temporal y accesses was measured
//  8 % slower at k = 20, d = 36.5 M: 1.79 ms against 1.66 ms = 5.3 TB/s for this form.)
template <int VEC>
__global__ __launch_bounds__(kRedBlock) void multi_axpby_kernel(AxpbyTable tab, int64_t nvec, float a,
                                                                float b) {
  float* __restrict__ y = tab.y[blockIdx.y];
  const float* __restrict__ x = tab.x[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * kRedBlock;
  for (int64_t v = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; v < nvec; v += stride) {
    float yy[VEC], xx[VEC];
    using T = typename VecLoad<VEC>::T;
    const T yv = *reinterpret_cast<const T*>(y + v * VEC);
    if constexpr (VEC == 1) {
      yy[0] = yv;
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) yy[c] = yv[c];
    }
    load_stream<VEC>(x + v * VEC, xx);
#pragma unroll
    for (int c = 0; c < VEC; ++c) yy[c] = __builtin_fmaf(b, xx[c], a * yy[c]);
    T ov;
    if constexpr (VEC == 1) {
      ov = yy[0];
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) ov[c] = yy[c];
    }
    *reinterpret_cast<T*>(y + v * VEC) = ov;
  }
}

// Stable argsort of n (<= 64) fp64 keys on the device; NaN keys rank last (as +inf).
__global__ __launch_bounds__(64) void stable_argsort_kernel(const double* __restrict__ keys, int n,
                                                            int32_t* __restrict__ order) {
  __shared__ double k[BM_MAX_ROWS];
  const int i = threadIdx.x;
  if (i < n) {
    double v = keys[i];
    if (v != v) v = __builtin_inf();
    k[i] = v;
  }
  __syncthreads();
  if (i < n) {
    const double ki = k[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (k[j] < ki || (k[j] == ki && j < i)) ? 1 : 0;
    order[rank] = i;
  }
}

}  // namespace bm

extern "C" int bm_selected_mean(const float* const* rows, int n, const int32_t* idx, int m,
                                int64_t d, float* out, void* stream) {
  using namespace bm;
  if (rows == nullptr || idx == nullptr || (out == nullptr && d > 0) || n < 1 || n > BM_MAX_ROWS || m < 1 ||
      m > BM_MAX_ROWS || d < 0)
    return BM_EINVAL;
  if (d == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  RowTable tab{};
  for (int i = 0; i < n; ++i) tab.p[i] = rows[i];
  const int vec = common_vec_width(reinterpret_cast<const void* const*>(rows), n, out);
  int64_t body = 0;
  int rc = 0;
  // (the d % VEC trailing columns ride in the last workgroup of the body's launch; a launch of their own only when
  //  there is no body)
  if (vec == 4 && d >= 4) {
    body = d;
    rc = launch_selected_mean<4>(tab, idx, m, d / 4, out, s, (int)(d % 4));
  } else if (vec >= 2 && d >= 2) {
    body = d;
    rc = launch_selected_mean<2>(tab, idx, m, d / 2, out, s, (int)(d % 2));
  }
  if (rc != 0) return rc;
  if (body < d) {
    RowTable tail{};
    for (int i = 0; i < n; ++i) tail.p[i] = rows[i] + body;
    rc = launch_selected_mean<1>(tail, idx, m, d - body, out + body, s);
  }
  return rc;
}

extern "C" int bm_stack_stats(const float* const* rows, int k, int64_t d, float* avg_out,
                              float* scaled_out, float scale, int attack_kind, double* out3, void* ws,
                              void* stream) {
  using namespace bm;
  // d == 0 is legal (an empty trailing shard): the finish kernel then writes zeros, so that every rank
  // of a sharded job reaches its collective
  if (rows == nullptr || out3 == nullptr || ws == nullptr || k < 1 || k > BM_MAX_ROWS || d < 0 ||
      ((attack_kind & ~BM_ATTACK_DIRECTION) != BM_ATTACK_EMPIRE && (attack_kind & ~BM_ATTACK_DIRECTION) != BM_ATTACK_LITTLE))
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  RowTable tab{};
  for (int i = 0; i < k; ++i) tab.p[i] = rows[i];
  double* partial = static_cast<double*>(ws);
  int vec = common_vec_width(reinterpret_cast<const void* const*>(rows), k, avg_out);
  if ((reinterpret_cast<uintptr_t>(scaled_out) & 15u) != 0) vec = (reinterpret_cast<uintptr_t>(scaled_out) & 7u) ? 1 : (vec > 2 ? 2 : vec);
  int64_t body = 0;
  int nparts = 0;
  int rc = 0;
  if (vec >= 2) {
    const int64_t nvec = d / vec;
    if (nvec > 0) {
      const int grid = stream_grid(nvec, kRedBlock, kMaxPartialBlocks - 1);
      rc = (vec == 4)
               ? dispatch_stack_stats<4>(tab, k, nvec, avg_out, scaled_out, scale, attack_kind, partial, grid, s)
               : dispatch_stack_stats<2>(tab, k, nvec, avg_out, scaled_out, scale, attack_kind, partial, grid, s);
      if (rc != 0) return rc;
      nparts = grid;
      body = nvec * vec;
    }
  }
  if (body < d) {
    RowTable tail{};
    for (int i = 0; i < k; ++i) tail.p[i] = rows[i] + body;
    const int64_t rest = d - body;
    const int grid = (body == 0) ? stream_grid(rest, kRedBlock, kMaxPartialBlocks) : 1;
    rc = dispatch_stack_stats<1>(tail, k, rest, avg_out ? avg_out + body : nullptr,
                                 scaled_out ? scaled_out + body : nullptr, scale, attack_kind,
                                 partial + (int64_t)nparts * 3, grid, s);
    if (rc != 0) return rc;
    nparts += grid;
  }
  hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(64), 0, s, partial, nparts, out3);
  BM_LAUNCH_CHECK();
  return 0;
}

extern "C" int bm_multi_dot(const float* const* core, int nc, const float* const* extra, int ne,
                            int64_t d, double* out, void* ws, void* stream) {
  using namespace bm;
  if (core == nullptr || out == nullptr || ws == nullptr || nc < 1 || nc > kMaxCore || ne < 0 ||
      ne > kMaxExtra || (ne > 0 && extra == nullptr) || d < 0)
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DotTable tab{};
  uintptr_t bits = 0;
  for (int i = 0; i < nc; ++i) {
    tab.core[i] = core[i];
    bits |= reinterpret_cast<uintptr_t>(core[i]);
  }
  for (int i = 0; i < ne; ++i) {
    tab.extra[i] = extra[i];
    bits |= reinterpret_cast<uintptr_t>(extra[i]);
  }
  double* partial = static_cast<double*>(ws);
  const int vec = (bits & 15u) == 0 ? 4 : ((bits & 7u) == 0 ? 2 : 1);
  int nparts = 0;
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    const int grid = stream_grid(nvec, kRedBlock, 1024);
    if (vec == 4)
      hipLaunchKernelGGL(multi_dot_kernel<4>, dim3(grid), dim3(kRedBlock), 0, s, tab, nc, ne, nvec,
                         partial);
    else
      hipLaunchKernelGGL(multi_dot_kernel<2>, dim3(grid), dim3(kRedBlock), 0, s, tab, nc, ne, nvec,
                         partial);
    BM_LAUNCH_CHECK();
    nparts = grid;
    body = nvec * vec;
  }
  if (body < d) {
    DotTable tail = tab;
    for (int i = 0; i < nc; ++i) tail.core[i] += body;
    for (int i = 0; i < ne; ++i) tail.extra[i] += body;
    const int64_t rest = d - body;
    const int grid = (body == 0) ? stream_grid(rest, kRedBlock, 1024) : 1;
    hipLaunchKernelGGL(multi_dot_kernel<1>, dim3(grid), dim3(kRedBlock), 0, s, tail, nc, ne, rest,
                       partial + (int64_t)nparts * kDotSlots);
    BM_LAUNCH_CHECK();
    nparts += grid;
  }
  hipLaunchKernelGGL(dot_finish_kernel, dim3(kDotSlots), dim3(64), 0, s, partial, nparts, nc, ne, out);
  BM_LAUNCH_CHECK();
  return 0;
}

extern "C" int bm_multi_axpby(float* const* y, const float* const* x, int k, int64_t d, float a,
                              float b, void* stream) {
  using namespace bm;
  if (y == nullptr || x == nullptr || k < 1 || k > BM_MAX_ROWS || d < 0) return BM_EINVAL;
  if (d == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  AxpbyTable tab{};
  uintptr_t bits = 0;
  for (int i = 0; i < k; ++i) {
    tab.y[i] = y[i];
    tab.x[i] = x[i];
    bits |= reinterpret_cast<uintptr_t>(y[i]) | reinterpret_cast<uintptr_t>(x[i]);
  }
  const int vec = (bits & 15u) == 0 ? 4 : ((bits & 7u) == 0 ? 2 : 1);
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    const int grid = stream_grid(nvec, kRedBlock, 2048);
    if (vec == 4)
      hipLaunchKernelGGL(multi_axpby_kernel<4>, dim3(grid, k), dim3(kRedBlock), 0, s, tab, nvec, a, b);
    else
      hipLaunchKernelGGL(multi_axpby_kernel<2>, dim3(grid, k), dim3(kRedBlock), 0, s, tab, nvec, a, b);
    BM_LAUNCH_CHECK();
    body = nvec * vec;
  }
  if (body < d) {
    AxpbyTable tail = tab;
    for (int i = 0; i < k; ++i) {
      tail.y[i] += body;
      tail.x[i] += body;
    }
    const int64_t rest = d - body;
    const int grid = stream_grid(rest, kRedBlock, 2048);
    hipLaunchKernelGGL(multi_axpby_kernel<1>, dim3(grid, k), dim3(kRedBlock), 0, s, tail, rest, a, b);
    BM_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int bm_stable_argsort(const double* keys, int n, int32_t* order_out, void* stream) {
  using namespace bm;
  if (keys == nullptr || order_out == nullptr || n < 1 || n > BM_MAX_ROWS) return BM_EINVAL;
  hipLaunchKernelGGL(stable_argsort_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     keys, n, order_out);
  BM_LAUNCH_CHECK();
  return 0;
}

extern "C" int bm_row_sqnorms(const float* const* rows, int k, int64_t d, double* sq_out, void* ws, void* stream) {
  using namespace bm;
  if (rows == nullptr || sq_out == nullptr || ws == nullptr || k < 1 || k > BM_MAX_ROWS || d < 0) return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  RowTable tab{};
  for (int i = 0; i < k; ++i) tab.p[i] = rows[i];
  const int vec = common_vec_width(reinterpret_cast<const void* const*>(rows), k, nullptr);
  double* body_part = static_cast<double*>(ws);
  double* tail_part = body_part + (int64_t)BM_MAX_ROWS * kNormBlocks;
  int nbody = 0, ntail = 0;
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    nbody = stream_grid(nvec, kRedBlock, kNormBlocks);
    if (vec == 4)
      hipLaunchKernelGGL(row_sqnorms_kernel<4>, dim3(nbody, k), dim3(kRedBlock), 0, s, tab, nvec, body_part);
    else
      hipLaunchKernelGGL(row_sqnorms_kernel<2>, dim3(nbody, k), dim3(kRedBlock), 0, s, tab, nvec, body_part);
    BM_LAUNCH_CHECK();
    body = nvec * vec;
  }
  if (body < d) {
    RowTable tail{};
    for (int i = 0; i < k; ++i) tail.p[i] = rows[i] + body;
    const int64_t rest = d - body;
    ntail = (body == 0) ? stream_grid(rest, kRedBlock, kNormBlocks) : 1;
    hipLaunchKernelGGL(row_sqnorms_kernel<1>, dim3(ntail, k), dim3(kRedBlock), 0, s, tail, rest, tail_part);
    BM_LAUNCH_CHECK();
  }
  // d == 0: no partial at all, the finish kernel writes zeros
  hipLaunchKernelGGL(row_sqnorms_finish_kernel, dim3(k), dim3(64), 0, s, body_part, nbody, tail_part, ntail, sq_out);
  BM_LAUNCH_CHECK();
  return 0;
}
