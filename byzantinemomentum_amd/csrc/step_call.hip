// step_call.hip — one simulation step (attack.py:786-868, worker-side momentum) as ONE C call.
//
// The kernels are the ones the Python host mirror launches one by one (byzantinemomentum_amd/step.py);
// this entry point issues the same sequence from C, on the caller's stream, with every collective of a
// dim-sharded job (row norms when clipping, the n x n squared distances, the packed statistics) going
// through the library's RCCL communicator.  At one GPU the host cost of a step is irrelevant (the kernels
// take milliseconds); at 8 ranks they take tens of microseconds each and a dozen Python-side calls plus
// three torch.distributed collectives per step would dominate.
//
// Sequence (d = the rank's slice):
//   [clip]   row squared norms of the sampled gradients -> all-reduce -> clipping factors   attack.py:791-794
//   pass 1   bm_momentum_stats: momentum in place, sampled / honest statistics, Byzantine vector   :800-804,846-847
//   rule     bm_sharded_krum / bm_sharded_bulyan (distances -> all-reduce -> rank -> mean / pass 2) or bm_colwise   :821
//   study    bm_study_stats, ONE pass: statistics of the attack stack and of the defense vector (:848,851-852),
//            Gram of (sampled avg, honest avg, defense, attack avg) + <s, newest past>, <s, C> (:854-866),
//            ||params - origin||^2 (:830), and the curvature combination C <- s + mu * (C - mu^(P-1) * oldest)
//            for the next step (see step.py)
//   pack     every scalar into one vector -> all-gather -> fixed-order sums / maxima -> stats_out
#include "bm_common.h"

namespace bm {

constexpr int kStatSums = 26;   // s2 sd h2 hd d2 a2 ad l2 | gram 4x4 | ex0 ex1
constexpr int kStatMaxes = 4;   // smax hmax dmax amax
constexpr int kStatSlots = 32;  // kStatSums + kStatMaxes, padded
static_assert(kStatSums + kStatMaxes <= kStatSlots, "statistics vector layout");

// scratch scalars of one call, all fp64 on the device
struct StepScalars {
  double out6[6];
  double study[BM_STUDY_SLOTS];  // bm_study_stats
  double rowsq[BM_MAX_ROWS];
  double mine[kStatSlots];
  double all[kStatSlots * BM_MAX_ROWS];  // up to 64 ranks
  float clipf[BM_MAX_ROWS];
};

// direct_out: NULL, or stats_out when there is ONE rank — the reduction over the ranks is then the identity and costs
// no launch of its own (step_reduce_kernel: a NaN maximum stays NaN, a sum of one term is the term)
__global__ void step_pack_kernel(StepScalars* sc, int has_attack, int has_past, int has_l2, double* __restrict__ direct_out) {
  if (threadIdx.x != 0) return;
  double* m = sc->mine;
  const double* st = sc->study;
  for (int i = 0; i < kStatSlots; ++i) m[i] = 0.0;
  m[0] = sc->out6[0];
  m[1] = sc->out6[1];
  m[2] = sc->out6[3];
  m[3] = sc->out6[4];
  m[4] = st[2 * 4 + 2];  // |defense|^2
  m[5] = has_attack ? st[18] : 0.0;
  m[6] = has_attack ? st[19] : 0.0;
  m[7] = has_l2 ? st[22] : 0.0;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) m[8 + a * 4 + b] = (has_attack || (a < 3 && b < 3)) ? st[a * 4 + b] : 0.0;
  if (has_past) {
    m[24] = st[16];
    m[25] = st[17];
  }
  m[kStatSums + 0] = sc->out6[2];
  m[kStatSums + 1] = sc->out6[5];
  m[kStatSums + 2] = st[21];
  m[kStatSums + 3] = has_attack ? st[20] : 0.0;
  if (direct_out != nullptr)
    for (int i = 0; i < kStatSlots; ++i) direct_out[i] = m[i];
}

// stats_out[slot] = sum over ranks in rank order (slots < kStatSums) or NaN-propagating maximum
__global__ void step_reduce_kernel(const double* __restrict__ all, int nranks, double* __restrict__ out) {
  const int slot = threadIdx.x;
  if (slot >= kStatSlots) return;
  double acc = all[slot];
  bool nan = acc != acc;
  for (int r = 1; r < nranks; ++r) {
    const double v = all[r * kStatSlots + slot];
    if (slot < kStatSums) {
      acc += v;
    } else {
      nan |= (v != v);
      acc = v > acc ? v : acc;
    }
  }
  out[slot] = (slot >= kStatSums && nan) ? __builtin_nan("") : acc;
}

struct StepLayout {
  int64_t scalars, ws_step, ws_study, ws_dot, ws_rule, total;
};

static StepLayout step_layout(int n, int64_t d) {
  StepLayout l;
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  int64_t off = 0;
  l.scalars = off;
  off += up((int64_t)sizeof(StepScalars));
  l.ws_step = off;
  off += up(bm_workspace_bytes(BM_WS_STEP, 1, d));
  l.ws_study = off;
  off += up(bm_workspace_bytes(BM_WS_STUDY, 1, d));
  l.ws_dot = off;
  off += up(bm_workspace_bytes(BM_WS_DOT, 1, d));
  l.ws_rule = off;
  off += up(bm_sharded_workspace_bytes(n, d));
  l.total = off;
  return l;
}

}  // namespace bm

extern "C" int bm_step_stats_count(void) { return bm::kStatSlots; }

extern "C" int64_t bm_step_workspace_bytes(int n, int64_t d_local) {
  if (n < 1 || n > BM_MAX_ROWS || d_local < 0) return BM_EINVAL;
  return bm::step_layout(n, d_local).total;
}

extern "C" int bm_step_worker(bm_comm* comm, const bm_step_params* p, const float* const* sampled,
                              float* const* buffers, int64_t d, int64_t d_total, float* defense_out,
                              float* sampled_avg_out,
                              float* honest_avg_out, float* byz_out, float* attack_avg_out,
                              const float* past_newest, float* curv, const float* past_oldest, const float* params,
                              const float* origin, double* stats_out, void* ws, void* stream) {
  using namespace bm;
  if (p == nullptr || sampled == nullptr || buffers == nullptr || stats_out == nullptr || ws == nullptr || d < 0 ||
      d_total < d)
    return BM_EINVAL;
  const int n = p->n, h = p->n - p->f_real, ks = p->ks, fr = p->f_real;
  if (n < 1 || n > BM_MAX_ROWS || h < 1 || ks < h || ks > BM_MAX_ROWS || fr < 0 ||
      (d > 0 && (defense_out == nullptr || sampled_avg_out == nullptr || honest_avg_out == nullptr)) ||
      (fr > 0 && d > 0 && byz_out == nullptr))
    return BM_EINVAL;
  const bool distance_rule = p->rule == BM_RULE_KRUM || p->rule == BM_RULE_BULYAN;
  if (!distance_rule && p->rule != BM_RULE_MEDIAN && p->rule != BM_RULE_TRMEAN && p->rule != BM_RULE_PHOCAS &&
      p->rule != BM_RULE_MEAMED)
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(ws);
  const StepLayout lay = step_layout(n, d);
  StepScalars* sc = reinterpret_cast<StepScalars*>(base + lay.scalars);
  int rc;

  // ---- clipping factors (device scalars, global under sharding) ----
  const float* clipf = nullptr;
  if (p->clip > 0.0f) {
    rc = bm_row_sqnorms(sampled, ks, d, sc->rowsq, base + lay.ws_dot, stream);
    if (rc != 0) return rc;
    rc = bm_allreduce_sum_f64(comm, sc->rowsq, ks, stream);
    if (rc != 0) return rc;
    rc = bm_clip_factors(sc->rowsq, ks, p->clip, sc->clipf, stream);
    if (rc != 0) return rc;
    clipf = sc->clipf;
  }

  // ---- pass 1: momentum, statistics of the sampled and honest stacks, Byzantine vector; what the rule needs rides
  //      along: a coordinate-wise rule itself, or the distance pass of Krum / Bulyan (inside the same kernel at h = 20) ----
  const bool distance_rule_ = p->rule == BM_RULE_KRUM || p->rule == BM_RULE_BULYAN;
  const bool rides_along = fr >= 1;  // the rule (or its distance pass) is fed from the first pass's registers
  const float* rows[BM_MAX_ROWS];
  for (int i = 0; i < h; ++i) rows[i] = buffers[i];
  for (int i = h; i < n; ++i) rows[i] = byz_out;
  const int m = p->m > 0 ? p->m : n - p->f_decl - 2;
  if (rides_along && !distance_rule_) {
    const int op = p->rule == BM_RULE_MEDIAN ? BM_OP_MEDIAN
                   : p->rule == BM_RULE_TRMEAN ? BM_OP_TRMEAN : p->rule == BM_RULE_PHOCAS ? BM_OP_PHOCAS : BM_OP_MEAMED;
    rc = bm_momentum_stats_colwise(sampled, ks, buffers, h, d, p->mu, p->one_minus_damp, clipf, sampled_avg_out,
                                   honest_avg_out, byz_out, p->attack_scale, p->attack_kind, op, p->f_decl, fr,
                                   defense_out, sc->out6, base + lay.ws_step, stream);
    if (rc != 0) return rc;
  } else if (rides_along) {
    // Krum / Bulyan: the squared distances of this shard come out of the first pass (the plan of the distance pass
    // follows the length of the whole vector, d_total, as in bm_sharded_krum); then all-reduce -> rank -> rule
    char* rule_ws = base + lay.ws_rule;
    rc = bm_momentum_stats_sqdist(sampled, ks, buffers, h, d, d_total, p->mu, p->one_minus_damp,
                                  clipf, sampled_avg_out, honest_avg_out, byz_out, p->attack_scale, p->attack_kind, fr,
                                  bm_sharded_sq_slot(rule_ws), sc->out6, base + lay.ws_step,
                                  bm_sharded_pair_workspace(rule_ws), stream);
    if (rc != 0) return rc;
    rc = bm_sharded_rule_from_sq(comm, p->rule, rows, n, d, p->f_decl, m, defense_out, nullptr, rule_ws, stream);
    if (rc != 0) return rc;
  } else {
    rc = bm_momentum_stats(sampled, ks, buffers, h, d, p->mu, p->one_minus_damp, clipf, sampled_avg_out, honest_avg_out,
                           nullptr, p->attack_scale, p->attack_kind, sc->out6, base + lay.ws_step, stream);
    if (rc != 0) return rc;
    // ---- the rule over the honest rows alone (no attack) ----
    switch (p->rule) {
      case BM_RULE_KRUM:
        rc = bm_sharded_krum(comm, rows, n, d, d_total, p->f_decl, m, defense_out, nullptr, base + lay.ws_rule, stream);
        break;
      case BM_RULE_BULYAN:
        rc = bm_sharded_bulyan(comm, rows, n, d, d_total, p->f_decl, m, defense_out, nullptr, base + lay.ws_rule, stream);
        break;
      case BM_RULE_MEDIAN: rc = bm_colwise(BM_OP_MEDIAN, rows, n, d, 0, defense_out, stream); break;
      case BM_RULE_TRMEAN: rc = bm_colwise(BM_OP_TRMEAN, rows, n, d, p->f_decl, defense_out, stream); break;
      case BM_RULE_PHOCAS: rc = bm_colwise(BM_OP_PHOCAS, rows, n, d, p->f_decl, defense_out, stream); break;
      default: rc = bm_colwise(BM_OP_MEAMED, rows, n, d, p->f_decl, defense_out, stream); break;
    }
    if (rc != 0) return rc;
  }

  // ---- the study block in one pass (attack / defense statistics, dots, l2, curvature combination) ----
  int curv_mode = 0;
  if (p->nb_past > 0 && curv != nullptr) {
    if (p->past_count == 0 || past_newest == nullptr)
      curv_mode = 1;
    else
      curv_mode = past_oldest != nullptr ? 3 : 2;
  }
  const bool has_past = curv_mode >= 2;
  const bool has_l2 = params != nullptr && origin != nullptr;
  rc = bm_study_stats(sampled_avg_out, honest_avg_out, defense_out, byz_out, fr, attack_avg_out, past_newest, curv,
                      past_oldest, curv_mode, p->mu, p->oldest_weight, params, origin, d, sc->study,
                      base + lay.ws_study, stream);
  if (rc != 0) return rc;

  // ---- one packed exchange of every scalar ----
  hipLaunchKernelGGL(step_pack_kernel, dim3(1), dim3(64), 0, s, sc, fr > 0 ? 1 : 0, has_past ? 1 : 0, has_l2 ? 1 : 0,
                     comm == nullptr ? stats_out : static_cast<double*>(nullptr));
  BM_LAUNCH_CHECK();
  if (comm == nullptr) return 0;
  const int nranks = bm_comm_size(comm);
  if (nranks > BM_MAX_ROWS) return BM_EINVAL;
  const double* gathered = sc->mine;
  if (comm != nullptr) {
    rc = bm_allgather_f32(comm, reinterpret_cast<const float*>(sc->mine), reinterpret_cast<float*>(sc->all),
                          2 * kStatSlots, stream);  // doubles moved as pairs of 4-byte words
    if (rc != 0) return rc;
    gathered = sc->all;
  }
  hipLaunchKernelGGL(step_reduce_kernel, dim3(1), dim3(64), 0, s, gathered, comm != nullptr ? nranks : 1, stats_out);
  BM_LAUNCH_CHECK();
  return 0;
}
