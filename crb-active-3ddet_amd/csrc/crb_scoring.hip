// CRB stage 3 — greedy point-cloud-density balancing (row a28 of SURVEY §8) as two launches on gfx950.
//
// Replaces the Python triple loop of pcdet/query_strategies/crb_sampling.py:276-331: for each of SELECT_NUMS picks, for
// every remaining candidate frame and every class, an sklearn KernelDensity(gaussian, bw).fit + score_samples(400) +
// scipy.stats.entropy(uniform prior, exp(logp)) on the CPU (~74k KDE fits per selection round).
//
// Reformulation (same numbers, f64 like sklearn/scipy):
//   KDE(x) over (already selected boxes of class c) U (candidate i's boxes of class c)
//        = (A_c(x) + B_ic(x)) / ((n_c + m_ic) * bw * sqrt(2 pi)),   K(u) = exp(-u^2 / 2)
//   A_c(x) = sum over selected boxes K((x - d)/bw)   — a running sum, updated once per pick
//   B_ic(x) = sum over candidate boxes                — computed ONCE for all candidates (kernel 1)
//   KL = sum_x pk * log(pk / qk) with pk, qk normalised to 1 over the 400-point axis (scipy.stats.entropy semantics,
//   terms with pk == 0 contribute 0, qk == 0 under pk > 0 gives +inf), proportion = 2/pi * atan(pi/2 * KL),
//   absent class -> proportion 1; score_i = mean_c(1 - proportion); pick = first arg-max in candidate order.
// Kernel 2 runs ALL picks in one launch with one 1024-thread workgroup: A_c lives in LDS, each wave owns
// (candidate, class) tasks with a 64-lane strided pass over the 400 axis points and wave reductions; no host round
// trip per pick.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

constexpr int AX = 400;          // axis points per class (crb_sampling.py:259)
constexpr int MAXC = 8;          // classes

// B[(i*C + c)*AX + x], cnt[i*C + c]
__global__ __launch_bounds__(256) void kde_candidate_sums(const float* __restrict__ dens, const int* __restrict__ lab,
                                                          int dmax, int C, const double* __restrict__ xaxis, double bw,
                                                          double* __restrict__ Bsum, int* __restrict__ cnt) {
  const int i = blockIdx.x, c = blockIdx.y;
  const float* d = dens + (int64_t)i * dmax;
  const int* l = lab + (int64_t)i * dmax;
  __shared__ int n_sh;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int k = 0; k < dmax; ++k) n += (l[k] == c + 1);
    n_sh = n;
    cnt[i * C + c] = n;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < AX; x += 256) {
    const double xv = xaxis[c * AX + x];
    double s = 0.0;
    if (n_sh > 0)
      for (int k = 0; k < dmax; ++k)
        if (l[k] == c + 1) {
          const double u = (xv - (double)d[k]) / bw;
          s += exp(-0.5 * u * u);
        }
    Bsum[((int64_t)i * C + c) * AX + x] = s;
  }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

__global__ __launch_bounds__(1024) void density_greedy_kernel(int N, int C, int select_nums,
                                                              const double* __restrict__ Bsum,
                                                              const int* __restrict__ cnt,
                                                              const double* __restrict__ prior, double bw,
                                                              int* __restrict__ order, double* __restrict__ best_scores,
                                                              double* __restrict__ prop_ws /* N*C */,
                                                              unsigned char* __restrict__ used /* N */) {
  __shared__ double A[MAXC * AX];
  __shared__ double pk[MAXC * AX];
  __shared__ int nsel[MAXC];
  __shared__ double red_val[16];
  __shared__ int red_idx[16];
  __shared__ int pick_sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int t = tid; t < C * AX; t += 1024) A[t] = 0.0;
  for (int t = tid; t < N; t += 1024) used[t] = 0;
  if (tid < C) nsel[tid] = 0;
  __syncthreads();
  // normalised prior per class
  for (int c = wave; c < C; c += 16) {
    double s = 0.0;
    for (int x = lane; x < AX; x += 64) s += prior[c * AX + x];
    s = wave_sum_d(s);
    for (int x = lane; x < AX; x += 64) pk[c * AX + x] = prior[c * AX + x] / s;
  }
  __syncthreads();
  const double norm_c = bw * 2.5066282746310002;   // sqrt(2 pi)
  const int picks = select_nums < N ? select_nums : N;
  for (int j = 0; j < picks; ++j) {
    int pick;
    if (j == 0) {
      pick = 0;                                     // the reference starts from the first candidate (:277-286)
      if (tid == 0 && best_scores) best_scores[0] = -1.0;
    } else {
      // (candidate, class) tasks, one wave each
      for (int task = wave; task < N * C; task += 16) {
        const int i = task / C, c = task - i * C;
        if (used[i]) continue;
        const int m = cnt[i * C + c];
        double prop = 1.0;
        if (m > 0) {
          const double* B = Bsum + (int64_t)task * AX;
          const double denom = (double)(nsel[c] + m) * norm_c;
          double q[(AX + 63) / 64];
          double sq = 0.0;
#pragma unroll
          for (int t = 0; t < (AX + 63) / 64; ++t) {
            const int x = lane + t * 64;
            q[t] = (x < AX) ? (A[c * AX + x] + B[x]) / denom : 0.0;
            sq += q[t];
          }
          sq = wave_sum_d(sq);
          double kl = 0.0;
#pragma unroll
          for (int t = 0; t < (AX + 63) / 64; ++t) {
            const int x = lane + t * 64;
            if (x < AX) {
              const double p = pk[c * AX + x];
              if (p > 0.0) {
                const double qn = q[t] / sq;
                kl += (qn > 0.0) ? p * log(p / qn) : INFINITY;
              }
            }
          }
          kl = wave_sum_d(kl);
          prop = 0.6366197723675814 * atan(1.5707963267948966 * kl);      // 2/pi * atan(pi/2 * KL)
          // degenerate prior (the class's 95 % density interval has zero width: scipy's uniform.pdf(scale = 0) is NaN, so is
          // the reference's entropy and inverse_coff, and `NaN > best` never holds, crb_sampling.py:256-259,317): the
          // candidate's score becomes NaN and it is never preferred; if every candidate is NaN the first unused one is taken
          // (the reference would crash on best_frame_index = None there)
          if (pk[c * AX] != pk[c * AX]) prop = NAN;
        }
        if (lane == 0) prop_ws[task] = prop;
      }
      __threadfence_block();
      __syncthreads();
      // score + first arg-max
      double bv = -2.0;
      int bi = 0x7fffffff;
      for (int i = tid; i < N; i += 1024) {
        if (used[i]) continue;
        double s = 0.0;
        for (int c = 0; c < C; ++c) s += 1.0 - prop_ws[i * C + c];
        s = s / (double)C;
        if (s > bv || (s == bv && i < bi)) { bv = s; bi = i; }
      }
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) {
        double ov = __shfl_xor(bv, s, 64);
        int oi = __shfl_xor(bi, s, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (lane == 0) { red_val[wave] = bv; red_idx[wave] = bi; }
      __syncthreads();
      if (tid == 0) {
        double v = red_val[0];
        int ix = red_idx[0];
        for (int w = 1; w < 16; ++w)
          if (red_val[w] > v || (red_val[w] == v && red_idx[w] < ix)) { v = red_val[w]; ix = red_idx[w]; }
        // the reference initialises best = -1 and requires strictly greater (:292,317): scores are >= 0 so a
        // candidate always wins; NaN scores (degenerate priors) never win there either -> fall back to the first unused
        pick_sh = ix;
        if (best_scores) best_scores[j] = v;
      }
      __syncthreads();
      pick = pick_sh;
      if (pick == 0x7fffffff) {                     // every remaining score was NaN: first unused candidate
        if (tid == 0) {
          int f = 0;
          while (f < N && used[f]) ++f;
          pick_sh = f;
        }
        __syncthreads();
        pick = pick_sh;
      }
    }
    // commit the pick: order, running sums, counts
    if (tid == 0) { order[j] = pick; used[pick] = 1; }
    for (int t = tid; t < C * AX; t += 1024) A[t] += Bsum[(int64_t)pick * C * AX + t];
    if (tid < C) nsel[tid] += cnt[pick * C + tid];
    __threadfence_block();
    __syncthreads();
  }
  for (int j = picks + tid; j < select_nums; j += 1024) order[j] = -1;
}

}  // namespace

extern "C" int64_t crb_density_greedy_workspace_bytes(int n_candidates, int num_class) {
  int64_t b = 0;
  b += crb_align_up((int64_t)n_candidates * num_class * AX * 8, 256);   // Bsum
  b += crb_align_up((int64_t)n_candidates * num_class * 4, 256);        // cnt
  b += crb_align_up((int64_t)n_candidates * num_class * 8, 256);        // prop
  b += crb_align_up((int64_t)n_candidates, 256);                        // used
  return b + 256;
}

extern "C" int crb_density_greedy(const float* densities, const int32_t* labels, int n_candidates, int dmax,
                                  int num_class, const double* xaxis, const double* prior, double bandwidth,
                                  int select_nums, int32_t* order, double* best_scores, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  if (n_candidates <= 0 || dmax <= 0 || num_class <= 0 || num_class > MAXC || select_nums <= 0 || bandwidth <= 0.0)
    return CRB_ERR_ARG;
  CrbArena a(workspace, (size_t)workspace_bytes);
  double* Bsum = a.take<double>((int64_t)n_candidates * num_class * AX);
  int* cnt = a.take<int>((int64_t)n_candidates * num_class);
  double* prop = a.take<double>((int64_t)n_candidates * num_class);
  unsigned char* used = a.take<unsigned char>(n_candidates);
  if (!a.ok) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(kde_candidate_sums, dim3(n_candidates, num_class), dim3(256), 0, st, densities, labels, dmax,
                     num_class, xaxis, bandwidth, Bsum, cnt);
  hipLaunchKernelGGL(density_greedy_kernel, dim3(1), dim3(1024), 0, st, n_candidates, num_class, select_nums, Bsum, cnt,
                     prior, bandwidth, order, best_scores, prop, used);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
