// Active-learning label update on the GPU: the per-sample scoring of /root/reference/update_label.py:125-169
// (get_uncert_rank) and the pseudo-label re-derivation of :85-123 (renew_label), with their helpers from
// /root/reference/utils/utils_hual.py (fill_isactivate :37-59, get_segment :63-76, center_width_gauss :79-89,
// get_distance_score(_shift) :92-124, get_uncert_model :144-161).  The reference walks the training set sample by sample
// in Python (and re-sorts inside the loop); here one workgroup owns one sample, the whole training set is one launch.
//
// Arithmetic types follow numpy's in the reference, because the results are argmax indices:
//   * gaussians: float32 on the float32 grid np.linspace(-1, 1, T) (python scalars are weak operands -> cast to f32),
//     normalised by their maximum over ALL T grid points, then scaled by width/vlen;
//   * model uncertainty: float32 sigmoid differences; probabilities: float32 1/(1+exp(-x));
//   * score mixtures / uncert_frame: float64 sums of float32 terms.
// Active points arrive as one CSR list per sample (frame index + positive flag), in the order they were annotated.
#include "al.h"
#include "prof.h"

using namespace hual;

#define AL_THREADS 256
#define AL_MAX_SEG 64

namespace {

struct ApInfo {
  int npos, nneg;
  int lo, hi;          // hull of the positive points
  int negL, negR;      // nearest negative left of lo (-1: none) / right of hi (INT_MAX: none)
};

__device__ __forceinline__ ApInfo scan_ap(const int32_t* idx, const int8_t* pos, int n) {
  ApInfo a;
  a.npos = 0; a.nneg = 0; a.lo = 0x7fffffff; a.hi = -1; a.negL = -1; a.negR = 0x7fffffff;
  for (int k = 0; k < n; ++k) {
    if (pos[k]) { ++a.npos; a.lo = min(a.lo, idx[k]); a.hi = max(a.hi, idx[k]); }
    else ++a.nneg;
  }
  if (a.npos > 0) {
    for (int k = 0; k < n; ++k) {
      if (pos[k]) continue;
      if (idx[k] < a.lo) a.negL = max(a.negL, idx[k]);
      if (idx[k] > a.hi) a.negR = min(a.negR, idx[k]);
    }
  }
  return a;
}

// fill_isactivate (utils_hual.py:37-59) evaluated at frame t
__device__ __forceinline__ int isactive_at(const ApInfo& a, const int32_t* idx, const int8_t* pos, int n, int t, int vlen) {
  if (t >= vlen) return -100;
  if (a.npos > 0) {
    if (t <= a.negL || t >= a.negR) return -1;
    return (t >= a.lo && t <= a.hi) ? 1 : 0;
  }
  for (int k = 0; k < n; ++k)
    if (!pos[k] && idx[k] == t) return -1;
  return 0;
}

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max64(v);
  __syncthreads();                                    // protects `red` against the previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = red[0];
  for (int w = 1; w < AL_THREADS / 64; ++w) m = fmaxf(m, red[w]);
  return m;
}

// grid point t of np.linspace(-1, 1, T, dtype=float32): float64 arange * step + start, last point = stop, then cast
__device__ __forceinline__ float grid_x(int t, int T) {
  if (t == T - 1) return 1.0f;
  const double step = 2.0 / (double)(T - 1);
  return (float)((double)t * step + (-1.0));
}

// center_width_gauss (utils_hual.py:79-89) for all T frames into out[] (LDS).  Block-uniform call.
__device__ void gauss_block(double center, double width, int vlen, int T, float* out, float* red) {
  double sig = (double)vlen / (double)T;
  sig *= width / (double)vlen * 0.4;
  const double u = (center / (double)(T - 1)) * 2.0 - 1.0;
  const float uf = (float)u;
  const float den = (float)(2.0 * (sig * sig));
  const float nrm = (float)(sqrt(2.0 * 3.141592653589793) * sig);
  const float peak = (float)(width / (double)vlen);
  float mx = -1.0f;
  for (int t = threadIdx.x; t < T; t += AL_THREADS) {
    const float d = grid_x(t, T) - uf;
    const float w = expf(-(d * d) / den) / nrm;
    out[t] = w;
    mx = fmaxf(mx, w);
  }
  mx = block_max(mx, red);
  for (int t = threadIdx.x; t < T; t += AL_THREADS) out[t] = t < vlen ? (out[t] / mx) * peak : 0.0f;
  __syncthreads();
}

// get_segment (utils_hual.py:63-76): maximal runs of isactive == 0, found by one thread (T <= 1024, a few hundred cycles)
__device__ void find_segments(const ApInfo& a, const int32_t* idx, const int8_t* pos, int n, int vlen, int T, int* seg,
                              int* nseg) {
  if (threadIdx.x == 0) {
    int k = 0, start = -1;
    for (int t = 0; t <= T; ++t) {
      const int v = t < T ? isactive_at(a, idx, pos, n, t, vlen) : -100;
      if (v == 0 && start < 0) start = t;
      else if (v != 0 && start >= 0) {
        if (k < AL_MAX_SEG) { seg[2 * k] = start; seg[2 * k + 1] = t - 1; ++k; }
        start = -1;
      }
    }
    *nseg = k;
  }
  __syncthreads();
}

// get_distance_score / one half of get_distance_score_shift: dist[t] (float32 values) for center shifted by
// width*shift/2 (shift = 0: utils_hual.py:92-103)
__device__ void distance_block(const int* seg, int nseg, double shift, int vlen, int T, float* dist, float* tmp, float* red) {
  for (int t = threadIdx.x; t < T; t += AL_THREADS) dist[t] = 0.0f;
  __syncthreads();
  for (int s = 0; s < nseg; ++s) {
    const int a = seg[2 * s], b = seg[2 * s + 1];
    const double width = (double)(b - a + 1);
    const double center = (double)(b - a) / 2.0 + (double)a + width * shift / 2.0;
    gauss_block(center, width, vlen, T, tmp, red);
    for (int t = a + (int)threadIdx.x; t <= b; t += AL_THREADS) dist[t] = tmp[t];
    __syncthreads();
  }
}

__device__ __forceinline__ float sigmoid_np(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace

// ------------------------------------------------------------------------------------------------------
// get_uncert_rank body (update_label.py:125-169) for one sample per block
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AL_THREADS) void al_score_kernel(AlScoreArgs a) {
  extern __shared__ float lds[];            // dist[T] | tmp[T]
  __shared__ float red[AL_THREADS / 64];
  __shared__ double redd[AL_THREADS / 64];
  __shared__ double vs[AL_THREADS / 64];
  __shared__ int redi[AL_THREADS / 64];
  __shared__ int seg[2 * AL_MAX_SEG];
  __shared__ int nseg;
  const int n = blockIdx.x;
  const int T = a.tlen[n], V = a.vlen[n];
  float* dist = lds;
  float* tmp = lds + a.ld;
  const int ap0 = a.ap_off[n], napn = a.ap_off[n + 1] - ap0;
  const int32_t* aidx = a.ap_idx + ap0;
  const int8_t* apos = a.ap_pos + ap0;
  const ApInfo ap = scan_ap(aidx, apos, napn);
  find_segments(ap, aidx, apos, napn, V, T, seg, &nseg);
  distance_block(seg, nseg, 0.0, V, T, dist, tmp, red);

  const size_t row = (size_t)n * a.ld;
  double vsum = 0.0;
  double best = -1.0;
  int besti = 0x7fffffff;
  for (int t = threadIdx.x; t < T; t += AL_THREADS) {
    a.sprob[row + t] = sigmoid_np(a.s0[row + t]);
    a.eprob[row + t] = sigmoid_np(a.e0[row + t]);
    float um = 0.0f;
    if (t < V)
      um = fabsf(sigmoid_np(a.s1[row + t]) - sigmoid_np(a.s2[row + t])) + fabsf(sigmoid_np(a.e1[row + t]) - sigmoid_np(a.e2[row + t]));
    vsum += (double)um;
    const double uf = (double)dist[t] + (double)(um * a.coff_uncert);
    a.uncert_frame[row + t] = uf;
    if (uf > best) { best = uf; besti = t; }       // ascending t per thread: first maximum kept
  }
  // block reduction: sum of the model uncertainty; argmax of uncert_frame with first-index ties
  for (int o = 32; o > 0; o >>= 1) {
    vsum += __shfl_xor(vsum, o);
    const double ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(besti, o);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if ((threadIdx.x & 63) == 0) { vs[threadIdx.x >> 6] = vsum; redd[threadIdx.x >> 6] = best; redi[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < AL_THREADS / 64; ++w) {
      s += vs[w];
      if (redd[w] > best || (redd[w] == best && redi[w] < besti)) { best = redd[w]; besti = redi[w]; }
    }
    a.uncert_video[n] = (float)s;
    a.observe[n] = besti;
  }
}

// ------------------------------------------------------------------------------------------------------
// renew_label (update_label.py:85-123) for one selected sample per block
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AL_THREADS) void al_renew_kernel(AlRenewArgs a) {
  extern __shared__ float lds[];            // ds[T] | de[T] | tmp[T] | (8-byte aligned) ss[T] | es[T] doubles
  __shared__ float red[AL_THREADS / 64];
  __shared__ double redd[2][AL_THREADS / 64];
  __shared__ int redi[2][AL_THREADS / 64];
  __shared__ int seg[2 * AL_MAX_SEG];
  __shared__ int nseg;
  const int n = a.sel ? a.sel[blockIdx.x] : (int)blockIdx.x;
  const int T = a.tlen[n], V = a.vlen[n];
  const int ldp = (a.ld + 1) & ~1;
  float* ds = lds;
  float* de = lds + ldp;
  float* tmp = lds + 2 * ldp;
  double* ss = reinterpret_cast<double*>(lds + 4 * ldp);
  double* es = ss + ldp;
  const int ap0 = a.ap_off[n], napn = a.ap_off[n + 1] - ap0;
  const int32_t* aidx = a.ap_idx + ap0;
  const int8_t* apos = a.ap_pos + ap0;
  const ApInfo ap = scan_ap(aidx, apos, napn);
  const bool has_pos = ap.npos > 0;
  const double a1 = has_pos ? a.coff[0] : a.coff[3];
  const float a2 = (float)(has_pos ? a.coff[1] : a.coff[4]);
  const float a3 = (float)(has_pos ? a.coff[2] : a.coff[5]);
  const double shift = has_pos ? -0.3 : 0.9;
  const size_t row = (size_t)n * a.ld;

  find_segments(ap, aidx, apos, napn, V, T, seg, &nseg);
  distance_block(seg, nseg, -shift, V, T, ds, tmp, red);     // start: centre - width*shift/2
  distance_block(seg, nseg, shift, V, T, de, tmp, red);      // end:   centre + width*shift/2
  // score = distance*a1 + prob*a2 + gaussian around the old index * a3   (float64 + float32 + float32)
  gauss_block((double)a.old_idx[2 * n], 0.5 * (double)V, V, T, tmp, red);
  for (int t = threadIdx.x; t < T; t += AL_THREADS)
    ss[t] = ((double)ds[t] * a1 + (double)(a.sprob[row + t] * a2)) + (double)(tmp[t] * a3);
  __syncthreads();
  gauss_block((double)a.old_idx[2 * n + 1], 0.5 * (double)V, V, T, tmp, red);
  for (int t = threadIdx.x; t < T; t += AL_THREADS)
    es[t] = ((double)de[t] * a1 + (double)(a.eprob[row + t] * a2)) + (double)(tmp[t] * a3);
  __syncthreads();

  double bs = -1.0, be = -1.0;
  int bsi = 0x7fffffff, bei = 0x7fffffff;
  if (has_pos) {
    // mask_activepoints, positive branch (update_label.py:69-82), then two plain argmaxes
    for (int t = threadIdx.x; t < T; t += AL_THREADS) {
      double s = ss[t], e = es[t];
      if (t > ap.lo || t <= ap.negL) s = 0.0;
      if (t < ap.hi || t >= ap.negR) e = 0.0;
      if (s > bs) { bs = s; bsi = t; }
      if (e > be) { be = e; bei = t; }
    }
  } else {
    // negative branch: damp around every negative point, then the best span that contains no negative point
    for (int k = 0; k < napn; ++k) {
      gauss_block((double)aidx[k], 0.3 * (double)V, V, T, tmp, red);
      for (int t = threadIdx.x; t < T; t += AL_THREADS) {
        const double m = (double)(1.0f - tmp[t]);
        ss[t] = m * ss[t];
        es[t] = m * es[t];
      }
      __syncthreads();
    }
    for (int t = threadIdx.x; t < T; t += AL_THREADS) {
      double rmax = 0.0, cmax = 0.0;
      bool cut = t >= V;
      int lo = -1, hi = V;                    // nearest cuts around t (update_label.py:113: sorted(neg + [-1, vlen]))
      for (int k = 0; k < napn; ++k) {
        const int c = aidx[k];
        if (c == t) cut = true;
        if (c < t) lo = max(lo, c);
        if (c > t) hi = min(hi, c);
      }
      if (!cut) {
        const double st = ss[t], et = es[t];
        for (int j = t; j < hi; ++j) rmax = fmax(rmax, st * es[j]);
        for (int i = lo + 1; i <= t; ++i) cmax = fmax(cmax, ss[i] * et);
      }
      if (rmax > bs) { bs = rmax; bsi = t; }
      if (cmax > be) { be = cmax; bei = t; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double obs = __shfl_xor(bs, o), obe = __shfl_xor(be, o);
    const int osi = __shfl_xor(bsi, o), oei = __shfl_xor(bei, o);
    if (obs > bs || (obs == bs && osi < bsi)) { bs = obs; bsi = osi; }
    if (obe > be || (obe == be && oei < bei)) { be = obe; bei = oei; }
  }
  if ((threadIdx.x & 63) == 0) {
    redd[0][threadIdx.x >> 6] = bs; redi[0][threadIdx.x >> 6] = bsi;
    redd[1][threadIdx.x >> 6] = be; redi[1][threadIdx.x >> 6] = bei;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 0; w < AL_THREADS / 64; ++w) {
      if (redd[0][w] > bs || (redd[0][w] == bs && redi[0][w] < bsi)) { bs = redd[0][w]; bsi = redi[0][w]; }
      if (redd[1][w] > be || (redd[1][w] == be && redi[1][w] < bei)) { be = redd[1][w]; bei = redi[1][w]; }
    }
    a.new_idx[2 * n] = bsi;
    a.new_idx[2 * n + 1] = bei;
  }
}

namespace hual {

int launch_al_score(const AlScoreArgs& a, hipStream_t s) {
  HUAL_REQUIRE(a.s0 && a.e0 && a.s1 && a.e1 && a.s2 && a.e2 && a.vlen && a.tlen && a.ap_off, "al_score: null input");
  HUAL_REQUIRE(a.sprob && a.eprob && a.uncert_frame && a.uncert_video && a.observe, "al_score: null output");
  HUAL_REQUIRE(a.N > 0 && a.ld >= 2 && a.ld <= HUAL_AL_MAX_T, "al_score: need N > 0 and 2 <= ld <= 1024");
  HUAL_LAUNCH(0.0, 40.0 * a.N * a.ld, al_score_kernel, dim3(a.N), dim3(AL_THREADS), 2 * a.ld * sizeof(float), s, a);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_al_renew(const AlRenewArgs& a, int nsel, hipStream_t s) {
  HUAL_REQUIRE(a.sprob && a.eprob && a.vlen && a.tlen && a.ap_off && a.old_idx && a.new_idx, "al_renew: null pointer");
  HUAL_REQUIRE(a.ld >= 2 && a.ld <= HUAL_AL_MAX_T, "al_renew: need 2 <= ld <= 1024");
  if (nsel <= 0) return 0;
  const int ldp = (a.ld + 1) & ~1;
  HUAL_LAUNCH(0.0, 16.0 * nsel * a.ld, al_renew_kernel, dim3(nsel), dim3(AL_THREADS), 4 * ldp * sizeof(float) + 2 * ldp * sizeof(double),
              s, a);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual

extern "C" {

int hual_al_score(const hual_al_set* set, const float* s0, const float* e0, const float* s1, const float* e1,
                  const float* s2, const float* e2, float coff_uncert, float* sprob, float* eprob, double* uncert_frame,
                  float* uncert_video, int32_t* observe_point, void* stream) {
  HUAL_REQUIRE(set, "hual_al_score: null set");
  AlScoreArgs a{};
  a.s0 = s0; a.e0 = e0; a.s1 = s1; a.e1 = e1; a.s2 = s2; a.e2 = e2;
  a.ld = set->ld; a.N = set->N; a.vlen = set->vlen; a.tlen = set->tlen;
  a.ap_off = set->ap_off; a.ap_idx = set->ap_idx; a.ap_pos = set->ap_pos;
  a.coff_uncert = coff_uncert;
  a.sprob = sprob; a.eprob = eprob; a.uncert_frame = uncert_frame; a.uncert_video = uncert_video; a.observe = observe_point;
  return launch_al_score(a, (hipStream_t)stream);
}

int hual_al_renew(const hual_al_set* set, const int32_t* sel, int nsel, const float* sprob, const float* eprob,
                  const int32_t* old_idx, const double* coff6, int32_t* new_idx, void* stream) {
  HUAL_REQUIRE(set && coff6, "hual_al_renew: null pointer");
  AlRenewArgs a{};
  a.sel = sel; a.sprob = sprob; a.eprob = eprob; a.ld = set->ld; a.vlen = set->vlen; a.tlen = set->tlen;
  a.ap_off = set->ap_off; a.ap_idx = set->ap_idx; a.ap_pos = set->ap_pos; a.old_idx = old_idx; a.new_idx = new_idx;
  for (int k = 0; k < 6; ++k) a.coff[k] = coff6[k];
  return launch_al_renew(a, sel ? nsel : set->N, (hipStream_t)stream);
}

}  // extern "C"
