// Post-processing of the RON heads on gfx950: softmax + objectness gate + decode + select,
// top-k, class-aware greedy NMS.  One pass over the head tensors (HBM-bound: 2.3 MB/image
// read), then one workgroup per image for the order-dependent part (LDS sort, ballot-free
// bitmask NMS).  Arithmetic that decides an index (IoU, decode, clip) is float32 with one
// rounding per operation (-ffp-contract=off), i.e. what numpy computes for np_methods.py.
//
// Candidate identity: p = anchor_global * (C-1) + (class-1).  p grows exactly in the order
// np.where() enumerates candidates in the reference (layers coarse->fine, anchor-major,
// class-minor; np_methods.py:91-95,117-131), so the 64-bit key
//        key = (float_bits(score) << 32) | (0xFFFFFFFF - p)
// sorted descending reproduces "score descending, position ascending".
#include "common.h"

using ron::kMaxClasses;
namespace {

constexpr int kSelectThreads = 256;
constexpr int kSortCap = 4096;     // keys sorted in LDS by one workgroup
constexpr int kMaxTopK = RON_MAX_TOPK;
constexpr int kTopkThreads = 1024;
constexpr int kMaskWords = kMaxTopK / 64;
constexpr int kCountStride = 32;   // ints: every per-image candidate counter on its own 128-B line (they are atomic targets)
constexpr int kSelectCap = 1024;   // the radix select narrows to at most this many keys before the LDS sort
// Dense images (most of the 425 k (anchor, class) pairs pass): the radix select of one workgroup reads every key once per pass,
// ~70 us per pass over 3.4 MB.  Images with more than kPartMin candidates are first cut into kPartChunks ranges, one workgroup
// each, which keep their own top_k; the image's workgroup then selects among the kPartChunks * top_k survivors (the global top_k
// is a subset of the union of the ranges' top_k: same result, bit for bit).
constexpr int kPartChunks = 8;
constexpr int kPartMin = 16384;
static_assert(kPartChunks * kMaxTopK <= kSortCap, "the survivors of the partial pass must fit the LDS sort");

typedef unsigned long long u64;

struct HeadsDev {
  int num_layers;
  int num_classes;
  int cells[RON_MAX_LAYERS];        // H*W
  int num_anchors[RON_MAX_LAYERS];  // A
  int anchor_base[RON_MAX_LAYERS + 1];   // prefix sum of cells*A
  int block_base[RON_MAX_LAYERS + 1];    // prefix sum of ceil(cells*A / kSelectThreads)
  const float* cls[RON_MAX_LAYERS];
  const float* obj[RON_MAX_LAYERS];
  const float* loc[RON_MAX_LAYERS];
  const float* ay[RON_MAX_LAYERS];
  const float* ax[RON_MAX_LAYERS];
  const float* ah[RON_MAX_LAYERS];
  const float* aw[RON_MAX_LAYERS];
};

struct PostDev {
  float obj_thr, sel_thr, nms_thr;
  int top_k;
  float ref[4];
  float ps[4];
  unsigned flags;
};

struct DetDev {
  int capacity;
  int* classes;
  float* scores;
  float* bboxes;
  int* anchor_index;
  int* count;
};

// score >= 0 (probabilities): the float's bit pattern is already monotone
__device__ __forceinline__ u64 make_key(float score, unsigned p) {
  return ((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - p);
}
// any float (caller-supplied lists): the usual order-preserving map (negative: all bits flipped, else the sign bit set);
// -0 == +0 and NaN goes last, the order np.argsort(-scores) gives (np_methods.py:137-150)
__device__ __forceinline__ u64 make_key_any(float score, unsigned p) {
  unsigned u = __float_as_uint(score + 0.f);
  u = (score != score) ? 0u : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
  return ((u64)u << 32) | (u64)(0xFFFFFFFFu - p);
}

// ------------------------------------------------------------------------------------------
// K-select: one thread per anchor.  The block's class tensor chunk (256 anchors x C floats,
// contiguous) is staged through LDS with coalesced loads; each thread then reads its C values
// at stride C (C odd -> conflict free).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSelectThreads) void select_kernel(HeadsDev hd, PostDev pc, u64* keys,
                                                                int* counts, int cap) {
  extern __shared__ __attribute__((aligned(16))) float stage[];
  const int img = blockIdx.y;
  const int tid = threadIdx.x;
  int layer = 0;
#pragma unroll
  for (int l = 1; l < RON_MAX_LAYERS; ++l)
    if (l < hd.num_layers && (int)blockIdx.x >= hd.block_base[l]) layer = l;
  const int C = hd.num_classes;
  const int n_anchor_layer = hd.cells[layer] * hd.num_anchors[layer];
  const int first = ((int)blockIdx.x - hd.block_base[layer]) * kSelectThreads;
  const int n_here = min(kSelectThreads, n_anchor_layer - first);

  const float* cls = hd.cls[layer] + ((size_t)img * n_anchor_layer + first) * C;
  for (int i = tid; i < n_here * C; i += kSelectThreads) stage[i] = cls[i];
  __syncthreads();

  const bool active = tid < n_here;
  const int local = first + tid;
  bool gate = active;
  if (active && hd.obj[layer] != nullptr) {
    float objp;
    if (pc.flags & RON_IN_OBJ_IS_PROB) {
      objp = hd.obj[layer][(size_t)img * n_anchor_layer + local];
    } else {
      const float2 o = *reinterpret_cast<const float2*>(hd.obj[layer] + ((size_t)img * n_anchor_layer + local) * 2);
      const float m = fmaxf(o.x, o.y);
      const float e0 = expf(o.x - m), e1 = expf(o.y - m);
      objp = e1 / (e0 + e1);
    }
    gate = objp > pc.obj_thr;   // eval_ron_network.py:227-229
  }

  // scores of this anchor.  The row's exponentials are computed ONCE and kept in the row's own LDS slots (the three passes
  // below divide the same values by the same sum: the bits of a score do not depend on which pass computed it).
  float inv_sum = 1.f, mx = 0.f;
  float* row = stage + tid * C;
  const bool is_prob = (pc.flags & RON_IN_CLS_IS_PROB) != 0;
  const bool argmax_mode = pc.sel_thr == 0.f;
  if (gate && !is_prob) {
    mx = row[0];
    float mx1 = -INFINITY;                       // best non-background logit
    for (int c = 1; c < C; ++c) mx1 = fmaxf(mx1, row[c]);
    mx = fmaxf(mx, mx1);
    // softmax_c = exp(x_c - mx) / sum <= exp(x_c - mx) (the sum holds exp(0) = 1): when even the best class stays below the
    // threshold by a margin far above any rounding of the real computation, the anchor selects nothing and its 2 C
    // exponentials and C divisions are skipped (background-dominated anchors: most of a real image)
    if (!argmax_mode && mx1 - mx < logf(pc.sel_thr) - 1e-2f) gate = false;
  }
  if (gate && !is_prob) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      const float e = expf(row[c] - mx);
      row[c] = e;
      s += e;
    }
    inv_sum = s;
  }
  int n_sel = 0;
  // select_threshold 0 / None: the "score > no-label" branch of ssd_bboxes_select_layer (np_methods.py:82-89): ONE candidate per
  // anchor, the arg-max over ALL classes (first maximum, like np.argmax), kept when it is not the background class.  A gated-out
  // anchor is an all-zero row there: arg-max 0, dropped.
  int best_c = 0;
  float best_sc = 0.f;
  if (gate && argmax_mode) {
    best_sc = is_prob ? row[0] : row[0] / inv_sum;
    for (int c = 1; c < C; ++c) {
      const float sc = is_prob ? row[c] : row[c] / inv_sum;
      if (sc > best_sc) { best_sc = sc; best_c = c; }
    }
    n_sel = best_c > 0 ? 1 : 0;
  } else if (gate) {
    for (int c = 1; c < C; ++c) {
      const float sc = is_prob ? row[c] : row[c] / inv_sum;
      n_sel += (sc > pc.sel_thr) ? 1 : 0;
    }
  }
  // wave-aggregated reservation of output slots
  const int lane = tid & 63;
  int incl = n_sel;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const int wave_total = __shfl(incl, 63, 64);
  int base = 0;
  if (lane == 63 && wave_total > 0) base = atomicAdd(&counts[img * kCountStride], wave_total);
  base = __shfl(base, 63, 64);
  if (n_sel > 0) {
    int pos = base + incl - n_sel;
    const unsigned p0 = (unsigned)(hd.anchor_base[layer] + local) * (unsigned)(C - 1);
    u64* out = keys + (size_t)img * cap;
    if (argmax_mode) {
      if (pos < cap) out[pos] = make_key(best_sc, p0 + (unsigned)(best_c - 1));
      return;
    }
    for (int c = 1; c < C; ++c) {
      const float sc = is_prob ? row[c] : row[c] / inv_sum;
      if (sc > pc.sel_thr) {
        if (pos < cap) out[pos] = make_key(sc, p0 + (unsigned)(c - 1));
        ++pos;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Shared pieces of the per-image workgroup.
// ------------------------------------------------------------------------------------------
struct ImageLds {
  u64 sort[kSortCap];              // keys; reused as the NMS suppression bit matrix
  float box[kMaxTopK][4];
  float score[kMaxTopK];
  int cls[kMaxTopK];
  int anchor[kMaxTopK];
  unsigned hist[256];
  int scalars[8];
  int order[kMaxTopK];             // class-grouped position -> sorted row (class-wise NMS)
  int keep[kMaxTopK];              // sorted row -> kept?
  int cstart[kMaxClasses + 2];     // first class-grouped position of every class
  float gbox[kMaxTopK][4];         // boxes in class-grouped order (class-wise NMS: no double indirection in the pair loop)
  int rbeg[kMaxTopK];              // class-grouped position -> first position of its class
  float gvol[kMaxTopK];            // area of gbox[pos], as bboxes_jaccard computes it
  u64 keptw[kTopkThreads / 64][kMaskWords];  // class-wise NMS: the scanning wave's kept rows, one word per 64 grouped positions
};

// gfx950 only: 64 KB is the static LDS limit of every other target, this struct is 65 832 bytes (the CU has 160 KB)
static_assert(sizeof(ImageLds) <= 160 * 1024, "ImageLds exceeds the 160 KB of LDS a gfx950 CU has");

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int j) {
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(v & 0xFFFFFFFFull), j, 64);
  const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), j, 64);
  return ((u64)hi << 32) | lo;
}

__device__ void bitonic_sort_desc(u64* s, int n2, int tid, int nthreads) {
  if (n2 <= nthreads) {
    // one key per thread: exchanges at distance < 64 are wave shuffles (45 of the 55 stages of a 1024-key sort), only the
    // others go through LDS and a barrier
    u64 key = tid < n2 ? s[tid] : 0;
    for (int k = 2; k <= n2; k <<= 1) {
      const bool desc = (tid & k) == 0;
      for (int j = k >> 1; j > 0; j >>= 1) {
        u64 other;
        if (j >= 64) {
          __syncthreads();                       // the previous stage's reads of s are done
          if (tid < n2) s[tid] = key;
          __syncthreads();
          other = tid < n2 ? s[tid ^ j] : 0;
        } else {
          other = shfl_xor_u64(key, j);
        }
        const bool take_max = ((tid & j) == 0) == desc;
        const u64 hi = key > other ? key : other, lo = key > other ? other : key;
        key = take_max ? hi : lo;
      }
    }
    __syncthreads();
    if (tid < n2) s[tid] = key;
    __syncthreads();
    return;
  }
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const u64 a = s[i], b = s[ixj];
          const bool desc_block = (i & k) == 0;
          if (desc_block ? (a < b) : (a > b)) {
            s[i] = b;
            s[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

// Puts the top_k largest keys of keys[0..m) (unique keys) sorted descending at lds.sort[0..).
// Returns number of valid keys (min(m, top_k)).
// With `below` != ~0: only keys < below take part (m_live of them; the caller knows how many it has consumed).
__device__ int topk_keys(const u64* __restrict__ keys, int m, int top_k, ImageLds& lds, u64 below = ~0ull, int m_live = -1) {
  const int tid = threadIdx.x;
  const int nth = blockDim.x;
  const bool bounded = below != ~0ull;
  if (m_live < 0) m_live = m;
  int n_sel;
  if (m_live <= kSelectCap) {
    if (!bounded) {
      for (int i = tid; i < m; i += nth) lds.sort[i] = keys[i];
      n_sel = m;
      __syncthreads();
    } else {
      if (tid == 0) lds.scalars[3] = 0;
      __syncthreads();
      for (int i = tid; i < m; i += nth) {
        const u64 k = keys[i];
        if (k < below) {
          const int pos = atomicAdd(&lds.scalars[3], 1);
          if (pos < kSortCap) lds.sort[pos] = k;
        }
      }
      __syncthreads();
      n_sel = min(lds.scalars[3], kSortCap);
      __syncthreads();
    }
  } else {
    // radix select from the most significant byte down until few enough keys survive for a short LDS sort
    u64 prefix = 0, mask = 0;
    int need = top_k, above_total = 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
      for (int i = tid; i < 256; i += nth) lds.hist[i] = 0;
      __syncthreads();
      for (int i = tid; i < m; i += nth) {
        const u64 k = keys[i];
        if ((k & mask) == prefix && k < below) atomicAdd(&lds.hist[(unsigned)(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid < 64) {
        // wave 0: lane l owns bins 4l..4l+3; suffix sums over lanes by shuffles, then the crossing bin inside one lane
        const int l = tid;
        const int c0 = (int)lds.hist[4 * l], c1 = (int)lds.hist[4 * l + 1], c2 = (int)lds.hist[4 * l + 2], c3 = (int)lds.hist[4 * l + 3];
        const int mine = c0 + c1 + c2 + c3;
        int suf = mine;                                   // inclusive suffix: bins >= 4l
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int v = __shfl_down(suf, d, 64);
          if (l + d < 64) suf += v;
        }
        const int above = suf - mine;                     // keys in bins > 4l+3
        if (above < need && suf >= need) {                // the crossing bin is one of this lane's four (exactly one lane)
          int cum = above, bin = 4 * l + 3;
          if (cum + c3 >= need) { bin = 4 * l + 3; }
          else { cum += c3; if (cum + c2 >= need) { bin = 4 * l + 2; } else { cum += c2; if (cum + c1 >= need) { bin = 4 * l + 1; } else { cum += c1; bin = 4 * l; } } }
          lds.scalars[0] = bin;
          lds.scalars[1] = cum;
          lds.scalars[2] = (int)lds.hist[bin];
        }
      }
      __syncthreads();
      const int bin = lds.scalars[0], cum = lds.scalars[1], bin_count = lds.scalars[2];
      __syncthreads();
      above_total += cum;
      need -= cum;
      prefix |= (u64)bin << shift;
      mask |= (u64)0xFF << shift;
      if (above_total + bin_count <= kSelectCap) break;
    }
    if (tid == 0) lds.scalars[3] = 0;
    __syncthreads();
    for (int i = tid; i < m; i += nth) {
      const u64 k = keys[i];
      if ((k & mask) >= prefix && k < below) {
        const int pos = atomicAdd(&lds.scalars[3], 1);
        if (pos < kSortCap) lds.sort[pos] = k;
      }
    }
    __syncthreads();
    n_sel = min(lds.scalars[3], kSortCap);
  }
  int n2 = 64;
  while (n2 < n_sel) n2 <<= 1;
  for (int i = n_sel + tid; i < n2; i += nth) lds.sort[i] = 0;
  __syncthreads();
  bitonic_sort_desc(lds.sort, n2, tid, nth);
  return min(n_sel, top_k);
}

// np_methods.py:186-205 (bboxes_jaccard) + :229-242: does box i remove box j?  float32, one rounding per operation:
//     iou = inter / (vol1 + vol2 - inter);  suppressed = !(iou < thr)      (NaN suppresses, as logical_or(overlap < thr, ...) does)
// with the two areas given (they are per-row values).  The pair loop of the class-wise scan is bound by the vector ALU of its one
// CU: the quotient is first estimated with v_rcp_f32 (1 ulp) and only a pair whose estimate lies within 1e-6 of the threshold - or
// whose denominator is not a normal positive number - takes the IEEE division; every other pair is decided by the estimate, whose
// error (< 3e-7 relative) cannot move it across the threshold.
__device__ __forceinline__ bool nms_suppresses_vol(const float* bi, float vol1, const float* bj, float vol2, float thr) {
  const float ih = fmaxf(fminf(bi[2], bj[2]) - fmaxf(bi[0], bj[0]), 0.f);
  const float iw = fmaxf(fminf(bi[3], bj[3]) - fmaxf(bi[1], bj[1]), 0.f);
  const float inter = ih * iw;
  const float den = vol1 + vol2 - inter;
  const float est = inter * __builtin_amdgcn_rcpf(den);
  const bool sure_below = est < thr * (1.f - 1e-6f), sure_above = est > thr * (1.f + 1e-6f);
  if ((sure_below || sure_above) && den > 1e-30f && den < 1e30f) return sure_above;
  const float iou = inter / den;
  return !(iou < thr);
}

// tf_extended/bboxes.py:195-211 + :226: kept box i against a later box j; mode 1 = 'min', 2 = 'union'
__device__ __forceinline__ bool tfe_suppresses(const float* bi, const float* bj, float thr, int mode) {
  const float ih = fmaxf(fminf(bj[2], bi[2]) - fmaxf(bj[0], bi[0]), 0.f);
  const float iw = fmaxf(fminf(bj[3], bi[3]) - fmaxf(bj[1], bi[1]), 0.f);
  const float inner = ih * iw;
  const float this_vol = (bi[2] - bi[0]) * (bi[3] - bi[1]);
  const float vol = (bj[3] - bj[1]) * (bj[2] - bj[0]);
  const float den = mode == 2 ? (vol - inner + this_vol) : fminf(vol, this_vol);
  const float sc = den > 0.f ? inner / den : 0.f;      // safe_divide, bboxes.py:192-193
  return !(sc < thr);
}

// The pair test of the three NMS flavours with the two areas given: `be` / `ve` the EARLIER row of the score order (the box that
// would do the suppressing), `bl` / `vl` the later one.  mode 0 = np_methods IoU (plain division: 0 / 0 = NaN suppresses), 1 = 'min',
// 2 = 'union' of tf_extended/bboxes.py:195-211 (safe_divide: 0 when the denominator is not positive).  The operand order of the
// reference's sums is kept ('union': (vol_later - inner) + vol_earlier).  The quotient is estimated with v_rcp_f32 first, as in
// nms_suppresses_vol: only estimates within 1e-6 of the threshold or unusual denominators take the IEEE division.
__device__ __forceinline__ bool pair_suppresses(const float* be, float ve, const float* bl, float vl, float thr, int mode) {
  const float ih = fmaxf(fminf(bl[2], be[2]) - fmaxf(bl[0], be[0]), 0.f);
  const float iw = fmaxf(fminf(bl[3], be[3]) - fmaxf(bl[1], be[1]), 0.f);
  const float inner = ih * iw;
  const float den = mode == 0 ? (ve + vl - inner) : mode == 2 ? (vl - inner + ve) : fminf(vl, ve);
  const float est = inner * __builtin_amdgcn_rcpf(den);
  const bool sure_below = est < thr * (1.f - 1e-6f), sure_above = est > thr * (1.f + 1e-6f);
  if ((sure_below || sure_above) && den > 1e-30f && den < 1e30f) return sure_above;
  const float q = (mode == 0 || den > 0.f) ? inner / den : 0.f;
  return !(q < thr);
}

// Greedy NMS of the n sorted boxes in lds, any flavour, all rows one segment (the TF variant: one workgroup per class; np_methods
// with class ids beyond the class-wise scan's kMaxClasses - not reachable through the entry points: mode 0 with the label test).  Same scheme as nms_scan_classwise below: the bit
// matrix by COLUMNS (row b: the earlier rows that overlap it), a quarter wave per row, then per 64-row block the fixed point of
// kept = alive & ((column & kept) == 0) on wave 0.  Stopping after max_keep kept rows = keeping the first max_keep of them.
// Leaves in lds.hist: [0..15] low / [16..31] high halves of the keep bits per 64-row word, [32..47] exclusive kept counts per
// word; lds.scalars[4] = number kept (<= max_keep).
__device__ void nms_scan(ImageLds& lds, int n, float nms_thr, int mode, int max_keep) {
  const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nth >> 6;
  const int words = (n + 63) >> 6;
  u64* mask = lds.sort;    // [n][kMaskWords]
  for (int i = tid; i < n; i += nth) lds.gvol[i] = (lds.box[i][2] - lds.box[i][0]) * (lds.box[i][3] - lds.box[i][1]);
  __syncthreads();
  {
    typedef unsigned short __attribute__((may_alias)) u16a;
    u16a* mask16 = reinterpret_cast<u16a*>(mask);
    const int sub = lane >> 4, sl = lane & 15;
    for (int r0 = wave * 4; r0 < n; r0 += nwaves * 4) {
      const int b = r0 + sub;
      const bool live = b < n;
      float bb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bb[q] = live ? lds.box[b][q] : 0.f;
      const float vb = live ? lds.gvol[b] : 0.f;
      const int cb = live ? lds.cls[b] : -1;
      if (live && sl <= (b >> 6)) mask[b * kMaskWords + sl] = 0;
      const int iters_me = (live && b > 0) ? ((b - 1) >> 4) + 1 : 0;   // 16-row slots [0, (b - 1) >> 4] hold the rows in [0, b)
      int iters = max(iters_me, __shfl_xor(iters_me, 16, 64));
      iters = max(iters, __shfl_xor(iters, 32, 64));
      // (no unroll pragma: hipcc refuses to unroll a loop with a run-time trip count around a ballot and says so with a warning)
      for (int it = 0; it < iters; ++it) {
        const int a = (it << 4) + sl;
        const int ac = min(a, kMaxTopK - 1);
        const bool sup = pair_suppresses(lds.box[ac], lds.gvol[ac], bb, vb, nms_thr, mode) && (mode != 0 || lds.cls[ac] == cb) &&
                         it < iters_me && a < b;
        const u64 bal = __ballot(sup);
        const unsigned half = (sub & 2) ? (unsigned)(bal >> 32) : (unsigned)(bal & 0xFFFFFFFFull);
        if (sl == 0 && it < iters_me) mask16[(b * kMaskWords << 2) + it] = (unsigned short)(half >> ((sub & 1) << 4));
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    u64* keptw = lds.keptw[0];
    for (int wb = 0; wb < words; ++wb) {
      const int r = (wb << 6) + lane;
      const bool in = r < n;
      u64 hit = 0;
      for (int w = 0; w < wb; ++w) hit |= (in ? mask[r * kMaskWords + w] : 0ull) & keptw[w];
      const u64 col = in ? mask[r * kMaskWords + wb] : 0ull;
      const bool alive = in && hit == 0;
      u64 kept = __ballot(alive);
      for (;;) {
        const u64 next = __ballot(alive && (col & kept) == 0);
        if (next == kept) break;
        kept = next;
      }
      if (lane == 0) keptw[wb] = kept;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // lane w: kept rows of word w, cut after the max_keep-th kept row of the score order
    u64 kb = lane < words ? keptw[lane] : 0ull;
    int cnt = __popcll(kb);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    const int excl = incl - cnt;
    if (excl >= max_keep) kb = 0;
    else if (incl > max_keep) {
      u64 rest = kb;
      for (int k = 0; k < max_keep - excl; ++k) rest &= rest - 1;      // the set bits beyond the first max_keep - excl
      kb &= ~rest;
    }
    if (lane < kMaskWords) {
      lds.hist[lane] = (unsigned)(kb & 0xFFFFFFFFull);
      lds.hist[16 + lane] = (unsigned)(kb >> 32);
      lds.hist[32 + lane] = (unsigned)min(excl, max_keep);
    }
    if (lane == 63) lds.scalars[4] = min(incl, max_keep);
  }
  __syncthreads();
}

// Class-wise form of the np_methods NMS (np_methods.py:229-242 suppresses only boxes of the SAME class): rows are
// grouped by class (stable, so score order is kept inside a class) and every class is scanned by its own wave in parallel.
// Pairs examined: sum_c n_c^2 / 2 instead of n^2 / 2.
//   The bit matrix is held by COLUMNS: row b of `mask` says which EARLIER rows of its class overlap b beyond the threshold (the
// test is symmetric in the two boxes, bit for bit: min / max / + commute).  A row is kept iff no kept earlier row is in its
// column.  64 rows at a time, lane = row: the kept rows of the earlier blocks are final (wave-uniform words), so "removed by an
// earlier block" is an AND per lane; inside the block kept = alive & ((column & kept) == 0) is iterated from kept = alive until
// it stops changing - row i is final after i + 1 rounds whatever the start, so the fixed point is the greedy answer, and the
// rounds needed are the longest suppression chain in the block (a handful), not the 64 serial steps of the row-major scan.
// Same outputs in lds.hist / lds.scalars[4] as nms_scan.
__device__ void nms_scan_classwise(ImageLds& lds, int n, float nms_thr, int num_classes) {
  const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nth >> 6;
  const int words = (n + 63) >> 6;
  u64* mask = lds.sort;    // [n][kMaskWords], indexed by class-grouped position
  // class-grouped order, stable inside a class (n <= kMaxTopK <= blockDim: one row per thread).  Rank of a row among the
  // rows of its class = same-class rows in earlier waves + same-class lanes below it in its own wave (ballots).
  int* wcnt = reinterpret_cast<int*>(lds.sort);      // [kMaskWords waves][kMaxClasses]; the mask is written after this phase
  for (int i = tid; i < kMaskWords * kMaxClasses; i += nth) wcnt[i] = 0;
  __syncthreads();
  const int c_me = tid < n ? (lds.cls[tid] & (kMaxClasses - 1)) : -1;
  int rank_me = 0;
  if ((wave << 6) < n) {
    u64 todo = __ballot(c_me >= 0);
    while (todo != 0) {
      const int leader = __ffsll((long long)todo) - 1;
      const int lc = __shfl(c_me, leader, 64);
      const u64 same = __ballot(c_me == lc);
      if (c_me == lc) rank_me = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == leader) wcnt[wave * kMaxClasses + lc] = __popcll(same);
      todo &= ~same;
    }
  }
  __syncthreads();
  if (tid < kMaxClasses) {   // per class: exclusive prefix over the waves; the class totals go through lds.rbeg (written for real below)
    int run = 0;
    for (int w = 0; w < words; ++w) {
      const int t = wcnt[w * kMaxClasses + tid];
      wcnt[w * kMaxClasses + tid] = run;
      run += t;
    }
    lds.rbeg[tid] = run;
  }
  __syncthreads();
  if (tid <= kMaxClasses) {  // exclusive scan of the class totals (classes >= num_classes are empty)
    int before = 0;
    for (int c = 0; c < tid && c < num_classes; ++c) before += lds.rbeg[c];
    lds.cstart[tid] = before;
  }
  __syncthreads();
  if (tid < n) {
    const int pos = lds.cstart[c_me] + wcnt[wave * kMaxClasses + c_me] + rank_me;
    lds.order[pos] = tid;
    lds.rbeg[pos] = lds.cstart[c_me];
    lds.keep[tid] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) lds.gbox[pos][q] = lds.box[tid][q];     // boxes in grouped order
    lds.gvol[pos] = (lds.box[tid][2] - lds.box[tid][0]) * (lds.box[tid][3] - lds.box[tid][1]);
  }
  __syncthreads();                             // wcnt (in lds.sort) is dead from here: the mask goes there
  // Column of grouped row b: the earlier rows of its class that overlap it.  A quarter wave (16 lanes) per row - classes are
  // short (top_k rows over the classes), a whole wave per row left most lanes without a pair - taking the earlier rows 16 at a
  // time, aligned, so that its 16 ballot bits are one 16-bit slot of the row's words.  Words [s0 >> 6, b >> 6] of row b are
  // zeroed first; the scan reads no others.  (One CU's vector ALU bounds this loop: ~36 issue slots per 64 pairs.)
  {
    typedef unsigned short __attribute__((may_alias)) u16a;
    u16a* mask16 = reinterpret_cast<u16a*>(mask);
    const int sub = lane >> 4, sl = lane & 15;
    for (int r0 = wave * 4; r0 < n; r0 += nwaves * 4) {
      const int b = r0 + sub;
      const bool live = b < n;
      const int s0 = live ? lds.rbeg[b] : 0;
      float bb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bb[q] = live ? lds.gbox[b][q] : 0.f;
      const float vb = live ? lds.gvol[b] : 0.f;
      if (live && sl <= (b >> 6) - (s0 >> 6)) mask[b * kMaskWords + (s0 >> 6) + sl] = 0;
      const int k0 = s0 >> 4;                                 // 16-row slots [k0, (b - 1) >> 4] hold the rows in [s0, b)
      const int iters_me = (live && b > s0) ? ((b - 1) >> 4) - k0 + 1 : 0;
      int iters = max(iters_me, __shfl_xor(iters_me, 16, 64));
      iters = max(iters, __shfl_xor(iters, 32, 64));
      // (loads and arithmetic unconditional on a clamped row, validity applied to the result: straight-line code)
      // (no unroll pragma: hipcc refuses to unroll a loop with a run-time trip count around a ballot and says so with a warning)
      for (int it = 0; it < iters; ++it) {
        const int a = ((k0 + it) << 4) + sl;
        const int ac = min(a, kMaxTopK - 1);
        const bool sup = nms_suppresses_vol(lds.gbox[ac], lds.gvol[ac], bb, vb, nms_thr) && it < iters_me && a >= s0 && a < b;
        const u64 bal = __ballot(sup);
        const unsigned half = (sub & 2) ? (unsigned)(bal >> 32) : (unsigned)(bal & 0xFFFFFFFFull);
        if (sl == 0 && it < iters_me) mask16[(b * kMaskWords << 2) + k0 + it] = (unsigned short)(half >> ((sub & 1) << 4));
      }
    }
  }
  __syncthreads();
  for (int c = wave; c < kMaxClasses && c < num_classes; c += nwaves) {   // one wave per class
    // (read through readfirstlane: the class bounds are wave-uniform)
    const int s0 = __builtin_amdgcn_readfirstlane(lds.cstart[c]), s1 = __builtin_amdgcn_readfirstlane(lds.cstart[c + 1]);
    if (s0 >= s1) continue;
    const int wfirst = s0 >> 6, wlast = (s1 - 1) >> 6;
    u64* keptw = lds.keptw[wave];              // kept rows of the blocks done so far, private to this wave
    for (int wb = wfirst; wb <= wlast; ++wb) {
      const int r = (wb << 6) + lane;
      const bool in_cls = r >= s0 && r < s1;
      u64 hit = 0;                             // kept rows of the earlier blocks in this row's column
      for (int w = wfirst; w < wb; ++w) hit |= (in_cls ? mask[r * kMaskWords + w] : 0ull) & keptw[w];
      const u64 col = in_cls ? mask[r * kMaskWords + wb] : 0ull;
      const bool alive = in_cls && hit == 0;
      u64 kept = __ballot(alive);
      for (;;) {
        const u64 next = __ballot(alive && (col & kept) == 0);
        if (next == kept) break;
        kept = next;
      }
      if ((kept >> lane) & 1ull) lds.keep[lds.order[r]] = 1;
      if (lane == 0) keptw[wb] = kept;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (tid < 64) {            // keep bits per 64-row word of the score order + exclusive counts
    u64 kb = 0;
    for (int w = 0; w < words; ++w) {
      const int i = (w << 6) + lane;
      const u64 bits = __ballot(i < n && lds.keep[i] != 0);
      if (lane == w) kb = bits;
    }
    int cnt = (lane < words) ? __popcll(kb) : 0;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    if (lane < kMaskWords) {
      lds.hist[lane] = (unsigned)(kb & 0xFFFFFFFFull);
      lds.hist[16 + lane] = (unsigned)(kb >> 32);
      lds.hist[32 + lane] = (unsigned)(incl - cnt);
    }
    if (lane == 63) lds.scalars[4] = incl;
  }
  __syncthreads();
}

// output slot of sorted row i after the scan, or -1 when it was suppressed
__device__ __forceinline__ int nms_slot(const ImageLds& lds, int i) {
  const int w = i >> 6, b = i & 63;
  const u64 kb = ((u64)lds.hist[16 + w] << 32) | (u64)lds.hist[w];
  if (!((kb >> b) & 1ull)) return -1;
  return (int)lds.hist[32 + w] + __popcll(kb & ((1ull << b) - 1ull));
}

// Greedy scan over the n sorted boxes in lds (np_methods.py:229-242).  Writes kept rows,
// compacted, to `out` (image `img`), after the resize by `ref` when do_resize.
__device__ void nms_and_store(ImageLds& lds, int n, float nms_thr, const float* ref, bool do_resize,
                              const DetDev& out, int img, int num_classes) {
  const int tid = threadIdx.x;
  const int nth = blockDim.x;
  if (num_classes > 0 && num_classes <= kMaxClasses) nms_scan_classwise(lds, n, nms_thr, num_classes);
  else nms_scan(lds, n, nms_thr, 0, n);       // class ids outside [0, kMaxClasses): generic all-pairs form
  const int total = lds.scalars[4];
  const float sy = ref[2] - ref[0], sx = ref[3] - ref[1];
  for (int i = tid; i < out.capacity; i += nth) {
    // zero rows at and after `total` (records are fixed-capacity, zero padded)
    if (i >= total) {
      const size_t o = (size_t)img * out.capacity + i;
      out.classes[o] = 0;
      out.scores[o] = 0.f;
      out.anchor_index[o] = 0;
      out.bboxes[o * 4 + 0] = 0.f; out.bboxes[o * 4 + 1] = 0.f;
      out.bboxes[o * 4 + 2] = 0.f; out.bboxes[o * 4 + 3] = 0.f;
    }
  }
  for (int i = tid; i < n; i += nth) {
    const int pos = nms_slot(lds, i);
    if (pos >= 0 && pos < out.capacity) {
      const size_t o = (size_t)img * out.capacity + pos;
      out.classes[o] = lds.cls[i];
      out.scores[o] = lds.score[i];
      out.anchor_index[o] = lds.anchor[i];
      float b0 = lds.box[i][0], b1 = lds.box[i][1], b2 = lds.box[i][2], b3 = lds.box[i][3];
      if (do_resize) {   // np_methods.py:167-183
        b0 = (b0 - ref[0]) / sy; b1 = (b1 - ref[1]) / sx;
        b2 = (b2 - ref[0]) / sy; b3 = (b3 - ref[1]) / sx;
      }
      out.bboxes[o * 4 + 0] = b0; out.bboxes[o * 4 + 1] = b1;
      out.bboxes[o * 4 + 2] = b2; out.bboxes[o * 4 + 3] = b3;
    }
  }
  if (tid == 0) out.count[img] = min(total, out.capacity);
}

__device__ void store_sorted(const ImageLds& lds, int n, const DetDev& out, int img) {
  for (int i = threadIdx.x; i < out.capacity; i += blockDim.x) {
    const size_t o = (size_t)img * out.capacity + i;
    const bool v = i < n;
    out.classes[o] = v ? lds.cls[i] : 0;
    out.scores[o] = v ? lds.score[i] : 0.f;
    out.anchor_index[o] = v ? lds.anchor[i] : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) out.bboxes[o * 4 + k] = v ? lds.box[i][k] : 0.f;
  }
  if (threadIdx.x == 0) out.count[img] = min(n, out.capacity);
}

__device__ __forceinline__ void decode_box(const float* l, float ya, float xa, float ha, float wa,
                                           const float* ps, float* box) {
  // np_methods.py:41-50 (== ssd_common.py:464-472)
  const float cx = l[0] * wa * ps[0] + xa;
  const float cy = l[1] * ha * ps[1] + ya;
  const float w = wa * expf(l[2] * ps[2]);
  const float h = ha * expf(l[3] * ps[3]);
  box[0] = cy - h / 2.f;
  box[1] = cx - w / 2.f;
  box[2] = cy + h / 2.f;
  box[3] = cx + w / 2.f;
}

// ------------------------------------------------------------------------------------------
// K-topk-partial: blockIdx.x = range of the image's key list, blockIdx.y = image.  Only for images with > kPartMin keys.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTopkThreads) void topk_partial_kernel(const u64* keys, const int* counts, int cap, int top_k,
                                                                    u64* part_keys, int* part_total) {
  __shared__ ImageLds lds;
  const int img = blockIdx.y, c = blockIdx.x;
  const int tid = threadIdx.x;
  const int m = min(counts[img * kCountStride], cap);
  if (m <= kPartMin) return;
  const int beg = (int)((long long)m * c / kPartChunks), end = (int)((long long)m * (c + 1) / kPartChunks);
  const int n = topk_keys(keys + (size_t)img * cap + beg, end - beg, top_k, lds);
  if (tid == 0) lds.scalars[5] = atomicAdd(&part_total[img * kCountStride], n);
  __syncthreads();
  u64* dst = part_keys + (size_t)img * (kPartChunks * kMaxTopK) + lds.scalars[5];
  for (int r = tid; r < n; r += blockDim.x) dst[r] = lds.sort[r];
}

// ------------------------------------------------------------------------------------------
// K-topk-nms: one workgroup per image.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTopkThreads) void topk_nms_kernel(HeadsDev hd, PostDev pc, const u64* keys,
                                                                const int* counts, int cap, const u64* part_keys,
                                                                const int* part_total, DetDev out,
                                                                DetDev sorted_out, int* n_candidates) {
  __shared__ ImageLds lds;
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  const int m_raw = counts[img * kCountStride];
  if (tid == 0 && n_candidates != nullptr) n_candidates[img] = m_raw;
  int m = min(m_raw, cap);
  const u64* my_keys = keys + (size_t)img * cap;
  if (m > kPartMin) {                 // the partial pass ran for this image: select among its survivors
    my_keys = part_keys + (size_t)img * (kPartChunks * kMaxTopK);
    m = part_total[img * kCountStride];
  }
  const int n = topk_keys(my_keys, m, pc.top_k, lds);
  // keys -> records.  Each thread keeps its key in a register before the LDS region is reused.
  const int C1 = hd.num_classes - 1;
  for (int r = tid; r < n; r += blockDim.x) {
    const u64 k = lds.sort[r];
    const float score = __uint_as_float((unsigned)(k >> 32));
    const unsigned p = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
    const int anchor = (int)(p / (unsigned)C1);
    const int cls = (int)(p - (unsigned)anchor * (unsigned)C1) + 1;
    int layer = 0;
#pragma unroll
    for (int l = 1; l < RON_MAX_LAYERS; ++l)
      if (l < hd.num_layers && anchor >= hd.anchor_base[l]) layer = l;
    const int local = anchor - hd.anchor_base[layer];
    const int A = hd.num_anchors[layer];
    const int cell = local / A, a = local - cell * A;
    const size_t n_anchor_layer = (size_t)hd.cells[layer] * A;
    const float* l4 = hd.loc[layer] + ((size_t)img * n_anchor_layer + local) * 4;
    float box[4];
    if (pc.flags & RON_IN_LOC_DECODED) {
      box[0] = l4[0]; box[1] = l4[1]; box[2] = l4[2]; box[3] = l4[3];
    } else {
      decode_box(l4, hd.ay[layer][cell], hd.ax[layer][cell], hd.ah[layer][a], hd.aw[layer][a], pc.ps, box);
    }
    // np_methods.py:153-164 (clip: no ymin<=ymax repair)
    lds.box[r][0] = fmaxf(box[0], pc.ref[0]);
    lds.box[r][1] = fmaxf(box[1], pc.ref[1]);
    lds.box[r][2] = fminf(box[2], pc.ref[2]);
    lds.box[r][3] = fminf(box[3], pc.ref[3]);
    lds.score[r] = score;
    lds.cls[r] = cls;
    lds.anchor[r] = anchor;
  }
  __syncthreads();
  // self-cleaning workspace (ron_detect's own): this workgroup is the last reader of its image's counters - every thread took them
  // at the top and has passed barriers since - so it leaves them zero for the next call and the 5 us memset in front of every call goes
  if ((pc.flags & ron::kPostWsClean) && tid == 0) {
    const_cast<int*>(counts)[img * kCountStride] = 0;
    const_cast<int*>(part_total)[img * kCountStride] = 0;
  }
  if (sorted_out.classes != nullptr) store_sorted(lds, n, sorted_out, img);
  nms_and_store(lds, n, pc.nms_thr, pc.ref, true, out, img, hd.num_classes);
}

// Explicit lists (np_methods.bboxes_sort -> bboxes_nms), one workgroup per image.
__global__ __launch_bounds__(kTopkThreads) void list_sort_nms_kernel(const int* classes, const float* scores,
                                                                     const float* bboxes, const int* n_valid,
                                                                     int n_in, int top_k, float nms_thr,
                                                                     u64* keys, DetDev out, DetDev sorted_out) {
  __shared__ ImageLds lds;
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  const int m = n_valid ? min(n_valid[img], n_in) : n_in;
  u64* my_keys = keys + (size_t)img * n_in;
  for (int i = tid; i < m; i += blockDim.x) my_keys[i] = make_key_any(scores[(size_t)img * n_in + i], (unsigned)i);
  __threadfence_block();
  __syncthreads();
  const int n = topk_keys(my_keys, m, top_k, lds);
  for (int r = tid; r < n; r += blockDim.x) {
    const u64 k = lds.sort[r];
    const unsigned p = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
    const size_t src = (size_t)img * n_in + p;
    lds.score[r] = scores[src];
    lds.cls[r] = classes[src];
    lds.anchor[r] = (int)p;
#pragma unroll
    for (int q = 0; q < 4; ++q) lds.box[r][q] = bboxes[src * 4 + q];
  }
  __syncthreads();
  if (sorted_out.classes != nullptr) store_sorted(lds, n, sorted_out, img);
  const float ref[4] = {0.f, 0.f, 1.f, 1.f};
  nms_and_store(lds, n, nms_thr, ref, false, out, img, 0);     // arbitrary class ids in explicit lists
}

__global__ void decode_layer_kernel(const float* loc, int n, int cells, int A, const float* ay, const float* ax,
                                    const float* ah, const float* aw, float ps0, float ps1, float ps2, float ps3,
                                    float* out) {
  const size_t total = (size_t)n * cells * A;
  const float ps[4] = {ps0, ps1, ps2, ps3};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const int cell = (int)((i / A) % cells);
    const float4 l = reinterpret_cast<const float4*>(loc)[i];
    const float l4[4] = {l.x, l.y, l.z, l.w};
    float box[4];
    decode_box(l4, ay[cell], ax[cell], ah[a], aw[a], ps, box);
    reinterpret_cast<float4*>(out)[i] = make_float4(box[0], box[1], box[2], box[3]);
  }
}

__global__ void softmax_last_kernel(const float* x, long long rows, int c, int pick, float* y) {
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows;
       r += (long long)gridDim.x * blockDim.x) {
    const float* in = x + r * c;
    float m = in[0];
    for (int i = 1; i < c; ++i) m = fmaxf(m, in[i]);
    float s = 0.f;
    for (int i = 0; i < c; ++i) s += expf(in[i] - m);
    if (pick >= 0) {
      y[r] = expf(in[pick] - m) / s;
    } else {
      for (int i = 0; i < c; ++i) y[r * c + i] = expf(in[i] - m) / s;
    }
  }
}

int build_heads_dev(const ron_heads* h, HeadsDev* d, bool need_anchors) {
  RON_REQUIRE(h != nullptr, "heads is NULL");
  RON_REQUIRE(h->num_layers >= 1 && h->num_layers <= RON_MAX_LAYERS, "num_layers %d out of range", h->num_layers);
  RON_REQUIRE(h->num_classes >= 2 && h->num_classes <= RON_MAX_CLASSES, "num_classes %d out of range [2, %d]", h->num_classes, RON_MAX_CLASSES);
  d->num_layers = h->num_layers;
  d->num_classes = h->num_classes;
  d->anchor_base[0] = 0;
  d->block_base[0] = 0;
  for (int l = 0; l < RON_MAX_LAYERS; ++l) {
    if (l < h->num_layers) {
      RON_REQUIRE(h->feat_h[l] > 0 && h->feat_w[l] > 0 && h->num_anchors[l] > 0 &&
                      h->num_anchors[l] <= RON_MAX_ANCHORS_PER_CELL,
                  "layer %d: bad shape", l);
      RON_REQUIRE(h->cls[l] != nullptr && h->loc[l] != nullptr, "layer %d: cls/loc pointer is NULL", l);
      if (need_anchors)
        RON_REQUIRE(h->anchor_y[l] && h->anchor_x[l] && h->anchor_h[l] && h->anchor_w[l],
                    "layer %d: anchors are required when loc holds raw offsets", l);
      d->cells[l] = h->feat_h[l] * h->feat_w[l];
      d->num_anchors[l] = h->num_anchors[l];
      const int na = d->cells[l] * d->num_anchors[l];
      d->anchor_base[l + 1] = d->anchor_base[l] + na;
      d->block_base[l + 1] = d->block_base[l] + (na + kSelectThreads - 1) / kSelectThreads;
      d->cls[l] = h->cls[l]; d->obj[l] = h->obj[l]; d->loc[l] = h->loc[l];
      d->ay[l] = h->anchor_y[l]; d->ax[l] = h->anchor_x[l]; d->ah[l] = h->anchor_h[l]; d->aw[l] = h->anchor_w[l];
    } else {
      d->cells[l] = 0; d->num_anchors[l] = 0;
      d->anchor_base[l + 1] = d->anchor_base[l];
      d->block_base[l + 1] = d->block_base[l];
      d->cls[l] = d->obj[l] = d->loc[l] = d->ay[l] = d->ax[l] = d->ah[l] = d->aw[l] = nullptr;
    }
  }
  return RON_OK;
}

int to_det_dev(const ron_detections* d, DetDev* out, int top_k, const char* what, bool optional) {
  if (d == nullptr || d->classes == nullptr) {
    if (!optional) { ron::set_error("%s is NULL", what); return RON_ERR_INVALID; }
    *out = DetDev{0, nullptr, nullptr, nullptr, nullptr, nullptr};
    return RON_OK;
  }
  RON_REQUIRE(d->capacity >= top_k, "%s: capacity %d < top_k %d", what, d->capacity, top_k);
  RON_REQUIRE(d->scores && d->bboxes && d->anchor_index && d->count, "%s: NULL member", what);
  *out = DetDev{d->capacity, d->classes, d->scores, d->bboxes, d->anchor_index, d->count};
  return RON_OK;
}

}  // namespace

extern "C" int64_t ron_post_np_workspace_bytes(const ron_heads* heads, int n) {
  HeadsDev hd;
  if (build_heads_dev(heads, &hd, false) != RON_OK || n <= 0) return -1;
  const int64_t cap = (int64_t)hd.anchor_base[RON_MAX_LAYERS] * (heads->num_classes - 1);
  // [candidate counts][survivor counts of the partial pass][keys][survivors]
  return 2 * ron::align_up((int64_t)n * kCountStride * 4, 256) + (int64_t)n * cap * 8 + (int64_t)n * kPartChunks * kMaxTopK * 8;
}

extern "C" int ron_post_np(const ron_heads* heads, int n, const ron_post_cfg* cfg, void* workspace,
                           int64_t workspace_bytes, ron_detections* out, ron_detections* sorted_out,
                           int32_t* n_candidates, void* stream) {
  RON_REQUIRE(cfg != nullptr && n > 0, "bad cfg / n");
  RON_REQUIRE(cfg->top_k >= 1 && cfg->top_k <= kMaxTopK, "top_k %d not in [1, %d]", cfg->top_k, kMaxTopK);
  RON_REQUIRE(cfg->select_threshold >= 0.f, "select_threshold must be >= 0 (0 = the arg-max branch of np_methods.py:82-89)");
  HeadsDev hd;
  int rc = build_heads_dev(heads, &hd, (cfg->input_flags & RON_IN_LOC_DECODED) == 0);
  if (rc != RON_OK) return rc;
  const int64_t need = ron_post_np_workspace_bytes(heads, n);
  RON_REQUIRE(workspace != nullptr && workspace_bytes >= need, "workspace too small: %lld < %lld",
              (long long)workspace_bytes, (long long)need);
  DetDev d_out, d_sorted;
  if ((rc = to_det_dev(out, &d_out, cfg->top_k, "out", false)) != RON_OK) return rc;
  if ((rc = to_det_dev(sorted_out, &d_sorted, cfg->top_k, "sorted_out", true)) != RON_OK) return rc;
  PostDev pc;
  pc.obj_thr = cfg->objectness_thres; pc.sel_thr = cfg->select_threshold; pc.nms_thr = cfg->nms_threshold;
  pc.top_k = cfg->top_k; pc.flags = cfg->input_flags;
  for (int i = 0; i < 4; ++i) { pc.ref[i] = cfg->bbox_img[i]; pc.ps[i] = cfg->prior_scaling[i]; }
  hipStream_t s = (hipStream_t)stream;
  // Layout: [counts][part_total][keys of n images][partial keys].  A self-cleaning workspace (kPostWsClean) is reused by calls of
  // different n and must keep its counters where they are: they are laid out for the largest batch the workspace holds, not for n
  // (laid out for n, a batch of 1 wrote its keys over the counters of a later batch of 32)
  const int cap = hd.anchor_base[RON_MAX_LAYERS] * (hd.num_classes - 1);
  auto bytes_for = [&](int64_t k) {           // = ron_post_np_workspace_bytes(heads, k)
    return 2 * ron::align_up(k * kCountStride * 4, 256) + k * cap * 8 + k * kPartChunks * kMaxTopK * 8;
  };
  int n_layout = n;
  if (cfg->input_flags & ron::kPostWsClean)
    while (bytes_for(n_layout + 1) <= workspace_bytes) ++n_layout;
  const int64_t cnt_bytes = ron::align_up((int64_t)n_layout * kCountStride * 4, 256);
  int* counts = (int*)workspace;
  int* part_total = (int*)((char*)workspace + cnt_bytes);
  u64* keys = (u64*)((char*)workspace + 2 * cnt_bytes);
  u64* part_keys = keys + (size_t)n_layout * cap;
  if (!(cfg->input_flags & ron::kPostWsClean)) RON_HIP_CHECK(hipMemsetAsync(counts, 0, 2 * cnt_bytes, s));
  dim3 grid(hd.block_base[RON_MAX_LAYERS], n);
  const size_t lds = (size_t)kSelectThreads * hd.num_classes * sizeof(float);
  if (lds > 48 * 1024) {          // beyond ~48 classes the staging tile needs the dynamic-LDS limit raised (128 classes: 128 KB of the CU's 160)
    static ron::PerDeviceOnce once;
    RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&select_kernel), (int)lds));
  }
  RON_LAUNCH(select_kernel, grid, dim3(kSelectThreads), lds, s, hd, pc, keys, counts, cap);
  // only lists that can exceed kPartMin need the partial pass at all (its workgroups return at once for shorter ones)
  if (cap > kPartMin)
    RON_LAUNCH(topk_partial_kernel, dim3(kPartChunks, n), dim3(kTopkThreads), 0, s, keys, counts, cap, pc.top_k, part_keys, part_total);
  RON_LAUNCH(topk_nms_kernel, dim3(n), dim3(kTopkThreads), 0, s, hd, pc, keys, counts, cap, part_keys, part_total, d_out,
                     d_sorted, n_candidates);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

extern "C" int64_t ron_np_sort_nms_workspace_bytes(int n, int n_in) {
  if (n <= 0 || n_in < 0) return -1;
  return (int64_t)n * (n_in > 0 ? n_in : 1) * 8;
}

extern "C" int ron_np_sort_nms(const int32_t* classes, const float* scores, const float* bboxes,
                               const int32_t* n_valid, int n, int n_in, int top_k, float nms_threshold,
                               void* workspace, int64_t workspace_bytes, ron_detections* out,
                               ron_detections* sorted_out, void* stream) {
  RON_REQUIRE(n > 0 && n_in >= 0, "bad n / n_in");
  RON_REQUIRE(top_k >= 1 && top_k <= kMaxTopK, "top_k %d not in [1, %d]", top_k, kMaxTopK);
  RON_REQUIRE(n_in == 0 || (classes && scores && bboxes), "NULL input list");
  RON_REQUIRE(workspace != nullptr && workspace_bytes >= ron_np_sort_nms_workspace_bytes(n, n_in), "workspace too small");
  DetDev d_out, d_sorted;
  int rc;
  if ((rc = to_det_dev(out, &d_out, top_k, "out", false)) != RON_OK) return rc;
  if ((rc = to_det_dev(sorted_out, &d_sorted, top_k, "sorted_out", true)) != RON_OK) return rc;
  RON_LAUNCH(list_sort_nms_kernel, dim3(n), dim3(kTopkThreads), 0, (hipStream_t)stream, classes, scores,
                     bboxes, n_valid, n_in, top_k, nms_threshold, (u64*)workspace, d_out, d_sorted);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

extern "C" int ron_bboxes_decode_layer(const float* loc, int n, int feat_h, int feat_w, int num_anchors,
                                       const float* anchor_y, const float* anchor_x, const float* anchor_h,
                                       const float* anchor_w, const float prior_scaling[4], float* out,
                                       void* stream) {
  RON_REQUIRE(loc && out && anchor_y && anchor_x && anchor_h && anchor_w && prior_scaling, "NULL argument");
  RON_REQUIRE(n > 0 && feat_h > 0 && feat_w > 0 && num_anchors > 0, "bad shape");
  const size_t total = (size_t)n * feat_h * feat_w * num_anchors;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  RON_LAUNCH(decode_layer_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, loc, n, feat_h * feat_w,
                     num_anchors, anchor_y, anchor_x, anchor_h, anchor_w, prior_scaling[0], prior_scaling[1],
                     prior_scaling[2], prior_scaling[3], out);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

namespace {
__global__ void pack_records_kernel(DetDev d, float* __restrict__ rec) {
  const int img = blockIdx.x, cap = d.capacity;
  const int cnt = d.count[img];
  float* r = rec + (size_t)img * (cap + 1) * 7;
  for (int i = threadIdx.x; i < (cap + 1) * 7; i += blockDim.x) {
    const int row = i / 7, col = i - row * 7;
    float v = 0.f;
    if (row == cap) v = (float)cnt;
    else if (row < cnt) {
      const size_t k = (size_t)img * cap + row;
      v = col == 0 ? (float)d.classes[k] : col == 1 ? d.scores[k] : col == 6 ? (float)d.anchor_index[k] : d.bboxes[k * 4 + col - 2];
    }
    r[i] = v;
  }
}
}  // namespace

extern "C" int ron_pack_records(const ron_detections* det, int n, float* records, void* stream) {
  RON_REQUIRE(det != nullptr && records != nullptr && n > 0, "bad argument");
  RON_REQUIRE(det->capacity >= 1, "ron_pack_records: empty detection buffers");
  DetDev d{det->capacity, det->classes, det->scores, det->bboxes, det->anchor_index, det->count};
  RON_REQUIRE(d.classes && d.scores && d.bboxes && d.anchor_index && d.count, "ron_pack_records: every detection array is needed");
  RON_LAUNCH(pack_records_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, d, records);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

extern "C" int ron_softmax_last(const float* x, int64_t rows, int c, int pick, float* y, void* stream) {
  RON_REQUIRE(x && y && rows > 0 && c > 0 && pick < c, "bad argument");
  const int blocks = (int)std::min<int64_t>((rows + 255) / 256, 4096);
  RON_LAUNCH(softmax_last_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, c,
                     pick, y);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

// ==========================================================================================
// TF-evaluation variant (what eval_ron_network.py:226-236 runs through RONNet.detected_bboxes,
// nets/ron_vgg_320.py:234-256): per class select (ssd_common.py:504-589) -> clip with repair
// (tf_extended/bboxes.py:105-144) -> bboxes_filter_min (ron_vgg_320.py:196-233) -> top_k sort
// (bboxes.py:60-101) -> greedy NMS, mode 'min' | 'union', at most keep_top_k (bboxes.py:173-234)
// -> zero padding (tensors.py:59-86).  The dense zeroed rows of the TF graph are never
// materialised: rows with score <= thr are zero rows there, which neither suppress nor survive
// the size filter, so only the selected candidates are listed (per image AND class).
// ==========================================================================================
namespace {

struct TfeDev {
  float obj_thr, sel_thr, nms_thr;
  int top_k, keep_top_k, nms_mode, clip;
  float ref[4];
  float min_size;
  float ps[4];
  unsigned flags;
};

__device__ __forceinline__ void tfe_box(const HeadsDev& hd, const TfeDev& pc, int img, int layer, int local, float* box) {
  const int A = hd.num_anchors[layer];
  const int cell = local / A, a = local - cell * A;
  const size_t n_anchor_layer = (size_t)hd.cells[layer] * A;
  const float* l4 = hd.loc[layer] + ((size_t)img * n_anchor_layer + local) * 4;
  if (pc.flags & RON_IN_LOC_DECODED) {
    box[0] = l4[0]; box[1] = l4[1]; box[2] = l4[2]; box[3] = l4[3];
  } else {
    decode_box(l4, hd.ay[layer][cell], hd.ax[layer][cell], hd.ah[layer][a], hd.aw[layer][a], pc.ps, box);
  }
  if (pc.clip) {   // tf_extended/bboxes.py:130-142
    box[0] = fmaxf(box[0], pc.ref[0]); box[1] = fmaxf(box[1], pc.ref[1]);
    box[2] = fminf(box[2], pc.ref[2]); box[3] = fminf(box[3], pc.ref[3]);
    box[0] = fminf(box[0], box[2]);    box[1] = fminf(box[1], box[3]);
  }
}

// one thread per anchor; candidates go to per-(image, class) lists keyed by (score, anchor index)
__global__ __launch_bounds__(kSelectThreads) void tfe_select_kernel(HeadsDev hd, TfeDev pc, u64* keys, int* counts,
                                                                    int cap) {
  extern __shared__ __attribute__((aligned(16))) float stage[];
  const int img = blockIdx.y;
  const int tid = threadIdx.x;
  int layer = 0;
#pragma unroll
  for (int l = 1; l < RON_MAX_LAYERS; ++l)
    if (l < hd.num_layers && (int)blockIdx.x >= hd.block_base[l]) layer = l;
  const int C = hd.num_classes;
  const int n_anchor_layer = hd.cells[layer] * hd.num_anchors[layer];
  const int first = ((int)blockIdx.x - hd.block_base[layer]) * kSelectThreads;
  const int n_here = min(kSelectThreads, n_anchor_layer - first);
  const float* cls = hd.cls[layer] + ((size_t)img * n_anchor_layer + first) * C;
  for (int i = tid; i < n_here * C; i += kSelectThreads) stage[i] = cls[i];
  __syncthreads();
  if (tid >= n_here) return;
  const int local = first + tid;
  if (hd.obj[layer] != nullptr) {
    float objp;
    if (pc.flags & RON_IN_OBJ_IS_PROB) {
      objp = hd.obj[layer][(size_t)img * n_anchor_layer + local];
    } else {
      const float2 o = *reinterpret_cast<const float2*>(hd.obj[layer] + ((size_t)img * n_anchor_layer + local) * 2);
      const float m = fmaxf(o.x, o.y);
      const float e0 = expf(o.x - m), e1 = expf(o.y - m);
      objp = e1 / (e0 + e1);
    }
    if (!(objp > pc.obj_thr)) return;
  }
  const float* row = stage + tid * C;
  const bool is_prob = (pc.flags & RON_IN_CLS_IS_PROB) != 0;
  float sum = 1.f, mx = 0.f;
  if (!is_prob) {
    mx = row[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(row[c] - mx);
    sum = s;
  }
  bool have_box = false, size_ok = true;
  for (int c = 1; c < C; ++c) {
    const float sc = is_prob ? row[c] : expf(row[c] - mx) / sum;
    if (!(sc > pc.sel_thr)) continue;
    if (!have_box) {
      float box[4];
      tfe_box(hd, pc, img, layer, local, box);
      if (pc.min_size >= 0.f) {   // ron_vgg_320.py:222-225
        const float h = box[2] - box[0], w = box[3] - box[1];
        size_ok = (w > pc.min_size) && (h > pc.min_size);
      }
      have_box = true;
    }
    if (!size_ok) break;
    const size_t list = (size_t)img * (C - 1) + (c - 1);
    const int pos = atomicAdd(&counts[list], 1);
    if (pos < cap) keys[list * cap + pos] = make_key(sc, (unsigned)(hd.anchor_base[layer] + local));
  }
}

__global__ __launch_bounds__(kTopkThreads) void tfe_topk_nms_kernel(HeadsDev hd, TfeDev pc, const u64* keys,
                                                                    const int* counts, int cap, float* out_scores,
                                                                    float* out_boxes) {
  __shared__ ImageLds lds;
  const int c1 = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  const int C1 = hd.num_classes - 1;
  const size_t list = (size_t)img * C1 + c1;
  const int m = min(counts[list], cap);
  const int n = topk_keys(keys + list * cap, m, pc.top_k, lds);
  for (int r = tid; r < n; r += blockDim.x) {
    const u64 k = lds.sort[r];
    const int anchor = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
    int layer = 0;
#pragma unroll
    for (int l = 1; l < RON_MAX_LAYERS; ++l)
      if (l < hd.num_layers && anchor >= hd.anchor_base[l]) layer = l;
    float box[4];
    tfe_box(hd, pc, img, layer, anchor - hd.anchor_base[layer], box);
    lds.box[r][0] = box[0]; lds.box[r][1] = box[1]; lds.box[r][2] = box[2]; lds.box[r][3] = box[3];
    lds.score[r] = __uint_as_float((unsigned)(k >> 32));
    lds.cls[r] = c1 + 1;
    lds.anchor[r] = anchor;
  }
  __syncthreads();
  nms_scan(lds, n, pc.nms_thr, pc.nms_mode == 1 ? 2 : 1, pc.keep_top_k);
  float* os = out_scores + list * pc.keep_top_k;
  float* ob = out_boxes + list * pc.keep_top_k * 4;
  const int total = lds.scalars[4];
  for (int i = tid; i < pc.keep_top_k; i += blockDim.x)
    if (i >= total) { os[i] = 0.f; ob[i * 4 + 0] = 0.f; ob[i * 4 + 1] = 0.f; ob[i * 4 + 2] = 0.f; ob[i * 4 + 3] = 0.f; }
  for (int i = tid; i < n; i += blockDim.x) {
    const int pos = nms_slot(lds, i);
    if (pos >= 0 && pos < pc.keep_top_k) {
      os[pos] = lds.score[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) ob[pos * 4 + q] = lds.box[i][q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// ron_eval.py variant (the reference's second, per-image harness): flaten_predict (ron_eval.py:111-144) ->
// tfe.bboxes_clip -> filter_boxes (:369-392) -> tf_bboxes_nms (:146-206, class agnostic) -> bboxes_resize.
// ------------------------------------------------------------------------------------------
struct EvalDev {
  float obj_thr, sel_thr, nms_thr;
  int keep_top_k, nms_mode;
  float ref[4];
  float ps[4];
  unsigned flags;
  const float* min_size;     // [n] filter_boxes' min_size per image
};

__device__ __forceinline__ void eval_box(const HeadsDev& hd, const EvalDev& pc, int img, int layer, int local, float* box) {
  TfeDev t;
  t.flags = pc.flags; t.clip = 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) { t.ref[i] = pc.ref[i]; t.ps[i] = pc.ps[i]; }
  tfe_box(hd, t, img, layer, local, box);
}

// one thread per anchor: score[c] = objectness * class probability, label = argmax over ALL classes (first maximum),
// kept when label > 0 and objectness > thres, the clipped box passes filter_boxes and score[label] > select_threshold.
// key = (score, anchor * 64 + label): score descending, then the flattened anchor order like tf.nn.top_k.
__global__ __launch_bounds__(kSelectThreads) void eval_select_kernel(HeadsDev hd, EvalDev pc, u64* keys, int* counts, int cap) {
  extern __shared__ __attribute__((aligned(16))) float stage[];
  const int img = blockIdx.y;
  const int tid = threadIdx.x;
  int layer = 0;
#pragma unroll
  for (int l = 1; l < RON_MAX_LAYERS; ++l)
    if (l < hd.num_layers && (int)blockIdx.x >= hd.block_base[l]) layer = l;
  const int C = hd.num_classes;
  const int n_anchor_layer = hd.cells[layer] * hd.num_anchors[layer];
  const int first = ((int)blockIdx.x - hd.block_base[layer]) * kSelectThreads;
  const int n_here = min(kSelectThreads, n_anchor_layer - first);
  const float* cls = hd.cls[layer] + ((size_t)img * n_anchor_layer + first) * C;
  for (int i = tid; i < n_here * C; i += kSelectThreads) stage[i] = cls[i];
  __syncthreads();
  if (tid >= n_here) return;
  const int local = first + tid;
  float objp;
  if (pc.flags & RON_IN_OBJ_IS_PROB) {
    objp = hd.obj[layer][(size_t)img * n_anchor_layer + local];
  } else {
    const float2 o = *reinterpret_cast<const float2*>(hd.obj[layer] + ((size_t)img * n_anchor_layer + local) * 2);
    const float m = fmaxf(o.x, o.y);
    const float e0 = expf(o.x - m), e1 = expf(o.y - m);
    objp = e1 / (e0 + e1);
  }
  if (!(objp > pc.obj_thr)) return;
  const float* row = stage + tid * C;
  const bool is_prob = (pc.flags & RON_IN_CLS_IS_PROB) != 0;
  float sum = 1.f, mx = 0.f;
  if (!is_prob) {
    mx = row[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
    float sacc = 0.f;
    for (int c = 0; c < C; ++c) sacc += expf(row[c] - mx);
    sum = sacc;
  }
  int label = 0;
  float best = objp * (is_prob ? row[0] : expf(row[0] - mx) / sum);
  for (int c = 1; c < C; ++c) {
    const float sc = objp * (is_prob ? row[c] : expf(row[c] - mx) / sum);
    if (sc > best) { best = sc; label = c; }
  }
  if (label == 0 || !(best > pc.sel_thr)) return;
  float box[4];
  eval_box(hd, pc, img, layer, local, box);
  const float ws = box[3] - box[1], hs = box[2] - box[0];
  const float xc = box[1] + ws / 2.f, yc = box[0] + hs / 2.f;
  const float ms = pc.min_size[img];
  if (!(ws > ms && hs > ms && xc > 0.f && yc > 0.f && xc < 1.f && yc < 1.f)) return;
  if (pc.nms_mode & 4) {
    // tf_bboxes_nms_by_class (ron_eval.py:212-280): one list per score column, background included; a row is a candidate of every
    // class whose score passes select_threshold (:226)
    for (int c = 0; c < C; ++c) {
      const float sc = objp * (is_prob ? row[c] : expf(row[c] - mx) / sum);
      if (!(sc > pc.sel_thr)) continue;
      const size_t list = (size_t)img * C + c;
      const int pos = atomicAdd(&counts[list * kCountStride], 1);
      if (pos < cap) keys[list * cap + pos] = make_key(sc, (unsigned)(hd.anchor_base[layer] + local) * (unsigned)kMaxClasses + (unsigned)c);
    }
    return;
  }
  const int pos = atomicAdd(&counts[img * kCountStride], 1);
  if (pos < cap) keys[(size_t)img * cap + pos] = make_key(best, (unsigned)(hd.anchor_base[layer] + local) * (unsigned)kMaxClasses + (unsigned)label);
}

constexpr int kEvalCand = 1024;     // NMS candidates of the ron_eval.py variant per pass: one thread each

// The reference takes ALL candidates (ron_eval.py:111-144 -> tf_bboxes_nms).  Greedy NMS only ever looks at candidates
// in score order, so they are taken kEvalCand at a time: a pass sorts the next kEvalCand highest keys, drops the ones a
// box kept by an earlier pass overlaps, and continues the greedy scan; it ends when keep_top_k rows are kept or the
// candidates run out.  One pass in practice (the reference's thresholds let a few dozen through), exact for any count.
// nms_mode | 4 (tf_bboxes_nms_by_class, ron_eval.py:212-280): the grid's y dimension walks the score columns; a list's kept rows go to
// `best[img][anchor]` = max over the classes that kept the anchor of (score, lowest class first) - keep_scores' reduce_max / argmax of
// :268-272 - and eval_merge_kernel writes the records in anchor order.
__global__ __launch_bounds__(kTopkThreads) void eval_nms_kernel(HeadsDev hd, EvalDev pc, const u64* keys, const int* counts,
                                                                int cap, DetDev out, u64* best) {
  static_assert(kEvalCand == kTopkThreads && kSortCap >= kEvalCand + kEvalCand * 2 + kEvalCand / 2,
                "one thread per candidate; boxes and labels live behind the keys");
  __shared__ ImageLds lds;
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool per_class_lists = (pc.nms_mode & 4) != 0;
  const size_t list = per_class_lists ? (size_t)img * hd.num_classes + blockIdx.y : (size_t)img;
  keys += list * cap - (size_t)img * cap;
  const int m = min(counts[list * kCountStride], cap);
  float* box = reinterpret_cast<float*>(&lds.sort[kEvalCand]);              // [kEvalCand][4]
  int* lab = reinterpret_cast<int*>(&lds.sort[kEvalCand + kEvalCand * 2]);  // [kEvalCand] labels of this pass's candidates
  u64* words = reinterpret_cast<u64*>(lds.hist);                            // alive bits, one word per wave
  const int mode = (pc.nms_mode & 1) ? 2 : 1;
  // tf_bboxes_nms_by_class_v1 (ron_eval.py:282-366): a kept box only suppresses boxes of ITS label; classes are independent,
  // so one greedy pass in score order with the label test is the reference's loop over the classes, and "the first keep_top_k of the
  // kept rows in score order" (its final cut) is where this scan stops anyway
  const bool by_class = (pc.nms_mode & 2) != 0;
  const int max_keep = per_class_lists ? pc.keep_top_k : min(pc.keep_top_k, out.capacity);
  int n_kept = 0;                                                           // kept records: lds.box / score / cls / anchor
  u64 below = ~0ull;
  for (int consumed = 0; consumed < m && n_kept < max_keep; consumed += kEvalCand) {
    const int n = topk_keys(keys + (size_t)img * cap, m, kEvalCand, lds, below, m - consumed);   // sorted, the next highest scores
    float my[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned my_p = 0;
    float my_score = 0.f;
    if (tid < n) {
      const u64 k = lds.sort[tid];
      my_score = __uint_as_float((unsigned)(k >> 32));
      my_p = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
      const int anchor = (int)(my_p / (unsigned)kMaxClasses);
      int layer = 0;
#pragma unroll
      for (int l = 1; l < RON_MAX_LAYERS; ++l)
        if (l < hd.num_layers && anchor >= hd.anchor_base[l]) layer = l;
      eval_box(hd, pc, img, layer, anchor - hd.anchor_base[layer], my);
    }
    if (n > 0) below = lds.sort[n - 1];                                     // the next pass continues under this key
    __syncthreads();                                                        // every key is read before boxes overwrite the tail
    box[tid * 4 + 0] = my[0]; box[tid * 4 + 1] = my[1]; box[tid * 4 + 2] = my[2]; box[tid * 4 + 3] = my[3];
    const int my_label = (int)(my_p & (unsigned)(kMaxClasses - 1));
    lab[tid] = my_label;
    bool alive = tid < n;
    for (int j = 0; j < n_kept && alive; ++j)                               // boxes kept by earlier passes come first in score order
      if ((!by_class || lds.cls[j] == my_label) && tfe_suppresses(lds.box[j], my, pc.nms_thr, mode)) alive = false;
    // greedy, class agnostic, in score order (ron_eval.py:187-203): pick the first live row, drop every live row it overlaps
    while (n_kept < max_keep) {
      const u64 bal = __ballot(alive);
      if (lane == 0) words[wave] = bal;
      __syncthreads();
      int first = -1;
#pragma unroll
      for (int w = kTopkThreads / 64 - 1; w >= 0; --w)
        if (words[w] != 0ull) first = w * 64 + __ffsll((long long)words[w]) - 1;
      if (first < 0) break;
      if (tid == first) {
        alive = false;
        lds.box[n_kept][0] = my[0]; lds.box[n_kept][1] = my[1]; lds.box[n_kept][2] = my[2]; lds.box[n_kept][3] = my[3];
        lds.score[n_kept] = my_score;
        lds.cls[n_kept] = (int)(my_p & (unsigned)(kMaxClasses - 1));
        lds.anchor[n_kept] = (int)(my_p / (unsigned)kMaxClasses);
      }
      ++n_kept;
      if (alive && (!by_class || lab[first] == my_label) && tfe_suppresses(&box[first * 4], my, pc.nms_thr, mode)) alive = false;
      __syncthreads();
    }
    __syncthreads();
    if (n < kEvalCand) break;                                               // that was the last of them
  }
  __syncthreads();
  const int total = n_kept;
  if (per_class_lists) {
    // score > 0 (it passed select_threshold >= 0): a zero word is "not kept"; among equal scores the lowest class wins (argmax)
    for (int i = tid; i < total; i += blockDim.x)
      atomicMax(&best[(size_t)img * cap + lds.anchor[i]], ((u64)__float_as_uint(lds.score[i]) << 8) | (u64)(255 - lds.cls[i]));
    return;
  }
  const float sy = pc.ref[2] - pc.ref[0], sx = pc.ref[3] - pc.ref[1];
  for (int i = tid; i < out.capacity; i += blockDim.x) {
    const size_t o = (size_t)img * out.capacity + i;
    if (i >= total) {
      out.classes[o] = 0; out.scores[o] = 0.f; out.anchor_index[o] = 0;
      out.bboxes[o * 4 + 0] = 0.f; out.bboxes[o * 4 + 1] = 0.f; out.bboxes[o * 4 + 2] = 0.f; out.bboxes[o * 4 + 3] = 0.f;
    } else {
      out.classes[o] = lds.cls[i];
      out.scores[o] = lds.score[i];
      out.anchor_index[o] = lds.anchor[i];
      out.bboxes[o * 4 + 0] = (lds.box[i][0] - pc.ref[0]) / sy;      // tfe.bboxes_resize, tf_extended/bboxes.py:147-171
      out.bboxes[o * 4 + 1] = (lds.box[i][1] - pc.ref[1]) / sx;
      out.bboxes[o * 4 + 2] = (lds.box[i][2] - pc.ref[0]) / sy;
      out.bboxes[o * 4 + 3] = (lds.box[i][3] - pc.ref[1]) / sx;
    }
  }
  if (tid == 0) out.count[img] = total;
}

// tf_bboxes_nms_by_class, :268-274: the anchors some class kept, in the flattened anchor order (boolean_mask), each with its best
// kept score and that score's class; boxes clipped like the candidates', then tfe.bboxes_resize.  One workgroup per image.
__global__ __launch_bounds__(kTopkThreads) void eval_merge_kernel(HeadsDev hd, EvalDev pc, const u64* best, int cap, DetDev out) {
  __shared__ int wave_sum[kTopkThreads / 64];
  __shared__ int base_s;
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float sy = pc.ref[2] - pc.ref[0], sx = pc.ref[3] - pc.ref[1];
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int a0 = 0; a0 < cap; a0 += kTopkThreads) {
    const int anchor = a0 + tid;
    const u64 v = anchor < cap ? best[(size_t)img * cap + anchor] : 0ull;
    const bool has = v != 0ull;
    const u64 bal = __ballot(has);
    if (lane == 0) wave_sum[wave] = __popcll(bal);
    __syncthreads();
    int pos = base_s + __popcll(bal & ((1ull << lane) - 1ull));
    int all = 0;
#pragma unroll
    for (int w = 0; w < kTopkThreads / 64; ++w) {
      if (w < wave) pos += wave_sum[w];
      all += wave_sum[w];
    }
    if (has && pos < out.capacity) {
      int layer = 0;
#pragma unroll
      for (int l = 1; l < RON_MAX_LAYERS; ++l)
        if (l < hd.num_layers && anchor >= hd.anchor_base[l]) layer = l;
      float b[4];
      eval_box(hd, pc, img, layer, anchor - hd.anchor_base[layer], b);
      const size_t o = (size_t)img * out.capacity + pos;
      out.classes[o] = 255 - (int)(v & 0xFFull);
      out.scores[o] = __uint_as_float((unsigned)(v >> 8));
      out.anchor_index[o] = anchor;
      out.bboxes[o * 4 + 0] = (b[0] - pc.ref[0]) / sy;
      out.bboxes[o * 4 + 1] = (b[1] - pc.ref[1]) / sx;
      out.bboxes[o * 4 + 2] = (b[2] - pc.ref[0]) / sy;
      out.bboxes[o * 4 + 3] = (b[3] - pc.ref[1]) / sx;
    }
    __syncthreads();
    if (tid == 0) base_s += all;
    __syncthreads();
  }
  const int total = min(base_s, out.capacity);
  for (int i = total + tid; i < out.capacity; i += blockDim.x) {
    const size_t o = (size_t)img * out.capacity + i;
    out.classes[o] = 0; out.scores[o] = 0.f; out.anchor_index[o] = 0;
    out.bboxes[o * 4 + 0] = 0.f; out.bboxes[o * 4 + 1] = 0.f; out.bboxes[o * 4 + 2] = 0.f; out.bboxes[o * 4 + 3] = 0.f;
  }
  if (tid == 0) out.count[img] = total;
}

// RONNet.bboxes_filter_min (nets/ron_vgg_320.py:196-233) on its own: per list, the rows with w > minsize and h > minsize in their
// order (tf.boolean_mask), zeros behind them (pad_axis).  One workgroup per list.
__global__ __launch_bounds__(kTopkThreads) void filter_min_kernel(const float* scores, const float* bboxes, int rows, float minsize,
                                                                  float* out_scores, float* out_bboxes, int out_rows, int* counts) {
  __shared__ int wave_sum[kTopkThreads / 64];
  __shared__ int base_s;
  const int list = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sc = scores + (size_t)list * rows;
  const float* bb = bboxes + (size_t)list * rows * 4;
  float* os = out_scores + (size_t)list * out_rows;
  float* ob = out_bboxes + (size_t)list * out_rows * 4;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int r0 = 0; r0 < rows; r0 += kTopkThreads) {
    const int r = r0 + tid;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    bool has = false;
    if (r < rows) {
      b = *reinterpret_cast<const float4*>(bb + (size_t)r * 4);
      const float h = b.z - b.x, w = b.w - b.y;                  // (ymin, xmin, ymax, xmax)
      has = (w > minsize) && (h > minsize);
    }
    const u64 bal = __ballot(has);
    if (lane == 0) wave_sum[wave] = __popcll(bal);
    __syncthreads();
    int pos = base_s + __popcll(bal & ((1ull << lane) - 1ull));
    int all = 0;
#pragma unroll
    for (int w = 0; w < kTopkThreads / 64; ++w) {
      if (w < wave) pos += wave_sum[w];
      all += wave_sum[w];
    }
    if (has) {
      os[pos] = sc[r];
      *reinterpret_cast<float4*>(ob + (size_t)pos * 4) = b;
    }
    __syncthreads();
    if (tid == 0) base_s += all;
    __syncthreads();
  }
  const int total = base_s;
  for (int i = total + tid; i < out_rows; i += blockDim.x) {
    os[i] = 0.f;
    *reinterpret_cast<float4*>(ob + (size_t)i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (tid == 0) counts[list] = total;
}

}  // namespace

extern "C" int ron_bboxes_filter_min(const float* scores, const float* bboxes, int num_lists, int rows, float minsize,
                                     float* out_scores, float* out_bboxes, int out_rows, int32_t* counts, void* stream) {
  RON_REQUIRE(scores != nullptr && bboxes != nullptr && out_scores != nullptr && out_bboxes != nullptr && counts != nullptr, "bad argument");
  RON_REQUIRE(num_lists > 0 && rows >= 0 && out_rows >= rows, "bboxes_filter_min: %d lists of %d rows into %d rows", num_lists, rows, out_rows);
  RON_REQUIRE(((uintptr_t)bboxes & 15) == 0 && ((uintptr_t)out_bboxes & 15) == 0, "bboxes_filter_min: box arrays must be 16-byte aligned");
  RON_LAUNCH(filter_min_kernel, dim3(num_lists), dim3(kTopkThreads), 0, (hipStream_t)stream, scores, bboxes, rows, minsize, out_scores,
             out_bboxes, out_rows, counts);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

static int64_t eval_lists(const ron_heads* heads, int n, int nms_mode) { return (nms_mode & 4) ? (int64_t)n * heads->num_classes : (int64_t)n; }

// workspace: [counters, one 128-byte line per list][keys: lists x anchors][nms_mode | 4: best, n x anchors]
extern "C" int64_t ron_post_eval_workspace_bytes_mode(const ron_heads* heads, int n, int nms_mode) {
  HeadsDev hd;
  if (heads == nullptr || build_heads_dev(heads, &hd, false) != RON_OK || n <= 0) return -1;
  const int64_t lists = eval_lists(heads, n, nms_mode);
  return ron::align_up(lists * kCountStride * 4, 256) + lists * hd.anchor_base[RON_MAX_LAYERS] * 8 +
         ((nms_mode & 4) ? (int64_t)n * hd.anchor_base[RON_MAX_LAYERS] * 8 : 0);
}

extern "C" int64_t ron_post_eval_workspace_bytes(const ron_heads* heads, int n) { return ron_post_eval_workspace_bytes_mode(heads, n, 0); }

extern "C" int ron_post_eval(const ron_heads* heads, int n, const float* min_sizes, const ron_eval_cfg* cfg, void* workspace,
                             int64_t workspace_bytes, ron_detections* out, void* stream) {
  RON_REQUIRE(cfg != nullptr && n > 0 && out != nullptr && min_sizes != nullptr, "bad argument");
  RON_REQUIRE(cfg->keep_top_k >= 1 && cfg->keep_top_k <= kMaxTopK, "keep_top_k %d not in [1, %d]", cfg->keep_top_k, kMaxTopK);
  RON_REQUIRE(cfg->nms_mode >= 0 && cfg->nms_mode <= 5, "unknown mode to use for nms.");
  RON_REQUIRE(heads != nullptr && heads->num_classes <= RON_MAX_CLASSES, "num_classes must be <= %d", RON_MAX_CLASSES);
  const bool per_class_lists = (cfg->nms_mode & 4) != 0;
  if (per_class_lists) {
    RON_REQUIRE(cfg->select_threshold >= 0.f, "tf_bboxes_nms_by_class: select_threshold must be >= 0");
    RON_REQUIRE((int64_t)out->capacity >= (int64_t)heads->num_classes * cfg->keep_top_k,
                "tf_bboxes_nms_by_class: every class may keep keep_top_k rows: capacity %d < %d x %d", out->capacity, heads->num_classes, cfg->keep_top_k);
  }
  HeadsDev hd;
  int rc = build_heads_dev(heads, &hd, (cfg->input_flags & RON_IN_LOC_DECODED) == 0);
  if (rc != RON_OK) return rc;
  for (int i = 0; i < hd.num_layers; ++i) RON_REQUIRE(hd.obj[i] != nullptr, "ron_post_eval needs the objectness tensors");
  DetDev d_out;
  if ((rc = to_det_dev(out, &d_out, cfg->keep_top_k, "out", false))) return rc;
  const int64_t need = ron_post_eval_workspace_bytes_mode(heads, n, cfg->nms_mode);
  RON_REQUIRE(workspace != nullptr && workspace_bytes >= need, "workspace too small: %lld < %lld", (long long)workspace_bytes, (long long)need);
  EvalDev pc;
  pc.obj_thr = cfg->objectness_thres; pc.sel_thr = cfg->select_threshold; pc.nms_thr = cfg->nms_threshold;
  pc.keep_top_k = cfg->keep_top_k; pc.nms_mode = cfg->nms_mode; pc.flags = cfg->input_flags; pc.min_size = min_sizes;
  for (int i = 0; i < 4; ++i) { pc.ref[i] = cfg->bbox_img[i]; pc.ps[i] = cfg->prior_scaling[i]; }
  hipStream_t s = (hipStream_t)stream;
  int* counts = (int*)workspace;
  const int64_t lists = eval_lists(heads, n, cfg->nms_mode);
  const int64_t cbytes = ron::align_up(lists * kCountStride * 4, 256);
  u64* keys = (u64*)((char*)workspace + cbytes);
  const int cap = hd.anchor_base[RON_MAX_LAYERS];
  u64* best = per_class_lists ? keys + lists * cap : nullptr;
  RON_HIP_CHECK(hipMemsetAsync(counts, 0, cbytes, s));
  if (per_class_lists) RON_HIP_CHECK(hipMemsetAsync(best, 0, (size_t)n * cap * 8, s));
  const size_t lds = (size_t)kSelectThreads * hd.num_classes * sizeof(float);
  if (lds > 48 * 1024) {          // beyond ~48 classes the staging tile needs the dynamic-LDS limit raised (128 classes: 128 KB of the CU's 160)
    static ron::PerDeviceOnce once;
    RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&eval_select_kernel), (int)lds));
  }
  RON_LAUNCH(eval_select_kernel, dim3(hd.block_base[RON_MAX_LAYERS], n), dim3(kSelectThreads), lds, s, hd, pc, keys, counts, cap);
  RON_LAUNCH(eval_nms_kernel, dim3(n, per_class_lists ? hd.num_classes : 1), dim3(kTopkThreads), 0, s, hd, pc, keys, counts, cap, d_out, best);
  if (per_class_lists) RON_LAUNCH(eval_merge_kernel, dim3(n), dim3(kTopkThreads), 0, s, hd, pc, best, cap, d_out);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

extern "C" int64_t ron_post_tfe_workspace_bytes(const ron_heads* heads, int n) {
  HeadsDev hd;
  if (build_heads_dev(heads, &hd, false) != RON_OK || n <= 0) return -1;
  const int64_t lists = (int64_t)n * (heads->num_classes - 1);
  return ron::align_up(lists * 4, 256) + lists * hd.anchor_base[RON_MAX_LAYERS] * 8;
}

extern "C" int ron_post_tfe(const ron_heads* heads, int n, const ron_tfe_cfg* cfg, void* workspace,
                            int64_t workspace_bytes, float* scores, float* bboxes, void* stream) {
  RON_REQUIRE(cfg != nullptr && n > 0 && scores != nullptr && bboxes != nullptr, "bad argument");
  RON_REQUIRE(cfg->top_k >= 1 && cfg->top_k <= kMaxTopK, "top_k %d not in [1, %d]", cfg->top_k, kMaxTopK);
  RON_REQUIRE(cfg->keep_top_k >= 1 && cfg->keep_top_k <= cfg->top_k, "keep_top_k %d not in [1, top_k]", cfg->keep_top_k);
  RON_REQUIRE(cfg->nms_mode == 0 || cfg->nms_mode == 1, "unknown mode to use for nms.");
  RON_REQUIRE(cfg->select_threshold >= 0.f, "select_threshold must be >= 0");
  HeadsDev hd;
  int rc = build_heads_dev(heads, &hd, (cfg->input_flags & RON_IN_LOC_DECODED) == 0);
  if (rc != RON_OK) return rc;
  const int64_t need = ron_post_tfe_workspace_bytes(heads, n);
  RON_REQUIRE(workspace != nullptr && workspace_bytes >= need, "workspace too small: %lld < %lld",
              (long long)workspace_bytes, (long long)need);
  TfeDev pc;
  pc.obj_thr = cfg->objectness_thres; pc.sel_thr = cfg->select_threshold; pc.nms_thr = cfg->nms_threshold;
  pc.top_k = cfg->top_k; pc.keep_top_k = cfg->keep_top_k; pc.nms_mode = cfg->nms_mode; pc.clip = cfg->clip;
  pc.min_size = cfg->min_size; pc.flags = cfg->input_flags;
  for (int i = 0; i < 4; ++i) { pc.ref[i] = cfg->clipping_bbox[i]; pc.ps[i] = cfg->prior_scaling[i]; }
  hipStream_t s = (hipStream_t)stream;
  const int C1 = hd.num_classes - 1;
  const int64_t lists = (int64_t)n * C1;
  int* counts = (int*)workspace;
  u64* keys = (u64*)((char*)workspace + ron::align_up(lists * 4, 256));
  const int cap = hd.anchor_base[RON_MAX_LAYERS];
  RON_HIP_CHECK(hipMemsetAsync(counts, 0, ron::align_up(lists * 4, 256), s));
  const size_t lds = (size_t)kSelectThreads * hd.num_classes * sizeof(float);
  if (lds > 48 * 1024) {          // beyond ~48 classes the staging tile needs the dynamic-LDS limit raised (128 classes: 128 KB of the CU's 160)
    static ron::PerDeviceOnce once;
    RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&tfe_select_kernel), (int)lds));
  }
  RON_LAUNCH(tfe_select_kernel, dim3(hd.block_base[RON_MAX_LAYERS], n), dim3(kSelectThreads), lds, s, hd, pc, keys,
                     counts, cap);
  RON_LAUNCH(tfe_topk_nms_kernel, dim3(C1, n), dim3(kTopkThreads), 0, s, hd, pc, keys, counts, cap, scores, bboxes);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}
