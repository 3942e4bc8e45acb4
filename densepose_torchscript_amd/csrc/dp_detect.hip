// Proposal / detection selection kernels: RPN top-k + decode, batched greedy NMS (wavefront bitmask),
// NHWC ROIAlign, second-stage decode + score filter, box post-processing.
// All box arithmetic is fp32 with FMA contraction OFF and IEEE division so that IoU / clip / level
// decisions follow the reference's CPU ops bit for bit wherever the inputs are equal.
#include "dp_common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ uint32_t f32_to_key(float f) {  // monotonic: larger float -> larger key
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_f32(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ bool finitef(float x) { return fabsf(x) <= 3.402823466e+38f; }

// In-LDS bitonic sort, descending, of n2 (power of two) 64-bit keys with `nthreads` threads.
__device__ void bitonic_sort_desc(unsigned long long* s, int n2, int tid, int nthreads) {
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < (n2 >> 1); t += nthreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = s[lo], b = s[hi];
        if (desc ? (a < b) : (a > b)) {
          s[lo] = b;
          s[hi] = a;
        }
      }
    }
  }
  __syncthreads();
}

// =====================================================================================================
// K9 + K8: per (image, level) top-k of the objectness logits, then decode of the survivors only.
// =====================================================================================================
constexpr int kSelThreads = 1024;

// One step of the MSB-first radix select: the bucket d of a 256-bin histogram that holds the `need`-th largest key - the
// largest d >= 1 with sum_{j >= d} hist[j] >= need, else 0 - and `above` = the number of keys in the buckets over d.
// Threads 0..255 of the workgroup each own a bin (suffix sums by wave shuffles + four wave totals); every thread of the
// workgroup must call it (it contains a barrier). A serial scan by one thread costs up to 255 dependent LDS reads per
// radix pass - that was most of the 66 / 87 us of the two select kernels.
__device__ __forceinline__ void radix_pick(const unsigned int* hist, unsigned int need, int tid, unsigned int* wave_tot /*[4]*/,
                                           unsigned int* out_d, unsigned int* out_above, unsigned int* out_count) {
  unsigned int h = 0, s = 0;
  const int lane = tid & 63, w = tid >> 6;
  if (tid < 256) {
    h = hist[tid];
    s = h;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned int v = __shfl_down(s, off, 64);
      if (lane + off < 64) s += v;
    }
    if (lane == 0) wave_tot[w] = s;     // sum of this wave's 64 bins
  }
  __syncthreads();
  if (tid < 256) {
    for (int ww = w + 1; ww < 4; ++ww) s += wave_tot[ww];    // s = sum_{j >= tid} hist[j]
    const unsigned int above = s - h;
    if (above < need && (s >= need || tid == 0)) {           // exactly one bin qualifies
      *out_d = (unsigned int)tid;
      *out_above = above;
      *out_count = h;
    }
  }
}

// ---- stage 1, ALL levels of a batch in one launch: workgroup = (8192-anchor chunk of a level, image).
// Small levels: the chunk's logits become sortable keys, nothing else. Large levels (>= 2 chunks): the workgroup keeps only the top
// min(k, chunk) keys of its chunk, held in LDS. The global top-k (ordered by key desc, anchor index asc) is a subset of the union
// of the per-chunk top-k's, so the single-workgroup-per-image stage 2 then scans n_chunks*k candidates instead of Hi*Wi*A keys
// (p2: 25k instead of 202k). Survivors are written in ascending anchor-index order (ordered compaction), which keeps stage 2's
// tie rule exact. (One launch per level - two chunk selections and three key conversions - was 5 launches in a row on the
// proposal chain, ~90 us where this one takes the time of the largest.)
constexpr int kChunk = 8192;
constexpr int kChunkThreads = 256;

struct RpnPrepLevel {
  const float* head;
  int cells, A, head_c, n, k, kcap, n_chunks, chunked;
  uint32_t* out_keys;    // chunked: [n_img][n_chunks][kcap] candidate keys, else [n_img][n] keys
  uint32_t* out_idx;     // chunked: the candidates' anchor indices
};
struct RpnPrepMulti {
  RpnPrepLevel lv[5];
  int first_block[6];
  int n_levels;
};

__global__ __launch_bounds__(kChunkThreads) void rpn_prep_kernel(const RpnPrepMulti pm) {
  __shared__ uint32_t keys[kChunk];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh_prefix, sh_need, sh_bucket_count;
  __shared__ unsigned int pick_tot[4];
  int lvl = 0;
  while (lvl + 1 < pm.n_levels && (int)blockIdx.x >= pm.first_block[lvl + 1]) ++lvl;
  const RpnPrepLevel& p = pm.lv[lvl];
  const int chunk = blockIdx.x - pm.first_block[lvl], img = blockIdx.y, tid = threadIdx.x;
  const int A = p.A, head_c = p.head_c, n = p.n, k = p.k, kcap = p.kcap, n_chunks = p.n_chunks;
  const int i0 = chunk * kChunk;
  const int len = min(kChunk, n - i0);
  const float* hbase = p.head + (long long)img * p.cells * head_c;
  if (!p.chunked) {
    uint32_t* ok = p.out_keys + (long long)img * n;
    for (int t = tid; t < len; t += kChunkThreads) {
      const int idx = i0 + t;
      ok[idx] = f32_to_key(hbase[(long long)(idx / A) * head_c + (idx % A)]);
    }
    return;
  }
  uint32_t* const cand_keys = p.out_keys;
  uint32_t* const cand_idx = p.out_idx;
  for (int t = tid; t < len; t += kChunkThreads) {
    const int idx = i0 + t;
    keys[t] = f32_to_key(hbase[(long long)(idx / A) * head_c + (idx % A)]);
  }
  uint32_t* okeys = cand_keys + ((long long)img * n_chunks + chunk) * kcap;
  uint32_t* oidx = cand_idx + ((long long)img * n_chunks + chunk) * kcap;
  const int kk = min(k, len);
  __syncthreads();
  uint32_t T = 0;
  unsigned int need_eq = 0xffffffffu;  // len <= k: everything is kept (T = 0: all keys >= T, unlimited ties)
  if (len > kk) {
    uint32_t prefix = 0, mask = 0;
    unsigned int need = (unsigned)kk;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int t = tid; t < 256; t += kChunkThreads) hist[t] = 0;
      __syncthreads();
      for (int t = tid; t < len; t += kChunkThreads) {
        const uint32_t key = keys[t];
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      radix_pick(hist, need, tid, pick_tot, &sh_prefix, &sh_need, &sh_bucket_count);   // (bucket, keys above it, keys in it)
      __syncthreads();
      prefix |= sh_prefix << shift;
      need -= sh_need;
      mask |= 255u << shift;
      __syncthreads();
    }
    T = prefix;
    need_eq = need;
  }
  // ordered compaction: position = (#kept before me); kept = key > T, or key == T while fewer than need_eq ties were taken.
  // Two sweeps instead of one with three barriers per 256 keys (that loop was most of the launch's 40 us): sweep 1 counts the kept
  // and the tied keys of every (256-key slice, wave) pair, one scan of the 128 counts turns them into start positions, sweep 2
  // recomputes the ballots and writes - no barrier inside either sweep.
  const int lane = tid & 63, w = tid >> 6;
  constexpr int NIT = kChunk / kChunkThreads, NCNT = NIT * (kChunkThreads / 64);
  static_assert(NCNT <= kChunkThreads && NCNT == 128, "one thread per (slice, wave) count, two waves of them");
  __shared__ unsigned int cnt[NCNT];      // kept | tied << 16 (both <= 8192)
  __shared__ unsigned int cnt_wave0;
  const bool all = len <= kk;
  for (int it = 0; it < NIT; ++it) {
    const int t = it * kChunkThreads + tid;
    const uint32_t key = t < len ? keys[t] : 0u;
    const bool gt = (t < len) && (all ? true : key > T);
    const bool eq = (t < len) && !all && key == T;
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    if (lane == 0) cnt[it * (kChunkThreads / 64) + w] = (unsigned)__popcll(bg) | ((unsigned)__popcll(be) << 16);
  }
  __syncthreads();
  {
    unsigned int own = tid < NCNT ? cnt[tid] : 0u, incl = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (tid == 63) cnt_wave0 = incl;
    __syncthreads();
    if (tid < NCNT) cnt[tid] = incl - own + (w == 1 ? cnt_wave0 : 0u);     // exclusive prefix in (slice, wave) order
  }
  __syncthreads();
  for (int it = 0; it < NIT; ++it) {
    const int t = it * kChunkThreads + tid;
    if (it * kChunkThreads >= len) break;
    const uint32_t key = t < len ? keys[t] : 0u;
    const bool gt = (t < len) && (all ? true : key > T);
    const bool eq = (t < len) && !all && key == T;
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    const unsigned int before = cnt[it * (kChunkThreads / 64) + w];
    const unsigned long long lm = (1ull << lane) - 1ull;
    const unsigned int my_gt = (before & 0xffffu) + (unsigned)__popcll(bg & lm);
    const unsigned int my_eq = (before >> 16) + (unsigned)__popcll(be & lm);
    const bool take_eq = eq && my_eq < need_eq;
    if (gt || take_eq) {
      // ties already taken before me (capped at need_eq) + greater keys before me
      const unsigned int pos = my_gt + (my_eq < need_eq ? my_eq : need_eq);
      okeys[pos] = key;
      oidx[pos] = (uint32_t)(i0 + t);
    }
  }
  // padding slots (chunks shorter than kcap): lowest key, sentinel index
  for (int t = kk + tid; t < kcap; t += kChunkThreads) { okeys[t] = 0u; oidx[t] = 0xffffffffu; }
}

struct RpnSelArgs {
  const float* head;
  const uint32_t* keys;
  const uint32_t* kidx;  // anchor index of each key (null: identity)
  int n_keys;            // keys per image
  int n_img, Hi, Wi, A, head_c, stride_px, level, kmax, slot_off, slots_per_img;
  float ca[3][4];
  float clip_x, clip_y;
  float* cand_boxes;
  float* cand_scores;
  int32_t* cand_level;
  int32_t* cand_valid;
};

struct RpnSelMulti {
  RpnSelArgs lv[5];
  int n2max;       // sort slots of the largest level: the key copy below starts behind them
  int keys_cap;    // keys of a (image, level) that fit the LDS copy
};

__global__ __launch_bounds__(kSelThreads) void rpn_select_kernel(const RpnSelMulti pm) {
  const RpnSelArgs& p = pm.lv[blockIdx.y];  // one workgroup per (image, level): all levels of a batch in ONE launch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem_raw);  // [n2]
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh_prefix, sh_need, sh_count, sh_eq_total, sh_eq_taken;
  __shared__ unsigned int wave_sums[kSelThreads / 64], pick_tot[4];

  const int img = blockIdx.x, tid = threadIdx.x;
  const int n = p.n_keys;
  const int n_anchors = p.Hi * p.Wi * p.A;
  const int k = n_anchors < p.kmax ? n_anchors : p.kmax;
  int n2 = 1;
  while (n2 < k) n2 <<= 1;
  const uint32_t* keys = p.keys + (long long)img * n;
  const uint32_t* kidx = p.kidx ? p.kidx + (long long)img * n : nullptr;
#define DP_AIDX(i) (kidx ? kidx[i] : (uint32_t)(i))
  // the keys are swept five to six times (four radix passes, the gather, the tie pass): one copy into LDS when they fit (the
  // p2 level's 25 x 1000 chunk candidates do), each sweep then costs LDS reads instead of an L2 round trip per 1024 keys
  uint32_t* const lkeys = reinterpret_cast<uint32_t*>(smem_raw + (size_t)pm.n2max * 8);
  const bool in_lds = n > p.kmax && n <= pm.keys_cap;
#define DP_KEY(i) (in_lds ? lkeys[i] : keys[i])

  for (int i = tid; i < n2; i += kSelThreads) sel[i] = 0ull;
  if (in_lds)
    for (int i = tid; i < n; i += kSelThreads) lkeys[i] = keys[i];
  if (tid == 0) { sh_count = 0; sh_eq_taken = 0; }
  __syncthreads();

  if (n <= p.kmax) {
    for (int i = tid; i < n; i += kSelThreads) sel[i] = ((unsigned long long)keys[i] << 32) | (uint32_t)(~DP_AIDX(i));
  } else {
    // ---- radix select (4 x 8 bits, MSB first) of the k-th largest key ----
    uint32_t prefix = 0, mask = 0;
    unsigned int need = (unsigned)k;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int i = tid; i < 256; i += kSelThreads) hist[i] = 0;
      __syncthreads();
      for (int i = tid; i < n; i += kSelThreads) {
        const uint32_t key = DP_KEY(i);
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      radix_pick(hist, need, tid, pick_tot, &sh_prefix, &sh_need, &sh_eq_total);   // (bucket d, keys above it, keys in it)
      __syncthreads();
      prefix |= sh_prefix << shift;
      need -= sh_need;              // how many are still needed inside bucket d
      mask |= 255u << shift;
      __syncthreads();
    }
    const uint32_t T = prefix;             // k-th largest key
    const unsigned int need_eq = need;     // number of entries == T to take
    const unsigned int eq_total = sh_eq_total;
    // ---- gather keys > T (any order: they are sorted afterwards) ----
    for (int i = tid; i < n; i += kSelThreads) {
      const uint32_t key = DP_KEY(i);
      if (key > T || (key == T && need_eq == eq_total)) {
        const unsigned int pos = atomicAdd(&sh_count, 1u);
        sel[pos] = ((unsigned long long)key << 32) | (uint32_t)(~DP_AIDX(i));
      }
    }
    __syncthreads();
    if (need_eq != eq_total) {
      // ties at the threshold: take the need_eq lowest indices (deterministic), ordered block scan
      const unsigned int base = sh_count;
      for (int i0 = 0; i0 < n; i0 += kSelThreads) {
        const int i = i0 + tid;
        const bool f = (i < n) && (DP_KEY(i) == T);
        const unsigned long long bal = __ballot(f);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) wave_sums[w] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned int before = sh_eq_taken;
        for (int ww = 0; ww < w; ++ww) before += wave_sums[ww];
        const unsigned int my = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (f && my < need_eq) sel[base + my] = ((unsigned long long)T << 32) | (uint32_t)(~DP_AIDX(i));
        __syncthreads();
        if (tid == 0) {
          unsigned int tot = 0;
          for (int ww = 0; ww < kSelThreads / 64; ++ww) tot += wave_sums[ww];
          sh_eq_taken += tot;
        }
        __syncthreads();
        if (sh_eq_taken >= need_eq) break;
      }
    }
  }
  bitonic_sort_desc(sel, n2, tid, kSelThreads);

  // ---- decode the k survivors (box_regression.py:74-112 with weights (1,1,1,1); anchors analytic) ----
  const float clampv = 4.135166556742356f;  // log(1000/16)
  for (int j = tid; j < p.kmax; j += kSelThreads) {
    const long long slot = (long long)img * p.slots_per_img + p.slot_off + j;
    if (j >= k) {
      p.cand_valid[slot] = 0;
      p.cand_scores[slot] = 0.f;
      p.cand_level[slot] = p.level;
      p.cand_boxes[slot * 4 + 0] = 0.f; p.cand_boxes[slot * 4 + 1] = 0.f; p.cand_boxes[slot * 4 + 2] = 0.f; p.cand_boxes[slot * 4 + 3] = 0.f;
      continue;
    }
    const unsigned long long e = sel[j];
    const float score = key_to_f32((uint32_t)(e >> 32));
    const int idx = (int)(~(uint32_t)(e & 0xffffffffull));
    const int a = idx % p.A;
    const int cell = idx / p.A;
    const int x = cell % p.Wi, y = cell / p.Wi;
    const float sx = (float)(x * p.stride_px), sy = (float)(y * p.stride_px);
    const float ax1 = sx + p.ca[a][0], ay1 = sy + p.ca[a][1], ax2 = sx + p.ca[a][2], ay2 = sy + p.ca[a][3];
    const float* d = p.head + ((long long)img * p.Hi * p.Wi + cell) * p.head_c + p.A + a * 4;
    const float widths = ax2 - ax1, heights = ay2 - ay1;
    const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;
    const float dx = d[0], dy = d[1];
    const float dw = fminf(d[2], clampv), dh = fminf(d[3], clampv);
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
    bool ok = finitef(x1) && finitef(y1) && finitef(x2) && finitef(y2) && finitef(score) && !(d[2] != d[2]) && !(d[3] != d[3]);
    // Q1: x clamped to clip_x (= padded HEIGHT), y to clip_y (= padded WIDTH)
    x1 = fminf(fmaxf(x1, 0.f), p.clip_x); y1 = fminf(fmaxf(y1, 0.f), p.clip_y);
    x2 = fminf(fmaxf(x2, 0.f), p.clip_x); y2 = fminf(fmaxf(y2, 0.f), p.clip_y);
    ok = ok && ((x2 - x1) >= 0.f) && ((y2 - y1) >= 0.f);
    p.cand_boxes[slot * 4 + 0] = x1; p.cand_boxes[slot * 4 + 1] = y1; p.cand_boxes[slot * 4 + 2] = x2; p.cand_boxes[slot * 4 + 3] = y2;
    p.cand_scores[slot] = score;
    p.cand_level[slot] = p.level;
    p.cand_valid[slot] = ok ? 1 : 0;
  }
}

// =====================================================================================================
// K10 / K13: batched greedy NMS. sort (1 workgroup / image) -> 64x64 bitmask tiles -> wavefront scan.
// =====================================================================================================
struct NmsWs {
  float* sboxes;        // [n_img][n_slots][4] boxes in score order (+ coordinate-trick offsets when applicable)
  int32_t* sgroup;      // [n_img][n_slots]
  int32_t* sslot;       // [n_img][n_slots]
  int32_t* nvalid;      // [n_img] (+ padding)
  unsigned long long* mask;  // [n_img][n_slots][ncb]
};
__host__ __device__ inline long long align256(long long x) { return (x + 255) & ~255ll; }
inline NmsWs carve_nms_ws(void* ws, int n_img, int n_slots) {
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  NmsWs w;
  long long off = 0;
  w.sboxes = reinterpret_cast<float*>(b + off); off = align256(off + (long long)n_img * n_slots * 16);
  w.sgroup = reinterpret_cast<int32_t*>(b + off); off = align256(off + (long long)n_img * n_slots * 4);
  w.sslot = reinterpret_cast<int32_t*>(b + off); off = align256(off + (long long)n_img * n_slots * 4);
  w.nvalid = reinterpret_cast<int32_t*>(b + off); off = align256(off + (long long)n_img * 4);
  w.mask = reinterpret_cast<unsigned long long*>(b + off);
  return w;
}

constexpr int kSortThreads = 1024;
constexpr int kSortMaxRuns = 8;

// Score order of one image's candidates. The valid candidates are compacted first (block scan; the same scan counts the RUNS
// of equal group id). The RPN's candidates arrive as one run per pyramid level, each already in (score desc, slot asc) order -
// rpn_select_kernel wrote them so - and then the global order is a MERGE: an element's position is its position in its own run
// plus, for every other run, the number of elements that precede it there (a binary search; the 64-bit key (score, ~slot) is
// unique, so there are no ties to break). That is ~40 dependent LDS reads per element instead of the 91 barrier-separated passes
// of a bitonic sort of 8192 keys (86 us -> see profiles/). Anything else (more than kSortMaxRuns runs, or a run out of order:
// the box head's candidates) takes the bitonic sort, over the next power of two of the VALID count only.
__global__ __launch_bounds__(kSortThreads) void nms_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                                 const int32_t* __restrict__ group, const int32_t* __restrict__ valid,
                                                                 int n_slots, int n2, int trick_max_numel, NmsWs w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* s = reinterpret_cast<unsigned long long*>(smem_raw);
  __shared__ float red[kSortThreads];
  __shared__ unsigned int wsum[kSortThreads / 64];
  __shared__ unsigned int sh_carry;               // (runs << 16 | valid candidates) before the current slice of 1024 slots
  __shared__ int seg_start[kSortMaxRuns + 1];     // first compacted position of run r
  __shared__ int sh_unsorted;
  const int img = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const long long base = (long long)img * n_slots;
  for (int i = tid; i < n2; i += kSortThreads) s[i] = 0ull;    // padding / invalid: sorts last
  if (tid == 0) { sh_carry = 0u; sh_unsorted = 0; }
  __syncthreads();
  float mx = -INFINITY;
  for (int i0 = 0; i0 < n_slots; i0 += kSortThreads) {
    const int i = i0 + tid;
    const bool in = i < n_slots;
    const bool v = in && valid[base + i];
    const bool bd = in && (i == 0 || group[base + i] != group[base + i - 1]);
    const unsigned int pk = (v ? 1u : 0u) | (bd ? 0x10000u : 0u);
    unsigned int incl = pk;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned int before = sh_carry;
    for (int ww = 0; ww < wv; ++ww) before += wsum[ww];
    const unsigned int excl = before + incl - pk;
    if (bd) {
      const unsigned int r = excl >> 16;
      if (r < (unsigned)kSortMaxRuns) seg_start[r] = (int)(excl & 0xffffu);
    }
    if (v) {
      s[excl & 0xffffu] = ((unsigned long long)f32_to_key(scores[base + i]) << 32) | (uint32_t)(~(uint32_t)i);
      const float* b = boxes + (base + i) * 4;
      mx = fmaxf(mx, fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
    }
    __syncthreads();
    if (tid == kSortThreads - 1) sh_carry = before + incl;
    __syncthreads();
  }
  red[tid] = mx;
  __syncthreads();
  for (int o = kSortThreads / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  const float max_coord = red[0];
  const int nvalid = (int)(sh_carry & 0xffffu);
  const int nruns = (int)(sh_carry >> 16);
  const bool trick = (4 * nvalid <= trick_max_numel);
  const float off_unit = max_coord + 1.0f;
  auto emit = [&](int i, unsigned long long e) __attribute__((always_inline)) {    // candidate e is the i-th in score order
    const int slot = (int)(~(uint32_t)(e & 0xffffffffull));
    const float* b = boxes + (base + slot) * 4;
    const int g = group[base + slot];
    float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
    if (trick) {
      const float o = (float)g * off_unit;  // torchvision: boxes + idxs.to(boxes) * (boxes.max() + 1)
      x1 = x1 + o; y1 = y1 + o; x2 = x2 + o; y2 = y2 + o;
    }
    float* d = w.sboxes + (base + i) * 4;
    d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2;
    w.sgroup[base + i] = trick ? 0 : g;  // with the trick, overlap across groups is impossible by construction
    w.sslot[base + i] = slot;
  };
  auto run_of = [&](int j) __attribute__((always_inline)) -> int {
    int r = 0;
    while (r + 1 < nruns && seg_start[r + 1] <= j) ++r;
    return r;
  };
  const bool few_runs = nruns <= kSortMaxRuns;
  if (few_runs) {
    if (tid == 0) seg_start[nruns] = nvalid;
    __syncthreads();
    for (int j = tid + 1; j < nvalid; j += kSortThreads)
      if (j > seg_start[run_of(j)] && s[j - 1] <= s[j]) sh_unsorted = 1;
    __syncthreads();
  }
  if (few_runs && !sh_unsorted) {
    for (int j = tid; j < nvalid; j += kSortThreads) {
      const unsigned long long e = s[j];
      const int r = run_of(j);
      int rank = j - seg_start[r];
      for (int b = 0; b < nruns; ++b) {
        if (b == r) continue;
        int lo = seg_start[b], hi = seg_start[b + 1];     // first position of run b whose key is below e
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s[mid] > e) lo = mid + 1; else hi = mid;
        }
        rank += lo - seg_start[b];
      }
      emit(rank, e);
    }
  } else {
    int n2v = 2;
    while (n2v < nvalid) n2v <<= 1;     // <= n2: the compacted keys fill [0, nvalid), zeros behind them
    bitonic_sort_desc(s, n2v, tid, kSortThreads);
    for (int i = tid; i < nvalid; i += kSortThreads) emit(i, s[i]);
  }
  for (int i = nvalid + tid; i < n_slots; i += kSortThreads) w.sslot[base + i] = -1;
  if (tid == 0) w.nvalid[img] = nvalid;
}

// 64 x 64 tiles of the suppression matrix (row i, column j > i, same group, IoU > thr), one single-wave workgroup per tile. (Tried:
// four tiles per workgroup - the RPN's launch has 50 k workgroups, half of them below the diagonal - and deciding the pairs far from
// the threshold without the IEEE division: 64 -> 67 and 74 us. The launch is bound by the VALU work of its 93 M pairs.)
__global__ __launch_bounds__(64) void nms_mask_kernel(int n_slots, int ncb, float thr, NmsWs w) {
  const int cb = blockIdx.x, rb = blockIdx.y, img = blockIdx.z;
  if (cb < rb) return;
  const int nvalid = w.nvalid[img];
  if (rb * 64 >= nvalid) return;
  const long long base = (long long)img * n_slots;
  __shared__ float cbx[64][4];
  __shared__ int cgrp[64];
  const int t = threadIdx.x;
  {
    const int j = cb * 64 + t;
    if (j < nvalid) {
      const float* b = w.sboxes + (base + j) * 4;
      cbx[t][0] = b[0]; cbx[t][1] = b[1]; cbx[t][2] = b[2]; cbx[t][3] = b[3];
      cgrp[t] = w.sgroup[base + j];
    }
  }
  __syncthreads();
  const int i = rb * 64 + t;
  unsigned long long bits = 0ull;
  if (i < nvalid) {
    const float* b = w.sboxes + (base + i) * 4;
    const float ix1 = b[0], iy1 = b[1], ix2 = b[2], iy2 = b[3];
    const int ig = w.sgroup[base + i];
    const float iarea = (ix2 - ix1) * (iy2 - iy1);
    const int jmax = min(64, nvalid - cb * 64);
    for (int jj = 0; jj < jmax; ++jj) {
      const int j = cb * 64 + jj;
      if (j <= i || cgrp[jj] != ig) continue;
      const float jx1 = cbx[jj][0], jy1 = cbx[jj][1], jx2 = cbx[jj][2], jy2 = cbx[jj][3];
      const float jarea = (jx2 - jx1) * (jy2 - jy1);
      const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1), xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
      const float ww = fmaxf(0.f, xx2 - xx1), hh = fmaxf(0.f, yy2 - yy1);
      const float inter = ww * hh;
      const float ovr = inter / (iarea + jarea - inter);
      if (ovr > thr) bits |= (1ull << jj);
    }
    w.mask[(base + i) * ncb + cb] = bits;
  }
}

constexpr int kScanThreads = 256;

__global__ __launch_bounds__(kScanThreads) void nms_scan_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int n_slots,
                                                                 int ncb, int max_out, NmsWs w, float* __restrict__ out_boxes,
                                                                 float* __restrict__ out_scores, int32_t* __restrict__ out_index,
                                                                 int32_t* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* remv = reinterpret_cast<unsigned long long*>(smem_raw);  // [ncb]
  unsigned long long* kmask = remv + ncb;                                       // [ncb] kept rows of a chunk
  int* kbase = reinterpret_cast<int*>(kmask + ncb);                             // [ncb] rows kept before the chunk
  __shared__ unsigned long long sh_km;
  int n_done = 0;
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long base = (long long)img * n_slots;
  const int nvalid = w.nvalid[img];
  for (int i = tid; i < ncb; i += kScanThreads) remv[i] = 0ull;
  __syncthreads();
  int kept = 0;
  const int nchunks = (nvalid + 63) / 64;
  // The chunks form one dependent chain (diagonal word -> resolve -> rows of the kept boxes -> next chunk's removal word); every
  // global round trip in it is exposed. Wave 0 walks the chain; the diagonal word of the NEXT chunk is fetched before this chunk's
  // resolve, and the OR of the kept rows into the later removal words is spread over all four waves so that it is ONE round trip
  // per chunk (thread = (word, half of the chunk's rows): at most 32 independent loads each), not one per 32 kept rows per word.
  unsigned long long D_next = (wave == 0 && lane < nvalid) ? w.mask[(base + lane) * ncb] : 0ull;
  for (int c = 0; c < nchunks && kept < max_out; ++c) {
    if (wave == 0) {
      const int row = c * 64 + lane;
      const unsigned long long D = D_next;
      if (c + 1 < nchunks) D_next = (row + 64 < nvalid) ? w.mask[(base + row + 64) * ncb + c + 1] : 0ull;
      // Greedy resolve of the chunk on the SCALAR unit: the removal word is wave-uniform (readfirstlane tells the compiler so),
      // and only the rows that are still alive are visited - the next kept row is the lowest set bit of `alive`, its mask word
      // comes from v_readlane with a scalar lane index. (A 64-step loop on 64-bit VALU values cost ~3 us per chunk.)
      const unsigned long long cur_v = remv[c];
      unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur_v >> 32)) << 32) |
                               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur_v & 0xffffffffull));
      const int rows_here = min(64, nvalid - c * 64);
      const int d_lo = (int)(unsigned)(D & 0xffffffffull), d_hi = (int)(unsigned)(D >> 32);
      unsigned long long alive = ~cur & (rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull));
      // A row's word D holds columns ABOVE the row only (nms_mask_kernel: j > i), so a visit never clears the visiting row nor one below it:
      // a row is kept iff it is alive when its turn comes, and rows whose word is empty suppress nobody - only the others need a turn. With
      // boxes that barely overlap (64 kept rows per chunk) that is a handful of turns instead of 64.
      unsigned long long pending = alive & __ballot(D != 0ull);
      while (pending) {
        const int j = __ffsll((long long)pending) - 1;
        const unsigned long long dj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(d_hi, j) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readlane(d_lo, j);
        alive &= ~dj;                              // rows above j that j suppresses die
        pending &= alive & ~((2ull << j) - 1ull);  // ... and lose their turn; rows up to j have had theirs
      }
      const unsigned long long km = alive;
      // the kept rows are written out AFTER the chain (their two dependent loads - sort slot, then box - would sit in it)
      if (lane == 0) { kmask[c] = km; kbase[c] = kept; sh_km = km; }
    }
    __syncthreads();
    const unsigned long long km = sh_km;
    n_done = c + 1;
    kept += (int)__popcll(km);
    // OR the kept rows' masks into the removal words of the later chunks
    if (c + 1 < nchunks && kept < max_out) {
      const int n_items = (ncb - (c + 1)) * 2;
      for (int item = tid; item < n_items; item += kScanThreads) {
        const int wd = c + 1 + (item >> 1), half = item & 1;
        unsigned int m = half ? (unsigned int)(km >> 32) : (unsigned int)(km & 0xffffffffull);
        const unsigned long long* col = w.mask + (base + (long long)c * 64 + half * 32) * ncb + wd;
        unsigned long long v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          v[u] = 0ull;
          if (m) {
            const int j = __ffs((int)m) - 1;
            m &= m - 1;
            v[u] = col[(long long)j * ncb];
          }
        }
        unsigned long long acc = 0ull;
#pragma unroll
        for (int u = 0; u < 32; ++u) acc |= v[u];
        if (acc) atomicOr(&remv[wd], acc);
      }
    }
    __syncthreads();
  }
  // emit the kept rows of every processed chunk in order: position = kept rows before the chunk + kept rows before me in it
  for (int c = wave; c < n_done; c += kScanThreads / 64) {
    const int row = c * 64 + lane;
    const unsigned long long km = kmask[c];
    const int pos = kbase[c] + (int)__popcll(km & ((1ull << lane) - 1ull));
    if (row < nvalid && ((km >> lane) & 1ull) && pos < max_out) {
      const int slot = w.sslot[base + row];
      const long long o = (long long)img * max_out + pos;
      const float* b = boxes + (base + slot) * 4;
      out_boxes[o * 4 + 0] = b[0]; out_boxes[o * 4 + 1] = b[1]; out_boxes[o * 4 + 2] = b[2]; out_boxes[o * 4 + 3] = b[3];
      out_scores[o] = scores[base + slot];
      out_index[o] = slot;
    }
  }
  if (tid == 0) out_count[img] = kept < max_out ? kept : max_out;
}

// =====================================================================================================
// K11 / K15: ROIAlign (aligned=False, legacy pixel model), NHWC, one workgroup per (roi, bin row)
// =====================================================================================================
struct RoiArgs {
  const void* feat[4];
  int Hl[4], Wl[4];
  float scale[4];
  int n_levels, min_level, C, P, sampling;
  const float* boxes;
  const int32_t* counts;
  int n_img, max_rois;
  void* out;
  int compact;
  const int32_t* roi_offsets;
  int items_per_block;
};

template <typename T>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiArgs p) {
  const int img = blockIdx.z, j = blockIdx.y;
  if (j >= p.counts[img]) {
    // padded slot of the fixed-size box-head input: zeros (finite) so that the GEMM rows behind it stay finite;
    // the compact DensePose layout has no padded rows
    if (!p.compact) {
      const int C8 = p.C >> 3;
      const int item0 = blockIdx.x * p.items_per_block;
      const int item1 = min(p.P * p.P * C8, item0 + p.items_per_block);
      T* __restrict__ out = reinterpret_cast<T*>(p.out) + ((long long)img * p.max_rois + j) * (long long)p.P * p.P * p.C;
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int item = item0 + threadIdx.x; item < item1; item += blockDim.x) store8(out + (long long)item * 8, z);
    }
    return;
  }
  const float* b = p.boxes + ((long long)img * p.max_rois + j) * 4;
  const float bx1 = b[0], by1 = b[1], bx2 = b[2], by2 = b[3];
  int lvl = 0;
  if (p.n_levels > 1) {
    // poolers.py:43-51
    const float area = (bx2 - bx1) * (by2 - by1);
    const float sz = sqrtf(area);
    float lv = floorf(4.0f + log2f(sz / 224.0f + 1e-8f));
    const float lo = (float)p.min_level, hi = (float)(p.min_level + p.n_levels - 1);
    lv = fminf(fmaxf(lv, lo), hi);   // NaN (negative area) -> fmaxf returns lo, torch.clamp would give NaN->int; never happens (w,h >= 0)
    lvl = (int)lv - p.min_level;
  }
  const int H = p.Hl[lvl], W = p.Wl[lvl], C = p.C, P = p.P, g = p.sampling;
  const float scale = p.scale[lvl];
  const T* __restrict__ feat = reinterpret_cast<const T*>(p.feat[lvl]) + (long long)img * H * W * C;
  const float rsw = bx1 * scale, rsh = by1 * scale, rew = bx2 * scale, reh = by2 * scale;
  const float roi_w = fmaxf(rew - rsw, 1.0f), roi_h = fmaxf(reh - rsh, 1.0f);
  const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
  const float count = (float)(g * g > 1 ? g * g : 1);
  const long long orow = p.compact ? (long long)(p.roi_offsets[img] + j) : ((long long)img * p.max_rois + j);
  T* __restrict__ out = reinterpret_cast<T*>(p.out) + orow * (long long)P * P * C;
  // work item = (bin column px, 8-channel group): 16-byte (bf16) / 32-byte (fp32) vector per corner, so the 7 bins x 32
  // groups of a box-head row fill one 256-thread pass and every corner read is a full 16-byte lane access
  // one workgroup per (ROI, slab of bins): ~2048 (bin, 8-channel group) items per workgroup - enough work per thread
  // to amortise the block's start-up and keep several independent corner loads in flight, while a 28x28 DensePose
  // pooling of a handful of ROIs still spreads over the whole chip
  const int C8 = C >> 3;
  const int item0 = blockIdx.x * p.items_per_block;
  const int item1 = min(P * P * C8, item0 + p.items_per_block);
  for (int item = item0 + threadIdx.x; item < item1; item += blockDim.x) {
    const int bin = item / C8, c8 = item - bin * C8;
    const int py = bin / P, px = bin - py * P;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int iy = 0; iy < g; ++iy) {
      const float yy = rsh + (float)py * bin_h + ((float)iy + 0.5f) * bin_h / (float)g;
      for (int ix = 0; ix < g; ++ix) {
        const float xx = rsw + (float)px * bin_w + ((float)ix + 0.5f) * bin_w / (float)g;
        float y = yy, x = xx;
        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;  // contributes 0
        if (y <= 0.f) y = 0.f;
        if (x <= 0.f) x = 0.f;
        int y_low = (int)y, x_low = (int)x, y_high, x_high;
        if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
        if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
        const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const T* p1 = feat + ((long long)y_low * W + x_low) * C + c8 * 8;
        const T* p2 = feat + ((long long)y_low * W + x_high) * C + c8 * 8;
        const T* p3 = feat + ((long long)y_high * W + x_low) * C + c8 * 8;
        const T* p4 = feat + ((long long)y_high * W + x_high) * C + c8 * 8;
        float v1[8], v2[8], v3[8], v4[8];
        load8(p1, v1); load8(p2, v2); load8(p3, v3); load8(p4, v4);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += w1 * v1[k] + w2 * v2[k] + w3 * v3[k] + w4 * v4[k];
      }
    }
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = acc[k] / count;
    store8(out + (long long)bin * C + c8 * 8, r);
  }
}

// ---- the same pooling with the sample geometry tabulated once per ROI (sampling ratio G known at compile time) ----
// A bin's G x G samples use G sample rows and G sample columns of the ROI's P*G x P*G sample grid; row s of that grid has ONE
// (y_low, y_high, ly, hy, inside) whatever the column and vice versa, so the workgroup's first 2*P*G threads work the
// coordinates out once (same expressions, same order as above: the weights are bit-identical) and every (bin, 8-channel)
// item only looks them up. All 4*G*G corner loads of an item are issued before the first is consumed.
template <typename T> struct RoiRaw { uint4 u; };
template <> struct RoiRaw<float> { float4 a, b; };
template <typename T>
__device__ __forceinline__ void roi_raw_load(const T* p, RoiRaw<T>& r) { r.u = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void roi_raw_load(const float* p, RoiRaw<float>& r) {
  r.a = *reinterpret_cast<const float4*>(p);
  r.b = *reinterpret_cast<const float4*>(p + 4);
}
template <typename T>
__device__ __forceinline__ void roi_raw_unpack(const RoiRaw<T>& r, float (&o)[8]) {
  o[0] = Elem<T>::unpack(r.u.x & 0xffffu); o[1] = Elem<T>::unpack(r.u.x >> 16);
  o[2] = Elem<T>::unpack(r.u.y & 0xffffu); o[3] = Elem<T>::unpack(r.u.y >> 16);
  o[4] = Elem<T>::unpack(r.u.z & 0xffffu); o[5] = Elem<T>::unpack(r.u.z >> 16);
  o[6] = Elem<T>::unpack(r.u.w & 0xffffu); o[7] = Elem<T>::unpack(r.u.w >> 16);
}
__device__ __forceinline__ void roi_raw_unpack(const RoiRaw<float>& r, float (&o)[8]) {
  o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w;
}

constexpr int kRoiTab = 64;   // P * G <= 64 sample rows / columns per ROI

template <typename T, int G>
__global__ __launch_bounds__(256) void roi_align_tab_kernel(const RoiArgs p) {
  const int img = blockIdx.z, j = blockIdx.y;
  const int C = p.C, P = p.P, C8 = C >> 3;
  const int item0 = blockIdx.x * p.items_per_block;
  const int item1 = min(P * P * C8, item0 + p.items_per_block);
  if (j >= p.counts[img]) {
    if (!p.compact) {
      T* __restrict__ out = reinterpret_cast<T*>(p.out) + ((long long)img * p.max_rois + j) * (long long)P * P * C;
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int item = item0 + threadIdx.x; item < item1; item += blockDim.x) store8(out + (long long)item * 8, z);
    }
    return;
  }
  __shared__ int t_lo[2][kRoiTab], t_hi[2][kRoiTab];      // [0]: row offsets (y * W * C), [1]: column offsets (x * C), elements
  __shared__ float t_l[2][kRoiTab], t_h[2][kRoiTab];      // ly / hy, lx / hx
  __shared__ int t_in[2][kRoiTab];
  const float* b = p.boxes + ((long long)img * p.max_rois + j) * 4;
  const float bx1 = b[0], by1 = b[1], bx2 = b[2], by2 = b[3];
  int lvl = 0;
  if (p.n_levels > 1) {
    const float area = (bx2 - bx1) * (by2 - by1);
    const float sz = sqrtf(area);
    float lv = floorf(4.0f + log2f(sz / 224.0f + 1e-8f));
    const float lo = (float)p.min_level, hi = (float)(p.min_level + p.n_levels - 1);
    lv = fminf(fmaxf(lv, lo), hi);
    lvl = (int)lv - p.min_level;
  }
  const int H = p.Hl[lvl], W = p.Wl[lvl];
  const float scale = p.scale[lvl];
  const T* __restrict__ feat = reinterpret_cast<const T*>(p.feat[lvl]) + (long long)img * H * W * C;
  {
    const int tid = threadIdx.x;
    const int axis = tid >> 6, s = tid & 63;    // wave 0: sample rows, wave 1: sample columns
    if (axis < 2 && s < P * G) {
      const float r0 = (axis == 0 ? by1 : bx1) * scale, r1 = (axis == 0 ? by2 : bx2) * scale;
      const float roi_len = fmaxf(r1 - r0, 1.0f);
      const float bin_len = roi_len / (float)P;
      const int pb = s / G, i = s - pb * G;
      const int lim = axis == 0 ? H : W;
      float v = r0 + (float)pb * bin_len + ((float)i + 0.5f) * bin_len / (float)G;
      const bool inside = !(v < -1.0f || v > (float)lim);
      if (v <= 0.f) v = 0.f;
      int lo = (int)v, hi;
      if (!inside) { lo = 0; v = 0.f; }           // never contributes; keeps the (unused) corner loads in bounds
      if (lo >= lim - 1) { hi = lo = lim - 1; v = (float)lo; } else { hi = lo + 1; }
      const float l = v - (float)lo;
      const int unit = axis == 0 ? W * C : C;
      t_lo[axis][s] = lo * unit; t_hi[axis][s] = hi * unit;
      t_l[axis][s] = l; t_h[axis][s] = 1.f - l;
      t_in[axis][s] = inside ? 1 : 0;
    }
  }
  __syncthreads();
  const float count = (float)(G * G > 1 ? G * G : 1);
  const long long orow = p.compact ? (long long)(p.roi_offsets[img] + j) : ((long long)img * p.max_rois + j);
  T* __restrict__ out = reinterpret_cast<T*>(p.out) + orow * (long long)P * P * C;
  for (int item = item0 + threadIdx.x; item < item1; item += blockDim.x) {
    const int bin = item / C8, c8 = item - bin * C8;
    const int py = bin / P, px = bin - py * P;
    const T* base = feat + c8 * 8;
    RoiRaw<T> raw[G * G][4];
#pragma unroll
    for (int iy = 0; iy < G; ++iy) {
      const int ylo = t_lo[0][py * G + iy], yhi = t_hi[0][py * G + iy];
#pragma unroll
      for (int ix = 0; ix < G; ++ix) {
        const int xlo = t_lo[1][px * G + ix], xhi = t_hi[1][px * G + ix];
        roi_raw_load(base + ylo + xlo, raw[iy * G + ix][0]);
        roi_raw_load(base + ylo + xhi, raw[iy * G + ix][1]);
        roi_raw_load(base + yhi + xlo, raw[iy * G + ix][2]);
        roi_raw_load(base + yhi + xhi, raw[iy * G + ix][3]);
      }
    }
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int iy = 0; iy < G; ++iy) {
      const float ly = t_l[0][py * G + iy], hy = t_h[0][py * G + iy];
      const int in_y = t_in[0][py * G + iy];
#pragma unroll
      for (int ix = 0; ix < G; ++ix) {
        const float lx = t_l[1][px * G + ix], hx = t_h[1][px * G + ix];
        if (!(in_y && t_in[1][px * G + ix])) continue;   // contributes 0
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        float v1[8], v2[8], v3[8], v4[8];
        roi_raw_unpack(raw[iy * G + ix][0], v1); roi_raw_unpack(raw[iy * G + ix][1], v2);
        roi_raw_unpack(raw[iy * G + ix][2], v3); roi_raw_unpack(raw[iy * G + ix][3], v4);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += w1 * v1[k] + w2 * v2[k] + w3 * v3[k] + w4 * v4[k];
      }
    }
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = acc[k] / count;
    store8(out + (long long)bin * C + c8 * 8, r);
  }
}

// =====================================================================================================
// K13 (first half): softmax + apply_deltas + finite filter + score threshold
// =====================================================================================================
__global__ void box_decode_kernel(const dp_box_decode_params p) {
  const int total = p.n_img * p.max_rois;
  const float clampv = 4.135166556742356f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int img = i / p.max_rois, j = i - img * p.max_rois;
    bool ok = j < p.prop_counts[img];
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, p0 = 0.f;
    if (ok) {
      const float* l = p.logits + (long long)i * p.ld;
      const float l0 = l[0], l1 = l[1];
      const float m = fmaxf(l0, l1);
      const float e0 = expf(l0 - m), e1 = expf(l1 - m);
      const float sum = e0 + e1;
      p0 = e0 / sum;
      const float p1 = e1 / sum;
      const float* b = p.prop_boxes + (long long)i * 4;
      const float widths = b[2] - b[0], heights = b[3] - b[1];
      const float ctr_x = b[0] + 0.5f * widths, ctr_y = b[1] + 0.5f * heights;
      const float dx = l[2] / p.wx, dy = l[3] / p.wy;
      const float dw = fminf(l[4] / p.ww, clampv), dh = fminf(l[5] / p.wh, clampv);
      const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
      const float pw = expf(dw) * widths, ph = expf(dh) * heights;
      x1 = pcx - 0.5f * pw; y1 = pcy - 0.5f * ph; x2 = pcx + 0.5f * pw; y2 = pcy + 0.5f * ph;
      ok = finitef(x1) && finitef(y1) && finitef(x2) && finitef(y2) && finitef(p0) && finitef(p1) && (l[4] == l[4]) && (l[5] == l[5]);
      ok = ok && (p0 > p.score_thresh);  // Q2: boxes are NOT clipped here
    }
    p.cand_boxes[(long long)i * 4 + 0] = x1; p.cand_boxes[(long long)i * 4 + 1] = y1;
    p.cand_boxes[(long long)i * 4 + 2] = x2; p.cand_boxes[(long long)i * 4 + 3] = y2;
    p.cand_scores[i] = p0;
    p.cand_group[i] = 0;
    p.cand_valid[i] = ok ? 1 : 0;
  }
}

__global__ void postprocess_kernel(const dp_postprocess_params p) {
  const int total = p.n_img * p.max_dets;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int img = i / p.max_dets, j = i - img * p.max_dets;
    int keep = 0;
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
    if (j < p.counts[img]) {
      const float sx = p.scale_xy[img * 2 + 0], sy = p.scale_xy[img * 2 + 1];
      const float H = p.out_hw[img * 2 + 0], W = p.out_hw[img * 2 + 1];
      const float* b = p.boxes + (long long)i * 4;
      x1 = b[0] * sx; y1 = b[1] * sy; x2 = b[2] * sx; y2 = b[3] * sy;
      keep = ((x2 - x1) >= 0.f && (y2 - y1) >= 0.f) ? 1 : 0;
      x1 = fminf(fmaxf(x1, 0.f), W); y1 = fminf(fmaxf(y1, 0.f), H);
      x2 = fminf(fmaxf(x2, 0.f), W); y2 = fminf(fmaxf(y2, 0.f), H);
    }
    p.out_boxes[(long long)i * 4 + 0] = x1; p.out_boxes[(long long)i * 4 + 1] = y1;
    p.out_boxes[(long long)i * 4 + 2] = x2; p.out_boxes[(long long)i * 4 + 3] = y2;
    p.keep[i] = keep;
  }
}

inline int next_pow2(int v) {
  int n = 1;
  while (n < v) n <<= 1;
  return n;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
extern "C" int64_t dp_rpn_topk_workspace_bytes(int n_img, int Hi, int Wi, int A) {
  // compact sortable keys of one level (single-stage path) or the (key, index) candidates of the chunked path;
  // the candidate list is never larger than the key list (kcap <= chunk length), so one bound covers both
  return (int64_t)n_img * Hi * Wi * A * 8 + 256;
}

namespace {
// validates one level, fills its slot of the stage-1 launch (key extraction / chunk pre-selection) and the select-kernel arguments
int rpn_prepare_level(const dp_rpn_level_params* p, RpnSelArgs& a, RpnPrepLevel& q, int& n2) {
  DP_REQUIRE(p->head && p->cand_boxes && p->cand_scores && p->cand_level && p->cand_valid && p->workspace, "dp_rpn_topk_decode: null pointer");
  DP_REQUIRE(p->n_img > 0 && p->Hi > 0 && p->Wi > 0 && p->A > 0 && p->A <= 3 && p->head_c >= 5 * p->A, "dp_rpn_topk_decode: bad shape");
  DP_REQUIRE(p->kmax > 0 && p->kmax <= 4096, "dp_rpn_topk_decode: kmax=%d outside (0, 4096]", p->kmax);
  DP_REQUIRE(p->slot_off >= 0 && p->slot_off + p->kmax <= p->slots_per_img, "dp_rpn_topk_decode: slot range");
  DP_REQUIRE(p->n_img <= 65535, "dp_rpn_topk_decode: n_img");
  const long long n = (long long)p->Hi * p->Wi * p->A;
  DP_REQUIRE(n < (1ll << 30), "dp_rpn_topk_decode: level too large");
  a.head = p->head; a.n_img = p->n_img; a.Hi = p->Hi; a.Wi = p->Wi; a.A = p->A; a.head_c = p->head_c;
  a.stride_px = p->stride_px; a.level = p->level; a.kmax = p->kmax; a.slot_off = p->slot_off; a.slots_per_img = p->slots_per_img;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) a.ca[i][j] = p->cell_anchors[i][j];
  a.clip_x = p->clip_x; a.clip_y = p->clip_y;
  a.cand_boxes = p->cand_boxes; a.cand_scores = p->cand_scores; a.cand_level = p->cand_level; a.cand_valid = p->cand_valid;
  uint32_t* ws = reinterpret_cast<uint32_t*>(p->workspace);
  q.head = p->head; q.cells = p->Hi * p->Wi; q.A = p->A; q.head_c = p->head_c; q.n = (int)n; q.k = p->kmax;
  if (n >= 2 * kChunk) {
    // two-stage: per-chunk pre-selection (many workgroups), then one workgroup per image over the candidates
    const int n_chunks = (int)((n + kChunk - 1) / kChunk);
    const int kcap = p->kmax < kChunk ? p->kmax : kChunk;
    q.chunked = 1; q.n_chunks = n_chunks; q.kcap = kcap;
    q.out_keys = ws; q.out_idx = ws + (long long)p->n_img * n_chunks * kcap;
    a.keys = q.out_keys; a.kidx = q.out_idx; a.n_keys = n_chunks * kcap;
  } else {
    q.chunked = 0; q.n_chunks = (int)((n + kChunk - 1) / kChunk); q.kcap = 0;
    q.out_keys = ws; q.out_idx = nullptr;
    a.keys = ws; a.kidx = nullptr; a.n_keys = (int)n;
  }
  const int k = (int)(n < p->kmax ? n : p->kmax);
  n2 = next_pow2(k);
  return DP_OK;
}
}  // namespace

extern "C" int dp_rpn_topk_decode_levels(const dp_rpn_level_params* levels, int n_levels, dp_stream_t stream) {
  DP_REQUIRE(levels && n_levels >= 1 && n_levels <= 5, "dp_rpn_topk_decode_levels: 1..5 levels");
  hipStream_t s = as_stream(stream);
  RpnSelMulti m;
  RpnPrepMulti pm;
  int n2max = 1, n_blocks = 0;
  for (int l = 0; l < n_levels; ++l) {
    DP_REQUIRE(levels[l].n_img == levels[0].n_img, "dp_rpn_topk_decode_levels: all levels must share n_img");
    int n2 = 1;
    const int rc = rpn_prepare_level(&levels[l], m.lv[l], pm.lv[l], n2);
    if (rc != DP_OK) return rc;
    if (n2 > n2max) n2max = n2;
    pm.first_block[l] = n_blocks;
    n_blocks += pm.lv[l].n_chunks;
  }
  pm.first_block[n_levels] = n_blocks;
  pm.n_levels = n_levels;
  hipLaunchKernelGGL(rpn_prep_kernel, dim3(n_blocks, levels[0].n_img), dim3(kChunkThreads), 0, s, pm);
  m.n2max = n2max;
  m.keys_cap = 25600;      // 100 KB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rpn_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 8 + 25600 * 4);
    attr_set = true;
  }
  hipLaunchKernelGGL(rpn_select_kernel, dim3(levels[0].n_img, n_levels), dim3(kSelThreads), n2max * 8 + m.keys_cap * 4, s, m);
  return dp_check_launch("rpn_select_kernel");
}

extern "C" int dp_rpn_topk_decode(const dp_rpn_level_params* p, dp_stream_t stream) {
  DP_REQUIRE(p, "dp_rpn_topk_decode: null params");
  return dp_rpn_topk_decode_levels(p, 1, stream);
}

extern "C" int64_t dp_nms_workspace_bytes(int n_img, int n_slots) {
  const long long ncb = (n_slots + 63) / 64;
  long long off = 0;
  off = align256(off + (long long)n_img * n_slots * 16);
  off = align256(off + (long long)n_img * n_slots * 4);
  off = align256(off + (long long)n_img * n_slots * 4);
  off = align256(off + (long long)n_img * 4);
  off = align256(off + (long long)n_img * n_slots * ncb * 8);
  return off;
}

extern "C" int dp_batched_nms(const dp_nms_params* p, dp_stream_t stream) {
  DP_REQUIRE(p, "dp_batched_nms: null params");
  DP_REQUIRE(p->boxes && p->scores && p->group && p->valid && p->out_boxes && p->out_scores && p->out_index && p->out_count && p->workspace,
             "dp_batched_nms: null pointer");
  DP_REQUIRE(p->n_img > 0 && p->n_slots > 0 && p->n_slots <= 8192, "dp_batched_nms: n_slots=%d outside (0, 8192]", p->n_slots);
  DP_REQUIRE(p->max_out > 0, "dp_batched_nms: max_out");
  hipStream_t s = as_stream(stream);
  NmsWs w = carve_nms_ws(p->workspace, p->n_img, p->n_slots);
  const int n2 = next_pow2(p->n_slots);
  const int ncb = (p->n_slots + 63) / 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_sort_kernel, dim3(p->n_img), dim3(kSortThreads), n2 * 8, s, p->boxes, p->scores, p->group, p->valid, p->n_slots, n2,
                     p->trick_max_numel, w);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(ncb, ncb, p->n_img), dim3(64), 0, s, p->n_slots, ncb, p->iou_thr, w);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(p->n_img), dim3(kScanThreads), ncb * 20, s, p->boxes, p->scores, p->n_slots, ncb, p->max_out, w, p->out_boxes,
                     p->out_scores, p->out_index, p->out_count);
  return dp_check_launch("nms kernels");
}

extern "C" int dp_roi_align_nhwc(const dp_roi_align_params* p, dp_stream_t stream) {
  DP_REQUIRE(p, "dp_roi_align_nhwc: null params");
  DP_REQUIRE(p->boxes && p->counts && p->out, "dp_roi_align_nhwc: null pointer");
  DP_REQUIRE(p->n_levels == 1 || p->n_levels == 4, "dp_roi_align_nhwc: n_levels=%d", p->n_levels);
  DP_REQUIRE(p->n_img > 0 && p->max_rois > 0 && p->C > 0 && p->C % 8 == 0 && p->P > 0 && p->sampling > 0, "dp_roi_align_nhwc: bad shape");
  DP_REQUIRE(!p->compact || p->roi_offsets, "dp_roi_align_nhwc: compact mode needs roi_offsets");
  RoiArgs a;
  for (int i = 0; i < 4; ++i) {
    a.feat[i] = i < p->n_levels ? p->feat[i] : nullptr;
    a.Hl[i] = p->Hl[i]; a.Wl[i] = p->Wl[i]; a.scale[i] = p->scale[i];
    if (i < p->n_levels) DP_REQUIRE(a.feat[i] && a.Hl[i] > 0 && a.Wl[i] > 0, "dp_roi_align_nhwc: level %d map", i);
  }
  a.n_levels = p->n_levels; a.min_level = p->min_level; a.C = p->C; a.P = p->P; a.sampling = p->sampling;
  a.boxes = p->boxes; a.counts = p->counts; a.n_img = p->n_img; a.max_rois = p->max_rois; a.out = p->out;
  a.compact = p->compact; a.roi_offsets = p->roi_offsets;
  hipStream_t s = as_stream(stream);
  const int C8 = p->C / 8;
  const int bins_per_block = 2048 / C8 > 0 ? 2048 / C8 : 1;
  a.items_per_block = bins_per_block * C8;
  const dim3 grid((p->P * p->P + bins_per_block - 1) / bins_per_block, p->max_rois, p->n_img), block(256);
  // policy key roi_tab = 0: the per-sample kernel for every sampling ratio (A/B: 288 -> 193 us for the box head's 8 x 1000 ROIs, + 2.3 % images/s).
  // Pinning image i's ROIs to XCD i % 8 on top (one L2 per feature map) cut the bytes fetched beyond L2 by 7 - 15 % and the time by
  // nothing: the kernel is bound by L2 -> L1 requests (3.2 GB of corner reads per launch), not by what L2 misses.
  const bool tab_mode = dp_policy().roi_tab != 0;
  if (tab_mode && p->sampling == 2 && p->P * 2 <= kRoiTab) {
    if (p->dtype == DP_F32) hipLaunchKernelGGL((roi_align_tab_kernel<float, 2>), grid, block, 0, s, a);
    else if (p->dtype == DP_BF16) hipLaunchKernelGGL((roi_align_tab_kernel<uint16_t, 2>), grid, block, 0, s, a);
    else if (p->dtype == DP_F16) hipLaunchKernelGGL((roi_align_tab_kernel<f16_t, 2>), grid, block, 0, s, a);
    else return dp_fail(DP_ERR_BAD_ARG, "dp_roi_align_nhwc: bad dtype");
    return dp_check_launch("roi_align_tab_kernel");
  }
  if (p->dtype == DP_F32) hipLaunchKernelGGL(roi_align_kernel<float>, grid, block, 0, s, a);
  else if (p->dtype == DP_BF16) hipLaunchKernelGGL(roi_align_kernel<uint16_t>, grid, block, 0, s, a);
  else if (p->dtype == DP_F16) hipLaunchKernelGGL(roi_align_kernel<f16_t>, grid, block, 0, s, a);
  else return dp_fail(DP_ERR_BAD_ARG, "dp_roi_align_nhwc: bad dtype");
  return dp_check_launch("roi_align_kernel");
}

extern "C" int dp_box_decode_score(const dp_box_decode_params* p, dp_stream_t stream) {
  DP_REQUIRE(p, "dp_box_decode_score: null params");
  DP_REQUIRE(p->logits && p->prop_boxes && p->prop_counts && p->cand_boxes && p->cand_scores && p->cand_group && p->cand_valid,
             "dp_box_decode_score: null pointer");
  DP_REQUIRE(p->n_img > 0 && p->max_rois > 0 && p->ld >= 6, "dp_box_decode_score: bad shape");
  const int total = p->n_img * p->max_rois;
  hipLaunchKernelGGL(box_decode_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), *p);
  return dp_check_launch("box_decode_kernel");
}

extern "C" int dp_postprocess_boxes(const dp_postprocess_params* p, dp_stream_t stream) {
  DP_REQUIRE(p, "dp_postprocess_boxes: null params");
  DP_REQUIRE(p->boxes && p->counts && p->scale_xy && p->out_hw && p->out_boxes && p->keep, "dp_postprocess_boxes: null pointer");
  DP_REQUIRE(p->n_img > 0 && p->max_dets > 0, "dp_postprocess_boxes: bad shape");
  const int total = p->n_img * p->max_dets;
  hipLaunchKernelGGL(postprocess_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), *p);
  return dp_check_launch("postprocess_kernel");
}
